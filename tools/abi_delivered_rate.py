"""Reference-ABI tier (gen_and_load_hip_code -> init_frame/init_slice/calc_lines), host buffer to host buffer: time per
frame of the Mandelbrot example at 8192 x 8192 into the host's pageable buffer `q`, from the difference between a run
of 1 + N frames and a run of 1 frame.  GPU box only."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import filters as F  # noqa: E402
from mathmap_amd._lib import selftest_lib  # noqa: E402


def run(src, w, h, warm, bands, out):
    os.environ["MMHIP_SELFTEST_WARM_FRAMES"] = str(warm)
    t0 = time.perf_counter()
    rc = selftest_lib().mmhip_selftest_abi_roundtrip(src.encode(), 1, None, 0, 0, 0, w, h, 0.25, bands,
                                                     out.ctypes.data_as(C.c_void_p))
    dt = time.perf_counter() - t0
    assert rc == 0, selftest_lib().mmhip_selftest_error()
    return dt


def main():
    w = h = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    src = F.ir_text("mandelbrot")
    out = np.zeros((h, w, 4), np.uint8)
    out[:] = 1          # touch the pages
    res = {"size": [w, h], "frames": n}
    for bands in (1, 8):
        run(src, w, h, 0, bands, out)
        a = min(run(src, w, h, 0, bands, out) for _ in range(2))
        b = min(run(src, w, h, n, bands, out) for _ in range(2))
        per = (b - a) / n
        res["bands_%d" % bands] = {"ms_per_frame": per * 1e3, "Mpixels_per_s": w * h / per / 1e6,
                                   "GB_per_s": w * h * 4 / per / 1e9}
    print(json.dumps(res))


if __name__ == "__main__":
    main()

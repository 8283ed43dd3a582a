#!/usr/bin/env python3
"""Every example filter of the reference (from the IR fixtures) at 2048^2 with the per-row slice on and off
(MMHIP_NO_ROW_SLICE): kernel ms, to see where the rows kernel is used and that it never costs time.
usage (GPU box): python tools/examples_row_slice_sweep.py > gpurun_out/row_slice_sweep.txt"""
import gzip
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import mathmap_amd as mm  # noqa: E402
from mathmap_amd._lib import lib  # noqa: E402
from mathmap_amd.api import UV_IMAGE  # noqa: E402
from tests import filters as F  # noqa: E402


def timed(ir, size, img):
    flt = mm.Filter(ir_json=ir)
    inv = flt.invoke(size, size)
    for u in flt.uservals:
        if u["kind"] == UV_IMAGE:
            inv.set_image(u["name"], img)
    dev = lib().mmhip_device_alloc(size * size * 4)
    inv.enable_timing(True)
    try:
        for i in range(7):
            inv.render_rows(dev, 0, size, t=0.25)
        ts = sorted(inv.drain_kernel_ms()[2:])
    finally:
        lib().mmhip_device_free(dev)
    return ts[len(ts) // 2], "mm_rows(mm_args" in flt.kernel_source


def main():
    size = 2048
    img = F.synthetic_image(512, 512, seed=2)
    d = os.path.join(ROOT, "tests", "golden", "ir_examples")
    names = json.load(open(os.path.join(d, "manifest.json")))
    used = 0
    for n in names:
        ir = gzip.open(os.path.join(d, n + ".json.gz"), "rt").read()
        try:
            os.environ.pop("MMHIP_NO_ROW_SLICE", None)
            on, has = timed(ir, size, img)
            if not has:
                continue
            os.environ["MMHIP_NO_ROW_SLICE"] = "1"
            off, _ = timed(ir, size, img)
        except mm.MathMapError as e:
            print("%-48s error: %s" % (n, str(e).splitlines()[0][:80]))
            continue
        used += 1
        print("%-48s rows kernel %.4f ms   per pixel %.4f ms   %+.1f %%" % (n, on, off, (on / off - 1) * 100), flush=True)
    print("%d of %d example filters have a per-row slice" % (used, len(names)))


if __name__ == "__main__":
    main()

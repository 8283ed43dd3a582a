#!/usr/bin/env python3
"""Differential fuzz of the render options of the boundary (GPU box): the generated filters of tests/fuzz_filters.py
rendered through mmhip_render with a random region (new_template.c.in:238-262: region_x / region_y / region_width /
region_height), random row bands inside it, output_bpp 1-4 or the float-map output (new_template.c.in:279-305), a row
stride with padding, and two times -- against the oracle's full frame of the same output kind, cropped.
usage: fuzz_options.py [lo hi]"""
import ctypes as C
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import mathmap_amd as mm  # noqa: E402
from mathmap_amd._lib import lib  # noqa: E402
from tests import filters as F  # noqa: E402
from oracle.ccgen import CpuFilter  # noqa: E402
from fuzz_filters import make_filter, make_filter_ex  # noqa: E402


def main():
    import faulthandler
    faulthandler.enable()
    lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 150)
    w, h = 112, 72
    imgs = {"in": F.synthetic_image(w, h, seed=1), "in2": F.synthetic_image(50, 70, seed=2)}
    bad, ran = [], 0
    for seed in range(lo, hi):
        r = random.Random(seed)
        if seed % 2:
            src, needs = make_filter(seed)
            names, opts = (["in"] if needs else []), {}
        else:
            src, names, opts = make_filter_ex(seed)
        uv = {"k": seed % 7, "m": 0.3 + (seed % 5) * 0.4}
        floatmap = r.random() < 0.3
        bpp = 4 if floatmap else r.choice([1, 2, 3, 4, 4])
        rw, rh = r.randint(1, w), r.randint(1, h)
        rx, ry = r.randint(0, w - rw), r.randint(0, h - rh)
        cuts = sorted({0, rh} | {r.randint(0, rh) for _ in range(r.randint(0, 3))})
        px = 16 if floatmap else bpp
        stride = rw * px + r.choice([0, 0, 4 * px, 64]) if not floatmap else w * 16      # (a float map's rows are the frame's render width apart, new_template.c.in:297)
        try:
            flt = mm.Filter(src, specialize=bool(seed % 3 == 0), **opts)
            inv = flt.invoke(w, h)
            for k, v in uv.items():
                inv.set(k, v)
            for n in names:
                inv.set_image(n, imgs[n])
            cf = CpuFilter(flt.ir_json_raw)
            for t in (0.4, 0.9):
                nbytes = stride * rh
                dev = lib().mmhip_device_alloc(nbytes)
                host = np.zeros(nbytes, np.uint8)
                try:
                    for a, b in zip(cuts[:-1], cuts[1:]):
                        if a == b:
                            continue
                        inv.render_rows(dev + a * stride, ry + a, ry + b, t=t, row_stride=stride, bpp=bpp, floatmap=floatmap,
                                        region=(rx, ry, rw, rh))
                    inv.sync()
                    assert lib().mmhip_copy_to_host(host.ctypes.data_as(C.c_void_p), C.c_void_p(dev), nbytes) == 0
                finally:
                    lib().mmhip_device_free(C.c_void_p(dev))
                rows = host.reshape(rh, stride)[:, :rw * px]
                got = rows.reshape(rh, rw, px).copy().view(np.float32).reshape(rh, rw, 4) if floatmap else rows.reshape(rh, rw, bpp)
                want = cf.render(w, h, uservals=uv, images={n: imgs[n] for n in names}, t=t, bpp=bpp, floatmap=floatmap,
                                 intersample=opts.get("intersample", True), edge=(opts.get("edge_x", 0), opts.get("edge_y", 0)))
                want = want[ry:ry + rh, rx:rx + rw]
                if floatmap:
                    same = np.array_equal(got.view(np.uint32), want.view(np.uint32)) or np.allclose(got, want, rtol=2e-6, atol=1e-7, equal_nan=True)
                    if not same:
                        bad.append((seed, "floatmap", t, float(np.nanmax(np.abs(got - want)))))
                else:
                    d = np.abs(got.astype(int) - want.astype(int))
                    if d.max() > 1 or (d > 0).sum() > 0.01 * d.size:
                        bad.append((seed, "bpp%d" % bpp, t, int(d.max()), int((d > 0).sum()), (rx, ry, rw, rh)))
            ran += 1
        except Exception as e:
            bad.append((seed, "error", str(e).splitlines()[0][:160]))
        if seed % 25 == 0:
            print("seed", seed, "ran", ran, "bad so far", len(bad), flush=True)
    print("filters:", ran, "bad:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python3
"""Long differential run of the fetching kernels (tests/fuzz_filters.py make_filter / make_filter_ex): HIP vs the
oracle for seeds [lo, hi), each filter in three kernel shapes -- the generic kernel, the user-value specialised one
and the one-pixel shape with the early-exit fetch (MMHIP_SINGLE_PIXEL=1) -- with an input smaller than the frame (so
that a good part of every frame samples outside the image), non-trivial edge colours and a frame size that is not a
multiple of any tile.  Prints every case that is not bit-exact.  usage: fuzz_fetch.py [lo hi]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mathmap_amd as mm
from tests import filters as F
from oracle.ccgen import CpuFilter
from tests.fuzz_filters import make_filter, make_filter_ex


def main():
    lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100, 400)
    w, h = 203, 131
    colors = (0x20406080, 0xC0A01055)
    imgs = {"in": F.synthetic_image(150, 97, seed=1), "in2": F.synthetic_image(50, 70, seed=2)}
    inexact, errors, kernels, fetched, exact = [], [], 0, 0, 0
    for seed in range(lo, hi):
        for kind in ("plain", "ex"):
            try:
                if kind == "plain":
                    src, needs = make_filter(seed)
                    names, opts = (["in"] if needs else []), {}
                else:
                    src, names, opts = make_filter_ex(seed)
                if not names:
                    continue
                uv = {"k": seed % 7, "m": 0.3 + (seed % 5) * 0.4}
                images = {n: imgs[n] for n in names}
                edge = (opts.get("edge_x", 0), opts.get("edge_y", 0))
                want = CpuFilter(mm.Filter(src, **opts).ir_json_raw).render(
                    w, h, uservals=uv, images=images, t=0.4, intersample=opts.get("intersample", True), edge=edge, edge_colors=colors)
                for shape in ("generic", "specialised", "one-pixel"):
                    if shape == "one-pixel":
                        os.environ["MMHIP_SINGLE_PIXEL"] = "1"
                    try:
                        flt = mm.Filter(src + (" " if shape == "one-pixel" else ""), specialize=shape == "specialised", **opts)
                    finally:
                        os.environ.pop("MMHIP_SINGLE_PIXEL", None)
                    kernels += 1
                    fetched += "mm_store_fetched_pixel(A, rl_raw" in flt.kernel_source
                    inv = flt.invoke(w, h)
                    for k, v in uv.items():
                        inv.set(k, v)
                    for n in names:
                        inv.set_image(n, imgs[n])
                    inv.set_edge_colors(*colors)
                    got = inv.render(t=0.4)
                    if np.array_equal(got, want):
                        exact += 1
                    else:
                        d = np.abs(got.astype(int) - want.astype(int))
                        inexact.append((seed, kind, shape, int(d.max()), int((d > 0).sum()), int((d > 1).sum())))
            except Exception as e:
                errors.append((seed, kind, str(e).splitlines()[0][:140]))
        if seed % 25 == 0:
            print("seed", seed, "kernels", kernels, "exact", exact, "not exact", len(inexact), "errors", len(errors), flush=True)
    print("kernels:", kernels, "with the fetched-pixel store:", fetched, "bit-exact:", exact)
    print("not bit-exact (seed, generator, shape, max, n_diff, n_gt1):", inexact)
    print("errors:", errors)


if __name__ == "__main__":
    main()

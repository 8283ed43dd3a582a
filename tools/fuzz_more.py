#!/usr/bin/env python3
"""Long differential fuzz run on the GPU box (tests/fuzz_filters.py generators): HIP vs oracle and
specialised vs generic, for seeds [lo, hi).  usage: fuzz_more.py [lo hi]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import mathmap_amd as mm
from tests import filters as F
from oracle.ccgen import CpuFilter
from fuzz_filters import make_filter, make_filter_ex


def main():
    lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (400, 1000)
    w, h = 96, 64
    imgs = {"in": F.synthetic_image(w, h, seed=1), "in2": F.synthetic_image(50, 70, seed=2)}
    bad = []
    for seed in range(lo, hi):
        for variant in ("plain", "ex"):
            if variant == "plain":
                src, needs = make_filter(seed)
                names, opts = (["in"] if needs else []), {}
            else:
                src, names, opts = make_filter_ex(seed)
            uv = {"k": seed % 7, "m": 0.3 + (seed % 5) * 0.4}
            try:
                outs = []
                for spec in (False, True):
                    flt = mm.Filter(src, specialize=spec, **opts)
                    inv = flt.invoke(w, h)
                    for k, v in uv.items():
                        inv.set(k, v)
                    for n in names:
                        inv.set_image(n, imgs[n])
                    outs.append(inv.render(t=0.4))
                if not np.array_equal(outs[0], outs[1]):
                    bad.append((seed, variant, "spec != generic"))
                    continue
                want = CpuFilter(mm.Filter(src, **opts).ir_json_raw).render(
                    w, h, uservals=uv, images={n: imgs[n] for n in names}, t=0.4,
                    intersample=opts.get("intersample", True), edge=(opts.get("edge_x", 0), opts.get("edge_y", 0)))
                d = np.abs(outs[0].astype(int) - want.astype(int))
                if d.max() > 1 and (d > 1).sum() >= 0.01 * want.size:
                    bad.append((seed, variant, "vs oracle max %d n>1 %d" % (d.max(), (d > 1).sum())))
            except Exception as e:
                bad.append((seed, variant, str(e).splitlines()[0][:200]))
        if seed % 100 == 0:
            print("seed", seed, "bad so far", len(bad), flush=True)
    print("bad:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

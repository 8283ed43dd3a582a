#!/usr/bin/env python3
"""Long differential run of pair mode (tests/fuzz_filters.py make_filter_arith): pair kernel vs oracle,
generic and specialised, for seeds [lo, hi).  usage: fuzz_pair.py [lo hi]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["MMHIP_PAIR"] = "1"
import numpy as np
import mathmap_amd as mm
from oracle.ccgen import CpuFilter
from fuzz_filters import make_filter_arith


def main():
    lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100, 400)
    w, h = 97, 63
    bad, paired = [], 0
    for seed in range(lo, hi):
        src = make_filter_arith(seed)
        uv = {"k": seed % 9, "m": 0.1 + (seed % 7) * 0.3}
        try:
            want = CpuFilter(mm.Filter(src).ir_json_raw).render(w, h, uservals=uv, t=0.6)
            for spec in (False, True):
                flt = mm.Filter(src, specialize=spec)
                paired += "const mm_pf mm_y2" in flt.kernel_source
                inv = flt.invoke(w, h)
                for k, v in uv.items():
                    inv.set(k, v)
                got = inv.render(t=0.6)
                if not np.array_equal(got, want):
                    d = np.abs(got.astype(int) - want.astype(int))
                    bad.append((seed, spec, int(d.max()), int((d > 0).sum())))
        except Exception as e:
            bad.append((seed, str(e).splitlines()[0][:160]))
        if seed % 50 == 0:
            print("seed", seed, "bad so far", len(bad), "paired kernels", paired, flush=True)
    print("paired kernels:", paired, "bad:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

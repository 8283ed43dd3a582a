#!/usr/bin/env python3
"""Why tests/golden/map_ifs_functional.png differs from the oracle (and from the GPU) by 2 in two values of one pixel.

The oracle's generated C is rebuilt with `acos` wrapped (gcc -include): the wrapper returns glibc's result rounded to float
(what the assignment to a float compvar does anyway) and nudges ONE call of the row -- the j-th of pixel (165, 224) -- up
by one float ulp.  Run where /root/reference is present:  python tools/ifs_functional_diagnosis.py
Output of this script at round 3: profiles/r03_ifs_functional_diagnosis.txt.  Test infrastructure (drives the oracle)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mathmap_amd as mm  # noqa: E402
from oracle.ccgen import CpuFilter  # noqa: E402
from tests.conftest import load_png_rgb  # noqa: E402

PX, ROW, CALLS_PER_PIXEL = 165, 224, 7          # depth 8: seven recursive levels, one acos (the polar angle) each

HDR = '''#include <math.h>
#include <stdio.h>
#include <stdlib.h>
static long pert_cnt = 0;
static inline double pert_acos(double a) { float f = (float)acos(a); const char *e = getenv("PERT_T"); long t = e ? atol(e) : -1;
  if (pert_cnt == t) { fprintf(stderr, "    acos(%a = %.9g) -> float %a = %.9g (glibc's double: %a); nudged up one float ulp\\n", a, a, (double)f, (double)f, acos(a)); f = nextafterf(f, INFINITY); }
  ++pert_cnt; return (double)f; }
#define acos pert_acos
'''


def main():
    src = open("/root/reference/examples/Map/IFS Functional.mm").read()
    flt = mm.Filter(src)
    img = load_png_rgb("marlene.png")
    want = load_png_rgb("map_ifs_functional.png")
    full = CpuFilter(flt.ir_json_raw).render(256, 256, images={"in": img})
    d = full[:, :, :3].astype(int) - want.astype(int)
    print("oracle vs golden: %d of %d values differ, %d by 2 -- all of the latter in pixel (x=%d, y=%d): oracle %s, golden %s"
          % ((d != 0).sum(), d.size, (abs(d) == 2).sum(), PX, ROW, full[ROW, PX, :3], want[ROW, PX]))
    hdr = "/tmp/mm_pert_acos.h"
    open(hdr, "w").write(HDR)
    cf = CpuFilter(flt.ir_json_raw, extra_cflags=("-include", hdr))
    os.environ["PERT_T"] = "-1"
    base = cf.render(256, 256, images={"in": img}, rows=(ROW, ROW + 1))[ROW].copy()
    n = 256 * CALLS_PER_PIXEL                     # acos calls per rendered row; the wrapper's counter runs on across renders
    for j in range(CALLS_PER_PIXEL):
        os.environ["PERT_T"] = str((j + 1) * n + PX * CALLS_PER_PIXEL + j)
        sys.stderr.flush()
        row = cf.render(256, 256, images={"in": img}, rows=(ROW, ROW + 1))[ROW]
        changed = np.nonzero((row != base).any(axis=1))[0].tolist()
        print("  acos call %d of the pixel nudged: pixels of the row that change: %s%s"
              % (j, changed, "  -> %s %s" % (row[PX, :3], "= the golden's bytes" if (row[PX, :3] == want[ROW, PX]).all() else "") if PX in changed else ""))


if __name__ == "__main__":
    main()

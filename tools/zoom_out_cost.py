#!/usr/bin/env python3
"""Kernel time of filters that sample mostly outside / partly outside / inside their input (in(xy * k)) at 8192^2.
Used to try the general fetch's outside-the-image shortcut in the hot fetch's not-all-inside branch as well (late
round 2): in(xy * 3) 0.196 -> 0.161 ms, but in(xy) 0.184 -> 0.216, Ident 0.175 -> 0.206 and Pond 0.60 -> 0.67 ms -- the
extra exit changed how the compiler lays out the all-inside path.  Not adopted."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mathmap_amd as mm
from tests import filters as F


def main():
    size = 8192
    out = torch.empty((size, size), dtype=torch.int32, device="cuda")
    img = F.synthetic_image(size, size)
    for k in ("3", "1.2", "1"):
        flt = mm.Filter("filter z (image in) in(xy * %s) end" % k)
        inv = flt.invoke(size, size)
        inv.set_image("in", img)
        inv.enable_timing(True)
        for _ in range(40):
            inv.render_rows(out.data_ptr(), 0, size, t=0.1)
        ts = sorted(inv.drain_kernel_ms()[5:])
        print("in(xy * %s): %.4f ms" % (k, ts[len(ts) // 2]), flush=True)


if __name__ == "__main__":
    main()

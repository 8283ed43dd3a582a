#!/usr/bin/env python3
"""Times a ladder of tiny filters at 8192^2 (HIP events, median of 20) to attribute the per-pixel
cost of the generated kernel: launch + store, pack, coordinates, nearest fetch, bilinear fetch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mathmap_amd as mm
from tests import filters as F

CASES = [
    ("const colour (store+pack)", "filter c () rgba:[0.2,0.4,0.6,1] end", {}, False),
    ("grayColor(x) (coords+pack)", "filter c () grayColor(x) end", {}, False),
    ("rgba:[x,y,x*y,1]", "filter c () rgba:[x,y,x*y,1] end", {}, False),
    ("ident nearest", "ident", dict(intersample=False), True),
    ("ident bilinear", "ident", {}, True),
    ("ident bilinear, half-pixel shift", "filter s (image in) in(xy + xy:[0.37/X/1000, 0.21/Y/1000]) end", {}, True),
    ("pond", "pond", {}, True),
]


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    dev = torch.device("cuda:0")
    img = torch.randint(0, 2 ** 31 - 1, (size, size), dtype=torch.int32, device=dev)
    out = torch.empty((size, size), dtype=torch.int32, device=dev)
    for name, src, opts, needs in CASES:
        tw = int(os.environ.get('TILE_W', '0'))
        flt = F.load(src, tile_w=tw, **opts) if src in F.NAMES else mm.Filter(src, tile_w=tw, **opts)
        inv = flt.invoke(size, size)
        if needs:
            inv.set_image_device("in", img.data_ptr(), size, size)
        inv.enable_timing(True)
        ts = []
        for i in range(25):
            inv.render_rows(out.data_ptr(), 0, size, t=0.1)
            inv.sync()
            ts.append(inv.last_kernel_ms())
        ts = sorted(ts[5:])
        print("%-36s %.4f ms  (min %.4f)" % (name, ts[len(ts) // 2], ts[0]), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""A/B of the per-row slice (hipgen.cpp find_row_slice, MMHIP_NO_ROW_SLICE=1 switches it off): kernel ms at 8192^2 of
filters whose row-only part holds library calls.  usage (GPU box): python tools/row_slice_cost.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mathmap_amd as mm

FILTERS = [
    ("wave: in(xy + [sin(y*10 + t*6)*amp, 0])", "filter wave (image in, float amp: 0-1 (0.1)) in(xy + xy:[sin(y * 10 + t * 6) * amp, 0]) end", True),
    ("flag: two row waves", "filter flag (image in) in(xy + xy:[sin(y * 10 + t * 6) * 0.1 + cos(y * 23) * 0.02, exp(y) * 0.01]) end", True),
    ("row gradient: grayColor(pow-like)", "filter g () grayColor(exp(y * 2) / 8 + sin(y * 40) * 0.1 + x * 0.01) end", False),
]


def main():
    size = 8192
    img = torch.randint(0, 2 ** 31 - 1, (size, size), dtype=torch.int32, device="cuda")
    out = torch.empty((size, size), dtype=torch.int32, device="cuda")
    for label, src, needs in FILTERS:
        res = []
        for off in (False, True):
            if off:
                os.environ["MMHIP_NO_ROW_SLICE"] = "1"
            else:
                os.environ.pop("MMHIP_NO_ROW_SLICE", None)
            flt = mm.Filter(src)
            inv = flt.invoke(size, size)
            if needs:
                inv.set_image_device("in", img.data_ptr(), size, size)
            inv.enable_timing(True)
            for i in range(40):
                inv.render_rows(out.data_ptr(), 0, size, t=i / 120.0)
            ts = sorted(inv.drain_kernel_ms()[8:])
            res.append(ts[len(ts) // 2])
        print("%-44s rows kernel + table: %.4f ms   recomputed per pixel: %.4f ms" % (label, res[0], res[1]), flush=True)


if __name__ == "__main__":
    main()

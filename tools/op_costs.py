#!/usr/bin/env python3
"""Marginal cost of single ops in the generated pixel kernel at 8192^2 (kernel ms above the
`grayColor(x*y)` base): which libm calls are worth restating."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mathmap_amd as mm

# every argument depends on x and y: an x-only expression is hoisted out of the rows-per-work-item loop
OPS = ["x*y", "sin(x*y*9)", "cos(x*y*9)", "tan(x*y)", "asin(x*y)", "acos(x*y)", "atan(x*y*5)", "atan(y, x)", "abs(ri:[x, y])",
       "x/(y+2)", "exp(x*y*4)", "log(x*y+1.5)", "(x+1.5)^(y*3)", "sqrt(x*y+1.5)", "sinh(x*y*3)", "tanh(x*y*3)", "(x*y)%0.37",
       "floor(x*y*9)", "abs(exp(ri:[x, y*6]))", "abs(log(ri:[x, y]))", "abs(ri:[x,y]^ri:[1.3,0.4])", "abs(sin(ri:[x*3,y]))",
       "abs(sqrt(ri:[x,y]))", "log(abs(ri:[x,y])+0.001)", "log(abs(ri:[x,y])+0.001)+atan(y,x)", "(log(ri:[x,y]))[0]",
       "(log(ri:[x,y]))[1]", "r", "a", "r+a", "sin(r / 0.04) * 0.05 + a", "cos(a) * r", "cos(a) * r + sin(a) * r"]


def main():
    size = 8192
    out = torch.empty((size, size), dtype=torch.int32, device="cuda")
    base = None
    for e in OPS:
        flt = mm.Filter("filter p () grayColor(%s) end" % e)
        inv = flt.invoke(size, size)
        inv.enable_timing(True)
        for _ in range(12):
            inv.render_rows(out.data_ptr(), 0, size, t=0.1)
        ts = sorted(inv.drain_kernel_ms()[2:])
        ms = ts[len(ts) // 2]
        if base is None:
            base = ms
        print("%-28s %.4f ms   +%.4f   ~%d VALU/px" % (e, ms, ms - base, round((ms - base) * 1e-3 * 1024 * 2.4e9 / 4 / (size * size / 64))), flush=True)


if __name__ == "__main__":
    main()

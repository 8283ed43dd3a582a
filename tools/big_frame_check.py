#!/usr/bin/env python3
"""Frames beyond 2^30 pixels / 4 GiB: 64-bit addressing of stores and (with the hot fetch path
disabled by its own size test) of texel loads.  Mandelbrot 40000^2 and Ident 36000x32768, sampled
row bands compared with the oracle."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mathmap_amd as mm
from tests import filters as F
from oracle.ccgen import CpuFilter


def main():
    # --- Mandelbrot 40000 x 40000 (6.4 GB of RGBA8)
    w = h = 40000
    flt = F.load("mandelbrot", specialize=True)
    inv = flt.invoke(w, h)
    out = torch.empty((h, w), dtype=torch.int32, device="cuda")
    inv.render_rows(out.data_ptr(), 0, h)
    inv.sync()
    cf = CpuFilter(F.load("mandelbrot").ir_json_raw)
    for lo in (0, 13333, 26844, 39992):
        want = cf.render(w, h, rows=(lo, lo + 8))[lo:lo + 8]
        got = out[lo:lo + 8].cpu().numpy().view(np.uint8).reshape(8, w, 4)
        assert np.array_equal(got, want), ("mandelbrot", lo)
    print("mandelbrot 40000^2 ok", flush=True)
    del out, inv
    torch.cuda.empty_cache()
    # --- Ident 36000 x 32768 with an input of the same size (hot path off: w*h >= 2^30)
    w, h = 36000, 32768
    rng = np.random.default_rng(3)
    band = rng.integers(0, 256, (64, w, 3), dtype=np.uint8)
    img = np.tile(band, (h // 64, 1, 1))
    img[:, :, 0] ^= (np.arange(h, dtype=np.uint32)[:, None] & 255).astype(np.uint8)     # rows differ
    flt = F.load("ident")
    inv = flt.invoke(w, h)
    inv.set_image("in", img)
    out = torch.empty((h, w), dtype=torch.int32, device="cuda")
    inv.render_rows(out.data_ptr(), 0, h)
    inv.sync()
    cf = CpuFilter(flt.ir_json_raw)
    for lo in (0, 16383, 29999, h - 8):
        got = out[lo:lo + 8].cpu().numpy().view(np.uint8).reshape(8, w, 4)
        want = cf.render(w, h, images={"in": img}, rows=(lo, lo + 8))[lo:lo + 8]
        # (not the identity at this size: float coordinates carry ~0.002 px of rounding, which the
        # bilinear weights of random texels turn into byte differences -- in the reference too)
        assert np.array_equal(got, want), ("ident", lo, int(np.abs(got.astype(int) - want).max()))
    print("ident 36000x32768 ok", flush=True)


if __name__ == "__main__":
    main()

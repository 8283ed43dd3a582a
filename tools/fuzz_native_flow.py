#!/usr/bin/env python3
"""Differential fuzz of native-filter calls in control flow (GPU box): random nests of frame-constant conditionals
(a user value), frame-constant loops (a user value / t as the count), pixel-dependent conditionals and chains of
gaussian_blur / render calls -- the call sites inside loops (numbered as made), the calls under pixel-dependent
control (moved to the frame-constant slice), calls before / inside / behind loops in program order.  HIP through the
C ABI against the oracle, for seeds [lo, hi), at two times each; also through the reference-ABI tier for every 4th seed.
usage: fuzz_native_flow.py [lo hi]"""
import ctypes as C
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import mathmap_amd as mm  # noqa: E402
from tests import filters as F  # noqa: E402
from oracle.ccgen import CpuFilter  # noqa: E402


class Gen:
    def __init__(self, seed, rich=False):
        self.r = random.Random(seed)
        self.rich = rich             # closures handed to natives, the FFT filters
        self.in_pixel_now = False
        self.calls_in_loops = 0      # static bound on calls made from in-loop sites (16 dynamic entries per frame)
        self.tmp = 0

    def dev(self):
        # (rich: no sigma below 0.5 px -- the FIR path picks do_full_lre or do_encoded_lre per line by counting *equal*
        # neighbours (gauss.c:333-375), so behind an FFT filter, whose last bits differ between FFT implementations, it
        # switches paths on rounding noise: seed 36 of the first rich run, 32 levels apart.  The reference would differ
        # from itself with another FFTW build.)
        return "s * %g" % self.r.choice([1, 1.5, 2, 3] if self.rich else [0.5, 1, 1.5, 2, 3])

    def call(self, img):
        k = self.r.random()
        if k < 0.2:
            return "render(%s)" % img
        if self.rich and k < 0.3 and not self.in_loop and not self.in_pixel_now:      # a closure image (refused inside loops)
            return self.closure_call(img)
        if self.rich and k < 0.38:
            return self.r.choice(["convolve(%s, in, 1, 0)", "half_convolve(%s, in, 0)", "visualize_fft(%s, 0)"]) % img
        return "gaussian_blur(%s, %s, %s)" % (img, self.dev(), self.dev())

    def closure_call(self, img):
        c = "inner(%s, %g)" % (img, self.r.choice([0.7, 1, 1.2]))
        if self.r.random() < 0.3:
            return "render(%s)" % c
        return "gaussian_blur(%s, %s, %s)" % (c, self.dev(), self.dev())

    def fresh(self):
        self.tmp += 1
        return "q%d" % self.tmp

    def block(self, depth, loop_mult, in_pixel):
        """Statements that update `img` (frame-constant image chain, only outside pixel-dependent control) and `acc`."""
        out = []
        for _ in range(self.r.randint(1, 3)):
            k = self.r.random()
            if k < 0.35 or depth >= 3:
                self.in_pixel_now = in_pixel
                if in_pixel:
                    q = self.fresh()
                    out.append("%s = %s; acc = acc + %s(xy * %g) * %g" % (q, self.call("img"), q, self.r.choice([1, 0.9, 1.1]),
                                                                          self.r.choice([0.2, 0.3, 0.5])))
                else:
                    out.append("img = %s" % self.call("img"))
                self.calls_in_loops += loop_mult if loop_mult > 1 or self.in_loop else 0
            elif k < 0.55:
                c = self.r.choice(["mode > 0", "mode > 1", "mode == 1", "t > 0.5"])
                a, b = self.block(depth + 1, loop_mult, in_pixel), self.block(depth + 1, loop_mult, in_pixel)
                out.append("if %s then %s; 0 else %s; 0 end" % (c, "; ".join(a), "; ".join(b)))
            elif k < 0.75 and not in_pixel:
                c = self.r.choice(["x > %g" % self.r.choice([-0.3, 0, 0.4]), "y < %g" % self.r.choice([-0.2, 0.1]), "r < 0.6",
                                   "x * y > 0"])
                a = self.block(depth + 1, loop_mult, True)
                out.append("if %s then %s; 0 else acc = acc + img(xy) * 0.1; 0 end" % (c, "; ".join(a)))
            elif k < 0.9 and not in_pixel and loop_mult == 1:
                n = self.r.choice([1, 2, 3])
                bound = self.r.choice(["%d" % n, "n", "1 + t * 2"])
                worst = {"n": 3, "1 + t * 2": 3}.get(bound, n)
                was = self.in_loop
                self.in_loop = True
                i = self.fresh()
                body = self.block(depth + 1, worst, in_pixel)
                self.in_loop = was
                out.append("%s = 0; while %s < %s do %s; %s = %s + 1 end" % (i, i, bound, "; ".join(body), i, i))
            else:
                out.append("acc = acc + img(xy * %g) * %g" % (self.r.choice([1, 0.8]), self.r.choice([0.1, 0.2])))
        return out

    def filter(self):
        self.in_loop = False
        body = self.block(0, 1, False)
        inner = "filter inner (image in, float k: 0-2 (1.0))\n  in(xy * k) * 0.8 + rgba:[t * 0.3, 0, 0.1, 0]\nend\n\n" if self.rich else ""
        return (inner + "filter f (image in, float s: 0-1 (0.02), int mode: 0-2 (1), int n: 0-3 (2))\n  img = in; acc = rgba:[0, 0, 0, 0];\n  "
                + ";\n  ".join(body) + ";\n  acc + img(xy) * 0.4\nend\n")


def main():
    import faulthandler
    faulthandler.enable()      # a host-side crash names the call it happened in
    from mathmap_amd._lib import selftest_lib
    lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 200)
    rich = len(sys.argv) > 3 and sys.argv[3] == "rich"
    bad, ran, skipped, calls, refused = [], 0, 0, 0, 0
    for seed in range(lo, hi):
        w, h = [(96, 64), (64, 96), (80, 80), (112, 48)][seed % 4] if rich else (96, 64)
        iw, ih = [(w, h), (50, 70), (w, h), (131, 40)][(seed // 4) % 4] if rich else (w, h)      # the input's own size
        img = np.ascontiguousarray(F.synthetic_image(iw, ih, seed=3))
        g = Gen(seed, rich)
        src = g.filter()
        if g.calls_in_loops > 16 or "gaussian_blur" not in src and "render" not in src:
            skipped += 1
            continue
        print("seed", seed, flush=True)      # (a GPU fault ends the process: the last line names the filter)
        faulthandler.dump_traceback_later(120, exit=True)      # a hang: say where, and end
        try:
            flt = mm.Filter(src)
            cf = CpuFilter(flt.ir_json_raw)
            inv = flt.invoke(w, h)
            inv.set_image("in", img)
            for mode, n, t in ((1, 2, 0.25), (2, 3, 0.75), (0, 1, 0.6)):
                inv.set("mode", mode)
                inv.set("n", n)
                got = inv.render(t=t)
                want = cf.render(w, h, uservals={"mode": mode, "n": n}, images={"in": img}, t=t)
                if not np.array_equal(got, want):
                    d = np.abs(got.astype(int) - want.astype(int))
                    fft = "convolve(" in src or "visualize_fft(" in src      # hipFFT against the oracle's own DFT: 1 LSB
                    if not (fft and d.max() <= 1 and (d > 0).sum() <= 0.001 * d.size):
                        bad.append((seed, mode, n, t, int(d.max()), int((d > 0).sum())))
            ran += 1
            calls += flt.num_native_calls
            if seed % 4 == 0:      # the reference-ABI tier on the same text (defaults of the user values)
                want = flt.invoke(w, h)
                want.set_image("in", img)
                ref = want.render(t=0.25)
                out = np.zeros((h, w, 4), np.uint8)
                rc = selftest_lib().mmhip_selftest_abi_roundtrip(src.encode(), 1, img.ctypes.data_as(C.c_void_p), iw, ih, 3, w, h, 0.25, 2,
                                                                 out.ctypes.data_as(C.c_void_p))
                if rc != 0:
                    bad.append((seed, "abi", selftest_lib().mmhip_selftest_error().decode()[:200]))
                elif not np.array_equal(out, ref):
                    bad.append((seed, "abi differs"))
        except Exception as e:
            msg = str(e).splitlines()[0][:200]
            if rich and ("inside a loop is not supported" in msg or "pixel-dependent arguments" in msg or "needs frame-constant arguments" in msg):
                refused += 1      # documented refusals (closures in loops / under pixel-dependent control): loud, not wrong
            else:
                bad.append((seed, "error", msg))
        if seed % 25 == 0:
            print("  ran", ran, "skipped", skipped, "bad so far", len(bad), flush=True)
    print("filters:", ran, "skipped:", skipped, "refused:", refused, "native call sites:", calls, "bad:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

"""Native filters called from inside a loop of the frame-constant code: one call per iteration, each with a result of
its own that the next iteration (or the code behind the loop) reads.  The reference simply runs the calls as its
init_frame code reaches them (new_template.c.in:314-337, native-filters/cache.c:110-147); here the prologue kernel
numbers the calls as it makes them and the host runs the recorded calls in that order (hipgen.cpp mm_native_call_in_loop,
runtime.cpp recorded_calls).  HIP through the C ABI against the oracle, and against the same chain written out."""
import ctypes as C

import numpy as np
import pytest

import mathmap_amd as mm
from oracle.ccgen import CpuFilter
from tests import filters as F
from tests.gpu_util import make_invocation, stats

pytestmark = pytest.mark.gpu

# n blurs in a row; n is a user value (the loop survives into the kernel) -- and with it the number of calls
REPEATED = """
filter repeated (image in, int n: 0-40 (2), float s: 0-1 (0.02))
  img = in; i = 0;
  while i < n do img = gaussian_blur(img, s, s * (i + 1)); i = i + 1 end;
  img(xy)
end
"""

UNROLLED = {
    0: "filter u (image in, float s: 0-1 (0.02)) in(xy) end",
    1: "filter u (image in, float s: 0-1 (0.02)) p = gaussian_blur(in, s, s); p(xy) end",
    3: "filter u (image in, float s: 0-1 (0.02)) p = gaussian_blur(in, s, s); q = gaussian_blur(p, s, s * 2); "
       "w = gaussian_blur(q, s, s * 3); w(xy) end",
}

# a call before the loop, calls in the loop (their number follows t), a call behind the loop that reads the loop's
# result, and a second in-loop site under a condition: the order the calls are made in is what the host must follow
MIXED = """
filter mixed (image in, float s: 0-1 (0.015))
  first = gaussian_blur(in, s, s);
  img = first; i = 0;
  while i < 1 + t * 4 do
    if i % 2 == 0 then img = gaussian_blur(img, s * 2, s) else img = render(img) end;
    i = i + 1
  end;
  last = gaussian_blur(img, s, s * 2);
  last(xy) * 0.5 + first(xy) * 0.5
end
"""


@pytest.mark.parametrize("n", [0, 1, 3])
def test_blur_repeated_in_a_loop_equals_the_chain_written_out(n):
    w, h = 160, 96
    img = F.synthetic_image(w, h, seed=5)
    flt, inv = make_invocation(REPEATED, w, h, {"n": n}, {"in": img})
    got = inv.render(t=0.25)
    want = CpuFilter(flt.ir_json_raw).render(w, h, uservals={"n": n}, images={"in": img}, t=0.25)
    assert np.array_equal(got, want), stats(got, want)
    _, ui = make_invocation(UNROLLED[n], w, h, {}, {"in": img})
    assert np.array_equal(got, ui.render(t=0.25))
    # changing the count on the same invocation: fewer calls, then more
    for m in (1, 2):
        inv.set("n", m)
        want = CpuFilter(flt.ir_json_raw).render(w, h, uservals={"n": m}, images={"in": img}, t=0.25)
        assert np.array_equal(inv.render(t=0.25), want), m


def test_calls_before_inside_and_behind_a_loop_run_in_program_order():
    w, h = 128, 80
    img = F.synthetic_image(w, h, seed=6)
    flt, inv = make_invocation(MIXED, w, h, {}, {"in": img})
    cf = CpuFilter(flt.ir_json_raw)
    for t in (0.0, 0.3, 0.6, 0.3):      # 1, 3, 4, 3 iterations
        got = inv.render(t=t)
        want = cf.render(w, h, images={"in": img}, t=t)
        assert np.array_equal(got, want), (t, stats(got, want))


def test_more_calls_than_dynamic_entries_is_an_error_not_a_wrong_frame():
    w, h = 64, 48
    img = F.synthetic_image(w, h, seed=7)
    _, inv = make_invocation(REPEATED, w, h, {"n": 17}, {"in": img})
    with pytest.raises(mm.MathMapError, match="more than 16 times from inside a loop"):
        inv.render(t=0.0)
    inv.set("n", 16)
    inv.render(t=0.0)


def test_loop_of_native_calls_through_the_reference_abi():
    from mathmap_amd._lib import selftest_lib
    w, h = 128, 80
    img = np.ascontiguousarray(F.synthetic_image(w, h, seed=6))
    _, inv = make_invocation(MIXED, w, h, {}, {"in": img})
    want = inv.render(t=0.5)
    got = np.zeros((h, w, 4), np.uint8)
    rc = selftest_lib().mmhip_selftest_abi_roundtrip(MIXED.encode(), 1, img.ctypes.data_as(C.c_void_p), w, h, 3, w, h, 0.5, 2,
                                                     got.ctypes.data_as(C.c_void_p))
    assert rc == 0, selftest_lib().mmhip_selftest_error().decode()
    assert np.array_equal(got, want), stats(got, want)


# ---- native calls with frame-constant arguments under pixel-dependent control ----
# The reference runs such a call when the first pixel reaches it and finds the result in its cache afterwards
# (native-filters/cache.c:110-147); here the call moves in front of the outermost pixel-dependent statement around it
# (passes.cpp hoist_native_calls) and runs once per frame.
SPLIT = """
filter split (image in, float s: 0-1 (0.02), int mode: 0-1 (1))
  if mode > 0 then
    if x > 0 then
      b = gaussian_blur(in, s, s * 2); b(xy)
    else
      if y > t - 0.5 then c = gaussian_blur(in, s * 3, s); rr = render(in); c(xy) * 0.5 + rr(xy * 0.9) * 0.5 else in(xy) end
    end
  else
    in(xy * 0.5)
  end
end
"""


@pytest.mark.parametrize("mode", [1, 0])
def test_native_calls_under_pixel_dependent_conditionals(mode):
    w, h = 160, 96
    img = F.synthetic_image(w, h, seed=9)
    flt, inv = make_invocation(SPLIT, w, h, {"mode": mode}, {"in": img})
    cf = CpuFilter(flt.ir_json_raw)
    for t in (0.25, 0.75):
        got = inv.render(t=t)
        want = cf.render(w, h, uservals={"mode": mode}, images={"in": img}, t=t)
        assert np.array_equal(got, want), (mode, t, stats(got, want))


def test_native_calls_under_pixel_dependent_conditionals_through_the_reference_abi():
    from mathmap_amd._lib import selftest_lib
    w, h = 160, 96
    img = np.ascontiguousarray(F.synthetic_image(w, h, seed=9))
    _, inv = make_invocation(SPLIT, w, h, {}, {"in": img})
    want = inv.render(t=0.25)
    got = np.zeros((h, w, 4), np.uint8)
    rc = selftest_lib().mmhip_selftest_abi_roundtrip(SPLIT.encode(), 1, img.ctypes.data_as(C.c_void_p), w, h, 3, w, h, 0.25, 2,
                                                     got.ctypes.data_as(C.c_void_p))
    assert rc == 0, selftest_lib().mmhip_selftest_error().decode()
    assert np.array_equal(got, want), stats(got, want)


def test_pixel_dependent_arguments_are_still_refused():
    with pytest.raises(mm.MathMapError, match="pixel-dependent arguments"):
        mm.Filter("filter f (image in) b = gaussian_blur(in, 0.01 + abs(x) * 0.01, 0.01); b(xy) end")


# every call of an in-loop site has a result of its own: the last one and the one before it are both still readable
# behind the loop (the reference caches per argument set, native-filters/cache.c:110-156)
KEEP = """
filter keep (image in, float s: 0-1 (0.02))
  img = in; prev = in; i = 0;
  while i < 3 do prev = img; img = gaussian_blur(img, s * (i + 1), s); i = i + 1 end;
  prev(xy) * 0.5 + img(xy) * 0.5
end
"""
KEEP_UNROLLED = ("filter u (image in, float s: 0-1 (0.02)) p = gaussian_blur(in, s, s); q = gaussian_blur(p, s * 2, s); "
                 "w = gaussian_blur(q, s * 3, s); q(xy) * 0.5 + w(xy) * 0.5 end")


def test_results_of_earlier_iterations_stay_readable():
    w, h = 160, 96
    img = F.synthetic_image(w, h, seed=11)
    flt, inv = make_invocation(KEEP, w, h, {}, {"in": img})
    got = inv.render(t=0.25)
    want = CpuFilter(flt.ir_json_raw).render(w, h, images={"in": img}, t=0.25)
    assert np.array_equal(got, want), stats(got, want)
    _, ui = make_invocation(KEEP_UNROLLED, w, h, {}, {"in": img})
    assert np.array_equal(got, ui.render(t=0.25))


# ---- loops that hold native calls *and* per-pixel code: emitted in both slices (passes.cpp mark_dual_loops) ----
# the loop-constant part (the image chain, its counters and conditions) runs in the prologue, where the calls are
# numbered; the pixel slice runs the whole loop and takes the n-th call's result for its n-th call
GLOW = """
filter glow (image in, float s: 0-1 (0.015), int n: 0-4 (3), int mode: 0-1 (1))
  img = in; acc = in(xy) * 0.4; i = 0;
  while i < n do
    img = gaussian_blur(img, s * (i + 1), s * (i + 1));
    acc = acc + img(xy) * 0.2;
    if x > -0.2 then q = gaussian_blur(img, s, s * 3); acc = acc + q(xy * 0.9) * 0.1 else acc = acc * 0.9 end;
    if mode > 0 then img = render(img); 0 else 0 end;
    i = i + 1
  end;
  last = gaussian_blur(img, s, s);
  acc + last(xy) * 0.2
end
"""

GLOW_UNROLLED_2 = """
filter glow2 (image in, float s: 0-1 (0.015))
  acc = in(xy) * 0.4;
  a1 = gaussian_blur(in, s, s); acc = acc + a1(xy) * 0.2;
  q1 = gaussian_blur(a1, s, s * 3);
  if x > -0.2 then acc = acc + q1(xy * 0.9) * 0.1 else acc = acc * 0.9 end;
  r1 = render(a1);
  a2 = gaussian_blur(r1, s * 2, s * 2); acc = acc + a2(xy) * 0.2;
  q2 = gaussian_blur(a2, s, s * 3);
  if x > -0.2 then acc = acc + q2(xy * 0.9) * 0.1 else acc = acc * 0.9 end;
  r2 = render(a2);
  last = gaussian_blur(r2, s, s);
  acc + last(xy) * 0.2
end
"""


@pytest.mark.parametrize("n,mode", [(0, 1), (1, 0), (2, 1), (4, 1), (3, 0)])
def test_loop_with_native_calls_and_per_pixel_code(n, mode):
    w, h = 160, 96
    img = F.synthetic_image(w, h, seed=13)
    flt, inv = make_invocation(GLOW, w, h, {"n": n, "mode": mode}, {"in": img})
    got = inv.render(t=0.25)
    want = CpuFilter(flt.ir_json_raw).render(w, h, uservals={"n": n, "mode": mode}, images={"in": img}, t=0.25)
    assert np.array_equal(got, want), stats(got, want)
    if (n, mode) == (2, 1):
        _, ui = make_invocation(GLOW_UNROLLED_2, w, h, {}, {"in": img})
        assert np.array_equal(got, ui.render(t=0.25))


def test_frame_constant_fetch_from_a_native_result_waits_for_the_result():
    """`b(xy:[0.1, 0.2])` has frame-constant coordinates, but b's pixels exist only after the host has run the blur,
    behind the prologue: the fetch is per-pixel code (it was hoisted into the prologue until round 3 and read the map of
    the previous frame -- or none)."""
    w, h = 128, 80
    img = F.synthetic_image(w, h, seed=14)
    src = ("filter f (image in, float s: 0-1 (0.02)) b = gaussian_blur(in, s * (1 + t), s); c = b(xy:[0.1, 0.2]); "
           "d = if c[0] > 0.3 then b(xy) else in(xy) end; d * 0.5 + c * 0.5 end")
    flt, inv = make_invocation(src, w, h, {}, {"in": img})
    cf = CpuFilter(flt.ir_json_raw)
    for t in (0.0, 0.5, 0.25):
        got = inv.render(t=t)
        want = cf.render(w, h, images={"in": img}, t=t)
        assert np.array_equal(got, want), (t, stats(got, want))


@pytest.mark.parametrize("size", [(96, 64), (64, 96), (80, 80)])
def test_render_of_a_native_result_on_a_non_square_frame(size):
    """What filter code hands to render() is the image behind its resize wrapper (every image value is
    RESIZE_IMAGE(STRIP_RESIZE(..)), drawable.c:213-227): of type IMAGE_RESIZE, so render_image's `a float map is its own
    rendering' shortcut (builtins.c:273-274) does not apply and the map is *sampled* into a new one -- each new pixel
    at its own unit coordinates times the wrapper's factors, nearest texel, zeros outside (builtins.c:303-343).  On a
    square frame that is a copy; on a non-square one it resamples (every third row twice, one black row at 96 x 64).
    Until round 3 both the oracle and the HIP path took the shortcut (differently): found by tools/fuzz_native_flow.py."""
    w, h = size
    img = F.synthetic_image(w, h, seed=15)
    src = ("filter f (image in, float s: 0-1 (0.02)) b = gaussian_blur(in, s * 3, s * 1.5); rr = render(b); "
           "c = gaussian_blur(rr, s, s); rr(xy) * 0.5 + c(xy * 0.9) * 0.3 + render(rr)(xy) * 0.2 end")
    src = src.replace("render(rr)(xy)", "r2(xy)").replace("c = gaussian_blur", "r2 = render(rr); c = gaussian_blur")
    flt, inv = make_invocation(src, w, h, {}, {"in": img})
    got = inv.render(t=0.25)
    want = CpuFilter(flt.ir_json_raw).render(w, h, images={"in": img}, t=0.25)
    assert np.array_equal(got, want), stats(got, want)
    plain = "filter f (image in, float s: 0-1 (0.02)) b = gaussian_blur(in, s * 3, s * 1.5); c = gaussian_blur(b, s, s); b(xy) * 0.7 + c(xy * 0.9) * 0.3 end"
    _, pi = make_invocation(plain, w, h, {}, {"in": img})
    assert np.array_equal(got, pi.render(t=0.25)) == (w == h)      # a copy on the square frame only


@pytest.mark.parametrize("seed,rich", [(5, False), (6, False), (21, False), (27, False), (45, False),
                                       (11, True), (17, True), (19, True), (65, True), (157, True)])
def test_generated_filters_of_the_native_flow_fuzzer(seed, rich):
    """A slice of tools/fuzz_native_flow.py in the suite: the seeds its first runs failed on (call sites in loops and
    under pixel-dependent control, loops of both slices, render() behind a resize wrapper, closures on results of natives
    on closures, unassigned image handles), at three (mode, n, t) settings each."""
    import importlib.util
    import os
    from tests.conftest import ROOT
    spec = importlib.util.spec_from_file_location("fuzz_native_flow", os.path.join(ROOT, "tools", "fuzz_native_flow.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    w, h = [(96, 64), (64, 96), (80, 80), (112, 48)][seed % 4] if rich else (96, 64)
    iw, ih = [(w, h), (50, 70), (w, h), (131, 40)][(seed // 4) % 4] if rich else (w, h)
    img = np.ascontiguousarray(F.synthetic_image(iw, ih, seed=3))
    g = fz.Gen(seed, rich)
    src = g.filter()
    assert g.calls_in_loops <= 16
    flt, inv = make_invocation(src, w, h, {}, {"in": img})
    cf = CpuFilter(flt.ir_json_raw)
    fft = "convolve(" in src or "visualize_fft(" in src
    for mode, n, t in ((1, 2, 0.25), (2, 3, 0.75), (0, 1, 0.6)):
        inv.set("mode", mode)
        inv.set("n", n)
        got = inv.render(t=t)
        want = cf.render(w, h, uservals={"mode": mode, "n": n}, images={"in": img}, t=t)
        mx, nd, _ = stats(got, want)
        assert (mx == 0) or (fft and mx <= 1 and nd <= 0.001 * got.size), (seed, rich, mode, n, t, mx, nd)

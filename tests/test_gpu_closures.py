"""Native filters and render() applied to closure images -- render_image's closure branch
(builtins/builtins.c:267-302): the closure's own calc_lines is launched over the whole frame with
floatmap = 1 at frame 0, t = 0.0 (its arguments being values of the main code at the current time), and the native
filter works on that float map.  HIP (through the
C ABI) against the oracle, which runs the closure's code as a filter of its own."""
import numpy as np
import pytest

import mathmap_amd as mm
from tests import filters as F
from oracle.ccgen import CpuFilter
from tests.gpu_util import make_invocation, render_device, stats

pytestmark = pytest.mark.gpu

INNER = """
filter inner (image in, float k: 0-2 (1.0))
  in(xy * k) * 0.8 + rgba:[t * 0.5, 0, 0.1, 0]
end
"""

BLUR_OF_CLOSURE = INNER + """
filter outer (image in, float s: 0-1 (0.03), float k: 0-2 (0.7))
  b = gaussian_blur(inner(in, k), s, s);
  b(xy)
end
"""

RENDER_OF_CLOSURE = INNER + """
filter outer (image in, float k: 0-2 (0.7))
  rendered = render(inner(in, k));
  rendered(xy * 0.9) * 0.5 + rendered(xy) * 0.5
end
"""

CONVOLVE_OF_CLOSURE = INNER + """
filter outer (image in, image kernel, float k: 0-2 (0.7))
  c = convolve(inner(in, k), kernel, 1, 0);
  c(xy)
end
"""

TWO_CLOSURES = INNER + """
filter tint (image in, float g: 0-1 (0.5))
  in(xy) * rgba:[1, g, 1, 1]
end

filter outer (image in, float s: 0-1 (0.02), float k: 0-2 (0.7))
  first = gaussian_blur(inner(in, k), s, s);
  second = gaussian_blur(tint(in, k * 0.5), s * 2, s);
  first(xy) * 0.5 + second(xy) * 0.5
end
"""


# closures that meet their native filter inside conditionals (two levels; the condition is a user value, so the
# branches survive into the kernel): render_image runs where the native call runs
CONDITIONAL = INNER + """
filter outer (image in, float s: 0-1 (0.03), int mode: 0-2 (%d))
  if mode > 0 then
    if mode > 1 then
      b = gaussian_blur(inner(in, 0.5 + t), s * 2, s);
      b(xy) * 0.5
    else
      b = gaussian_blur(inner(in, 0.75), s, s);
      b(xy)
    end
  else
    in(xy)
  end
end
"""


def test_closure_handed_to_a_native_filter_inside_a_conditional():
    """Every mode against the oracle, and against the same blur written without the conditional."""
    w, h = 160, 96
    img = F.synthetic_image(w, h, seed=3)
    flt, inv = make_invocation(CONDITIONAL % 1, w, h, {}, {"in": img})
    cf = CpuFilter(flt.ir_json_raw)
    for mode, k, sx, gain in ((0, None, None, None), (1, "0.75", 1, 1.0), (2, "0.5 + t", 2, 0.5)):
        inv.set("mode", mode)
        got = inv.render(t=0.25)
        want = cf.render(w, h, uservals={"mode": mode}, images={"in": img}, t=0.25)
        assert np.array_equal(got, want), (mode, stats(got, want))
        if mode:
            flat = INNER + "filter flat (image in, float s: 0-1 (0.03)) b = gaussian_blur(inner(in, %s), s * %d, s); b(xy) * %g end" % (k, sx, gain)
            _, fi = make_invocation(flat, w, h, {}, {"in": img})
            assert np.array_equal(got, fi.render(t=0.25)), mode
        else:
            _, ii = make_invocation("ident", w, h, {}, {"in": img})
            assert np.array_equal(got, ii.render())
        # the variant with the user values baked in (the conditionals fold away) renders the same bytes
        _, si = make_invocation(CONDITIONAL % 1, w, h, {"mode": mode}, {"in": img}, specialize=True)
        assert np.array_equal(si.render(t=0.25), got), mode


# a closure whose own body calls a native filter: the closure's calc_lines runs its init_frame first, and that is
# where its gaussian_blur happens (builtins.c:273-298, new_template.c.in:314-337)
BLUR_OF_BLURRING_CLOSURE = """
filter soft (image in, float s: 0-1 (0.02), float k: 0-2 (0.8))
  b = gaussian_blur(in, s, s * 2);
  b(xy * k) * 0.9
end

filter outer (image in, float s: 0-1 (0.03))
  c = gaussian_blur(soft(in, s * 0.5, 0.8 + t * 0.1), s, s);
  c(xy)
end
"""


def test_closure_that_calls_a_native_filter_itself():
    w, h = 160, 96
    img = F.synthetic_image(w, h, seed=3)
    flt, inv = make_invocation(BLUR_OF_BLURRING_CLOSURE, w, h, {}, {"in": img})
    cf = CpuFilter(flt.ir_json_raw)
    for t, s in ((0.25, 0.03), (0.8, 0.05), (0.25, 0.03)):
        inv.set("s", s)
        got = inv.render(t=t)
        want = cf.render(w, h, uservals={"s": s}, images={"in": img}, t=t)
        assert np.array_equal(got, want), (t, s, stats(got, want))
    # float-map output of the same (no byte quantisation in between)
    gm = render_device(inv, w, h, floatmap=True, t=0.25)
    wm = cf.render(w, h, uservals={"s": 0.03}, images={"in": img}, t=0.25, floatmap=True)
    assert np.array_equal(gm, wm)


@pytest.mark.parametrize("name,src,tol", [("blur", BLUR_OF_CLOSURE, 0), ("render", RENDER_OF_CLOSURE, 0),
                                          ("two", TWO_CLOSURES, 0)])
def test_native_filter_on_closure_image(name, src, tol):
    w, h = 212, 131
    img = F.synthetic_image(w, h, seed=3)
    flt, inv = make_invocation(src, w, h, {}, {"in": img})
    cf = CpuFilter(flt.ir_json_raw)
    for k, t in ((0.7, 0.0), (1.3, 0.6), (1.3, 0.2), (0.4, 0.9)):
        inv.set("k", k)
        got = inv.render(t=t)
        want = cf.render(w, h, uservals={"k": k}, images={"in": img}, t=t)
        mx, nd, n1 = stats(got, want)
        assert mx <= tol, (name, k, t, mx, nd, n1)
    # the closure is rendered at t = 0 whatever the frame's t (invocation_new_frame(invocation, image, 0, 0.0))
    if name == "blur":
        assert np.array_equal(inv.render(t=0.1), inv.render(t=0.8))
    # row bands and a float-map render go through the same path
    full = inv.render(t=0.3)
    banded = render_device(inv, w, h, rows=[(0, 50), (50, 51), (51, h)], t=0.3)
    assert np.array_equal(full, banded)


TIMED_ARG = F.CLOSURE_TIMED_ARG


def test_closure_arguments_are_values_of_the_current_frame_and_its_body_runs_at_t_zero():
    """render_image runs the closure's own calc_lines on a frame with t = 0.0 and frame number 0
    (invocation_new_frame(invocation, image, 0, 0.0), builtins.c:291), but the closure's arguments were evaluated by
    the main filter's code at the current time (backends/cc.c:158-188).  So k * (1 + t) at t = 0.5, k = 0.5 must give
    what the constant 0.75 gives at any t and frame -- and HIP = oracle."""
    w, h = 160, 96
    img = F.synthetic_image(w, h, seed=3)
    flt, inv = make_invocation(TIMED_ARG, w, h, {}, {"in": img})
    got = inv.render(t=0.5, frame=7)
    want = CpuFilter(flt.ir_json_raw).render(w, h, images={"in": img}, t=0.5, frame=7)
    assert np.array_equal(got, want), stats(got, want)
    const = TIMED_ARG.replace("k * (1 + t)", "k")
    f2, i2 = make_invocation(const, w, h, {"k": 0.75}, {"in": img})
    assert np.array_equal(i2.render(t=0.0, frame=0), got)
    assert np.array_equal(i2.render(t=0.9, frame=3), got)


@pytest.mark.parametrize("name", ["blur", "render", "two", "timed_arg", "conditional_else", "conditional_then", "blurring_closure"])
def test_closure_images_for_native_filters_through_the_reference_abi(name):
    """The same through gen_and_load_hip_code: the importer finds the closure images that reach native filters in the
    reference-layout IR, builds each one's render code from the main filter's code plus a call of the closure's own
    filter_code (abi_backend.cpp), and calc_lines must deliver the standalone tier's frame byte for byte."""
    import ctypes as C
    from mathmap_amd._lib import selftest_lib
    src = {"blur": BLUR_OF_CLOSURE, "render": RENDER_OF_CLOSURE, "two": TWO_CLOSURES, "timed_arg": TIMED_ARG,
           "conditional_else": CONDITIONAL % 1, "conditional_then": CONDITIONAL % 2, "blurring_closure": BLUR_OF_BLURRING_CLOSURE}[name]
    w, h = 192, 128
    img = np.ascontiguousarray(F.synthetic_image(w, h, seed=3))
    flt, inv = make_invocation(src, w, h, {}, {"in": img})
    want = inv.render(t=0.25)
    got = np.zeros((h, w, 4), np.uint8)
    rc = selftest_lib().mmhip_selftest_abi_roundtrip(src.encode(), 1, img.ctypes.data_as(C.c_void_p), w, h, 3, w, h, 0.25, 2,
                                                     got.ctypes.data_as(C.c_void_p))
    assert rc == 0, selftest_lib().mmhip_selftest_error().decode()
    assert np.array_equal(got, want), stats(got, want)


def test_native_filter_on_a_recursive_closure():
    """The closure handed to the blur is a recursive filter whose body ends in an `if` (its result values are exit
    phis of a top-level construct) and calls itself at run time: the closure's render kernel calls the main code's
    filter functions (generate_hip(..., functions_of)).  Generic kernel = oracle = kernel with the depth baked in."""
    w, h = 160, 96
    img = F.synthetic_image(w, h, seed=3)
    src = F.RECURSIVE + """
filter blurred_tree (image in, int depth: 1-16 (3), float dev: 0-1 (0.02))
  b = gaussian_blur(tree(in, depth, 0.7), dev, dev);
  b(xy)
end
"""
    flt, inv = make_invocation(src, w, h, {}, {"in": img})
    cf = CpuFilter(flt.ir_json_raw)
    for depth in (1, 3, 6):
        inv.set("depth", depth)
        got = inv.render(t=0.25)
        want = cf.render(w, h, uservals={"depth": depth}, images={"in": img}, t=0.25)
        assert np.array_equal(got, want), (depth, stats(got, want))
        sp, si = make_invocation(src, w, h, {"depth": depth}, {"in": img}, specialize=True)
        assert np.array_equal(si.render(t=0.25), got), depth


def test_convolve_on_closure_image():
    w, h = 96, 64
    img = F.synthetic_image(w, h, seed=3)
    yy, xx = np.mgrid[0:h, 0:w]
    blob = np.exp(-(((xx - w // 2) / 3.0) ** 2 + ((yy - (h // 2 - 1)) / 2.0) ** 2))
    kern = np.repeat((blob * 255).astype(np.uint8)[:, :, None], 3, axis=2)
    flt, inv = make_invocation(CONVOLVE_OF_CLOSURE, w, h, {}, {"in": img, "kernel": kern})
    got = inv.render(t=0.4)
    want = CpuFilter(flt.ir_json_raw).render(w, h, images={"in": img, "kernel": kern}, t=0.4)
    assert stats(got, want)[0] <= 1, stats(got, want)


def test_closure_render_survives_an_ir_round_trip():
    """The IR dump carries the closure's render code ("closure_renders"): a filter rebuilt from it
    (mmhip_compile_ir_json, what the reference-ABI tier and the fixtures use) renders the same frame."""
    w, h = 128, 80
    img = F.synthetic_image(w, h, seed=5)
    flt, inv = make_invocation(BLUR_OF_CLOSURE, w, h, {"k": 1.1}, {"in": img})
    a = inv.render(t=0.5)
    flt2 = mm.Filter("", ir_json=flt.ir_json_raw)
    inv2 = flt2.invoke(w, h)
    inv2.set("k", 1.1)
    inv2.set_image("in", img)
    assert np.array_equal(a, inv2.render(t=0.5))


CHAINED = INNER + """
filter outer (image in, float s: 0-1 (0.02))
  b = gaussian_blur(inner(in, 0.7), s, s * 2);
  c = gaussian_blur(inner(b, 1.2), s * 2, s);
  d = render(inner(c, 0.9));
  c(xy) * 0.5 + d(xy) * 0.3 + b(xy) * 0.2
end
"""


@pytest.mark.parametrize("size", [(128, 128), (160, 96)])
def test_closure_on_the_result_of_a_native_filter_on_a_closure(size):
    """`c = gaussian_blur(inner(b, ..))` with `b = gaussian_blur(inner(in, ..))`: the second closure's render kernel evaluates
    the main code once more to get at its argument `b`, and that code holds the first blur with *its* closure.  The
    runtime recognises a call the main code has already made this frame (same filter, same arguments, images by what
    they refer to) and hands the render kernel the main code's map -- in the reference `b` simply is that image, and
    its cache would answer (native-filters/cache.c:110-147).  Until round 3: refused at run time, and a main-code
    native call without a closure was run a second time."""
    w, h = size
    img = F.synthetic_image(w, h, seed=17)
    flt, inv = make_invocation(CHAINED, w, h, {}, {"in": img})
    cf = CpuFilter(flt.ir_json_raw)
    for t in (0.25, 0.7):
        got = inv.render(t=t)
        want = cf.render(w, h, images={"in": img}, t=t)
        assert np.array_equal(got, want), (t, stats(got, want))

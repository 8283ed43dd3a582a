"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs, against the reference's golden PNGs, and -- at the BASELINE sizes --
through size-independent properties (stripe invariance, row-band composition).

Tolerances: bit-exact where the arithmetic is +,-,*,/ and sqrt only (Mandelbrot, Ident);
<= 1 LSB per 8-bit channel where double libm / float complex functions are involved
(device OCML vs host glibc), as BASELINE.json's north_star states."""
import numpy as np
import pytest

import mathmap_amd as mm
from tests import filters as F
from oracle.ccgen import CpuFilter
from tests.conftest import load_png_rgb
from tests.expectations import Expectations
from tests.gpu_util import float_ulps, render_device

pytestmark = pytest.mark.gpu

# per-case records: tests/golden/expected_gpu_*.json (see tests/expectations.py); unlisted = <= 1 LSB
EXP_ORACLE = Expectations("gpu_vs_oracle")
EXP_GOLDEN = Expectations("gpu_vs_golden")


def hip_render(src, w, h, uservals=None, image=None, t=0.0, **opts):
    """`src`: .mm text, or the name of a filter of tests/filters.py (the reference's filters load from IR fixtures)."""
    flt = F.load(src, **opts) if src in F.NAMES else mm.Filter(src, **opts)
    inv = flt.invoke(w, h)
    for k, v in (uservals or {}).items():
        inv.set(k, v)
    if image is not None:
        inv.set_image("in", image)
    return flt, inv.render(t=t)


def cpu_render(flt, w, h, uservals=None, image=None, t=0.0, intersample=True):
    images = {"in": image} if image is not None else {}
    return CpuFilter(flt.ir_json_raw).render(w, h, uservals=uservals, images=images, t=t, intersample=intersample)


def stats(a, b):
    d = np.abs(a.astype(int) - b.astype(int))
    return int(d.max()), int((d > 0).sum()), int((d > 1).sum())


GOLDEN_CASES = [
    ("mandelbrot", "render_mandelbrot.png", {}, False, 0),
    ("ident", "utilities_ident.png", {}, True, 0),
    ("pond", "distorts_pond.png", {}, True, 1),
    ("droste", "map_droste.png", {}, True, 1),
    ("gaussian_blur", "blur_gaussian_blur.png", {"dev": 0.1}, True, 1),
    ("closure_value", "apply.png", {}, False, 0),
    ("closure_call", "circle.png", {}, True, 0),
    ("closure_arg", "closure.png", {}, True, 0),
    ("nested_calls", "twice.png", {}, True, 0),
    ("visualize_fft", "utilities_visualize_fft.png", {}, True, 1),
]


@pytest.mark.parametrize("name,golden,uv,needs_image,tol", GOLDEN_CASES)
def test_hip_matches_reference_golden(name, golden, uv, needs_image, tol, marlene):
    """256x256, -i: what tests/run_tests.sh of the reference renders."""
    _, got = hip_render(name, 256, 256, uv, marlene if needs_image else None)
    want = load_png_rgb(golden)
    mx, nd, n1 = stats(got[:, :, :3], want)
    assert mx <= tol, "%s: max diff %d (%d values differ, %d by more than 1)" % (name, mx, nd, n1)


@pytest.mark.parametrize("name,uv,tol", [
    ("mandelbrot", {}, 0), ("mandelbrot", {"num_iterations": 100, "pj": 0.3, "ck": -0.2}, 0),
    ("ident", {}, 0), ("pond", {}, 1), ("pond", {"height": 0.2, "wavelength": 0.11}, 1),
    ("droste", {}, 1), ("droste", {"NoTransparency": 1}, 1),
    ("droste", {"ShowGrid": 1, "ShowFrame": 1}, 1), ("gauss_direct", {"hdev": 0.02, "vdev": 0.035}, 1),
])
@pytest.mark.parametrize("size", [(317, 203), (640, 480)])
def test_hip_matches_oracle(name, uv, tol, size):
    """Ragged (non tile-multiple, non-square) sizes on a seeded synthetic image."""
    w, h = size
    img = F.synthetic_image(w, h, seed=3)
    needs = bool(F.image_names(F.load(name)))
    flt, got = hip_render(name, w, h, uv, img if needs else None, t=0.37)
    want = cpu_render(flt, w, h, uv, img if needs else None, t=0.37)
    mx, nd, n1 = stats(got, want)
    case = "oracle/%s/%s/%dx%d" % (name, ",".join("%s=%s" % kv for kv in sorted(uv.items())), w, h)
    EXP_ORACLE.check(case, mx, nd, n1, got.size, default=(tol, 0))


@pytest.mark.parametrize("segments", [None, "auto"])
@pytest.mark.parametrize("size", [(37, 19), (64, 48), (129, 65), (5, 3), (16, 16), (17, 33), (300, 7), (700, 523),
                                  (1037, 650, 11.0, 7.5), (2051, 1500, 20.0, 12.75)])
def test_gauss_iir_float_map_is_bit_exact(size, segments, monkeypatch):
    """The recursive Gaussian keeps the reference's operation order in f64 (gauss.c:126-262), so
    the blurred float map -- read back through float-map output, no byte quantisation -- must
    equal the oracle's bit for bit; sizes below, at and across the 16-step block boundaries of
    the scan kernels.  With MMHIP_GAUSS_SEGMENTS=auto lines of 2 * (22.7 sigma + 2) steps or more
    are split into concurrently swept segments (the last three sizes, and the rows of 300 x 7): a
    segment's warmed-up start agrees with the full sweep to ~1e-14 relative in f64 (the recurrence's
    own rounding-noise floor), so 1e-7 (sigma 7 px) to 3e-6 (sigma 20 px) of the float32 values round
    the other way by one ulp there; tolerance of that mode: <= 1 ulp on <= 1e-5 of the values."""
    if segments:
        monkeypatch.setenv("MMHIP_GAUSS_SEGMENTS", segments)
    else:
        monkeypatch.delenv("MMHIP_GAUSS_SEGMENTS", raising=False)
    import ctypes as C
    from mathmap_amd._lib import lib
    w, h = size[:2]
    sx, sy = size[2:] if len(size) > 2 else (2.0, 1.5)                  # sigma in pixels: IIR path
    img = F.synthetic_image(w, h, seed=11)
    uv = {"hdev": 2 * sx / max(w - 1, 1), "vdev": 2 * sy / max(h - 1, 1)}
    flt = F.load("gauss_direct")
    inv = flt.invoke(w, h)
    for k, v in uv.items():
        inv.set(k, v)
    inv.set_image("in", img)
    dev = lib().mmhip_device_alloc(w * h * 16)
    try:
        inv.render_rows(dev, 0, h, floatmap=True)
        inv.sync()
        got = np.empty((h, w, 4), np.float32)
        assert lib().mmhip_copy_to_host(got.ctypes.data_as(C.c_void_p), C.c_void_p(dev), w * h * 16) == 0
    finally:
        lib().mmhip_device_free(C.c_void_p(dev))
    want = CpuFilter(flt.ir_json_raw).render(w, h, uservals=uv, images={"in": img}, floatmap=True)
    diff = got.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64)
    if not segments or w * h < 100000:
        assert not diff.any(), np.abs(got - want).max()
    else:
        assert np.abs(diff).max() <= 1 and np.count_nonzero(diff) <= max(2, 1e-5 * diff.size), \
            (np.abs(diff).max(), np.count_nonzero(diff), diff.size)


@pytest.mark.parametrize("size,sig", [((300, 200), (3.0, 2.5)), ((129, 65), (2.0, 1.5)), ((1037, 650), (11.0, 7.5))])
def test_gauss_direct_output_equals_pixel_kernel(size, sig, monkeypatch):
    """`soft = gaussian_blur(in, ..); soft(xy)`: the blur's second pass packs the RGBA8 pixels itself and
    the pixel kernel is skipped (hipgen find_direct_native, runtime run_natives).  Same bytes as the
    pixel kernel sampling the float map, and as the oracle.  A whole-frame launch does not even write
    the map the first time an argument set is seen; the second identical frame computes it (and still
    writes the pixels directly), the third is a memo hit and goes through the pixel kernel.  Row
    bands and a region that does not start at column 0 exercise the memo and the offsets of the
    direct write."""
    import ctypes as C
    from mathmap_amd._lib import lib
    w, h = size
    img = F.synthetic_image(w, h, seed=21)
    uv = {"hdev": 2 * sig[0] / (w - 1), "vdev": 2 * sig[1] / (h - 1)}

    def make():
        flt = F.load("gauss_direct")
        inv = flt.invoke(w, h)
        for k, v in uv.items():
            inv.set(k, v)
        inv.set_image("in", img)
        return flt, inv

    def device_render(inv, rows, region=None):
        rx, ry, rw, rh = region or (0, 0, w, h)
        dev = lib().mmhip_device_alloc(rw * rh * 4)
        try:
            for lo, hi in rows:
                inv.render_rows(dev + (lo - ry) * rw * 4, lo, hi, region=region)
            inv.sync()
            out = np.empty((rh, rw, 4), np.uint8)
            assert lib().mmhip_copy_to_host(out.ctypes.data_as(C.c_void_p), C.c_void_p(dev), rw * rh * 4) == 0
        finally:
            lib().mmhip_device_free(C.c_void_p(dev))
        return out

    monkeypatch.delenv("MMHIP_NO_DIRECT_NATIVE", raising=False)
    flt, inv = make()
    got = inv.render()                                     # direct, map not written
    assert inv.direct_native_launches() == 1
    second = inv.render()                                  # same arguments again: direct, map written and memoised
    assert inv.direct_native_launches() == 2
    third = inv.render()                                   # memo hit: pixel kernel on the memoised map
    assert inv.direct_native_launches() == 2
    assert np.array_equal(got, second) and np.array_equal(got, third)
    want = CpuFilter(flt.ir_json_raw).render(w, h, uservals=uv, images={"in": img})
    assert np.array_equal(got, want), stats(got, want)

    monkeypatch.setenv("MMHIP_NO_DIRECT_NATIVE", "1")
    _, inv2 = make()
    plain = inv2.render()
    assert inv2.direct_native_launches() == 0
    assert np.array_equal(got, plain)

    monkeypatch.delenv("MMHIP_NO_DIRECT_NATIVE")
    _, inv3 = make()
    cut = h // 3 + 1
    banded = device_render(inv3, [(0, cut), (cut, h)])     # band 1: map + direct rows; band 2: memo hit, pixel kernel
    assert inv3.direct_native_launches() == 1
    assert np.array_equal(banded, got)

    # one invocation per stripe, each asking the blur for its own rows only (the multi-GPU stripe mode):
    # every stripe computes its window + halo and writes its rows directly
    dev = lib().mmhip_device_alloc(w * h * 4)
    try:
        bounds = [0, h // 3, 2 * h // 3 + 1, h]
        for lo, hi in zip(bounds[:-1], bounds[1:]):
            _, invs = make()
            invs.set_native_row_margin(0)
            invs.render_rows(dev + lo * w * 4, lo, hi)
            invs.sync()
            assert invs.direct_native_launches() == 1
        striped = np.empty((h, w, 4), np.uint8)
        assert lib().mmhip_copy_to_host(striped.ctypes.data_as(C.c_void_p), C.c_void_p(dev), w * h * 4) == 0
    finally:
        lib().mmhip_device_free(C.c_void_p(dev))
    assert np.array_equal(striped, got)

    _, inv4 = make()                                       # a region inside the frame: direct write with offsets
    rx, ry, rw, rh = 16, 5, w - 40, h - 9
    reg = device_render(inv4, [(ry, ry + rh)], region=(rx, ry, rw, rh))
    assert inv4.direct_native_launches() == 1
    assert np.array_equal(reg, got[ry:ry + rh, rx:rx + rw])


def test_gauss_row_stripes_with_local_halo_equal_full_frame():
    """Multi-GPU striping of a blurred frame: each stripe's render fills only its rows
    of the blur map plus a halo of ceil(22.7 sigma) rows computed locally from the replicated
    input -- no exchange -- and must reproduce the full-frame result bit for bit (float map)."""
    import ctypes as C
    from mathmap_amd._lib import lib
    w, h = 640, 1500
    img = F.synthetic_image(w, h, seed=13)
    uv = {"hdev": 2 * 3.0 / (w - 1), "vdev": 2 * 2.5 / (h - 1)}       # sigma 3 px / 2.5 px -> halo 59 rows
    flt = F.load("gauss_direct")

    def render(stripes, margin):
        inv = flt.invoke(w, h)
        for k, v in uv.items():
            inv.set(k, v)
        inv.set_image("in", img)
        inv.set_native_row_margin(margin)
        dev = lib().mmhip_device_alloc(w * h * 16)
        try:
            for lo, hi in stripes:
                inv.render_rows(dev + lo * w * 16, lo, hi, floatmap=True)
            inv.sync()
            out = np.empty((h, w, 4), np.float32)
            assert lib().mmhip_copy_to_host(out.ctypes.data_as(C.c_void_p), C.c_void_p(dev), w * h * 16) == 0
        finally:
            lib().mmhip_device_free(C.c_void_p(dev))
        return out

    full = render([(0, h)], -1)
    from mathmap_amd.striping import stripe_rows
    striped = render([stripe_rows(h, g, 4) for g in range(4)], 0)
    assert np.array_equal(full.view(np.uint32), striped.view(np.uint32)), np.abs(full - striped).max()


@pytest.mark.parametrize("intersample", [True, False])
@pytest.mark.parametrize("ex,ey", [(0, 0), (1, 1), (2, 2), (3, 3), (1, 2), (3, 0), (0, 2), (2, 1)])
def test_edge_behaviours_match_oracle(ex, ey, intersample):
    """apply_edge_behaviour (builtins.c:40-119): COLOR with non-trivial edge colours, WRAP,
    REFLECT, ROTATE in every mix of axes, sampling far outside the image, through the branch-free
    hot fetch and the early-exit one (odd image sizes so the C `%` cases differ): bit-exact."""
    w, h = 160, 96
    img = F.synthetic_image(53, 37, seed=21)
    src = "filter e (image in) in(xy * 2.7 + xy:[0.31, -0.23]) end"
    colors = (0x20406080, 0xC0A01055)
    for extra in ("", " * in(xy * 0.9)" * 0):
        flt = mm.Filter(src, intersample=intersample, edge_x=ex, edge_y=ey)
        inv = flt.invoke(w, h)
        inv.set_image("in", img)
        inv.set_edge_colors(*colors)
        got = inv.render()
        want = CpuFilter(flt.ir_json_raw).render(w, h, images={"in": img}, intersample=intersample, edge=(ex, ey), edge_colors=colors)
        assert np.array_equal(got, want), stats(got, want)
    # the same through the one-pixel kernel shape (large bodies use the early-exit fetch)
    import os
    os.environ["MMHIP_SINGLE_PIXEL"] = "1"
    try:
        flt = mm.Filter(src + " ", intersample=intersample, edge_x=ex, edge_y=ey)
        inv = flt.invoke(w, h)
        inv.set_image("in", img)
        inv.set_edge_colors(*colors)
        assert np.array_equal(inv.render(), want)
    finally:
        del os.environ["MMHIP_SINGLE_PIXEL"]


@pytest.mark.parametrize("intersample", [True, False])
def test_nan_and_huge_coordinates_follow_x86_conversion(intersample):
    """Sampling at NaN, +-inf and beyond-int coordinates: the reference converts with cvttss2si
    (INT_MIN for all of them: outside the image -> edge colour), the device conversion saturates
    and maps NaN to 0 unless told otherwise (mm_f2i); also floor() of such values.  With WRAP/
    REFLECT/ROTATE edges the bilinear weights themselves are garbage then (x - (float)INT_MIN) and
    the reference's byte is the low byte of a 64-bit conversion of ~1e32: the branch-free fetch
    flags such a pixel and the work-item redoes it in the generic loop (mm_x86_byte)."""
    w, h = 128, 64
    img = F.synthetic_image(w, h, seed=8)
    nan = "filter n (image in) q = exp(x * 1000 + 900) * 0; in(xy + xy:[q, 0]) end"
    wild = ("filter n (image in) q = exp(x * 1000 + 900); big = x * 1000000 * 1000000 * 1000000 * 1000000 * 1000000; "
            "k = floor(big) + floor(q * 0); in(xy + xy:[q, 0]) * 0.5 + in(xy:[big, y]) * 0.25 + in(xy * (1 + k * 0)) * 0.25 end")
    colors = (0x30507090, 0xA0B0C0D0)
    cases = [(nan, e) for e in ((0, 0), (1, 1), (2, 2), (3, 3), (1, 2))]
    cases += [(wild, e) for e in ((0, 0), (1, 1), (2, 2), (3, 3), (2, 1))]
    for src, (ex, ey) in cases:
        flt = mm.Filter(src, intersample=intersample, edge_x=ex, edge_y=ey)
        inv = flt.invoke(w, h)
        inv.set_image("in", img)
        inv.set_edge_colors(*colors)
        got = inv.render()
        want = CpuFilter(flt.ir_json_raw).render(w, h, images={"in": img}, intersample=intersample, edge=(ex, ey), edge_colors=colors)
        assert np.array_equal(got, want), (src[:40], ex, ey, stats(got, want))


@pytest.mark.parametrize("inc", [2, 3, 4, 7])
def test_preview_strided_sampling_matches_oracle(inc):
    """drawable_get_pixel_inc > 1 (the GIMP dialog's preview reads a source sampled at fast_image_source_scale,
    mathmap.c:1320-1327): the bilinear fetch takes its taps `inc` apart on the grid of multiples of inc and weighs in
    units of inc (builtins.c:186-216).  The branch-free hot fetch, the pixel that is one fetch, the early-exit fetch
    (Droste; the one-pixel kernel shape), every edge behaviour, NaN / huge coordinates: bit-exact against the oracle;
    and a frame that differs from the full-resolution one (the option is live)."""
    w, h = 160, 96
    img = F.synthetic_image(53, 37, seed=21)
    colors = (0x20406080, 0xC0A01055)
    scaled = "filter e (image in) in(xy * 1.3 + xy:[0.11, -0.07]) end"
    mixed = "filter e (image in) in(xy * 2.7 + xy:[0.31, -0.23]) * 0.75 + in(xy * 0.8) * 0.25 end"
    wild = ("filter n (image in) q = exp(x * 1000 + 900); big = x * 1000000 * 1000000 * 1000000 * 1000000 * 1000000; "
            "in(xy + xy:[q, 0]) * 0.5 + in(xy:[big, y]) * 0.25 + in(xy + xy:[q * 0, 0]) * 0.25 end")
    for src in (scaled, mixed, wild):
        for ex, ey in ((0, 0), (1, 2), (3, 3)):
            flt = mm.Filter(src, edge_x=ex, edge_y=ey, pixel_inc=inc)
            inv = flt.invoke(w, h)
            inv.set_image("in", img)
            inv.set_edge_colors(*colors)
            got = inv.render()
            want = CpuFilter(flt.ir_json_raw).render(w, h, images={"in": img}, edge=(ex, ey), edge_colors=colors, pixel_inc=inc)
            assert np.array_equal(got, want), (src[:40], ex, ey, stats(got, want))
    plain = mm.Filter(scaled).invoke(w, h)
    plain.set_image("in", img)
    flt = mm.Filter(scaled, pixel_inc=inc)
    inv = flt.invoke(w, h)
    inv.set_image("in", img)
    assert not np.array_equal(inv.render(), plain.render())
    # the early-exit fetch: Droste's body, and the one-pixel kernel shape
    big = F.synthetic_image(w, h, seed=4)
    flt = F.load("droste", pixel_inc=inc)
    inv = flt.invoke(w, h)
    inv.set_image("in", big)
    want = CpuFilter(flt.ir_json_raw).render(w, h, images={"in": big}, pixel_inc=inc)
    assert np.array_equal(inv.render(), want), stats(inv.render(), want)
    import os
    os.environ["MMHIP_SINGLE_PIXEL"] = "1"
    try:
        flt = mm.Filter(mixed + " ", edge_x=2, edge_y=1, pixel_inc=inc)
        inv = flt.invoke(w, h)
        inv.set_image("in", img)
        inv.set_edge_colors(*colors)
        want = CpuFilter(flt.ir_json_raw).render(w, h, images={"in": img}, edge=(2, 1), edge_colors=colors, pixel_inc=inc)
        assert np.array_equal(inv.render(), want)
    finally:
        del os.environ["MMHIP_SINGLE_PIXEL"]


@pytest.mark.parametrize("size", [(1, 1), (1, 7), (9, 1), (2, 2), (3, 5), (17, 2)])
def test_degenerate_frame_sizes(size):
    """One-pixel-wide / -high frames: (W-1)/2 = 0 makes the virtual coordinates inf or NaN in the
    reference too; every workload must still agree with the oracle."""
    w, h = size
    img = np.ascontiguousarray(F.synthetic_image(max(w, 2), max(h, 2), seed=3)[:h, :w])
    for name in ("mandelbrot", "ident", "pond", "droste", "gauss_direct"):
        uv = {"hdev": 0.9, "vdev": 0.8} if name == "gauss_direct" else {}
        flt = F.load(name)
        needs = bool(F.image_names(flt))
        inv = flt.invoke(w, h)
        for k, v in uv.items():
            inv.set(k, v)
        if needs:
            inv.set_image("in", img)
        got = inv.render(t=0.2)
        want = CpuFilter(flt.ir_json_raw).render(w, h, uservals=uv, images={"in": img} if needs else {}, t=0.2)
        assert stats(got, want)[0] <= 1, (name, size, stats(got, want))


def test_nearest_sampling_matches_oracle():
    w, h = 300, 200
    img = F.synthetic_image(w, h, seed=5)
    flt, got = hip_render("pond", w, h, {}, img, t=0.1, intersample=False)
    want = cpu_render(flt, w, h, {}, img, t=0.1, intersample=False)
    assert stats(got, want)[0] <= 1


def test_row_bands_compose_to_full_frame():
    """calc_lines on disjoint row bands (mathmap_common.c:991-1003) == one full render."""
    import ctypes as C
    from mathmap_amd._lib import lib
    w, h = 512, 384
    flt = F.load("mandelbrot")
    inv = flt.invoke(w, h)
    full = inv.render()
    dev = lib().mmhip_device_alloc(w * h * 4)
    try:
        bands = [(0, 100), (100, 101), (101, 384)]
        for lo, hi in bands:
            inv.render_rows(dev + lo * w * 4, lo, hi)
        inv.sync()
        out = np.empty((h, w, 4), np.uint8)
        assert lib().mmhip_copy_to_host(out.ctypes.data_as(C.c_void_p), C.c_void_p(dev), w * h * 4) == 0
    finally:
        lib().mmhip_device_free(C.c_void_p(dev))
    assert np.array_equal(out, full)


def test_mandelbrot_8192_stripe_property():
    """BASELINE size: the full 8192x8192 frame, checked against the oracle on sampled row
    bands (the oracle would need minutes for the whole frame)."""
    w = h = 8192
    flt = F.load("mandelbrot")
    inv = flt.invoke(w, h)
    got = inv.render()
    cf = CpuFilter(flt.ir_json_raw)
    for lo in (0, 2048, 4090, 8184):
        want = cf.render(w, h, rows=(lo, lo + 8))
        assert np.array_equal(got[lo:lo + 8], want[lo:lo + 8])
    # symmetric in y for the default parameters: row r mirrors row h-1-r
    assert np.array_equal(got[:64], got[::-1][:64])


@pytest.mark.parametrize("name,bands", [("mandelbrot", 1), ("pond", 3), ("droste", 2), ("gaussian_blur", 1),
                                        ("closure_arg", 2), ("curve_gradient", 2),
                                        # RHS_FILTER statements + the callee's filter_code (run-time filter_$name calls)
                                        ("recursive", 2), ("recursive_data", 1), ("recursive_mutual", 3),
                                        # TYPE_TREE_VECTOR values, RHS_TREE_VECTOR, TREE_VECTOR_NTH / SET_TREE_VECTOR_NTH
                                        ("tree_vector", 2)])
def test_reference_abi_boundary_roundtrip(name, bands, marlene):
    """gen_and_load_hip_code + the returned mathfuncs, driven with reference-layout
    structures (include/mathmap_abi.h) the way mathmap_common.c drives the cc backend,
    must produce the same frame as the standalone C API."""
    import ctypes as C
    from mathmap_amd._lib import lib, selftest_lib
    w, h = 256, 256
    flt = F.load(name)
    src = F.SOURCES.get(name) or flt.ir_json_raw      # the self-test takes .mm text or an IR dump
    needs = bool(F.image_names(flt))
    inv = flt.invoke(w, h)
    if needs:
        inv.set_image("in", marlene)
    if name == "curve_gradient":     # the self-test passes the same tables through curve_t / gradient_t
        inv.set_curve("tone", F.test_curve())
        inv.set_gradient("colors", F.test_gradient())
    want = inv.render(t=0.25)   # gaussian_blur: default dev = 0 -> sigma 0 -> FIR path with both passes skipped
    got = np.zeros((h, w, 4), np.uint8)
    img = np.ascontiguousarray(marlene)
    rc = selftest_lib().mmhip_selftest_abi_roundtrip(src.encode(), 1, img.ctypes.data_as(C.c_void_p) if needs else None,
                                            img.shape[1], img.shape[0], 3, w, h, 0.25, bands,
                                            got.ctypes.data_as(C.c_void_p))
    assert rc == 0, selftest_lib().mmhip_selftest_error().decode()
    assert np.array_equal(got, want)
    # ... and, directly, as the oracle (the two tiers share the engine below the importer: a lowering bug common to both
    # would pass the comparison above)
    tables = {"tone": F.test_curve(), "colors": F.test_gradient()} if name == "curve_gradient" else {}
    oracle = CpuFilter(flt.ir_json_raw).render(w, h, uservals=tables, images={"in": marlene} if needs else {}, t=0.25)
    mx, nd, n1 = stats(got, oracle)
    EXP_ORACLE.check("abi/%s/256x256" % name, mx, nd, n1, got.size, default=(1, 0))


def test_reference_abi_tier_asks_the_host_for_the_preview_stride(marlene):
    """The reference's compiled filters call the host's drawable_get_pixel_inc per bilinear fetch (builtins.c:182-184);
    the HIP backend asks it once per frame and runs the kernel variant of that stride.  The self-test host answers 3,
    then 1 again (the dialog's preview followed by the final render on the same module)."""
    import ctypes as C
    from mathmap_amd._lib import selftest_lib
    w, h = 192, 128
    src = "filter e (image in) in(xy * 1.3 + xy:[0.11, -0.07]) * 0.75 + in(xy * 0.6) * 0.25 end"
    img = np.ascontiguousarray(marlene)

    def through_abi():
        got = np.zeros((h, w, 4), np.uint8)
        rc = selftest_lib().mmhip_selftest_abi_roundtrip(src.encode(), 1, img.ctypes.data_as(C.c_void_p), img.shape[1], img.shape[0], 3,
                                                         w, h, 0.25, 2, got.ctypes.data_as(C.c_void_p))
        assert rc == 0, selftest_lib().mmhip_selftest_error().decode()
        return got

    frames = {}
    try:
        for inc in (3, 1):
            selftest_lib().mmhip_selftest_set_pixel_inc(inc)
            frames[inc] = through_abi()
            flt = mm.Filter(src, pixel_inc=inc)
            want = CpuFilter(flt.ir_json_raw).render(w, h, images={"in": marlene}, t=0.25, pixel_inc=inc)
            assert np.array_equal(frames[inc], want), (inc, stats(frames[inc], want))
    finally:
        selftest_lib().mmhip_selftest_set_pixel_inc(1)
    assert not np.array_equal(frames[1], frames[3])


def test_reference_abi_tier_specialises_from_the_second_frame(marlene):
    """calc_lines driven for three frames of an animation (fixed user values, t advancing): the first
    frame runs the generic kernel, later ones the variant built from the imported IR; the last
    frame must equal the standalone render at its t."""
    import ctypes as C
    import os
    from mathmap_amd._lib import lib, selftest_lib
    w, h = 256, 256
    for name in ("mandelbrot", "pond"):
        flt = F.load(name)
        src = flt.ir_json_raw
        needs = bool(F.image_names(flt))
        inv = flt.invoke(w, h)
        if needs:
            inv.set_image("in", marlene)
        want = inv.render(t=0.5)
        got = np.zeros((h, w, 4), np.uint8)
        img = np.ascontiguousarray(marlene)
        os.environ["MMHIP_SELFTEST_WARM_FRAMES"] = "2"
        try:
            rc = selftest_lib().mmhip_selftest_abi_roundtrip(src.encode(), 1, img.ctypes.data_as(C.c_void_p) if needs else None,
                                                    img.shape[1], img.shape[0], 3, w, h, 0.5, 2,
                                                    got.ctypes.data_as(C.c_void_p))
        finally:
            del os.environ["MMHIP_SELFTEST_WARM_FRAMES"]
        assert rc == 0, selftest_lib().mmhip_selftest_error().decode()
        assert np.array_equal(got, want), name


MATH_PROBE = """
filter probe (float k: 0-100 (1))
  u = x * k; v = y * k;
  rgba:[%s]
end
"""

PROBES = [
    ("sqrt(abs(u))", "hypot-free sqrt", 0),
    # sin / cos of a float: mm_fastmath.h, verified equal to glibc for every float below 2^22
    ("sin(u*7)", "sin", 0), ("cos(v*7)", "cos", 0), ("sin(u*3000000+v)", "sin wide", 0), ("cos(v*4000000+u)", "cos wide", 0),
    ("sin(u*1000000000)", "sin beyond 2^22 (OCML)", 0), ("tan(u)", "tan", 0), ("atan(u*9)", "atan", 0), ("atan(u*9, v*9)", "atan2", 1),
    ("exp(u*3)", "exp", 0), ("exp(v*120)", "exp wide", 0), ("log(abs(u)+0.001)", "log", 0), ("log(abs(u*v)*1000000)", "log wide", 0),
    # hypot: glibc's own arithmetic for two floats (mm_fastmath.h); every one-argument op: equal to glibc for every float
    # (test_unary_libm_equals_glibc_for_every_float)
    ("abs(ri:[u,v])", "hypot", 0), ("abs(ri:[u*1000,v*0.001])", "hypot wide", 0),
    ("asin(u)", "asin", 0), ("acos(v)", "acos", 0), ("asin(u*v*0.001)", "asin small", 0), ("acos(1-abs(u*v)*0.0001)", "acos near 1", 0), ("a", "polar angle", 0), ("(abs(u)+0.01)^(v*3)", "pow", 0), ("(abs(u*v)+0.5)^2", "pow int", 0), ("(abs(u)*40+0.1)^(v*9-2)", "pow wide", 0), ("(u*9)^floor(v*6)", "pow of a negative base", 0),
    ("sinh(u*2)", "sinh", 0), ("cosh(v*2)", "cosh", 0), ("tanh(u*2)", "tanh", 0), ("u % 0.37", "fmod", 0),
]


@pytest.mark.parametrize("expr,label,max_ulp", PROBES)
def test_real_math_float_ulps(expr, label, max_ulp):
    """Raw float outputs (float-map mode) of the real math ops: device OCML double function
    rounded to float vs glibc's.  Bound in float ulps, and 99.9% must be identical."""
    import ctypes as C
    from mathmap_amd._lib import lib
    w, h = 512, 256
    src = MATH_PROBE % ", ".join([expr] * 4)
    flt = mm.Filter(src)
    inv = flt.invoke(w, h)
    dev = lib().mmhip_device_alloc(w * h * 16)
    try:
        inv.render_rows(dev, 0, h, floatmap=True)
        inv.sync()
        got = np.empty((h, w, 4), np.float32)
        assert lib().mmhip_copy_to_host(got.ctypes.data_as(C.c_void_p), C.c_void_p(dev), w * h * 16) == 0
    finally:
        lib().mmhip_device_free(C.c_void_p(dev))
    want = CpuFilter(flt.ir_json_raw).render(w, h, floatmap=True)
    a, b = got[:, :, 0], want[:, :, 0]
    finite = np.isfinite(a) & np.isfinite(b)
    assert np.array_equal(np.isnan(a), np.isnan(b))
    ulps = np.abs(a[finite].view(np.int32).astype(np.int64) - b[finite].view(np.int32).astype(np.int64))
    assert ulps.max() <= max_ulp, "%s: max %d ulps" % (label, ulps.max())
    assert (ulps == 0).mean() > (0.98 if "beyond" in label else 0.999), "%s: only %.5f identical" % (label, (ulps == 0).mean())


ROW_SLICE_FILTERS = [
    # a wave distortion: sin of the row coordinate (and t) is a per-row value
    ("wave", "filter wave (image in, float amp: 0-1 (0.1)) in(xy + xy:[sin(y * 10 + t * 6) * amp, 0]) end", True),
    # per-row and per-column library calls side by side (the per-column one stays in the pixel loop: hoisted by the compiler)
    ("both", "filter both (image in) in(xy + xy:[sin(y * 10) * 0.1, cos(x * 7) * 0.05]) end", True),
    # a complex per-row value: its real and imaginary parts travel, the complex value itself does not
    ("cplx", "filter cplx () w = exp(ri:[0, y * 3]); grayColor(w[0] * x + w[1]) end", True),
    # integer per-row values (floor) and a row value that is the result by itself
    ("ints", "filter ints () k = floor(abs(sin(y * 9)) * 5); rgba:[k / 5, exp(y) / 3, x * k / 5, 1] end", True),
    # nothing worth a table: plain arithmetic on y is recomputed per pixel
    ("none", "filter none (image in) in(xy + xy:[y * 0.1, 0]) end", False),
]


@pytest.mark.parametrize("name,src,has_rows", ROW_SLICE_FILTERS, ids=[f[0] for f in ROW_SLICE_FILTERS])
def test_per_row_slice_matches_oracle(name, src, has_rows, monkeypatch):
    """The per-row (x-const) slice: values that depend on the row alone come from the rows kernel's table.  Whole frames,
    row bands, a region with an offset, float-map output, and the same filter with the slice switched off."""
    w, h = 333, 207
    img = F.synthetic_image(w, h, seed=12)
    flt = mm.Filter(src)
    assert ("mm_rows(mm_args" in flt.kernel_source) == has_rows
    needs = bool(F.image_names(flt))
    images = {"in": img} if needs else {}
    cf = CpuFilter(flt.ir_json_raw)
    inv = flt.invoke(w, h)
    if needs:
        inv.set_image("in", img)
    for t in (0.0, 0.37):
        want = cf.render(w, h, images=images, t=t)
        assert np.array_equal(inv.render(t=t), want), (name, t)
        got = render_device(inv, w, h, rows=[(0, 50), (50, 51), (51, 207)], t=t)
        assert np.array_equal(got, want), (name, "bands")
    gm = render_device(inv, w, h, floatmap=True, t=0.37)
    assert np.array_equal(gm, cf.render(w, h, images=images, t=0.37, floatmap=True))
    monkeypatch.setenv("MMHIP_NO_ROW_SLICE", "1")
    off = mm.Filter(src)
    assert "mm_rows(mm_args" not in off.kernel_source
    oi = off.invoke(w, h)
    if needs:
        oi.set_image("in", img)
    assert np.array_equal(oi.render(t=0.37), want)


def test_unary_libm_equals_glibc_for_every_float():
    """Every real one-argument math op of a float -- sin cos tan asin acos atan exp log sinh cosh tanh asinh acosh atanh, as
    the kernels compute them (csrc/fastmath_selftest.hip calls the functions hipgen.cpp names) -- for EVERY float (2^32
    arguments each), against the host's glibc double function rounded to float (oracle/libm_ref.c: what the reference's
    generated C computes).  0 differences: the table-driven forms (sin cos exp log) are exact by construction, the
    platform's functions (OCML) round like glibc's on all 2^32 arguments except six of asinh / acosh, which
    mm_libm_exceptions.h lists with glibc's value -- so every filter built from these ops, Pond's polar angle included, is
    bit-exact by enumeration."""
    from tools.libm_exceptions import compare_all
    res = compare_all(stride=1, progress=False)
    assert set(res) == {"sin", "cos", "tan", "asin", "acos", "atan", "exp", "log", "sinh", "cosh", "tanh", "asinh", "acosh", "atanh"}
    for name, r in res.items():
        assert r["checked"] == 1 << 32
        assert r["mismatches"] == 0, (name, r["mismatches"], ["0x%08x" % x for x, _ in r["pairs"][:8]])


def test_binary_libm_against_glibc_on_sampled_pairs():
    """Two-argument ops cannot be enumerated: 2^28 pseudo-random pairs each (any two floats; both near 1; the second up to
    30 binades below the first; small integers and halves, either sign) against the host's glibc.  hypot (glibc's own
    arithmetic), pow and fmod must agree everywhere -- the full run of tools/libm_exceptions.py (2^32 pairs each,
    profiles/r03_libm_every_float.json) found pow an ulp off at (-154.5)^3-like ties, since fixed; atan2 is the platform's
    function: 1 differing pair in 2^32."""
    from tools.libm_exceptions import compare_pairs
    res = compare_pairs(runs=16, progress=False)
    for name in ("hypot", "pow", "fmod"):
        assert res[name]["mismatches"] == 0, (name, res[name])
    assert res["atan2"]["mismatches"] <= 2, res["atan2"]


# (expression, max float ulps allowed per component over ALL finite results).  0: the device runs glibc's own
# float algorithm (mm_glibcf.h, verified bit for bit against the host libm by tools/verify_glibcf.c).  cgamma (the
# reference's own spec_func.c: double complex inside, with the float roundings of its `creal(z) < 0` branch) is held
# to 1 ulp -- its double exp / log / sincos / atan2 / hypot are this project's or OCML's, not glibc's; measured: every
# value of the three probes identical (tools/gamma_probe.py).
COMPLEX_PROBES = [("exp(z)", 0), ("log(z)", 0), ("sqrt(z)", 0), ("sin(z)", 0), ("cos(z)", 0), ("tan(z)", 0),
                  ("z^ri:[1.3,0.4]", 0), ("ri:[0.3,-0.8]^z", 0), ("sinh(z)", 0), ("cosh(z)", 0), ("tanh(z)", 0), ("arg(z)", 0),
                  ("asin(z)", 0), ("acos(z)", 0), ("atan(z)", 0), ("asinh(z)", 0), ("acosh(z)", 0),
                  ("atanh(z)", 0), ("gamma(z)", 1)]


@pytest.mark.parametrize("expr,max_ulp", COMPLEX_PROBES, ids=[p[0] for p in COMPLEX_PROBES])
@pytest.mark.parametrize("scale", [3.0, 40.0, 0.01])
def test_complex_math_float_ulps(expr, max_ulp, scale):
    """float-complex functions in float-map mode (raw float outputs) over z = scale * (x + i y): device
    vs glibc.  For the functions restated after glibc (max_ulp = 0) every finite component must be
    identical and the NaN / inf patterns must agree; cgamma may differ by one ulp in at most 1e-4 of the values."""
    w, h = 256, 256
    if expr == "arg(z)":
        src = "filter probe () z = ri:[x*%g, y*%g]; w = arg(z); rgba:[w, w, w, w] end" % (scale, scale)
    else:
        src = "filter probe () z = ri:[x*%g, y*%g]; w = %s; rgba:[w[0], w[1], w[0], w[1]] end" % (scale, scale, expr)
    flt = mm.Filter(src)
    inv = flt.invoke(w, h)
    got = render_device(inv, w, h, floatmap=True)
    want = CpuFilter(flt.ir_json_raw).render(w, h, floatmap=True)
    ulps = float_ulps(got[:, :, :2], want[:, :, :2])
    assert ulps.max() <= max_ulp, "%s: max %d ulps, %d of %d values differ" % (expr, ulps.max(), (ulps > 0).sum(), ulps.size)
    assert (ulps == 0).mean() >= 0.9999, "%s: only %.6f identical" % (expr, (ulps == 0).mean())


@pytest.mark.parametrize("hdev,vdev", [(0.001, 0.001), (0.0012, 0.04), (0.05, 0.0009), (0.0, 0.02)])
@pytest.mark.parametrize("flat", [False, True])
def test_gauss_fir_path_matches_oracle(hdev, vdev, flat):
    """sigma < 0.5 px on an axis selects the reference's FIR/RLE blur (gauss.c:500-639); a mostly
    flat image additionally selects its run-length branch (do_encoded_lre) per line."""
    w, h = 333, 251
    if flat:
        img = np.full((h, w, 3), 40, np.uint8)
        img[100:140, 50:300] = (200, 120, 30)
        img[:, 170:173] = 255
    else:
        img = F.synthetic_image(w, h, seed=9)
    uv = {"hdev": hdev, "vdev": vdev}
    flt, got = hip_render("gauss_direct", w, h, uv, img)
    want = cpu_render(flt, w, h, uv, img)
    assert stats(got, want)[0] <= 1


@pytest.mark.parametrize("bpp", [1, 2, 3])
def test_output_bpp_variants(bpp):
    """output_bpp 1..3 of calc_lines (grey, grey+alpha, RGB; new_template.c.in:279-293)."""
    import ctypes as C
    from mathmap_amd._lib import lib
    w, h = 200, 120
    img = F.synthetic_image(w, h, seed=2)
    flt = F.load("pond")
    inv = flt.invoke(w, h)
    inv.set_image("in", img)
    dev = lib().mmhip_device_alloc(w * h * bpp)
    try:
        inv.render_rows(dev, 0, h, t=0.2, bpp=bpp)
        inv.sync()
        got = np.empty((h, w, bpp), np.uint8)
        assert lib().mmhip_copy_to_host(got.ctypes.data_as(C.c_void_p), C.c_void_p(dev), w * h * bpp) == 0
    finally:
        lib().mmhip_device_free(C.c_void_p(dev))
    want = CpuFilter(flt.ir_json_raw).render(w, h, images={"in": img}, t=0.2, bpp=bpp)
    assert np.abs(got.astype(int) - want.astype(int)).max() <= 1


@pytest.mark.parametrize("intersample", [False, True])
def test_supersampling_matches_oracle(intersample):
    """-o: two slices (offset 0 and -0.5, the long one a column wider) combined 1-1-2-1-1 / 6."""
    import ctypes as C
    from mathmap_amd._lib import lib
    from oracle.ccgen import render_supersampled
    w, h = 211, 157
    img = F.synthetic_image(w, h, seed=4)
    flt = F.load("pond", intersample=intersample, supersampling=True)
    inv = flt.invoke(w, h)
    inv.set_image("in", img)
    dev = lib().mmhip_device_alloc(w * h * 4)
    try:
        inv.render_supersampled(dev, t=0.3)
        inv.sync()
        got = np.empty((h, w, 4), np.uint8)
        assert lib().mmhip_copy_to_host(got.ctypes.data_as(C.c_void_p), C.c_void_p(dev), w * h * 4) == 0
    finally:
        lib().mmhip_device_free(C.c_void_p(dev))
    want = render_supersampled(CpuFilter(flt.ir_json_raw), w, h, images={"in": img}, t=0.3, intersample=intersample)
    assert np.abs(got.astype(int) - want.astype(int)).max() <= 1


def test_command_line_reproduces_golden(tmp_path, marlene):
    """The `mathmap` command line (mathmap_hip_cli): -i -f script -Din=png out.png, as the
    reference's tests/run_tests.sh invokes it; PNG decode/encode through the bundled codec."""
    import subprocess
    from PIL import Image
    from tests.conftest import GOLDEN, ROOT
    import os
    cli = os.path.join(ROOT, "mathmap_amd", "mathmap_hip_cli")
    script = tmp_path / "pond.mm"
    script.write_text(F.ir_text("pond"))      # the CLI takes an IR dump in place of .mm text
    out = tmp_path / "out.png"
    r = subprocess.run([cli, "-i", "-f", str(script), "-Din=" + os.path.join(GOLDEN, "marlene.png"), str(out)],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    got = np.array(Image.open(out))
    assert got.shape == (256, 256, 3)
    assert np.abs(got.astype(int) - load_png_rgb("distorts_pond.png").astype(int)).max() <= 1
    # render test with -s and -D for a scalar user value
    script2 = tmp_path / "mandel.mm"
    script2.write_text(F.ir_text("mandelbrot"))
    out2 = tmp_path / "m.png"
    r = subprocess.run([cli, "-i", "-s", "256x256", "-f", str(script2), str(out2)], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    assert np.array_equal(np.array(Image.open(out2)), load_png_rgb("render_mandelbrot.png"))


@pytest.mark.parametrize("name,uv", [
    ("mandelbrot", {}), ("mandelbrot", {"num_iterations": 77, "pj": 0.25}), ("mandelbrot", {"ci": 0.3, "ck": -0.4}),
    ("droste", {}), ("droste", {"NoTransparency": 1, "Zoom": 3}), ("droste", {"ShowGrid": 1, "ShowFrame": 1, "Untwist": 1}),
    ("pond", {"height": 0.0}), ("pond", {"wavelength": 0.0}), ("closure_call", {"radius": 0.0}),
])
def test_userval_specialisation_is_bit_identical(name, uv):
    """The specialising JIT (scalar user values baked in, reference-style literal folds,
    optimistic constant propagation through loop phis) must not change a single byte."""
    w, h = 400, 300
    img = F.synthetic_image(w, h, seed=6) if F.image_names(F.load(name)) else None
    _, generic = hip_render(name, w, h, uv, img, t=0.4)
    _, special = hip_render(name, w, h, uv, img, t=0.4, specialize=True)
    assert np.array_equal(generic, special)
    flt = F.load(name, specialize=True)
    # changing a user value after the first render must re-specialise
    inv = flt.invoke(w, h)
    if img is not None:
        inv.set_image("in", img)
    first = inv.render(t=0.4)
    for k, v in uv.items():
        inv.set(k, v)
    assert np.array_equal(inv.render(t=0.4), special)
    assert first.shape == special.shape


def test_ir_origin_filters_specialise_too():
    """A filter that arrives as IR (reference-ABI tier, mmhip_compile_ir_json) has no source to
    re-lower; its variants are built from its own IR dump (bake_uservals + the same constant
    propagation) and must equal the generic kernel byte for byte."""
    w, h = 320, 200
    img = F.synthetic_image(w, h, seed=4)
    for name, uv in (("mandelbrot", {"num_iterations": 40, "pj": 0.2}), ("pond", {"height": 0.1}), ("droste", {})):
        outs = []
        for spec in (False, True):
            flt = F.load(name, specialize=spec)
            inv = flt.invoke(w, h)
            for k, v in uv.items():
                inv.set(k, v)
            if F.image_names(flt):
                inv.set_image("in", img)
            outs.append(inv.render(t=0.2))
        assert np.array_equal(outs[0], outs[1]), name


def test_specialised_mandelbrot_8192_equals_generic():
    w = h = 8192
    a = F.load("mandelbrot").invoke(w, h).render()
    b = F.load("mandelbrot", specialize=True).invoke(w, h).render()
    assert np.array_equal(a, b)


GSL_PROBES = [
    # v2 / m2x2 and v3 / m3x3: gsl_linalg_HH_solve (opmacros.h:84-96)
    ("q = v2:[x + 2, y - 1] / m2x2:[x + 2.5, y, 0.3, y + 1.5]; rgba:[q[0], q[1], q[0], q[1]]", 2),
    ("q = v3:[x, y, 1] / m3x3:[2 + x, y, 0.1, 0.3, 1.5 + y, x, 0.2, 0.1, 3]; rgba:[q[0], q[1], q[2], q[0]]", 2),
    ("q = v2:[1, 2] / m2x2:[1, 2, 2, 4]; rgba:[q[0], q[1], 0, 1]", 0),       # singular: x as far as HH_svx got
    # gsl_sf_elljac_e (opmacros.h:118-126), real and complex argument
    ("rgba:[ell_jac_sn(x * 3, 0.5), ell_jac_cn(y * 3, 0.3), ell_jac_dn(x * y * 4, 0.8), ell_jac_sn(x, 1.5)]", 4),
    ("w = ell_jac_cn(ri:[x * 2, y * 2], 0.5); rgba:[w[0], w[1], w[0], w[1]]", 16),
    # gsl_sf_ellint_* (opmacros.h:102-117): complete, Legendre (phi beyond pi/2: the periodic terms) and Carlson
    # forms, domain errors (k^2 >= 1, negative arguments) as NaN
    ("rgba:[ell_int_Kcomp(x * 0.99), ell_int_Ecomp(y * 0.99), ell_int_Kcomp(x * 1.2), ell_int_Ecomp(0.99999999)]", 4),
    ("rgba:[ell_int_F(y * 4, x * 0.9), ell_int_E(y * 4, x * 0.9), ell_int_P(y * 4, x * 0.9, 0.3), ell_int_D(y * 4, x * 0.9, 0)]", 8),
    ("rgba:[ell_int_RC(x + 1.2, y + 1.1), ell_int_RD(x + 1.2, y + 1.1, 0.7), ell_int_RF(x + 1.2, y + 1.1, 0.7), "
     "ell_int_RJ(x + 1.2, y + 1.1, 0.7, 2.5)]", 4),
    ("rgba:[ell_int_RC(x, y), ell_int_RF(x, y, x * y), ell_int_RD(x + 1, y, 0.5), ell_int_RJ(x + 1, y + 1, 1, x)]", 8),
]


@pytest.mark.parametrize("body,max_ulp", GSL_PROBES)
def test_gsl_operators_match_restatement(body, max_ulp):
    """SOLVE_LINEAR_2/3, ELL_JAC and ELL_INT_*: device vs the oracle build of the same restated GSL algorithms
    (mm_gslmath.h; GSL itself is absent -- parity with the reference is unpinned).  Float-map
    output; differences come from OCML vs glibc sqrt/sin/cos/hypot inside the algorithms."""
    import ctypes as C
    from mathmap_amd._lib import lib
    w, h = 192, 128
    flt = mm.Filter("filter probe () %s end" % body)
    inv = flt.invoke(w, h)
    dev = lib().mmhip_device_alloc(w * h * 16)
    try:
        inv.render_rows(dev, 0, h, floatmap=True)
        inv.sync()
        got = np.empty((h, w, 4), np.float32)
        assert lib().mmhip_copy_to_host(got.ctypes.data_as(C.c_void_p), C.c_void_p(dev), w * h * 16) == 0
    finally:
        lib().mmhip_device_free(C.c_void_p(dev))
    want = CpuFilter(flt.ir_json_raw).render(w, h, floatmap=True)
    finite = np.isfinite(got) & np.isfinite(want)
    assert np.array_equal(np.isfinite(got), np.isfinite(want))
    ulps = np.abs(got[finite].view(np.int32).astype(np.int64) - want[finite].view(np.int32).astype(np.int64))
    assert ulps.max() <= max_ulp, ulps.max()
    assert (ulps == 0).mean() > 0.98


def test_rand_is_deterministic_and_stripe_invariant():
    """rand(a, b): a counter-based hash of (column, row, frame, call number) stands in for the
    reference's clock-seeded global generator (mm_gslmath.h).  Device == oracle build of the same
    hash, bit for bit; rendering in stripes changes nothing; two calls in one pixel differ; the
    values fill [a, b)."""
    import ctypes as C
    from mathmap_amd._lib import lib
    w, h = 256, 192
    flt = mm.Filter("filter r () u = rand(-2, 3); v = rand(0, 1); rgba:[u, v, rand(10, 11), u - v] end")
    inv = flt.invoke(w, h)

    def grab(stripes):
        dev = lib().mmhip_device_alloc(w * h * 16)
        try:
            for lo, hi in stripes:
                inv.render_rows(dev + lo * w * 16, lo, hi, floatmap=True)
            inv.sync()
            out = np.empty((h, w, 4), np.float32)
            assert lib().mmhip_copy_to_host(out.ctypes.data_as(C.c_void_p), C.c_void_p(dev), w * h * 16) == 0
        finally:
            lib().mmhip_device_free(C.c_void_p(dev))
        return out

    full = grab([(0, h)])
    assert np.array_equal(full, grab([(0, 50), (50, 51), (51, h)]))
    want = CpuFilter(flt.ir_json_raw).render(w, h, floatmap=True)
    assert np.array_equal(full.view(np.uint32), want.view(np.uint32))
    u, v, t3 = full[:, :, 0], full[:, :, 1], full[:, :, 2]
    assert u.min() >= -2 and u.max() < 3 and v.min() >= 0 and v.max() < 1 and t3.min() >= 10 and t3.max() <= 11
    assert abs(float(u.mean()) - 0.5) < 0.02 and abs(float(v.mean()) - 0.5) < 0.01
    assert abs(np.corrcoef(u.ravel(), v.ravel())[0, 1]) < 0.02
    assert abs(np.corrcoef(v[:, :-1].ravel(), v[:, 1:].ravel())[0, 1]) < 0.02


def test_dynamic_subscripts_on_gpu():
    """Tree vectors (dynamic tuple subscripts) are lowered to element variables and select chains:
    HIP == oracle bit for bit, for several values of the run-time index user value."""
    import ctypes as C
    from mathmap_amd._lib import lib
    w, h = 192, 96
    flt = F.load("tree_vector")
    for k in (0, 2, 3, 7):
        inv = flt.invoke(w, h)
        inv.set("k", k)
        dev = lib().mmhip_device_alloc(w * h * 16)
        try:
            inv.render_rows(dev, 0, h, floatmap=True)
            inv.sync()
            got = np.empty((h, w, 4), np.float32)
            assert lib().mmhip_copy_to_host(got.ctypes.data_as(C.c_void_p), C.c_void_p(dev), w * h * 16) == 0
        finally:
            lib().mmhip_device_free(C.c_void_p(dev))
        want = CpuFilter(flt.ir_json_raw).render(w, h, uservals={"k": k}, floatmap=True)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), k


def test_curve_and_gradient_user_values(marlene):
    """Curve / gradient LUT user values (APPLY_CURVE / APPLY_GRADIENT, opmacros.h:192-194): default
    ramps and explicitly set tables, HIP vs oracle, bit-exact."""
    w = h = 128
    img = np.ascontiguousarray(marlene[:h, :w])
    flt = F.load("curve_gradient")
    for tables in ({}, {"tone": F.test_curve(), "colors": F.test_gradient()}):
        inv = flt.invoke(w, h)
        inv.set_image("in", img)
        if tables:
            inv.set_curve("tone", tables["tone"])
            inv.set_gradient("colors", tables["colors"])
        got = inv.render()
        want = CpuFilter(flt.ir_json_raw).render(w, h, uservals=tables, images={"in": img})
        assert np.array_equal(got, want), stats(got, want)
    assert not np.array_equal(got, F.load("ident").invoke(w, h).render())


def _fft_case(src, w, h, uv, images):
    flt = F.load(src) if src in F.NAMES else mm.Filter(src)
    inv = flt.invoke(w, h)
    for k, v in uv.items():
        inv.set(k, v)
    for k, v in images.items():
        inv.set_image(k, v)
    got = inv.render()
    want = CpuFilter(flt.ir_json_raw).render(w, h, uservals=uv, images=images)
    return got, want


@pytest.mark.parametrize("w,h", [(96, 64), (75, 50), (64, 33)])
def test_fft_native_filters_match_oracle(w, h):
    """convolve / half_convolve / visualize_fft (hipFFT double + hand-written kernels) against
    the oracle's direct long-double DFT, even and odd sizes, every flag combination.
    Tolerance: the reference itself uses FFTW, whose round-off differs from any other FFT's
    by O(1e-15) relative; after /n, the float store and the byte pack that is <= 1 LSB."""
    img = F.synthetic_image(w, h, seed=3)
    yy, xx = np.mgrid[0:h, 0:w]
    blob = np.exp(-(((xx - w // 2) / 3.0) ** 2 + ((yy - (h // 2 - 1)) / 2.0) ** 2))
    kern = np.repeat((blob * 255).astype(np.uint8)[:, :, None], 3, axis=2)
    kern[:, :, 1] = kern[:, :, 1] // 2
    mask = F.synthetic_image(w, h, seed=9)
    for normalize in (0, 1):
        for copy_alpha in (0, 1):
            got, want = _fft_case("convolve", w, h, {"normalize": normalize, "copy_alpha": copy_alpha},
                                  {"in": img, "kernel": kern})
            mx, nd, n1 = stats(got, want)
            assert mx <= 1, ("convolve", w, h, normalize, copy_alpha, mx, nd, n1)
    for copy_alpha in (0, 1):
        got, want = _fft_case("half_convolve", w, h, {"copy_alpha": copy_alpha}, {"in": img, "mask": mask})
        mx, nd, n1 = stats(got, want)
        assert mx <= 1, ("half_convolve", w, h, copy_alpha, mx, nd, n1)
    for ignore_alpha in (0, 1):
        got, want = _fft_case("visualize_fft", w, h, {"ignore_alpha": ignore_alpha}, {"in": img})
        mx, nd, n1 = stats(got, want)
        assert mx <= 1, ("visualize_fft", w, h, ignore_alpha, mx, nd, n1)


def test_convolve_with_impulse_is_identity_at_2048():
    """Size-independent property at a size the oracle cannot reach: convolving with the unit
    impulse at flat index n - nhalf (convolve.c:119-122) returns the input (<= 1 LSB), and
    the native-filter memo returns the same map on a second frame."""
    w = h = 2048
    img = F.synthetic_image(w, h, seed=5)
    kern = np.zeros((h, w, 3), np.uint8)
    kern[h // 2 - 1, w // 2] = 255
    flt = F.load("convolve")
    inv = flt.invoke(w, h)
    inv.set("normalize", 1)
    inv.set_image("in", img)
    inv.set_image("kernel", kern)
    got = inv.render()
    ident = F.load("ident").invoke(w, h)
    ident.set_image("in", img)
    want = ident.render()
    mx, nd, n1 = stats(got, want)
    assert mx <= 1, (mx, nd, n1)
    assert np.array_equal(inv.render(t=0.5), got)


def test_recursive_filter_calls_filter_functions_on_the_gpu():
    """Run-time filter_$name calls (RHS_FILTER): one generic kernel, the recursion depth is a user value read
    per render.  Bit-exact against the oracle's recursive C functions, and against the kernel with the
    depth baked in (recursion unrolled while lowering, no calls)."""
    w, h = 96, 64
    img = F.synthetic_image(w, h, seed=2)
    flt = F.load("recursive")
    assert "mm_filter_0<0>" in flt.kernel_source
    cpu = CpuFilter(flt.ir_json_raw)
    inv = flt.invoke(w, h)
    inv.set_image("in", img)
    outs = []
    for depth in (1, 3, 5, 16):
        inv.set("depth", depth)
        got = inv.render()
        want = cpu.render(w, h, uservals={"depth": depth}, images={"in": img})
        assert np.array_equal(got, want), (depth, stats(got, want))
        if depth <= 5:
            sp = flt.specialized({"depth": depth})
            assert "mm_filter_" not in sp.kernel_source
            si = sp.invoke(w, h)
            si.set_image("in", img)
            si.set("depth", depth)
            assert np.array_equal(si.render(), got), depth
        outs.append(got)
    assert not np.array_equal(outs[0], outs[1]) and not np.array_equal(outs[1], outs[2])


PACK_RAMP = """
filter ramp (float spread: 0-1 (0.00001))
  kk = floor((x + 1) * 128.5);
  off = y * spread;
  c = (kk + off) / 255;
  rgba:[c, 1 - c, c * c, (kk - 3 + off * 7) / 249]
end
"""


def test_rgba8_pack_and_fetched_pixel_store_on_values_around_every_byte_boundary():
    """The pack without f64 (one round-toward-zero fma per channel, mm_pack_rgba8) against the template's
    (unsigned char)(c * 255.0), which the oracle executes: channel values within a few ulps of every k / 255, on
    both sides, and values outside [0, 1] and NaN-free extremes.  Then the store of an unchanged bilinear fetch
    (mm_store_fetched_pixel: the fetch's rounded byte sums, no division / product) for all 256 byte values in
    every channel, at fractional offsets."""
    w, h = 1024, 512
    for spread in (1e-5, 3e-7, 0.4):
        flt, got = hip_render(PACK_RAMP, w, h, {"spread": spread})
        want = cpu_render(flt, w, h, {"spread": spread})
        assert np.array_equal(got, want), (spread, stats(got, want))
        assert len(np.unique(got[..., 0])) >= 250
    # every byte value in every channel, sampled between texel centres
    img = np.zeros((64, 256, 4), np.uint8)
    v = np.arange(256, dtype=np.uint8)[None, :]
    img[..., 0] = v; img[..., 1] = 255 - v; img[..., 2] = (v.astype(int) * 7 % 256).astype(np.uint8); img[..., 3] = (v[:, ::-1] // 2 + 64)
    img[1::2] = img[1::2, ::-1]
    src = "filter shift (image in, float dx: -1-1 (0.001), float dy: -1-1 (0.002))\n  in(xy + xy:[dx, dy])\nend\n"
    for dx, dy in ((0.0, 0.0), (0.0013, 0.0021), (-0.004, 0.0097), (0.3, -0.2)):
        flt = mm.Filter(src)
        assert "mm_store_fetched_pixel(A, rl_raw" in flt.kernel_source
        inv = flt.invoke(256, 64)
        inv.set_image("in", img)
        inv.set("dx", dx); inv.set("dy", dy)
        got = inv.render()
        want = CpuFilter(flt.ir_json_raw).render(256, 64, uservals={"dx": dx, "dy": dy}, images={"in": img})
        assert np.array_equal(got, want), ((dx, dy), stats(got, want))
        # the same kernel asked for a float map (and for RGB): the fetched sums go through k / 255 after all
        from tests.gpu_util import render_device
        gf = render_device(inv, 256, 64, floatmap=True)
        wf = CpuFilter(flt.ir_json_raw).render(256, 64, uservals={"dx": dx, "dy": dy}, images={"in": img}, floatmap=True)
        assert np.array_equal(gf.view(np.uint32), wf.view(np.uint32)), (dx, dy)
        g3 = render_device(inv, 256, 64, bpp=3)
        w3 = CpuFilter(flt.ir_json_raw).render(256, 64, uservals={"dx": dx, "dy": dy}, images={"in": img}, bpp=3)
        assert np.array_equal(g3, w3), (dx, dy)


def test_data_dependent_recursion_on_the_gpu():
    """The recursion depth differs per pixel (it follows the image content and the position), which no
    lowering-time unrolling could serve; mutual recursion between two filters; RAND inside the callee keeps the
    pixel's call counter.  Bit-exact against the oracle, including the cut-off at MM_MAX_CALL_DEPTH."""
    w, h = 128, 96
    img = F.synthetic_image(w, h, seed=5)
    cases = [(F.RECURSIVE_DATA, {}), (F.RECURSIVE_MUTUAL, {"n": 6}), (F.RECURSIVE_MUTUAL, {"n": 40})]
    for src, uv in cases:
        flt = mm.Filter(src)
        inv = flt.invoke(w, h)
        inv.set_image("in", img)
        for k, v in uv.items():
            inv.set(k, v)
        got = inv.render(t=0.3)
        want = CpuFilter(flt.ir_json_raw).render(w, h, uservals=uv, images={"in": img}, t=0.3)
        assert np.array_equal(got, want), (uv, stats(got, want))
        assert got[..., :3].any()


@pytest.mark.parametrize("seed", range(80))
def test_random_filters_hip_vs_oracle_and_specialised_vs_generic(seed):
    """Differential fuzzing (tests/fuzz_filters.py): a random filter with loops, conditionals, libm
    calls and image fetches is printed twice from the same IR -- as a HIP kernel (prologue/pixel
    split, hoisting, unrolled hot variant; after all passes) and as C by the oracle (the IR before
    any pass) -- and must agree within 1 LSB unless tests/golden/expected_gpu_vs_oracle.json records
    more for that seed.  The user-value specialised kernel must equal the generic one byte for byte."""
    from tests.fuzz_filters import make_filter
    src, needs = make_filter(seed)
    w, h = 96, 64
    img = F.synthetic_image(w, h, seed=1)
    uv = {"k": 5, "m": 1.3}
    outs = []
    for spec in (False, True):
        flt = mm.Filter(src, specialize=spec)
        inv = flt.invoke(w, h)
        for k, v in uv.items():
            inv.set(k, v)
        if needs:
            inv.set_image("in", img)
        outs.append(inv.render(t=0.4))
    assert np.array_equal(outs[0], outs[1]), "specialised kernel differs from the generic one"
    want = CpuFilter(mm.Filter(src).ir_json_raw).render(w, h, uservals=uv, images={"in": img} if needs else {}, t=0.4)
    mx, nd, n1 = stats(outs[0], want)
    EXP_ORACLE.check("fuzz/%d" % seed, mx, nd, n1, want.size)


@pytest.mark.parametrize("seed", range(40))
def test_random_filters_with_closures_complex_ops_and_options(seed):
    """The richer generator (filter flags, a second image, complex arithmetic, a helper filter
    applied as a closure, random intersampling / edge behaviours): HIP vs oracle, specialised vs
    generic."""
    from tests.fuzz_filters import make_filter_ex
    src, names, opts = make_filter_ex(seed)
    w, h = 96, 64
    imgs = {"in": F.synthetic_image(w, h, seed=1), "in2": F.synthetic_image(50, 70, seed=2)}
    uv = {"k": seed % 7, "m": 0.3 + (seed % 5) * 0.4}
    outs = []
    for spec in (False, True):
        flt = mm.Filter(src, specialize=spec, **opts)
        inv = flt.invoke(w, h)
        for k, v in uv.items():
            inv.set(k, v)
        for n in names:
            inv.set_image(n, imgs[n])
        outs.append(inv.render(t=0.4))
    assert np.array_equal(outs[0], outs[1]), "specialised kernel differs from the generic one"
    want = CpuFilter(mm.Filter(src, **opts).ir_json_raw).render(
        w, h, uservals=uv, images={n: imgs[n] for n in names}, t=0.4, intersample=opts["intersample"],
        edge=(opts["edge_x"], opts["edge_y"]))
    mx, nd, n1 = stats(outs[0], want)
    EXP_ORACLE.check("fuzz_ex/%d" % seed, mx, nd, n1, want.size)


@pytest.mark.parametrize("seed", range(40))
def test_pair_mode_matches_scalar_kernel_and_oracle(seed, monkeypatch):
    """Arithmetic-only filters are evaluated two pixels at a time in lockstep (hipgen.cpp pair mode):
    bytes identical to the one-pixel-at-a-time kernel (MMHIP_PAIR=0) and to the oracle, generic and with the
    user values baked in, on a ragged frame (odd height: the last pair has one row)."""
    from tests.fuzz_filters import make_filter_arith
    src = make_filter_arith(seed)
    w, h = 83, 61
    uv = {"k": seed % 7, "m": 0.3 + (seed % 5) * 0.4}
    outs = {}
    for pair in ("1", "0"):
        monkeypatch.setenv("MMHIP_PAIR", pair)
        for spec in (False, True):
            flt = mm.Filter(src, specialize=spec)
            inv = flt.invoke(w, h)
            for k, v in uv.items():
                inv.set(k, v)
            outs[pair, spec] = inv.render(t=0.3)
    want = CpuFilter(mm.Filter(src).ir_json_raw).render(w, h, uservals=uv, t=0.3)
    for key, got in outs.items():
        assert np.array_equal(got, want), (key, stats(got, want), src)


PAIR_NESTED = """filter t (int k: 0-8 (3), float m: 0-2 (0.7))
  v0 = if x * y < 0.1 then (w = x + 0.3; n = 0; while (w * w < 4) && (n < k + 3) do w = w * w * 0.5 + y; n = n + 1 end; w + n * 0.1) else (y * 0.5) end;
  u = 0; q = x;
  while u < 3 do
    p = 0; rr = q;
    while (rr * rr < 2.5) && (p < 4) do rr = rr * 1.3 + 0.2; p = p + 1 end;
    q = q * 0.7 + rr * 0.1 + p * m * 0.01;
    u = u + 1
  end;
  rgba:[v0, q, x * y, 1]
end
"""


PAIR_BIG_INTS = """filter t (int k: 0-8 (3), float m: 0-2 (0.7))
  n = 0; w = x;
  while (w * w < 4) && (n < k + 9) do w = w * w + y; n = n + 1 end;
  big = n * 3000000 + k;
  rgba:[big * 0.0000001, (big + 0.25) * 0.00000003 * m, if big < 16777217.5 then 0.2 else 0.9 end, 1]
end
"""


@pytest.mark.parametrize("src", [PAIR_NESTED, PAIR_BIG_INTS], ids=["nested", "big_ints"])
def test_pair_mode_nested_control_flow(src, monkeypatch):
    """Pair mode with a data-dependent loop inside a conditional and a loop nest whose inner trip count differs
    between the two pixels of a pair; and with ints beyond 2^24 next to float literals (C computes those in
    double and rounds once): same bytes as the one-pixel kernel and the oracle."""
    w, h = 131, 77
    uv = {"k": 4, "m": 1.1}
    outs = {}
    for pair in ("1", "0"):
        monkeypatch.setenv("MMHIP_PAIR", pair)
        flt = mm.Filter(src)
        assert ("mm_p += 2)" in flt.kernel_source) == (pair == "1")
        inv = flt.invoke(w, h)
        for k, v in uv.items():
            inv.set(k, v)
        outs[pair] = inv.render()
    want = CpuFilter(mm.Filter(src).ir_json_raw).render(w, h, uservals=uv)
    assert np.array_equal(outs["0"], want), stats(outs["0"], want)
    assert np.array_equal(outs["1"], want), stats(outs["1"], want)


def _example_manifest():
    import json
    import os
    from tests.conftest import GOLDEN
    p = os.path.join(GOLDEN, "ir_examples", "manifest.json")
    return json.load(open(p)) if os.path.exists(p) else []


@pytest.mark.parametrize("stem", _example_manifest())
def test_reference_examples_on_gpu(stem, marlene):
    """Every filter under the reference's examples/ (189; IR fixtures, default user values, every
    image input bound to the 128x128 corner of marlene.png, t = 0.3): HIP vs oracle.
    <= 1 LSB unless tests/golden/expected_gpu_vs_oracle.json records more for that exact example
    (discontinuities: modulo, comparisons on libm results, noise lattice look-ups)."""
    import gzip
    import os
    from tests.conftest import GOLDEN
    ir = gzip.open(os.path.join(GOLDEN, "ir_examples", stem + ".json.gz"), "rt").read()
    flt = mm.Filter("", ir_json=ir)
    w = h = 128
    img = np.ascontiguousarray(marlene[:h, :w])
    inv = flt.invoke(w, h)
    images = {}
    for u in flt.uservals:
        if u["kind"] == mm.api.UV_IMAGE:
            inv.set_image(u["name"], img)
            images[u["name"]] = img
    got = inv.render(t=0.3)
    want = CpuFilter(ir).render(w, h, images=images, t=0.3)
    mx, nd, n1 = stats(got, want)
    EXP_ORACLE.check("example/" + stem, mx, nd, n1, want.size)


def _ir_manifest():
    import json
    import os
    from tests.conftest import GOLDEN
    p = os.path.join(GOLDEN, "ir", "manifest.json")
    return json.load(open(p)) if os.path.exists(p) else []


@pytest.mark.parametrize("case", _ir_manifest(), ids=lambda c: c["golden"])
def test_reference_suite_on_gpu(case, marlene):
    """Every filter of the reference's tests/run_tests.sh that compiles (IR fixtures made by
    tests/make_ir_fixtures.py), rendered on the GPU like the suite does (-i, 256x256 or
    -Din=marlene.png) and compared with the reference's golden PNG.  <= 1 LSB unless
    tests/golden/expected_gpu_vs_golden.json records more for that case -- never more than the CPU
    oracle itself shows against the same golden (tests/golden/expected_oracle_vs_golden.json)."""
    import gzip
    import os
    from tests.conftest import GOLDEN
    ir = gzip.open(os.path.join(GOLDEN, "ir", case["ir"]), "rt").read()
    flt = mm.Filter("", ir_json=ir)
    inv = flt.invoke(256, 256)
    for k, v in case["uservals"].items():
        inv.set(k, v)
    if case["needs_image"]:
        for u in flt.uservals:
            if u["kind"] == mm.api.UV_IMAGE:
                inv.set_image(u["name"], marlene)
    try:
        got = inv.render()
    except mm.MathMapError as e:
        if "not implemented" in str(e):
            pytest.skip(str(e))
        raise
    want = load_png_rgb(case["golden"])
    mx, nd, n1 = stats(got[:, :, :3], want)
    EXP_GOLDEN.check(case["golden"], mx, nd, n1, want.size)


SPECIAL_SCALARS = [0.0, -0.0, 1.0, -1.0, 0.5, 2.5, -3.75, 1e-40, -1e-40, 1e-30, 88.5, -104.0, 200.0, 3e38, -3e38,
                   float("inf"), float("-inf"), float("nan")]
COMPLEX_SPECIAL_FUNCS = ["exp(z)", "log(z)", "sqrt(z)", "sin(z)", "cos(z)", "tan(z)", "z^ri:[1.3,0.4]", "ri:[0.3,-0.8]^z",
                         "sinh(z)", "cosh(z)", "tanh(z)", "asin(z)", "acos(z)", "atan(z)", "asinh(z)", "acosh(z)", "atanh(z)"]


@pytest.mark.parametrize("expr", COMPLEX_SPECIAL_FUNCS)
def test_complex_math_special_values(expr):
    """Zeros of both signs, subnormals, overflow thresholds, infinities and NaN in either component (the
    values arrive as user values, so nothing folds at compile time): the device's restated glibc
    functions must take the same special-case branches as glibc -- same finite bits, same infinities
    and zeros with the same signs, NaN where glibc returns NaN."""
    import itertools
    src = "filter probe (float a: -1-1 (0), float b: -1-1 (0)) z = ri:[a, b]; w = %s; rgba:[w[0], w[1], w[0], w[1]] end" % expr
    flt = mm.Filter(src)
    cf = CpuFilter(flt.ir_json_raw)
    inv = flt.invoke(2, 2)
    bad = []
    for a, b in itertools.product(SPECIAL_SCALARS, SPECIAL_SCALARS):
        inv.set("a", a)
        inv.set("b", b)
        got = render_device(inv, 2, 2, floatmap=True)[0, 0, :2]
        want = cf.render(2, 2, uservals={"a": a, "b": b}, floatmap=True)[0, 0, :2]
        same = all((np.isnan(g) and np.isnan(w_)) or g.tobytes() == w_.tobytes() for g, w_ in zip(got, want))
        if not same:
            bad.append(((a, b), [float(v) for v in got], [float(v) for v in want]))
    assert not bad, "%s: %d of %d pairs differ, first: %s" % (expr, len(bad), len(SPECIAL_SCALARS) ** 2, bad[:6])

"""State kept across renders of one invocation: native-filter maps and their memo
(native-filters/cache.c), render-size changes (the GIMP preview -> full render flow,
mathmap.c:2191-2223), supersampling buffers, replaced input uploads.  HIP (through the C ABI)
against the CPU oracle on every step."""
import numpy as np
import pytest

import mathmap_amd as mm
from tests import filters as F
from oracle.ccgen import CpuFilter, render_supersampled
from tests.gpu_util import make_invocation, render_device, stats

pytestmark = pytest.mark.gpu

CHAIN = """
stretched filter chain (stretched image in, float s: 0-1 (0.02))
  b = gaussian_blur(in, s, s);
  c = gaussian_blur(b, 0.015, 0.01);
  c(xy)
end
"""


def test_render_size_change_reallocates_native_maps():
    """Preview (small render size) then the full size, then the preview again, on ONE invocation: the
    blur's float map must follow the render size (it used to stay preview-sized: a device heap
    overflow) and the memo must not return the other size's map."""
    w, h = 320, 200
    img = F.synthetic_image(w, h, seed=5)
    uv = {"hdev": 0.03, "vdev": 0.02}
    flt, inv = make_invocation(F.GAUSS_DIRECT, w, h, uv, {"in": img})
    cf = CpuFilter(flt.ir_json_raw)
    for rw, rh in ((80, 50), (w, h), (80, 50), (160, 100)):
        inv.set_render_size(rw, rh)
        got = inv.render()
        assert got.shape == (rh, rw, 4)
        want = cf.render(w, h, uservals=uv, images={"in": img}, render_size=(rw, rh))
        assert np.array_equal(got, want), ((rw, rh), stats(got, want))
        again = inv.render()          # second frame at this size: the memo path
        assert np.array_equal(again, want), ((rw, rh), "memo", stats(again, want))


@pytest.mark.parametrize("intersample", [False, True])
def test_supersampled_gaussian_blur(intersample):
    """-o on a filter with a native-filter call: the nested renders run gaussian_blur, which grows the
    native workspace -- the supersampling slices must not live in it."""
    w, h = 233, 141
    img = F.synthetic_image(w, h, seed=6)
    uv = {"hdev": 0.04, "vdev": 0.03}
    flt, inv = make_invocation(F.GAUSS_DIRECT, w, h, uv, {"in": img}, intersample=intersample, supersampling=True)
    got = render_device(inv, w, h, supersampled=True)
    want = render_supersampled(CpuFilter(flt.ir_json_raw), w, h, uservals=uv, images={"in": img}, intersample=intersample)
    assert np.array_equal(got, want), stats(got, want)
    # a larger frame afterwards on a fresh invocation of the same filter object, and the first one again
    got2 = render_device(inv, w, h, supersampled=True)
    assert np.array_equal(got2, want)


def test_native_memo_follows_its_producer():
    """c = gaussian_blur(b, ...) with b = gaussian_blur(in, s, s): changing s recomputes b in place;
    c's own arguments did not change, but its input did (the reference keys the cache on image ids,
    cache.c:65-68), so c must be recomputed as well."""
    w, h = 200, 120
    img = F.synthetic_image(w, h, seed=7)
    flt, inv = make_invocation(CHAIN, w, h, {}, {"in": img})
    cf = CpuFilter(flt.ir_json_raw)
    for s in (0.02, 0.06, 0.06, 0.02):
        inv.set("s", s)
        got = inv.render()
        want = cf.render(w, h, uservals={"s": s}, images={"in": img})
        assert np.array_equal(got, want), (s, stats(got, want))


def test_replacing_an_input_image_releases_the_old_upload():
    """set_image twice: the second upload replaces (and frees) the first; results follow the image."""
    w, h = 96, 64
    a, b = F.synthetic_image(w, h, seed=1), F.synthetic_image(w, h, seed=2)
    flt, inv = make_invocation("pond", w, h, {}, {"in": a})
    cf = CpuFilter(flt.ir_json_raw)
    first = inv.render(t=0.1)
    for k in range(20):
        inv.set_image("in", b if k % 2 == 0 else a)
    got = inv.render(t=0.1)
    assert stats(first, cf.render(w, h, images={"in": a}, t=0.1))[0] <= 1
    want = cf.render(w, h, images={"in": a}, t=0.1)
    assert np.array_equal(got, first) and stats(got, want)[0] <= 1
    inv.set_image("in", b)
    assert stats(inv.render(t=0.1), cf.render(w, h, images={"in": b}, t=0.1))[0] <= 1


EDGE_BLUR = """
filter eb (image in, float s: 0-1 (0.03))
  b = gaussian_blur(in, s, s);
  b(xy * 0.9)
end
"""


@pytest.mark.parametrize("intersample", [False, True])
@pytest.mark.parametrize("supersampling", [False, True])
@pytest.mark.parametrize("ex,ey", [(0, 0), (1, 2), (3, 3), (2, 1)])
def test_native_filter_input_honours_edges_and_supersampling(ex, ey, supersampling, intersample):
    """render_image's fetch is get_orig_val_pixel (builtins.c:306): always nearest, but with the
    invocation's supersampling flag (no +0.5), edge behaviours and edge colours.  A non-square input
    smaller than the canvas, so the blur's input map samples outside the image."""
    w, h = 160, 96
    img = F.synthetic_image(53, 37, seed=21)
    colors = (0x20406080, 0xC0A01055)
    flt = mm.Filter(EDGE_BLUR, intersample=intersample, supersampling=supersampling, edge_x=ex, edge_y=ey)
    inv = flt.invoke(w, h)
    inv.set_image("in", img)
    inv.set_edge_colors(*colors)
    got = inv.render()
    want = CpuFilter(flt.ir_json_raw).render(w, h, images={"in": img}, intersample=intersample, supersampling=supersampling,
                                             edge=(ex, ey), edge_colors=colors)
    assert np.array_equal(got, want), stats(got, want)

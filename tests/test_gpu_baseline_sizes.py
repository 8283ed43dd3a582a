"""Every BASELINE.json configuration at its stated size, HIP (through the C ABI) against the CPU
oracle.  The oracle needs minutes for a whole frame of these sizes, so frames are compared on
sampled row bands spread over the frame (first and last rows included); the blur, whose rows all
depend on the whole frame, is compared on sampled rows of a full CPU vertical pass
(oracle.ccgen.gauss_rows).  Results are identical to the reference's on the same inputs: bit-exact
for the blur's float map, <= 1 LSB per channel for the libm-heavy filters (per-case records in
tests/golden/expected_gpu_vs_oracle.json, see tests/expectations.py)."""
import ctypes as C
import os

import numpy as np
import pytest

import mathmap_amd as mm
from tests import filters as F
from mathmap_amd._lib import lib
from oracle.ccgen import CpuFilter, gauss_rows
from tests.expectations import Expectations
from tests.gpu_util import stats

pytestmark = pytest.mark.gpu
EXP = Expectations("gpu_vs_oracle")


def band_starts(h, n, bh):
    """n bands of bh rows: the first and the last rows of the frame and n - 2 spread between."""
    return sorted({0, h - bh} | {int((h - bh) * (k + 0.37) / (n - 2)) for k in range(n - 2)})


@pytest.mark.parametrize("uv", [{}, {"NoTransparency": 1}], ids=["defaults", "NoTransparency"])
def test_droste_8192_bands_match_oracle(uv):
    """BASELINE config 2 (examples/Map/Droste on an 8192 x 8192 input): defaults -- one bilinear tap
    per pixel -- and NoTransparency=1, which drives the multi-tap level loop."""
    w = h = 8192
    img = F.synthetic_image(w, h, seed=1)
    for specialize in (True, False):
        flt = F.load("droste", specialize=specialize)
        inv = flt.invoke(w, h)
        for k, v in uv.items():
            inv.set(k, v)
        inv.set_image("in", img)
        got = inv.render()
        if specialize:
            first = got
            cf = CpuFilter(flt.ir_json_raw)
            tot = [0, 0, 0]
            for lo in band_starts(h, 12, 8):
                want = cf.render(w, h, uservals=uv, images={"in": img}, rows=(lo, lo + 8))
                mx, nd, n1 = stats(got[lo:lo + 8], want[lo:lo + 8])
                tot = [max(tot[0], mx), tot[1] + nd, tot[2] + n1]
            EXP.check("droste8192/%s" % (",".join(sorted(uv)) or "defaults"), tot[0], tot[1], tot[2], 12 * 8 * w * 4)
        else:
            assert np.array_equal(got, first), "generic kernel differs from the specialised one"


@pytest.mark.parametrize("k", [0, 37, 119])
def test_pond_8192_frames_match_oracle(k):
    """BASELINE config 4/5 (examples/Distorts/Pond, 120-frame 8192 x 8192 animation): frame k of 120,
    t = k / 120 computed like the CLI does ((float)frame / (float)num_frames, mathmap_cmdline.c:835)."""
    w = h = 8192
    img = F.synthetic_image(w, h, seed=1)
    t = float(np.float32(k) / np.float32(120))
    flt = F.load("pond", specialize=True)
    inv = flt.invoke(w, h)
    inv.set_image("in", img)
    got = inv.render(t=t, frame=k)
    cf = CpuFilter(flt.ir_json_raw)
    tot = [0, 0, 0]
    for lo in band_starts(h, 10, 8):
        want = cf.render(w, h, images={"in": img}, rows=(lo, lo + 8), t=t, frame=k)
        mx, nd, n1 = stats(got[lo:lo + 8], want[lo:lo + 8])
        tot = [max(tot[0], mx), tot[1] + nd, tot[2] + n1]
    EXP.check("pond8192/frame%d" % k, tot[0], tot[1], tot[2], 10 * 8 * w * 4)


def _pack_rgba8(v):
    """new_template.c.in:279-293: CLAMP01 in float (NaN -> 0), x 255.0 in double, truncation."""
    c = np.where(v > 0, np.minimum(v, np.float32(1.0)), np.float32(0.0)).astype(np.float64)
    return (c * 255.0).astype(np.uint8)


@pytest.mark.parametrize("size,sigma", [(16384, 20.0)], ids=["16384_sigma20"])
def test_gauss_sigma20_16384_rows_are_bit_exact(size, sigma):
    """BASELINE config 3 (native-filters gauss, sigma = 20 px, 16384 x 16384): the blurred float map -- 4.29 GB,
    past 4 GiB, where the scan kernels' 32-bit lane offsets matter -- read back on sampled rows (float-map
    output, no byte quantisation) must equal the oracle's bit for bit; the RGBA8 frame the bench times
    (the blur's last kernel packs it, the pixel kernel is skipped) must be its exact pack.
    MM_TEST_GAUSS_SIZE overrides the size (development on smaller machines)."""
    w = h = int(os.environ.get("MM_TEST_GAUSS_SIZE", size))
    img = F.synthetic_image(w, h, seed=4)
    dev_uv = np.float32(sigma / ((w - 1) / 2.0))          # sigma_px = |dev * (W-1)/2| (gauss.c:659-660)
    uv = {"hdev": float(dev_uv), "vdev": float(dev_uv)}
    rows = sorted({0, 1, 2, 19, 454, 455, h // 2 - 1, h // 2, h - 456, h - 20, h - 2, h - 1} |
                  {int(h * (k + 0.5) / 8) for k in range(8)})
    rows = [r for r in rows if 0 <= r < h]
    threads = max(1, min(16, len(os.sched_getaffinity(0))))
    want = gauss_rows(img, dev_uv, dev_uv, rows, threads=threads)

    flt = F.load("gauss_direct")
    inv = flt.invoke(w, h)
    for k, v in uv.items():
        inv.set(k, v)
    inv.set_image("in", img)
    # (a) the frame as bench.py renders it: RGBA8, written by the blur's last kernel
    dev8 = lib().mmhip_device_alloc(w * h * 4)
    assert dev8
    try:
        inv.render_rows(dev8, 0, h)
        inv.sync()
        assert inv.direct_native_launches() == 1
        got8 = np.empty((len(rows), w, 4), np.uint8)
        for i, r in enumerate(rows):
            assert lib().mmhip_copy_to_host(got8[i].ctypes.data_as(C.c_void_p), C.c_void_p(dev8 + r * w * 4), w * 4) == 0
    finally:
        lib().mmhip_device_free(C.c_void_p(dev8))
    assert np.array_equal(got8, _pack_rgba8(want)), stats(got8, _pack_rgba8(want))
    # (b) the float map itself
    devf = lib().mmhip_device_alloc(w * h * 16)
    assert devf
    try:
        inv.render_rows(devf, 0, h, floatmap=True)
        inv.sync()
        got = np.empty((len(rows), w, 4), np.float32)
        for i, r in enumerate(rows):
            assert lib().mmhip_copy_to_host(got[i].ctypes.data_as(C.c_void_p), C.c_void_p(devf + r * w * 16), w * 16) == 0
    finally:
        lib().mmhip_device_free(C.c_void_p(devf))
    diff = got.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64)
    assert not diff.any(), (np.abs(diff).max(), np.count_nonzero(diff), np.abs(got - want).max())

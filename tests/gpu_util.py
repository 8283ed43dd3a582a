"""Helpers shared by the -m gpu test modules (all calls go through the C ABI)."""
import ctypes as C

import numpy as np

import mathmap_amd as mm
from mathmap_amd._lib import lib


def stats(a, b):
    """(max |a-b|, number of differing values, number differing by more than 1)."""
    d = np.abs(a.astype(int) - b.astype(int))
    return int(d.max()) if d.size else 0, int((d > 0).sum()), int((d > 1).sum())


def make_invocation(src, w, h, uservals=None, images=None, **opts):
    """`src`: .mm text or the name of a filter of tests/filters.py."""
    from tests import filters as F
    flt = F.load(src, **opts) if src in F.NAMES else mm.Filter(src, **opts)
    inv = flt.invoke(w, h)
    for k, v in (uservals or {}).items():
        inv.set(k, v)
    for k, v in (images or {}).items():
        inv.set_image(k, v)
    return flt, inv


def render_device(inv, w, h, rows=None, floatmap=False, bpp=4, t=0.0, frame=0, supersampled=False):
    """Renders row bands `rows` (default: the whole frame) into one device buffer and returns it
    as uint8 [h,w,bpp] or float32 [h,w,4]."""
    px = 16 if floatmap else bpp
    dev = lib().mmhip_device_alloc(w * h * px)
    assert dev
    try:
        if supersampled:
            inv.render_supersampled(dev, t=t, frame=frame, bpp=bpp)
        else:
            for lo, hi in (rows or [(0, h)]):
                inv.render_rows(dev + lo * w * px, lo, hi, t=t, frame=frame, floatmap=floatmap, bpp=bpp)
        inv.sync()
        out = np.empty((h, w, 4), np.float32) if floatmap else np.empty((h, w, bpp), np.uint8)
        assert lib().mmhip_copy_to_host(out.ctypes.data_as(C.c_void_p), C.c_void_p(dev), w * h * px) == 0
    finally:
        lib().mmhip_device_free(C.c_void_p(dev))
    return out


def float_ulps(a, b):
    """|a - b| in float32 ulps over the entries finite in both; NaN patterns must agree."""
    assert np.array_equal(np.isnan(a), np.isnan(b))
    fin = np.isfinite(a) & np.isfinite(b)
    assert np.array_equal(np.isfinite(a), np.isfinite(b))
    return np.abs(a[fin].view(np.int32).astype(np.int64) - b[fin].view(np.int32).astype(np.int64))

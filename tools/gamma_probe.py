import sys, numpy as np
sys.path.insert(0, '.')
import mathmap_amd as mm
from oracle.ccgen import CpuFilter
from tests.gpu_util import render_device
for scale in (3.0, 40.0, 0.01):
    src = "filter probe () z = ri:[x*%g, y*%g]; w = gamma(z); rgba:[w[0], w[1], w[0], w[1]] end" % (scale, scale)
    flt = mm.Filter(src); inv = flt.invoke(256, 256)
    got = render_device(inv, 256, 256, floatmap=True)[:, :, :2]
    want = CpuFilter(flt.ir_json_raw).render(256, 256, floatmap=True)[:, :, :2]
    fin = np.isfinite(got) & np.isfinite(want)
    print("scale", scale, "nan/inf patterns equal:", np.array_equal(np.isnan(got), np.isnan(want)), np.array_equal(np.isinf(got), np.isinf(want)), "finite", fin.mean())
    u = np.abs(got[fin].view(np.int32).astype(np.int64) - want[fin].view(np.int32).astype(np.int64))
    print("   ulps: max", u.max(), "mean", u.mean(), "frac 0:", (u == 0).mean(), "frac<=1:", (u <= 1).mean(), "p99.9", np.percentile(u, 99.9))
    mag = np.maximum(np.hypot(want[..., 0], want[..., 1]), 1e-30)
    err = np.hypot(got[..., 0].astype(float) - want[..., 0], got[..., 1].astype(float) - want[..., 1]) / mag
    f2 = fin.all(axis=2)
    print("   rel err vs |result|: max %.3g p99.5 %.3g" % (err[f2].max(), np.percentile(err[f2], 99.5)))
    big = np.argsort(u)[-3:]
    print("   worst:", got[fin][big], want[fin][big])

import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import mathmap_amd as mm
from mathmap_amd import workloads as W
from oracle.ccgen import CpuFilter
from fuzz_filters import make_filter
bad = []
for seed in range(80, 400):
    src, needs = make_filter(seed)
    w, h = 96, 64
    img = W.synthetic_image(w, h, seed=1)
    uv = {"k": seed % 7, "m": 0.3 + (seed % 5) * 0.4}
    try:
        outs = []
        for spec in (False, True):
            flt = mm.Filter(src, specialize=spec)
            inv = flt.invoke(w, h)
            for k, v in uv.items(): inv.set(k, v)
            if needs: inv.set_image("in", img)
            outs.append(inv.render(t=0.4))
        if not np.array_equal(outs[0], outs[1]): bad.append((seed, "spec != generic")); continue
        want = CpuFilter(mm.Filter(src).ir_json).render(w, h, uservals=uv, images={"in": img} if needs else {}, t=0.4)
        d = np.abs(outs[0].astype(int) - want.astype(int))
        if d.max() > 1 and (d > 1).sum() >= 0.01 * want.size: bad.append((seed, "vs oracle max %d n>1 %d" % (d.max(), (d > 1).sum())))
    except Exception as e:
        bad.append((seed, str(e).splitlines()[0][:200]))
print("bad:", bad)

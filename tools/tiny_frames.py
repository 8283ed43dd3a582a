import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import mathmap_amd as mm
from tests import filters as F
from oracle.ccgen import CpuFilter
bad = []
for (w, h) in [(1, 1), (1, 7), (9, 1), (2, 2), (3, 5), (17, 2), (255, 3)]:
    img = F.synthetic_image(max(w, 2), max(h, 2), seed=3)[:h, :w]
    for name in ("mandelbrot", "ident", "pond", "droste", "gauss_direct"):
        uv = {"hdev": 0.9, "vdev": 0.8} if name == "gauss_direct" else {}
        try:
            flt = F.load(name)
            needs = bool(F.image_names(flt))
            inv = flt.invoke(w, h)
            for k, v in uv.items(): inv.set(k, v)
            if needs: inv.set_image("in", np.ascontiguousarray(img))
            got = inv.render(t=0.2)
            want = CpuFilter(flt.ir_json_raw).render(w, h, uservals=uv, images={"in": np.ascontiguousarray(img)} if needs else {}, t=0.2)
            d = np.abs(got.astype(int) - want.astype(int)).max()
            if d > 1: bad.append((w, h, name, int(d)))
        except Exception as e:
            bad.append((w, h, name, str(e).splitlines()[0][:120]))
print("bad:", bad)

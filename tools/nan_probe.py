import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import mathmap_amd as mm
from tests import filters as F
from oracle.ccgen import CpuFilter
w, h = 128, 64
img = F.synthetic_image(w, h, seed=8)
colors = (0x30507090, 0xA0B0C0D0)
srcs = {
 "nan": "filter n (image in) q = exp(x * 1000 + 900) * 0; in(xy + xy:[q, 0]) end",
 "big": "filter n (image in) big = x * 1000000 * 1000000 * 1000000 * 1000000 * 1000000; in(xy:[big, y]) end",
 "inf": "filter n (image in) q = exp(x * 1000 + 900); in(xy + xy:[q, 0]) end",
 "floor": "filter n (image in) q = exp(x * 1000 + 900) * 0; big = x * 1000000 * 1000000 * 1000000 * 1000000 * 1000000; k = floor(big) + floor(q); in(xy * (1 + k * 0)) end",
}
for name, src in srcs.items():
    for inter in (True, False):
        for ex, ey in ((0, 0), (1, 1), (2, 2), (3, 3)):
            flt = mm.Filter(src, intersample=inter, edge_x=ex, edge_y=ey)
            inv = flt.invoke(w, h); inv.set_image("in", img); inv.set_edge_colors(*colors)
            got = inv.render()
            want = CpuFilter(flt.ir_json_raw).render(w, h, images={"in": img}, intersample=inter, edge=(ex, ey), edge_colors=colors)
            d = np.abs(got.astype(int) - want.astype(int))
            print(name, "bilinear" if inter else "nearest", (ex, ey), "max", d.max(), "n", (d > 0).sum(), flush=True)

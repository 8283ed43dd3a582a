#!/usr/bin/env python3
"""All 189 example filters (tests/golden/ir_examples) at a ragged frame size and two times, HIP vs
oracle -- the GPU suite's sweep with other geometry.  usage: examples_sweep.py [W H]"""
import gzip
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mathmap_amd as mm
from oracle.ccgen import CpuFilter
from tests.conftest import GOLDEN, load_png_rgb


def main():
    w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (203, 150)
    marlene = load_png_rgb(os.path.join(GOLDEN, "marlene.png"))
    img = np.ascontiguousarray(marlene[:h, :w])
    bad = []
    stems = json.load(open(os.path.join(GOLDEN, "ir_examples", "manifest.json")))
    for n, stem in enumerate(stems):
        ir = gzip.open(os.path.join(GOLDEN, "ir_examples", stem + ".json.gz"), "rt").read()
        try:
            flt = mm.Filter("", ir_json=ir)
            inv = flt.invoke(w, h)
            images = {}
            for u in flt.uservals:
                if u["kind"] == mm.api.UV_IMAGE:
                    inv.set_image(u["name"], img)
                    images[u["name"]] = img
            cf = CpuFilter(ir)
            for t in (0.0, 0.7):
                got = inv.render(t=t)
                want = cf.render(w, h, images=images, t=t)
                d = np.abs(got.astype(int) - want.astype(int))
                n1 = int((d > 1).sum())
                if d.max() > 1 and n1 >= 0.005 * want.size:
                    bad.append((stem, t, int(d.max()), n1))
        except Exception as e:
            bad.append((stem, str(e).splitlines()[0][:160]))
        if n % 40 == 0:
            print(n, "of", len(stems), "bad so far", len(bad), flush=True)
    print("bad:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

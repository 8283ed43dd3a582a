"""Stripe mode across two processes on the GPU box (both on the one GPU the box has; stripes are
gathered over gloo on host tensors -- the data path itself has no collective): the blur with
native_row_margin = 0 (each rank computes its own rows of the blur map plus a halo from its
replica of the input, and its blur kernel writes the stripe's pixels directly) and frames of the
Pond animation with the CLI's frame -> t convention.  The union of the stripes must equal the
single-process full-frame render and the oracle."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu

WORKER = textwrap.dedent("""
    import os, sys, ctypes as C
    sys.path.insert(0, %r)
    import numpy as np, torch, torch.distributed as dist
    import mathmap_amd as mm
    from tests import filters as F
    from mathmap_amd._lib import lib
    from mathmap_amd.striping import stripe_rows, gather_stripes, render_stripe, animation_frame_t
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    mm.set_device(0)
    w, h = 640, 451

    def stripe(inv, bpp=4, **kw):
        lo, hi = stripe_rows(h, rank, world)
        dev = lib().mmhip_device_alloc(w * (hi - lo) * bpp)
        try:
            render_stripe(inv, dev, rank, world, **kw)
            inv.sync()
            out = np.empty((hi - lo, w, bpp), np.uint8)
            assert lib().mmhip_copy_to_host(out.ctypes.data_as(C.c_void_p), C.c_void_p(dev), out.nbytes) == 0
        finally:
            lib().mmhip_device_free(C.c_void_p(dev))
        return torch.from_numpy(out)

    img = F.synthetic_image(w, h, seed=13)
    res = {}
    inv = F.load("gauss_direct").invoke(w, h)
    inv.set("hdev", 2 * 3.0 / (w - 1)); inv.set("vdev", 2 * 2.5 / (h - 1))
    inv.set_image("in", img)
    blur = gather_stripes(stripe(inv, native_row_margin=0), h, rank, world)
    assert inv.direct_native_launches() == 1            # the stripe's pixels came straight out of the blur kernel
    pond = F.load("pond", specialize=True).invoke(w, h)
    pond.set_image("in", img)
    frames = [gather_stripes(stripe(pond, t=animation_frame_t(k, 120), frame=k), h, rank, world) for k in (0, 37, 119)]
    if rank == 0:
        np.savez(sys.argv[1], blur=blur.numpy(), f0=frames[0].numpy(), f37=frames[1].numpy(), f119=frames[2].numpy())
    dist.barrier()
    dist.destroy_process_group()
""") % ROOT


def test_two_process_gpu_stripes_equal_full_frame(tmp_path):
    import mathmap_amd as mm
    from tests import filters as F
    from mathmap_amd.striping import animation_frame_t
    from oracle.ccgen import CpuFilter
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = tmp_path / "stripes.npz"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29631", str(script), str(out)],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    got = np.load(out)
    w, h = 640, 451
    img = F.synthetic_image(w, h, seed=13)
    uv = {"hdev": 2 * 3.0 / (w - 1), "vdev": 2 * 2.5 / (h - 1)}
    flt = F.load("gauss_direct")
    inv = flt.invoke(w, h)
    for k, v in uv.items():
        inv.set(k, v)
    inv.set_image("in", img)
    full = inv.render()
    assert np.array_equal(got["blur"], full)
    assert np.array_equal(full, CpuFilter(flt.ir_json_raw).render(w, h, uservals=uv, images={"in": img}))
    pflt = F.load("pond", specialize=True)
    pinv = pflt.invoke(w, h)
    pinv.set_image("in", img)
    cf = CpuFilter(pflt.ir_json_raw)
    for k in (0, 37, 119):
        t = animation_frame_t(k, 120)
        assert np.array_equal(got["f%d" % k], pinv.render(t=t, frame=k)), k
        d = np.abs(got["f%d" % k].astype(int) - cf.render(w, h, images={"in": img}, t=t, frame=k).astype(int))
        assert d.max() <= 1, (k, int(d.max()))

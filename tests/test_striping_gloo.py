"""N>1 path on CPU: two gloo ranks each render their row stripe (with the CPU oracle standing
in for the GPU renderer -- the stripe/gather logic under test is backend independent) and
rank 0 reassembles the frame, which must equal a single full render."""
import os
import subprocess
import sys
import textwrap

import numpy as np

from tests.conftest import ROOT

WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np, torch, torch.distributed as dist
    import mathmap_amd as mm
    from tests import filters as F
    from mathmap_amd.striping import stripe_rows, gather_stripes, replicate_input
    from oracle.ccgen import CpuFilter
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    w, h = 96, 67                      # odd height: stripes differ by one row
    cf = CpuFilter(F.load("mandelbrot").ir_json_raw)
    lo, hi = stripe_rows(h, rank, world)
    full = cf.render(w, h, rows=(lo, hi))          # only rows [lo,hi) are filled
    stripe = torch.from_numpy(np.ascontiguousarray(full[lo:hi]))
    out = gather_stripes(stripe, h, rank, world)
    if rank == 0:
        np.save(sys.argv[1], out.numpy())
    # BASELINE config 5 in small: frames of the 120-frame Pond animation, every frame row-striped across the
    # ranks, t by the CLI's convention; and the blur, whose stripe is its own rows only (what
    # native_row_margin = 0 asks of the GPU blur -- here the oracle's sampled-row blur stands in)
    from mathmap_amd.striping import animation_frame_t
    from oracle.ccgen import gauss_rows
    pw, ph = 120, 77
    img = F.synthetic_image(pw, ph, seed=2)
    pond = CpuFilter(F.load("pond").ir_json_raw)
    lo, hi = stripe_rows(ph, rank, world)
    frames = []
    for k in (0, 37, 119):
        t = animation_frame_t(k, 120)
        part = pond.render(pw, ph, images={"in": img}, rows=(lo, hi), t=t, frame=k)
        frames.append(gather_stripes(torch.from_numpy(np.ascontiguousarray(part[lo:hi])), ph, rank, world))
    dev = np.float32(2 * 3.0 / (pw - 1))
    fm = gauss_rows(img, dev, dev, list(range(lo, hi)))
    c = np.where(fm > 0, np.minimum(fm, np.float32(1.0)), np.float32(0.0)).astype(np.float64)
    blur = gather_stripes(torch.from_numpy((c * 255.0).astype(np.uint8)), ph, rank, world)
    if rank == 0:
        np.savez(sys.argv[1] + ".anim.npz", f0=frames[0].numpy(), f37=frames[1].numpy(), f119=frames[2].numpy(), blur=blur.numpy())
    # input replication (scatter + all-gather): every rank ends up with rank 0's image
    ih, iw = 37, 53
    src_img = torch.arange(ih * iw, dtype=torch.int32).view(ih, iw) * 2654435 if rank == 0 else None
    got = replicate_input(src_img, ih, iw, rank, world)
    want = torch.arange(ih * iw, dtype=torch.int32).view(ih, iw) * 2654435
    assert torch.equal(got, want), "replicate_input mismatch on rank %%d" %% rank
    dist.barrier()
    dist.destroy_process_group()
""") % ROOT


def test_two_rank_stripes_reassemble(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = tmp_path / "frame.npy"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29613")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29613", str(script), str(out)],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:]
    import mathmap_amd as mm
    from tests import filters as F
    from oracle.ccgen import CpuFilter
    want = CpuFilter(F.load("mandelbrot").ir_json_raw).render(96, 67)
    assert np.array_equal(np.load(out), want)
    # the striped animation frames and the striped blur equal single full-frame renders
    from mathmap_amd.striping import animation_frame_t
    anim = np.load(str(out) + ".anim.npz")
    pw, ph = 120, 77
    img = F.synthetic_image(pw, ph, seed=2)
    pond = CpuFilter(F.load("pond").ir_json_raw)
    for k in (0, 37, 119):
        t = animation_frame_t(k, 120)
        assert t == float(np.float32(k) / np.float32(120))
        assert np.array_equal(anim["f%d" % k], pond.render(pw, ph, images={"in": img}, t=t, frame=k)), k
    assert not np.array_equal(anim["f37"], anim["f119"])
    dev = float(np.float32(2 * 3.0 / (pw - 1)))
    gd = F.load("gauss_direct")
    full = CpuFilter(gd.ir_json_raw).render(pw, ph, uservals={"hdev": dev, "vdev": dev}, images={"in": img})
    assert np.array_equal(anim["blur"], full)

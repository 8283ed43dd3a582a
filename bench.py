#!/usr/bin/env python3
"""Headline benchmark: Mpixels/s of the per-pixel path on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W`.  With N > 1 and no WORLD_SIZE in the
environment the script starts its N ranks itself (`python -m torch.distributed.run`, before anything
touches the GPU) and fails if the printed line does not say `n_gpus == N`; under torchrun it is one of
the ranks.  One process per GPU; `--dist-backend gloo` keeps the (control-path only) collectives on the
CPU so that two ranks can rehearse the N > 1 path on a one-GPU box.

A step = one full frame of the hot path (prologue + pixel kernel) per rank; frames of an animation are
independent, so ranks never communicate on the data path (weak scaling; `--mode stripes` row-stripes every
frame across the ranks instead: strong scaling, BASELINE config 5).  Rank 0 prints ONE JSON line.

BASELINE.json's metric is "Mpixels/sec (Mandelbrot & Droste @8192x8192)": `value` is the Mandelbrot frame
rate (config 1, the configuration the >= 100x target is quoted on).  The default line carries every other
BASELINE config next to it, each with its own kernel time, roofline, verification against the CPU oracle
and CPU baseline: "droste" (config 2: defaults and -DNoTransparency=1), "gauss" (config 3: gaussian_blur
sigma = 20 px at 16384^2) and "pond" (config 4: the 120-frame Pond animation, t = frame/120), plus
"host_delivered" -- the rate with the frame's last byte in a host buffer (SURVEY 8d's end point,
PCIe-inclusive; never `value`).  After the timed region sampled row bands of the last timed frame are
compared with the CPU oracle ("verified").
"""
import argparse
import glob
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_PIXEL = {"mandelbrot": 4, "droste": 8, "pond": 8, "ident": 8, "gauss": 104}
GAUSS_FUSED_FLOOR_BYTES = 40            # render fused into pass 1, pack into pass 2
# f64 operations per pixel of the blur (4 channels, 2 passes).  "reference": gauss.c:181-185 as written, per element and
# direction 5 x (2 products, a difference, a sum) = 20, plus the sum of the two directions: 41.  "as_implemented": the
# terms that are provably neutral are dropped (d[0] = 0, n_m[0] = +0 on finite input: 18 causal + 16 anticausal), and the
# anticausal kernel re-runs the causal steps from checkpoints (18 more) instead of storing 8 bytes per value: 53
GAUSS_F64_OPS_PER_PIXEL = {"reference": 4 * 2 * 41, "as_implemented": 4 * 2 * (18 + 18 + 16 + 1)}
HBM_PEAK_GBS = 8000.0                   # MI355X_MICROARCH.md
F32_VECTOR_PEAK_TF = 157.3              # MI355X_MICROARCH.md, FMA counted as 2; exact-rounding code cannot fuse: 78.6
F64_VECTOR_PEAK_TOPS = 78.6 / 2         # f64 vector peak 78.6 TFLOP/s counts an FMA as 2: 39.3 T separately rounded operations/s
PCIE_PEAK_GBS = 63.0                    # MI355X_MICROARCH.md: host link PCIe Gen5 x16
NUM_FRAMES = 120                        # BASELINE config 5: 120-frame animation
NAMES = {"mandelbrot": "examples/Render/Mandelbrot", "droste": "examples/Map/Droste", "pond": "examples/Distorts/Pond",
         "ident": "examples/Utilities/Ident", "gauss": "native-filters gauss sigma=20 px (examples/Blur/Gaussian Blur shape)"}


def frame_t(frame):
    """t of animation frame `frame` as the CLI computes it: (float)frame / (float)num_frames
    (mathmap_cmdline.c:835)."""
    from mathmap_amd.striping import animation_frame_t
    return animation_frame_t(frame % NUM_FRAMES, NUM_FRAMES)


def device_image(torch, w, h):
    """Synthetic RGBA8 input generated directly in HBM (packed 0xRRGGBBAA, alpha 255)."""
    yy = torch.arange(h, device="cuda", dtype=torch.int64).view(h, 1)
    xx = torch.arange(w, device="cuda", dtype=torch.int64).view(1, w)
    packed = torch.full((h, w), 255, device="cuda", dtype=torch.int64)
    g = (xx * 255 // max(w - 1, 1) + yy * 255 // max(h - 1, 1)) // 2
    for c in range(3):
        v = ((xx * (131 + 17 * c) + yy * (71 + 29 * c) + (977 + c * 17)) ^ ((xx * yy) >> 3)) & 63
        packed |= ((v + (g * 3) // 4) & 255) << (24 - 8 * c)
        del v
    img = torch.where(packed >= 2 ** 31, packed - 2 ** 32, packed).to(torch.int32).contiguous()
    return img


def host_image_rows(torch, dev_img, lo, hi):
    """Rows [lo, hi) of the device image as uint8 [rows, W, 3] (what the oracle reads)."""
    import numpy as np
    p = dev_img[lo:hi].cpu().numpy().view(np.uint32)
    out = np.empty(p.shape + (3,), np.uint8)
    for c in range(3):
        out[..., c] = (p >> np.uint32(24 - 8 * c)) & np.uint32(255)
    return out


class Workload:
    """One filter bound to a frame size, an input and user values, rendering into HBM.  The reference's filters are
    loaded from their compiled-IR fixtures (tests/filters.py), not from text."""

    def __init__(self, mm, F, torch, name, size, uservals=None, specialize=True, tile_w=0, dev_img=None):
        self.name, self.w, self.h = name, size, size
        self.torch = torch
        self.uservals = dict(uservals or {})
        self.specialize = bool(specialize)
        if name == "gauss":
            sigma = 20.0 / ((size - 1) / 2.0)          # 20 px (gauss.c:659-660: sigma_px = |dev * (W-1)/2|)
            self.uservals.update(hdev=sigma, vdev=sigma)
        self.flt = F.load("gauss_direct" if name == "gauss" else name, tile_w=tile_w, specialize=specialize)
        t0 = time.perf_counter()
        self.inv = self.flt.invoke(size, size)
        self.needs_image = bool(F.image_names(self.flt))
        self.dev_img = dev_img
        if self.needs_image:
            if self.dev_img is None:
                self.dev_img = device_image(torch, size, size)
            self.inv.set_image_device("in", self.dev_img.data_ptr(), size, size, keepalive=self.dev_img)
        for k, v in self.uservals.items():
            self.inv.set(k, v)
        self.jit_seconds = time.perf_counter() - t0
        self.out = torch.empty((size, size, 4), dtype=torch.uint8, device="cuda")
        self.inv.enable_timing(False)     # per-launch event pairs only in per_launch_pass (they cost 7 us per frame)

    def render(self, frame, stream, rows=None, out=None):
        """Frame `frame` of the animation: t = frame / 120 and the frame number itself, as the command line passes
        them (mathmap_cmdline.c:835-842)."""
        lo, hi = rows or (0, self.h)
        if self.name == "gauss":
            # a new input generation per frame, otherwise the native-filter memo
            # (native-filters/cache.c semantics) would hand back the previous frame's map
            self.inv.set_image_device("in", self.dev_img.data_ptr(), self.w, self.h)
        o = self.out if out is None else out
        self.inv.render_rows(o.data_ptr() + lo * self.w * 4, lo, hi, t=frame_t(frame), frame=frame % NUM_FRAMES, stream=stream)

    def oracle(self):
        from oracle.ccgen import CpuFilter
        return CpuFilter(self.flt.ir_json_raw)

    def host_images(self):
        """The whole input on the host for the oracle (uint8 [H, W, 3])."""
        if not self.needs_image:
            return {}
        return {"in": host_image_rows(self.torch, self.dev_img, 0, self.h)}

    def verify(self, frame, bands=6, rows_per_band=4, images=None, rows=None, out=None):
        """Sampled row bands of `out` (default self.out; frame `frame`) against the CPU oracle, inside rows
        [rows[0], rows[1]) (default: the whole frame); returns {"ok", "max_diff", "n_diff", "n_gt1", "pixels"}.
        ok: <= 1 LSB everywhere (the north-star bar); the blur must be exact."""
        import numpy as np
        w = self.w
        r0, r1 = rows or (0, self.h)
        span = r1 - r0
        rows_per_band = min(rows_per_band, span)
        starts = sorted({r0, r1 - rows_per_band} |
                        {r0 + int((span - rows_per_band) * (k + 0.41) / max(bands - 2, 1)) for k in range(bands - 2)})
        o = self.out if out is None else out
        if self.name == "gauss":
            from oracle.ccgen import gauss_rows
            rr = sorted({r for s in starts for r in range(s, s + rows_per_band)})
            img = (images or self.host_images())["in"]
            dev = np.float32(self.uservals["hdev"])
            threads = max(1, min(16, len(os.sched_getaffinity(0))))
            fm = gauss_rows(img, dev, dev, rr, threads=threads)
            c = np.where(fm > 0, np.minimum(fm, np.float32(1.0)), np.float32(0.0)).astype(np.float64)
            want = (c * 255.0).astype(np.uint8)
            got = o[rr].cpu().numpy()
            d = np.abs(got.astype(int) - want.astype(int))
            return {"ok": bool(d.max() == 0), "max_diff": int(d.max()), "n_diff": int((d > 0).sum()), "n_gt1": int((d > 1).sum()),
                    "pixels": int(len(rr) * w), "against": "oracle.gauss_rows: full CPU vertical pass, sampled rows"}
        cf = self.oracle()
        images = images if images is not None else self.host_images()
        mx = nd = n1 = 0
        for lo in starts:
            want = cf.render(w, self.h, uservals=self.uservals, images=images, rows=(lo, lo + rows_per_band), t=frame_t(frame),
                             frame=frame % NUM_FRAMES)
            got = o[lo:lo + rows_per_band].cpu().numpy()
            d = np.abs(got.astype(int) - want[lo:lo + rows_per_band].astype(int))
            mx, nd, n1 = max(mx, int(d.max())), nd + int((d > 0).sum()), n1 + int((d > 1).sum())
        return {"ok": bool(mx <= 1 and n1 == 0), "max_diff": mx, "n_diff": nd, "n_gt1": n1,
                "pixels": int(len(starts) * rows_per_band * w), "rows": [r0, r1], "frame": frame,
                "against": "oracle (pre-pass IR, gcc -O2, glibc), sampled row bands"}

    def kernel_key(self):
        """Identifies the kernel text that ran (the specialised variant when there is one)."""
        try:
            f = self.flt
            if self.specialize and self.name != "gauss":        # filters with native calls keep the generic kernel
                f = self.flt.specialized(self.uservals)
            return hashlib.sha1(f.kernel_source.encode()).hexdigest()[:16]
        except Exception:
            return None


PER_LAUNCH_PASS_FRAMES = 40
KERNEL_MS_NOTE = ("kernel_ms: GPU time per step over the timed region, from one HIP event before its first and one behind its "
                  "last launch on the launch stream (back-to-back launches overlap at their edges, so this can be below the "
                  "isolated duration); kernel_ms_per_launch_events / kernel_ms_per_kernel: an untimed pass right after it with "
                  "an event pair around every launch -- the figures rocprofv3's kernel trace is compared with")


def per_launch_pass(torch, wl, render_one, frames):
    """Untimed: `frames` more frames with one HIP event pair around every kernel launch (on the launch stream), for the
    per-kernel figures rocprofv3's kernel trace is compared with.  The pairs themselves cost 7 us per frame on the
    headline kernel (tools/event_overhead.py: 0.207 ms per frame with them, 0.200 without), which is why the timed
    region carries two events, not two per launch.  Returns (mean pixel-kernel ms, [(native kernel label, ms)])."""
    import numpy as np
    settle_clocks(torch, render_one, 60.0, batch=4)      # it follows CPU work (the verification): the clocks have dropped
    wl.inv.enable_timing(True)
    wl.inv.drain_kernel_ms()
    wl.inv.drain_native_kernel_ms()
    for i in range(frames):
        render_one(i)
    torch.cuda.synchronize()
    ms = wl.inv.drain_kernel_ms()
    native = wl.inv.drain_native_kernel_ms()
    wl.inv.enable_timing(False)
    return (float(np.mean(ms)) if ms else None), native


def timed_frames(torch, wl, steps, warmup, stream, first_frame=0):
    """W untimed + K timed frames queued back to back, one HIP event before the first and one behind the last launch
    (torch's current stream is the launch stream); returns (elapsed_s, GPU ms per frame over the timed region)."""
    wl.inv.enable_timing(False)
    for i in range(warmup):
        wl.render(first_frame + i, stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for i in range(steps):
        wl.render(first_frame + i, stream)
    e1.record()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    return el, e0.elapsed_time(e1) / steps


def settle_clocks(torch, render_one, target_ms, batch=16, cap_frames=2000):
    """Untimed frames until `target_ms` of wall time have gone by with the GPU busy: a 20-step timed region of the
    headline kernel is 6 ms of GPU work, over before the clocks have settled (measured round 2: 0.295 ms per frame in
    a 20-frame run against 0.268 in a 240-frame run of the same kernel).  Returns (ms spent, frames rendered)."""
    if target_ms <= 0:
        return 0.0, 0
    n = 0
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < target_ms and n < cap_frames:
        for _ in range(batch):
            render_one(n)
            n += 1
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3, n


def pmc_traffic(workload, size, kernel_key):
    """HBM bytes per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    separate runs, tools/pmc_traffic.py): bench.py cannot run the profiler around itself.  A summary
    only counts when it was collected for this very kernel text (its `kernel_key`); otherwise null,
    with the reason."""
    best = None
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic_%s%d.json" % (workload, size)))):
        try:
            j = json.load(open(fn))
        except Exception:
            continue
        best = (fn, j)
    if not best:
        return None, {"file": None, "kernel_key_now": kernel_key, "why": "no PMC summary committed for this workload / size"}
    fn, j = best
    src = {"file": os.path.relpath(fn, ROOT), "collected_for_kernel_key": j.get("kernel_key"), "kernel_key_now": kernel_key,
           "collected_at_commit": j.get("commit"), "kind": "profile-sourced (rocprofv3 --pmc, committed summary), not measured by this run"}
    if workload != "gauss" and (j.get("kernel_key") is None or j.get("kernel_key") != kernel_key):
        src["why"] = "stale: collected for another kernel text"
        return None, src
    return j.get("traffic_bytes_per_launch"), src


def cpu_baseline_bands(cf, w, h, uservals, images, cores, rows_one=512, frame=0):
    """The oracle (cc-equivalent C, gcc -O2) on a bounded sample of the same frame: `nb` bands spread
    evenly over the frame height, so the sample sees the same mix of cheap and expensive rows."""
    def timed(threads, nb, bh):
        starts = [int((h - bh) * (k + 0.5) / nb) for k in range(nb)]
        tot = 0.0
        for lo in starts:
            tm = []
            cf.render(w, h, uservals=uservals, images=images, rows=(lo, lo + bh), threads=threads, timing=tm, t=frame_t(frame),
                      frame=frame % NUM_FRAMES)
            tot += tm[0]
        return w * bh * nb / 1e6 / tot
    nb, bh = 16, max(1, rows_one // 16)
    one = timed(1, nb, bh)
    many = timed(cores, 4, 16 * cores)
    return {"value": one, "unit": "Mpixels/s", "cores": 1, "kind": "port",
            "sample": "%d bands of %d rows spread over the same %dx%d frame; oracle cc-equivalent C (gcc -O2 -fPIC), "
                      "1 thread like the reference CLI (mathmap_cmdline.c:844)" % (nb, bh, w, h),
            "all_cores": {"value": many, "cores": cores,
                          "note": "4 bands of %d rows, row-band threads like the reference GIMP path "
                                  "(mathmap_common.c:972-1006)" % (16 * cores)}}


def host_delivered_rate(torch, wl, steps, rows=None, frame_of=lambda i: i):
    """the metric ends at "the last byte of q written on the host".  Rows `rows` of every frame
    (default: the whole frame) are rendered into two device buffers in turn; frame i's device-to-host copy (pinned
    destination, its own stream) overlaps the kernel of frame i+1.  PCIe-inclusive: reported beside `value`, never
    as it.  Returns (elapsed_s, pixels delivered, host copy equals device frame)."""
    w = wl.w
    lo, hi = rows or (0, wl.h)
    n = hi - lo
    outs = [torch.empty((wl.h, w, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
    host = [torch.empty((n, w, 4), dtype=torch.uint8).pin_memory() for _ in range(2)]
    comp, copy = torch.cuda.Stream(), torch.cuda.Stream()
    done = [torch.cuda.Event() for _ in range(2)]
    copied = [torch.cuda.Event() for _ in range(2)]

    def run(k):
        for i in range(k):
            b = i & 1
            if i >= 2:
                comp.wait_event(copied[b])           # the buffer's previous frame has left
            wl.render(frame_of(i), comp.cuda_stream, rows=(lo, hi), out=outs[b])
            done[b].record(comp)
            copy.wait_event(done[b])
            with torch.cuda.stream(copy):
                host[b].copy_(outs[b][lo:hi], non_blocking=True)
            copied[b].record(copy)
        torch.cuda.synchronize()
    run(3)
    t0 = time.perf_counter()
    run(steps)
    el = time.perf_counter() - t0
    same = bool(torch.equal(host[(steps - 1) & 1], outs[(steps - 1) & 1][lo:hi].cpu()))
    return el, w * n * steps, same


class Comm:
    """The control-path collectives of the bench (barrier, max of a time, gather of a few floats): RCCL on device
    tensors (`nccl`), or gloo on host tensors for rehearsing N > 1 on a box with fewer GPUs than ranks."""

    def __init__(self, torch, dist, backend, rank, world, local_rank, device_index):
        self.torch, self.dist, self.world, self.rank, self.backend = torch, dist, world, rank, backend
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
            else:
                dist.init_process_group("gloo")
        self.dev = "cuda" if backend == "nccl" else "cpu"

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def max(self, x):
        if self.world == 1:
            return float(x)
        t = self.torch.tensor([x], device=self.dev, dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum(self, x):
        if self.world == 1:
            return float(x)
        t = self.torch.tensor([x], device=self.dev, dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def gather(self, x):
        if self.world == 1:
            return [float(x)]
        ks = [self.torch.zeros(1, device=self.dev, dtype=self.torch.float64) for _ in range(self.world)]
        self.dist.all_gather(ks, self.torch.tensor([x], device=self.dev, dtype=self.torch.float64))
        return [float(k.item()) for k in ks]

    def close(self):
        if self.world > 1:
            self.dist.destroy_process_group()


def self_launch(n, argv):
    """`bench.py --gpus N` started by hand: run the N ranks as children of this process (which has not touched the GPU and
    never will), pass their output through and check the line they print."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    p = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    sys.stdout.write(p.stdout)
    sys.stdout.flush()
    if p.returncode != 0:
        sys.exit(p.returncode)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith("{"):
            try:
                line = json.loads(ln)
            except ValueError:
                pass
    if line is None or line.get("n_gpus") != n:
        sys.stderr.write("bench.py: asked for %d ranks, the result line says n_gpus = %r\n" % (n, line and line.get("n_gpus")))
        sys.exit(3)
    sys.exit(0)


def sub_record(torch, mm, F, name, size, uv, args, stream, cores, frames, warmup, pmc_name, first_frame=0,
               verify_frames=None, dev_img=None, host_img=None, cpu_rows=128):
    """One more BASELINE config in the same line: timed frames, kernel time (HIP events on the launch stream), roofline,
    verification of the last frame (and `verify_frames`) against the oracle, CPU baseline."""
    import numpy as np
    from oracle.ccgen import CpuFilter
    d = Workload(mm, F, torch, name, size, uv, bool(args.specialize), args.tile_w, dev_img=dev_img)
    # the same clock-settle phase as the headline workload (a handful of timed frames is a few milliseconds of GPU work)
    settle_clocks(torch, lambda i: d.render(first_frame + i, stream), min(args.settle_ms, 100.0), batch=4 if name == "gauss" else 16)
    el, kms = timed_frames(torch, d, frames, warmup, stream, first_frame)
    if host_img is None and d.needs_image:
        host_img = d.host_images()
    last = first_frame + frames - 1          # what d.out holds after the timed frames
    vs = [d.verify(last, images=host_img)]
    launch_ms, native = per_launch_pass(torch, d, lambda i: d.render(first_frame + i, stream), min(frames, PER_LAUNCH_PASS_FRAMES))
    bpp = ALGO_BYTES_PER_PIXEL[name]
    px = size * size
    ent = {"workload": "%s %dx%d, %s" % (NAMES[name], size, size, "-D" + " -D".join("%s=%g" % kv for kv in sorted(uv.items())) if uv else "defaults"),
           "value": px * frames / 1e6 / el, "unit": "Mpixels/s", "frames": frames, "ms_per_frame": el / frames * 1e3}
    if name == "gauss":
        # the chain's kernels, each timed with its own event pair on the launch stream
        per = {}
        for label, ms in native:
            per.setdefault(label, []).append(ms)
        per = {k: float(np.mean(v)) for k, v in per.items()}
        ent["kernel_ms_per_kernel"] = per
        ent["kernel_ms_per_kernel_sum"] = float(sum(per.values()))
    ent["kernel_ms"] = kms
    ent["kernel_ms_per_launch_events"] = launch_ms
    ent["kernel_ms_note"] = KERNEL_MS_NOTE
    ach = px * bpp / (kms * 1e-3) / 1e9
    tr, trs = pmc_traffic(pmc_name, size, d.kernel_key())
    roof = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "algorithmic_bytes_per_pixel": bpp, "traffic": tr, "traffic_source": trs, "kernel": "mm_pixels"}
    if name == "gauss":
        floor = px * GAUSS_FUSED_FLOOR_BYTES / (kms * 1e-3) / 1e9
        ops = GAUSS_F64_OPS_PER_PIXEL
        roof.update({
            "kernel": "gaussian_blur chain: 2x(k_iir_causal, k_iir_anticausal), render_image fused into the first pass, "
                      "the RGBA8 pack into the last",
            "fused_floor": {"algorithmic_bytes_per_pixel": GAUSS_FUSED_FLOOR_BYTES, "achieved": floor, "frac": floor / HBM_PEAK_GBS,
                            "note": "SURVEY 8(d): 104 B/px is the unfused chain, 40 B/px the floor with render fused into pass 1 "
                                    "and pack into pass 2 (what this implementation does)"},
            "f64": {"bound": "valu-f64", "unit": "T f64 operations/s", "peak": F64_VECTOR_PEAK_TOPS,
                    "achieved_reference_ops": px * ops["reference"] / (kms * 1e-3) / 1e12,
                    "achieved_issued_ops": px * ops["as_implemented"] / (kms * 1e-3) / 1e12,
                    "frac": px * ops["as_implemented"] / (kms * 1e-3) / 1e12 / F64_VECTOR_PEAK_TOPS,
                    "ops_per_pixel": ops,
                    "note": "the recurrence is sequential per line and rounds after every operation (no FMA): 39.3 T separately "
                            "rounded f64 operations/s is the vector unit's limit; the chain is bound by it, not by HBM"}})
    elif name == "droste":
        roof["bound"] = "valu"
        roof["note"] = ("8 B/px compulsory (4 read + 4 written); the kernel is bound by vector-instruction issue (complex log / exp / "
                        "pow chain, ~670 wave-instructions per pixel: profiles/*sq_counters_droste*), the HBM fraction is what "
                        "that leaves of the memory system")
    ent["roofline"] = roof
    for fr in sorted(set(verify_frames or []) - {last}):
        d.render(fr, stream)
        torch.cuda.synchronize()
        vs.append(d.verify(fr, images=host_img))
    ent["verified"] = all(v["ok"] for v in vs)
    ent["verification"] = vs if len(vs) > 1 else vs[0]
    if not args.no_cpu_baseline:
        if name == "gauss":
            cw = min(size, 2048)          # the CPU blur is timed on a smaller square frame (work is linear in pixels)
            cuv = {"hdev": 20.0 / ((cw - 1) / 2.0), "vdev": 20.0 / ((cw - 1) / 2.0)}
            tm = []
            CpuFilter(d.flt.ir_json).render(cw, cw, uservals=cuv, images={"in": F.synthetic_image(cw, cw)}, threads=1, timing=tm)
            ent["cpu_baseline"] = {
                "value": cw * cw / 1e6 / tm[0], "unit": "Mpixels/s", "cores": 1, "kind": "port",
                "sample": "whole %dx%d frame, sigma 20 px: render_image + gauss_iir (both passes) + sample/pack; oracle C "
                          "restatement of native-filters/gauss.c, gcc -O2, 1 thread (the reference's gauss is "
                          "single-threaded)" % (cw, cw)}
        else:
            ent["cpu_baseline"] = cpu_baseline_bands(CpuFilter(d.flt.ir_json), size, size, d.uservals, host_img or {}, cores,
                                                     rows_one=cpu_rows, frame=first_frame + frames - 1)
        ent["gpu_over_cpu_1thread"] = ent["value"] / ent["cpu_baseline"]["value"]
    return ent, d.dev_img, host_img


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks (one per GPU); default: WORLD_SIZE, else 1")
    # defaults: two passes over the 120-frame animation after 24 untimed frames
    ap.add_argument("--steps", type=int, default=240)
    ap.add_argument("--warmup", type=int, default=24)
    ap.add_argument("--workload", default="mandelbrot", choices=sorted(ALGO_BYTES_PER_PIXEL))
    ap.add_argument("--size", type=int, default=0, help="frame edge in pixels (default 8192; 16384 for gauss)")
    ap.add_argument("--tile-w", type=int, default=0)
    ap.add_argument("-D", dest="defs", action="append", default=[], metavar="name=value", help="user value, like the CLI's -D")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", default="frames", choices=["frames", "stripes"],
                    help="multi-GPU decomposition: one whole frame per rank and step (weak scaling, default) or every "
                         "frame row-striped across the ranks (strong scaling; BASELINE config 5)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="backend of the control-path collectives (barrier, max of the times): RCCL, or gloo to rehearse "
                         "N > 1 with several ranks on one GPU")
    ap.add_argument("--settle-ms", type=float, default=150.0,
                    help="untimed frames are rendered for this long (after the --warmup frames) before the timed region, so "
                         "that a short timed region sees settled clocks; reported as settle_ms (0 = off)")
    ap.add_argument("--no-generic", action="store_true", help="skip the generic-kernel comparison")
    ap.add_argument("--no-extras", action="store_true",
                    help="profiling runs: only the timed region (no verification, other configs, host-delivered rate, CPU baseline)")
    ap.add_argument("--no-configs", action="store_true", help="skip the other BASELINE configs (droste, gauss, pond) in the line")
    ap.add_argument("--specialize", type=int, default=1,
                    help="1 = user-value specialising JIT (default), 0 = generic kernel reading user values at run time")
    ap.add_argument("--cpu-rows", type=int, default=0, help="rows of the frame timed on the CPU (0 = auto)")
    args = ap.parse_args()

    env_world = int(os.environ["WORLD_SIZE"]) if "WORLD_SIZE" in os.environ else None
    if env_world is None and (args.gpus or 1) > 1:
        self_launch(args.gpus, sys.argv[1:])          # never returns
    world = env_world or 1
    if args.gpus is not None and args.gpus != world:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE = %d\n" % (args.gpus, world))
        sys.exit(2)

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    if ndev < 1:
        sys.stderr.write("bench.py: no GPU visible (there is no CPU fallback)\n")
        sys.exit(2)
    if args.dist_backend == "nccl" and world > ndev:
        sys.stderr.write("bench.py: %d ranks on %d GPUs needs --dist-backend gloo (RCCL wants one GPU per rank)\n" % (world, ndev))
        sys.exit(2)
    device_index = local_rank % ndev
    torch.cuda.set_device(device_index)
    comm = Comm(torch, dist, args.dist_backend, rank, world, local_rank, device_index)

    import mathmap_amd as mm
    from mathmap_amd.striping import stripe_rows
    from tests import filters as F
    mm.set_device(device_index)                  # the library's own handle on the rank's GPU (one process per GPU)

    size = args.size or (16384 if args.workload == "gauss" else 8192)
    uv = {}
    for d in args.defs:
        k, v = d.split("=", 1)
        uv[k] = float(v)
    wl = Workload(mm, F, torch, args.workload, size, uv, bool(args.specialize), args.tile_w)
    w = h = size
    stream = torch.cuda.current_stream().cuda_stream
    stripes = args.mode == "stripes" and world > 1
    row_lo, row_hi = stripe_rows(h, rank, world) if stripes else (0, h)
    if stripes:
        wl.inv.set_native_row_margin(0)          # a blur stripe computes its own rows + halo locally

    def frame_of(i):
        # frames mode: one animation frame per rank and step (frame i*world+rank of the 120-frame animation);
        # stripes mode: every rank renders its row stripe of frame i
        return i if stripes else i * world + rank

    def step(i):
        wl.render(frame_of(i), stream, rows=(row_lo, row_hi))

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    settle_ms, settle_frames = settle_clocks(torch, step, args.settle_ms)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    comm.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()                                  # HIP events on the launch stream (torch's current one) around the region
    for i in range(args.steps):
        step(i)                                  # queued back to back: no synchronisation per step
    e1.record()
    torch.cuda.synchronize()
    comm.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    my_kernel_ms = e0.elapsed_time(e1) / args.steps     # GPU time per step over the timed region
    elapsed = comm.max(elapsed)
    per_rank_ms = comm.gather(my_kernel_ms)

    last_frame = frame_of(args.steps - 1)
    extras = not args.no_extras
    # ---- every rank: its own rows of its last timed frame against the oracle; the host-delivered rate ----
    images = None
    my_ok, my_v = True, None
    if extras:
        try:
            images = wl.host_images()
            my_v = wl.verify(last_frame, images=images, rows=(row_lo, row_hi))
            my_ok = bool(my_v["ok"])
        except Exception as e:      # a verification that cannot run is a failed verification
            my_ok, my_v = False, {"ok": False, "error": repr(e)[:300]}
        all_ok = comm.sum(0.0 if my_ok else 1.0) == 0.0
        hd_err = None
        try:
            comm.barrier()
            hsteps = max(4, min(args.steps, 12))
            hel, hpx, hsame = host_delivered_rate(torch, wl, hsteps, rows=(row_lo, row_hi), frame_of=frame_of)
        except Exception as e:
            hd_err, hel, hpx, hsame, hsteps = repr(e)[:300], 1.0, 0, False, 0
        hel = comm.max(hel)
        hpx_all = comm.sum(hpx)
        hsame_all = comm.sum(0.0 if hsame else 1.0) == 0.0

    # (behind the verification: that one looks at the frame the timed region left in the output buffer)
    launch_ms, native = per_launch_pass(torch, wl, step, min(args.steps, PER_LAUNCH_PASS_FRAMES))
    if rank == 0:
        mpix = w * h * args.steps * (1 if stripes else world) / 1e6
        value = mpix / elapsed
        k_ms = my_kernel_ms
        per_kernel = None
        if args.workload == "gauss":
            per = {}
            for label, ms in native:
                per.setdefault(label, []).append(ms)
            per_kernel = {k: float(np.mean(v)) for k, v in per.items()}
        bpp = ALGO_BYTES_PER_PIXEL[args.workload]
        px_per_launch = w * (row_hi - row_lo)
        hbm_achieved = px_per_launch * bpp / (k_ms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(args.workload, w, wl.kernel_key())
        res = {
            "metric": "Mpixels/sec (%s @%dx%d)" % (args.workload, w, h),
            "value": value, "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if stripes else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s %dx%d, %s, RGBA8 out, frames of the %d-frame animation (t = frame/%d)"
                                   % (NAMES[args.workload], w, h,
                                      "defaults (num_iterations=32)" if args.workload == "mandelbrot" and not uv
                                      else ("-D" + " -D".join("%s=%g" % kv for kv in sorted(uv.items())) if uv else "defaults"),
                                      NUM_FRAMES, NUM_FRAMES),
                       "frames_per_step_per_gpu": (1.0 / world) if stripes else 1,
                       "parallelism": ("row stripes x %d of one frame (no data-path collective)" if stripes
                                       else "frames x %d (no data-path collective)") % world,
                       "dist_backend": args.dist_backend if world > 1 else None,
                       "jit_seconds": round(wl.jit_seconds, 3)},
            "settle_ms": round(settle_ms, 1), "settle_frames": settle_frames,
            "settle_note": "untimed frames rendered after the --warmup frames until the clocks have settled; the timed region "
                           "is exactly --steps frames",
            "per_rank_kernel_ms": per_rank_ms,
        }
        hbm = {"achieved": hbm_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_achieved / HBM_PEAK_GBS,
               "algorithmic_bytes_per_pixel": bpp}
        kernel_name = ("gaussian_blur chain: 2x(k_iir_causal, k_iir_anticausal), render_image fused into the first pass, "
                       "the RGBA8 pack into the last") if args.workload == "gauss" else "mm_pixels"
        if args.workload == "mandelbrot":
            # The kernel writes 4 B/px and reads nothing: it is bound by vector-ALU issue, not HBM.  Iterations are
            # recovered exactly from the grey level n/32; per iteration 5 mul + 5 add/sub on the complex plane
            # (the j,k parts are identically 0 for the default parameters).
            n_iter = torch.round(wl.out[row_lo:row_hi, :, 0].to(torch.float32) * (32.0 / 255.0)).sum().item()
            flops = n_iter * 10.0
            tf = flops / (k_ms * 1e-3) / 1e12
            res["roofline"] = {"bound": "valu", "achieved": tf, "peak": F32_VECTOR_PEAK_TF, "unit": "TFLOP/s",
                               "frac": tf / F32_VECTOR_PEAK_TF, "frac_of_no_fma_peak": tf / (F32_VECTOR_PEAK_TF / 2),
                               "note": "useful f32 flops; the reference's arithmetic rounds after every operation, so FMA "
                                       "(half of the 157.3 TF vector peak) is not available to it",
                               "pixel_iterations_per_launch": n_iter, "useful_flops_per_launch": flops,
                               "traffic": traffic, "traffic_source": traffic_src, "kernel": kernel_name, "kernel_ms": k_ms,
                               "kernel_ms_per_launch_events": launch_ms, "kernel_ms_note": KERNEL_MS_NOTE, "hbm": hbm}
        else:
            res["roofline"] = dict(hbm, bound="valu" if args.workload == "droste" else "hbm", traffic=traffic,
                                   traffic_source=traffic_src, kernel=kernel_name, kernel_ms=k_ms,
                                   kernel_ms_per_launch_events=launch_ms, kernel_ms_note=KERNEL_MS_NOTE)
            if args.workload == "gauss":
                floor = px_per_launch * GAUSS_FUSED_FLOOR_BYTES / (k_ms * 1e-3) / 1e9
                ops = GAUSS_F64_OPS_PER_PIXEL
                res["roofline"]["kernel_ms_per_kernel"] = per_kernel
                res["roofline"]["fused_floor"] = {"algorithmic_bytes_per_pixel": GAUSS_FUSED_FLOOR_BYTES, "achieved": floor,
                                                  "frac": floor / HBM_PEAK_GBS,
                                                  "note": "SURVEY 8(d): 104 B/px is the unfused chain, 40 B/px the floor with render "
                                                          "fused into pass 1 and pack into pass 2 (what this implementation does)"}
                res["roofline"]["f64"] = {"bound": "valu-f64", "unit": "T f64 operations/s", "peak": F64_VECTOR_PEAK_TOPS,
                                          "achieved_issued_ops": px_per_launch * ops["as_implemented"] / (k_ms * 1e-3) / 1e12,
                                          "frac": px_per_launch * ops["as_implemented"] / (k_ms * 1e-3) / 1e12 / F64_VECTOR_PEAK_TOPS,
                                          "ops_per_pixel": ops}
        if not extras:
            print(json.dumps(res))
        else:
            res["verified"] = bool(all_ok)
            res["verification"] = dict(my_v or {}, ranks_verified=world,
                                       note="every rank compares its own rows of its last timed frame with the oracle; "
                                            "`verified` is the conjunction (rank 0's record shown)")
            cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            cores = max(1, min(cores, 16))     # the GPU box grants 16 host cores per GPU
            # ---- host-delivered rate: the metric as SURVEY 8(d) / BASELINE.md define its end point ----
            if hd_err:
                res["host_delivered"] = {"error": hd_err}
            else:
                gbs = hpx_all * 4 / hel / 1e9
                res["host_delivered"] = {
                    "value": hpx_all / 1e6 / hel, "unit": "Mpixels/s", "frames": hsteps, "n_gpus": world,
                    "ms_per_step": hel / hsteps * 1e3, "host_copy_equals_device_frame": bool(hsame_all),
                    "roofline": {"bound": "pcie", "achieved": gbs, "peak": PCIE_PEAK_GBS * world, "unit": "GB/s",
                                 "frac": gbs / (PCIE_PEAK_GBS * world),
                                 "note": "4 B/px leave every GPU over its own PCIe Gen5 x16 link (63 GB/s each)"},
                    "note": "every rank: pinned host buffer, D2H of step i overlapped with the kernel of step i+1 (two device "
                            "buffers, two streams); bound by the copy, not the kernel; PCIe-inclusive, never `value`"}
            if world == 1 and args.specialize and not args.no_generic and args.workload != "gauss":
                # the generic kernel (user values read at run time) on the same frame: must be byte-identical
                g = Workload(mm, F, torch, args.workload, size, uv, False, args.tile_w, dev_img=wl.dev_img)
                gms = []
                g.inv.enable_timing(True)
                for i in range(3):
                    g.render(last_frame, stream)
                    gms.append(g.inv.last_kernel_ms())
                torch.cuda.synchronize()
                wl.render(last_frame, stream)
                torch.cuda.synchronize()
                res["generic_kernel"] = {"kernel_ms": float(np.mean(gms[1:])), "value": w * h / 1e6 / (float(np.mean(gms[1:])) * 1e-3),
                                         "byte_identical_to_specialised": bool(torch.equal(g.out, wl.out))}
                del g
            if world == 1 and not args.no_cpu_baseline:       # the CPU baseline is a rank-0, N = 1 figure
                from oracle.ccgen import CpuFilter
                if args.workload == "gauss":
                    cw = min(w, 2048)          # the CPU blur is timed on a smaller square frame (work is linear in pixels)
                    cuv = {"hdev": 20.0 / ((cw - 1) / 2.0), "vdev": 20.0 / ((cw - 1) / 2.0)}
                    tm = []
                    CpuFilter(wl.flt.ir_json).render(cw, cw, uservals=cuv, images={"in": F.synthetic_image(cw, cw)}, threads=1, timing=tm)
                    res["cpu_baseline"] = {
                        "value": cw * cw / 1e6 / tm[0], "unit": "Mpixels/s", "cores": 1, "kind": "port",
                        "sample": "whole %dx%d frame, sigma 20 px: render_image + gauss_iir (both passes) + sample/pack; oracle C "
                                  "restatement of native-filters/gauss.c, gcc -O2, 1 thread (the reference's gauss is "
                                  "single-threaded)" % (cw, cw)}
                else:
                    # timed on the IR after the passes (frame constants hoisted into init_frame, like the reference's
                    # own xy-const slice): the faster, fairer CPU figure
                    res["cpu_baseline"] = cpu_baseline_bands(CpuFilter(wl.flt.ir_json), w, h, wl.uservals, images, cores,
                                                             rows_one=args.cpu_rows or 512, frame=last_frame)
                one, allc = res["cpu_baseline"]["value"], res["cpu_baseline"].get("all_cores", {}).get("value")
                res["gpu_over_cpu_1thread"] = value / one
                res["gpu_over_cpu"] = {
                    "device_resident_over_1_thread": value / one,
                    "host_delivered_over_1_thread": (res["host_delivered"].get("value", 0.0) / one) if "value" in res["host_delivered"] else None,
                    "host_delivered_over_all_cores": (res["host_delivered"]["value"] / allc) if allc and "value" in res["host_delivered"] else None,
                    "note": "`value` ends with the frame in HBM; host_delivered ends with its last byte in a host buffer, "
                            "where the reference's timed region ends (mathmap_cmdline.c:833-854)"}
            # ---- the other BASELINE configs (N = 1 only: they are one-GPU configurations) ----
            if world == 1 and args.workload == "mandelbrot" and not args.no_configs:
                del images
                nfr = max(5, min(args.steps // 2, 120))
                res["droste"] = {}
                dimg = dhost = None
                for label, duv, pmc in (("defaults", {}, "droste"), ("NoTransparency=1", {"NoTransparency": 1}, "droste_nt")):
                    try:
                        res["droste"][label], dimg, dhost = sub_record(torch, mm, F, "droste", 8192, duv, args, stream, cores, nfr,
                                                                       max(2, args.warmup), pmc, dev_img=dimg, host_img=dhost)
                    except Exception as e:
                        res["droste"][label] = {"error": repr(e)[:300]}
                # config 4: the 120-frame animation literally -- frame i, t = i/120 (the timed frames are the animation's own)
                try:
                    res["pond"], _, _ = sub_record(torch, mm, F, "pond", 8192, {}, args, stream, cores, NUM_FRAMES, 8, "pond",
                                                   verify_frames=[37], dev_img=dimg, host_img=dhost, cpu_rows=256)
                    res["pond"]["workload"] += ", the %d frames of the animation in order (t = frame/%d, frame = i)" % (NUM_FRAMES, NUM_FRAMES)
                except Exception as e:
                    res["pond"] = {"error": repr(e)[:300]}
                del dimg, dhost
                torch.cuda.empty_cache()
                # config 3: gaussian_blur sigma = 20 px at 16384^2
                try:
                    res["gauss"], _, _ = sub_record(torch, mm, F, "gauss", 16384, {}, args, stream, cores, max(10, min(args.steps // 8, 30)),
                                                    3, "gauss")
                except Exception as e:
                    res["gauss"] = {"error": repr(e)[:300]}
            print(json.dumps(res))
    comm.close()


if __name__ == "__main__":
    main()

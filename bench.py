#!/usr/bin/env python3
"""Headline benchmark: Mpixels/s of examples/Render/Mandelbrot at 8192x8192 on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` (N > 1 under torch.distributed.run,
one rank per GPU).  A step = one full frame of the hot path (prologue + pixel kernel) per
rank; frames of an animation are independent, so ranks never communicate on the data path
(weak scaling).  Prints one JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_PIXEL = {"mandelbrot": 4, "droste": 8, "pond": 8, "ident": 8, "gauss": 104}
HBM_PEAK_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="mandelbrot", choices=sorted(ALGO_BYTES_PER_PIXEL))
    ap.add_argument("--size", type=int, default=0, help="frame edge in pixels (default 8192; 16384 for gauss)")
    ap.add_argument("--tile-w", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", default="frames", choices=["frames", "stripes"],
                    help="multi-GPU decomposition: one whole frame per rank and step (weak scaling, default) or every "
                         "frame row-striped across the ranks (strong scaling; BASELINE config 5)")
    ap.add_argument("--no-generic", action="store_true", help="skip the generic-kernel comparison (for profiling runs)")
    ap.add_argument("--specialize", type=int, default=1,
                    help="1 = user-value specialising JIT (default), 0 = generic kernel reading user values at run time")
    ap.add_argument("--cpu-rows", type=int, default=0, help="rows of the frame timed on the CPU (0 = auto)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)

    import mathmap_amd as mm
    from mathmap_amd import workloads as W
    mm.set_device(local_rank)                  # the library's own handle on the rank's GPU (one process per GPU)

    w = h = args.size or (16384 if args.workload == "gauss" else 8192)
    src = W.ALL["gauss_direct" if args.workload == "gauss" else args.workload]
    flt = mm.Filter(src, tile_w=args.tile_w, specialize=bool(args.specialize))
    t0 = time.perf_counter()
    inv = flt.invoke(w, h)
    jit_s = time.perf_counter() - t0
    needs_image = "image in" in src
    dev_img = None
    if needs_image:
        # synthetic RGBA8 input generated directly in HBM (packed 0xRRGGBBAA, alpha 255)
        yy = torch.arange(h, device="cuda", dtype=torch.int64).view(h, 1)
        xx = torch.arange(w, device="cuda", dtype=torch.int64).view(1, w)
        chans = []
        for c in range(3):
            v = ((xx * (131 + 17 * c) + yy * (71 + 29 * c) + (977 + c * 17)) ^ ((xx * yy) >> 3)) & 63
            g = (xx * 255 // max(w - 1, 1) + yy * 255 // max(h - 1, 1)) // 2
            chans.append(((v + (g * 3) // 4) & 255))
        packed = (chans[0] << 24) | (chans[1] << 16) | (chans[2] << 8) | 255
        dev_img = (packed & 0xFFFFFFFF).to(torch.int64)
        dev_img = torch.where(dev_img >= 2 ** 31, dev_img - 2 ** 32, dev_img).to(torch.int32).contiguous()
        del chans, packed, xx, yy
        inv.set_image_device("in", dev_img.data_ptr(), w, h, keepalive=dev_img)
    if args.workload == "gauss":
        sigma = 20.0 / ((w - 1) / 2.0)     # 20 px (gauss.c:659-660: sigma_px = |dev * (W-1)/2|)
        inv.set("hdev", sigma)
        inv.set("vdev", sigma)
    out = torch.empty((h, w, 4), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    inv.enable_timing(True)

    from mathmap_amd.striping import stripe_rows
    stripes = args.mode == "stripes" and world > 1
    row_lo, row_hi = stripe_rows(h, rank, world) if stripes else (0, h)
    if stripes:
        inv.set_native_row_margin(0)          # a blur stripe computes its own rows + halo locally

    def step(i):
        # frames mode: one animation frame per rank and step (frame index i*world+rank, t = frame/120);
        # stripes mode: every rank renders its row stripe of frame i
        fr = i if stripes else i * world + rank
        if args.workload == "gauss":
            # a new input generation per frame, otherwise the native-filter memo
            # (native-filters/cache.c semantics) would reuse the blurred map
            inv.set_image_device("in", dev_img.data_ptr(), w, h)
        inv.render_rows(out.data_ptr() + row_lo * w * 4, row_lo, row_hi, t=(fr % 120) / 120.0, frame=0, stream=stream)

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    inv.drain_kernel_ms()                        # forget the warm-up launches
    for i in range(args.steps):
        step(i)                                  # queued back to back: no synchronisation per step
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        mpix = w * h * args.steps * (1 if stripes else world) / 1e6
        value = mpix / elapsed
        kernel_ms = inv.drain_kernel_ms()        # HIP events recorded on the launch stream around every launch
        k_ms = float(np.mean(kernel_ms))
        if args.workload == "gauss":
            k_ms = elapsed / args.steps * 1e3     # whole chain: render + 4 scan kernels + sample/pack
        bpp = ALGO_BYTES_PER_PIXEL[args.workload]
        achieved = w * h * bpp / (k_ms * 1e-3) / 1e9
        # HBM traffic of the dominant kernel from the PMC passes (rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE, summarised by tools/pmc_traffic.py into profiles/): bench.py cannot run the
        # profiler around itself, so it reports the committed per-launch figure for this
        # exact workload/size, or null when none has been collected.
        traffic = None
        import glob
        for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic_%s%d.json" % (args.workload, w)))):
            try:
                traffic = json.load(open(fn)).get("traffic_bytes_per_launch")
            except Exception:
                pass
        res = {
            "metric": "Mpixels/sec (%s @%dx%d)" % (args.workload, w, h),
            "value": value, "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if stripes else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "examples/Render/Mandelbrot 8192x8192, defaults (num_iterations=32), RGBA8 out"
                       if args.workload == "mandelbrot" else "%s %dx%d" % (args.workload, w, h),
                       "frames_per_step_per_gpu": (1.0 / world) if stripes else 1,
                       "parallelism": ("row stripes x %d of one frame (no data-path collective)" if stripes
                                       else "frames x %d (no data-path collective)") % world,
                       "jit_seconds": round(jit_s, 3)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "gaussian_blur chain: 2x(k_iir_causal, k_iir_anticausal_T), render_image fused into the first pass, the RGBA8 pack into the last"
                         if args.workload == "gauss" else "mm_pixels",
                         "kernel_ms": k_ms, "algorithmic_bytes_per_pixel": bpp},
        }
        if args.workload == "mandelbrot":
            # compute-side view (the kernel writes 4 B/px and reads nothing, so HBM is not its
            # bound): iterations are recovered exactly from the grey level n/32
            n_iter = torch.round(out[:, :, 0].to(torch.float32) * (32.0 / 255.0)).sum().item()
            flops = n_iter * 10.0     # per iteration: 5 mul + 5 add/sub on the complex plane (j,k parts are 0)
            res["valu"] = {"pixel_iterations_per_launch": n_iter, "useful_flops_per_launch": flops,
                           "achieved_tflops": flops / (k_ms * 1e-3) / 1e12, "peak_tflops_f32_no_fma": 78.6,
                           "note": "exact-rounding code cannot use FMA, so the peak is half the 157.3 TF vector peak"}
        if world == 1 and args.specialize and not args.no_generic and args.workload != "gauss":
            # the generic kernel (user values read at run time) on the same frame: must be
            # byte-identical; its rate is reported beside the specialised one
            g_flt = mm.Filter(src, tile_w=args.tile_w, specialize=False)
            g_inv = g_flt.invoke(w, h)
            if needs_image:
                g_inv.set_image_device("in", dev_img.data_ptr(), w, h, keepalive=dev_img)
            if args.workload == "gauss":
                g_inv.set("hdev", sigma)
                g_inv.set("vdev", sigma)
            g_out = torch.empty_like(out)
            g_inv.enable_timing(True)
            gms = []
            step(args.steps - 1)     # re-render the last timed frame for the comparison
            torch.cuda.synchronize()
            for i in range(3):
                g_inv.render_rows(g_out.data_ptr(), 0, h, t=(((args.steps - 1) * world + rank) % 120) / 120.0, stream=stream)
                gms.append(g_inv.last_kernel_ms())
            torch.cuda.synchronize()
            res["generic_kernel"] = {"kernel_ms": float(np.mean(gms[1:])), "value": w * h / 1e6 / (float(np.mean(gms[1:])) * 1e-3),
                                     "byte_identical_to_specialised": bool(torch.equal(g_out, out))}
        if not args.no_cpu_baseline and world == 1:
            from oracle.ccgen import CpuFilter
            cf = CpuFilter(flt.ir_json)
            cw = w
            if args.workload == "gauss":
                cw = min(w, 2048)          # the CPU blur is timed on a smaller square frame (work is linear in pixels)
            images = {"in": W.synthetic_image(cw, cw)} if needs_image else {}
            cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            cores = max(1, min(cores, 16))     # the GPU box grants 16 host cores per GPU
            if args.workload == "gauss":
                uv = {"hdev": 20.0 / ((cw - 1) / 2.0), "vdev": 20.0 / ((cw - 1) / 2.0)}
                tm = []
                cf.render(cw, cw, uservals=uv, images=images, threads=1, timing=tm)
                one = cw * cw / 1e6 / tm[0]
                res["cpu_baseline"] = {
                    "value": one, "unit": "Mpixels/s", "cores": 1, "kind": "port",
                    "sample": "whole %dx%d frame, sigma 20 px: render_image + gauss_iir (both passes) + sample/pack; "
                              "oracle C restatement of native-filters/gauss.c, gcc -O2, 1 thread (the reference's "
                              "gauss is single-threaded)" % (cw, cw)}
            else:
                # bounded sample: `nb` bands of `bh` rows spread evenly over the frame height, so the
                # sample sees the same mix of cheap and expensive rows as the whole frame
                def timed(threads, nb, bh):
                    starts = [int((h - bh) * (k + 0.5) / nb) for k in range(nb)]
                    tot = 0.0
                    for lo in starts:
                        tm = []
                        cf.render(w, h, images=images, rows=(lo, lo + bh), threads=threads, timing=tm)
                        tot += tm[0]
                    return w * bh * nb / 1e6 / tot
                nb, bh = 16, max(1, (args.cpu_rows or 512) // 16)
                one = timed(1, nb, bh)
                many = timed(cores, 4, 64 * cores)
                res["cpu_baseline"] = {
                    "value": one, "unit": "Mpixels/s", "cores": 1, "kind": "port",
                    "sample": "%d bands of %d rows spread over the same %dx%d frame; oracle cc-equivalent C "
                              "(gcc -O2 -fPIC), 1 thread like the reference CLI (mathmap_cmdline.c:844)" % (nb, bh, w, h),
                    "all_cores": {"value": many, "cores": cores,
                                  "note": "4 bands of %d rows, row-band threads like the reference GIMP path "
                                          "(mathmap_common.c:972-1006)" % (64 * cores)},
                }
            res["gpu_over_cpu_1thread"] = value / res["cpu_baseline"]["value"]
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

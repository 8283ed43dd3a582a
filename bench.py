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

ALGO_BYTES_PER_PIXEL = {"mandelbrot": 4, "droste": 8, "pond": 8, "ident": 8}
HBM_PEAK_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="mandelbrot", choices=sorted(ALGO_BYTES_PER_PIXEL))
    ap.add_argument("--size", type=int, default=8192)
    ap.add_argument("--tile-w", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
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

    w = h = args.size
    src = W.ALL[args.workload]
    flt = mm.Filter(src, tile_w=args.tile_w)
    t0 = time.perf_counter()
    inv = flt.invoke(w, h)
    jit_s = time.perf_counter() - t0
    needs_image = "image in" in src
    if needs_image:
        inv.set_image("in", W.synthetic_image(w, h))
    out = torch.empty((h, w, 4), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    inv.enable_timing(True)

    def step(i):
        # one animation frame per rank and step: frame index i*world+rank, t = frame/120
        fr = i * world + rank
        inv.render_rows(out.data_ptr(), 0, h, t=(fr % 120) / 120.0, frame=0, stream=stream)

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kernel_ms = []
    for i in range(args.steps):
        step(i)
        kernel_ms.append(inv.last_kernel_ms())   # HIP events on the launch stream
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
        mpix = w * h * args.steps * world / 1e6
        value = mpix / elapsed
        k_ms = float(np.mean(kernel_ms))
        bpp = ALGO_BYTES_PER_PIXEL[args.workload]
        achieved = w * h * bpp / (k_ms * 1e-3) / 1e9
        res = {
            "metric": "Mpixels/sec (%s @%dx%d)" % (args.workload, w, h),
            "value": value, "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "examples/Render/Mandelbrot 8192x8192, defaults (num_iterations=32), RGBA8 out"
                       if args.workload == "mandelbrot" else "%s %dx%d" % (args.workload, w, h),
                       "frames_per_step_per_gpu": 1, "parallelism": "frames x %d (no data-path collective)" % world,
                       "jit_seconds": round(jit_s, 3)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "mm_pixels", "kernel_ms": k_ms, "algorithmic_bytes_per_pixel": bpp},
        }
        if not args.no_cpu_baseline and world == 1:
            from oracle.ccgen import CpuFilter
            cf = CpuFilter(flt.ir_json)
            images = {"in": W.synthetic_image(w, h)} if needs_image else {}
            # bounded sample: `nb` bands of `bh` rows spread evenly over the frame height, so the
            # sample sees the same mix of cheap and expensive rows as the whole frame
            nb, bh = 16, max(1, (args.cpu_rows or 512) // 16)
            starts = [int((h - bh) * (k + 0.5) / nb) for k in range(nb)]
            cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)

            def timed(threads):
                tot = 0.0
                for lo in starts:
                    tm = []
                    cf.render(w, h, images=images, rows=(lo, lo + bh), threads=threads, timing=tm)
                    tot += tm[0]
                return w * bh * nb / 1e6 / tot
            one, many = timed(1), timed(cores)
            res["cpu_baseline"] = {
                "value": one, "unit": "Mpixels/s", "cores": 1, "kind": "port",
                "sample": "%d bands of %d rows spread over the same %dx%d frame; oracle cc-equivalent C "
                          "(gcc -O2 -fPIC), 1 thread like the reference CLI (mathmap_cmdline.c:844)" % (nb, bh, w, h),
                "all_cores": {"value": many, "cores": cores,
                              "note": "same sample, row bands on all host cores like the reference GIMP path"},
            }
            res["gpu_over_cpu_1thread"] = value / res["cpu_baseline"]["value"]
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

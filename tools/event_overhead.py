"""What the per-launch HIP event pairs of bench.py's kernel timing cost: the Mandelbrot example at 8192 x 8192, frames
queued back to back, with and without mmhip_enable_timing.  GPU box only."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
import mathmap_amd as mm  # noqa: E402
from tests import filters as F  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    torch.cuda.set_device(0)
    wl = bench.Workload(mm, F, torch, "mandelbrot", 8192)
    stream = torch.cuda.current_stream().cuda_stream
    res = {}
    for rep in range(2):
        for timing in (True, False, True, False):
            wl.inv.enable_timing(timing)
            for i in range(300):
                wl.render(i, stream)
            torch.cuda.synchronize()
            wl.inv.drain_kernel_ms()
            t0 = time.perf_counter()
            for i in range(n):
                wl.render(i, stream)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            k = wl.inv.drain_kernel_ms()
            res.setdefault("events" if timing else "no_events", []).append(
                {"ms_per_frame": el / n * 1e3, "kernel_ms": sum(k) / len(k) if k else None})
    print(json.dumps(res))


if __name__ == "__main__":
    main()

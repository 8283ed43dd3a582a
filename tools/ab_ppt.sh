#!/bin/bash
# A/B of rows per work-item (MMHIP_PPT) and tile width (--tile-w) with tiles in dispatch order, 8192^2.
#   usage (repo root, GPU box): bash tools/ab_ppt.sh > gpurun_out/ab_ppt.txt
for wl in mandelbrot pond; do
  for tw in 16 32 64; do
    for ppt in 2 4 8 16 32; do
      line=$(MMHIP_PPT=$ppt python bench.py --workload $wl --tile-w $tw --steps 120 --warmup 12 --settle-ms 50 --no-extras 2>/dev/null | tail -1)
      echo "$wl tile_w=$tw ppt=$ppt $(echo "$line" | python -c 'import json,sys; j=json.loads(sys.stdin.read()); print("kernel_ms=%.4f" % j["per_rank_kernel_ms"][0])')"
    done
  done
done

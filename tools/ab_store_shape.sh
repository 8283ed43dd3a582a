#!/bin/bash
# A/B of the tile width (how many pixels of a row a wave stores together) and of non-temporal output stores, 8192^2.
#   usage (repo root, GPU box): bash tools/ab_store_shape.sh > gpurun_out/ab_store_shape.txt
for wl in droste "droste -DNoTransparency=1" pond; do
  for tw in 16 32 64; do
    for nt in 1 0; do
      line=$(MMHIP_NT_STORE=$nt python bench.py --workload $wl --tile-w $tw --steps 60 --warmup 6 --settle-ms 50 --no-extras 2>/dev/null | tail -1)
      echo "$wl tile_w=$tw nt_store=$nt $(echo "$line" | python -c 'import json,sys; j=json.loads(sys.stdin.read()); print("kernel_ms=%.4f" % j["per_rank_kernel_ms"][0])')"
    done
  done
done

#!/bin/bash
# A/B of the workgroup -> tile order (MM_XCD_ORDER in hipgen.cpp): 0 = dispatch order, 1 = one contiguous band per XCD,
# 2 = runs of about one tile row dealt to the XCDs in turn.
#   usage (from the repo root on the GPU box): bash tools/ab_xcd_order.sh > gpurun_out/ab_xcd_order.txt
for wl in mandelbrot ident pond droste "droste -DNoTransparency=1"; do
  for xo in 0 1 2; do
    for rep in 1 2; do
      line=$(MMHIP_XCD_ORDER=$xo python bench.py --workload $wl --steps 120 --warmup 12 --no-extras 2>/dev/null | tail -1)
      echo "$wl xcd_order=$xo rep=$rep $(echo "$line" | python -c 'import json,sys; j=json.loads(sys.stdin.read()); print("kernel_ms=%.4f value=%.0f" % (j["per_rank_kernel_ms"][0], j["value"]))')"
    done
  done
done

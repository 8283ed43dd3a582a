#!/bin/bash
# A/B of the blur chain's kernel shapes at 16384^2, sigma 20 px (native_filters.hip gaussian_blur at commit 732273e; the
# producer / consumer kernel was measured, not adopted, and removed again -- record: profiles/r03_ab_gauss_pc.txt):
#   MMHIP_GAUSS_PC=0            one-wave anticausal kernel (two chains per lane), checkpoints every 16 steps
#   MMHIP_GAUSS_PC=1 CK=16|32   producer / consumer pair of waves, checkpoints every 16 / 32 steps
#   usage (from the repo root on the GPU box): bash tools/ab_gauss_pc.sh > gpurun_out/ab_gauss_pc.txt
for cfg in "0 16" "1 16" "1 32"; do
  set -- $cfg
  for rep in 1 2; do
    line=$(MMHIP_GAUSS_PC=$1 MMHIP_GAUSS_CK=$2 timeout -k 10 300 python bench.py --workload gauss --steps 12 --warmup 3 --no-cpu-baseline --no-generic 2>/dev/null | tail -1)
    echo "pc=$1 ck=$2 rep=$rep $(echo "$line" | python -c 'import json,sys; j=json.loads(sys.stdin.read()); r=j["roofline"]; print("chain_ms=%.3f ms_per_step=%.3f verified=%s %s" % (r["kernel_ms"], j["ms_per_step"], j.get("verified"), {k: round(v, 3) for k, v in r["kernel_ms_per_kernel"].items()}))')"
  done
done

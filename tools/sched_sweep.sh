# LLVM AMDGPU scheduling strategies on the fetching kernels (do two unrolled pixel bodies interleave?)
for w in ident pond droste; do
  for st in default max-ilp iterative-ilp; do
    if [ $st = default ]; then fl=""; else fl="-mllvm -amdgpu-sched-strategy=$st"; fi
    MMHIP_HIPRTC_FLAGS="$fl" MMHIP_CACHE_DIR=/tmp/mmc_s$RANDOM python3 bench.py --workload $w --no-cpu-baseline --no-generic --steps 20 --warmup 3 > gpurun_out/ss.log 2>&1
    python3 -c "
import json
l=[x for x in open('gpurun_out/ss.log') if x.startswith('{')]
print('$w $st', json.loads(l[-1])['roofline']['kernel_ms'] if l else open('gpurun_out/ss.log').read()[-300:])"
  done
done

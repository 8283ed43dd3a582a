set -e
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gauss or blur" > gpurun_out/gauss_tests.log 2>&1 || { tail -20 gpurun_out/gauss_tests.log; exit 1; }
tail -2 gpurun_out/gauss_tests.log
hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/f64_rate.hip -o /tmp/f64_rate 2>/dev/null && /tmp/f64_rate > gpurun_out/f64_rate.txt; cat gpurun_out/f64_rate.txt
for ns in 1 2 3 4; do
  MMHIP_GAUSS_SEGMENTS=$ns python bench.py --workload gauss --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/b_gauss_$ns.log 2>&1 || true
  echo "segments $ns: $(python - <<PY
import json
l=[x for x in open('gpurun_out/b_gauss_$ns.log') if x.startswith('{')]
if l:
    j=json.loads(l[-1]); print(j['ms_per_step'], j['roofline'].get('kernel_ms'), j['roofline'].get('breakdown'))
else: print(open('gpurun_out/b_gauss_$ns.log').read()[-400:])
PY
)"
done

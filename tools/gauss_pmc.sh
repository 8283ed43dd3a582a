# HBM traffic of the blur chain at 16384^2: one rocprofv3 --pmc pass per counter (3 profiled steps each).
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/gpmc_$c
  rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/gpmc_$c -o pm --output-format csv -- python3 $R/bench.py --workload gauss --steps 2 --warmup 1 --settle-ms 0 --no-extras > $R/gpurun_out/gpmc_$c.log 2>&1
  if grep -qi "memory access fault\|HSA_STATUS_ERROR" $R/gpurun_out/gpmc_$c.log; then exit 1; fi
done
python3 $R/tools/pmc_traffic_chain.py $R/gpurun_out/gpmc_FETCH_SIZE $R/gpurun_out/gpmc_WRITE_SIZE k_iir 3 $R/gpurun_out/pmc_traffic_gauss16384.json

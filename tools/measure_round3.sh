# Round-3 measurements on one MI355X: bench lines (default and with the driver's flags), rocprofv3 kernel stats, HBM traffic
# (FETCH_SIZE / WRITE_SIZE in separate passes), one SQ-counter pass and one GRBM_GUI_ACTIVE (clock) pass per workload.
# Counters are collected with --kernel-trace only (no other trace domains).  Output: gpurun_out/r03m/, summarised into
# profiles/r03_* by tools/collect_profiles2.py r03.
#   usage (from the repo root on the GPU box): MM_COMMIT=<short hash> bash tools/measure_round3.sh [workloads...]
set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd $R
O=gpurun_out/r03m
mkdir -p $O
echo "${MM_COMMIT:-unknown}" > $O/commit.txt
WL="${@:-mandelbrot ident pond droste droste_nt gauss}"
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err
for w in $WL; do
  case $w in
    droste_nt) ARGS="--workload droste -D NoTransparency=1" ;;
    *) ARGS="--workload $w" ;;
  esac
  STEPS=120; PSTEPS=4
  if [ $w = gauss ]; then STEPS=30; PSTEPS=2; fi
  python3 bench.py $ARGS --no-extras --steps $STEPS > $O/bench_$w.log 2>&1
  rm -rf $O/stats_$w $O/pmc_fetch_$w $O/pmc_write_$w $O/pmc_sq_$w $O/pmc_clk_$w
  rocprofv3 --kernel-trace --stats -d $O/stats_$w -o st --output-format csv -- python3 bench.py $ARGS --no-extras --steps $STEPS > $O/stats_$w.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch_$w -o pm --output-format csv -- python3 bench.py $ARGS --no-extras --steps $PSTEPS --warmup 1 --settle-ms 0 > $O/pmc_fetch_$w.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write_$w -o pm --output-format csv -- python3 bench.py $ARGS --no-extras --steps $PSTEPS --warmup 1 --settle-ms 0 > $O/pmc_write_$w.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace -d $O/pmc_sq_$w -o pm --output-format csv -- python3 bench.py $ARGS --no-extras --steps $PSTEPS --warmup 1 --settle-ms 0 > $O/pmc_sq_$w.log 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_clk_$w -o pm --output-format csv -- python3 bench.py $ARGS --no-extras --steps 24 --warmup 4 --settle-ms 0 > $O/pmc_clk_$w.log 2>&1
  if grep -qi "memory access fault\|gpu fault\|hsa_status_error" $O/pmc_sq_$w.log; then echo "fault in $w"; exit 1; fi
done
find $O -name "*.csv" | head -80

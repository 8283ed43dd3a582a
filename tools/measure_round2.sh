# Round-2 measurements on one MI355X: bench lines, rocprofv3 kernel stats, HBM traffic (FETCH_SIZE / WRITE_SIZE in
# separate passes) and one SQ-counter pass per workload.  Counters are collected with --kernel-trace only (no other
# trace domains).  Output: gpurun_out/r02m/, summarised into profiles/r02_* by tools/collect_profiles2.py.
#   usage (from the repo root on the GPU box): MM_COMMIT=<short hash> bash tools/measure_round2.sh [workloads...]
set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd $R
O=gpurun_out/r02m
mkdir -p $O
echo "${MM_COMMIT:-unknown}" > $O/commit.txt
WL="${@:-mandelbrot ident pond droste droste_nt gauss}"
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
for w in $WL; do
  case $w in
    droste_nt) ARGS="--workload droste -D NoTransparency=1" ;;
    *) ARGS="--workload $w" ;;
  esac
  STEPS=120; PSTEPS=4
  if [ $w = gauss ]; then STEPS=30; PSTEPS=2; fi
  python3 bench.py $ARGS --no-extras --steps $STEPS > $O/bench_$w.log 2>&1
  rm -rf $O/stats_$w $O/pmc_fetch_$w $O/pmc_write_$w $O/pmc_sq_$w
  rocprofv3 --kernel-trace --stats -d $O/stats_$w -o st --output-format csv -- python3 bench.py $ARGS --no-extras --steps $STEPS > $O/stats_$w.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch_$w -o pm --output-format csv -- python3 bench.py $ARGS --no-extras --steps $PSTEPS --warmup 1 > $O/pmc_fetch_$w.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write_$w -o pm --output-format csv -- python3 bench.py $ARGS --no-extras --steps $PSTEPS --warmup 1 > $O/pmc_write_$w.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace -d $O/pmc_sq_$w -o pm --output-format csv -- python3 bench.py $ARGS --no-extras --steps $PSTEPS --warmup 1 > $O/pmc_sq_$w.log 2>&1
  if grep -qi "memory access fault\|gpu fault\|hsa_status_error" $O/pmc_sq_$w.log; then echo "fault in $w"; exit 1; fi
done
find $O -name "*.csv" | head -60

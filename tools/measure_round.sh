set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r01m
rm -rf $O; mkdir -p $O
python3 bench.py > $O/bench_mandelbrot.log 2>&1
for w in ident pond droste gauss; do python3 bench.py --workload $w > $O/bench_$w.log 2>&1; done
rocprofv3 --kernel-trace --stats -d $O/stats_mandelbrot -o st --output-format csv -- python3 bench.py --no-generic --no-cpu-baseline > $O/stats_mandelbrot.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/stats_gauss -o st --output-format csv -- python3 bench.py --workload gauss --no-generic --no-cpu-baseline --steps 5 > $O/stats_gauss.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/stats_ident -o st --output-format csv -- python3 bench.py --workload ident --no-generic --no-cpu-baseline > $O/stats_ident.log 2>&1
for w in mandelbrot ident; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch_$w -o pm --output-format csv -- python3 bench.py --workload $w --no-generic --no-cpu-baseline --steps 5 --warmup 1 > $O/pmc_fetch_$w.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write_$w -o pm --output-format csv -- python3 bench.py --workload $w --no-generic --no-cpu-baseline --steps 5 --warmup 1 > $O/pmc_write_$w.log 2>&1
done
ls -R $O | head -50

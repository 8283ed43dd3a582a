# Per-kernel times of the 16384^2 blur chain (rocprofv3 kernel trace), for each MMHIP_GAUSS_SEGMENTS in "$@".
set -e
cd /tmp && export TMPDIR=/tmp
for ns in "$@"; do
  export MMHIP_GAUSS_SEGMENTS=$ns
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/gp_$ns
  rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/gp_$ns -o g --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload gauss --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/gp_$ns.log 2>&1
  python3 - <<PY
import csv, glob, json
f = glob.glob("$GRAFT_REPO_ROOT/gpurun_out/gp_$ns/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "k_iir" in n or n.startswith("mm_pixels"):
        short = n[n.find("k_iir"):].split("(")[0] if "k_iir" in n else n
        print("segments $ns  %-70s %s calls  %.3f ms" % (short.replace("mm::(anonymous namespace)::", ""), r["Calls"], float(r["AverageNs"]) / 1e6))
l = [x for x in open("$GRAFT_REPO_ROOT/gpurun_out/gp_$ns.log") if x.startswith("{")]
print("segments $ns  ms_per_step", json.loads(l[-1])["ms_per_step"])
PY
done

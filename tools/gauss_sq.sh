# SQ counters of the blur's scan kernels at 16384^2 (one rocprofv3 --pmc pass per counter group).
# MMHIP_GAUSS_PC / MMHIP_GAUSS_CK in the environment select the kernel shape (see native_filters.hip gaussian_blur).
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/gsq_$i
  rocprofv3 --pmc $grp --kernel-trace -d $R/gpurun_out/gsq_$i -o q --output-format csv -- python3 $R/bench.py --workload gauss --steps 2 --warmup 1 --settle-ms 0 --no-extras > $R/gpurun_out/gsq_$i.log 2>&1
  if grep -qi "memory access fault\|HSA_STATUS_ERROR" $R/gpurun_out/gsq_$i.log; then exit 1; fi
  python3 - <<PY
import csv, glob, collections
f = glob.glob("$R/gpurun_out/gsq_$i/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "k_iir" not in n: continue
    short = n[n.find("k_iir"):].split("(")[0].replace("mm::(anonymous namespace)::", "")
    acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k, {c: sum(v) / len(v) for c, v in acc[k].items()})
PY
done

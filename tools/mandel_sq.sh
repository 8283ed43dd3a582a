# SQ instruction counters of the Mandelbrot pixel kernel, pair mode on / off.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for pm in 1 0; do
  export MMHIP_PAIR=$pm
  rm -rf $R/gpurun_out/msq_$pm
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace -d $R/gpurun_out/msq_$pm -o q --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-generic > $R/gpurun_out/msq_$pm.log 2>&1
  python3 - <<PY
import csv, glob, collections
f = glob.glob("$R/gpurun_out/msq_$pm/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Kernel_Name"] == "mm_pixels": acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("MMHIP_PAIR=$pm", {k: round(sum(v)/len(v)/1e6, 1) for k, v in acc.items()})
PY
done

#!/bin/bash
# Effective shader clock of the blur's scan kernels: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel duration
# (MI355X_MICROARCH.md, DVFS give-back).  MMHIP_GAUSS_PC / MMHIP_GAUSS_CK select the kernel shape.
#   usage (repo root, GPU box): bash tools/gauss_clock.sh
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/gclk
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/gclk -o q --output-format csv -- python3 $R/bench.py --workload gauss --steps 6 --warmup 2 --settle-ms 0 --no-extras > $R/gpurun_out/gclk.log 2>&1
python3 - <<PY
import csv, glob, collections
d = "$R/gpurun_out/gclk"
dur = {}
for r in csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
acc = collections.defaultdict(list)
for r in csv.DictReader(open(glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0])):
    n = r["Kernel_Name"]
    if "k_iir" not in n or r["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
    short = n[n.find("k_iir"):].split("(")[0].replace("mm::(anonymous namespace)::", "")
    t = dur[r["Dispatch_Id"]]
    acc[short].append((float(r["Counter_Value"]) / 8 / t / 1e9, t * 1e3))
for k in sorted(acc):
    v = acc[k][2:]          # skip the warm-up frames
    print("%-48s clock %.3f GHz  duration %.3f ms  (%d launches, profiled)" % (k, sum(x[0] for x in v) / len(v), sum(x[1] for x in v) / len(v), len(v)))
PY

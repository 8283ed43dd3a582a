#!/usr/bin/env python3
"""Summarises gpurun_out/<round>m/ (tools/measure_round2.sh / measure_round3.sh; round = argv[1], default r03) into
profiles/<round>_*: the bench JSON line, the rocprofv3
kernel-stats CSV, HBM traffic per launch from the FETCH_SIZE / WRITE_SIZE passes (FETCH doubled: gfx950 counts
128-byte requests as 64 B, MI355X_MICROARCH.md) and the SQ counters per launch with derived per-pixel figures.
Every summary records the commit and the kernel-text key it was collected for (bench.py checks the key)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = sys.argv[1] if len(sys.argv) > 1 else "r03"
G = os.path.join(ROOT, "gpurun_out", RND + "m")
P = os.path.join(ROOT, "profiles")
SIZES = {"mandelbrot": 8192, "ident": 8192, "pond": 8192, "droste": 8192, "droste_nt": 8192, "gauss": 16384}


def counters(d, match):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if match not in n:
                continue
            short = n.replace("mm::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def main():
    commit = open(os.path.join(G, "commit.txt")).read().strip() if os.path.exists(os.path.join(G, "commit.txt")) else None
    if os.path.exists(os.path.join(G, "bench_default.json")):
        shutil.copy(os.path.join(G, "bench_default.json"), os.path.join(P, RND + "_bench_default.json"))
    if os.path.exists(os.path.join(G, "bench_driver_flags.json")):      # python bench.py --gpus 1 --steps 20 --warmup 5
        shutil.copy(os.path.join(G, "bench_driver_flags.json"), os.path.join(P, RND + "_bench_driver_flags.json"))
    for w, sz in SIZES.items():
        log = os.path.join(G, "bench_%s.log" % w)
        if not os.path.exists(log):
            continue
        lines = [l for l in open(log) if l.startswith("{")]
        if not lines:
            continue
        bench = json.loads(lines[-1])
        open(os.path.join(P, RND + "_bench_%s%d.json" % (w, sz)), "w").write(lines[-1])
        key = (bench.get("roofline", {}).get("traffic_source") or {}).get("kernel_key_now")
        kms = bench["roofline"]["kernel_ms"]
        for st in glob.glob(os.path.join(G, "stats_%s" % w, "**", "*kernel_stats.csv"), recursive=True):
            shutil.copy(st, os.path.join(P, RND + "_kernel_stats_%s%d.csv" % (w, sz)))
        match = "k_iir" if w == "gauss" else "mm_pixels"
        fetch, write = counters(os.path.join(G, "pmc_fetch_%s" % w), match), counters(os.path.join(G, "pmc_write_%s" % w), match)
        res = {"workload": w, "size": sz, "commit": commit, "kernel_key": key, "kernels": {},
               "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KiB) in separate passes with --kernel-trace; FETCH doubled "
                       "(gfx950 tallies 128-byte requests as 64 B); per launch, averaged over the profiled launches"}
        total = 0.0
        for k in sorted(set(fetch) | set(write)):
            fb = sum(fetch[k]["FETCH_SIZE"]) / len(fetch[k]["FETCH_SIZE"]) * 1024 * 2 if fetch.get(k, {}).get("FETCH_SIZE") else None
            wb = sum(write[k]["WRITE_SIZE"]) / len(write[k]["WRITE_SIZE"]) * 1024 if write.get(k, {}).get("WRITE_SIZE") else None
            res["kernels"][k] = {"fetch_bytes_corrected": fb, "write_bytes": wb, "launches_profiled": len(fetch.get(k, {}).get("FETCH_SIZE", []))}
            total += (fb or 0) + (wb or 0)
        if res["kernels"]:
            res["traffic_bytes_per_launch"] = total      # gauss: one launch of each of the chain's kernels = one frame
            res["bytes_per_pixel"] = total / (sz * sz)
            json.dump(res, open(os.path.join(P, RND + "_pmc_traffic_%s%d.json" % (w, sz)), "w"), indent=1)
        sq = counters(os.path.join(G, "pmc_sq_%s" % w), match)
        out = {"workload": w, "size": sz, "commit": commit, "kernel_key": key, "bench_kernel_ms": kms, "kernels": {},
               "note": "rocprofv3 --pmc (SQ block, one pass, --kernel-trace) around bench.py --no-extras; per launch; "
                       "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* are quad-cycles summed over waves, SQ_BUSY_CYCLES quad-cycles per SE"}
        for k, c in sq.items():
            d = {n: sum(v) / len(v) for n, v in c.items()}
            px = sz * sz
            if "SQ_INSTS_VALU" in d:
                d["valu_instructions_per_pixel"] = d["SQ_INSTS_VALU"] * 64 / px
                d["salu_instructions_per_pixel"] = d.get("SQ_INSTS_SALU", 0) * 64 / px
                if w != "gauss":
                    n_inst = d["SQ_INSTS_VALU"] + d.get("SQ_INSTS_SALU", 0) + d.get("SQ_INSTS_VMEM_RD", 0)
                    d["simd_cycles_per_instruction_at_2.4GHz"] = kms * 1e-3 * 2.4e9 / (n_inst / 1024)
                    d["simd_cycles_per_valu_instruction_at_2.4GHz"] = kms * 1e-3 * 2.4e9 / (d["SQ_INSTS_VALU"] / 1024)
                if d.get("SQ_WAVE_CYCLES"):
                    d["wave_quadcycles_per_valu_instruction"] = d["SQ_WAVE_CYCLES"] / d["SQ_INSTS_VALU"]
            out["kernels"][k] = d
        # effective shader clock per kernel: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / duration (MI355X_MICROARCH.md)
        clk_dir = os.path.join(G, "pmc_clk_%s" % w)
        traces = glob.glob(os.path.join(clk_dir, "**", "*kernel_trace.csv"), recursive=True)
        if traces:
            dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9 for r in csv.DictReader(open(traces[0]))}
            per = collections.defaultdict(list)
            for f2 in glob.glob(os.path.join(clk_dir, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f2)):
                    if match in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE" and dur.get(r["Dispatch_Id"]):
                        short = r["Kernel_Name"].replace("mm::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                        per[short].append(float(r["Counter_Value"]) / 8 / dur[r["Dispatch_Id"]] / 1e9)
            for k2, v in per.items():
                v = v[len(v) // 4:]          # the first launches run on clocks still ramping
                out["kernels"].setdefault(k2, {})["effective_clock_GHz_profiled"] = sum(v) / len(v)
        if out["kernels"]:
            json.dump(out, open(os.path.join(P, RND + "_sq_counters_%s%d.json" % (w, sz)), "w"), indent=1)
        print(w, "kernel_ms", round(kms, 4), "key", key, "traffic B/px", round(res.get("bytes_per_pixel", 0), 2),
              {k: round(v.get("valu_instructions_per_pixel", 0), 1) for k, v in out["kernels"].items()})


if __name__ == "__main__":
    main()

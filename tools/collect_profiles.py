#!/usr/bin/env python3
"""Copies the artefacts of tools/measure_round.sh (gpurun_out/r01m/) into profiles/ under the
round's prefix and summarises the PMC passes.  usage: collect_profiles.py r01"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out", "r01m")
P = os.path.join(ROOT, "profiles")


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
    sizes = {"mandelbrot": 8192, "ident": 8192, "pond": 8192, "droste": 8192, "gauss": 16384}
    for w, sz in sizes.items():
        log = os.path.join(G, "bench_%s.log" % w)
        if os.path.exists(log):
            lines = [l for l in open(log) if l.startswith("{")]
            if lines:
                open(os.path.join(P, "%s_bench_%s%d.json" % (rnd, w, sz)), "w").write(lines[-1])
        st = os.path.join(G, "stats_%s" % w, "st_kernel_stats.csv")
        if os.path.exists(st):
            shutil.copy(st, os.path.join(P, "%s_kernel_stats_%s%d.csv" % (rnd, w, sz)))
    for w in ("mandelbrot", "ident"):
        for kind in ("fetch", "write"):
            src = os.path.join(G, "pmc_%s_%s" % (kind, w), "pm_counter_collection.csv")
            if os.path.exists(src):
                shutil.copy(src, os.path.join(P, "%s_pmc_%s_%s8192.csv" % (rnd, kind, w)))
        if os.path.isdir(os.path.join(G, "pmc_fetch_%s" % w)):
            subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), os.path.join(G, "pmc_fetch_%s" % w),
                            os.path.join(G, "pmc_write_%s" % w), "mm_pixels",
                            os.path.join(P, "%s_pmc_traffic_%s8192.json" % (rnd, w))], check=True)
    for w in ("ident", "mandelbrot", "pond"):
        for f in glob.glob(os.path.join(G, "pmc_sq_%s" % w, "**", "*counter_collection.csv"), recursive=True):
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if r["Kernel_Name"].startswith("mm_pixels"):
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            d = {k: sum(v) / len(v) for k, v in acc.items()}
            d["note"] = ("rocprofv3 --pmc (SQ block, one pass) around bench.py --workload %s --no-generic; per launch of "
                         "mm_pixels at 8192x8192; SQ_WAVE_CYCLES/WAIT/ACTIVE are quad-cycles summed over waves" % w)
            d["valu_instructions_per_pixel"] = d["SQ_INSTS_VALU"] * 64 / (8192 * 8192)
            d["valu_issue_bound_ms_at_2.4GHz"] = d["SQ_INSTS_VALU"] * 4 / 1024 / 2.4e9 * 1e3
            json.dump(d, open(os.path.join(P, "%s_sq_counters_%s8192.json" % (rnd, w)), "w"), indent=1)
            print(w, round(d["valu_instructions_per_pixel"], 1), round(d["valu_issue_bound_ms_at_2.4GHz"], 3))


if __name__ == "__main__":
    main()

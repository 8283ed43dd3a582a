#!/usr/bin/env python3
"""Like pmc_traffic.py, for a chain of kernels (the blur): per-kernel and per-step HBM bytes from
rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes.  A kernel matches when its name contains the given
substring; the per-step figure is the sum over one step's launches (launch count / steps).

usage: pmc_traffic_chain.py <fetch_dir> <write_dir> <substring> <steps_profiled> <out.json>
"""
import collections
import csv
import glob
import json
import re
import sys


def short(name):
    m = re.search(r"(k_\w+)<.*?(DrawableSrc|MapSrcT<true>|MapSrcT<false>|MapSrc)", name)
    return "%s<%s>" % (m.group(1), m.group(2)) if m else name.split("(")[0]


def totals(d, sub, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if sub in row["Kernel_Name"] and row["Counter_Name"] == counter:
                acc[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    return acc


def main():
    fetch_dir, write_dir, sub, steps, out = sys.argv[1:6]
    steps = int(steps)
    f, w = totals(fetch_dir, sub, "FETCH_SIZE"), totals(write_dir, sub, "WRITE_SIZE")
    res = {"note": "FETCH_SIZE doubled (gfx950: 128-byte requests counted as 64 B); WRITE_SIZE as reported", "kernels": {}}
    step = 0.0
    for k in sorted(set(f) | set(w)):
        fb = sum(f[k]) / len(f[k]) * 1024 * 2 if f.get(k) else None
        wb = sum(w[k]) / len(w[k]) * 1024 if w.get(k) else None
        res["kernels"][k] = {"launches": len(f.get(k, [])), "fetch_bytes_corrected": fb, "write_bytes": wb}
        step += ((fb or 0) + (wb or 0)) * (len(f.get(k, [])) / steps)
    res["traffic_bytes_per_launch"] = step          # per bench step (one whole chain)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()

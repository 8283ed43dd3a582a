#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs (separate passes, as
MI355X_MICROARCH.md prescribes) into per-launch HBM bytes for one kernel.

FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950 correction from the guide: FETCH_SIZE counts
128-byte requests at 64 B, i.e. reports half the bytes of a wide coalesced stream, so it
is doubled; WRITE_SIZE is exact for 16-B/lane streaming stores (narrower stores are
uncalibrated -- stated with the numbers).

usage: pmc_traffic.py <fetch_dir> <write_dir> <kernel_name> <out.json>
"""
import csv
import glob
import json
import sys


def mean_counter(d, kernel, counter):
    vals = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Kernel_Name"] == kernel and row["Counter_Name"] == counter:
                vals.append(float(row["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def main():
    fetch_dir, write_dir, kernel, out = sys.argv[1:5]
    f, nf = mean_counter(fetch_dir, kernel, "FETCH_SIZE")
    w, nw = mean_counter(write_dir, kernel, "WRITE_SIZE")
    res = {"kernel": kernel, "launches_fetch": nf, "launches_write": nw,
           "FETCH_SIZE_KiB_raw": f, "WRITE_SIZE_KiB_raw": w,
           "fetch_bytes_corrected": None if f is None else f * 1024 * 2,
           "write_bytes": None if w is None else w * 1024}
    if f is not None and w is not None:
        res["traffic_bytes_per_launch"] = res["fetch_bytes_corrected"] + res["write_bytes"]
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()

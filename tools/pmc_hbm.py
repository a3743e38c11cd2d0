#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel trace only):
   pmc_hbm.py <fetch_dir> <write_dir> <drop_first_n> <name-filter,...>  -> JSON {kernel: {launches, FETCH_SIZE_KiB, WRITE_SIZE_KiB, bytes}}
bytes = 1024 * (2 * FETCH_SIZE + WRITE_SIZE): FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 correction (re-calibrated in
profiles/r01_fetch_size_calibration.txt), WRITE_SIZE 1:1."""
import collections
import csv
import glob
import json
import sys

fd, wd, drop, keys = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4].split(",")


def collect(d, name):
    vals = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name and any(x in r["Kernel_Name"] for x in keys):
                vals[r["Kernel_Name"].split("(")[0][:120]].append((int(r.get("Dispatch_Id", 0)), float(r["Counter_Value"])))
    return {k: [x for _, x in sorted(v)][drop:] or [x for _, x in sorted(v)] for k, v in vals.items()}


f, w = collect(fd, "FETCH_SIZE"), collect(wd, "WRITE_SIZE")
out = {}
for k in sorted(set(f) | set(w)):
    fv, wv = f.get(k, [0.0]), w.get(k, [0.0])
    fm, wm = sum(fv) / len(fv), sum(wv) / len(wv)
    out[k] = {"launches": min(len(fv), len(wv)), "FETCH_SIZE_KiB": round(fm, 1), "WRITE_SIZE_KiB": round(wm, 1), "bytes_per_launch": int(1024 * (2 * fm + wm))}
print(json.dumps(out, indent=1))

#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc CSV output: python tools/pmc_summary.py <dir>... (prints per-kernel averages).
DAAC_PMC_FILTER=substr[,substr] picks the kernels (default: gram, scan_kernel)."""
import collections
import csv
import glob
import os
import sys

KEYS = [k.replace(". ", ", ") for k in os.environ.get("DAAC_PMC_FILTER", "gram,scan_kernel").split(",")]  # (". " stands for ", " inside a template argument list)

for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, c in agg.items():
            if not any(x in k for x in KEYS):
                continue
            print(d, k)
            for name, v in sorted(c.items()):
                print(f"   {name:28s} {sum(v) / len(v):.5g}  (n={len(v)})")
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        if os.environ.get("DAAC_PMC_MEAN"):  # one line per kernel instead of one per launch
            dur = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if any(x in r["Kernel_Name"] for x in KEYS):
                    dur[r["Kernel_Name"][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            for k, v in dur.items():
                print(f"   mean_us {sum(v) / len(v):10.1f}  (n={len(v)})  {k}")
            continue
        for r in csv.DictReader(open(f)):
            if any(x in r["Kernel_Name"] for x in KEYS):
                print("   duration_us", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "vgpr", r.get("VGPR_Count"),
                      "scratch", r.get("Scratch_Size"), "lds", r.get("LDS_Block_Size"))

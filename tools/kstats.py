#!/usr/bin/env python3
"""Per-kernel rows of a rocprofv3 --kernel-trace --stats output directory: python tools/kstats.py <dir> [substr,substr...] -> name calls avg_us total_ms"""
import csv
import glob
import sys

keys = sys.argv[2].split(",") if len(sys.argv) > 2 else [""]
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in keys):
            print(f"   {r['Name'][:64]:64s} calls {int(r['Calls']):4d}  avg {float(r['AverageNs']) / 1e3:9.1f} us  total {float(r['TotalDurationNs']) / 1e6:8.2f} ms")

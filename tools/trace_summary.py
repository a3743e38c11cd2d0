#!/usr/bin/env python3
"""Per-kernel statistics from a rocprofv3 --kernel-trace CSV with the warm-up launches dropped:
   trace_summary.py <dir> <drop_first_n_per_kernel> [name-filter,...]  -> CSV lines: kernel, launches, avg_us, min_us, max_us, total_us, pct
(rocprofv3's own --stats averages every launch, warm-ups included; a kernel cannot take longer than the timed step that contains it)"""
import collections
import csv
import glob
import sys

d, drop = sys.argv[1], int(sys.argv[2])
keys = sys.argv[3].split(",") if len(sys.argv) > 3 else None
rows = collections.defaultdict(list)
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
out = []
for k, v in rows.items():
    if keys and not any(x in k for x in keys):
        continue
    v.sort()
    v = v[drop:] if len(v) > drop else v
    us = [(e - s) / 1e3 for s, e in v]
    out.append((sum(us), k, len(us), sum(us) / len(us), min(us), max(us)))
tot = sum(o[0] for o in out) or 1.0
print('"Name","Calls","TotalDurationUs","AverageUs","MinUs","MaxUs","Percentage","WarmupsDropped"')
for t, k, n, avg, mn, mx in sorted(out, reverse=True):
    print(f'"{k[:150]}",{n},{t:.1f},{avg:.2f},{mn:.2f},{mx:.2f},{100 * t / tot:.2f},{drop}')

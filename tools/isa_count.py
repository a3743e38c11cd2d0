#!/usr/bin/env python3
"""Counts instruction classes per basic block of one kernel in a hipcc -save-temps .s file (a CPU-side proxy for
SQ_INSTS_VALU / SALU / LDS while no GPU is at hand).  usage: isa_count.py file.s kernel-substring [min_block_size]"""
import re, sys
path, key = sys.argv[1], sys.argv[2]
minsz = int(sys.argv[3]) if len(sys.argv) > 3 else 12
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and key in l.split(":")[0])
end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.end_amdhsa_kernel") or lines[i].strip().startswith("s_endpgm"))
blocks, cur, name = [], [], "entry"
for l in lines[start + 1:end + 1]:
    s = l.strip()
    if not s or s.startswith(";") or s.startswith("."):
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            blocks.append((name, cur)); cur, name = [], m.group(1)
        continue
    cur.append(s.split()[0])
blocks.append((name, cur))
def cls(op):
    if op.startswith("v_"): return "VALU"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"): return "WAIT"
    if op.startswith("s_"): return "SALU"
    if op.startswith("ds_"): return "LDS"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"): return "VMEM"
    return "OTHER"
tot = {}
for name, ops in blocks:
    c = {}
    for o in ops:
        c[cls(o)] = c.get(cls(o), 0) + 1
        tot[cls(o)] = tot.get(cls(o), 0) + 1
    if len(ops) >= minsz:
        print(f"{name:14s} n={len(ops):4d} " + " ".join(f"{k}={v}" for k, v in sorted(c.items())))
print("total", tot)

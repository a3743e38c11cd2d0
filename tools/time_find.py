#!/usr/bin/env python3
"""find_iter / leftmost_find_iter count + checksum throughput on cfg3 (bytewise) — tools/time_find.py [mib] [sparse|dense] [find|leftmost]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import daachorse_amd as da
from daachorse_amd import ScanMode, synth
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
hk = sys.argv[2] if len(sys.argv) > 2 else "sparse"
pats = synth.patterns_cfg3()
hay = torch.empty(mib << 20, dtype=torch.uint8, device="cuda")
if hk == "sparse":
    synth.device_uniform(hay, synth.SEEDS["cfg3_hay"], synth.ALPHA_LOWER_SPACE)
else:
    synth.device_wordsoup(hay, synth.SEEDS["cfg3_dense"], pats, 20)
res = {}
only = sys.argv[3] if len(sys.argv) > 3 else ""   # "find" / "leftmost": that iterator alone (one kernel family per profile)
for name, kind, mode, opts in (("find_iter (double array)", da.MatchKind.Standard, ScanMode.Find, {}),
                               ("leftmost_find_iter LL", da.MatchKind.LeftmostLongest, ScanMode.LeftmostFind, {})):
    if (only == "find" and "double array" not in name) or (only == "leftmost" and "leftmost" not in name):
        continue
    for k, v in opts.items():
        da.set_option(k, v)
    pma = da.DoubleArrayAhoCorasickBuilder().match_kind(kind).build(pats)
    pma.upload(0)
    r = pma.scan_count(mode, hay)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        r2 = pma.scan_count(mode, hay)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    assert r2 == r
    name = f"{name} [engine {da.last_engine()}]"
    print(f"{name:40s} {hk} {mib} MiB: {best * 1e3:8.2f} ms  {hay.numel() / best / 1e9:7.1f} GB/s  count={r[0]} checksum={r[1]:016x}", flush=True)
    res[(kind, name.split(' (')[0])] = r

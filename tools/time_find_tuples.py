#!/usr/bin/env python3
"""The tuple LIST of the restart iterators left in HBM (daac_scan_device16 in Find / LeftmostFind mode) on cfg3 — tools/time_find_tuples.py [mib] [sparse|dense]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import daachorse_amd as da
from daachorse_amd import ScanMode, synth
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
hk = sys.argv[2] if len(sys.argv) > 2 else "sparse"
pats = synth.patterns_cfg3()
da.set_option("max_result_bytes", 64 << 30)
hay = torch.empty(mib << 20, dtype=torch.uint8, device="cuda")
if hk == "sparse":
    synth.device_uniform(hay, synth.SEEDS["cfg3_hay"], synth.ALPHA_LOWER_SPACE)
else:
    synth.device_wordsoup(hay, synth.SEEDS["cfg3_dense"], pats, 20)
for name, kind, mode in (("find_iter", da.MatchKind.Standard, ScanMode.Find), ("leftmost_find_iter LL", da.MatchKind.LeftmostLongest, ScanMode.LeftmostFind)):
    pma = da.DoubleArrayAhoCorasickBuilder().match_kind(kind).build(pats)
    pma.upload(0)
    dm = pma.scan_device(mode, hay, fmt16=True)
    cnt = dm.count
    dm.free()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        dm = pma.scan_device(mode, hay, fmt16=True)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
        dm.free()
    want = pma.scan_count(mode, hay)
    print(f"{name:24s} tuples [engine {da.last_engine()}] {hk} {mib} MiB: {cnt} tuples (count says {want[0]}), {best * 1e3:.2f} ms -> {hay.numel() / best / 1e9:.1f} GB/s of haystack", flush=True)

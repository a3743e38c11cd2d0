#!/usr/bin/env python3
"""find_iter / leftmost_find_iter (cfg3, 1 GiB) under launch-shape options: ab_find.py [mib] [sparse|dense]"""
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
pm = {"find": (da.DoubleArrayAhoCorasickBuilder().match_kind(da.MatchKind.Standard).build(pats), ScanMode.Find),
      "leftmost": (da.DoubleArrayAhoCorasickBuilder().match_kind(da.MatchKind.LeftmostLongest).build(pats), ScanMode.LeftmostFind)}
for v in pm.values():
    v[0].upload(0)
ref = {}
for opts in ({}, {"restart_bpc": 4}, {"restart_bpc": 2}, {"restart_bpc": 6}, {"seg_bytes": 1024}, {"seg_bytes": 4096}, {"seg_bytes": 8192}, {"seg_bytes": 16384}, {"seg_bytes": 65536},
             {"seg_bytes": 8192, "restart_bpc": 4}, {"seg_bytes": 16384, "restart_bpc": 2}):
    for k, v in {"restart_bpc": 8, "seg_bytes": 0, **opts}.items():
        da.set_option(k, v)
    for name, (pma, mode) in pm.items():
        r = pma.scan_count(mode, hay)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            r = pma.scan_count(mode, hay)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        ref.setdefault(name, r)
        print(f"{name:9s} {str(opts):45s} {best * 1e3:8.2f} ms {hay.numel() / best / 1e9:7.1f} GB/s {'ok' if r == ref[name] else 'MISMATCH'}", flush=True)

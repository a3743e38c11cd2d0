#!/usr/bin/env python3
"""Device-resident tuple emission (daac_scan_device) on cfg3: GB/s of haystack and of tuples written, GRAM emitter vs the
segment scanners.  usage: python tools/time_emit.py [mib] [sparse|dense] [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import daachorse_amd as da
from daachorse_amd import ScanMode, synth

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 512
hk = sys.argv[2] if len(sys.argv) > 2 else "sparse"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
pats = synth.patterns_cfg3()
da.set_option("max_result_bytes", 64 << 30)
hay = torch.empty(mib << 20, dtype=torch.uint8, device="cuda")
if hk == "sparse":
    synth.device_uniform(hay, synth.SEEDS["cfg3_hay"], synth.ALPHA_LOWER_SPACE)
else:
    synth.device_wordsoup(hay, synth.SEEDS["cfg3_dense"], pats, 20)
for emit in (2, 1, 0):  # 2: staged writes, 1: a pair of stores per tuple, 0: segment scanners
    da.set_option("emit", 1 if emit else 0)
    da.set_option("emit_staged", 1 if emit == 2 else 0)
    pma = da.DoubleArrayAhoCorasick.new(pats)
    pma.upload(0)
    n = hay.numel() if emit else min(hay.numel(), 256 << 20)  # the segment scanners are slow: a prefix is enough
    h = hay[:n]
    dm = pma.scan_device(ScanMode.FindOverlapping, h)
    cnt = dm.count
    dm.free()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        dm = pma.scan_device(ScanMode.FindOverlapping, h)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
        dm.free()
    print(f"emit={emit} engine_used={da.last_engine()} {hk} {n >> 20} MiB: {cnt} tuples, {best * 1e3:.2f} ms  ->  {n / best / 1e9:.1f} GB/s of haystack, "
          f"{cnt * 24 / best / 1e9:.1f} GB/s of tuples written ({cnt / n:.3f} tuples/byte)", flush=True)

#!/usr/bin/env python3
"""Device-resident tuple emission (daac_scan_device) on cfg3: GB/s of haystack and of tuples written, GRAM emitter vs the
segment scanners.  usage: python tools/time_emit.py [mib] [sparse|dense] [reps] [-] [only16]"""
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
ver = 0  # (sys.argv[4] was the round-3 emitter's version: ignored, kept so that the positions of the arguments stay)
for k, v in os.environ.items():  # DAAC_OPT_<name>=<value>: tuning options for A/B runs
    if k.startswith("DAAC_OPT_"):
        da.set_option(k[len("DAAC_OPT_"):], int(v))
only16 = len(sys.argv) > 5 and sys.argv[5] == "only16"
for emit, fmt16 in (((1, True),) if only16 else ((1, True), (1, False), (0, False))):  # the GRAM emitter in both device formats, then the segment scanners
    da.set_option("emit", emit)
    pma = da.DoubleArrayAhoCorasick.new(pats)
    pma.upload(0)
    n = hay.numel() if emit else min(hay.numel(), 256 << 20)  # the segment scanners are slow: a prefix is enough
    h = hay[:n]
    dm = pma.scan_device(ScanMode.FindOverlapping, h, fmt16=fmt16)
    cnt = dm.count
    dm.free()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        dm = pma.scan_device(ScanMode.FindOverlapping, h, fmt16=fmt16)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
        dm.free()
    tb = 16 if fmt16 else 24
    print(f"emit={emit} version={ver} tuple_bytes={tb} engine_used={da.last_engine()} {hk} {n >> 20} MiB: {cnt} tuples, {best * 1e3:.2f} ms  ->  {n / best / 1e9:.1f} GB/s of haystack, "
          f"{cnt * tb / best / 1e9:.1f} GB/s of tuples written ({cnt / n:.3f} tuples/byte)", flush=True)

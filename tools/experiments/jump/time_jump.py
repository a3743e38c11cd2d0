#!/usr/bin/env python3
"""find_iter count + checksum of cfg3 on the JUMP engine only (for rocprofv3): tools/time_jump.py [mib] [sparse|dense]"""
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
pma = da.DoubleArrayAhoCorasick.new(pats)
pma.upload(0)
for _ in range(4):
    t0 = time.perf_counter()
    r = pma.scan_count(ScanMode.Find, hay)
    torch.cuda.synchronize()
    print(f"{hk} {mib} MiB: {(time.perf_counter() - t0) * 1e3:8.2f} ms engine {da.last_engine()} {r}", flush=True)

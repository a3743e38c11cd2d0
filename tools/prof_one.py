#!/usr/bin/env python3
"""Runs ONE configuration a few times (for rocprofv3): python tools/prof_one.py cfg3 sparse gram 1024 [mib] [gram_version] [count_only]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import daachorse_amd as da
from daachorse_amd import Engine, ScanMode, synth

wl, hk, eng, thr = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
mib = int(sys.argv[5]) if len(sys.argv) > 5 else 1024
pats = synth.patterns_cfg3() if wl == "cfg3" else synth.patterns_cfg2()
da.set_option("threads", thr)
pma = da.DoubleArrayAhoCorasick.new(pats)
hay = torch.empty(mib << 20, dtype=torch.uint8, device="cuda")
if hk == "sparse":
    synth.device_uniform(hay, synth.SEEDS[f"{wl}_hay"], synth.ALPHA_LOWER_SPACE if wl == "cfg3" else synth.ALPHA_PRINTABLE)
else:
    synth.device_wordsoup(hay, synth.SEEDS[f"{wl}_dense"], pats, 20 if wl == "cfg3" else 13, noise_256=77 if wl == "cfg3" else 0)
e = {"gram": Engine.Gram, "tiered": Engine.Tiered, "darray": Engine.DArray, "auto": Engine.Auto}[eng]
da.set_option("gram_version", int(sys.argv[6]) if len(sys.argv) > 6 else 0)
count_only = len(sys.argv) > 7 and sys.argv[7] == "1"
for _ in range(3):
    print(pma.count(ScanMode.FindOverlapping, hay, engine=e) if count_only else pma.scan_count(ScanMode.FindOverlapping, hay, engine=e))

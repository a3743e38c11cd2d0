#!/usr/bin/env python3
"""The lazy iterator over page-locked HOST haystacks (daac_iter_open / daac_iter_next_batch to exhaustion): cfg3's 1 GiB (0.6 matches per
byte: the tuples' way back over PCIe decides) and cfg2's sparse haystack (the haystack's way to the device decides).
usage: python tools/time_iter.py [mib] [window_mib ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import daachorse_amd as da
from daachorse_amd import ScanMode, synth

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
windows = [int(x) for x in sys.argv[2:]] or [64]
n = mib << 20
host = torch.empty(n, dtype=torch.uint8).pin_memory()
dev = torch.empty(n, dtype=torch.uint8, device="cuda")
for name, pats, fill in (("cfg3", synth.patterns_cfg3(), lambda: synth.device_uniform(dev, synth.SEEDS["cfg3_hay"], synth.ALPHA_LOWER_SPACE)),
                         ("cfg2_sparse", synth.patterns_cfg2(), lambda: synth.device_uniform(dev, synth.SEEDS["cfg2_hay"], synth.ALPHA_PRINTABLE))):
    pma = da.DoubleArrayAhoCorasick.new(pats)
    pma.upload(0)
    fill()
    host.copy_(dev)
    torch.cuda.synchronize()
    h = host.numpy()
    want = pma.count(ScanMode.FindOverlapping, dev)
    for w in windows:
        da.set_option("iter_window", w << 20)
        for compact, tb in ((True, 8), (False, 16)):
            best = None
            for rep in range(3):
                t0 = time.perf_counter()
                it = pma.find_overlapping_iter(h, compact=compact)
                cnt = 0
                while True:
                    run = it.next_batch8() if compact else it.next_batch()
                    if run is None:
                        break
                    cnt += len(run[0]) if compact else len(run)
                it.close()
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            assert cnt == want, (cnt, want)
            print(f"{name:12s} window {w:4d} MiB, {tb}-byte tuples: {best * 1e3:8.2f} ms  {n / best / 1e9:6.2f} GB/s of haystack, {cnt * tb / best / 1e9:6.2f} GB/s of tuples over PCIe ({cnt} matches)", flush=True)
    # Iterator::next one match at a time through ctypes is a Python number, not the library's: a C++ caller's loop is tests/native/cpp_facade_test.cpp
da.set_option("iter_window", 64 << 20)

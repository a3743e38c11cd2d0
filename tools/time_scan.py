#!/usr/bin/env python3
"""Times daac_scan (C ABI only, no numpy copy of the result) on a prefix of the cfg3 haystack."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import daachorse_amd as da
from daachorse_amd import _ffi, synth, ScanMode
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 64
pma = da.DoubleArrayAhoCorasick.new(synth.patterns_cfg3())
hay = torch.empty(mib << 20, dtype=torch.uint8, device="cuda")
synth.device_uniform(hay, synth.SEEDS["cfg3_hay"], synth.ALPHA_LOWER_SPACE)
L = _ffi.lib()
for mode in (ScanMode.FindOverlapping, ScanMode.Find):
    for rep in range(3):
        out = C.c_void_p()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _ffi.check(L.daac_scan(pma._h, int(mode), 0, hay.data_ptr(), hay.numel(), 1, None, C.byref(out)))
        dt = time.perf_counter() - t0
        n = L.daac_matches_count(out)
        t1 = time.perf_counter()
        L.daac_matches_free(out)
        print(mode.name, f"{mib} MiB: {n} matches, daac_scan {dt*1e3:.1f} ms ({(mib<<20)/dt/1e9:.2f} GB/s, {n*24/dt/1e9:.2f} GB/s of tuples), free {1e3*(time.perf_counter()-t1):.1f} ms")

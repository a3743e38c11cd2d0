#!/usr/bin/env python3
"""Times `.count()` of the wide-alphabet dictionaries (PFX engine vs the double array): binary256 (100k random 3-12-byte patterns over all
256 byte values, uniform random haystack) and utf8jp (cfg5's 50k UTF-8 patterns scanned bytewise over cfg5's Zipf text).
usage: ab_pfx.py [mib]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import daachorse_amd as da
from daachorse_amd import Engine, ScanMode, synth

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
res = torch.zeros(3, dtype=torch.int64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
for name in ("binary256", "utf8jp"):
    pats = synth.patterns_binary256() if name == "binary256" else synth.patterns_cfg5()
    pma = da.DoubleArrayAhoCorasick.new(pats)
    pma.upload(0)
    n = mib << 20
    if name == "utf8jp":
        n -= n % synth.CFG5_SLOT
    hay = torch.empty(n, dtype=torch.uint8, device="cuda")
    if name == "binary256":
        synth.device_uniform(hay, synth.SEEDS["bin_hay"], synth.ALPHA_BYTES)
    else:
        synth.device_zipf_text(hay)
    torch.cuda.synchronize()
    ref = None
    for ename, eng in (("pfx", Engine.Pfx), ("auto", Engine.Auto), ("darray", Engine.DArray)):
        try:
            for _ in range(2):
                pma.count(ScanMode.FindOverlapping, hay, engine=eng, stream=stream, result_dev=res.data_ptr())
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            reps = 5 if ename != "darray" else 2
            for _ in range(reps):
                pma.count(ScanMode.FindOverlapping, hay, engine=eng, stream=stream, result_dev=res.data_ptr())
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            cnt = int(res[0].item())
            ref = cnt if ref is None else ref
            print(f"{name:10s} {ename:7s} {hay.numel() / ms / 1e6:8.1f} GB/s {ms:9.3f} ms  count {cnt} {'ok' if cnt == ref else 'MISMATCH'} engine_used={da.last_engine()}", flush=True)
        except Exception as ex:  # noqa
            print(f"{name:10s} {ename:7s} failed: {ex}", flush=True)
    for ename, eng in (("pfx", Engine.Pfx), ("darray", Engine.DArray)):  # count + checksum
        for _ in range(2):
            pma.scan_count(ScanMode.FindOverlapping, hay, engine=eng, stream=stream, result_dev=res.data_ptr())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(2):
            pma.scan_count(ScanMode.FindOverlapping, hay, engine=eng, stream=stream, result_dev=res.data_ptr())
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 2
        print(f"{name:10s} {ename:7s} count+checksum {hay.numel() / ms / 1e6:8.1f} GB/s {ms:9.3f} ms  {[int(x) & 0xffffffff for x in res.tolist()]} engine_used={da.last_engine()}", flush=True)
    del hay

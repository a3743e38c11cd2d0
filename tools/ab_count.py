#!/usr/bin/env python3
"""A/B sweep of the `.count()` kernels in ONE process: for every variant (a set of daac_set_option pairs) times
find_overlapping_iter(haystack).count() on cfg3 sparse / dense (and cfg2) with HIP events and checks the count.
usage: ab_count.py [mib] [variant-file]   -> one line per (workload, variant): GB/s, ms, count ok"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import daachorse_amd as da
from daachorse_amd import Engine, ScanMode, synth

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
DEFAULTS = {"gram3_tail": -1, "gram4_arith": 1, "gram4_filter": 1, "gram_version": 0, "gram_ppl": 0, "gram2_rfull": 1, "gram_region": 0, "threads": 1024, "blocks_per_cu": 0, "gram_slab": 4096}
VARIANTS = [
    ("gram4 auto", {"gram_version": 4}),
    ("gram4 p16", {"gram_version": 4, "gram_ppl": 16}),
    ("gram4 p32", {"gram_version": 4, "gram_ppl": 32}),
    ("gram4 p32 plain", {"gram_version": 4, "gram_ppl": 32, "gram3_tail": 0}),
    ("gram4 p32 tail", {"gram_version": 4, "gram_ppl": 32, "gram3_tail": 1}),
    ("gram4 p32 class table", {"gram_version": 4, "gram_ppl": 32, "gram4_arith": 0}),
    ("gram4 p32 coarse directory", {"gram_version": 4, "gram_ppl": 32, "gram2_rfull": 0}),
]
if len(sys.argv) > 2:
    VARIANTS = [(n, o) for n, o in json.load(open(sys.argv[2]))]


def setopts(o):
    for k, v in {**DEFAULTS, **o}.items():
        try:
            da.set_option(k, v)
        except da.DaachorseError:
            pass


res = torch.zeros(3, dtype=torch.int64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
for wl, hk in (("cfg3", "sparse"), ("cfg3", "dense"), ("cfg2", "sparse")):
    pats = synth.patterns_cfg3() if wl == "cfg3" else synth.patterns_cfg2()
    pma = da.DoubleArrayAhoCorasick.new(pats)
    pma.upload(0)
    hay = torch.empty(mib << 20, dtype=torch.uint8, device="cuda")
    if hk == "sparse":
        synth.device_uniform(hay, synth.SEEDS[f"{wl}_hay"], synth.ALPHA_LOWER_SPACE if wl == "cfg3" else synth.ALPHA_PRINTABLE)
    else:
        synth.device_wordsoup(hay, synth.SEEDS[f"{wl}_dense"], pats, 20, noise_256=77)
    torch.cuda.synchronize()
    ref = None
    if VARIANTS:   # the first timed variant of a process ran 4-5 % slow (clocks, first touches): ten launches of it that are not timed
        setopts(VARIANTS[0][1])
        for _ in range(10):
            pma.count(ScanMode.FindOverlapping, hay, engine=Engine.Gram, stream=stream, result_dev=res.data_ptr())
        torch.cuda.synchronize()
    for name, opts in VARIANTS:
        setopts(opts)
        try:
            for _ in range(2):
                pma.count(ScanMode.FindOverlapping, hay, engine=Engine.Gram, stream=stream, result_dev=res.data_ptr())
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                pma.count(ScanMode.FindOverlapping, hay, engine=Engine.Gram, stream=stream, result_dev=res.data_ptr())
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            cnt = int(res[0].item())
            if ref is None:
                ref = cnt
            print(f"{wl} {hk:6s} {name:28s} {hay.numel() / ms / 1e6:8.1f} GB/s  {ms:8.3f} ms  count {cnt} {'ok' if cnt == ref else 'MISMATCH'}", flush=True)
        except Exception as ex:  # noqa
            print(f"{wl} {hk:6s} {name:28s} failed: {ex}", flush=True)
    setopts({})
    del hay

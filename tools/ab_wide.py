#!/usr/bin/env python3
"""The wide look-alike dictionaries (synth.patterns_unidic_like / patterns_o200k_like) per engine: `.count()`, count + checksum and 16-byte
tuples — tools/ab_wide.py [mib] [name,name] [engine,engine] [what,what]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import daachorse_amd as da
from daachorse_amd import Engine, ScanMode, synth

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
da.set_option("max_result_bytes", 64 << 30)
names = sys.argv[2].split(",") if len(sys.argv) > 2 else ("unidic_like", "o200k_like", "utf8jp")
engines = [Engine[e] for e in sys.argv[3].split(",")] if len(sys.argv) > 3 else (Engine.Auto, Engine.DArray, Engine.Tiered)
whats = sys.argv[4].split(",") if len(sys.argv) > 4 else ("count", "count+checksum", "tuples16")
for name in names:
    pats = {"unidic_like": synth.patterns_unidic_like, "o200k_like": synth.patterns_o200k_like, "utf8jp": synth.patterns_cfg5}[name]()
    n = (mib << 20) - (mib << 20) % synth.CFG5_SLOT
    hay = torch.empty(n, dtype=torch.uint8, device="cuda")
    if name == "o200k_like":
        synth.device_wordsoup(hay, synth.SEEDS["o200k_hay"], synth.o200k_soup_words(), 17)
    else:
        synth.device_zipf_text(hay)
    p = da.DoubleArrayAhoCorasick.new(pats)
    p.upload(0)
    ref = None
    for eng in engines:
        for what, fn in (("count", lambda: p.count(ScanMode.FindOverlapping, hay, engine=eng)),
                         ("count+checksum", lambda: p.scan_count(ScanMode.FindOverlapping, hay, engine=eng)[0]),
                         ("tuples16", lambda: p.scan_device(ScanMode.FindOverlapping, hay, engine=eng, fmt16=True))):
            if what not in whats:
                continue
            try:
                r = fn()
                torch.cuda.synchronize()
                best = 1e9
                for _ in range(2):
                    if what == "tuples16":
                        r.free()
                    t0 = time.perf_counter()
                    r = fn()
                    torch.cuda.synchronize()
                    best = min(best, time.perf_counter() - t0)
                cnt = r.count if what == "tuples16" else r
                if what == "tuples16":
                    r.free()
                ref = cnt if ref is None else ref
                print(f"{name:12s} {eng.name:7s} {what:15s} {best * 1e3:9.2f} ms {n / best / 1e9:8.1f} GB/s  matches={cnt} {'' if cnt == ref else 'MISMATCH'} engine_used={da.last_engine()}", flush=True)
            except da.DaachorseError as e:
                print(f"{name:12s} {eng.name:7s} {what:15s} -> {e}", flush=True)

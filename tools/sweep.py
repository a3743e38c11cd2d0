#!/usr/bin/env python3
"""Tuning sweep on one MI355X: GB/s of the count+checksum scan for a grid of options.
usage: python tools/sweep.py [--workload cfg3|cfg2] [--mib 1024] [--grid 'engine=tiered,darray;lds_budget=65536,98304;...']
Prints one line per configuration; every configuration's (count, checksum) must agree."""
import argparse
import itertools
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

import daachorse_amd as da
from daachorse_amd import Engine, ScanMode, synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--haystack", default="sparse")
    ap.add_argument("--mib", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--grid", default="engine=tiered,darray")
    ap.add_argument("--mode", default="overlapping", choices=["overlapping", "find", "leftmost"])
    args = ap.parse_args()
    pats = synth.patterns_cfg3() if args.workload == "cfg3" else synth.patterns_cfg3_wide() if args.workload == "cfg3w" else synth.patterns_cfg2()
    kind = da.MatchKind.LeftmostLongest if args.mode == "leftmost" else da.MatchKind.Standard
    mode = {"overlapping": ScanMode.FindOverlapping, "find": ScanMode.Find, "leftmost": ScanMode.LeftmostFind}[args.mode]
    blob = da.DoubleArrayAhoCorasickBuilder().match_kind(kind).build(pats).serialize()
    n = args.mib << 20
    hay = torch.empty(n, dtype=torch.uint8, device="cuda")
    wl = "cfg3" if args.workload == "cfg3w" else args.workload
    if args.haystack == "sparse":
        synth.device_uniform(hay, synth.SEEDS[f"{wl}_hay"], synth.ALPHA_WIDE_SPACE if args.workload == "cfg3w" else
                             synth.ALPHA_LOWER_SPACE if wl == "cfg3" else synth.ALPHA_PRINTABLE)
    else:
        synth.device_wordsoup(hay, synth.SEEDS[f"{wl}_dense"], pats, 20 if wl == "cfg3" else 13, noise_256=77 if wl == "cfg3" else 0,
                              alphabet=synth.ALPHA_WIDE if args.workload == "cfg3w" else synth.ALPHA_LOWER)
    res = torch.zeros(3, dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    keys, vals = [], []
    for part in args.grid.split(";"):
        k, v = part.split("=")
        keys.append(k)
        vals.append(v.split(","))
    ref = None
    defaults = {"overlap_micro": 1, "gram_version": 0, "gram2_dpp": 1, "gram_rank_in_lds": -1, "gram_dense": -1, "gram_ppl": 0, "gram_region": 0, "gram_lds_budget": 161792, "gram_slab": 4096, "restart_chain": 1, "seg_bytes": 0, "lds_budget": 96 * 1024, "dense_depth": -1, "rows_share_pct": 45, "blocks_per_cu": 0, "threads": 1024}
    for combo in itertools.product(*vals):
        cfg = dict(zip(keys, combo))
        for k, v in defaults.items():
            da.set_option(k, int(cfg.get(k, v)))
        eng = {"tiered": Engine.Tiered, "darray": Engine.DArray, "auto": Engine.Auto, "gram": Engine.Gram}[cfg.get("engine", "auto")]
        pma, _ = da.DoubleArrayAhoCorasick.deserialize(blob)
        try:
            pma.upload(0)
            info = pma.info()
            count_only = int(cfg.get("count_only", 0)) != 0   # `.count()` alone (daac_scan_count_only_range)
            run = (lambda: pma.count(mode, hay, engine=eng, stream=stream, result_dev=res.data_ptr())) if count_only else \
                (lambda: pma.scan_count(mode, hay, engine=eng, stream=stream, result_dev=res.data_ptr()))
            run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.reps
            r = res.tolist()
            cc = (int(r[0]), int(r[1]) & 0xFFFFFFFF, int(r[2]) & 0xFFFFFFFF)
            if count_only:
                cc = (cc[0],) + (ref[1:] if ref else (0, 0))
            ok = ref is None or cc == ref
            ref = ref or cc
            print(f"{cfg} used={da.last_engine()}  NA={info.tier_dense_states} NB={info.tier_lds_states} lds={info.tier_lds_bytes}  "
                  f"{ms:8.3f} ms  {n / ms / 1e6:9.1f} GB/s  frac={n / ms / 1e6 / 8000:.3f}  count={cc[0]} {'OK' if ok else 'MISMATCH'}", flush=True)
        except da.DaachorseError as e:
            print(f"{cfg}  ERROR {e}", flush=True)


if __name__ == "__main__":
    main()

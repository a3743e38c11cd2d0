#!/usr/bin/env python3
"""BASELINE.json configs[4]: CharwiseDoubleArrayAhoCorasick leftmost_find_iter (LeftmostLongest), 50 k UTF-8
patterns over a 1 GiB multi-byte haystack on one MI355X.  Not the headline bench line (bench.py is): a measurement
of the charwise path with a parity check against the CPU oracle on a prefix.
usage: python tools/bench_cfg5.py [--mib 1024] [--steps 5] [--mode leftmost|overlapping|find]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

import daachorse_amd as da
from daachorse_amd import ScanMode, synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mib", type=int, default=1024)
    ap.add_argument("--block-mib", type=int, default=64)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--mode", default="leftmost", choices=["leftmost", "overlapping", "find"])
    ap.add_argument("--cpu-mib", type=int, default=16)
    ap.add_argument("--opt", action="append", default=[])
    args = ap.parse_args()
    for kv in args.opt:
        k, v = kv.split("=")
        da.set_option(k, int(v))
    kind = da.MatchKind.LeftmostLongest if args.mode == "leftmost" else da.MatchKind.Standard
    mode = {"leftmost": ScanMode.LeftmostFind, "overlapping": ScanMode.FindOverlapping, "find": ScanMode.Find}[args.mode]
    pats = synth.patterns_cfg5()
    t0 = time.time()
    pma = da.CharwiseDoubleArrayAhoCorasickBuilder().match_kind(kind).build(pats)
    build_s = time.time() - t0
    pma.upload(0)
    info = pma.info()
    block = synth.cfg5_text_block(pats, args.block_mib << 20)
    reps = max(1, (args.mib << 20) // len(block))
    hay = torch.from_numpy(block.copy()).cuda().repeat(reps)
    nbytes = hay.numel()
    res = torch.zeros(3, dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    pma.scan_count(mode, hay, stream=stream, result_dev=res.data_ptr())
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    for a, b in ev:
        a.record()
        pma.scan_count(mode, hay, stream=stream, result_dev=res.data_ptr())
        b.record()
    torch.cuda.synchronize()
    ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    r = res.tolist()
    out = {"metric": f"haystack GB/s scanned (charwise {args.mode}, 50k UTF-8 patterns)", "value": round(nbytes / ms / 1e6, 2), "unit": "GB/s",
           "ms_per_step": round(ms, 3), "steps": args.steps, "dtype": "u8/u32 integer", "data": "synthetic",
           "config": {"workload": "cfg5: CharwiseDoubleArrayAhoCorasick, 50k patterns of 1-8 three-byte characters, word-soup text",
                      "haystack_bytes": nbytes, "num_states": info.num_states, "states_len": info.states_len,
                      "alphabet_size": info.alphabet_size, "automaton_bytes": info.heap_bytes, "host_build_seconds": round(build_s, 2)},
           "roofline": {"bound": "hbm", "achieved": round(nbytes / ms / 1e6, 2), "peak": 8000.0, "unit": "GB/s",
                        "frac": round(nbytes / ms / 1e6 / 8000.0, 5), "traffic": None},
           "match_count": int(r[0]), "matches_per_byte": round(int(r[0]) / nbytes, 4)}
    if args.cpu_mib > 0:
        from oracle import oracle as orc
        o = orc.OracleCharwisePma.deserialize(pma.serialize())
        n = min(len(block), args.cpu_mib << 20)
        n -= n % 3
        sample = block[:n]
        api = {"leftmost": "leftmost_find_iter", "overlapping": "find_overlapping_iter", "find": "find_iter"}[args.mode]
        t0 = time.perf_counter()
        want = getattr(o, api)(sample)
        dt = time.perf_counter() - t0
        got = pma.scan_count(mode, sample)
        out["cpu_baseline"] = {"value": round(n / dt / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port",
                               "sample": f"first {n >> 20} MiB of the text block, materialising oracle iterator",
                               "parity_with_gpu_on_sample": bool(got == (len(want), orc.matches_checksum(want)))}
    print(json.dumps(out))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py — haystack GB/s of the MI355X-native daachorse overlapping scan.

One "step" = one find_overlapping scan (count + checksum, `daac_scan_count`) of this rank's
haystack shard, resident in HBM, with the 100 000-pattern bytewise automaton (BASELINE.json
configs[2] = the configuration the metric is quoted on), plus the RCCL all-reduce of
{count, S1, S2} when more than one GPU takes part.  Shards are independent haystacks
(shard k is seeded 0xDAAC0014 + k), so scaling is weak and no data-path collective exists.

Prints ONE JSON line on rank 0 (see the contract in the task description).  Extra objects:
  roofline      dominant kernel vs the HBM roof: algorithmic bytes (= haystack bytes, 1 B read per
                haystack byte) / average kernel time measured with HIP events on the launch stream
  cpu_baseline  the C restatement of the reference's CPU path (oracle/, kind "port") on the host
                cores, on a bounded prefix of the same haystack, count + checksum checked against
                the GPU's for that prefix
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg3", choices=["cfg2", "cfg3"])
    ap.add_argument("--haystack", default="sparse", choices=["sparse", "dense"])
    ap.add_argument("--bytes", type=int, default=0, help="haystack bytes per GPU (default: the config's size)")
    ap.add_argument("--engine", default="auto", choices=["auto", "gram", "tiered", "darray"])
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--opt", action="append", default=[], help="name=value tuning option (daac_set_option)")
    ap.add_argument("--materialize-mib", type=int, default=64, help="also time a materialising scan of this prefix")
    args = ap.parse_args()

    import numpy as np
    import torch

    import daachorse_amd as da
    from daachorse_amd import Engine, ScanMode, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world != 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # one process per GPU; DAAC_DIST_BACKEND=gloo lets the control flow be exercised on a single-GPU box
    backend = os.environ.get("DAAC_DIST_BACKEND", "nccl")
    local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # RCCL over xGMI
        else:
            dist.init_process_group(backend)

    def reduce_counts(t):
        if backend == "nccl":
            dist.all_reduce(t)
        else:  # host-side reduction for backends without device tensors
            c = t.cpu()
            dist.all_reduce(c)
            t.copy_(c)

    for kv in args.opt:
        k, v = kv.split("=")
        da.set_option(k, int(v))
    engine = {"auto": Engine.Auto, "gram": Engine.Gram, "tiered": Engine.Tiered, "darray": Engine.DArray}[args.engine]
    mat_engine = Engine.Auto if engine == Engine.Gram else engine  # GRAM only counts

    # ---- automaton (host CPU, not timed) --------------------------------------------------------
    if args.workload == "cfg3":
        patterns = synth.patterns_cfg3()
        nbytes = args.bytes or (4 << 30)
        seed_sparse, seed_dense, alpha, slot = synth.SEEDS["cfg4_hay"] + rank if world > 1 else synth.SEEDS["cfg3_hay"], \
            synth.SEEDS["cfg3_dense"] + rank, synth.ALPHA_LOWER_SPACE, 20
        wl_name = "100k-pattern bytewise automaton (words_100000-style), 4 GiB haystack per GPU, find_overlapping count+checksum"
    else:
        patterns = synth.patterns_cfg2()
        nbytes = args.bytes or (256 << 20)
        seed_sparse, seed_dense, alpha, slot = synth.SEEDS["cfg2_hay"] + rank, synth.SEEDS["cfg2_dense"] + rank, \
            synth.ALPHA_PRINTABLE, 13
        wl_name = "1000-pattern bytewise automaton, 256 MiB random-ASCII haystack per GPU, find_overlapping count+checksum"
    t0 = time.time()
    pma = da.DoubleArrayAhoCorasick.new(patterns)
    build_s = time.time() - t0
    pma.upload(local_rank)
    info = pma.info()

    # ---- haystack shard, generated in HBM (not timed) ----------------------------------------------
    hay = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    if args.haystack == "sparse":
        synth.device_uniform(hay, seed_sparse, alpha)
    else:
        synth.device_wordsoup(hay, seed_dense, patterns, slot, noise_256=77 if args.workload == "cfg3" else 0)
    result = torch.zeros(3, dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        pma.scan_count(ScanMode.FindOverlapping, hay, engine=engine, stream=stream, result_dev=result.data_ptr())
        if dist is not None:
            reduce_counts(result)  # RCCL over xGMI: the trivial match-count reduction

    for _ in range(args.warmup):
        step()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        pma.scan_count(ScanMode.FindOverlapping, hay, engine=engine, stream=stream, result_dev=result.data_ptr())
        b.record()
        if dist is not None:
            reduce_counts(result)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = [a.elapsed_time(b) for a, b in ev]  # memset + scan kernel on the launch stream
    from daachorse_amd import dist as ddist
    elapsed = ddist.max_over_ranks(elapsed, device="cuda" if backend == "nccl" else None)
    total_count = int(result[0].item())
    checksum = ((int(result[1].item()) & 0xFFFFFFFF) << 32) | (int(result[2].item()) & 0xFFFFFFFF)

    if dist is not None:  # every rank leaves the group together; rank 0 reports on its own
        dist.barrier()
        dist.destroy_process_group()
        dist = None
    if rank != 0:
        return

    total_bytes = nbytes * world
    value = total_bytes * args.steps / elapsed / 1e9
    avg_kernel_s = float(np.mean(kernel_ms)) / 1e3
    achieved = nbytes / avg_kernel_s / 1e9

    out = {
        "metric": "haystack GB/s scanned (find_overlapping, 100k-pattern bytewise automaton)" if args.workload == "cfg3"
        else "haystack GB/s scanned (find_overlapping, 1000-pattern bytewise automaton)",
        "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8/u32 integer", "data": "synthetic",
        "config": {"workload": wl_name, "haystack": args.haystack, "haystack_bytes_per_gpu": nbytes,
                   "patterns": len(patterns), "engine": args.engine, "num_states": info.num_states,
                   "automaton_bytes": info.heap_bytes, "byte_classes": info.num_classes,
                   "lds_dense_states": info.tier_dense_states, "lds_states": info.tier_lds_states,
                   "lds_table_bytes": info.tier_lds_bytes, "gram_k": info.gram_k, "gram_lds_bytes": info.gram_lds_bytes,
                   "parallelism": f"haystack-shard x{world}",
                   "matches_per_byte": round(total_count / total_bytes, 4), "host_build_seconds": round(build_s, 2)},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                     "kernel": "daac::gram_count_kernel" if info.gram_available and args.engine in ("auto", "gram") else "daac::scan_kernel",
                     "kernel_ms": round(avg_kernel_s * 1e3, 4),
                     "algorithmic_bytes_per_launch": nbytes},
        "match_count": total_count, "match_checksum": f"{checksum:016x}" if world == 1 else None,
    }
    pmc = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(pmc):
        try:
            out["roofline"]["traffic"] = json.load(open(pmc)).get(f"{args.workload}_{args.haystack}_bytes_per_launch")
        except Exception:
            pass

    # ---- materialising scan of a prefix (reported, not the metric) -----------------------------------
    if args.materialize_mib > 0 and world == 1:
        n = min(nbytes, args.materialize_mib << 20)
        pma.scan(ScanMode.FindOverlapping, hay[:n], engine=mat_engine)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m = pma.scan(ScanMode.FindOverlapping, hay[:n], engine=mat_engine)
        dt = time.perf_counter() - t0
        out["materialize"] = {"bytes": n, "matches": int(len(m)), "seconds": round(dt, 4), "GB/s": round(n / dt / 1e9, 3),
                              "note": "count pass + scan + write pass + D2H of 24-byte tuples"}

    # ---- CPU baseline: the C restatement of the reference CPU path, on a bounded prefix ----------------
    if not args.no_cpu and world == 1:
        from oracle import oracle as orc
        o = orc.OraclePma.deserialize(pma.serialize())
        cores = os.cpu_count() or 1
        probe = hay[:8 << 20].cpu().numpy()
        t0 = time.perf_counter()
        o.overlapping_count(probe, threads=1)
        rate1 = len(probe) / (time.perf_counter() - t0)
        n = int(min(nbytes, max(64 << 20, rate1 * cores * args.cpu_seconds * 0.5)))
        n -= n % (1 << 20)
        sample = hay[:n].cpu().numpy()
        t0 = time.perf_counter()
        c1 = o.overlapping_count(sample[:n // max(1, cores // 2)], threads=1)
        dt1 = time.perf_counter() - t0
        t0 = time.perf_counter()
        cN = o.overlapping_count(sample, threads=cores)
        dtN = time.perf_counter() - t0
        gpu_cc = pma.scan_count(ScanMode.FindOverlapping, hay[:n], engine=engine)
        out["cpu_baseline"] = {"value": round(n / dtN / 1e9, 4), "unit": "GB/s", "cores": cores, "kind": "port",
                               "sample": f"first {n >> 20} MiB of the same haystack, {cores} threads with (Lmax-1)-byte halos",
                               "single_thread_GB/s": round((n // max(1, cores // 2)) / dt1 / 1e9, 4),
                               "parity_with_gpu_on_sample": bool(gpu_cc == cN)}
        del c1
    print(json.dumps(out))


if __name__ == "__main__":
    main()

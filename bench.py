#!/usr/bin/env python3
"""bench.py — haystack GB/s of the MI355X-native daachorse overlapping scan.

One "step" = one find_overlapping scan of this rank's part of the haystack, resident in HBM, with the 100 000-pattern
bytewise automaton (BASELINE.json configs[2] = the configuration the metric is quoted on), plus the RCCL all-reduce of
{count, S1, S2} when more than one GPU takes part.  No data-path collective exists: the automaton is replicated and the
haystack is sharded.  The scan is `find_overlapping_iter(haystack).count()` (`daac_scan_count_only_range`, --op count,
the default) or count + order-independent checksum of the (start, end, value) stream (`daac_scan_count_range`,
--op checksum); the line carries both at N = 1 (`with_checksum`), each checked against the CPU oracle.

  --scaling weak    (default) every rank owns a 4 GiB haystack of its own (shard k seeded 0xDAAC0014 + k: cfg4)
  --scaling strong  ONE haystack of --bytes (default 4 GiB) split over the ranks with daac_scan_count_range: rank r
                    generates bytes [lo_r - halo, hi_r) of the stream and counts the matches ending in (lo_r, hi_r]

`python bench.py --gpus N` launches itself: without WORLD_SIZE in the environment it re-executes under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`, one rank per GPU
(N must not exceed hipGetDeviceCount).  Under a launcher (WORLD_SIZE set) it is one rank of the job.

Prints ONE JSON line on rank 0.  Extra objects:
  roofline      dominant kernel vs the HBM roof: algorithmic bytes (= haystack bytes, 1 B read per haystack byte) /
                average kernel time measured with HIP events on the launch stream
  dense         the same automaton and kernel over the word-soup haystack (cfg3 (ii)), N = 1 only
  cpu_baseline  the C restatement of the reference's CPU path (oracle/, kind "port"), built -O3 -march=native on
                this host, timed on a bounded prefix of the same haystack with a 1/2/4/... thread ladder; count +
                checksum checked against the GPU's for that prefix
"""
import argparse
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)
ENGINE_NAMES = {0: "auto", 1: "tiered", 2: "darray", 3: "gram", 4: "pfx"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg3", choices=["cfg2", "cfg3"])
    ap.add_argument("--haystack", default="sparse", choices=["sparse", "dense"])
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--bytes", type=int, default=0, help="haystack bytes per GPU (weak) / in total (strong); default: the config's size")
    ap.add_argument("--engine", default="auto", choices=["auto", "gram", "tiered", "darray"])
    ap.add_argument("--op", default="count", choices=["count", "checksum"],
                    help="what a step computes: .count() of the iterator, or count + checksum of the match stream")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-dense", action="store_true", help="skip the extra dense-haystack object")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--opt", action="append", default=[], help="name=value tuning option (daac_set_option)")
    ap.add_argument("--materialize-mib", type=int, default=64, help="also time a materialising scan of this prefix")
    ap.add_argument("--no-extra", action="store_true", help="skip the restart-iterator, cfg5 and lazy-iterator objects")
    ap.add_argument("--plumbing", action="store_true",
                    help="no GPU, no scan: every rank contributes a fixed triple; exercises launch, rendezvous, reduction and the "
                         "JSON line only (used by the CPU test of the N > 1 path; prints value null)")
    return ap.parse_args(argv)


def hbm_traffic(workload_key, kernel_filter=None):
    """HBM bytes per launch of a workload's dominant kernel from the committed PMC passes (tools/profile_round.sh: rocprofv3 --pmc FETCH_SIZE and
    --pmc WRITE_SIZE in separate runs of this same command, warm-ups dropped; 2 * FETCH + WRITE per MI355X_MICROARCH.md's gfx950 correction).
    -> (bytes or None, where the figure comes from).  Counters cannot be collected inside the timed run itself."""
    path = os.environ.get("DAAC_HBM_TRAFFIC_JSON") or os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        t = json.load(open(path))
        kernels = t[workload_key]
        pick = [v for k, v in kernels.items() if kernel_filter is None or kernel_filter in k]
        best = max(pick, key=lambda v: v["bytes_per_launch"])
        name = [k for k, v in kernels.items() if v is best][0]
        return best["bytes_per_launch"], (f"{os.path.relpath(path, ROOT)} [{workload_key}][{name[:60]}]: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over this command "
                                          f"({best['launches']} timed launches), 2*FETCH + WRITE; " + str(t.get("_source", "")))
    except Exception:
        return None, None


def hbm_traffic_sum(workload_key, kernel_filters):
    """A workload served by several kernels in a row (find3: DETECT, BIN, the tails, SELECT): the sum of their bytes per launch.
    -> (bytes or None, {kernel: bytes})"""
    path = os.environ.get("DAAC_HBM_TRAFFIC_JSON") or os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        kernels = json.load(open(path))[workload_key]
        parts = {k[:48]: v["bytes_per_launch"] for k, v in kernels.items() if any(f in k for f in kernel_filters)}
        return (sum(parts.values()) if parts else None), parts
    except Exception:
        return None, None


def emitter_write_ratio(n_tuples, hay_bytes, tuple_bytes=16):
    """HBM bytes the three kernels of the tuple emitter WRITE per launch (WRITE_SIZE of the committed PMC pass over tools/time_emit.py 1024
    sparse: DETECT's annotated stream and records, BIN's bins, EXPAND's tuples) over (tuples + one byte per haystack byte): the bound the
    round-3 verdict set is 1.15.  -> (ratio or None, bytes or None)"""
    path = os.environ.get("DAAC_HBM_TRAFFIC_JSON") or os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        k = json.load(open(path))["emit"]
        want = ("emit3_detect_kernel", "emit3_bin_kernel", "emit3_expand_kernel<3, true" if tuple_bytes == 16 else "emit3_expand_kernel<3, false")
        w = 0
        for part in want:
            pick = [v for name, v in k.items() if part in name]
            if not pick:
                return None, None
            w += int(pick[0]["WRITE_SIZE_KiB"] * 1024)
        return round(w / (n_tuples * tuple_bytes + hay_bytes), 3), w
    except Exception:
        return None, None


def traffic_stale():
    """True when the counters in profiles/hbm_traffic.json were collected from other kernel sources than the ones this run was built from
    (tools/profile_round.sh stamps `_csrc_sha256` = daachorse_amd._build.source_hash()); None when there is no stamp to compare."""
    path = os.environ.get("DAAC_HBM_TRAFFIC_JSON") or os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        from daachorse_amd import _build
        stamp = json.load(open(path)).get("_csrc_sha256")
        return None if stamp is None else stamp != _build.source_hash()
    except Exception:
        return None


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args):
    """--gpus N without a launcher: check N against the device count and re-exec under torchrun."""
    if not args.plumbing:
        import torch
        have = torch.cuda.device_count()  # hipGetDeviceCount
        if args.gpus > have and os.environ.get("DAAC_BENCH_OVERSUBSCRIBE") != "1":
            raise SystemExit(f"bench.py: --gpus {args.gpus} but this node exposes {have} GPU(s) (hipGetDeviceCount); "
                             f"set DAAC_BENCH_OVERSUBSCRIBE=1 to share devices between ranks (tests only)")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def cpu_limits():
    """cores this process may really use: affinity mask and cgroup CPU quota"""
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else float(q) / float(per)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            quota = None if q <= 0 else q / per
        except Exception:
            pass
    return usable, quota


def plumbing(args, rank, world):
    """launch + rendezvous + reduction + reporting without a device (gloo)"""
    import torch
    import torch.distributed as dist
    from daachorse_amd import dist as ddist
    if world > 1:
        dist.init_process_group("gloo")
    t0 = time.perf_counter()
    tot = None
    for _ in range(args.warmup + args.steps):
        tot = ddist.all_reduce_counts(rank + 1, 10 * (rank + 1), 100 * (rank + 1))
    if world > 1:
        dist.barrier()
    elapsed = ddist.max_over_ranks(time.perf_counter() - t0)
    if world > 1:
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "plumbing only (no scan)", "value": None, "unit": "GB/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(elapsed / max(1, args.steps) * 1e3, 4), "higher_is_better": True,
                          "scaling": args.scaling, "vs_baseline": None, "plumbing_only": True, "reduced": [tot[0], tot[1]]}))


def _event_ms(torch, fn, reps):
    """average milliseconds of fn() over `reps` launches, HIP events on the current stream (one warm-up)"""
    fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in ev) / reps


def restart_legs(da, synth, torch, patterns, local_rank, stream, result, no_cpu, seed_sparse, seed_dense, alpha):
    import numpy as np
    """find_iter / leftmost_find_iter (bytewise/iter.rs:58-113, 272-340) of the cfg3 dictionary: count + checksum over 1 GiB of the
    sparse and of the dense haystack, under the same clock as the headline number; prefix parity against the oracle."""
    from daachorse_amd import ScanMode
    n = 1 << 30
    hay = torch.empty(n, dtype=torch.uint8, device="cuda")
    out = {"bytes": n, "op": "count + checksum of the iterator's match stream (daac_scan_count)"}
    for kind_name, kind, mode, api in (("find_iter", da.MatchKind.Standard, ScanMode.Find, "find_iter"),
                                       ("leftmost_find_iter", da.MatchKind.LeftmostLongest, ScanMode.LeftmostFind, "leftmost_find_iter")):
        pma = da.DoubleArrayAhoCorasickBuilder().match_kind(kind).build(patterns)
        pma.upload(local_rank)
        oo = None
        if not no_cpu:
            from oracle import oracle as orc
            oo = orc.OraclePma.deserialize(pma.serialize())
        for hk in ("sparse", "dense"):
            if hk == "sparse":
                synth.device_uniform(hay, seed_sparse, alpha)
            else:
                synth.device_wordsoup(hay, seed_dense, patterns, 20, noise_256=77)
            torch.cuda.synchronize()
            fn = lambda: pma.scan_count(mode, hay, stream=stream, result_dev=result.data_ptr())
            ms = _event_ms(torch, fn, 3)
            r = result.tolist()
            ok = None
            if oo is not None:
                pn = 16 << 20
                sample = hay[:pn].cpu().numpy()
                want = getattr(oo, api)(sample)
                ok = bool(pma.scan_count(mode, sample) == (len(want), orc.matches_checksum(want)))
            eng = ENGINE_NAMES.get(da.last_engine(), "?")
            tr = hbm_traffic(f"{'find' if kind_name == 'find_iter' else 'leftmost'}_{hk}", "chain")
            if eng == "gram":   # find3 / left3: DETECT + BIN of the tuple emitter, the tiles' tails, SELECT — one launch each per GiB
                tr = hbm_traffic_sum(f"{'find' if kind_name == 'find_iter' else 'leftmost'}_{hk}", ("find3_", "left3_", "emit3_detect", "emit3_bin"))
            out[f"{kind_name}_{hk}"] = {"value": round(n / ms / 1e6, 2), "unit": "GB/s", "frac": round(n / ms / 1e6 / HBM_PEAK_GBS, 4), "kernel_ms": round(ms, 3),
                                        "engine_used": eng, "match_count": int(r[0]),
                                        "matches_per_byte": round(int(r[0]) / n, 4), "traffic": tr[0], "parity_16mib_prefix_vs_oracle": ok}
            if hk == "sparse":   # the iterator's tuple LIST left in HBM (daac_scan_device16 in this mode), wall time of the call, best of 3
                da.set_option("max_result_bytes", 64 << 30)
                best, cnt_list = None, None
                for _ in range(4):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    dm = pma.scan_device(mode, hay, fmt16=True)
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t0
                    cnt_list = dm.count
                    dm.free()
                    best = dt if best is None else min(best, dt)
                da.set_option("max_result_bytes", 8 << 30)
                okl = None
                if oo is not None:
                    dm = pma.scan_device(mode, hay[:pn], fmt16=True)
                    got = dm.to_numpy()
                    dm.free()
                    okl = bool(len(got) == len(want) and np.array_equal(got["end"], want["end"]) and np.array_equal(got["value"], want["value"]) and
                               np.array_equal(got["length"].astype(np.uint64), want["end"] - want["start"]))
                out[f"{kind_name}_{hk}"]["tuples_device"] = {"GB/s": round(n / best / 1e9, 2), "seconds": round(best, 5), "tuples": int(cnt_list),
                                                             "count_agrees": bool(cnt_list == int(r[0])), "engine_used": ENGINE_NAMES.get(da.last_engine(), "?"),
                                                             "list_of_16mib_prefix_equals_oracle": okl,
                                                             "note": "daac_scan_device16 in this mode: the iterator's matches as {end, length, value} in its order, left in HBM"}
            if eng == "gram":
                out[f"{kind_name}_{hk}"]["method"] = ("find3" if kind_name == "find_iter" else "left3") + ": selection over the emitter's per-position flags, no state chain (the chain walkers serve text made of dictionary words)"
        del pma
    del hay
    torch.cuda.empty_cache()
    return out


def cfg5_legs(da, synth, torch, local_rank, stream, result, no_cpu):
    """BASELINE.json configs[4]: CharwiseDoubleArrayAhoCorasick, 50 k UTF-8 patterns, the full 1 GiB multi-byte haystack:
    leftmost_find_iter (LeftmostLongest; charwise/iter.rs:328-399), find_iter, find_overlapping_iter — count + checksum each."""
    from daachorse_amd import ScanMode
    pats = synth.patterns_cfg5()
    nbytes = synth.cfg5_haystack_bytes(1 << 30)
    hay = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    synth.device_zipf_text(hay)
    torch.cuda.synchronize()
    out = {"bytes": nbytes, "workload": "cfg5 (SURVEY 8d): 50k patterns of 2-8 scalars, Zipf(1.0) over 6000 symbols; text of the same distribution + 10% ASCII"}
    for name, kind, mode, api in (("leftmost_find_iter", da.MatchKind.LeftmostLongest, ScanMode.LeftmostFind, "leftmost_find_iter"),
                                  ("find_iter", da.MatchKind.Standard, ScanMode.Find, "find_iter"),
                                  ("find_overlapping_iter", da.MatchKind.Standard, ScanMode.FindOverlapping, "find_overlapping_iter")):
        pma = da.CharwiseDoubleArrayAhoCorasickBuilder().match_kind(kind).build(pats)
        pma.upload(local_rank)
        fn = lambda: pma.scan_count(mode, hay, stream=stream, result_dev=result.data_ptr())
        ms = _event_ms(torch, fn, 3)
        r = result.tolist()
        ok = None
        if not no_cpu:
            from oracle import oracle as orc
            oo = orc.OracleCharwisePma.deserialize(pma.serialize())
            pn = synth.cfg5_haystack_bytes(16 << 20)
            sample = hay[:pn].cpu().numpy()
            want = getattr(oo, api)(sample)
            ok = bool(pma.scan_count(mode, sample) == (len(want), orc.matches_checksum(want)))
        tr = hbm_traffic({"leftmost_find_iter": "cfg5_leftmost", "find_iter": "cfg5_find", "find_overlapping_iter": "cfg5_overlapping"}[name], "daac::char_")
        out[name] = {"value": round(nbytes / ms / 1e6, 2), "unit": "GB/s", "frac": round(nbytes / ms / 1e6 / HBM_PEAK_GBS, 4), "kernel_ms": round(ms, 3),
                     "engine_used": ENGINE_NAMES.get(da.last_engine(), "?"), "match_count": int(r[0]), "matches_per_byte": round(int(r[0]) / nbytes, 4),
                     "traffic": tr[0], "parity_16mib_prefix_vs_oracle": ok}
        del pma
    del hay
    torch.cuda.empty_cache()
    return out


def multi_legs(da, synth, torch, pma_cfg3, local_rank, alpha):
    """daac_scan_count_multi (the product's multi-device entry point, SURVEY 8e / BASELINE configs[3]) with every visible device — on a
    one-GPU box: 8 virtual shards of 512 MiB that all name device 0 (the shard arithmetic, the host threads and the host-side sum are
    those of the 8-GPU node; the shards then share one device, so the figure is a single device's).  Shard k is seeded cfg4's
    0xDAAC0014 + k and carries the last max_pattern_len - 1 bytes of shard k - 1 in front; checked against one scan of the shards laid
    end to end."""
    from daachorse_amd import ScanMode
    ndev = torch.cuda.device_count()
    nsh, shard = 8, 512 << 20
    halo = pma_cfg3.info().max_pattern_len - 1
    whole = torch.empty(nsh * shard, dtype=torch.uint8, device="cuda")
    for k in range(nsh):
        synth.device_uniform(whole[k * shard:(k + 1) * shard], synth.SEEDS["cfg4_hay"] + k, alpha)
    shards = []
    keep = []
    for k in range(nsh):
        dev = k % ndev
        h = halo if k else 0
        piece = whole[k * shard - h:(k + 1) * shard]
        if dev != torch.cuda.current_device():
            piece = piece.to(f"cuda:{dev}")
            keep.append(piece)
        shards.append((dev, piece, h, k * shard))
    torch.cuda.synchronize()
    out = {"shards": nsh, "shard_bytes": shard, "devices": ndev, "op": "daac_scan_count_multi: one persistent worker thread + stream per device, a device's shards queued back to back and waited for once, counts (+ checksum sums) added on the host"}
    for name, cs in (("count", False), ("count_checksum", True)):
        best, got = None, None
        for _ in range(3):
            t0 = time.perf_counter()
            got = da.scan_count_multi(pma_cfg3, ScanMode.FindOverlapping, shards, checksum=cs)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        one = pma_cfg3.scan_count(ScanMode.FindOverlapping, whole) if cs else pma_cfg3.count(ScanMode.FindOverlapping, whole)
        out[name] = {"GB/s": round(nsh * shard / best / 1e9, 2), "seconds": round(best, 5), "equals_one_scan_of_the_whole": bool(got == one)}
    return out


def stream_legs(da, synth, torch, np, pma_cfg3, local_rank, seed_sparse, alpha, no_cpu):
    """The chunk-fed steppers (daac_stream_*: FindOverlappingStepper / FindStepper / find_overlapping_no_suffix fed 64 MiB device chunks, SURVEY
    8f-2): every feed scans [kept tail | chunk] on the engines the eager scans use and hands the chunk's matches to the host; 512 MiB of the cfg3
    haystack (bound by the 24-byte tuples' way back over PCIe) and of cfg2's sparse haystack; the first chunks' tuples against the oracle."""
    from daachorse_amd import ScanMode
    n, chunk = 512 << 20, 64 << 20
    out = {"bytes": n, "chunk_bytes": chunk, "op": "daac_stream_open + daac_stream_feed over 64 MiB device chunks, every chunk's tuple list handed to the host (24-byte tuples; `compact`: daac_stream_feed_compact, 8-byte tuples)"}
    dev = torch.empty(n, dtype=torch.uint8, device="cuda")
    p2 = da.DoubleArrayAhoCorasick.new(synth.patterns_cfg2())
    p2.upload(local_rank)
    for name, pma, fill in (("cfg3", pma_cfg3, lambda: synth.device_uniform(dev, seed_sparse, alpha)),
                            ("cfg2_sparse", p2, lambda: synth.device_uniform(dev, synth.SEEDS["cfg2_hay"], synth.ALPHA_PRINTABLE))):
        fill()
        torch.cuda.synchronize()
        leg = {}
        for mode_name, make in (("find_overlapping_stepper", pma.find_overlapping_stepper), ("find_stepper", pma.find_stepper),
                                ("find_overlapping_no_suffix_stepper", pma.find_overlapping_no_suffix_stepper)):
            best, cnt, first = None, 0, None
            for rep in range(2):
                st = make()
                cnt = 0
                t0 = time.perf_counter()
                for b in range(0, n, chunk):
                    m = st.feed(dev[b:b + chunk])
                    cnt += len(m)
                    if b == 0 and rep == 0:
                        first = np.array(m[:200000], copy=True)
                dt = time.perf_counter() - t0
                del st
                best = dt if best is None else min(best, dt)
            leg[mode_name] = {"GB/s": round(n / best / 1e9, 2), "seconds": round(best, 4), "matches": cnt,
                              "engine_used": ENGINE_NAMES.get(da.last_engine(), "?")}
            # the same stream through daac_stream_feed_compact: 8-byte tuples in the stream object's page-locked block (round 6)
            best8, cnt8, first8 = None, 0, None
            for rep in range(2):
                st = make()
                cnt8 = 0
                t0 = time.perf_counter()
                for b in range(0, n, chunk):
                    run, base, bits = st.feed_compact(dev[b:b + chunk])
                    cnt8 += len(run)
                    if b == 0 and rep == 0:
                        first8 = st.decode8(run[:200000], base, bits)
                dt = time.perf_counter() - t0
                del st
                best8 = dt if best8 is None else min(best8, dt)
            leg[mode_name]["compact"] = {"GB/s": round(n / best8 / 1e9, 2), "seconds": round(best8, 4), "matches": cnt8,
                                         "same_first_tuples_as_feed": bool(cnt8 == cnt and first8 is not None and len(first8) == len(first) and
                                                                           np.array_equal(first8["start"], first["start"]) and np.array_equal(first8["end"], first["end"]) and
                                                                           np.array_equal(first8["value"], first["value"]))}
            if not no_cpu and mode_name != "find_overlapping_no_suffix_stepper":
                from oracle import oracle as orc
                o = orc.OraclePma.deserialize(pma.serialize())
                pre = dev[:8 << 20].cpu().numpy()
                want = (o.find_overlapping_iter(pre) if mode_name == "find_overlapping_stepper" else o.find_iter(pre))[:len(first)]
                k = min(len(want), len(first), 100000)   # (the prefix: what lies well inside the first 8 MiB)
                leg[mode_name]["parity_first_tuples_vs_oracle"] = bool(k > 0 and np.array_equal(first["start"][:k], want["start"][:k]) and
                                                                     np.array_equal(first["end"][:k], want["end"][:k]) and np.array_equal(first["value"][:k], want["value"][:k]))
        out[name] = leg
    return out


def iterator_legs(da, synth, torch, np, pma_cfg3, local_rank, seed_sparse, alpha):
    """Iterator::next driven to exhaustion over HOST haystacks (page-locked): daac_iter_next_batch run by run, every tuple looked at
    (runs counted; a third pass also sums the ends on one host thread).  cfg3 (0.6 matches per byte: bound by the tuples' way back over PCIe) and cfg2's sparse haystack
    (bound by the haystack's way to the device)."""
    from daachorse_amd import ScanMode
    n = 1 << 30
    out = {"bytes": n, "op": "daac_iter_open_compact + daac_iter_next_batch8 to exhaustion over a page-locked host haystack (zero-copy runs of 8-byte tuples, counted)",
           "window_bytes": "16, 32, then 64 MiB (option iter_window)"}
    host = torch.empty(n, dtype=torch.uint8).pin_memory()
    dev = torch.empty(n, dtype=torch.uint8, device="cuda")
    p2 = da.DoubleArrayAhoCorasick.new(synth.patterns_cfg2())
    p2.upload(local_rank)
    for name, pma, fill in (("cfg3", pma_cfg3, lambda: synth.device_uniform(dev, seed_sparse, alpha)),
                            ("cfg2_sparse", p2, lambda: synth.device_uniform(dev, synth.SEEDS["cfg2_hay"], synth.ALPHA_PRINTABLE))):
        fill()
        host.copy_(dev)
        torch.cuda.synchronize()
        h = host.numpy()
        best, cnt, best_sum, ends, best16, cnt16 = None, 0, None, 0, None, 0
        for rep in range(3):   # the first pass also pays for pinning the window buffers; the last one reads every tuple on the host
            t0 = time.perf_counter()
            it = pma.find_overlapping_iter(h, compact=True)
            cnt, ends = 0, 0
            while True:
                got = it.next_batch8()
                if got is None:
                    break
                run, base, eb = got
                cnt += len(run)
                if rep == 2:
                    ends += int((run["end_len"] & np.uint32((1 << eb) - 1)).sum(dtype=np.uint64)) + base * len(run)
            it.close()
            dt = time.perf_counter() - t0
            if rep == 2:
                best_sum = dt
            else:
                best = dt if best is None else min(best, dt)
        for rep in range(2):   # the 16-byte runs (daac_iter_open / daac_iter_next_batch) beside it
            t0 = time.perf_counter()
            it = pma.find_overlapping_iter(h)
            cnt16 = 0
            while True:
                run = it.next_batch()
                if run is None:
                    break
                cnt16 += len(run)
            it.close()
            dt = time.perf_counter() - t0
            best16 = dt if best16 is None else min(best16, dt)
        if name == "cfg3":   # find_iter() of the same automaton over the same host haystack (its windows' lists come from the selection kernels)
            bestf, cntf = None, 0
            for rep in range(2):
                t0 = time.perf_counter()
                it = pma.find_iter(h, compact=True)
                cntf = 0
                while True:
                    got = it.next_batch8()
                    if got is None:
                        break
                    cntf += len(got[0])
                it.close()
                dt = time.perf_counter() - t0
                bestf = dt if bestf is None else min(bestf, dt)
            wantf = pma.count(ScanMode.Find, dev)
            out["cfg3_find_iter"] = {"GB/s": round(n / bestf / 1e9, 2), "seconds": round(bestf, 4), "matches": cntf, "matches_per_byte": round(cntf / n, 4),
                                     "tuple_GB/s_over_pcie": round(cntf * 8 / bestf / 1e9, 2), "count_agrees_with_count_kernel": bool(cntf == wantf),
                                     "engine_used": ENGINE_NAMES.get(da.last_engine(), "?"),
                                     "op": "pma.find_iter(host haystack) to exhaustion, 8-byte tuples (daac_iter_open_compact in DAAC_FIND mode)"}
        want = pma.count(ScanMode.FindOverlapping, dev)
        out[name] = {"GB/s": round(n / best / 1e9, 2), "seconds": round(best, 4), "matches": cnt, "matches_per_byte": round(cnt / n, 4),
                     "wire": "8-byte tuples (daac_iter_open_compact / daac_iter_next_batch8: {value, end relative to the window | length << end_bits})",
                     "tuple_GB/s_over_pcie": round(cnt * 8 / best / 1e9, 2), "count_agrees_with_count_kernel": bool(cnt == want and cnt16 == want),
                     "GB/s_16_byte_runs": round(n / best16 / 1e9, 2),
                     "GB/s_with_a_numpy_pass_over_every_tuple": round(n / best_sum / 1e9, 2),
                     "engine_used": ENGINE_NAMES.get(da.last_engine(), "?")}
    del host, dev, p2
    torch.cuda.empty_cache()
    return out


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)  # does not return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if args.plumbing:
        return plumbing(args, rank, world)

    import numpy as np
    import torch

    import daachorse_amd as da
    from daachorse_amd import Engine, ScanMode, synth
    from daachorse_amd import dist as ddist

    ndev = torch.cuda.device_count()
    if ndev == 0:
        raise SystemExit("bench.py: no GPU visible (the scan has no CPU path)")
    if world > ndev and os.environ.get("DAAC_BENCH_OVERSUBSCRIBE") != "1":
        raise SystemExit(f"bench.py: {world} ranks but {ndev} GPU(s) on this node")
    # one process per GPU; DAAC_DIST_BACKEND=gloo lets the control flow be exercised on a single-GPU box
    backend = os.environ.get("DAAC_DIST_BACKEND", "nccl")
    local_rank = local_rank % ndev
    torch.cuda.set_device(local_rank)
    dist = None
    # DAAC_BENCH_FORCE_DIST=1: take the distributed branch (process group, device-tensor all-reduce, barriers) at world size 1 too —
    # the -m gpu suite proves RCCL initialisation and the reduce on the one-GPU box that way (RANK / WORLD_SIZE / MASTER_* set)
    if world > 1 or os.environ.get("DAAC_BENCH_FORCE_DIST") == "1":
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
        total_default = 4 << 30
        seed_sparse = synth.SEEDS["cfg4_hay"] + rank if (world > 1 and args.scaling == "weak") else synth.SEEDS["cfg3_hay"]
        seed_dense, alpha, slot, noise = synth.SEEDS["cfg3_dense"] + (rank if args.scaling == "weak" else 0), synth.ALPHA_LOWER_SPACE, 20, 77
        wl_name = "cfg3: 100k-pattern bytewise automaton (words_100000-style), 4 GiB haystack per GPU, find_overlapping"
    else:
        patterns = synth.patterns_cfg2()
        total_default = 256 << 20
        seed_sparse = synth.SEEDS["cfg2_hay"] + (rank if args.scaling == "weak" else 0)
        seed_dense, alpha, slot, noise = synth.SEEDS["cfg2_dense"] + (rank if args.scaling == "weak" else 0), synth.ALPHA_PRINTABLE, 13, 0
        wl_name = "cfg2: 1000-pattern bytewise automaton, 256 MiB random-ASCII haystack per GPU, find_overlapping"
    t0 = time.time()
    pma = da.DoubleArrayAhoCorasick.new(patterns)
    build_s = time.time() - t0
    pma.upload(local_rank)
    info = pma.info()

    # ---- this rank's part of the haystack, generated in HBM (not timed) ---------------------------------
    total = args.bytes or total_default
    if args.scaling == "weak":
        lo, hi, lead = 0, total, 0          # a haystack of its own
        job_bytes = total * world
    else:
        lo, hi = ddist.shard_range(total, rank, world)
        lead = min(lo, (info.max_pattern_len + 15) & ~15)  # bytes before lo that the range scan may read (>= Lmax - 1), 16-aligned
        job_bytes = total
    nbytes = hi - lo                         # bytes this rank accounts for per step
    hay = torch.empty(max(16, nbytes + lead), dtype=torch.uint8, device="cuda")[:nbytes + lead]

    def fill(kind):
        if kind == "sparse":
            synth.device_uniform(hay, seed_sparse, alpha, offset=lo - lead)
        else:
            synth.device_wordsoup(hay, seed_dense, patterns, slot, noise_256=noise, offset=lo - lead)
        torch.cuda.synchronize()

    fill(args.haystack)
    result = torch.zeros(3, dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    base32 = (lo - lead) & 0xFFFFFFFF

    op = {"v": args.op}

    def scan_step():
        if op["v"] == "count":
            pma.count(ScanMode.FindOverlapping, hay, engine=engine, stream=stream, result_dev=result.data_ptr(), begin=lead)
            return
        pma.scan_count(ScanMode.FindOverlapping, hay, engine=engine, stream=stream, result_dev=result.data_ptr(), begin=lead)
        if base32:  # ends were relative to this rank's buffer: S2 += S1 * (offset of the buffer in the haystack)  (mod 2^32)
            result[2] += (result[1] & 0xFFFFFFFF) * base32

    diag = {}  # what the last timed() saw on THIS rank: kernel and reduce times by HIP events

    def timed(steps, warmup):
        for _ in range(warmup):
            scan_step()
            if dist is not None:
                reduce_counts(result)  # RCCL over xGMI: the trivial match-count reduction
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for a, b, c in ev:
            a.record()
            scan_step()
            b.record()
            if dist is not None:
                reduce_counts(result)
                c.record()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        kernel_ms = [a.elapsed_time(b) for a, b, _ in ev]  # memset + scan kernel on the launch stream
        diag["kernel_ms"] = float(np.mean(kernel_ms))
        diag["reduce_ms"] = float(np.mean([b.elapsed_time(c) for _, b, c in ev])) if dist is not None else None
        diag["elapsed_s"] = elapsed
        return ddist.max_over_ranks(elapsed, device="cuda" if backend == "nccl" else None), float(np.mean(kernel_ms)) / 1e3

    elapsed, avg_kernel_s = timed(args.steps, args.warmup)
    engine_used = ENGINE_NAMES.get(da.last_engine(), "?")
    total_count = int(result[0].item())
    checksum = ((int(result[1].item()) & 0xFFFFFFFF) << 32) | (int(result[2].item()) & 0xFFFFFFFF) if args.op == "checksum" else None
    v2 = info.gram2_available and engine_used == "gram" and args.op == "count"

    dist_used = None
    if dist is not None:  # every rank leaves the group together; rank 0 reports on its own
        dist_used = {"backend": backend, "world_size": dist.get_world_size(), "all_reduces": args.steps + args.warmup}
        # what each rank measured, gathered to rank 0: where a non-linear scaling curve comes from (a slow rank, the reduce, the launch)
        per_rank = torch.tensor([diag["kernel_ms"], diag["reduce_ms"] or 0.0, diag["elapsed_s"] * 1e3 / max(1, args.steps), float(nbytes)],
                                dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        gathered = [torch.zeros_like(per_rank) for _ in range(dist.get_world_size())]
        dist.all_gather(gathered, per_rank)
        rows = [[float(x) for x in g.tolist()] for g in gathered]
        km = [r[0] for r in rows]
        dist_used.update({
            "n_ranks_seen": len(rows),
            "per_rank_kernel_ms": {"min": round(min(km), 4), "max": round(max(km), 4), "mean": round(sum(km) / len(km), 4),
                                   "slowest_rank": int(km.index(max(km))), "all": [round(x, 4) for x in km]},
            "reduce_ms": {"mean_over_ranks": round(sum(r[1] for r in rows) / len(rows), 4), "max": round(max(r[1] for r in rows), 4),
                          "note": "HIP events around the all-reduce of {count, S1, S2} on the launch stream (includes waiting for the slowest rank)"},
            "per_rank_ms_per_step": [round(r[2], 4) for r in rows],
            "per_rank_bytes": [int(r[3]) for r in rows],
            "rccl_version": (".".join(str(x) for x in torch.cuda.nccl.version()) if backend == "nccl" and hasattr(torch.cuda, "nccl") else None),
        })
        # strong scaling: the job's count (+ checksum) must be what ONE device finds in the whole haystack — checked here, not only in the tests
        if args.scaling == "strong" and world > 1 and total <= (8 << 30):
            ok = None
            if rank == 0:
                whole = torch.empty(total, dtype=torch.uint8, device="cuda")
                if args.haystack == "sparse":
                    synth.device_uniform(whole, seed_sparse, alpha)
                else:
                    synth.device_wordsoup(whole, seed_dense, patterns, slot, noise_256=noise)
                if args.op == "count":
                    ok = bool(pma.count(ScanMode.FindOverlapping, whole, engine=engine) == int(result[0].item()))
                else:
                    one = pma.scan_count(ScanMode.FindOverlapping, whole, engine=engine)
                    ok = bool(one == (int(result[0].item()), ((int(result[1].item()) & 0xFFFFFFFF) << 32) | (int(result[2].item()) & 0xFFFFFFFF)))
                del whole
            dist_used["strong_equals_one_rank"] = ok
        dist.barrier()
        dist.destroy_process_group()
        dist = None
    if rank != 0:
        return

    value = job_bytes * args.steps / elapsed / 1e9
    achieved = nbytes / avg_kernel_s / 1e9
    gram = engine_used == "gram"

    out = {
        # (round 1 timed count + checksum under the metric string of round 2; since round 3 the string names the op, and the
        # count + checksum figure rides along as `with_checksum` / `value_count_checksum` for like-for-like comparison)
        "metric": f"haystack GB/s scanned (find_overlapping_iter(..){'.count()' if args.op == 'count' else ' count + checksum'}, "
                  f"{'100k' if args.workload == 'cfg3' else '1000'}-pattern bytewise automaton)",
        "op": "find_overlapping_iter(haystack).count()" if args.op == "count" else "count + checksum of the find_overlapping match stream",
        "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": "u8/u32 integer", "data": "synthetic",
        "config": {"workload": wl_name, "haystack": args.haystack, "haystack_bytes_per_gpu": nbytes, "haystack_bytes_job": job_bytes,
                   "patterns": len(patterns), "engine": args.engine, "engine_used": engine_used, "num_states": info.num_states,
                   "automaton_bytes": info.heap_bytes, "byte_classes": info.num_classes,
                   "lds_dense_states": info.tier_dense_states, "lds_states": info.tier_lds_states,
                   "lds_table_bytes": info.tier_lds_bytes, "gram_k": info.gram_k, "gram_lds_bytes": info.gram_lds_bytes,
                   "gram2_lds_bytes_count": info.gram2_lds_count, "gram2_lds_bytes_checksum": info.gram2_lds_exact,
                   "parallelism": f"haystack-shard x{world} ({args.scaling})",
                   "matches_per_byte": round(total_count / job_bytes, 4), "host_build_seconds": round(build_s, 2)},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "traffic_source": None,
                     "kernel": ("daac::gram4_kernel" if v2 else "daac::gram_count_kernel") if gram else "daac::scan_kernel",
                     "kernel_ms": round(avg_kernel_s * 1e3, 4),
                     "algorithmic_bytes_per_launch": nbytes},
        "match_count": total_count, "match_checksum": f"{checksum:016x}" if checksum is not None else None,
        "distributed": dist_used,
    }
    # what the METHOD allows on this chip, measured: the same kernel with everything behind the per-position lookup compiled out
    # (tools/method_ceiling.sh -> profiles/method_ceiling.json; a figure from a separate run, like `traffic`)
    try:
        mc = json.load(open(os.path.join(ROOT, "profiles", "method_ceiling.json")))
        key = f"{args.workload}_{args.haystack}"
        if gram and args.op == "count" and key in mc:
            out["roofline"]["method_ceiling"] = {"value": mc[key]["GB/s"], "unit": "GB/s", "frac": mc[key]["frac"],
                                                 "achieved_over_ceiling": round(achieved / mc[key]["GB/s"], 3),
                                                 "what": mc["what"], "source": "profiles/method_ceiling.json (" + str(mc.get("round")) + ", " + str(mc.get("build")) + ")"}
    except Exception:
        pass
    tr = hbm_traffic({"cfg3": "cfg3", "cfg2": "cfg2"}[args.workload] + ("_count" if args.workload == "cfg2" else f"_{args.haystack}_{args.op}"),
                     out["roofline"]["kernel"].split("::")[-1])  # the entry of the kernel the roofline names
    out["roofline"]["traffic"], out["roofline"]["traffic_source"] = tr
    out["roofline"]["traffic_stale"] = traffic_stale()

    # ---- tuples (reported, not the metric): device-resident list of a 1 GiB prefix in both device formats, and a list copied to the host ----
    if args.materialize_mib > 0 and world == 1:
        da.set_option("max_result_bytes", 64 << 30)
        n = min(nbytes, 1 << 30)
        for key, fmt16, tb in (("tuples_device", True, 16), ("tuples_device_24", False, 24)):
            dm = pma.scan_device(ScanMode.FindOverlapping, hay[:n], engine=mat_engine, fmt16=fmt16)
            dm.free()
            torch.cuda.synchronize()
            dt = None
            for _ in range(3):   # (best of three calls: a call is ~5 ms, and the allocator's first 10 GB block costs the first one)
                t0 = time.perf_counter()
                dm = pma.scan_device(ScanMode.FindOverlapping, hay[:n], engine=mat_engine, fmt16=fmt16)
                torch.cuda.synchronize()
                d1 = time.perf_counter() - t0
                dt = d1 if dt is None else min(dt, d1)
                n_t = int(dm.count)
                dm.free()
            ratio, wbytes = emitter_write_ratio(n_t, n, tb) if (args.workload == "cfg3" and args.haystack == "sparse" and n == 1 << 30) else (None, None)
            out[key] = {"bytes": n, "matches": n_t, "tuple_bytes": tb, "seconds": round(dt, 5), "GB/s": round(n / dt / 1e9, 2),
                        "tuple_GB/s": round(n_t * tb / dt / 1e9, 1), "engine_used": ENGINE_NAMES.get(da.last_engine(), "?"),
                        "hbm_write_bytes": wbytes, "write_traffic_over_tuples_plus_text": ratio,
                        "note": ("daac_scan_device16: {end u64, length u32, value u32} = the crate's Match fields" if fmt16 else
                                 "daac_scan_device: daac_match {start, end, value, pad}") +
                                ", reference order, left in HBM; wall time of the call (DETECT, scans of the tile counts, BIN, allocation, EXPAND), best of 3; "
                                "write traffic: WRITE_SIZE of the emitter's kernels from the committed PMC pass (profiles/hbm_traffic.json [emit])"}
        n = min(nbytes, args.materialize_mib << 20)
        pma.scan(ScanMode.FindOverlapping, hay[:n], engine=mat_engine)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m = pma.scan(ScanMode.FindOverlapping, hay[:n], engine=mat_engine)
        dt = time.perf_counter() - t0
        out["materialize"] = {"bytes": n, "matches": int(len(m)), "seconds": round(dt, 4), "GB/s": round(n / dt / 1e9, 3),
                              "engine_used": ENGINE_NAMES.get(da.last_engine(), "?"),
                              "note": "daac_scan: the same + D2H of the 24-byte tuples into page-locked memory (PCIe-bound)"}
        del m

    # ---- CPU baseline: the C restatement of the reference CPU path, on a bounded prefix ----------------
    if not args.no_cpu and world == 1:
        os.environ["DAAC_ORACLE_NATIVE"] = "1"  # -O3 -march=native, built on this host
        from oracle import oracle as orc
        o = orc.OraclePma.deserialize(pma.serialize())
        usable, quota = cpu_limits()
        budget = max(4.0, args.cpu_seconds)
        probe = hay[:8 << 20].cpu().numpy()
        o.overlapping_count(probe[:1 << 20], threads=1)
        t0 = time.perf_counter()
        o.overlapping_count(probe, threads=1)
        rate1 = len(probe) / (time.perf_counter() - t0)
        ladder_t = [1]
        while ladder_t[-1] * 2 <= usable:
            ladder_t.append(ladder_t[-1] * 2)
        if ladder_t[-1] != usable:
            ladder_t.append(usable)
        per_rung = budget * 0.6 / len(ladder_t)
        cap = int(nbytes)
        sample = hay[:cap].cpu().numpy()
        ladder, best = [], None
        for t in ladder_t:
            n = int(min(cap, max(16 << 20, rate1 * min(t, 64) * per_rung)))
            n -= n % (1 << 20)
            t0 = time.perf_counter()
            o.overlapping_count(sample[:n], threads=t)
            r = n / (time.perf_counter() - t0)
            ladder.append({"threads": t, "GB/s": round(r / 1e9, 4), "MiB": n >> 20})
            if best is None or r > best[1]:
                best = (t, r)
        # the reported value: the best rung again on a sample sized for ~40 % of the budget, checked against the GPU
        n = int(min(cap, max(64 << 20, best[1] * budget * 0.4)))
        n -= n % (1 << 20)
        t0 = time.perf_counter()
        cN = o.overlapping_count(sample[:n], threads=best[0])
        dtN = time.perf_counter() - t0
        gpu_cc = pma.scan_count(ScanMode.FindOverlapping, hay[:n], engine=engine)     # count + checksum kernel
        gpu_c = pma.count(ScanMode.FindOverlapping, hay[:n], engine=engine)            # count-only kernel
        # ... and the WHOLE haystack of the timed step (not timed: the oracle on all usable threads; 4 GiB take a few seconds)
        whole = {"bytes": cap, "ok": None}
        if cap > n:
            t0 = time.perf_counter()
            cW = o.overlapping_count(sample[:cap], threads=best[0])
            whole["oracle_seconds"] = round(time.perf_counter() - t0, 2)
            whole["ok"] = bool(pma.scan_count(ScanMode.FindOverlapping, hay[:cap], engine=engine) == cW and
                               pma.count(ScanMode.FindOverlapping, hay[:cap], engine=engine) == cW[0] and
                               (args.haystack != "sparse" or args.op != "count" or total_count == cW[0]))
        else:
            whole["ok"] = bool(gpu_cc == cN and gpu_c == cN[0])
        out["cpu_baseline"] = {"value": round(n / dtN / 1e9, 4), "unit": "GB/s", "cores": best[0], "kind": "port",
                               "sample": f"first {n >> 20} MiB of the same haystack, {best[0]} threads with (Lmax-1)-byte halos, "
                                         f"gcc -O3 -march=native",
                               "single_thread_GB/s": round(rate1 / 1e9, 4),
                               "cores_usable": usable, "cores_os": os.cpu_count(), "cgroup_cpu_max": quota,
                               "effective_parallelism": round(best[1] / rate1, 1), "scaling": ladder,
                               "parity_with_gpu_on_sample": bool(gpu_cc == cN and gpu_c == cN[0]),
                               "parity_with_gpu_whole_haystack": whole["ok"], "parity_whole_haystack_bytes": whole["bytes"],
                               "parity_oracle_seconds": whole.get("oracle_seconds"),
                               "parity_checked": "whole haystack: count (count-only kernel, incl. the timed step's own result) and count + checksum "
                                                 "(checksum kernel) vs the oracle; the timed sample likewise"}
        del sample

    # ---- the other op beside the primary one: same haystack, few steps; the two kernels must agree on the count -----------
    if world == 1:
        other = "checksum" if args.op == "count" else "count"
        op["v"] = other
        _, k_s = timed(max(3, args.steps // 4), 1)
        oc = int(result[0].item())
        ocs = ((int(result[1].item()) & 0xFFFFFFFF) << 32) | (int(result[2].item()) & 0xFFFFFFFF)
        out["with_checksum" if other == "checksum" else "count_only"] = {
            "op": other, "value": round(nbytes / k_s / 1e9, 2), "unit": "GB/s", "frac": round(nbytes / k_s / 1e9 / HBM_PEAK_GBS, 4),
            "kernel_ms": round(k_s * 1e3, 4), "engine_used": ENGINE_NAMES.get(da.last_engine(), "?"),
            "match_count": oc, "match_checksum": f"{ocs:016x}" if other == "checksum" else None,
            "count_agrees_with_primary": bool(oc == total_count),
            "traffic": hbm_traffic(f"cfg3_{args.haystack}_{other}", "gram_count_kernel" if other == "checksum" else "gram4_kernel")[0] if args.workload == "cfg3" else None}
        if other == "checksum":
            out["value_count_checksum"] = out["with_checksum"]["value"]  # the op rounds 1 timed: compare THIS with BENCH_r01's `value`
        op["v"] = args.op

    # ---- the dense haystack (cfg3 (ii): word soup) beside the primary number --------------------------
    if world == 1 and not args.no_dense and args.haystack == "sparse":
        fill("dense")
        _, k_s = timed(max(3, args.steps // 4), 1)
        cnt = int(result[0].item())
        dense_parity = None
        if not args.no_cpu:   # the tail-record body of the count kernel runs here: the WHOLE word-soup haystack against the oracle (not timed)
            from oracle import oracle as orc
            od = orc.OraclePma.deserialize(pma.serialize())
            usable, _ = cpu_limits()
            cW = od.overlapping_count(hay[:int(nbytes)].cpu().numpy(), threads=usable)
            dense_parity = bool(cnt == cW[0] and pma.scan_count(ScanMode.FindOverlapping, hay[:int(nbytes)], engine=engine) == cW)
        dense_cs = None
        if args.op == "count":   # count + checksum of the same text (the checksum kernels have no tail records: profiles/r05_experiments_that_did_not_pay.txt)
            op["v"] = "checksum"
            _, kc_s = timed(3, 1)
            dense_cs = {"value": round(nbytes / kc_s / 1e9, 2), "unit": "GB/s", "kernel_ms": round(kc_s * 1e3, 4), "count_agrees": bool(int(result[0].item()) == cnt)}
            op["v"] = args.op
        out["dense"] = {"haystack": "dense (word soup)", "value": round(nbytes / k_s / 1e9, 2), "unit": "GB/s", "with_checksum": dense_cs,
                        "frac": round(nbytes / k_s / 1e9 / HBM_PEAK_GBS, 4), "kernel_ms": round(k_s * 1e3, 4),
                        "engine_used": ENGINE_NAMES.get(da.last_engine(), "?"), "matches_per_byte": round(cnt / nbytes, 4),
                        "parity_whole_haystack": dense_parity,
                        "traffic": hbm_traffic("cfg3_dense_count", "gram4_kernel")[0] if args.workload == "cfg3" and args.op == "count" else None}
    # ---- a dictionary beyond 31 byte classes: the cfg3 words in mixed case + digits (60 pattern bytes) -----------------
    if world == 1 and not args.no_dense and args.workload == "cfg3":
        del hay
        torch.cuda.empty_cache()
        wp = da.DoubleArrayAhoCorasick.new(synth.patterns_cfg3_wide())
        wp.upload(local_rank)
        wn = 1 << 30
        whay = torch.empty(wn, dtype=torch.uint8, device="cuda")
        synth.device_uniform(whay, seed_sparse, synth.ALPHA_WIDE_SPACE)
        wide = {"dictionary": "cfg3 words in lower / Capitalised / UPPER case, a quarter with a digit: 60 pattern bytes", "bytes": wn,
                "byte_classes": None}
        for name, fn in (("count", lambda: wp.count(ScanMode.FindOverlapping, whay, stream=stream, result_dev=result.data_ptr())),
                         ("checksum", lambda: wp.scan_count(ScanMode.FindOverlapping, whay, stream=stream, result_dev=result.data_ptr()))):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            wide[name] = {"value": round(wn / ms / 1e6, 2), "unit": "GB/s", "frac": round(wn / ms / 1e6 / HBM_PEAK_GBS, 4),
                          "engine_used": ENGINE_NAMES.get(da.last_engine(), "?"), "match_count": int(result[0].item())}
        wide["byte_classes"] = wp.info().num_classes
        out["wide_alphabet"] = wide
        del whay, wp
        torch.cuda.empty_cache()
        # ---- dictionaries no byte-class table serves: `.count()` on the PFX engine (any byte alphabet) ------------------
        anyab = {}
        for name in ("binary256", "utf8jp", "unidic_like", "o200k_like"):
            pats_w = {"binary256": synth.patterns_binary256, "utf8jp": synth.patterns_cfg5, "unidic_like": synth.patterns_unidic_like,
                      "o200k_like": synth.patterns_o200k_like}[name]()
            ap = da.DoubleArrayAhoCorasick.new(pats_w)
            ap.upload(local_rank)
            an = 1 << 30
            if name in ("utf8jp", "unidic_like"):
                an -= an % synth.CFG5_SLOT
            ahay = torch.empty(an, dtype=torch.uint8, device="cuda")
            if name == "binary256":
                synth.device_uniform(ahay, synth.SEEDS["bin_hay"], synth.ALPHA_BYTES)
            elif name == "o200k_like":
                synth.device_wordsoup(ahay, synth.SEEDS["o200k_hay"], synth.o200k_soup_words(), 17)
            else:
                synth.device_zipf_text(ahay)
            fn = lambda: ap.count(ScanMode.FindOverlapping, ahay, stream=stream, result_dev=result.data_ptr())
            ap.count(ScanMode.FindOverlapping, ahay)   # a synchronous call samples the text and leaves the engine choice (PFX / walker) in the handle
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            gpu_cnt = int(result[0].item())
            used = ENGINE_NAMES.get(da.last_engine(), "?")
            # the same with the checksum of every (start, end, value)
            fnx = lambda: ap.scan_count(ScanMode.FindOverlapping, ahay, stream=stream, result_dev=result.data_ptr())
            fnx()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fnx()
            e1.record()
            torch.cuda.synchronize()
            msx = e0.elapsed_time(e1) / 5
            used_x = ENGINE_NAMES.get(da.last_engine(), "?")
            # the tuples (16-byte device format) of the first 256 MiB: wall time of daac_scan_device16
            tn = (256 << 20) - ((256 << 20) % synth.CFG5_SLOT)
            da.set_option("max_result_bytes", 64 << 30)
            dm = ap.scan_device(ScanMode.FindOverlapping, ahay[:tn], fmt16=True)
            dm.free()
            torch.cuda.synchronize()
            t_best = None
            for _ in range(3):
                t0 = time.perf_counter()
                dm = ap.scan_device(ScanMode.FindOverlapping, ahay[:tn], fmt16=True)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                n_tuples = dm.count
                dm.free()
                t_best = dt if t_best is None else min(t_best, dt)
            used_t = ENGINE_NAMES.get(da.last_engine(), "?")
            # parity on a 64 MiB prefix against the oracle (the whole GiB would take the CPU a minute); tuples on 8 MiB
            pn = (64 << 20) - ((64 << 20) % synth.CFG5_SLOT if name in ("utf8jp", "unidic_like") else 0)
            ok = ok_t = None
            if not args.no_cpu:
                from oracle import oracle as orc2
                oo = orc2.OraclePma.deserialize(ap.serialize())
                want_a = oo.overlapping_count(ahay[:pn].cpu().numpy(), threads=16)
                ok = bool(ap.count(ScanMode.FindOverlapping, ahay[:pn]) == want_a[0] and ap.scan_count(ScanMode.FindOverlapping, ahay[:pn]) == want_a)
                qn = 48 * 174762
                want_t = oo.find_overlapping_iter(ahay[:qn].cpu().numpy())
                got_t = ap.scan(ScanMode.FindOverlapping, ahay[:qn])
                ok_t = bool(len(got_t) == len(want_t) and np.array_equal(got_t["start"], want_t["start"]) and np.array_equal(got_t["end"], want_t["end"]) and
                            np.array_equal(got_t["value"], want_t["value"]))
            anyab[name] = {"dictionary": {"binary256": "100 000 random patterns of 3-12 bytes over all 256 byte values; haystack: uniform random bytes",
                                          "utf8jp": "cfg5's 50 000 UTF-8 patterns (2-8 three-byte scalars, Zipf) scanned BYTEWISE; haystack: cfg5's Zipf text",
                                          "unidic_like": "look-alike of the crate's Unidic benchmark dictionary: 675 000 UTF-8 patterns of 1-8 three-byte scalars (Zipf); haystack: cfg5's Zipf text",
                                          "o200k_like": "look-alike of o200k_base: 200 000 byte-level tokens incl. all 256 one-byte patterns; haystack: soup of its base words"}[name],
                            "patterns": len(pats_w), "automaton_bytes": ap.heap_bytes(),
                            "bytes": an, "value": round(an / ms / 1e6, 2), "unit": "GB/s", "frac": round(an / ms / 1e6 / HBM_PEAK_GBS, 4),
                            "kernel_ms": round(ms, 4), "engine_used": used, "match_count": gpu_cnt, "matches_per_byte": round(gpu_cnt / an, 4),
                            "with_checksum": {"value": round(an / msx / 1e6, 2), "unit": "GB/s", "kernel_ms": round(msx, 4), "engine_used": used_x},
                            "tuples_device": {"bytes": tn, "matches": int(n_tuples), "GB/s": round(tn / t_best / 1e9, 2), "tuple_GB/s": round(n_tuples * 16 / t_best / 1e9, 1),
                                              "seconds": round(t_best, 5), "engine_used": used_t, "parity_8mib_prefix_vs_oracle": ok_t},
                            "parity_64mib_prefix_vs_oracle": ok}
            del ahay, ap
            torch.cuda.empty_cache()
        out["any_alphabet"] = anyab
    # ---- the other iterators of the path and configs[4], under this same clock; the lazy iterator over host haystacks ----
    if world == 1 and not args.no_extra and args.workload == "cfg3" and args.haystack == "sparse":
        try:
            hay_alive = hay  # noqa: F841
            del hay
        except NameError:
            pass
        torch.cuda.empty_cache()
        out["restart"] = restart_legs(da, synth, torch, patterns, local_rank, stream, result, args.no_cpu, seed_sparse, seed_dense, alpha)
        out["cfg5"] = cfg5_legs(da, synth, torch, local_rank, stream, result, args.no_cpu)
        out["iterator"] = iterator_legs(da, synth, torch, np, pma, local_rank, seed_sparse, alpha)
        out["stream"] = stream_legs(da, synth, torch, np, pma, local_rank, seed_sparse, alpha, args.no_cpu)
        out["multi"] = multi_legs(da, synth, torch, pma, local_rank, alpha)
    print(json.dumps(out))
    # a multi-GPU line that is not what it says fails loudly: the reason is in the line (`distributed`), the exit code says so
    if dist_used is not None and (dist_used.get("n_ranks_seen") != args.gpus or dist_used.get("strong_equals_one_rank") is False):
        sys.stdout.flush()
        raise SystemExit(f"bench.py: --gpus {args.gpus}: ranks seen {dist_used.get('n_ranks_seen')}, strong_equals_one_rank "
                         f"{dist_used.get('strong_equals_one_rank')} (see `distributed` in the line above)")


if __name__ == "__main__":
    main()

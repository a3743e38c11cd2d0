"""The N > 1 path on CPUs: two processes over gloo run the same sharding + reduction code that
bench.py runs over RCCL; the per-shard scanner is the CPU oracle here (the GPU is not involved)."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from daachorse_amd import dist as ddist
    from daachorse_amd import synth
    from oracle import oracle as orc

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pats = synth.patterns_cfg3(3000)
        o = orc.OraclePma.build(pats)
        halo = o.max_pattern_len() - 1
        n = 3_000_017  # not a multiple of anything
        hay = synth.wordsoup_haystack(n, synth.SEEDS["cfg3_dense"], pats, 20)

        # (1) one haystack split over the ranks: every rank counts the matches ending in its range
        def scan(lo, hi):
            start = max(0, lo - halo)
            m = o.find_overlapping_iter(hay[start:hi])
            m = m[m["end"] + start > lo] if lo > 0 else m
            m = m.copy()
            m["start"] += start
            m["end"] += start
            return len(m), orc.matches_checksum(m)

        got = ddist.sharded_count(scan, n, rank, world, align=4096)
        want = o.overlapping_count(hay, threads=1)
        assert got == want, (rank, got, want)

        # (2) independent shards (the bench's weak-scaling mode): totals add up
        mine = synth.uniform_haystack(200_000, synth.SEEDS["cfg4_hay"] + rank, synth.ALPHA_LOWER_SPACE)
        c, cs = o.overlapping_count(mine)
        s1, s2 = ddist.split_checksum(cs)
        tot = ddist.all_reduce_counts(c, s1, s2)
        ref_c, ref_1, ref_2 = 0, 0, 0
        for r in range(world):
            cc, ccs = o.overlapping_count(synth.uniform_haystack(200_000, synth.SEEDS["cfg4_hay"] + r, synth.ALPHA_LOWER_SPACE))
            a, b = ddist.split_checksum(ccs)
            ref_c, ref_1, ref_2 = ref_c + cc, ref_1 + a, ref_2 + b
        assert tot == (ref_c, ddist.join_checksum(ref_1, ref_2)), rank

        # (3) the timing reduction takes the slowest rank
        assert ddist.max_over_ranks(1.0 + rank) == float(world)
        with open(os.path.join(tmpdir, f"ok{rank}"), "w") as f:
            f.write("ok")
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_count(tmp_path):
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def test_shard_ranges_cover_exactly():
    from daachorse_amd.dist import shard_range
    for total in (0, 1, 4095, 4096, 1 << 20, (1 << 30) + 17):
        for world in (1, 2, 3, 8):
            edges = [shard_range(total, r, world, align=4096) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))


def _run_bench(extra, env_extra=None, timeout=600):
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout  # rank 0 prints the one line
    return json.loads(lines[0])


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks (torchrun on 127.0.0.1), reduces over
    them and prints n_gpus = 2.  --plumbing: no device and no scan, only launch + rendezvous + reduction + the line."""
    line = _run_bench(["--gpus", "2", "--plumbing", "--steps", "2", "--warmup", "1", "--scaling", "strong"])
    assert line["n_gpus"] == 2 and line["plumbing_only"] is True and line["value"] is None and line["scaling"] == "strong"
    assert line["reduced"] == [3, ((30 << 32) | 300)]  # sums over ranks 0 and 1 of {r + 1, 10 (r + 1), 100 (r + 1)}


def test_bench_refuses_more_ranks_than_gpus():
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "DAAC_BENCH_OVERSUBSCRIBE")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1"], env=env, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode != 0 and "hipGetDeviceCount" in (out.stderr + out.stdout)


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu_weak_and_strong():
    """the real step under the self-launcher: two ranks share the box's GPU (oversubscribed, gloo for the reduce).
    Strong scaling must reproduce the single-rank count + checksum of the same haystack exactly."""
    common = ["--steps", "2", "--warmup", "1", "--no-cpu", "--no-dense", "--materialize-mib", "0", "--workload", "cfg2", "--op", "checksum"]
    env = {"DAAC_DIST_BACKEND": "gloo", "DAAC_BENCH_OVERSUBSCRIBE": "1"}
    one = _run_bench(["--gpus", "1", "--bytes", str(96 << 20)] + common)
    strong = _run_bench(["--gpus", "2", "--scaling", "strong", "--bytes", str(96 << 20)] + common, env)
    assert strong["n_gpus"] == 2 and strong["scaling"] == "strong"
    assert strong["match_count"] == one["match_count"] and strong["match_checksum"] == one["match_checksum"]
    assert strong["config"]["haystack_bytes_job"] == 96 << 20 and strong["config"]["engine_used"] == "gram"
    d = strong["distributed"]   # the self-diagnosis of a multi-rank run: per-rank kernel times, the reduce, the strong-scaling self-check
    assert d["n_ranks_seen"] == 2 and len(d["per_rank_kernel_ms"]["all"]) == 2 and d["per_rank_kernel_ms"]["max"] >= d["per_rank_kernel_ms"]["min"] > 0
    assert d["reduce_ms"]["max"] >= 0 and sum(d["per_rank_bytes"]) == 96 << 20 and d["strong_equals_one_rank"] is True
    weak = _run_bench(["--gpus", "2", "--bytes", str(32 << 20)] + common, env)
    assert weak["n_gpus"] == 2 and weak["scaling"] == "weak" and weak["config"]["haystack_bytes_job"] == 64 << 20
    assert weak["value"] > 0 and weak["match_count"] > 0
    # the default op (.count() alone) under strong scaling: same count, no checksum in the line
    cnt = _run_bench(["--gpus", "2", "--scaling", "strong", "--bytes", str(96 << 20)] + common[:-2], env)
    assert cnt["match_count"] == one["match_count"] and cnt["match_checksum"] is None and cnt["op"].endswith(".count()")


@pytest.mark.gpu
def test_bench_takes_the_rccl_branch_on_one_gpu():
    """RCCL on hardware: bench.py's distributed branch with the `nccl` backend at world size 1 — init_process_group("nccl"),
    a device-tensor all-reduce per step, barriers, teardown — so that the first time that code runs is not on the 8-GPU node.
    The line must equal the plain single-process one."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    common = ["--steps", "3", "--warmup", "1", "--no-cpu", "--no-dense", "--materialize-mib", "0", "--workload", "cfg2", "--op", "checksum",
              "--bytes", str(64 << 20)]
    plain = _run_bench(["--gpus", "1"] + common)
    env = {"DAAC_BENCH_FORCE_DIST": "1", "RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
           "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    rccl = _run_bench(["--gpus", "1"] + common, env)
    d = rccl["distributed"]
    assert d["backend"] == "nccl" and d["world_size"] == 1 and d["all_reduces"] == 4 and d["n_ranks_seen"] == 1
    assert d["per_rank_kernel_ms"]["min"] > 0 and d["reduce_ms"]["max"] >= 0 and d["rccl_version"]
    assert plain["distributed"] is None
    assert rccl["match_count"] == plain["match_count"] and rccl["match_checksum"] == plain["match_checksum"] and rccl["value"] > 0

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

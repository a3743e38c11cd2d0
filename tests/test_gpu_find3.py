"""find_iter without a state chain (find3_kernels.hip: selection over the tuple emitter's per-position flags, relaxed bit-parallel per tile)
against the oracle's FindIterator (reference src/bytewise/iter.rs:58-113): count + checksum and `.count()`, dictionaries with and without
one-byte patterns, deep-only positions, restarts inside the haystack, texts that keep the relaxation busy (given up: the chain walkers answer)."""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

import daachorse_amd as da
from daachorse_amd import Engine, ScanMode, synth


def _pma(patterns):
    o = orc.OraclePma.build(patterns)
    p, rest = da.DoubleArrayAhoCorasick.deserialize(o.serialize())
    assert rest == b""
    return o, p


def _want(o, hay):
    m = o.find_iter(hay)
    return len(m), orc.matches_checksum(m)


def test_find3_against_the_oracle():
    import torch
    rng = np.random.default_rng(2031)
    pats3 = synth.patterns_cfg3(30000)
    with1 = synth.patterns_cfg3(5000) + [b"a", b"e", b"q"]
    deepish = [b"abcd", b"bcdefg", b"cdefghijklmnopqrs", b"defg", b"ghij", b"xy", b"yz", b"zab", b"nopqrstuvwxyzabcdef"]   # mostly deep patterns
    cases = [(pats3, synth.uniform_haystack((3 << 20) + 7, synth.SEEDS["cfg3_hay"], synth.ALPHA_LOWER_SPACE)),
             (pats3, synth.wordsoup_haystack(2 << 20, synth.SEEDS["cfg3_dense"], pats3, 20)),
             (synth.patterns_cfg3(), synth.uniform_haystack(4 << 20, 77, synth.ALPHA_LOWER_SPACE)),
             (with1, synth.uniform_haystack(1 << 20, 5, synth.ALPHA_LOWER_SPACE)),
             (with1, synth.wordsoup_haystack(1 << 20, 6, with1, 20)),
             (deepish, np.frombuffer((b"abcdefghijklmnopqrstuvwxyz" * 40000)[:1000003], dtype=np.uint8)),
             (deepish, synth.uniform_haystack(1 << 20, 9, b"abcdefghijklmnopqrstuvwxyz")),
             (synth.patterns_cfg2(500), synth.wordsoup_haystack(600000, 8, synth.patterns_cfg2(500), 13, noise_256=30)),
             (synth.patterns_cfg1(), synth.uniform_haystack(70000, 3, synth.ALPHA_ABCD))]
    served = 0
    for pats, hay in cases:
        o, p = _pma(pats)
        want = _want(o, hay)
        for shift in (0, 5):
            dev = torch.from_numpy(np.concatenate([np.zeros(shift, dtype=np.uint8), hay])).cuda()[shift:]
            got = p.scan_count(ScanMode.Find, dev)
            served += da.last_engine() == int(Engine.Gram)
            assert got == want, (len(pats), len(hay), shift, da.last_engine())
            assert p.count(ScanMode.Find, dev) == want[0]
        # a restart inside the haystack: the chain begins at `b` (a sync point of the caller's), ends stay absolute
        b = int(rng.integers(1, len(hay) - 1))
        p.set_option("find3", 0)   # (this handle's: daac_pma_set_option)
        ref = p.scan_count(ScanMode.Find, dev, begin=b)
        p.set_option("find3")
        assert p.scan_count(ScanMode.Find, dev, begin=b) == ref, (len(pats), b)
        # host haystack
        assert p.scan_count(ScanMode.Find, hay) == want
    assert served >= 12, served   # (the small dictionaries may have no K = 3 tables: the walkers answer, equally)


def test_find3_gives_up_where_the_relaxation_will_not_settle():
    """`aaaa...` against "aa": every position looks back at the one before for the tile's whole length — the kernel flags the tile and the chain
    walkers (or their fallback) answer; nothing but the engine changes."""
    o, p = _pma([b"aa", b"aaa", b"b", b"ab"] + synth.patterns_cfg3(3000))
    hay = np.frombuffer(b"a" * 300000 + b"b" + b"a" * 100001, dtype=np.uint8)
    want = _want(o, hay)
    assert p.scan_count(ScanMode.Find, hay) == want
    assert da.last_engine() != int(Engine.Gram)
    # and text on which it does settle, from the same handle
    hay2 = synth.uniform_haystack(1 << 20, 11, synth.ALPHA_LOWER_SPACE)
    assert p.scan_count(ScanMode.Find, hay2) == _want(o, hay2)


def test_find3_one_gib_of_cfg3():
    """BASELINE's dictionary over 1 GiB (the bench's `restart` leg): count + checksum against the chain walkers, and a 64 MiB prefix against the oracle"""
    import torch
    pats = synth.patterns_cfg3()
    p = da.DoubleArrayAhoCorasick.new(pats)
    o = orc.OraclePma.deserialize(p.serialize())
    dev = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
    for kind in ("sparse", "dense"):
        if kind == "sparse":
            synth.device_uniform(dev, synth.SEEDS["cfg3_hay"], synth.ALPHA_LOWER_SPACE)
        else:
            synth.device_wordsoup(dev, synth.SEEDS["cfg3_dense"], pats, 20)
        got = p.scan_count(ScanMode.Find, dev)
        # (text made of the dictionary's own words keeps the detection's walkers busy: the engine says so after its first look and the
        # chain walkers serve that handle's requests from then on; find3 = 2 insists)
        assert da.last_engine() == (int(Engine.Gram) if kind == "sparse" else int(Engine.DArray)), kind
        if kind == "dense":
            p.set_option("find3", 2)
            assert p.scan_count(ScanMode.Find, dev) == got and da.last_engine() == int(Engine.Gram)
            assert p.count(ScanMode.Find, dev) == got[0]
        p.set_option("find3", 0)
        ref = p.scan_count(ScanMode.Find, dev)
        assert da.last_engine() == int(Engine.DArray)
        p.set_option("find3")
        assert got == ref, kind
        pre = dev[:64 << 20]
        assert p.scan_count(ScanMode.Find, pre) == _want(o, pre.cpu().numpy()), kind


def test_find3_windows_restart_where_the_last_match_ended():
    """A haystack beyond one window is scanned window by window, each restarting at the end of the last match the one before selected (or,
    none near its end, 64 bytes before it).  Windows of 8 KiB .. 1 MiB over texts whose matches straddle every boundary, against the
    oracle; then 2.5 GiB of cfg3 in the real windows of 1 GiB against the chain walkers."""
    import torch
    pats3 = synth.patterns_cfg3(30000)
    with1 = synth.patterns_cfg3(5000) + [b"a", b"e", b"q"]
    deepish = [b"abcd", b"bcdefg", b"cdefghijklmnopqrs", b"defg", b"ghij", b"xy", b"yz", b"zab", b"nopqrstuvwxyzabcdef"]
    cases = [(pats3, synth.uniform_haystack((3 << 20) + 7, synth.SEEDS["cfg3_hay"], synth.ALPHA_LOWER_SPACE)),
             (pats3, synth.wordsoup_haystack(2 << 20, synth.SEEDS["cfg3_dense"], pats3, 20)),
             (with1, synth.wordsoup_haystack(1 << 20, 6, with1, 20)),
             (deepish, np.frombuffer((b"abcdefghijklmnopqrstuvwxyz" * 40000)[:1000003], dtype=np.uint8)),
             (deepish, np.frombuffer((b"abcdefghijklmnopqrstuvwxyz" + b"-" * 4000) * 200, dtype=np.uint8)),   # stretches without any match: the 64-byte rule
             (deepish, synth.uniform_haystack(1 << 20, 9, b"abcdefghijklmnopqrstuvwxyz"))]
    for pats, hay in cases:
        o, p = _pma(pats)
        p.set_option("find3", 2)
        want = _want(o, hay)
        dev = torch.from_numpy(hay.copy()).cuda()
        for win in (8192, 8192 + 4096 + 17, 65536, 1 << 20):
            p.set_option("find3_window", win)
            assert p.scan_count(ScanMode.Find, dev) == want, (len(pats), len(hay), win)
            assert da.last_engine() == int(Engine.Gram)
            assert p.count(ScanMode.Find, dev) == want[0]
        b = len(hay) // 3
        p.set_option("find3", 0)
        ref = p.scan_count(ScanMode.Find, dev, begin=b)
        p.set_option("find3", 2).set_option("find3_window", 100000)
        assert p.scan_count(ScanMode.Find, dev, begin=b) == ref
    pats = synth.patterns_cfg3()
    p = da.DoubleArrayAhoCorasick.new(pats)
    dev = torch.empty((5 << 29) + 4321, dtype=torch.uint8, device="cuda")
    synth.device_uniform(dev, synth.SEEDS["cfg3_hay"], synth.ALPHA_LOWER_SPACE)
    got = p.scan_count(ScanMode.Find, dev)
    assert da.last_engine() == int(Engine.Gram)
    p.set_option("find3", 0)
    assert p.scan_count(ScanMode.Find, dev) == got

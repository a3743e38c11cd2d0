"""leftmost_find_iter without a state chain (left3_kernels.hip: the emitter's detection run on a Standard automaton of the handle's own
patterns, selection by STARTS relaxed bit-parallel per tile) against the oracle's leftmost iterator (reference src/bytewise/iter.rs:272-340):
LeftmostLongest and LeftmostFirst, dictionaries with and without one-byte patterns, deep matches that cover tile and lane borders,
restarts inside the haystack, texts the relaxation gives up on (the chain walkers answer)."""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

import daachorse_amd as da
from daachorse_amd import Engine, ScanMode, synth


def _pma(patterns, kind):
    o = orc.OraclePma.build(patterns, kind=kind)
    p, rest = da.DoubleArrayAhoCorasick.deserialize(o.serialize())
    assert rest == b""
    return o, p


def _want(o, hay):
    m = o.leftmost_find_iter(hay)
    return len(m), orc.matches_checksum(m)


def test_left3_against_the_oracle():
    import torch
    rng = np.random.default_rng(2032)
    pats3 = synth.patterns_cfg3(30000)
    with1 = synth.patterns_cfg3(5000) + [b"a", b"e", b"q"]
    deepish = [b"abcd", b"bcdefg", b"cdefghijklmnopqrs", b"defg", b"ghij", b"xy", b"yz", b"zab", b"nopqrstuvwxyzabcdef", b"ab", b"abc"]
    # LeftmostFirst: later patterns below earlier ones are dropped by the builder; later SHORTER ones stay
    order = [b"abcde", b"abc", b"ab", b"bcd", b"b", b"cdefgh", b"cd", b"efghijklmnop", b"e", b"zzzz", b"zz"]
    cases = [(pats3, synth.uniform_haystack((3 << 20) + 7, synth.SEEDS["cfg3_hay"], synth.ALPHA_LOWER_SPACE)),
             (pats3, synth.wordsoup_haystack(2 << 20, synth.SEEDS["cfg3_dense"], pats3, 20)),
             (synth.patterns_cfg3(), synth.uniform_haystack(4 << 20, 77, synth.ALPHA_LOWER_SPACE)),
             (with1, synth.uniform_haystack(1 << 20, 5, synth.ALPHA_LOWER_SPACE)),
             (with1, synth.wordsoup_haystack(1 << 20, 6, with1, 20)),
             (deepish, np.frombuffer((b"abcdefghijklmnopqrstuvwxyz" * 40000)[:1000003], dtype=np.uint8)),
             (deepish, synth.uniform_haystack(1 << 20, 9, b"abcdefghijklmnopqrstuvwxyz")),
             (order, synth.uniform_haystack(1 << 20, 10, b"abcdefghz")),
             (order, np.frombuffer((b"abcdefghijklmnopqrstuvwxyz" * 40000)[:1000003], dtype=np.uint8)),
             (synth.patterns_cfg2(500), synth.wordsoup_haystack(600000, 8, synth.patterns_cfg2(500), 13, noise_256=30))]
    served = 0
    for kind_o, kind_name in ((orc.LEFTMOST_LONGEST, "longest"), (orc.LEFTMOST_FIRST, "first")):
        for pats, hay in cases:
            o, p = _pma(pats, kind_o)
            want = _want(o, hay)
            for shift in (0, 5):
                dev = torch.from_numpy(np.concatenate([np.zeros(shift, dtype=np.uint8), hay])).cuda()[shift:]
                got = p.scan_count(ScanMode.LeftmostFind, dev)
                served += da.last_engine() == int(Engine.Gram)
                assert got == want, (kind_name, len(pats), len(hay), shift, da.last_engine())
                assert p.count(ScanMode.LeftmostFind, dev) == want[0]
            b = int(rng.integers(1, len(hay) - 1))
            p.set_option("left3", 0)   # (this handle's: daac_pma_set_option)
            ref = p.scan_count(ScanMode.LeftmostFind, dev, begin=b)
            assert da.last_engine() != int(Engine.Gram)
            p.set_option("left3")
            assert p.scan_count(ScanMode.LeftmostFind, dev, begin=b) == ref, (kind_name, len(pats), b)
            assert p.scan_count(ScanMode.LeftmostFind, hay) == want
    assert served >= 24, served   # (the small dictionaries may have no K = 3 tables: the walkers answer, equally)


def test_left3_gives_up_where_the_relaxation_will_not_settle():
    o, p = _pma([b"aa", b"aaa", b"b", b"ab"] + synth.patterns_cfg3(3000), orc.LEFTMOST_LONGEST)
    hay = np.frombuffer(b"a" * 300000 + b"b" + b"a" * 100001, dtype=np.uint8)
    assert p.scan_count(ScanMode.LeftmostFind, hay) == _want(o, hay)
    hay2 = synth.uniform_haystack(1 << 20, 11, synth.ALPHA_LOWER_SPACE)
    assert p.scan_count(ScanMode.LeftmostFind, hay2) == _want(o, hay2)


def test_left3_one_gib_of_cfg3():
    """BASELINE's dictionary (LeftmostLongest) over 1 GiB (the bench's `restart` leg): count + checksum against the chain walkers, and a 64 MiB
    prefix against the oracle"""
    import torch
    pats = synth.patterns_cfg3()
    p = da.DoubleArrayAhoCorasickBuilder().match_kind(da.MatchKind.LeftmostLongest).build(pats)
    o = orc.OraclePma.deserialize(p.serialize())
    dev = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
    for kind in ("sparse", "dense"):
        if kind == "sparse":
            synth.device_uniform(dev, synth.SEEDS["cfg3_hay"], synth.ALPHA_LOWER_SPACE)
        else:
            synth.device_wordsoup(dev, synth.SEEDS["cfg3_dense"], pats, 20)
        got = p.scan_count(ScanMode.LeftmostFind, dev)
        assert da.last_engine() == (int(Engine.Gram) if kind == "sparse" else int(Engine.DArray)), kind
        if kind == "dense":
            p.set_option("left3", 2)
            assert p.scan_count(ScanMode.LeftmostFind, dev) == got and da.last_engine() == int(Engine.Gram)
        p.set_option("left3", 0)
        ref = p.scan_count(ScanMode.LeftmostFind, dev)
        assert da.last_engine() == int(Engine.DArray)
        p.set_option("left3")
        assert got == ref, kind
        pre = dev[:64 << 20]
        assert p.scan_count(ScanMode.LeftmostFind, pre) == _want(o, pre.cpu().numpy()), kind


def test_left3_windows_restart_where_the_last_match_ended():
    """A haystack beyond one window: a window's matches START in it (the detection looks 32 bytes further), the next one restarts at the end of
    its last match.  Windows of 8 KiB .. 1 MiB against the oracle, both kinds; then 2.5 GiB of cfg3 in the real windows against the walkers."""
    import torch
    pats3 = synth.patterns_cfg3(30000)
    with1 = synth.patterns_cfg3(5000) + [b"a", b"e", b"q"]
    deepish = [b"abcd", b"bcdefg", b"cdefghijklmnopqrs", b"defg", b"ghij", b"xy", b"yz", b"zab", b"nopqrstuvwxyzabcdef", b"ab", b"abc"]
    cases = [(pats3, synth.uniform_haystack((3 << 20) + 7, synth.SEEDS["cfg3_hay"], synth.ALPHA_LOWER_SPACE)),
             (pats3, synth.wordsoup_haystack(2 << 20, synth.SEEDS["cfg3_dense"], pats3, 20)),
             (with1, synth.wordsoup_haystack(1 << 20, 6, with1, 20)),
             (deepish, np.frombuffer((b"abcdefghijklmnopqrstuvwxyz" * 40000)[:1000003], dtype=np.uint8)),
             (deepish, np.frombuffer((b"abcdefghijklmnopqrstuvwxyz" + b"-" * 4000) * 200, dtype=np.uint8)),
             (deepish, synth.uniform_haystack(1 << 20, 9, b"abcdefghijklmnopqrstuvwxyz"))]
    for kind_o in (orc.LEFTMOST_LONGEST, orc.LEFTMOST_FIRST):
        for pats, hay in cases:
            o, p = _pma(pats, kind_o)
            p.set_option("left3", 2)
            want = _want(o, hay)
            dev = torch.from_numpy(hay.copy()).cuda()
            for win in (8192, 8192 + 4096 + 17, 65536, 1 << 20):
                p.set_option("find3_window", win)
                assert p.scan_count(ScanMode.LeftmostFind, dev) == want, (kind_o, len(pats), len(hay), win)
                # (the alphabet repeated: more than 255 deep matches selected in a tile — given up, the walkers answer)
                assert da.last_engine() == int(Engine.Gram) or (pats is deepish and hay[0] == ord("a") and hay[26] == ord("a"))
                assert p.count(ScanMode.LeftmostFind, dev) == want[0]
    pats = synth.patterns_cfg3()
    p = da.DoubleArrayAhoCorasickBuilder().match_kind(da.MatchKind.LeftmostLongest).build(pats)
    dev = torch.empty((5 << 29) + 4321, dtype=torch.uint8, device="cuda")
    synth.device_uniform(dev, synth.SEEDS["cfg3_hay"], synth.ALPHA_LOWER_SPACE)
    got = p.scan_count(ScanMode.LeftmostFind, dev)
    assert da.last_engine() == int(Engine.Gram)
    p.set_option("left3", 0)
    assert p.scan_count(ScanMode.LeftmostFind, dev) == got

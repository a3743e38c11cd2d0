"""Parity of the HIP scan path (through the C ABI) with the CPU oracle — runs on the MI355X.

Bit-exact comparison of (start, end, value) tuples in the reference's order; count + checksum on
larger inputs.  Both device engines (TIERED and DARRAY) are exercised.
"""
import itertools

import numpy as np
import pytest

from conftest import iter_vector_runs
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

import daachorse_amd as da
from daachorse_amd import Engine, ScanMode, synth

API_MODE = {"find_overlapping_iter": ScanMode.FindOverlapping,
            "find_overlapping_no_suffix_iter": ScanMode.FindOverlappingNoSuffix,
            "find_iter": ScanMode.Find, "leftmost_find_iter": ScanMode.LeftmostFind}
KIND_CODE = {"Standard": 0, "LeftmostLongest": 1, "LeftmostFirst": 2}
ENGINES = [Engine.Tiered, Engine.DArray]


def _pma(patterns, values=None, kind="Standard"):
    o = orc.OraclePma.build(patterns, values=values, kind=kind)
    p, rest = da.DoubleArrayAhoCorasick.deserialize(o.serialize())
    assert rest == b""
    return o, p


def _sev(m):
    return [(int(x["start"]), int(x["end"]), int(x["value"])) for x in m]


def _check_counts(p, mode, hay, want):
    """count + checksum on every engine that can serve the request (GRAM declines some automata)"""
    expect = (len(want), orc.matches_checksum(want))
    for eng in ENGINES + [Engine.Auto]:
        assert p.scan_count(mode, hay, engine=eng) == expect, eng
        assert p.count(mode, hay, engine=eng) == expect[0], eng  # `.count()` alone
    if mode == ScanMode.FindOverlapping:
        served = False
        for version in (1, 2, 0):  # first table set, second, the default choice
            p.set_option("gram_version", version)   # (the handle's own setting: daac_pma_set_option)
            try:
                got = p.scan_count(mode, hay, engine=Engine.Gram)
                assert got == expect, ("gram", version)
                served = True
            except da.DaachorseError as e:
                assert e.code == 6
            try:
                assert p.count(mode, hay, engine=Engine.Gram) == expect[0], ("gram count", version)
            except da.DaachorseError as e:
                assert e.code == 6
        p.set_option("gram_version")
        return served
    return False


def _same(a, b):
    return len(a) == len(b) and np.array_equal(a["start"], b["start"]) and np.array_equal(a["end"], b["end"]) and \
        np.array_equal(a["value"], b["value"])


# (launch shapes, table budgets and engine switches are set on the HANDLE under test — daac_pma_set_option, `p.set_option(...)` — so no test
# leaves anything process-wide behind and there is no reset fixture; the process-wide form has test_options below)


def test_golden_vectors_overlapping(vectors):
    """tests/aho_corasick_crate_test.rs search_standard_overlapping (BASICS + OVERLAPPING) on the GPU."""
    n = 0
    for runner, case in iter_vector_runs(vectors):
        if runner["api"] != "find_overlapping_iter":
            continue
        o, p = _pma(case["patterns"])
        want = [tuple(t) for t in case["matches"]]
        for eng in ENGINES:
            got = p.scan(ScanMode.FindOverlapping, case["haystack"], engine=eng)
            assert [(int(m["value"]), int(m["start"]), int(m["end"])) for m in got] == want, (case["name"], eng)
        _check_counts(p, ScanMode.FindOverlapping, case["haystack"], o.find_overlapping_iter(case["haystack"]))
        lazy = [(m.value(), m.start(), m.end()) for m in p.find_overlapping_iter(case["haystack"])]
        assert lazy == want, case["name"]
        n += 1
    assert n == 57


def _empty_pattern_leftmost(runner, case):
    return runner["api"] == "leftmost_find_iter" and "" in case["patterns"]


def test_golden_vectors_find_and_leftmost(vectors):
    """search_standard_non_overlapping, search_leftmost_longest, search_leftmost_first
    (tests/aho_corasick_crate_test.rs:537-589) on the GPU: eager scan, count + checksum, lazy iterator."""
    n = with_empty = 0
    for runner, case in iter_vector_runs(vectors):
        if runner["api"] not in ("find_iter", "leftmost_find_iter"):
            continue
        o, p = _pma(case["patterns"], kind=runner["kind"])
        mode = API_MODE[runner["api"]]
        if _empty_pattern_leftmost(runner, case):
            with_empty += 1  # "" under leftmost kinds (iter.rs:254-261): covered by the reference's vectors
        want = [tuple(t) for t in case["matches"]]
        got = p.scan(mode, case["haystack"])
        assert [(int(m["value"]), int(m["start"]), int(m["end"])) for m in got] == want, (runner, case["name"])
        assert p.scan_count(mode, case["haystack"]) == (len(want), orc.matches_checksum(got)), case["name"]
        it = p.find_iter(case["haystack"]) if mode == ScanMode.Find else p.leftmost_find_iter(case["haystack"])
        assert [(m.value(), m.start(), m.end()) for m in it] == want, case["name"]
        n += 1
    assert n == 61 + 93 + 91 and 0 < with_empty < 40


def test_known_answers(pins):
    for ka in pins["known_answers"]:
        if "patvals" in ka:
            pats, vals = [p for p, _ in ka["patvals"]], [v for _, v in ka["patvals"]]
        else:
            pats, vals = ka["patterns"], None
        _, p = _pma(pats, values=vals, kind=ka["kind"])
        engines = ENGINES if ka["api"].startswith("find_overlapping") else [Engine.Auto, Engine.DArray]
        for eng in engines:
            got = _sev(p.scan(API_MODE[ka["api"]], ka["haystack"], engine=eng))
            assert got == [tuple(t) for t in ka["matches_sev"]], (ka["cite"], eng)


def test_matchkind_mismatch_is_an_error(pins):
    """The reference panics (tests/matchkind_mismatch_test.rs); the C ABI returns status 5."""
    for e in pins["matchkind_mismatch"]["must_fail"]:
        mode = {"find_iter": ScanMode.Find, "find_overlapping_iter": ScanMode.FindOverlapping,
                "find_overlapping_no_suffix_iter": ScanMode.FindOverlappingNoSuffix,
                "leftmost_find_iter": ScanMode.LeftmostFind}[e["api"]]
        _, p = _pma(["a"], kind=e["kind"])
        with pytest.raises(da.DaachorseError) as ei:
            p.scan(mode, "")
        assert ei.value.code == 5


def test_empty_pattern_set_never_matches():
    _, p = _pma([])
    hay = bytes(itertools.chain.from_iterable((a, b) for a in range(0, 256, 3) for b in range(0, 256, 7)))
    for eng in ENGINES:
        assert len(p.scan(ScanMode.FindOverlapping, hay, engine=eng)) == 0
        assert p.scan_count(ScanMode.FindOverlapping, hay, engine=eng) == (0, 0)


@pytest.mark.parametrize("seg_bytes", [16, 48, 0])
def test_fuzz_small_alphabets(seg_bytes):
    """Random tiny pattern sets (incl. "", duplicates) over {a,b,c}; 16-byte segments make nearly
    every match straddle a segment boundary, which is what the halo has to get right."""
    rng = np.random.default_rng(1234 + seg_bytes)
    gram_runs = 0
    for it in range(60):
        npat = int(rng.integers(1, 7))
        pats = [bytes(rng.integers(97, 100, size=int(rng.integers(0, 6))).astype(np.uint8)) for _ in range(npat)]
        hay = rng.integers(97, 100 + (it % 2), size=int(rng.integers(0, 400)), dtype=np.uint8)
        o, p = _pma(pats)
        p.set_option("seg_bytes", seg_bytes)
        want = o.find_overlapping_iter(hay)
        want_ns = o.find_overlapping_no_suffix_iter(hay)
        for eng in ENGINES:
            got = p.scan(ScanMode.FindOverlapping, hay, engine=eng)
            assert _same(got, want), (pats, bytes(hay), eng, _sev(got)[:8], _sev(want)[:8])
            got_ns = p.scan(ScanMode.FindOverlappingNoSuffix, hay, engine=eng)
            assert _same(got_ns, want_ns), (pats, bytes(hay), eng)
        gram_runs += _check_counts(p, ScanMode.FindOverlapping, hay, want)
        _check_counts(p, ScanMode.FindOverlappingNoSuffix, hay, want_ns)
    assert gram_runs > 20  # only pattern sets containing "" are declined


def test_unaligned_and_device_haystacks():
    import torch
    rng = np.random.default_rng(5)
    pats = ["ab", "bca", "c", "abcab", "bb", "cabcabc"]
    o, p = _pma(pats)
    base = rng.integers(97, 100, size=70_000, dtype=np.uint8)
    dev = torch.from_numpy(base).cuda()
    for off in (0, 1, 7, 15, 16, 33):
        for n in (0, 1, 15, 16, 17, 4099, 65_536):
            want = o.find_overlapping_iter(base[off:off + n])
            for eng in ENGINES:
                got = p.scan(ScanMode.FindOverlapping, dev[off:off + n], engine=eng)
                assert _same(got, want), (off, n, eng)
                got_h = p.scan(ScanMode.FindOverlapping, base[off:off + n], engine=eng)
                assert _same(got_h, want), (off, n, eng)
            assert _check_counts(p, ScanMode.FindOverlapping, dev[off:off + n], want), (off, n)
            assert _check_counts(p, ScanMode.FindOverlapping, base[off:off + n], want), (off, n)


def test_lazy_iterator_windows():
    rng = np.random.default_rng(9)
    pats = ["abra", "cad", "abracadabra", "a", "ra"]
    o, p = _pma(pats)
    hay = rng.choice(np.frombuffer(b"abrcd", dtype=np.uint8), size=50_000)
    want = _sev(o.find_overlapping_iter(hay))
    p.set_option("iter_window", 4096)
    got = [(m.start(), m.end(), m.value()) for m in p.find_overlapping_iter(hay)]
    assert got == want


def test_byte_alphabet_falls_back_to_darray():
    """> 31 distinct pattern bytes: TIERED is unavailable, AUTO must still be exact."""
    rng = np.random.default_rng(11)
    pats = [bytes(rng.integers(0, 256, size=int(rng.integers(1, 9))).astype(np.uint8)) for _ in range(300)]
    o, p = _pma(pats)
    hay = rng.integers(0, 256, size=200_000, dtype=np.uint8)
    hay[1000:1000 + len(pats[0])] = np.frombuffer(pats[0], dtype=np.uint8)
    want = o.find_overlapping_iter(hay)
    assert _same(p.scan(ScanMode.FindOverlapping, hay), want)
    with pytest.raises(da.DaachorseError) as ei:
        p.scan(ScanMode.FindOverlapping, hay, engine=Engine.Tiered)
    assert ei.value.code == 6


def test_device_generators_match_numpy():
    import torch
    t = torch.empty(100_003, dtype=torch.uint8, device="cuda")
    synth.device_uniform(t, synth.SEEDS["cfg3_hay"], synth.ALPHA_LOWER_SPACE, offset=12345)
    assert np.array_equal(t.cpu().numpy(), synth.uniform_haystack(100_003, synth.SEEDS["cfg3_hay"], synth.ALPHA_LOWER_SPACE, offset=12345))
    words = synth.patterns_cfg3(500)
    synth.device_wordsoup(t, synth.SEEDS["cfg3_dense"], words, 20, offset=777)
    assert np.array_equal(t.cpu().numpy(), synth.wordsoup_haystack(100_003, synth.SEEDS["cfg3_dense"], words, 20, offset=777))


def test_cfg2_1000_patterns():
    """BASELINE configs[1] automaton on 8 MiB: sparse random-ASCII and dense pattern soup."""
    import torch
    pats = synth.patterns_cfg2()
    o, p = _pma(pats)
    n = 8 << 20
    sparse = synth.uniform_haystack(n, synth.SEEDS["cfg2_hay"], synth.ALPHA_PRINTABLE)
    dense = synth.wordsoup_haystack(n, synth.SEEDS["cfg2_dense"], pats, 13, noise_256=0)
    for hay in (sparse, dense):
        want = o.find_overlapping_iter(hay)
        dev = torch.from_numpy(hay).cuda()
        for eng in ENGINES:
            got = p.scan(ScanMode.FindOverlapping, dev, engine=eng)
            assert _same(got, want), eng
        assert _check_counts(p, ScanMode.FindOverlapping, dev, want)


def test_cfg3_100k_patterns():
    """BASELINE configs[2] automaton (100k words): full tuples on 4 MiB, count + checksum on 64 MiB."""
    import torch
    pats = synth.patterns_cfg3()
    o, p = _pma(pats)
    info = p.upload().info()
    assert info.tiered_available and info.gram_available and info.gram_k == 3
    small = synth.uniform_haystack(4 << 20, synth.SEEDS["cfg3_hay"], synth.ALPHA_LOWER_SPACE)
    want = o.find_overlapping_iter(small)
    for eng in ENGINES:
        got = p.scan(ScanMode.FindOverlapping, small, engine=eng)
        assert _same(got, want), eng
    n = 64 << 20
    dev = torch.empty(n, dtype=torch.uint8, device="cuda")
    synth.device_uniform(dev, synth.SEEDS["cfg3_hay"], synth.ALPHA_LOWER_SPACE)
    host = dev.cpu().numpy()
    want_cc = o.overlapping_count(host, threads=8)
    for eng in ENGINES + [Engine.Gram, Engine.Auto]:
        assert p.scan_count(ScanMode.FindOverlapping, dev, engine=eng) == want_cc, eng
    synth.device_wordsoup(dev, synth.SEEDS["cfg3_dense"], pats, 20)
    want_cc = o.overlapping_count(dev.cpu().numpy(), threads=8)
    for eng in ENGINES + [Engine.Gram, Engine.Auto]:
        assert p.scan_count(ScanMode.FindOverlapping, dev, engine=eng) == want_cc, eng
    # a small K (tables squeezed into a few KB of LDS) must give the same answer
    p2, _ = da.DoubleArrayAhoCorasick.deserialize(o.serialize())
    p2.set_option("gram_lds_budget", 9216)   # (read at upload)
    assert p2.upload().info().gram_k == 2
    assert p2.scan_count(ScanMode.FindOverlapping, dev, engine=Engine.Gram) == want_cc


def test_gram2_tables_on_the_device():
    """the GRAM engine's second table set (gram2.hpp: one M word per position) against the oracle: count alone and count +
    checksum, K = 3 and K = 2, both neighbour-exchange paths, unaligned haystacks, shards, dense and sparse text"""
    import torch
    rng = np.random.default_rng(77)
    pats3 = synth.patterns_cfg3(30000)
    cases = [(synth.patterns_cfg1(), synth.uniform_haystack(70001, 5, synth.ALPHA_ABCD)),
             (synth.patterns_cfg2(500), synth.uniform_haystack(1 << 20, 6, synth.ALPHA_LOWER)),
             (pats3, synth.uniform_haystack(3 << 20, synth.SEEDS["cfg3_hay"], synth.ALPHA_LOWER_SPACE)),
             (pats3, synth.wordsoup_haystack(3 << 20, synth.SEEDS["cfg3_dense"], pats3, 20)),
             (["ab", "ab", "b", "abab", "bababab"], np.frombuffer(b"abababbab" * 3000, dtype=np.uint8))]
    for pats, hay in cases:
        o, p = _pma(pats)
        info = p.upload().info()
        assert info.gram2_available and info.gram2_exact
        dev = torch.from_numpy(np.concatenate([np.zeros(5, dtype=np.uint8), hay])).cuda()[5:]  # not 16-byte aligned
        want = o.overlapping_count(hay, threads=8)
        for dpp, budget in ((1, 158 * 1024), (0, 158 * 1024), (1, 24 * 1024)):
            q, _ = da.DoubleArrayAhoCorasick.deserialize(o.serialize())
            q.set_option("gram2_dpp", dpp).set_option("gram_lds_budget", budget).set_option("gram_version", 2)
            assert q.scan_count(ScanMode.FindOverlapping, dev, engine=Engine.Gram) == want, (len(pats), dpp, budget)
            assert q.count(ScanMode.FindOverlapping, dev, engine=Engine.Gram) == want[0], (len(pats), dpp, budget)
            assert da.last_engine() == int(Engine.Gram)
            cut = int(rng.integers(1, len(hay)))
            head, tail = q.count(ScanMode.FindOverlapping, dev[:cut]), q.count(ScanMode.FindOverlapping, dev, begin=cut)
            assert head + tail == want[0], cut
        if budget == 24 * 1024 and len(pats) > 1000:
            assert q.info().gram2_k == 2
    # declined automata: the request falls through to the other engines / an error for engine = GRAM with version 2
    o, p = _pma(["", "a"])
    assert not p.upload().info().gram2_available
    assert p.count(ScanMode.FindOverlapping, b"aaa") == 7


def test_wide_alphabet_dictionary_stays_on_gram():
    """a 60-class dictionary (the cfg3 words in mixed case, some with a digit): beyond the 32-bit tables, served by the wide
    GRAM engine (gram2w.hpp) instead of falling to DARRAY; count alone, count + checksum, shards, tuples (segment scanners)"""
    import torch
    pats = synth.patterns_cfg3_wide(30000)
    o, p = _pma(pats)
    info = p.upload().info()
    assert info.gram_available and info.gram_wide and not info.tiered_available and info.num_classes == 61
    for hay in (synth.uniform_haystack((3 << 20) + 5, synth.SEEDS["cfg3_hay"], synth.ALPHA_WIDE_SPACE),
                synth.wordsoup_haystack(3 << 20, 11, pats, 20, alphabet=synth.ALPHA_WIDE)):
        dev = torch.from_numpy(np.concatenate([np.zeros(7, dtype=np.uint8), hay])).cuda()[7:]
        want = o.overlapping_count(hay, threads=8)
        assert p.scan_count(ScanMode.FindOverlapping, dev, engine=Engine.Gram) == want
        assert p.scan_count(ScanMode.FindOverlapping, dev) == want and da.last_engine() == int(Engine.Gram)
        assert p.count(ScanMode.FindOverlapping, dev) == want[0] and da.last_engine() == int(Engine.Gram)
        assert p.scan_count(ScanMode.FindOverlapping, dev, engine=Engine.DArray) == want
        cut = 1234567
        assert p.count(ScanMode.FindOverlapping, dev[:cut]) + p.count(ScanMode.FindOverlapping, dev, begin=cut) == want[0]
        small = hay[:200000]
        assert _same(p.scan(ScanMode.FindOverlapping, small), o.find_overlapping_iter(small))
    # 61 distinct pattern bytes, short patterns, text over a wider byte range
    rng = np.random.default_rng(8)
    some = list(dict.fromkeys(bytes(rng.integers(33, 94, size=int(rng.integers(1, 9))).astype(np.uint8)) for _ in range(3000)))
    o, p = _pma(some)
    hay = rng.integers(30, 97, size=2_000_000).astype(np.uint8)
    assert p.scan_count(ScanMode.FindOverlapping, hay, engine=Engine.Gram) == o.overlapping_count(hay, threads=8)


def _same16(got16, want):
    return len(got16) == len(want) and np.array_equal(got16["end"], want["end"]) and np.array_equal(got16["value"], want["value"]) and \
        np.array_equal(got16["length"].astype(np.uint64), want["end"] - want["start"])


def test_gram_tuple_emitter():
    """daac_scan_device / daac_scan_device16 / daac_scan through the GRAM tuple emitter (emit3_kernels.hip: detection once, then expansion): bit-exact tuples in
    the reference's order, in both device formats, on texts that stress its seams — matches that straddle tile (1024 B) and region
    boundaries, lazy windows that begin inside matches, unaligned device haystacks, both K, patterns longer than K + 16 bytes and
    duplicate patterns (placed as extras), record-list overflow (falls back) — and the list left in device memory equals the one
    copied to the host"""
    import torch
    rng = np.random.default_rng(123)
    pats3 = synth.patterns_cfg3(30000)
    long_pats = [b"abcdefghijklmnop", b"bcdefghijklmnopq", b"mnopqrs", b"ponmlkjihg", b"qrstuv", b"a", b"op", b"nop", b"lmnopqrstuvwxyzabc"]
    # a word list with what round 2's emitter declined: five words of 25-40 bytes, a duplicate, nested long words, copies of a long word
    extra_pats = pats3[:20000] + [pats3[i] + pats3[i + 1] + pats3[i + 2] + pats3[i + 3] for i in (3, 50, 700, 1200, 9000)] + \
        [next(w for w in pats3 if len(w) >= 6), pats3[4000] * 3, pats3[4000] * 3, pats3[4000] * 3 + b"s", b"q" * 30, b"q" * 31]
    assert max(len(w) for w in extra_pats) > 30 and len(set(extra_pats)) < len(extra_pats)
    # (the special words spread over the text: a tile places at most 64 extras, denser texts fall back to the segment scanners — below)
    extra_text = b" ".join(extra_pats[i] for i in rng.permutation(np.concatenate([rng.integers(0, len(extra_pats), size=40000), rng.integers(20000, 20009, size=1500),
                                                                                   rng.integers(20009, len(extra_pats), size=150)])).tolist()) + b" " + b"q" * 40
    cases = [(synth.patterns_cfg1(), synth.uniform_haystack(9000, 3, synth.ALPHA_ABCD)),
             (long_pats, np.frombuffer((b"abcdefghijklmnopqrstuvwxyzabc" * 400)[:11000], dtype=np.uint8)),
             (long_pats, synth.uniform_haystack(20000, 4, b"abcdefghijklmnopqrstuvwxyz")),
             (synth.patterns_cfg2(500), synth.wordsoup_haystack(300000, 8, synth.patterns_cfg2(500), 13, noise_256=30)),
             (pats3, synth.uniform_haystack((1 << 20) + 777, synth.SEEDS["cfg3_hay"], synth.ALPHA_LOWER_SPACE)),
             (pats3, synth.wordsoup_haystack(1 << 20, synth.SEEDS["cfg3_dense"], pats3, 20)),
             (extra_pats, np.frombuffer(extra_text, dtype=np.uint8))]
    for pats, hay in cases:
        o, _ = _pma(pats)
        want = o.find_overlapping_iter(hay)
        for tiles, budget, shift in ((64, 158 * 1024, 0), (1, 158 * 1024, 3), (2, 24 * 1024, 9), (3, 158 * 1024, 5)):
            p, _ = da.DoubleArrayAhoCorasick.deserialize(o.serialize())
            p.set_option("gram_region", 2048 * tiles if tiles < 64 else 0)   # regions of one to three DETECT steps: every seam between waves
            p.set_option("gram_lds_budget", budget)   # (read at upload)
            dev = torch.from_numpy(np.concatenate([np.zeros(shift, dtype=np.uint8), hay])).cuda()[shift:]
            got = p.scan(ScanMode.FindOverlapping, dev, engine=Engine.Gram)  # Engine.Gram: no silent fallback
            assert _same(got, want), (len(pats), len(hay), tiles, budget)
            dm = p.scan_device(ScanMode.FindOverlapping, dev)
            assert da.last_engine() == int(Engine.Gram) and dm.count == len(want)
            assert _same(dm.to_numpy(), want)
            if len(want) > 100:
                mid = dm.to_numpy(first=len(want) // 2, n=50)
                assert _same(mid, want[len(want) // 2:len(want) // 2 + 50])
            with pytest.raises(da.DaachorseError):
                dm.to_numpy(first=len(want), n=1)
            dm.free()
            d16 = p.scan_device(ScanMode.FindOverlapping, dev, fmt16=True)
            assert da.last_engine() == int(Engine.Gram) and d16.count == len(want) and _same16(d16.to_numpy(), want)
            d16.free()
        # lazy windows begin wherever the previous one ended: inside matches, off the tile grid
        p, _ = da.DoubleArrayAhoCorasick.deserialize(o.serialize())
        p.set_option("iter_window", 4096 + 37)
        sub = hay[:60000]
        wsub = o.find_overlapping_iter(sub)
        lazy = [(m.start(), m.end(), m.value()) for m in p.find_overlapping_iter(sub)]
        assert lazy == [(int(x["start"]), int(x["end"]), int(x["value"])) for x in wsub]
    # more deep matches between two checkpoints than a chunk holds: the scan falls back and stays exact
    pats = [b"a" * k for k in range(1, 17)]
    hay = np.frombuffer(b"a" * 5000 + b"b" + b"a" * 3000, dtype=np.uint8)
    o, p = _pma(pats)
    want = o.find_overlapping_iter(hay)
    got = p.scan(ScanMode.FindOverlapping, hay)
    # (thirteen deep matches per position: the emitter gives this text up and the segment scanners serve it)
    assert _same(got, want) and da.last_engine() != int(Engine.Gram)
    d16 = p.scan_device(ScanMode.FindOverlapping, hay, fmt16=True)  # another engine's list, repacked on the device
    assert da.last_engine() != int(Engine.Gram) and _same16(d16.to_numpy(), want)
    d16.free()
    # more extras in one tile than the expansion places (every prefix of a long run a pattern): falls back too
    o, p = _pma([b"z" * k for k in range(1, 60)])
    hay = np.frombuffer(b"z" * 4000, dtype=np.uint8)
    assert _same(p.scan(ScanMode.FindOverlapping, hay), o.find_overlapping_iter(hay)) and da.last_engine() != int(Engine.Gram)
    # a record list sized too small: DETECT counts what it cannot store and the scan is rerun with the exact size
    o, p = _pma(pats3)
    p.set_option("emit_rec_per_kib", 1)
    hay = synth.wordsoup_haystack(3 << 20, 5, pats3, 20)
    got = p.scan(ScanMode.FindOverlapping, hay, engine=Engine.Gram)
    assert _same(got, o.find_overlapping_iter(hay))
    # the other iterators in the 16-byte format (repacked)
    o, p = _pma(pats3[:2000])
    hay = synth.wordsoup_haystack(50000, 3, pats3[:2000], 20)
    d16 = p.scan_device(ScanMode.Find, hay, fmt16=True)
    assert _same16(d16.to_numpy(), o.find_iter(hay))
    # automata the emitter declines (duplicates among the short patterns): still served, by the segment scanners
    o, p = _pma(["ab", "ab", "abc"])
    assert _same(p.scan(ScanMode.FindOverlapping, b"xabcabab"), o.find_overlapping_iter(b"xabcabab"))
    with pytest.raises(da.DaachorseError) as ei:
        p.scan(ScanMode.FindOverlapping, b"xabcabab", engine=Engine.Gram)
    assert ei.value.code == 6


def test_shard_tail_counts_add_up():
    """daac_scan_count_range: the matches with end in (begin, len] — what one device of a sharded
    haystack contributes.  Shards of one haystack must add up to the whole, at any split point."""
    import torch
    from daachorse_amd import dist as ddist
    pats = synth.patterns_cfg3(5000)
    o, p = _pma(pats)
    n = 1_000_003
    hay = synth.wordsoup_haystack(n, synth.SEEDS["cfg3_dense"], pats, 20)
    dev = torch.from_numpy(hay).cuda()
    whole = o.overlapping_count(hay)
    all_m = o.find_overlapping_iter(hay)
    for cuts in ([0, n], [0, 1, n], [0, 333_333, 666_671, n], [0, 17, 4096, 500_000, n - 1, n]):
        tot_c, tot_1, tot_2 = 0, 0, 0
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            want = all_m[(all_m["end"] > lo) & (all_m["end"] <= hi)]
            for src in (dev[:hi], hay[:hi]):
                got = p.scan_count(ScanMode.FindOverlapping, src, begin=lo)
                assert got == (len(want), orc.matches_checksum(want)), (lo, hi)
            s1, s2 = ddist.split_checksum(got[1])
            tot_c, tot_1, tot_2 = tot_c + got[0], tot_1 + s1, tot_2 + s2
        assert (tot_c, ddist.join_checksum(tot_1, tot_2)) == whole, cuts


@pytest.mark.parametrize("chain", [1, 0, 2])
@pytest.mark.parametrize("seg_bytes", [16, 64, 0])
def test_fuzz_find_and_leftmost(seg_bytes, chain):
    """find_iter / leftmost_find_iter against the literal iterators of the oracle: tiny alphabets (few
    sync points, long chains across segments), periodic texts that never synchronise, separators.
    chain = 1: speculate / reconcile / emit; 0: the sync-point scanners; 2: one reconciliation round only,
    so that texts whose chains do not fall in step at once exercise the fallback."""
    rng = np.random.default_rng(4321 + seg_bytes)
    for it in range(50):
        npat = int(rng.integers(1, 7))
        pats = [bytes(rng.integers(97, 100, size=int(rng.integers(1, 6))).astype(np.uint8)) for _ in range(npat)]
        if it % 7 == 0:
            pats.append(b"")
        if it % 5 == 0:
            hay = np.frombuffer((b"ab" * 150)[:int(rng.integers(1, 300))], dtype=np.uint8)  # periodic: no sync point
        else:
            hay = rng.integers(97, 100 + (it % 3), size=int(rng.integers(0, 500)), dtype=np.uint8)
        for kind in ("Standard", "LeftmostLongest", "LeftmostFirst"):
            o, p = _pma(pats, kind=kind)
            p.set_option("seg_bytes", seg_bytes).set_option("restart_chain", 1 if chain else 0).set_option("chain_rounds", 1 if chain == 2 else 24)
            if kind == "Standard":
                want, mode = o.find_iter(hay), ScanMode.Find
            else:
                mode = ScanMode.LeftmostFind
                try:
                    want = o.leftmost_find_iter(hay)
                except orc.OracleError as e:
                    # "" in the set and the haystack ends inside a longer pattern: the reference never terminates
                    # (SURVEY 8a note D); the boundary says so instead of imitating it
                    assert e.code == 6 and b"" in pats
                    with pytest.raises(da.DaachorseError) as ei:
                        p.scan(mode, hay)
                    assert ei.value.code == 6
                    with pytest.raises(da.DaachorseError) as ei:
                        p.scan_count(mode, hay)
                    assert ei.value.code == 6
                    continue
            got = p.scan(mode, hay)
            assert _same(got, want), (kind, pats, bytes(hay), _sev(got)[:6], _sev(want)[:6])
            assert p.scan_count(mode, hay) == (len(want), orc.matches_checksum(want)), (kind, pats)


def test_sync_point_verdicts_do_not_depend_on_the_warm_up():
    """Regression (found by tools/stress.py): text that never returns to ROOT, a pattern of maximal length ending
    exactly at a segment cut.  A lane warming up over Lmax - 1 bytes saw ROOT there, a lane that had followed the
    text did not, and their regions overlapped."""
    for pats, unit in ((["cab", "cbabab"], "cabcbabab"), (["ab", "abcabc", "cabca"], "abcabc"), (["aaaa", "a"], "aaaa")):
        text = (unit * 400).encode()
        for kind, api, mode in (("LeftmostLongest", "leftmost_find_iter", ScanMode.LeftmostFind), ("LeftmostFirst", "leftmost_find_iter", ScanMode.LeftmostFind),
                                ("Standard", "find_iter", ScanMode.Find)):
            o, p = _pma(pats, kind=kind)
            p.set_option("restart_chain", 0)
            want = getattr(o, api)(text)
            for seg in (16, 32, 48, 64, 0):
                p.set_option("seg_bytes", seg)
                assert _same(p.scan(mode, text), want), (pats, kind, seg)
                assert p.scan_count(mode, text) == (len(want), orc.matches_checksum(want)), (pats, kind, seg)


def test_find_and_leftmost_dictionaries():
    """cfg2 / cfg3-style automata, all three kinds, MiB-sized sparse and dense haystacks, lazy windows."""
    import torch
    for pats, dense_slot in ((synth.patterns_cfg2(), 13), (synth.patterns_cfg3(20000), 20)):
        n = 2 << 20
        sparse = synth.uniform_haystack(n, synth.SEEDS["cfg3_hay"], synth.ALPHA_LOWER_SPACE)
        dense = synth.wordsoup_haystack(n, synth.SEEDS["cfg3_dense"], pats, dense_slot)
        for kind in ("Standard", "LeftmostLongest", "LeftmostFirst"):
            o, p = _pma(pats, kind=kind)
            for hay in (sparse, dense):
                want = o.find_iter(hay) if kind == "Standard" else o.leftmost_find_iter(hay)
                mode = ScanMode.Find if kind == "Standard" else ScanMode.LeftmostFind
                dev = torch.from_numpy(hay).cuda()
                assert _same(p.scan(mode, dev), want), kind
                assert p.scan_count(mode, dev) == (len(want), orc.matches_checksum(want)), kind
    # lazy iterator across windows that end at sync points
    o, p = _pma(pats, kind="LeftmostLongest")
    p.set_option("iter_window", 8192)
    small = dense[:200_000]
    want = _sev(o.leftmost_find_iter(small))
    assert [(m.start(), m.end(), m.value()) for m in p.leftmost_find_iter(small)] == want
    o, p = _pma(pats)
    want = _sev(o.find_iter(small))
    assert [(m.start(), m.end(), m.value()) for m in p.find_iter(small)] == want


def test_cpp_facade_on_gpu(tmp_path):
    """The C++ host façade (include/daachorse_amd.hpp) end to end: the README examples of the reference."""
    import os
    import subprocess
    from conftest import ROOT
    exe = str(tmp_path / "cpp_facade_test")
    libdir = os.path.join(ROOT, "daachorse_amd", "lib")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "native", "cpp_facade_test.cpp"),
                           "-L" + libdir, "-ldaachorse_amd", "-Wl,-rpath," + libdir])
    r = subprocess.run([exe, "gpu"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0 and r.stdout.decode().strip().endswith("OK gpu"), r.stdout.decode()


def test_concurrent_scans_share_one_handle():
    """Handles are immutable after upload: several host threads scan through one handle, each on its own HIP
    stream (include/daachorse_amd.h conventions); all four iterators, materialising and counting."""
    import threading
    import torch
    pats = synth.patterns_cfg3(3000)
    o, p = _pma(pats)
    ol, pl = _pma(pats, kind="LeftmostLongest")
    hay = synth.wordsoup_haystack(400_000, synth.SEEDS["cfg3_dense"], pats, 20)
    dev = torch.from_numpy(hay).cuda()
    want = {"ov": o.find_overlapping_iter(hay), "find": o.find_iter(hay), "lm": ol.leftmost_find_iter(hay)}
    p.set_option("pfx", 2)   # (read at upload: PFX tables as well)
    p.upload(0)
    pl.upload(0)
    errors = []

    def worker(tid):
        try:
            s = torch.cuda.Stream()
            for rep in range(6):
                assert _same(p.scan(ScanMode.FindOverlapping, dev, stream=s.cuda_stream), want["ov"])
                assert p.scan_count(ScanMode.FindOverlapping, dev, stream=s.cuda_stream) == (len(want["ov"]), orc.matches_checksum(want["ov"]))
                assert p.count(ScanMode.FindOverlapping, dev, stream=s.cuda_stream) == len(want["ov"])                      # gram3
                assert p.count(ScanMode.FindOverlapping, dev, engine=Engine.Pfx, stream=s.cuda_stream) == len(want["ov"])   # PFX
                assert _same(p.scan(ScanMode.Find, dev, stream=s.cuda_stream), want["find"])
                assert pl.scan_count(ScanMode.LeftmostFind, dev, stream=s.cuda_stream) == (len(want["lm"]), orc.matches_checksum(want["lm"]))
                assert [(m.start(), m.end()) for m, _ in zip(p.find_iter(hay[:5000]), range(50))] == \
                    [(int(x["start"]), int(x["end"])) for x in o.find_iter(hay[:5000])[:50]]
        except Exception as e:  # noqa: BLE001
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_gram_haystack_beyond_4_gib():
    """BASELINE-sized property: positions above 2^32 (the walker slabs change epoch at every multiple of 4 GiB,
    checksums use the low 32 bits of `end`).  5 GiB + 5 bytes of word soup at an unaligned address, GRAM
    against the oracle (128 host threads); the two halves of the haystack must also add up."""
    import os
    import torch
    pats = synth.patterns_cfg3(20000)
    o, p = _pma(pats)
    n = (5 << 30) + 5
    buf = torch.empty(n + 16, dtype=torch.uint8, device="cuda")
    dev = buf[3:3 + n]
    synth.device_wordsoup(dev, synth.SEEDS["cfg3_dense"], pats, 20)
    p.set_option("pfx", 2)
    p.upload()
    got = p.scan_count(ScanMode.FindOverlapping, dev, engine=Engine.Gram)
    host = dev.cpu().numpy()
    want = o.overlapping_count(host, threads=min(128, os.cpu_count() or 1))
    assert got == want
    assert p.count(ScanMode.FindOverlapping, dev) == want[0] and da.last_engine() == int(Engine.Gram)   # the gram3 kernel
    assert p.count(ScanMode.FindOverlapping, dev, engine=Engine.Pfx) == want[0]
    assert p.scan_count(ScanMode.FindOverlapping, dev, engine=Engine.Pfx) == want
    cut = (4 << 30) - 7  # tail shard through the materialising engines' count mode
    head = p.scan_count(ScanMode.FindOverlapping, dev[:cut], engine=Engine.Gram)
    tail = p.scan_count(ScanMode.FindOverlapping, dev, begin=cut)
    from daachorse_amd import dist as ddist
    h1, h2 = ddist.split_checksum(head[1])
    t1, t2 = ddist.split_checksum(tail[1])
    assert (head[0] + tail[0], ddist.join_checksum(h1 + t1, h2 + t2)) == want


def test_gram_text_made_of_patterns_overflows_nothing():
    """Regression (found by tools/stress.py): text in which EVERY position continues deep in the trie queues a
    walker per position — more than the slab margin of a step allowed for, so entries spilled into the
    neighbouring wave's slab.  Pattern-made text over small alphabets, both positions-per-lane variants."""
    import torch
    rng = np.random.default_rng(1051)
    for nsym, npat, lo, hi, sep in ((4, 500, 4, 8, b" "), (7, 5, 2, 7, b""), (3, 40, 5, 12, b""), (2, 6, 3, 9, b"")):
        syms = np.arange(97, 97 + nsym, dtype=np.uint8)
        pats = [bytes(syms[rng.integers(0, nsym, size=int(rng.integers(lo, hi + 1)))]) for _ in range(npat)]
        idx = rng.integers(0, npat, size=200_000)
        hay = np.frombuffer(sep.join(pats[i] for i in idx.tolist())[:900_000], dtype=np.uint8).copy()
        o, p = _pma(pats)
        dev = torch.from_numpy(hay).cuda()[1:]
        want = o.overlapping_count(dev.cpu().numpy(), threads=8)
        for eng in (Engine.Auto, Engine.Tiered, Engine.DArray):
            assert p.scan_count(ScanMode.FindOverlapping, dev, engine=eng) == want, (nsym, npat, eng)
    # second defect of the same family: every position hits AND continues, so a step retires its own 64 * P hits plus
    # what the previous step left in the stack and in flight (K = 2 tables, 16 positions per lane, smallest slab)
    syms = np.frombuffer(b"acinrs", dtype=np.uint8)
    pats = [bytes(syms[rng.integers(0, 6, size=int(rng.integers(4, 9)))]) for _ in range(5000)]
    hay = np.frombuffer(b"".join(pats[i] for i in rng.integers(0, 5000, size=60_000).tolist())[:300_000], dtype=np.uint8).copy()
    o, p = _pma(pats)
    p.set_option("gram_lds_budget", 9216).set_option("gram_ppl", 16).set_option("gram_slab", 0)
    dev = torch.from_numpy(hay).cuda()[13:]
    want = o.overlapping_count(dev.cpu().numpy(), threads=8)
    assert p.upload().info().gram_k == 2
    assert p.scan_count(ScanMode.FindOverlapping, dev, engine=Engine.Gram) == want
    p.set_option("gram_ppl")
    assert p.scan_count(ScanMode.FindOverlapping, dev, engine=Engine.Gram) == want


def test_chain_walkers_step_back_and_window_edges():
    """The micro-step chain walkers keep 32 resident bytes of text and slide forward; a leftmost match that ends far
    behind the walker (a long pattern that fails at its last byte) steps back beyond the window, short texts end inside the
    first granule, and multi-byte characters straddle granules.  Against the oracle's literal iterators, several segment sizes."""
    long_a = b"a" * 45 + b"b"
    cases = [
        ([b"a", long_a], (b"a" * 45 + b"c") * 40 + b"a" * 45 + b"b" + b"a" * 7),
        ([b"ab", b"abababababababababababababababababababababc"], b"ab" * 300),
        ([b"xyz", b"x" * 30 + b"y"], b"x" * 29 + b"z" + b"xyz" * 50 + b"x" * 31),
        ([b"a", b"ab"], b"a"), ([b"a", b"ab"], b""), ([b"a", b"ab"], b"ab"), ([b"abcdefgh" * 3], b"abcdefgh" * 3),
    ]
    for pats, text in cases:
        hay = np.frombuffer(text, dtype=np.uint8)
        for kind, api, mode in (("LeftmostLongest", "leftmost_find_iter", ScanMode.LeftmostFind), ("LeftmostFirst", "leftmost_find_iter", ScanMode.LeftmostFind),
                                ("Standard", "find_iter", ScanMode.Find)):
            o, p = _pma(pats, kind=kind)
            want = getattr(o, api)(hay)
            for seg in (0, 16, 48, 256):
                p.set_option("seg_bytes", seg)
                assert _same(p.scan(mode, hay), want), (pats, kind, seg)
                assert p.scan_count(mode, hay) == (len(want), orc.matches_checksum(want)), (pats, kind, seg)
                assert p.count(mode, hay) == len(want), (pats, kind, seg)
    # charwise: three-byte characters (a granule boundary falls inside every other one), ASCII in between, a long pattern
    cpats = ["世界", "全世界", "世", "にほんごのながいぱたーんです", "abc", "b"]
    text = ("全世界中に世界の世abcにほんごのながいぱたーんでx" * 60 + "にほんごのながいぱたーんです" + "世") * 3
    for kind, api, mode in ((1, "leftmost_find_iter", ScanMode.LeftmostFind), (2, "leftmost_find_iter", ScanMode.LeftmostFind), (0, "find_iter", ScanMode.Find)):
        co = orc.OracleCharwisePma.build(cpats, kind=kind)
        cp, _ = da.CharwiseDoubleArrayAhoCorasick.deserialize(co.serialize())
        want = getattr(co, api)(text)
        for seg in (0, 16, 64, 1024):
            for rows in (1, 0):
                cp2, _ = da.CharwiseDoubleArrayAhoCorasick.deserialize(co.serialize())
                cp2.set_option("seg_bytes", seg).set_option("char_row_lds", rows)   # (the second one read at upload)
                assert _same(cp2.scan(mode, text), want), (kind, seg, rows)
                assert cp2.scan_count(mode, text) == (len(want), orc.matches_checksum(want)), (kind, seg, rows)


def test_gram_tail_records():
    """`.count()` folds single trie paths into tail records compared with the text eight bytes at a time: words that are
    prefixes of longer words on one path, paths longer than eight bytes (a tail below ordinary records), the byte 0x00 as
    a pattern byte (the bytes shifted into a spent look-ahead are zeros too), paths running into the end of the text."""
    rng = np.random.default_rng(77)
    stems = [bytes(rng.integers(97, 101, size=5).astype(np.uint8)) for _ in range(40)]
    pats = set()
    for s in stems:
        tail = bytes(rng.integers(97, 123, size=16).astype(np.uint8))
        for cut in (0, 1, 3, 7, 8, 9, 12, 16):
            pats.add(s + tail[:cut])
    pats = sorted(pats)
    o, p = _pma(pats)
    text = bytearray()
    for i in range(3000):
        w = pats[int(rng.integers(0, len(pats)))]
        text += w[:int(rng.integers(4, len(w) + 1))] if i % 3 == 0 else w
        text += bytes(rng.integers(97, 123, size=int(rng.integers(0, 3))).astype(np.uint8))
    text += pats[-1][:-1]  # a path that runs into the end of the text
    hay = np.frombuffer(bytes(text), dtype=np.uint8)
    want = o.find_overlapping_iter(hay)
    assert _check_counts(p, ScanMode.FindOverlapping, hay, want)
    for cut in (1, 5, 9, 13):  # ... and at every distance from the end
        h2 = hay[:len(hay) - cut]
        assert p.count(ScanMode.FindOverlapping, h2) == len(o.find_overlapping_iter(h2)), cut
    # 0x00 .. 0x03 as the alphabet
    zpats = sorted({bytes(rng.integers(0, 4, size=int(rng.integers(1, 14))).astype(np.uint8)) for _ in range(300)})
    zo, zp = _pma(zpats)
    zhay = np.concatenate([np.frombuffer(zpats[int(rng.integers(0, len(zpats)))], dtype=np.uint8) for _ in range(4000)] + [np.zeros(9, dtype=np.uint8)])
    zwant = zo.find_overlapping_iter(zhay)
    _check_counts(zp, ScanMode.FindOverlapping, zhay, zwant)


def test_engine_options_do_not_change_answers():
    """The knobs that pick between implementations of the same scan (the micro-step walker
    or the byte-at-a-time segment scanners for counts, mapper / ROOT's row in LDS or in L2, sync-point scanners or chains)
    must not change a single number."""
    rng = np.random.default_rng(5)
    pats = synth.patterns_cfg3(3000)
    hay = synth.wordsoup_haystack(300_000, synth.SEEDS["cfg3_dense"], pats, 20)
    o, _ = _pma(pats)
    ol, _ = _pma(pats, kind="LeftmostLongest")
    want = {"ov": o.find_overlapping_iter(hay), "find": o.find_iter(hay), "lm": ol.leftmost_find_iter(hay)}
    cpats = ["".join(chr(0x4E00 + int(c)) for c in rng.integers(0, 300, size=int(rng.integers(1, 5)))) for _ in range(2000)] + ["ab", "b", "全世界"]
    ctext = "".join(chr(0x4E00 + int(c)) if c < 300 else "ab "[int(c) % 3] for c in rng.integers(0, 340, size=60_000))
    co = orc.OracleCharwisePma.build(cpats)
    col = orc.OracleCharwisePma.build(cpats, kind=1)
    cwant = {"ov": co.find_overlapping_iter(ctext), "find": co.find_iter(ctext), "lm": col.leftmost_find_iter(ctext)}
    for opts in ({}, {"overlap_micro": 0}, {"overlap_micro": 2}, {"char_map_lds": 0}, {"char_row_lds": 0}, {"restart_chain": 0},
                 {"seg_bytes": 4096, "overlap_micro": 2}):
        _, p = _pma(pats)       # (tables are laid out at upload: a fresh handle per setting, the setting the handle's own)
        _, pl = _pma(pats, kind="LeftmostLongest")
        cp, _ = da.CharwiseDoubleArrayAhoCorasick.deserialize(co.serialize())
        cpl, _ = da.CharwiseDoubleArrayAhoCorasick.deserialize(col.serialize())
        for handle in (p, pl, cp, cpl):
            for k, v in opts.items():
                handle.set_option(k, v)
        for eng in (Engine.Auto, Engine.DArray, Engine.Tiered):
            assert p.scan_count(ScanMode.FindOverlapping, hay, engine=eng) == (len(want["ov"]), orc.matches_checksum(want["ov"])), (opts, eng)
        assert p.scan_count(ScanMode.Find, hay) == (len(want["find"]), orc.matches_checksum(want["find"])), opts
        assert pl.scan_count(ScanMode.LeftmostFind, hay) == (len(want["lm"]), orc.matches_checksum(want["lm"])), opts
        assert _same(pl.scan(ScanMode.LeftmostFind, hay), want["lm"]), opts
        dm = p.scan_device(ScanMode.FindOverlapping, hay)
        assert _same(dm.to_numpy(), want["ov"]), opts
        dm.free()
        assert cp.scan_count(ScanMode.FindOverlapping, ctext) == (len(cwant["ov"]), orc.matches_checksum(cwant["ov"])), opts
        assert cp.scan_count(ScanMode.Find, ctext) == (len(cwant["find"]), orc.matches_checksum(cwant["find"])), opts
        assert cpl.scan_count(ScanMode.LeftmostFind, ctext) == (len(cwant["lm"]), orc.matches_checksum(cwant["lm"])), opts
        assert _same(cp.scan(ScanMode.FindOverlapping, ctext), cwant["ov"]), opts


def test_compact_lazy_iterator():
    """daac_iter_open_compact / daac_iter_next_batch8: the same match stream with 8 bytes per tuple over PCIe ({value, end relative to the
    run's base | length << end_bits}) — every iterator of the crate, bytewise and charwise, windows small enough for many runs, matches that straddle windows;
    the two run formats refuse each other's iterators."""
    rng = np.random.default_rng(12)
    pats = synth.patterns_cfg3(5000)
    hay = synth.wordsoup_haystack(400_000, 9, pats, 20)
    cpats = synth.patterns_cfg5(3000)
    chay = synth.zipf_text(48 * 6000)
    for kind, mode, api in ((0, ScanMode.FindOverlapping, "find_overlapping_iter"), (0, ScanMode.Find, "find_iter"),
                            (0, ScanMode.FindOverlappingNoSuffix, "find_overlapping_no_suffix_iter"), (1, ScanMode.LeftmostFind, "leftmost_find_iter")):
        for charwise in (False, True):
            if charwise:
                o = orc.OracleCharwisePma.build(cpats, kind=kind)
                p, _ = da.CharwiseDoubleArrayAhoCorasick.deserialize(o.serialize())
                h = chay
            else:
                o = orc.OraclePma.build(pats, kind=kind)
                p, _ = da.DoubleArrayAhoCorasick.deserialize(o.serialize())
                h = hay
            p.set_option("iter_window", 50_000)   # (the handle's own: daac_pma_set_option)
            want = getattr(o, api)(h)
            it = getattr(p, api)(h, compact=True)
            ends, lens, vals, nruns = [], [], [], 0
            while True:
                got = it.next_batch8()
                if got is None:
                    break
                run, base, eb = got
                nruns += 1
                ends.append((run["end_len"] & np.uint32((1 << eb) - 1)).astype(np.uint64) + np.uint64(base))
                lens.append(run["end_len"] >> np.uint32(eb)); vals.append(run["value"].copy())
            it.close()
            assert nruns > 3, (api, charwise)
            e, l, v = np.concatenate(ends), np.concatenate(lens), np.concatenate(vals)
            assert len(e) == len(want) and np.array_equal(e, want["end"]) and np.array_equal(l, (want["end"] - want["start"]).astype(np.uint32)) and \
                np.array_equal(v, want["value"]), (api, charwise)
            # match by match on a compact iterator, and the 16-byte runs on an ordinary one
            k = int(rng.integers(1000, 3000))
            got = [(m.start(), m.end(), m.value()) for _, m in zip(range(k), getattr(p, api)(h, compact=True))]
            assert got == orc.triples_sev(want[:k]), (api, charwise)
    p, _ = da.DoubleArrayAhoCorasick.deserialize(orc.OraclePma.build(pats).serialize())
    p.set_option("iter_window", 50_000)
    with pytest.raises(da.DaachorseError) as ei:
        p.find_overlapping_iter(hay, compact=True).next_batch()
    assert ei.value.code == 6
    with pytest.raises(da.DaachorseError) as ei:
        p.find_overlapping_iter(hay).next_batch8()
    assert ei.value.code == 6
    # a dictionary with a pattern of several KB has no compact form (the 16-byte iterator serves it)
    o = orc.OraclePma.build([b"ab", b"x" * 5000])
    q, _ = da.DoubleArrayAhoCorasick.deserialize(o.serialize())
    q.set_option("iter_window", 50_000)
    with pytest.raises(da.DaachorseError) as ei:
        q.find_overlapping_iter(hay, compact=True)
    assert ei.value.code == 6
    assert [(m.start(), m.end(), m.value()) for m in q.find_overlapping_iter(b"zabab" + b"x" * 5001)] == orc.triples_sev(o.find_overlapping_iter(b"zabab" + b"x" * 5001))


def test_scan_count_multi_shards_on_one_device():
    """daac_scan_count_multi (SURVEY 8e in the product: one haystack sharded across the devices of a node, host-side sum): shards that all
    name device 0 — random cuts, halos of exactly max_pattern_len - 1 bytes and longer, device and host buffers, bytewise (GRAM, PFX,
    DARRAY dictionaries) and charwise — equal the whole haystack's count + checksum from the oracle; a halo that is too short and the
    chain iterators are refused."""
    import torch
    rng = np.random.default_rng(808)
    pats3 = synth.patterns_cfg3(20000)
    cases = [(pats3, synth.wordsoup_haystack(3 << 20, synth.SEEDS["cfg3_dense"], pats3, 20), False),
             (synth.patterns_cfg2(500), synth.uniform_haystack(2 << 20, 6, synth.ALPHA_LOWER), False),
             (synth.patterns_binary256(3000), rng.integers(0, 256, size=1 << 20).astype(np.uint8), False),
             (["全世界", "世界", "に", "世", "界に"], np.frombuffer(("全世界に世界に" * 40000).encode(), dtype=np.uint8).copy(), True)]
    for pats, hay, charwise in cases:
        if charwise:
            o = orc.OracleCharwisePma.build(pats)
            p, _ = da.CharwiseDoubleArrayAhoCorasick.deserialize(o.serialize())
            want = o.find_overlapping_iter(hay)
        else:
            o = orc.OraclePma.build(pats)
            p, _ = da.DoubleArrayAhoCorasick.deserialize(o.serialize())
            want = o.find_overlapping_iter(hay)
        expect = (len(want), orc.matches_checksum(want))
        lmax = p.info().max_pattern_len
        need = max(lmax, 3) if charwise else lmax - 1
        n = len(hay)
        dev = torch.from_numpy(hay).cuda()
        for nshards in (1, 2, 8):
            cuts = [0] + sorted(int(x) for x in rng.integers(1, n, size=nshards - 1)) + [n]
            for extra in (0, 37):
                shards_dev, shards_host = [], []
                for k in range(nshards):
                    b, e = cuts[k], cuts[k + 1]
                    halo = min(b, need + extra)
                    shards_dev.append((0, dev[b - halo:e], halo, b))
                    shards_host.append((0, hay[b - halo:e], halo, b))
                assert da.scan_count_multi(p, ScanMode.FindOverlapping, shards_dev) == expect, (len(pats), nshards, extra)
                assert da.scan_count_multi(p, ScanMode.FindOverlapping, shards_dev, checksum=False) == expect[0]
            assert da.scan_count_multi(p, ScanMode.FindOverlapping, shards_host) == expect, (len(pats), nshards, "host")
        if need > 1:
            with pytest.raises(da.DaachorseError) as ei:
                da.scan_count_multi(p, ScanMode.FindOverlapping, [(0, dev[:100], 0, 0), (0, dev[100 - (need - 1):], need - 1, 100)])
            assert ei.value.code == 1
        with pytest.raises(da.DaachorseError) as ei:
            da.scan_count_multi(p, ScanMode.Find, [(0, dev, 0, 0)])
        assert ei.value.code == 6
        with pytest.raises(da.DaachorseError) as ei:   # a halo that reaches in front of the haystack (halo > base): the re-basing would wrap
            da.scan_count_multi(p, ScanMode.FindOverlapping, [(0, dev, 50, 10)])
        assert ei.value.code == 1
    # The HANDLE's options are in scope on every shard's worker thread (round-5 verdict: they were on the first shard only, the others ran with
    # the process-wide values): a handle with its own launch shape for the `.count()` kernel, while the process-wide options say something
    # else — daac_last_kernel() names the kernel and shape that served a shard, as its device's worker ran it.
    o = orc.OraclePma.build(pats3)
    q, _ = da.DoubleArrayAhoCorasick.deserialize(o.serialize())
    q.upload()
    hay = cases[0][1]
    dev = torch.from_numpy(hay).cuda()
    want = o.overlapping_count(hay, threads=8)[0]
    cuts = [0, 700_001, 1_500_000, 2_222_222, len(hay)]
    need = q.info().max_pattern_len - 1
    shards = [(0, dev[b - min(b, need):e], min(b, need), b) for b, e in zip(cuts[:-1], cuts[1:])]
    da.set_option("gram_ppl", 32)
    da.set_option("gram_tail", 0)
    try:
        q.set_option("gram_version", 4).set_option("gram_ppl", 16).set_option("gram_tail", 1).set_option("threads", 512).set_option("gram2_rfull", 0).set_option("gram4_filter", 0)
        assert da.scan_count_multi(q, ScanMode.FindOverlapping, shards, engine=Engine.Gram, checksum=False) == want
        assert da.last_engine() == int(Engine.Gram)
        for k in range(len(shards)):   # shard by shard: each goes through the device's worker
            b, e = cuts[k], cuts[k + 1]
            got = da.scan_count_multi(q, ScanMode.FindOverlapping, [shards[k]], engine=Engine.Gram, checksum=False)
            assert got == q.count(ScanMode.FindOverlapping, dev[:e], engine=Engine.Gram, begin=b) and da.last_engine() == int(Engine.Gram), k
            lk = da.last_kernel()
            assert lk.startswith("gram4 ppl=16 dir=1 waves=8 ") and lk.endswith(" tail=1"), (k, lk)
        for name in ("gram_ppl", "gram_tail", "threads", "gram2_rfull", "gram4_filter"):
            q.set_option(name)
        da.scan_count_multi(q, ScanMode.FindOverlapping, [shards[1]], engine=Engine.Gram, checksum=False)
        lk = da.last_kernel()
        assert lk.startswith("gram4 ppl=32 ") and "waves=16" in lk and lk.endswith(" tail=0"), lk   # (the process-wide values)
    finally:
        da.set_option("gram_ppl", 0)
        da.set_option("gram_tail", -1)
    with pytest.raises(da.DaachorseError) as ei:   # an option that is read at upload, set on a handle that has its tables already: said, not swallowed
        q.set_option("gram_lds_budget", 64 * 1024)
    assert ei.value.code == 6


def test_scan_count_multi_from_several_threads_and_handles():
    """the shard workers of daac_scan_count_multi (one per handle and device, round 6) under concurrent callers: four threads, two handles with
    different launch shapes, every call with its own random cuts — each call's sum is the oracle's, whatever is queued on the workers beside it;
    a handle is freed while the other's worker is busy"""
    import threading
    import torch
    pats = synth.patterns_cfg3(20000)
    o = orc.OraclePma.build(pats)
    hay = synth.wordsoup_haystack(5 << 20, synth.SEEDS["cfg3_dense"] + 3, pats, 20)
    dev = torch.from_numpy(hay).cuda()
    want = o.overlapping_count(hay, threads=8)
    a, _ = da.DoubleArrayAhoCorasick.deserialize(o.serialize())
    b, _ = da.DoubleArrayAhoCorasick.deserialize(o.serialize())
    a.set_option("gram_ppl", 16).set_option("gram4_filter", 0)
    b.set_option("gram_tail", 0).set_option("threads", 512)
    need = a.info().max_pattern_len - 1
    errs = []

    def work(p, seed):
        rng = np.random.default_rng(seed)
        try:
            for it in range(8):
                k = int(rng.integers(1, 9))
                cuts = [0] + sorted(int(x) for x in rng.integers(1, len(hay), size=k - 1)) + [len(hay)]
                shards = [(0, dev[lo - min(lo, need):hi], min(lo, need), lo) for lo, hi in zip(cuts[:-1], cuts[1:])]
                if it % 2:
                    assert da.scan_count_multi(p, ScanMode.FindOverlapping, shards) == want, (seed, it, cuts)
                else:
                    assert da.scan_count_multi(p, ScanMode.FindOverlapping, shards, checksum=False) == want[0], (seed, it, cuts)
        except Exception as e:  # noqa
            errs.append(e)
    ts = [threading.Thread(target=work, args=(p, s)) for p, s in ((a, 1), (b, 2), (a, 3), (b, 4))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    t = threading.Thread(target=work, args=(b, 5))
    t.start()
    del a          # joins a's worker while b's is scanning
    t.join()
    assert not errs, errs


def test_trim_and_iterators_that_are_never_pulled():
    """daac_pma_trim gives the kept scratch back and the next scans allocate again; a lazy iterator that is opened and closed without a
    next() never starts its worker (the Rust cursor's `.count()` opens one and counts beside it), one that is pulled after a trim works"""
    import torch
    pats = synth.patterns_cfg3(20000)
    o = orc.OraclePma.build(pats)
    p, _ = da.DoubleArrayAhoCorasick.deserialize(o.serialize())
    hay = synth.wordsoup_haystack(4 << 20, synth.SEEDS["cfg3_dense"], pats, 20)
    dev = torch.from_numpy(hay).cuda()
    want = o.find_overlapping_iter(hay)
    for _ in range(2):
        dm = p.scan_device(ScanMode.FindOverlapping, dev)
        assert dm.count == len(want)
        dm.free()
        free_before = torch.cuda.mem_get_info()[0]
        p.trim()
        assert torch.cuda.mem_get_info()[0] >= free_before
        for _ in range(3):
            it = p.find_overlapping_iter(hay)
            it.close()
        assert p.count(ScanMode.FindOverlapping, dev) == len(want)
        it = p.find_overlapping_iter(hay)
        first = next(iter(it))
        assert (first.start(), first.end(), first.value()) == (int(want["start"][0]), int(want["end"][0]), int(want["value"][0]))
        it.close()


def test_options_per_handle():
    """daac_pma_set_option: two handles of one dictionary with different settings — the engine version, the launch shape, the body of the
    count kernel, the emitter switched off — scanned from two threads at once give the oracle's answers, the process-wide options untouched
    (a third handle follows them); an unknown name and the allocator's options are refused."""
    import threading
    import torch
    pats = synth.patterns_cfg3(20000)
    o = orc.OraclePma.build(pats)
    hay = synth.wordsoup_haystack(6 << 20, synth.SEEDS["cfg3_dense"], pats, 20)
    dev = torch.from_numpy(hay).cuda()
    want = o.overlapping_count(hay, threads=8)
    a, _ = da.DoubleArrayAhoCorasick.deserialize(o.serialize())
    b, _ = da.DoubleArrayAhoCorasick.deserialize(o.serialize())
    c, _ = da.DoubleArrayAhoCorasick.deserialize(o.serialize())
    a.set_option("gram_ppl", 16).set_option("gram_tail", 0).set_option("gram4_arith", 0).set_option("emit", 0)
    b.set_option("gram_ppl", 32).set_option("gram_tail", 1).set_option("gram2_rfull", 0).set_option("threads", 512)
    errs = []

    def work(p, n):
        try:
            for _ in range(n):
                assert p.count(ScanMode.FindOverlapping, dev) == want[0]
                assert p.scan_count(ScanMode.FindOverlapping, dev) == want
        except Exception as e:  # noqa
            errs.append(e)
    ts = [threading.Thread(target=work, args=(p, 6)) for p in (a, b, c)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    m = 200_000
    ref = o.find_overlapping_iter(hay[:m])
    for p, by_emitter in ((a, False), (b, True), (c, True)):   # a's emitter is off: the segment scanners write its list
        dm = p.scan_device(ScanMode.FindOverlapping, dev[:m])
        got = dm.to_numpy()
        dm.free()
        assert (da.last_engine() == int(Engine.Gram)) == by_emitter, (da.last_engine(), by_emitter)
        assert len(got) == len(ref) and np.array_equal(got["end"], ref["end"]) and np.array_equal(got["value"], ref["value"])
    a.set_option("emit")   # the override goes: the process-wide value (1) again
    dm = a.scan_device(ScanMode.FindOverlapping, dev[:m])
    dm.free()
    assert da.last_engine() == int(Engine.Gram)
    for name in ("no_such_option", "pool"):
        with pytest.raises(da.DaachorseError) as ei:
            a.set_option(name, 1)
        assert ei.value.code == 1

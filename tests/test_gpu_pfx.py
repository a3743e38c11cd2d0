"""PFX engine (pfx_kernels.hip: hashed prefix filter + start-anchored goto-only walks; `.count()` of find_overlapping for
dictionaries over ANY byte alphabet) against the oracle: a 256-byte-alphabet dictionary, a UTF-8 Japanese-like one scanned
bytewise, one-byte patterns, duplicates, every key length G = 2 .. 6, unaligned and ragged haystacks, shards."""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

import daachorse_amd as da
from daachorse_amd import Engine, ScanMode, synth


def _pma(patterns, pfx=2):
    """pfx = 2 (this handle's own setting, daac_pma_set_option; read at upload): the PFX tables are built whatever else serves the automaton
    (default: only where GRAM does not apply)"""
    o = orc.OraclePma.build(patterns)
    p, rest = da.DoubleArrayAhoCorasick.deserialize(o.serialize())
    assert rest == b""
    if pfx is not None:
        p.set_option("pfx", pfx)
    return o, p


def _count(p, hay, **kw):
    got = p.count(ScanMode.FindOverlapping, hay, engine=Engine.Pfx, **kw)
    assert da.last_engine() == int(Engine.Pfx)
    return got


def _count_checksum(p, hay, **kw):
    got = p.scan_count(ScanMode.FindOverlapping, hay, engine=Engine.Pfx, **kw)
    assert da.last_engine() == int(Engine.Pfx)
    return got


def _add(a, b):
    """(count, checksum) of two shards: the checksum is two sums mod 2^32 side by side"""
    m = 0xffffffff
    return a[0] + b[0], ((((a[1] >> 32) + (b[1] >> 32)) & m) << 32) | ((a[1] + b[1]) & m)


def test_pfx_against_the_oracle():
    import torch
    rng = np.random.default_rng(2024)
    binp = synth.patterns_binary256(20000)
    jp = synth.patterns_cfg5(5000)
    mixed = list(dict.fromkeys([bytes(rng.integers(0, 256, size=int(rng.integers(1, 9))).astype(np.uint8)) for _ in range(3000)]))
    by_len = {g: [bytes(rng.integers(0, 256, size=int(rng.integers(g, g + 6))).astype(np.uint8)) for _ in range(4000)] for g in (2, 3, 4, 5, 6, 9)}
    cases = [(binp, rng.integers(0, 256, size=3 << 20).astype(np.uint8)),
             (binp, np.frombuffer(b"".join(binp[i] for i in rng.integers(0, len(binp), size=200000).tolist()), dtype=np.uint8)),
             (jp, synth.zipf_text(48 * 40000)),
             (mixed, rng.integers(0, 256, size=1 << 20).astype(np.uint8)),
             ([b"ab", b"ab", b"b", b"abab", b"bababab", b"\xff\x00", b"\xff\x00"], np.frombuffer(b"abababbab\xff\x00" * 30000, dtype=np.uint8)),
             (synth.patterns_cfg3(20000), synth.wordsoup_haystack(1 << 20, 5, synth.patterns_cfg3(20000), 20))]
    for g, pats in by_len.items():
        pats = list(dict.fromkeys(pats))
        soup = b"".join(pats[i] if rng.integers(0, 3) else bytes(rng.integers(0, 256, size=5).astype(np.uint8)) for i in rng.integers(0, len(pats), size=60000).tolist())
        cases.append((pats, np.frombuffer(soup, dtype=np.uint8)))
    for pats, hay in cases:
        o, p = _pma(pats)
        p.upload()
        both = o.overlapping_count(hay, threads=8)
        want = both[0]
        dev = torch.from_numpy(np.concatenate([np.zeros(7, dtype=np.uint8), hay])).cuda()[7:]  # not 16-byte aligned
        assert _count(p, dev) == want, (len(pats), len(hay))
        assert _count_checksum(p, dev) == both, (len(pats), len(hay))
        cut = int(rng.integers(1, len(hay)))
        assert _count(p, dev[:cut]) + _count(p, dev, begin=cut) == want, cut
        assert _add(_count_checksum(p, dev[:cut]), _count_checksum(p, dev, begin=cut)) == both, cut
        p.set_option("gram_region", 2048)
        assert _count(p, dev) == want
        assert _count_checksum(p, dev) == both


def test_pfx_short_and_ragged_haystacks():
    import torch
    rng = np.random.default_rng(77)
    pats = list(dict.fromkeys([bytes(rng.integers(0, 8, size=int(rng.integers(1, 7))).astype(np.uint8)) for _ in range(300)]))  # dense: 8 byte values, one-byte patterns too
    o, p = _pma(pats)
    p.upload()
    base = rng.integers(0, 8, size=1 << 18).astype(np.uint8)
    buf = torch.from_numpy(base).cuda()
    p.set_option("gram_region", 2048)
    lengths = [0, 1, 2, 3, 4, 5, 6, 7, 15, 16, 17, 63, 64, 1023, 1024, 1025, 2047, 2048, 2049, 4095, 4096, 4100, 65535, 65536, 65537]
    lengths += [int(x) for x in rng.integers(1, 1 << 17, size=12)]
    for n in lengths:
        off = int(rng.integers(0, 32))
        both = o.overlapping_count(base[off:off + n], threads=1) if n else (0, 0)
        assert _count(p, buf[off:off + n]) == both[0], (n, off)
        assert _count_checksum(p, buf[off:off + n]) == both, (n, off)


def test_pfx_is_what_auto_takes_for_wide_alphabets():
    """256 pattern bytes: no GRAM table set applies; `.count()` and count + checksum run on PFX"""
    import torch
    pats = synth.patterns_binary256(30000)
    o, p = _pma(pats, pfx=None)
    info = p.upload().info()
    assert not info.gram_available and not info.gram2_available
    dev = torch.empty(32 << 20, dtype=torch.uint8, device="cuda")
    synth.device_uniform(dev, synth.SEEDS["bin_hay"], synth.ALPHA_BYTES)
    want = o.overlapping_count(dev.cpu().numpy(), threads=16)
    assert p.count(ScanMode.FindOverlapping, dev) == want[0] and da.last_engine() == int(Engine.Pfx)
    assert p.scan_count(ScanMode.FindOverlapping, dev) == want and da.last_engine() == int(Engine.Pfx)
    # the UTF-8 dictionary of cfg5 scanned bytewise
    jp = synth.patterns_cfg5(20000)
    o, p = _pma(jp)
    p.upload()
    n = (32 << 20) - (32 << 20) % synth.CFG5_SLOT
    dev = torch.empty(n, dtype=torch.uint8, device="cuda")
    synth.device_zipf_text(dev)
    want = o.overlapping_count(dev.cpu().numpy(), threads=16)
    assert p.count(ScanMode.FindOverlapping, dev) == want[0] and da.last_engine() == int(Engine.Pfx)
    assert p.scan_count(ScanMode.FindOverlapping, dev) == want and da.last_engine() == int(Engine.Pfx)


def test_pfx_on_a_vector_of_window_counts():
    """one integer per call hides compensating errors: 1 024 windows (begin, len) of random sizes and alignments over 16 MiB of each of
    the two wide-alphabet workloads, each counted on its own and compared with the oracle's matches ending in (begin, len]"""
    import torch
    rng = np.random.default_rng(5)
    n = 16 << 20
    for name in ("binary256", "utf8jp"):
        if name == "binary256":
            pats = synth.patterns_binary256(50000)
            # a third of the text is made of patterns so that windows hold matches
            soup = b"".join(pats[i] if k % 3 == 0 else bytes(rng.integers(0, 256, size=24).astype(np.uint8)) for k, i in enumerate(rng.integers(0, len(pats), size=n // 12).tolist()))
            hay = np.frombuffer(soup[:n], dtype=np.uint8)
        else:
            pats = synth.patterns_cfg5(20000)
            hay = synth.zipf_text(n - n % synth.CFG5_SLOT)
        o, p = _pma(pats)
        p.upload()
        dev = torch.from_numpy(hay.copy()).cuda()
        ends = np.sort(o.find_overlapping_iter(hay)["end"].astype(np.int64))
        assert len(ends) > 100000
        los = rng.integers(0, len(hay) - (1 << 16), size=1024)
        sizes = np.concatenate([rng.integers(1, 64, size=256), rng.integers(64, 4096, size=384), rng.integers(4096, 1 << 16, size=384)])
        his = los + sizes
        want = np.searchsorted(ends, his, side="right") - np.searchsorted(ends, los, side="right")
        got = np.array([_count(p, dev[:int(h)], begin=int(l)) for l, h in zip(los, his)])
        bad = np.nonzero(got != want)[0]
        assert len(bad) == 0, (name, bad[:8], got[bad[:8]], want[bad[:8]], los[bad[:8]], his[bad[:8]])


def test_engine_plan_says_what_will_run():
    """daac_info.plan_* / daac_pma_explain: the engine a request gets is known before the scan, and it is the engine that then serves it"""
    import torch
    REQ = {"count": 0, "checksum": 1, "tuples": 2, "find": 3, "leftmost": 4, "nosuffix": 5}
    hay = torch.from_numpy(synth.uniform_haystack(1 << 20, 3, synth.ALPHA_LOWER_SPACE)).cuda()
    for pats, kind, expect in ((synth.patterns_cfg3(5000), 0, {"count": (Engine.Gram, 1), "checksum": (Engine.Gram, 2), "tuples": (Engine.Gram, 4), "find": (Engine.Gram, 9)}),
                               (synth.patterns_binary256(5000), 0, {"count": (Engine.Pfx, 5), "checksum": (Engine.Pfx, 5)}),
                               (synth.patterns_cfg3(2000) + [b""], 0, {})):
        o = orc.OraclePma.build(pats)
        p, _ = da.DoubleArrayAhoCorasick.deserialize(o.serialize())
        assert list(p.info().plan_reason)[:6] == [1] * 6  # not uploaded yet
        info = p.upload().info()
        assert info.struct_size > 100
        text = p.explain()
        assert len(text.splitlines()) == 6 and "leftmost_find_iter: - (the crate panics" in text
        for req, (eng, kernel) in expect.items():
            assert info.plan_engine[REQ[req]] == int(eng) and info.plan_kernel[REQ[req]] == kernel, (req, text)
        # what the plan says is what runs
        p.count(ScanMode.FindOverlapping, hay)
        assert da.last_engine() == info.plan_engine[0], text
        p.scan_count(ScanMode.FindOverlapping, hay)
        assert da.last_engine() == info.plan_engine[1], text
        p.scan_device(ScanMode.FindOverlapping, hay[:1 << 16]).free()
        assert da.last_engine() == info.plan_engine[2], text
        p.scan_count(ScanMode.Find, hay[:1 << 16])
        assert da.last_engine() == info.plan_engine[3], text
    o = orc.OraclePma.build(synth.patterns_cfg3(2000), kind="LeftmostLongest")
    p, _ = da.DoubleArrayAhoCorasick.deserialize(o.serialize())
    info = p.upload().info()
    assert info.plan_kernel[4] == 9 and info.plan_kernel[0] == 0  # left3's selection (the handle's patterns as a Standard automaton); find_overlapping does not apply to the kind
    p.scan_count(ScanMode.LeftmostFind, hay[:1 << 16])
    assert da.last_engine() == info.plan_engine[4] == int(Engine.Gram)
    q, _ = da.DoubleArrayAhoCorasick.deserialize(o.serialize())
    q.set_option("left3", 0)
    info = q.upload().info()
    assert info.plan_kernel[4] == 8   # chain walkers
    q.scan_count(ScanMode.LeftmostFind, hay[:1 << 16])
    assert da.last_engine() == info.plan_engine[4]


def test_round3_engines_take_host_haystacks_and_streams():
    """the same answers from a host buffer (staged by the library), from a device tensor, and on a caller's stream"""
    import torch
    rng = np.random.default_rng(9)
    for pats, hay in ((synth.patterns_binary256(5000), rng.integers(0, 256, size=(1 << 20) + 3).astype(np.uint8)),
                      (synth.patterns_cfg3(5000), synth.wordsoup_haystack((1 << 20) + 5, 3, synth.patterns_cfg3(5000), 20))):
        o, p = _pma(pats)
        p.upload()
        want = o.overlapping_count(hay, threads=4)
        dev = torch.from_numpy(hay).cuda()
        s = torch.cuda.Stream()
        for h in (hay, dev):
            assert p.count(ScanMode.FindOverlapping, h) == want[0]
            assert p.scan_count(ScanMode.FindOverlapping, h) == want
            assert p.count(ScanMode.FindOverlapping, h, stream=s.cuda_stream) == want[0]
            cut = 333_333
            assert p.count(ScanMode.FindOverlapping, h[:cut]) + p.count(ScanMode.FindOverlapping, h, begin=cut) == want[0]
        ref = o.find_overlapping_iter(hay[:200_000])
        for h in (hay[:200_000], dev[:200_000]):
            dm = p.scan_device(ScanMode.FindOverlapping, h, fmt16=True)
            got = dm.to_numpy()
            dm.free()
            assert np.array_equal(got["end"], ref["end"]) and np.array_equal(got["value"], ref["value"]) and np.array_equal(got["length"], ref["end"] - ref["start"])


def test_engines_that_cannot_serve_a_request_say_so():
    """An engine whose tables a handle does not have (PFX where the GRAM tables serve), or a number that is no engine at all, asked for
    tuples / a lazy iterator / a stepper: status 6 — not a silent scan on the double array with last_engine() = DARRAY (round-3 advisor).
    With its tables built PFX serves tuples and the lazy iterator (round 4); steppers it does not."""
    o = orc.OraclePma.build([b"ab", b"bc", b"abc"])
    hay = np.frombuffer(b"xxabcabyy" * 50, dtype=np.uint8)
    want = o.find_overlapping_iter(hay)
    p, _ = da.DoubleArrayAhoCorasick.deserialize(o.serialize())
    p.set_option("pfx", 2)   # PFX tables built (read at upload)
    assert _same(p.scan(ScanMode.FindOverlapping, hay, engine=Engine.Pfx), want)
    assert da.last_engine() == int(Engine.Pfx)
    assert [(m.start(), m.end(), m.value()) for m in p.find_overlapping_iter(hay, engine=Engine.Pfx)] == orc.triples_sev(want)
    with pytest.raises(da.DaachorseError) as ei:
        p.find_overlapping_stepper(engine=Engine.Pfx)
    assert ei.value.code == 6
    p, _ = da.DoubleArrayAhoCorasick.deserialize(o.serialize())   # (default: no PFX tables where GRAM serves)
    got = p.scan(ScanMode.FindOverlapping, hay)
    assert len(got) == len(want)
    for eng in (int(Engine.Pfx), 77):
        for call in (lambda: p.scan(ScanMode.FindOverlapping, hay, engine=eng),
                     lambda: p.scan_device(ScanMode.FindOverlapping, hay, engine=eng),
                     lambda: list(p.find_overlapping_iter(hay, engine=eng)),
                     lambda: p.find_overlapping_stepper(engine=eng)):
            with pytest.raises(da.DaachorseError) as ei:
                call()
            assert ei.value.code == 6, eng


def _same(a, b):
    return len(a) == len(b) and np.array_equal(a["start"], b["start"]) and np.array_equal(a["end"], b["end"]) and \
        np.array_equal(a["value"], b["value"])


def _tuples16(p, dev, **kw):
    dm = p.scan_device(ScanMode.FindOverlapping, dev, engine=Engine.Pfx, fmt16=True, **kw)
    assert da.last_engine() == int(Engine.Pfx)
    t = dm.to_numpy()
    dm.free()
    return t


def _same16(t, want):
    return len(t) == len(want) and np.array_equal(t["end"], want["end"]) and np.array_equal(t["length"], (want["end"] - want["start"]).astype(np.uint32)) and \
        np.array_equal(t["value"], want["value"])


def test_pfx_tuples_against_the_oracle():
    """The find_overlapping tuples of dictionaries over ANY byte alphabet through pfx_emit_kernel + EXPAND over the raw haystack
    (bytewise/iter.rs:133-176: by end, longest first), both device formats: a 256-byte-alphabet dictionary, UTF-8 Japanese scanned
    bytewise, one-byte patterns (all 256 of them in one case), patterns beyond the 17 bytes a tile's length bits carry (extras),
    several patterns ending on one byte, every key length G, unaligned haystacks, a shard, text shorter than a key."""
    import torch
    rng = np.random.default_rng(77)
    binp = synth.patterns_binary256(20000)
    jp = synth.patterns_cfg5(5000)
    mixed = list(dict.fromkeys([bytes(rng.integers(0, 256, size=int(rng.integers(1, 9))).astype(np.uint8)) for _ in range(3000)]))
    nested = [b"a", b"ab", b"abc", b"bc", b"c", b"abcabcabcabcabcabcabc", b"cabcabcabcabcabcabcabca", b"bcabcabcabcabcabcabcabcabcabc", b"\xff\x00", b"\x00\xff\x00\xff"]
    all1 = [bytes([b]) for b in range(256)] + [b"th", b"the", b"he", b" the ", b"\xe3\x81\x82", b"\xe3\x81"]
    by_len = {g: list(dict.fromkeys(bytes(rng.integers(0, 256, size=int(rng.integers(g, g + 6))).astype(np.uint8)) for _ in range(4000))) for g in (2, 3, 4, 5, 6, 9)}
    cases = [(binp, rng.integers(0, 256, size=1 << 20).astype(np.uint8)),
             (binp, np.frombuffer(b"".join(binp[i] for i in rng.integers(0, len(binp), size=100000).tolist()), dtype=np.uint8)),
             (jp, synth.zipf_text(48 * 20000)),
             (mixed, rng.integers(0, 256, size=1 << 19).astype(np.uint8)),
             (nested, np.frombuffer(((b"abcabcabcabcabcabcabcabcabcabcab\xff\x00\xff\x00xx" + b"q" * 300 + b"abcab" * 9) * 1500), dtype=np.uint8)),
             (all1, np.frombuffer(b"the cat and the \xe3\x81\x82 other " * 3000, dtype=np.uint8)),
             (all1, rng.integers(0, 256, size=70000).astype(np.uint8))]
    for g, pats in by_len.items():
        soup = b"".join(pats[i] if rng.integers(0, 3) else bytes(rng.integers(0, 256, size=5).astype(np.uint8)) for i in rng.integers(0, len(pats), size=30000).tolist())
        cases.append((pats, np.frombuffer(soup, dtype=np.uint8)))
    for pats, hay in cases:
        o, p = _pma(pats)
        p.upload()
        want = o.find_overlapping_iter(hay)
        dev = torch.from_numpy(np.concatenate([np.zeros(5, dtype=np.uint8), hay])).cuda()[5:]  # not 16-byte aligned
        got = p.scan(ScanMode.FindOverlapping, dev, engine=Engine.Pfx)
        assert da.last_engine() == int(Engine.Pfx)
        assert _same(got, want), (len(pats), len(hay), len(got), len(want))
        assert _same16(_tuples16(p, dev), want), (len(pats), len(hay))
        # a prefix, and the automatic engine choice on a dictionary no GRAM table serves
        cut = int(rng.integers(1, min(len(hay), 5000)))
        assert _same(p.scan(ScanMode.FindOverlapping, dev[:cut], engine=Engine.Pfx), o.find_overlapping_iter(hay[:cut])), cut
        p.set_option("gram_region", 2048)
        assert _same(p.scan(ScanMode.FindOverlapping, dev, engine=Engine.Pfx), want)
        p.set_option("gram_region")
        # the lazy iterator over the same bytes (windows with begin > 0)
        p.set_option("iter_window", 1 << 16)
        it = p.find_overlapping_iter(hay[:200000], engine=Engine.Pfx)
        runs = []
        while True:
            r = it.next_batch()
            if r is None:
                break
            runs.append(r.copy())
        it.close()
        got16 = np.concatenate(runs) if runs else np.zeros(0, dtype=bytewise_match16())
        assert _same16(got16, o.find_overlapping_iter(hay[:200000])), len(pats)
    # text shorter than a key, an empty text
    o, p = _pma([b"abcd", b"abcde", b"x"])
    for text in (b"", b"ab", b"x", b"abc", b"abcd", b"xabcdex"):
        h = np.frombuffer(text, dtype=np.uint8)
        assert _same(p.scan(ScanMode.FindOverlapping, h, engine=Engine.Pfx), o.find_overlapping_iter(h)), text
    # more long matches (beyond the 17 bytes a position's length bits carry) in one tile than EXPAND places: PFX says so, the default engine answers
    o, p = _pma(nested)
    h = np.frombuffer(b"abcabcabcabcabcabcabcabcabcabcab" * 3000, dtype=np.uint8)
    with pytest.raises(da.DaachorseError):
        p.scan(ScanMode.FindOverlapping, h, engine=Engine.Pfx)
    assert _same(p.scan(ScanMode.FindOverlapping, h), o.find_overlapping_iter(h))
    # a pattern registered twice: no PFX tuples (the engine says so), the default engine still answers
    o, p = _pma([b"ab\xff", b"ab\xff", b"b\xffq"])
    h = np.frombuffer(b"xab\xffqab\xff" * 100, dtype=np.uint8)
    with pytest.raises(da.DaachorseError):
        p.scan(ScanMode.FindOverlapping, h, engine=Engine.Pfx)
    assert _same(p.scan(ScanMode.FindOverlapping, h), o.find_overlapping_iter(h))


def bytewise_match16():
    from daachorse_amd.bytewise import MATCH16_DTYPE
    return MATCH16_DTYPE


def test_wide_dictionary_look_alikes():
    """Look-alikes of the two wide dictionaries of the crate's own benchmark (figures/overlapping.txt:1-5): Unidic-like (675 000 UTF-8
    patterns) and o200k-like (200 000 byte-level tokens, all 256 one-byte patterns among them), built by the PRODUCT's builder: count,
    count + checksum and tuples against the oracle on 8 MiB of their text."""
    import torch
    for name in ("unidic_like", "o200k_like"):
        if name == "unidic_like":
            pats = synth.patterns_unidic_like()
            n = (8 << 20) - (8 << 20) % synth.CFG5_SLOT
            dev = torch.empty(n, dtype=torch.uint8, device="cuda")
            synth.device_zipf_text(dev)
        else:
            pats = synth.patterns_o200k_like()
            n = 8 << 20
            dev = torch.empty(n, dtype=torch.uint8, device="cuda")
            synth.device_wordsoup(dev, synth.SEEDS["o200k_hay"], synth.o200k_soup_words(), 17)
        p = da.DoubleArrayAhoCorasick.new(pats)
        o = orc.OraclePma.deserialize(p.serialize())
        host = dev.cpu().numpy()
        want = o.find_overlapping_iter(host)
        assert len(want) > n // 8, name
        # (`.count()`: PFX, or the micro-step walker where most positions of the text survive PFX's filter — both against the oracle)
        for eng in (Engine.Auto, Engine.Pfx, Engine.DArray):
            assert p.count(ScanMode.FindOverlapping, dev, engine=eng) == len(want), (name, eng)
            assert p.scan_count(ScanMode.FindOverlapping, dev, engine=eng) == (len(want), orc.matches_checksum(want)), (name, eng)
        got = p.scan(ScanMode.FindOverlapping, dev)
        assert da.last_engine() == int(Engine.Pfx), name
        assert _same(got, want), name
        dm = p.scan_device(ScanMode.FindOverlapping, dev, fmt16=True)
        assert _same16(dm.to_numpy(), want), name
        dm.free()


def test_the_probe_sends_dense_text_to_the_walker():
    """AUTO on a dictionary PFX serves: a synchronous `.count()` of 32 MiB or more samples the text; where more than a quarter of the
    positions survive the filter (every character of the text is a pattern) the micro-step walker over the double array takes the scan, on
    text the filter thins out PFX does — the counts agree with each other and with a shard sum either way."""
    import torch
    pats = synth.patterns_unidic_like(60_000)
    p = da.DoubleArrayAhoCorasick.new(pats)
    n = (48 << 20) - (48 << 20) % synth.CFG5_SLOT
    dev = torch.empty(n, dtype=torch.uint8, device="cuda")
    synth.device_zipf_text(dev)
    want = p.count(ScanMode.FindOverlapping, dev, engine=Engine.Pfx)
    assert p.count(ScanMode.FindOverlapping, dev) == want
    assert da.last_engine() == int(Engine.DArray)
    res = torch.zeros(3, dtype=torch.int64, device="cuda")
    p.count(ScanMode.FindOverlapping, dev, result_dev=res.data_ptr())      # asynchronous: goes by the verdict in the handle
    assert int(res[0].item()) == want and da.last_engine() == int(Engine.DArray)
    synth.device_uniform(dev, synth.SEEDS["bin_hay"], synth.ALPHA_BYTES)  # random bytes: next to nothing survives
    want = p.count(ScanMode.FindOverlapping, dev, engine=Engine.DArray)
    assert p.count(ScanMode.FindOverlapping, dev) == want
    assert da.last_engine() == int(Engine.Pfx)
    p.set_option("pfx_probe", 0)
    synth.device_zipf_text(dev)
    p.count(ScanMode.FindOverlapping, dev)
    assert da.last_engine() == int(Engine.Pfx)


def test_pfx_tuples_piece_by_piece():
    """daac_scan_device16 of a range beyond 256 MiB on the PFX engine goes piece by piece (every tuple is a record first: the scratch of a
    whole GiB of match-dense text would be tens of GB per call); the pieces' lists, put together, are the oracle's list — matches that
    straddle a piece boundary included."""
    import torch
    pats = synth.patterns_cfg5(20_000)
    p = da.DoubleArrayAhoCorasick.new(pats)
    o = orc.OraclePma.deserialize(p.serialize())
    n = (300 << 20) - (300 << 20) % synth.CFG5_SLOT
    dev = torch.empty(n, dtype=torch.uint8, device="cuda")
    synth.device_zipf_text(dev)
    host = dev.cpu().numpy()
    want = o.find_overlapping_iter(host)
    assert len(want) > 5_000_000
    for rep in range(2):   # (the first scan of a handle counts as match-dense; the second goes by what the first met)
        dm = p.scan_device(ScanMode.FindOverlapping, dev, fmt16=True)
        assert da.last_engine() == int(Engine.Pfx)
        got = dm.to_numpy()
        dm.free()
        assert _same16(got, want), rep
    # the same text cut so that a match straddles the first piece boundary (256 MiB), on a fresh handle
    cut = 256 << 20
    j = int(np.searchsorted(want["start"], cut))
    shift = int(want["start"][j]) + 1 - cut
    assert 0 < shift < (1 << 20)
    want2 = o.find_overlapping_iter(host[shift:])
    assert np.any((want2["start"] < cut) & (want2["end"] > cut))
    q = da.DoubleArrayAhoCorasick.new(pats)
    dm = q.scan_device(ScanMode.FindOverlapping, dev[shift:])
    assert da.last_engine() == int(Engine.Pfx)
    assert _same(dm.to_numpy(), want2)
    dm.free()

"""BASELINE.json's configurations as SURVEY.md 8d specifies them, on the MI355X against the CPU oracle:
cfg1 (3 patterns, 1 KiB + "abcd"), cfg3 (100k words: tuples on the first 16 MiB and on 64 random 1 MiB windows of the
4 GiB haystack), cfg5 (charwise LeftmostLongest, 50k patterns over the 6 000-symbol Zipf alphabet, text with 10 % ASCII
generated on the device by index).  cfg2 lives in test_gpu_parity.py; cfg4 needs 8 GPUs (its shard logic is covered by
test_shard_tail_counts_add_up / test_dist_gloo)."""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

import daachorse_amd as da
from daachorse_amd import Engine, ScanMode, synth


def _same(a, b):
    return len(a) == len(b) and np.array_equal(a["start"], b["start"]) and np.array_equal(a["end"], b["end"]) and \
        np.array_equal(a["value"], b["value"])


def test_cfg1_three_patterns():
    pats = synth.patterns_cfg1()
    o = orc.OraclePma.build(pats)
    p = da.DoubleArrayAhoCorasick.new(pats)
    assert p.serialize() == o.serialize()
    got = [(m.start(), m.end(), m.value()) for m in p.find_overlapping_iter(b"abcd")]
    assert got == [(0, 1, 2), (0, 2, 1), (1, 4, 0)]  # README.md:57-71
    hay = synth.uniform_haystack(1024, synth.SEEDS["cfg1_hay"], synth.ALPHA_ABCD)
    want = o.find_overlapping_iter(hay)
    for eng in (Engine.Tiered, Engine.DArray, Engine.Auto):
        assert _same(p.scan(ScanMode.FindOverlapping, hay, engine=eng), want), eng
    for eng in (Engine.Tiered, Engine.DArray, Engine.Gram, Engine.Auto):
        assert p.scan_count(ScanMode.FindOverlapping, hay, engine=eng) == (len(want), orc.matches_checksum(want)), eng
    assert [(m.start(), m.end(), m.value()) for m in p.find_overlapping_iter(hay)] == \
        [(int(x["start"]), int(x["end"]), int(x["value"])) for x in want]


def test_cfg3_tuples_16mib_and_64_windows():
    """full tuple equality on the first 16 MiB and on 64 random 1 MiB windows of the 4 GiB haystack
    (each window generated on the device at its offset of the stream and scanned as a haystack of its own)"""
    import torch
    pats = synth.patterns_cfg3()
    o = orc.OraclePma.build(pats)
    p, _ = da.DoubleArrayAhoCorasick.deserialize(o.serialize())
    n = 16 << 20
    dev = torch.empty(n, dtype=torch.uint8, device="cuda")
    synth.device_uniform(dev, synth.SEEDS["cfg3_hay"], synth.ALPHA_LOWER_SPACE)
    host = dev.cpu().numpy()
    assert np.array_equal(host[:4096], synth.uniform_haystack(4096, synth.SEEDS["cfg3_hay"], synth.ALPHA_LOWER_SPACE))
    want = o.find_overlapping_iter(host)
    assert len(want) > 9_000_000
    for eng in (Engine.Auto, Engine.Tiered, Engine.DArray):
        assert _same(p.scan(ScanMode.FindOverlapping, dev, engine=eng), want), eng
    rng = np.random.default_rng(0xDAAC0013)
    win = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
    for off in rng.integers(0, (4 << 30) - (1 << 20), size=64).tolist():
        synth.device_uniform(win, synth.SEEDS["cfg3_hay"], synth.ALPHA_LOWER_SPACE, offset=off)
        w = win.cpu().numpy()
        want = o.find_overlapping_iter(w)
        assert _same(p.scan(ScanMode.FindOverlapping, win), want), off
        assert p.scan_count(ScanMode.FindOverlapping, win) == (len(want), orc.matches_checksum(want)), off
    # the dense haystack (word soup) too: first 4 MiB
    synth.device_wordsoup(dev[:4 << 20], synth.SEEDS["cfg3_dense"], pats, 20)
    want = o.find_overlapping_iter(dev[:4 << 20].cpu().numpy())
    assert _same(p.scan(ScanMode.FindOverlapping, dev[:4 << 20]), want)


def test_cfg5_charwise_leftmost_longest():
    """BASELINE configs[4] as SURVEY 8d states it: 50 000 patterns of 2-8 scalars, Zipf(1.0) over 6 000 symbols,
    LeftmostLongest; text = i.i.d. scalars + 10 % ASCII, generated in HBM by index.  Tuples on 8 MiB, count + checksum
    on the configuration's full 1 GiB, and the other three charwise iterators on 2 MiB."""
    import torch
    pats = synth.patterns_cfg5()
    assert len(pats) == 50_000
    o = orc.OracleCharwisePma.build(pats, kind=1)
    p, _ = da.CharwiseDoubleArrayAhoCorasick.deserialize(o.serialize())
    n = (1 << 30) - (1 << 30) % synth.CFG5_SLOT
    dev = torch.empty(n, dtype=torch.uint8, device="cuda")
    synth.device_zipf_text(dev)
    small = dev[:(8 << 20) - (8 << 20) % synth.CFG5_SLOT]
    host = small.cpu().numpy()
    assert np.array_equal(host[:48 * 5000], synth.zipf_text(48 * 5000))       # device generator == numpy definition
    off = 48 * 1234567 + 17
    assert np.array_equal(dev[off:off + 4099].cpu().numpy(), synth.zipf_text(4099, offset=off))
    want = o.leftmost_find_iter(host)
    assert len(want) > 300_000
    assert _same(p.scan(ScanMode.LeftmostFind, small), want)
    assert p.scan_count(ScanMode.LeftmostFind, small) == (len(want), orc.matches_checksum(want))
    big = o.leftmost_find_iter(dev.cpu().numpy())
    assert len(big) > 40_000_000
    want_big = (len(big), orc.matches_checksum(big))
    del big
    assert p.scan_count(ScanMode.LeftmostFind, dev) == want_big
    assert p.count(ScanMode.LeftmostFind, dev) == want_big[0]
    # the Standard-kind iterators of the same dictionary
    o0 = orc.OracleCharwisePma.build(pats, kind=0)
    p0, _ = da.CharwiseDoubleArrayAhoCorasick.deserialize(o0.serialize())
    two = dev[:48 * 43690]
    h2 = two.cpu().numpy()
    for api, mode in (("find_overlapping_iter", ScanMode.FindOverlapping), ("find_iter", ScanMode.Find),
                      ("find_overlapping_no_suffix_iter", ScanMode.FindOverlappingNoSuffix)):
        w = getattr(o0, api)(h2)
        assert _same(p0.scan(mode, two), w), api
        assert p0.scan_count(mode, two) == (len(w), orc.matches_checksum(w)), api


def test_product_builder_cfg3_and_cfg5_automata():
    """The automata every other test here scans come from the ORACLE's builder (deserialize(o.serialize())); this one scans the ones the
    PRODUCT's own builder makes (daac_bytewise_build / daac_charwise_build, src/bytewise/builder.rs:187-227, src/charwise/builder.rs:148-190)
    from BASELINE's two dictionaries: same bytes as the oracle's, and count + checksum + tuples against the oracle's scan."""
    import torch
    pats = synth.patterns_cfg3()
    o = orc.OraclePma.build(pats)
    p = da.DoubleArrayAhoCorasick.new(pats)
    assert p.serialize() == o.serialize()
    n = 64 << 20
    dev = torch.empty(n, dtype=torch.uint8, device="cuda")
    synth.device_uniform(dev, synth.SEEDS["cfg3_hay"], synth.ALPHA_LOWER_SPACE)
    host = dev.cpu().numpy()
    want = o.overlapping_count(host, threads=16)
    assert p.scan_count(ScanMode.FindOverlapping, dev) == want
    assert da.last_engine() == int(Engine.Gram)
    assert p.count(ScanMode.FindOverlapping, dev) == want[0]
    w4 = o.find_overlapping_iter(host[:4 << 20])
    assert _same(p.scan(ScanMode.FindOverlapping, dev[:4 << 20]), w4)
    w1 = o.find_iter(host[:4 << 20])
    assert _same(p.scan(ScanMode.Find, dev[:4 << 20]), w1)
    pl = da.DoubleArrayAhoCorasickBuilder().match_kind(da.MatchKind.LeftmostLongest).build(pats)
    ol = orc.OraclePma.build(pats, kind="LeftmostLongest")
    assert pl.serialize() == ol.serialize()
    wl = ol.leftmost_find_iter(host[:4 << 20])
    assert _same(pl.scan(ScanMode.LeftmostFind, dev[:4 << 20]), wl)
    assert pl.scan_count(ScanMode.LeftmostFind, dev[:4 << 20]) == (len(wl), orc.matches_checksum(wl))
    # cfg5: the charwise builder, LeftmostLongest
    cp = synth.patterns_cfg5()
    co = orc.OracleCharwisePma.build(cp, kind=1)
    c = da.CharwiseDoubleArrayAhoCorasickBuilder().match_kind(1).build(cp)
    assert c.serialize() == co.serialize()
    m = (64 << 20) - (64 << 20) % synth.CFG5_SLOT
    synth.device_zipf_text(dev[:m])
    ch = dev[:m].cpu().numpy()
    cw = co.leftmost_find_iter(ch)
    assert c.scan_count(ScanMode.LeftmostFind, dev[:m]) == (len(cw), orc.matches_checksum(cw))
    k = (4 << 20) - (4 << 20) % synth.CFG5_SLOT
    assert _same(c.scan(ScanMode.LeftmostFind, dev[:k]), co.leftmost_find_iter(ch[:k]))


def test_cfg2_full_256_mib():
    """BASELINE configs[1] at its stated size: count + checksum and `.count()` of the whole 256 MiB random-ASCII haystack
    (and of the dense pattern soup of the same size) against the oracle"""
    import torch
    pats = synth.patterns_cfg2()
    o = orc.OraclePma.build(pats)
    p, _ = da.DoubleArrayAhoCorasick.deserialize(o.serialize())
    n = 256 << 20
    dev = torch.empty(n, dtype=torch.uint8, device="cuda")
    for kind in ("sparse", "dense"):
        if kind == "sparse":
            synth.device_uniform(dev, synth.SEEDS["cfg2_hay"], synth.ALPHA_PRINTABLE)
        else:
            synth.device_wordsoup(dev, synth.SEEDS["cfg2_dense"], pats, 13, noise_256=0)
        want = o.overlapping_count(dev.cpu().numpy(), threads=16)
        assert want[0] > 0
        for eng in (Engine.Auto, Engine.Gram, Engine.Tiered):
            assert p.scan_count(ScanMode.FindOverlapping, dev, engine=eng) == want, (kind, eng)
            assert p.count(ScanMode.FindOverlapping, dev, engine=eng) == want[0], (kind, eng)


def test_cfg3_full_4_gib_count_and_checksum():
    """SURVEY 8d, whole-haystack equality: (count, checksum) of ALL 4 GiB of BOTH cfg3 haystacks (BASELINE configs[2]: uniform a-z + space;
    word soup of the patterns — the text on which the `.count()` kernel runs its tail-record body) from the count + checksum kernel and
    the count from every `.count()` kernel against the oracle (positions beyond 2^32 included)"""
    import torch
    pats = synth.patterns_cfg3()
    o = orc.OraclePma.build(pats)
    p, _ = da.DoubleArrayAhoCorasick.deserialize(o.serialize())
    n = 4 << 30
    dev = torch.empty(n, dtype=torch.uint8, device="cuda")
    for kind in ("sparse", "dense"):
        if kind == "sparse":
            synth.device_uniform(dev, synth.SEEDS["cfg3_hay"], synth.ALPHA_LOWER_SPACE)
        else:
            synth.device_wordsoup(dev, synth.SEEDS["cfg3_dense"], pats, 20, noise_256=77)
        host = dev.cpu().numpy()
        want = o.overlapping_count(host, threads=16)
        del host
        assert want[0] > 2_000_000_000
        assert p.scan_count(ScanMode.FindOverlapping, dev) == want, kind
        assert da.last_engine() == int(Engine.Gram)
        assert p.count(ScanMode.FindOverlapping, dev) == want[0], kind
        for version, ppl, tail in ((2, 0, -1), (4, 16, -1), (4, 32, -1), (4, 16, 0), (4, 16, 1), (4, 32, 0), (4, 32, 1)):
            p.set_option("gram_version", version).set_option("gram_ppl", ppl).set_option("gram3_tail", tail)   # (the handle's own: daac_pma_set_option)
            assert p.count(ScanMode.FindOverlapping, dev, engine=Engine.Gram) == want[0], (kind, version, ppl, tail)
        for name in ("gram_version", "gram_ppl", "gram3_tail"):
            p.set_option(name)


def test_cfg4_32_gib_as_eight_shards_on_one_device():
    """BASELINE configs[3] at its stated size in the form one GPU can run it: the same 100 k automaton, 32 GiB = 8 shards x 4 GiB, shard k
    seeded 0xDAAC0014 + k, through the product's multi-device entry point daac_scan_count_multi with all eight shards naming
    device 0 (the 8-GPU node differs in the device ordinals only: one host thread per shard, host-side sum).  The shards are independent
    haystacks' worth of bytes laid end to end: shard k > 0 carries the last max_pattern_len - 1 bytes of shard k - 1 as its halo, so a
    match across a seam is counted once, by the shard it ends in.  Against the oracle shard by shard (count + checksum re-based to
    haystack positions, 16 threads), and `.count()` alone."""
    import torch
    pats = synth.patterns_cfg3()
    o = orc.OraclePma.build(pats)
    p, _ = da.DoubleArrayAhoCorasick.deserialize(o.serialize())
    shard = 4 << 30
    halo = p.info().max_pattern_len - 1
    bufs, shards = [], []
    want_n, s1, s2 = 0, 0, 0
    prev_tail = None
    for k in range(8):
        buf = torch.empty(halo + shard, dtype=torch.uint8, device="cuda")
        synth.device_uniform(buf[halo:], synth.SEEDS["cfg4_hay"] + k, synth.ALPHA_LOWER_SPACE)
        if prev_tail is not None:
            buf[:halo] = prev_tail
        prev_tail = buf[-halo:].clone()
        view = buf if k else buf[halo:]
        bufs.append(buf)
        shards.append((0, view, halo if k else 0, k * shard))
        # the oracle on this shard alone: matches with their end inside it (the halo in front), ends counted from the haystack's first byte
        host = view.cpu().numpy()
        h = halo if k else 0
        c_all = o.overlapping_count(host, threads=16)
        c_halo = o.overlapping_count(host[:h], threads=1) if h else (0, 0)
        n_k = c_all[0] - c_halo[0]
        a1, a2 = (c_all[1] >> 32) - (c_halo[1] >> 32), (c_all[1] & 0xFFFFFFFF) - (c_halo[1] & 0xFFFFFFFF)
        base = k * shard - h
        want_n += n_k
        s1 = (s1 + a1) & 0xFFFFFFFF
        s2 = (s2 + a2 + (base & 0xFFFFFFFF) * a1) & 0xFFFFFFFF
        del host
    want = (want_n, (s1 << 32) | s2)
    assert want_n > 16_000_000_000
    assert da.scan_count_multi(p, ScanMode.FindOverlapping, shards) == want
    assert da.last_engine() == int(Engine.Gram)
    assert da.scan_count_multi(p, ScanMode.FindOverlapping, shards, checksum=False) == want[0]


def test_cfg3_count_kernel_on_a_vector_of_window_counts():
    """The `.count()` kernel is observed through ONE integer per call; a missed match here and a double count there would cancel.
    4 096 windows (begin, len) of random sizes and alignments over 64 MiB of the cfg3 haystack, each counted on its own
    (`daac_scan_count_only_range`) and compared with the oracle's matches ending in (begin, len]."""
    import torch
    pats = synth.patterns_cfg3()
    o = orc.OraclePma.build(pats)
    p, _ = da.DoubleArrayAhoCorasick.deserialize(o.serialize())
    n = 64 << 20
    dev = torch.empty(n, dtype=torch.uint8, device="cuda")
    synth.device_uniform(dev, synth.SEEDS["cfg3_hay"], synth.ALPHA_LOWER_SPACE)
    ends = np.sort(o.find_overlapping_iter(dev.cpu().numpy())["end"].astype(np.int64))
    rng = np.random.default_rng(4096)
    los = rng.integers(0, n - (1 << 17), size=4096)
    sizes = np.concatenate([rng.integers(1, 64, size=512), rng.integers(64, 4096, size=1536), rng.integers(4096, 1 << 17, size=2048)])
    his = los + sizes
    want = np.searchsorted(ends, his, side="right") - np.searchsorted(ends, los, side="right")
    got = np.array([p.count(ScanMode.FindOverlapping, dev[:int(h)], begin=int(l)) for l, h in zip(los, his)])
    bad = np.nonzero(got != want)[0]
    assert len(bad) == 0, (bad[:8], got[bad[:8]], want[bad[:8]], los[bad[:8]], his[bad[:8]])
    assert da.last_engine() == int(Engine.Gram)
    # the other launch shape (16 positions per lane) and the tail-record body on every eighth window
    for ppl, tail in ((16, 0), (32, 1)):
        p.set_option("gram_ppl", ppl).set_option("gram3_tail", tail)
        got3 = np.array([p.count(ScanMode.FindOverlapping, dev[:int(h)], begin=int(l), engine=Engine.Gram) for l, h in zip(los[::8], his[::8])])
        assert np.array_equal(got3, want[::8]), (ppl, tail)


class _DeviceWords:
    """a device buffer of n int64 words as a __cuda_array_interface__ object (torch.as_tensor takes it without a copy)"""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i8", "data": (ptr, False), "version": 2}


def _checksum_of_device_tuples16(dm):
    """(count, checksum) of a device list of daac_match16 {end u64, length u32, value u32}, computed on the device with torch's wrapping
    int64 arithmetic: h = low32(mix64(value << 32 | length)), S1 = sum h, S2 = sum h * low32(end) (include/daachorse_amd.h)"""
    import torch
    n = dm.count
    w = torch.as_tensor(_DeviceWords(dm.ptr, 2 * n), device="cuda").view(n, 2)
    M1, M2 = -4658895280553007687, -7723592293110705685   # 0xBF58476D1CE4E5B9, 0x94D049BB133111EB as int64

    def lsr(z, k):
        return (z >> k) & ((1 << (64 - k)) - 1)
    s1 = s2 = 0
    for lo in range(0, n, 1 << 27):   # (in pieces: a few temporaries of the piece's size at a time)
        end, z = w[lo:lo + (1 << 27), 0], w[lo:lo + (1 << 27), 1].clone()   # value << 32 | length is the tuple's second word as it stands
        z = (z ^ lsr(z, 30)) * M1
        z = (z ^ lsr(z, 27)) * M2
        h = (z ^ lsr(z, 31)) & 0xFFFFFFFF
        s1 += int(h.sum().item())
        s2 += int(((h * (end & 0xFFFFFFFF)) & 0xFFFFFFFF).sum().item())
    return n, ((s1 & 0xFFFFFFFF) << 32) | (s2 & 0xFFFFFFFF)


_ORACLE_CFG3 = []


def _oracle_cfg3(pats):
    if not _ORACLE_CFG3:
        _ORACLE_CFG3.append(orc.OraclePma.build(pats))
    return _ORACLE_CFG3[0]


def test_cfg3_tuples_at_size_checksum_of_the_list():
    """Full-size property of the tuple emitter: the (count, checksum) of the LIST daac_scan_device16 leaves in HBM — computed from the
    tuples themselves on the device — equals what the count + checksum kernel says of the same haystack (which the 4 GiB test pins to the
    oracle).  3.5 GiB of cfg3 = four emitter windows, 2.24 G tuples (tuple indices beyond 2^31 — a sign-extended tile offset faulted there
    until round 4 — and 34 GB of list), ends beyond 2^32; ends ascend; 64 random 1 MiB slices of the haystack compared tuple for tuple
    (order within an end included) with the oracle; and the same for 1 GiB of word soup."""
    import torch
    pats = synth.patterns_cfg3()
    p = da.DoubleArrayAhoCorasick.new(pats)
    p.set_option("max_result_bytes", 64 << 30)   # (this handle may hand out a 34 GB list)
    for kind, n in (("sparse", (7 << 29) + 12345), ("dense", 1 << 30)):
        dev = torch.empty(n, dtype=torch.uint8, device="cuda")
        if kind == "sparse":
            synth.device_uniform(dev, synth.SEEDS["cfg3_hay"], synth.ALPHA_LOWER_SPACE)
        else:
            synth.device_wordsoup(dev, synth.SEEDS["cfg3_dense"], pats, 20)
        want = p.scan_count(ScanMode.FindOverlapping, dev)
        dm = p.scan_device(ScanMode.FindOverlapping, dev, fmt16=True)
        assert da.last_engine() == int(Engine.Gram)
        assert _checksum_of_device_tuples16(dm) == want, kind
        assert kind != "sparse" or dm.count > (1 << 31)
        ends = torch.as_tensor(_DeviceWords(dm.ptr, 2 * dm.count), device="cuda").view(dm.count, 2)[:, 0]
        assert bool((ends[1:] >= ends[:-1]).all()), kind      # by end (as unsigned they are below 2^63: the comparison holds)
        assert int(ends[-1].item()) <= n and int(ends[0].item()) >= 1
        # ... and 64 random 1 MiB slices of the haystack, tuple for tuple against the oracle (order within an end included):
        # the list's tuples with lo < end <= hi are those the oracle reports on [lo - 64, hi) with an end beyond the 64 bytes of halo
        o = _oracle_cfg3(pats)
        rng = np.random.default_rng(20250 + len(kind))
        words = torch.as_tensor(_DeviceWords(dm.ptr, 2 * dm.count), device="cuda").view(dm.count, 2)
        for lo in [0, n - (1 << 20)] + [int(x) for x in rng.integers(64, n - (1 << 20), size=62)]:
            hi = lo + (1 << 20)
            i0, i1 = (int(x) for x in torch.searchsorted(ends, torch.tensor([lo, hi], device="cuda", dtype=ends.dtype), right=True))
            got = words[i0:i1].cpu().numpy()
            from_ = max(0, lo - 64)
            ref = o.find_overlapping_iter(dev[from_:hi].cpu().numpy())
            ref = ref[ref["end"].astype(np.int64) + from_ > lo]
            assert len(ref) == i1 - i0, (kind, lo, len(ref), i1 - i0)
            assert np.array_equal(got[:, 0], ref["end"].astype(np.int64) + from_), (kind, lo)
            assert np.array_equal(got[:, 1] & 0xFFFFFFFF, (ref["end"] - ref["start"]).astype(np.int64)), (kind, lo)      # length
            assert np.array_equal((got[:, 1] >> 32) & 0xFFFFFFFF, ref["value"].astype(np.int64)), (kind, lo)            # value
        dm.free()
        del dev, ends, words
        torch.cuda.empty_cache()
    # the PFX engine's list likewise (utf8jp scanned bytewise, 1 GiB: counted first, emitted piece by piece)
    q = da.DoubleArrayAhoCorasick.new(synth.patterns_cfg5())
    q.set_option("max_result_bytes", 64 << 30)
    m = (1 << 30) - (1 << 30) % synth.CFG5_SLOT
    dev = torch.empty(m, dtype=torch.uint8, device="cuda")
    synth.device_zipf_text(dev)
    want = q.scan_count(ScanMode.FindOverlapping, dev)
    dm = q.scan_device(ScanMode.FindOverlapping, dev, fmt16=True)
    assert da.last_engine() == int(Engine.Pfx)
    assert _checksum_of_device_tuples16(dm) == want
    dm.free()


"""`.count()` kernel of round 5 (gram4_kernels.hip, option gram_version = 4; what AUTO takes) against the oracle: every launch
shape (16 / 32 positions per lane, 1024- and 512-thread workgroups), plain records and tail records from the hit record on,
both rank directories, arithmetic byte classes and the class table, K = 3 and K = 2, dictionaries whose bytes are not one
range, unaligned haystacks, shards, haystacks shorter than one step, text in which every position hits and continues."""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

import daachorse_amd as da
from daachorse_amd import Engine, ScanMode, synth

# (positions per lane, tail records (-1: the workgroups' own probe), per-word directory, threads per workgroup, arithmetic classes, the filter in
# front of rank + gather (round 6; it serves the plain records only: with tail = 1 the option is moot))
VARIANTS = [(16, 0, 1, 1024, 1, 1), (16, 1, 1, 1024, 1, 1), (32, 0, 1, 1024, 1, 1), (32, 1, 0, 512, 1, 0), (16, 0, 0, 1024, 0, 1), (32, 0, 1, 512, 0, 1), (16, 1, 1, 512, 0, 0),
            (32, 1, 0, 1024, 1, 1), (32, 0, 1, 1024, 1, 0), (16, 0, 0, 512, 1, 0), (32, -1, 1, 1024, 1, 1), (32, 0, 0, 1024, 0, 1)]


def _pma(patterns):
    o = orc.OraclePma.build(patterns)
    p, rest = da.DoubleArrayAhoCorasick.deserialize(o.serialize())
    assert rest == b""
    return o, p


def _count3(p, hay, ppl, wtext, rfull, threads=1024, arith=1, filt=1, **kw):
    """the launch shape is the HANDLE's (daac_pma_set_option): nothing process-wide is touched, nothing to reset"""
    for k, v in (("gram_version", 4), ("gram4_arith", arith), ("gram_ppl", ppl), ("gram3_tail", wtext), ("gram2_rfull", rfull), ("threads", threads), ("gram4_filter", filt)):
        p.set_option(k, v)
    got = p.count(ScanMode.FindOverlapping, hay, engine=Engine.Gram, **kw)
    assert da.last_engine() == int(Engine.Gram)
    lk = da.last_kernel()   # (a dictionary squeezed into a few KB of tables may be served by the first table set: "gram")
    if lk.startswith("gram4"):
        assert lk.startswith(f"gram4 ppl={ppl} ") and f"waves={threads // 64}" in lk, lk
        if filt == 0:
            assert "filter=0" in lk, lk
    return got


def test_gram4_against_the_oracle():
    import torch
    rng = np.random.default_rng(303)
    pats3 = synth.patterns_cfg3(30000)
    cases = [(synth.patterns_cfg1(), synth.uniform_haystack(70001, 5, synth.ALPHA_ABCD)),
             (synth.patterns_cfg2(500), synth.uniform_haystack(1 << 20, 6, synth.ALPHA_LOWER)),
             (pats3, synth.uniform_haystack(3 << 20, synth.SEEDS["cfg3_hay"], synth.ALPHA_LOWER_SPACE)),
             (pats3, synth.wordsoup_haystack(3 << 20, synth.SEEDS["cfg3_dense"], pats3, 20)),
             (["ab", "ab", "b", "abab", "bababab"], np.frombuffer(b"abababbab" * 3000, dtype=np.uint8)),
             (sorted(set(bytes(rng.choice(np.frombuffer(b"acegikmoqsuwy", dtype=np.uint8), size=int(rng.integers(2, 9)))) for _ in range(3000))),
              rng.choice(np.frombuffer(b"abcdefghijklmnopqrstuvwxyz{ ", dtype=np.uint8), size=1 << 20)),
             (sorted(set(bytes(rng.choice(np.arange(0xf0, 0x100, dtype=np.uint8), size=int(rng.integers(2, 7)))) for _ in range(2000))),
              rng.choice(np.arange(0xe8, 0x100, dtype=np.uint8), size=1 << 20))]
    for pats, hay in cases:
        o, _ = _pma(pats)
        want = o.overlapping_count(hay, threads=8)[0]
        dev = torch.from_numpy(np.concatenate([np.zeros(5, dtype=np.uint8), hay])).cuda()[5:]  # not 16-byte aligned
        for budget in (158 * 1024, 24 * 1024):
            q, _ = da.DoubleArrayAhoCorasick.deserialize(o.serialize())
            q.set_option("gram_lds_budget", budget)   # (read at upload)
            assert q.upload().info().gram2_available
            for ppl, wtext, rfull, nb, ar, fl in VARIANTS:
                assert _count3(q, dev, ppl, wtext, rfull, nb, ar, fl) == want, (len(pats), budget, ppl, wtext, rfull, nb, ar, fl)
            cut = int(rng.integers(1, len(hay)))
            assert _count3(q, dev[:cut], 16, 0, 1) + _count3(q, dev, 16, 0, 1, begin=cut) == want, cut
            assert _count3(q, dev[:cut], 32, 1, 1) + _count3(q, dev, 32, 1, 1, begin=cut) == want, cut


def test_gram4_short_and_ragged_haystacks():
    """lengths around the step (1 KiB / 2 KiB) and region sizes, every alignment of the first byte"""
    import torch
    rng = np.random.default_rng(404)
    pats = synth.patterns_cfg3(5000)
    o, p = _pma(pats)
    p.upload()
    base = synth.wordsoup_haystack(1 << 18, 11, pats, 20)
    buf = torch.from_numpy(base).cuda()
    p.set_option("gram_region", 2048)
    lengths = [0, 1, 2, 3, 4, 5, 15, 16, 17, 63, 64, 1023, 1024, 1025, 2047, 2048, 2049, 4095, 4096, 4100, 6143, 6144, 65535, 65536, 65537]
    lengths += [int(x) for x in rng.integers(1, 1 << 17, size=12)]
    for n in lengths:
        off = int(rng.integers(0, 32))
        h = base[off:off + n]
        want = o.overlapping_count(h, threads=1)[0] if n else 0
        for ppl, wtext, rfull, nb, ar, fl in VARIANTS[:5] + VARIANTS[8:]:
            assert _count3(p, buf[off:off + n], ppl, wtext, rfull, nb, ar, fl) == want, (n, off, ppl, wtext, ar, fl)


def test_gram4_every_position_hits_and_continues():
    import torch
    rng = np.random.default_rng(1051)
    syms = np.frombuffer(b"acinrs", dtype=np.uint8)
    pats = [bytes(syms[rng.integers(0, 6, size=int(rng.integers(4, 9)))]) for _ in range(5000)]
    hay = np.frombuffer(b"".join(pats[i] for i in rng.integers(0, 5000, size=60_000).tolist())[:300_000], dtype=np.uint8).copy()
    for budget in (158 * 1024, 9216):
        o, p = _pma(pats)
        p.set_option("gram_lds_budget", budget).set_option("gram_slab", 0)
        p.upload()
        dev = torch.from_numpy(hay).cuda()[13:]
        want = o.overlapping_count(dev.cpu().numpy(), threads=8)[0]
        for ppl, wtext, rfull, nb, ar, fl in VARIANTS:
            assert _count3(p, dev, ppl, wtext, rfull, nb, ar, fl) == want, (budget, ppl, wtext, rfull, nb, ar, fl)
    # "aaaa...": every prefix a pattern, every position a hit with a long walk behind it
    o, p = _pma([b"a" * k for k in range(1, 40)])
    hay = np.frombuffer(b"a" * 100_000 + b"b" + b"a" * 5000, dtype=np.uint8)
    want = o.overlapping_count(hay, threads=4)[0]
    for ppl, wtext, rfull, nb, ar, fl in VARIANTS:
        assert _count3(p, torch.from_numpy(hay.copy()).cuda(), ppl, wtext, rfull, nb, ar, fl) == want


def test_gram4_cfg3_agrees_with_gram2_and_oracle():
    import torch
    pats = synth.patterns_cfg3()
    o, p = _pma(pats)
    p.upload()
    n = 64 << 20
    dev = torch.empty(n, dtype=torch.uint8, device="cuda")
    for fill in ("sparse", "dense"):
        if fill == "sparse":
            synth.device_uniform(dev, synth.SEEDS["cfg3_hay"], synth.ALPHA_LOWER_SPACE)
        else:
            synth.device_wordsoup(dev, synth.SEEDS["cfg3_dense"], pats, 20, noise_256=77)
        want = o.overlapping_count(dev.cpu().numpy(), threads=8)[0]
        p.set_option("gram_version", 2)
        assert p.count(ScanMode.FindOverlapping, dev, engine=Engine.Gram) == want
        for ppl, wtext, rfull, nb, ar, fl in VARIANTS:
            assert _count3(p, dev, ppl, wtext, rfull, nb, ar, fl) == want, (fill, ppl, wtext, rfull, nb, ar, fl)


def test_gram4_walkers_across_2_and_4_gib():
    """a walker that starts just below a multiple of 2 GiB / 4 GiB of the virtual position and walks across it (round-5 advisor: the slab kept
    the low word of a walker's position, which wrapped at 4 GiB while the epoch's high word stayed): a haystack of 4 GiB + 32 MiB, blank
    but for word soup — every position a hit, long walks — around both boundaries, with the dictionary's longest word laid across each
    boundary at several offsets, two alignments of the first byte, both bodies of the kernel"""
    import torch
    pats = synth.patterns_cfg3()
    longest = max(pats, key=len)
    assert len(longest) >= 15
    o, p = _pma(pats)
    p.upload()
    n = (4 << 30) + (32 << 20)
    buf = torch.full((n + 16,), 0x20, dtype=torch.uint8, device="cuda")
    w = 4 << 20
    word = torch.from_numpy(np.frombuffer(longest, dtype=np.uint8).copy()).cuda()
    checked_whole = False
    for off in (0, 5):   # virtual position = index + (address & 15)
        dev = buf[off:off + n]
        for k in (2, 5, 9, 13):
            want = 0
            for b in (1 << 31, 1 << 32):
                vb = b - off   # index of the byte whose virtual position is b
                synth.device_wordsoup(dev[vb - w:vb + w], synth.SEEDS["cfg3_dense"] + k, pats, 20, noise_256=0)
                dev[vb - k:vb - k + len(longest)] = word
                dev[vb - k - 1] = 0x20
                dev[vb - k + len(longest)] = 0x20
                want += o.overlapping_count(dev[vb - w:vb + w].cpu().numpy(), threads=16)[0]
            if not checked_whole:   # the blanks between the windows hold no match (' ' is no pattern byte): once against the whole haystack
                assert o.overlapping_count(dev.cpu().numpy(), threads=16)[0] == want
                checked_whole = True
            assert want > 1_000_000
            for ppl, tail, fl in ((32, 1, 1), (32, 0, 1), (32, 0, 0), (16, 1, 1), (32, -1, 1)):
                assert _count3(p, dev, ppl, tail, 1, filt=fl) == want, (off, k, ppl, tail, fl)

"""JUMP engine (jump_kernels.hip): find_iter count (+ checksum) of Standard bytewise automata over per-position jump tables — against
the oracle's FindIterator and against the chain walkers it replaces: uniform text and text made of the patterns, duplicates, one-byte
patterns, dictionaries without short patterns (every start is a walk), long stretches where nothing starts (the escape), unaligned
and ragged haystacks, shards that start at a sync point."""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

import daachorse_amd as da
from daachorse_amd import ScanMode, synth


@pytest.fixture(autouse=True)
def _opts():
    da.set_option("jump", 1)  # (read at upload: the tables are built only when the option is on; off by default)
    yield
    da.set_option("jump", 0)
    da.set_option("seg_bytes", 0)


def _pma(patterns):
    o = orc.OraclePma.build(patterns)
    p, rest = da.DoubleArrayAhoCorasick.deserialize(o.serialize())
    assert rest == b""
    return o, p


def _want(o, hay):
    m = o.find_iter(hay)
    return len(m), orc.matches_checksum(m)


def _both(p, dev, **kw):
    da.set_option("jump", 1)
    a = p.scan_count(ScanMode.Find, dev, **kw)
    assert da.last_engine() == int(da.Engine.Jump) or dev.numel() == 0
    da.set_option("jump", 0)
    b = p.scan_count(ScanMode.Find, dev, **kw)
    assert da.last_engine() == int(da.Engine.DArray)
    da.set_option("jump", 1)
    assert a == b, ("jump tables vs chain walkers", a, b)
    return a


def test_jump_against_the_oracle():
    import torch
    rng = np.random.default_rng(77)
    pats3 = synth.patterns_cfg3(30000)
    syms = np.frombuffer(b"abcdefg", dtype=np.uint8)
    longp = [bytes(syms[rng.integers(0, 7, size=int(rng.integers(4, 100)))]) for _ in range(400)]
    cases = [(synth.patterns_cfg1(), synth.uniform_haystack(70001, 5, synth.ALPHA_ABCD)),
             (pats3, synth.uniform_haystack(3 << 20, synth.SEEDS["cfg3_hay"], synth.ALPHA_LOWER_SPACE)),
             (pats3, synth.wordsoup_haystack(3 << 20, synth.SEEDS["cfg3_dense"], pats3, 20)),
             (longp, np.frombuffer(b"".join(longp[i] if k % 2 else bytes(syms[rng.integers(0, 7, size=9)]) for k, i in enumerate(rng.integers(0, 400, size=40000).tolist())), dtype=np.uint8)),
             (longp, np.frombuffer((b"zzzz" * 300 + longp[3] + b"z" * 700 + longp[5][:-1] + b"q" * 300 + longp[7]) * 50, dtype=np.uint8)),
             (["ab", "ab", "b", "abab", "bababab", "ba", "b"], np.frombuffer(b"abababbab" * 30000, dtype=np.uint8))]
    for pats, hay in cases:
        o, p = _pma(pats)
        p.upload()
        want = _want(o, hay)
        dev = torch.from_numpy(np.concatenate([np.zeros(5, dtype=np.uint8), hay])).cuda()[5:]  # not 16-byte aligned
        assert _both(p, dev) == want, (len(pats), len(hay))
        for seg in (1024, 4096):
            da.set_option("seg_bytes", seg)
            assert _both(p, dev) == want, (len(pats), len(hay), seg)
        da.set_option("seg_bytes", 0)


def test_jump_short_and_ragged_haystacks():
    import torch
    rng = np.random.default_rng(78)
    pats = synth.patterns_cfg3(5000)
    o, p = _pma(pats)
    p.upload()
    base = synth.wordsoup_haystack(1 << 18, 11, pats, 20)
    buf = torch.from_numpy(base).cuda()
    lengths = [0, 1, 2, 3, 4, 5, 15, 16, 17, 63, 64, 1021, 1022, 1023, 1024, 1025, 1790, 1791, 1792, 1793, 2047, 2048, 2049, 4095, 4096, 4100, 65535, 65536, 65537]
    lengths += [int(x) for x in rng.integers(1, 1 << 17, size=12)]
    for n in lengths:
        off = int(rng.integers(0, 32))
        h = base[off:off + n]
        want = _want(o, h) if n else (0, 0)
        assert _both(p, buf[off:off + n]) == want, (n, off)


def test_jump_cfg3_64_mib():
    import torch
    pats = synth.patterns_cfg3()
    o, p = _pma(pats)
    p.upload()
    dev = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    for fill in ("sparse", "dense"):
        if fill == "sparse":
            synth.device_uniform(dev, synth.SEEDS["cfg3_hay"], synth.ALPHA_LOWER_SPACE)
        else:
            synth.device_wordsoup(dev, synth.SEEDS["cfg3_dense"], pats, 20, noise_256=77)
        want = _want(o, dev.cpu().numpy())
        assert _both(p, dev) == want, fill

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
    on 256 MiB, and the other three charwise iterators on 2 MiB."""
    import torch
    pats = synth.patterns_cfg5()
    assert len(pats) == 50_000
    o = orc.OracleCharwisePma.build(pats, kind=1)
    p, _ = da.CharwiseDoubleArrayAhoCorasick.deserialize(o.serialize())
    n = (256 << 20) - (256 << 20) % synth.CFG5_SLOT
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
    assert p.scan_count(ScanMode.LeftmostFind, dev) == (len(big), orc.matches_checksum(big))
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

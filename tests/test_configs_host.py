"""BASELINE.json's configurations on the host: the cfg1 plumbing case (3 patterns, the seeded 1 KiB haystack and the
README's "abcd") through the oracle against answers obtained without any automaton, and the synthetic generators of
cfg5.  No GPU."""
import numpy as np

from daachorse_amd import synth
from oracle import oracle as orc


def _brute_overlapping(patterns, hay):
    """every occurrence of every pattern, in the reference's order: by end, longest first (the output list of the state
    reached at that end lists the longest suffix first, bytewise/iter.rs:147-172); value = index of the pattern"""
    out = []
    for end in range(1, len(hay) + 1):
        for i, p in sorted(enumerate(patterns), key=lambda t: -len(t[1])):
            if len(p) <= end and hay[end - len(p):end] == p:
                out.append((end - len(p), end, i))
    return out


def test_cfg1_readme_answer():
    """README.md:57-71: patterns bcd / ab / a over "abcd" -> (0,1,2), (0,2,1), (1,4,0)"""
    o = orc.OraclePma.build(synth.patterns_cfg1())
    got = [(int(m["start"]), int(m["end"]), int(m["value"])) for m in o.find_overlapping_iter(b"abcd")]
    assert got == [(0, 1, 2), (0, 2, 1), (1, 4, 0)]


def test_cfg1_seeded_haystack():
    """cfg1: 1 024 bytes uniform over {a,b,c,d}, seed 0xDAAC0001, against a brute-force search"""
    pats = synth.patterns_cfg1()
    hay = bytes(synth.uniform_haystack(1024, synth.SEEDS["cfg1_hay"], synth.ALPHA_ABCD))
    assert set(hay) <= set(b"abcd") and len(hay) == 1024
    want = _brute_overlapping(pats, hay)
    o = orc.OraclePma.build(pats)
    m = o.find_overlapping_iter(hay)
    got = [(int(x["start"]), int(x["end"]), int(x["value"])) for x in m]
    assert got == want and len(want) > 300
    assert o.overlapping_count(np.frombuffer(hay, dtype=np.uint8), threads=1) == (len(want), orc.matches_checksum(m))


def test_cfg5_generators():
    pats = synth.patterns_cfg5(2000)
    assert len(set(pats)) == 2000
    allowed = {int(c) for c in synth.CFG5_CODEPOINTS}
    for p in pats:
        s = p.decode("utf-8")
        assert 2 <= len(s) <= 8 and all(ord(ch) in allowed for ch in s)
    assert pats == synth.patterns_cfg5(5000)[:2000]  # a prefix of the stream, whatever n
    n = 48 * 4000
    text = synth.zipf_text(n)
    s = bytes(text).decode("utf-8")  # whole characters only
    ascii_share = sum(ch < "\x80" for ch in s) / len(s)
    assert 0.08 < ascii_share < 0.16  # 10 % of the draws + the 1-2 byte slot fillers
    assert all(ord(ch) in allowed or 0x20 <= ord(ch) < 0x7F for ch in s)
    # Zipf(1.0): the first symbol is about twice as frequent as the second
    c0, c1 = s.count(chr(int(synth.CFG5_CODEPOINTS[0]))), s.count(chr(int(synth.CFG5_CODEPOINTS[1])))
    assert 1.6 < c0 / c1 < 2.5
    # a pure function of (seed, byte index): any window equals the same bytes of the whole
    for off, ln in ((0, 100), (47, 3), (12345, 5000), (n - 7, 7)):
        assert np.array_equal(synth.zipf_text(ln, offset=off), text[off:off + ln])
    assert synth.cfg5_haystack_bytes() % synth.CFG5_SLOT == 0 and (1 << 30) - synth.cfg5_haystack_bytes() < synth.CFG5_SLOT

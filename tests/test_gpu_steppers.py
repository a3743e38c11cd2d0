"""Chunk-fed steppers on the MI355X (daac_stream_*): whatever the chunking, the concatenation of what the calls
return equals what the reference's steppers report byte by byte (oracle: orc_find_stepper /
orc_find_overlapping_stepper, pinned by the stepper runners of tests/aho_corasick_crate_test.rs) — bytewise and
charwise, cuts inside UTF-8 characters, "" among the patterns, empty chunks, device-resident chunks."""
import numpy as np
import pytest

from conftest import iter_vector_runs
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

import daachorse_amd as da


def _sev(m):
    return [(int(x["start"]), int(x["end"]), int(x["value"])) for x in m]


def _feed_all(stepper, raw, cuts, compact=False):
    """compact: daac_stream_feed_compact (8-byte tuples in the stream's page-locked block, round 6) instead of daac_stream_feed"""
    got = []
    prev = 0
    for c in list(cuts) + [len(raw)]:
        if compact:
            run, base, bits = stepper.feed_compact(raw[prev:c])
            got += _sev(stepper.decode8(run, base, bits))
        else:
            got += _sev(stepper.feed(raw[prev:c]))
        prev = c
    return got


def _cuts(rng, n):
    k = int(rng.integers(0, 8))
    return sorted(int(x) for x in rng.integers(0, n + 1, size=k))


STEPPERS = [("find_stepper", "find_stepper"), ("find_overlapping_stepper", "find_overlapping_stepper"),
            ("find_overlapping_no_suffix_stepper", "find_overlapping_no_suffix_iter")]


def test_golden_vectors_through_the_steppers(vectors):
    """find_stepper / find_overlapping_stepper runners (tests/aho_corasick_crate_test.rs:446-464, 500-521 and the
    charwise twins), each haystack fed in one piece, byte by byte and at seeded cuts"""
    rng = np.random.default_rng(3)
    n = 0
    for runner, case in iter_vector_runs(vectors):
        if runner["api"] not in ("find_stepper", "find_overlapping_stepper"):
            continue
        want = [(s, e, v) for v, s, e in (tuple(t) for t in case["matches"])]
        raw = case["haystack"].encode()
        for flavour in ("bytewise", "charwise"):
            p = (da.DoubleArrayAhoCorasick if flavour == "bytewise" else da.CharwiseDoubleArrayAhoCorasick).new(case["patterns"])
            for cuts in ([], list(range(1, len(raw))), _cuts(rng, len(raw))):
                for compact in (False, True):
                    got = _feed_all(getattr(p, runner["api"])(), raw, cuts, compact)
                    assert got == want, (flavour, runner["api"], case["name"], cuts, compact)
        n += 1
    assert n == 61 + 57


@pytest.mark.parametrize("flavour", ["bytewise", "charwise"])
def test_fuzz_chunkings(flavour):
    rng = np.random.default_rng(17 if flavour == "bytewise" else 18)
    alphabets = [list("ab"), list("abcd"), [chr(c) for c in range(0x3041, 0x3046)], list("aé世") + ["\U0001F600"]]
    for trial in range(60):
        A = alphabets[int(rng.integers(0, 2 if flavour == "bytewise" else len(alphabets)))]
        pats = ["".join(A[i] for i in rng.integers(0, len(A), size=int(rng.integers(1, 6)))) for _ in range(int(rng.integers(1, 30)))]
        if trial % 6 == 0:
            pats.append("")
        text = "".join(A[i] for i in rng.integers(0, len(A), size=int(rng.integers(0, 4000))))
        raw = text.encode()
        if flavour == "bytewise":
            o = orc.OraclePma.build(pats)
            p, _ = da.DoubleArrayAhoCorasick.deserialize(o.serialize())
        else:
            o = orc.OracleCharwisePma.build(pats)
            p, _ = da.CharwiseDoubleArrayAhoCorasick.deserialize(o.serialize())
        p.set_option("seg_bytes", int(rng.choice([0, 16, 64])))
        for api, oapi in STEPPERS:
            want = _sev(getattr(o, oapi)(text))
            cuts = _cuts(rng, len(raw))
            if trial % 5 == 0:
                cuts = sorted(cuts + cuts[:2])  # repeated cut positions = empty chunks
            got = _feed_all(getattr(p, api)(), raw, cuts, compact=trial % 2 == 1)
            assert got == want, (flavour, api, pats, cuts, text[:100])


def test_long_stream_device_chunks_and_bounded_carry():
    """a 6 MB word-soup stream in 64 KB device chunks: FIND must not hold on to more than a halo of old bytes when
    matches are rare, nor lose the chain when they are dense"""
    import torch
    from daachorse_amd import synth
    pats = synth.patterns_cfg3(5000)
    o = orc.OraclePma.build(pats)
    p, _ = da.DoubleArrayAhoCorasick.deserialize(o.serialize())
    hay = synth.wordsoup_haystack(6_000_000, synth.SEEDS["cfg3_dense"], pats, 20)
    hay[1_000_000:3_000_000] = ord("#")  # a long stretch without any pattern byte
    dev = torch.from_numpy(hay).cuda()
    for api, oapi in STEPPERS[:2]:
        want = getattr(o, oapi)(hay)
        st = getattr(p, api)()
        parts = [st.feed(dev[i:i + 65536]) for i in range(0, len(hay), 65536)]
        got = np.concatenate([x for x in parts if len(x)])
        assert len(got) == len(want) and np.array_equal(got["start"], want["start"]) and np.array_equal(got["end"], want["end"]) and \
            np.array_equal(got["value"], want["value"]), api
        # the compact form (8-byte tuples, the stream's own page-locked block) over chunks of changing sizes, host and device chunks mixed
        st = getattr(p, api)()
        parts, i, k = [], 0, 0
        while i < len(hay):
            n = [65536, 1 << 20, 3, 300_001, 4096][k % 5]
            chunk = dev[i:i + n] if k % 3 else hay[i:i + n]
            parts.append(st.decode8(*st.feed_compact(chunk)))   # (decoded at once: the view is the stream's until the next feed)
            i += n
            k += 1
        got = np.concatenate([x for x in parts if len(x)])
        assert len(got) == len(want) and np.array_equal(got["start"], want["start"]) and np.array_equal(got["end"], want["end"]) and \
            np.array_equal(got["value"], want["value"]), (api, "compact")


def test_compact_feed_refuses_what_it_cannot_say():
    """daac_stream_feed_compact packs end and length into one word: a dictionary whose longest pattern leaves no room for a chunk (status 6: the
    caller takes daac_stream_feed), and a chunk of 2^end_bits bytes or more; the stream goes on after a refusal"""
    pats = [b"abc", b"bcd", b"x" * 5000]   # 13 length bits: 512 KiB of room for ends, less than a chunk is allowed to need
    o = orc.OraclePma.build(pats)
    p, _ = da.DoubleArrayAhoCorasick.deserialize(o.serialize())
    hay = np.frombuffer(b"xabcdx" * 1000, dtype=np.uint8)
    st = p.find_overlapping_stepper()
    with pytest.raises(da.DaachorseError) as ei:
        st.feed_compact(hay)
    assert ei.value.code == 6
    got = st.feed(hay)
    want = o.find_overlapping_iter(hay)
    assert len(got) == len(want) and np.array_equal(got["end"], want["end"]) and np.array_equal(got["value"], want["value"])
    pats = [b"abc", b"bcd", b"y" * 1500]   # 11 length bits: chunks below 2 MiB
    o2 = orc.OraclePma.build(pats)
    q, _ = da.DoubleArrayAhoCorasick.deserialize(o2.serialize())
    long_hay = np.frombuffer(b"xabcdx" * 400_000, dtype=np.uint8)   # 2.4 MB
    for api, oapi in (("find_overlapping_stepper", "find_overlapping_iter"), ("find_stepper", "find_iter")):
        st = getattr(q, api)()
        with pytest.raises(da.DaachorseError) as ei:
            st.feed_compact(long_hay)
        assert ei.value.code == 6
        parts = [st.decode8(*st.feed_compact(long_hay[i:i + 700_001])) for i in range(0, len(long_hay), 700_001)]
        got = np.concatenate([x for x in parts if len(x)])
        want = getattr(o2, oapi)(long_hay)
        assert len(got) == len(want) and np.array_equal(got["start"], want["start"]) and np.array_equal(got["end"], want["end"]) and \
            np.array_equal(got["value"], want["value"]), api

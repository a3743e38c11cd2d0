"""CPU-side tests of the product's host logic (no GPU): C-ABI surface, blob handling, re-pack."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from oracle import oracle as orc

import daachorse_amd as da
from daachorse_amd import _ffi, synth


def test_abi_exports_every_declared_symbol():
    names = set()
    for hdr in ("daachorse_amd.h", "daac_synth.h"):
        src = open(os.path.join(ROOT, "include", hdr)).read()
        names |= set(re.findall(r"\b(daac_[a-z_0-9]+)\s*\(", src))
    assert len(names) >= 20
    lib = C.CDLL(_ffi._build.LIB_PATH)
    for n in sorted(names):
        assert hasattr(lib, n), f"{n} declared in include/ but not exported"


def test_no_oracle_in_product():
    """The product path must never route through oracle/ (or any CPU scan fallback)."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "daachorse_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in text.lower() or f == "__init__.py" and False, f"{f} mentions the oracle"


def test_blob_roundtrip_and_validation(pins):
    for e in pins["serialize_roundtrip"]:
        blob = orc.OraclePma.build(e["patterns"], kind=e["kind"]).serialize()
        pma, rest = da.DoubleArrayAhoCorasick.deserialize(blob + b"tail")
        assert rest == b"tail"
        assert pma.serialize() == blob
    with pytest.raises(da.DaachorseError) as ei:
        da.DoubleArrayAhoCorasick.deserialize(bytes(pins["invalid_blob"]["blob"]))
    assert ei.value.code == 4
    # truncated / corrupted blobs are rejected, never crash
    blob = orc.OraclePma.build(["abba", "baaba", "ababa"]).serialize()
    for cut in (0, 3, 4, 17, len(blob) - 1):
        with pytest.raises(da.DaachorseError):
            da.DoubleArrayAhoCorasick.deserialize(blob[:cut])
    bad = bytearray(blob)
    bad[4:8] = (0x7FFFFFFF).to_bytes(4, "little")  # base of state 0 out of range
    with pytest.raises(da.DaachorseError):
        da.DoubleArrayAhoCorasick.deserialize(bytes(bad))


def test_info_matches_reference_pins(pins):
    for e in pins["heap_bytes"]:
        pma, _ = da.DoubleArrayAhoCorasick.deserialize(orc.OraclePma.build(e["patterns"]).serialize())
        assert pma.heap_bytes() == e["heap_bytes"]
    for e in pins["num_states"]:
        pma, _ = da.DoubleArrayAhoCorasick.deserialize(orc.OraclePma.build(e["patterns"]).serialize())
        assert pma.num_states() == e["num_states"]


def test_from_parts_equals_blob():
    o = orc.OraclePma.build(["he", "she", "his", "hers"])
    a = da.DoubleArrayAhoCorasick.from_parts(0, o.num_states, o.outputs(), states=o.states())
    assert a.serialize() == o.serialize()
    o = orc.OraclePma.build(["he", "she", "his", "hers"], kind="LeftmostLongest")
    a = da.DoubleArrayAhoCorasick.from_parts(1, o.num_states, o.outputs(), leftmost_states=o.leftmost_states(), fails=o.fails())
    assert a.serialize() == o.serialize()


def test_from_parts_rejects_short_or_missing_arrays():
    """the C ABI copies n_lstates entries from `fails`: a short array must not be read past its end"""
    import ctypes as C
    from daachorse_amd import _ffi
    o = orc.OraclePma.build(["he", "she", "his", "hers"], kind="LeftmostLongest")
    with pytest.raises(da.DaachorseError) as ei:
        da.DoubleArrayAhoCorasick.from_parts(1, o.num_states, o.outputs(), leftmost_states=o.leftmost_states(), fails=o.fails()[:-1])
    assert ei.value.code == 1
    h = C.c_void_p()
    ls = np.ascontiguousarray(o.leftmost_states(), dtype=np.uint32)
    ou = np.ascontiguousarray(o.outputs(), dtype=np.uint32)
    st = _ffi.lib().daac_bytewise_from_parts(None, 0, ls.ctypes.data, None, len(ls), ou.ctypes.data, len(ou), 1, o.num_states, C.byref(h))
    assert st == 1 and not h.value  # NULL `fails` with a non-zero count
    st = _ffi.lib().daac_bytewise_from_parts(None, 5, None, None, 0, ou.ctypes.data, len(ou), 0, o.num_states, C.byref(h))
    assert st == 1 and not h.value


def test_numpy_haystack_must_be_bytes():
    """a non-uint8 array is rejected, not value-cast (torch tensors are checked the same way)"""
    p = da.DoubleArrayAhoCorasick.new(["ab"])
    for bad in (np.arange(8, dtype=np.int32), np.ones(4, dtype=np.float32)):
        with pytest.raises(da.DaachorseError) as ei:
            p.scan_count(da.ScanMode.FindOverlapping, bad)
        assert ei.value.code == 1


@pytest.fixture(scope="module")
def repack_check(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("native") / "repack_check")
    csrc = os.path.join(ROOT, "daachorse_amd", "csrc")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "native", "repack_check.cpp"),
                           os.path.join(csrc, "pma.cpp"), os.path.join(csrc, "repack.cpp")])
    return exe


def _run_check(exe, tmp_path, patterns, hay, budget, depth=-1):
    blob = tmp_path / "a.blob"
    h = tmp_path / "h.bin"
    blob.write_bytes(orc.OraclePma.build(patterns).serialize())
    np.asarray(hay, dtype=np.uint8).tofile(h)
    out = subprocess.check_output([exe, str(blob), str(budget), str(depth), str(h)]).decode()
    return out


def test_repack_tables_follow_reference_delta(repack_check, tmp_path):
    """Tier tables walked with the kernel's step rule land on the reference's state, byte for byte."""
    rng = np.random.default_rng(1)
    # tiny alphabets, deep failure chains, the empty pattern, duplicates
    cases = [(["a", "ab", "bab", "bc", "bca", "c", "caa"], b"abc"),
             (["", "a", "aa", "aaa"], b"ab"),
             (["abcabcabd", "bcabd", "cab", "ab", "ab"], b"abcd"),
             (synth.patterns_cfg2(200), synth.ALPHA_LOWER)]
    for pats, alpha in cases:
        hay = rng.choice(np.frombuffer(alpha, dtype=np.uint8), size=20000)
        for budget in (1024, 4096, 98304):
            for depth in (-1, 0, 1):
                out = _run_check(repack_check, tmp_path, pats, hay, budget, depth)
                assert out.startswith("OK") or out.startswith("UNAVAILABLE"), out
    out = _run_check(repack_check, tmp_path, pats, hay, 98304)
    assert out.startswith("OK")
    # more than 31 distinct pattern bytes: the tiered engine must decline (DARRAY takes over)
    wide = [bytes([i, i + 1]) for i in range(40)]
    assert _run_check(repack_check, tmp_path, wide, hay, 98304).startswith("UNAVAILABLE")


def test_repack_cfg3_dictionary(repack_check, tmp_path):
    pats = synth.patterns_cfg3(20000)
    hay = synth.uniform_haystack(1 << 18, synth.SEEDS["cfg3_hay"], synth.ALPHA_LOWER_SPACE)
    out = _run_check(repack_check, tmp_path, pats, hay, 98304)
    assert out.startswith("OK"), out


@pytest.fixture(scope="module")
def gram_check(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("native") / "gram_check")
    csrc = os.path.join(ROOT, "daachorse_amd", "csrc")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "native", "gram_check.cpp"),
                           os.path.join(csrc, "pma.cpp"), os.path.join(csrc, "repack.cpp"), os.path.join(csrc, "gram.cpp")])
    return exe


def test_gram_tables_reproduce_the_match_stream(gram_check, tmp_path):
    """count + checksum evaluated from the k-gram tables (the kernel's rule) == literal automaton walk"""
    rng = np.random.default_rng(2)
    cases = [(["a", "ab", "bab", "bc", "bca", "c", "caa", "abcabcab", "bb", "bb"], b"abc"),
             (["abcabcabd", "bcabd", "cab", "ab", "ab", "dddddddd"], b"abcd"),
             (synth.patterns_cfg2(300), synth.ALPHA_LOWER),
             (synth.patterns_cfg3(20000), synth.ALPHA_LOWER_SPACE)]
    for pats, alpha in cases:
        blob = tmp_path / "a.blob"
        blob.write_bytes(orc.OraclePma.build(pats).serialize())
        h = tmp_path / "h.bin"
        rng.choice(np.frombuffer(alpha, dtype=np.uint8), size=60000).tofile(h)
        for budget in (160000, 9000):
            out = subprocess.check_output([gram_check, str(blob), str(budget), str(h)]).decode()
            assert out.startswith("OK"), out
    # text made of the dictionary's words: the walkers meet the tail records of round 6 (one path, one pattern end, h inline) — also words cut
    # short, a word's last letters replaced, and duplicates / nested words (paths with two ends keep their per-letter records)
    pats = synth.patterns_cfg3(20000) + [b"abcdefghijk", b"abcdefghijklm", b"abcdefgh", b"zzzzzzzzzzzzzzzz", b"zzzzzzzzzzzzzzzz"]
    blob.write_bytes(orc.OraclePma.build(pats).serialize())
    soup = synth.wordsoup_haystack(120000, synth.SEEDS["cfg3_dense"], pats, 20).copy()
    soup[rng.integers(0, len(soup), size=3000)] = ord("q")
    soup[:40] = np.frombuffer(b"abcdefghijklm abcdefghijk abcdefghijklmn", dtype=np.uint8)
    soup.tofile(h)
    for budget in (160000, 9000):
        out = subprocess.check_output([gram_check, str(blob), str(budget), str(h)]).decode()
        f = dict(kv.split("=") for kv in out.split()[2:])
        assert out.startswith("OK") and int(f["tail_records"]) > 5000 and int(f["tails_met"]) > 2000, out
    # "" as a pattern: declined
    blob.write_bytes(orc.OraclePma.build(["", "a"]).serialize())
    assert subprocess.check_output([gram_check, str(blob), "160000", str(h)]).decode().startswith("UNAVAILABLE")


@pytest.fixture(scope="module")
def gram2_check(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("native") / "gram2_check")
    csrc = os.path.join(ROOT, "daachorse_amd", "csrc")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "native", "gram2_check.cpp"),
                           os.path.join(csrc, "pma.cpp"), os.path.join(csrc, "repack.cpp"), os.path.join(csrc, "gram2.cpp")])
    return exe


def test_gram2_tables_reproduce_the_match_stream(gram2_check, tmp_path):
    """the second GRAM table set (one M word per position: continuation bits + short-pattern count; CID -> H for the
    checksum) evaluated with the kernel's rules == literal automaton walk; K = 3 and K = 2; duplicates decline"""
    rng = np.random.default_rng(3)
    cases = [(["a", "ab", "bab", "bc", "bca", "c", "caa", "abcabcab", "bb"], b"abc", "OK"),
             (["abcabcabd", "bcabd", "cab", "ab", "dddddddd"], b"abcd", "OK"),
             (synth.patterns_cfg1(), synth.ALPHA_ABCD, "OK"),
             (synth.patterns_cfg2(300), synth.ALPHA_LOWER, "OK"),
             (synth.patterns_cfg3(20000), synth.ALPHA_LOWER_SPACE, "OK")]
    for pats, alpha, want in cases:
        blob = tmp_path / "a.blob"
        blob.write_bytes(orc.OraclePma.build(pats).serialize())
        h = tmp_path / "h.bin"
        rng.choice(np.frombuffer(alpha, dtype=np.uint8), size=60000).tofile(h)
        for budget in (160000, 9000):
            out = subprocess.check_output([gram2_check, str(blob), str(budget), str(h)]).decode()
            assert out.startswith(want), out
    # word-soup text: deep walks
    pats = synth.patterns_cfg3(20000)
    blob.write_bytes(orc.OraclePma.build(pats).serialize())
    synth.wordsoup_haystack(100000, synth.SEEDS["cfg3_dense"], pats, 20).tofile(h)
    out = subprocess.check_output([gram2_check, str(blob), "160000", str(h)]).decode()
    assert out.startswith("OK") and "K=3" in out, out
    # "" as a pattern, four copies of one short pattern (count bits overflow), 31 distinct bytes: declined
    for pats in (["", "a"], ["ab"] * 4 + ["abc"], [bytes([65 + i, 66 + i]) for i in range(31)]):
        blob.write_bytes(orc.OraclePma.build(pats).serialize())
        assert subprocess.check_output([gram2_check, str(blob), "160000", str(h)]).decode().startswith("UNAVAILABLE"), pats
    # duplicates below the limit are fine
    blob.write_bytes(orc.OraclePma.build(["ab", "ab", "b", "abab"]).serialize())
    np.frombuffer(b"abababbab" * 50, dtype=np.uint8).tofile(h)
    assert subprocess.check_output([gram2_check, str(blob), "160000", str(h)]).decode().startswith("OK")


def test_gram4_tables_reproduce_the_count(tmp_path_factory, tmp_path):
    """the `.count()` tables of round 5 (gram4.hpp: "no pattern" as the last class, arithmetic class map where the dictionary's
    bytes are one range, per-word rank directory, "ends a pattern" in bit 30) walked with the rules of gram4_kernels.hip — plain
    records, tail records from the hit record on, and (round 6) the hits sent through the filter of gram4_filter.hpp first: every hit
    that ends a pattern or goes on passes, the count is the same — == literal automaton walk"""
    exe = str(tmp_path_factory.mktemp("native") / "gram4_check")
    csrc = os.path.join(ROOT, "daachorse_amd", "csrc")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "native", "gram4_check.cpp"),
                           os.path.join(csrc, "pma.cpp"), os.path.join(csrc, "repack.cpp"), os.path.join(csrc, "gram2.cpp"), os.path.join(csrc, "gram4.cpp")])
    rng = np.random.default_rng(5)
    gapped = [bytes(rng.choice(np.frombuffer(b"acegikmoqsuwy", dtype=np.uint8), size=int(rng.integers(1, 9)))) for _ in range(400)]  # not one byte range
    gapped = sorted(set(gapped))
    cases = [(["a", "ab", "bab", "bc", "bca", "c", "caa", "abcabcab", "bb"], b"abc ", 1),
             (["abcabcabd", "bcabd", "cab", "ab", "dddddddd"], b"abcd`e", 1),
             (synth.patterns_cfg1(), synth.ALPHA_ABCD, 1),
             (synth.patterns_cfg2(300), synth.ALPHA_LOWER, 1),
             (synth.patterns_cfg3(20000), synth.ALPHA_LOWER_SPACE, 1),
             (gapped, b"abcdefghijklmnopqrstuvwxyz{ ", 0),
             ([b"\x00\x01\x01", b"\x01", b"\x02\x00\x01\x02\x02"], b"\x00\x01\x02\x03\xff", 1),
             ([b"\xff\xfe", b"\xfe\xfe\xfd\xff", b"\xfd"], b"\xfc\xfd\xfe\xff\x00", 1)]
    blob, h = tmp_path / "a.blob", tmp_path / "h.bin"
    for pats, alpha, arith in cases:
        blob.write_bytes(orc.OraclePma.build(pats).serialize())
        rng.choice(np.frombuffer(alpha, dtype=np.uint8), size=60000).tofile(h)
        for budget in (160000, 9000):
            for fbytes in (33000, 600):   # (the filter of round 6: as the upload sizes it, and squeezed — no false negatives either way)
                out = subprocess.check_output([exe, str(blob), str(budget), str(h), str(fbytes)]).decode()
                assert out.startswith("OK") and f"arith={arith}" in out, (out, pats[:3])
    pats = synth.patterns_cfg3(20000)
    blob.write_bytes(orc.OraclePma.build(pats).serialize())
    synth.wordsoup_haystack(100000, synth.SEEDS["cfg3_dense"], pats, 20).tofile(h)
    out = subprocess.check_output([exe, str(blob), "160000", str(h)]).decode()
    assert out.startswith("OK") and "K=3" in out and "arith=1 lo=97" in out and "filter=1" in out, out
    rng.choice(np.frombuffer(synth.ALPHA_LOWER_SPACE, dtype=np.uint8), size=400000).tofile(h)   # uniform text: most hits neither end a pattern nor go on
    out = subprocess.check_output([exe, str(blob), "160000", str(h)]).decode()
    f = dict(kv.split("=") for kv in out.split()[1:])
    assert out.startswith("OK") and int(f["useful"]) <= int(f["passed"]) < int(f["hits"]) // 2, out
    blob.write_bytes(orc.OraclePma.build(["ab", "ab", "b", "abab"]).serialize())
    np.frombuffer(b"abababbab" * 50, dtype=np.uint8).tofile(h)
    assert subprocess.check_output([exe, str(blob), "160000", str(h)]).decode().startswith("OK")


def test_gram2w_tables_reproduce_the_match_stream(tmp_path):
    """the wide-alphabet GRAM tables (31 .. 62 byte classes, 64-bit words, K = 2) walked with the kernel's rules == literal
    automaton walk; at most 61 pattern bytes"""
    exe = str(tmp_path / "gram2w_check")
    csrc = os.path.join(ROOT, "daachorse_amd", "csrc")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "native", "gram2w_check.cpp"),
                           os.path.join(csrc, "pma.cpp"), os.path.join(csrc, "repack.cpp"), os.path.join(csrc, "gram2w.cpp")])
    blob, h = tmp_path / "a.blob", tmp_path / "h.bin"
    wide = synth.patterns_cfg3_wide(20000)
    assert len({b for w in wide for b in w}) == 60
    rng = np.random.default_rng(8)
    some = [bytes(rng.integers(33, 94, size=int(rng.integers(1, 9))).astype(np.uint8)) for _ in range(400)]  # 61 distinct bytes
    for pats, hay in ((wide, synth.uniform_haystack(100000, 3, synth.ALPHA_WIDE_SPACE)),
                      (wide, synth.wordsoup_haystack(100000, 5, wide, 20, alphabet=synth.ALPHA_WIDE)),
                      (list(dict.fromkeys(some)), rng.integers(30, 97, size=60000).astype(np.uint8)),
                      (synth.patterns_cfg1(), synth.uniform_haystack(5000, 1, synth.ALPHA_ABCD))):
        blob.write_bytes(orc.OraclePma.build(pats).serialize())
        np.asarray(hay, dtype=np.uint8).tofile(h)
        out = subprocess.check_output([exe, str(blob), "147000", str(h)]).decode()
        assert out.startswith("OK"), out
    blob.write_bytes(orc.OraclePma.build([bytes([i]) for i in range(33, 33 + 62)]).serialize())  # 62 pattern bytes: one too many
    assert subprocess.check_output([exe, str(blob), "147000", str(h)]).decode().startswith("UNAVAILABLE")


def test_pfx_tables_reproduce_the_match_count(tmp_path):
    """the PFX tables (Bloom bitmap over hashed G-byte prefixes, hash-and-displace slots, goto-only walk records, one-byte
    counts) walked with the kernel's rules == literal automaton walk, for dictionaries over any byte alphabet"""
    exe = str(tmp_path / "pfx_check")
    csrc = os.path.join(ROOT, "daachorse_amd", "csrc")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "native", "pfx_check.cpp"),
                           os.path.join(csrc, "pma.cpp"), os.path.join(csrc, "repack.cpp"), os.path.join(csrc, "pfx.cpp")])
    blob, h = tmp_path / "a.blob", tmp_path / "h.bin"
    rng = np.random.default_rng(12)
    binp = synth.patterns_binary256(20000)
    assert len({b for w in binp for b in w}) == 256
    jp = synth.patterns_cfg5(5000)
    mixed = list(dict.fromkeys([bytes(rng.integers(0, 256, size=int(rng.integers(1, 9))).astype(np.uint8)) for _ in range(3000)]))  # incl. one-byte patterns
    dup = [b"ab", b"ab", b"b", b"abab", b"bababab", b"\xff\x00", b"\xff\x00"]
    cases = ((binp, rng.integers(0, 256, size=200000).astype(np.uint8), "G=3"),
             (binp, np.frombuffer(b"".join(binp[i] for i in rng.integers(0, len(binp), size=20000).tolist()), dtype=np.uint8), "G=3"),
             (jp, synth.zipf_text(48 * 4000), "G=6"),
             (mixed, rng.integers(0, 256, size=100000).astype(np.uint8), "len1=1"),
             (dup, np.frombuffer(b"abababbab\xff\x00" * 3000, dtype=np.uint8), "G=2"),
             (synth.patterns_cfg3(20000), synth.wordsoup_haystack(100000, 5, synth.patterns_cfg3(20000), 20), "G=2"))
    for pats, hay, expect in cases:
        blob.write_bytes(orc.OraclePma.build(pats).serialize())
        np.asarray(hay, dtype=np.uint8).tofile(h)
        out = subprocess.check_output([exe, str(blob), "120000", str(h)]).decode()
        assert out.startswith("OK") and expect in out, out
    for pats in (["", "a"], ["a", "b"]):  # "" in the set; one-byte patterns only
        blob.write_bytes(orc.OraclePma.build(pats).serialize())
        assert subprocess.check_output([exe, str(blob), "120000", str(h)]).decode().startswith("UNAVAILABLE"), pats

def test_emit_tables_reproduce_the_tuple_stream(tmp_path):
    """the tuple-emission tables (flag bits + value tables for short patterns, ehit / erec for deep ones) walked with the
    emitter's rules give the literal automaton's (start, end, value) list, order included"""
    exe = str(tmp_path / "emit_check")
    csrc = os.path.join(ROOT, "daachorse_amd", "csrc")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "native", "emit_check.cpp"),
                           os.path.join(csrc, "pma.cpp"), os.path.join(csrc, "repack.cpp"), os.path.join(csrc, "gram2.cpp")])
    rng = np.random.default_rng(4)
    pats3 = synth.patterns_cfg3(20000)
    cases = [(["a", "ab", "bab", "bc", "bca", "c", "caa", "abcabcab", "bb", "cabcabcab"], rng.choice(np.frombuffer(b"abc", dtype=np.uint8), size=30000)),
             (synth.patterns_cfg1(), synth.uniform_haystack(5000, 1, synth.ALPHA_ABCD)),
             (synth.patterns_cfg2(300), synth.wordsoup_haystack(60000, 9, synth.patterns_cfg2(300), 13, noise_256=40)),
             (pats3, synth.uniform_haystack(100000, synth.SEEDS["cfg3_hay"], synth.ALPHA_LOWER_SPACE)),
             (pats3, synth.wordsoup_haystack(100000, synth.SEEDS["cfg3_dense"], pats3, 20))]
    blob, h = tmp_path / "a.blob", tmp_path / "h.bin"
    for pats, hay in cases:
        blob.write_bytes(orc.OraclePma.build(pats).serialize())
        np.asarray(hay, dtype=np.uint8).tofile(h)
        for budget in (147000, 9000):
            out = subprocess.check_output([exe, str(blob), str(budget), str(h)]).decode()
            assert out.startswith("OK"), out
    # patterns longer than K + 16 bytes and duplicates among the longer patterns: served (the kernel places them as "extras")
    long_words = pats3[:3000] + [w + b"ological" * 3 for w in pats3[:40]] + [pats3[7] * 2] * 3 + [pats3[9] + b"xx"] * 2
    soup = b" ".join(long_words[i] for i in rng.integers(0, len(long_words), size=9000).tolist())
    blob.write_bytes(orc.OraclePma.build(long_words).serialize())
    np.frombuffer(soup, dtype=np.uint8).tofile(h)
    out = subprocess.check_output([exe, str(blob), "147000", str(h)]).decode()
    assert out.startswith("OK") and "maxlen=" in out and int(out.split("maxlen=")[1]) > 19, out
    # duplicates among the patterns of at most K bytes: no emission tables (the segment scanners serve)
    for pats in (["ab", "ab", "abc"], ["a", "a", "abcd"]):
        blob.write_bytes(orc.OraclePma.build(pats).serialize())
        assert subprocess.check_output([exe, str(blob), "147000", str(h)]).decode().startswith("UNAVAILABLE"), pats


def test_synth_definitions_are_stable():
    """Seeds and generators are part of the benchmark definition: pin a few bytes/patterns."""
    h = synth.uniform_haystack(64, synth.SEEDS["cfg3_hay"], synth.ALPHA_LOWER_SPACE)
    assert synth.uniform_haystack(40, synth.SEEDS["cfg3_hay"], synth.ALPHA_LOWER_SPACE, offset=13).tobytes() == h[13:53].tobytes()
    p2 = synth.patterns_cfg2(50)
    assert len(set(p2)) == 50 and all(4 <= len(p) <= 12 for p in p2)
    p3 = synth.patterns_cfg3(3000)
    assert len(set(p3)) == 3000 and all(2 <= len(p) <= 16 for p in p3)
    ws = synth.wordsoup_haystack(400, synth.SEEDS["cfg3_dense"], p3, 20)
    assert synth.wordsoup_haystack(100, synth.SEEDS["cfg3_dense"], p3, 20, offset=137).tobytes() == ws[137:237].tobytes()


@pytest.fixture(scope="module")
def cpp_facade(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("native") / "cpp_facade_test")
    libdir = os.path.join(ROOT, "daachorse_amd", "lib")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "native", "cpp_facade_test.cpp"),
                           "-L" + libdir, "-ldaachorse_amd", "-Wl,-rpath," + libdir])
    return exe


def test_cpp_facade_host(cpp_facade):
    """include/daachorse_amd.hpp compiles against the C ABI and behaves like the crate on the host side."""
    assert subprocess.check_output([cpp_facade, "host"]).decode().strip() == "OK host"


def test_matchkind_mismatch_needs_no_device(pins):
    """The crate panics on a MatchKind mismatch before looking at the haystack; so does the boundary."""
    for e in pins["matchkind_mismatch"]["must_fail"]:
        pma = da.DoubleArrayAhoCorasickBuilder().match_kind(da.MatchKind[e["kind"]]).build(["a"])
        with pytest.raises(da.DaachorseError) as ei:
            list(getattr(pma, e["api"])(""))
        assert ei.value.code == 5, e


def test_scan_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    pma, _ = da.DoubleArrayAhoCorasick.deserialize(orc.OraclePma.build(["a"]).serialize())
    with pytest.raises(da.DaachorseError) as ei:
        pma.scan_count(da.ScanMode.FindOverlapping, "aaa")
    assert ei.value.code == 7


def test_options_need_no_device():
    """daac_set_option / daac_pma_set_option (include/daachorse_amd.h, the list is DAAC_OPTIONS in api_internal.hpp): names, aliases, what a handle
    may not override, and that a handle without tables takes every option — none of it touches a device.  (An upload-time option on a handle
    that HAS tables answers 6: tests/test_gpu_parity.py.)"""
    import ctypes as C
    from daachorse_amd import _ffi
    for name in ("gram_ppl", "gram_tail", "gram3_tail", "gram4_filter", "gram_version", "find3_window", "workspace_keep", "threads"):
        da.set_option(name, {"gram_tail": -1, "gram3_tail": -1, "gram4_filter": 1, "find3_window": 1 << 30, "workspace_keep": 8 << 30, "threads": 1024}.get(name, 0))
    da.set_option("gram_version", 3)   # ABI 4's name of the `.count()` kernel: an alias of 4
    da.set_option("gram_version", 0)
    for gone in ("emit_version", "emit_tiles", "emit_rec_cap", "restart_tier", "emit_stagger", "emit_v3_lds", "no_such_option"):
        with pytest.raises(da.DaachorseError) as ei:
            da.set_option(gone, 1)
        assert ei.value.code == 1, gone
    p, _ = da.DoubleArrayAhoCorasick.deserialize(orc.OraclePma.build(["ab", "b"]).serialize())
    p.set_option("gram_lds_budget", 9216).set_option("pfx", 2).set_option("gram_ppl", 16).set_option("gram4_filter", 0)   # before any upload: all fine
    p.set_option("gram_lds_budget").set_option("gram_ppl")                                                               # ... and taken away again
    for name in ("pool", "pool_keep"):
        with pytest.raises(da.DaachorseError) as ei:
            p.set_option(name, 0)
        assert ei.value.code == 1
    with pytest.raises(da.DaachorseError) as ei:
        p.set_option("no_such_option", 1)
    assert ei.value.code == 1
    assert _ffi.lib().daac_pma_set_option(None, b"gram_ppl", 16, 0) == 1
    assert _ffi.lib().daac_abi_version() == _ffi.ABI_VERSION == 6
    assert isinstance(da.last_kernel(), str)

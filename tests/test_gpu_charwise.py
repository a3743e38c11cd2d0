"""Parity of the charwise HIP scan path (through the C ABI) with the CPU oracle — runs on the MI355X.

Bit-exact (start, end, value) tuples in the reference's order for the four iterators of
src/charwise/iter.rs, over the reference's own vector tables (ASCII), its in-module multi-byte
pins, and seeded multi-byte fuzz with segment / window / shard cuts falling inside characters.
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, iter_vector_runs
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

import daachorse_amd as da
from daachorse_amd import ScanMode

API_MODE = {"find_overlapping_iter": ScanMode.FindOverlapping,
            "find_overlapping_no_suffix_iter": ScanMode.FindOverlappingNoSuffix,
            "find_iter": ScanMode.Find, "leftmost_find_iter": ScanMode.LeftmostFind}
APIS_OF_KIND = {0: ["find_overlapping_iter", "find_overlapping_no_suffix_iter", "find_iter"], 1: ["leftmost_find_iter"],
                2: ["leftmost_find_iter"]}

ALPHABETS = {
    "ascii": [chr(c) for c in range(0x61, 0x65)],
    "mixed": list("abé¢д") + ["全", "世", "界", "\U0001F600", "\U00010348"],
    "kana": [chr(c) for c in range(0x3041, 0x3049)],
}


@pytest.fixture(scope="module")
def cpins():
    with open(os.path.join(GOLDEN, "charwise_pins.json")) as f:
        return json.load(f)


# (launch shapes and engine switches are set on the HANDLE under test, daac_pma_set_option: nothing process-wide to reset)


def _pair(patterns, kind=0, values=None):
    kind = orc.KIND.get(kind, kind)
    o = orc.OracleCharwisePma.build(patterns, values=values, kind=kind)
    b = da.CharwiseDoubleArrayAhoCorasickBuilder().match_kind(kind)
    p = b.build_with_values(zip(patterns, values)) if values is not None else b.build(patterns)
    assert p.serialize() == o.serialize()
    return o, p


def _sev(m):
    return [(int(x["start"]), int(x["end"]), int(x["value"])) for x in m]


def _oracle_or_diverged(o, api, hay):
    try:
        return getattr(o, api)(hay)
    except orc.OracleError as e:
        assert e.code == 6  # ORC_ERR_DIVERGED: the reference itself would not terminate
        return None


def _check_all_forms(o, p, api, hay, what):
    """eager scan, count + checksum, lazy iterator — against the oracle's stream"""
    mode = API_MODE[api]
    want = _oracle_or_diverged(o, api, hay)
    if want is None:
        with pytest.raises(da.DaachorseError) as ei:
            p.scan(mode, hay)
        assert ei.value.code == 6, what
        with pytest.raises(da.DaachorseError) as ei:
            p.scan_count(mode, hay)
        assert ei.value.code == 6, what
        return False
    got = p.scan(mode, hay)
    assert _sev(got) == _sev(want), what
    assert p.scan_count(mode, hay) == (len(want), orc.matches_checksum(want)), what
    lazy = [(m.start(), m.end(), m.value()) for m in getattr(p, api)(hay)]
    assert lazy == _sev(want), what
    return True


def test_golden_vector_tables_all_four_iterators(vectors):
    """search_*_charwise runners of tests/aho_corasick_crate_test.rs:592-645 that go through an iterator."""
    n = 0
    for runner, case in iter_vector_runs(vectors):
        if runner["api"] not in API_MODE:
            continue
        o, p = _pair(case["patterns"], runner["kind"])
        want = [tuple(t) for t in case["matches"]]
        got = p.scan(API_MODE[runner["api"]], case["haystack"])
        assert [(int(m["value"]), int(m["start"]), int(m["end"])) for m in got] == want, (runner, case["name"])
        for seg in (0, 16):  # default plan, and one lane per 16 bytes
            p.set_option("seg_bytes", seg)
            assert _check_all_forms(o, p, runner["api"], case["haystack"], (runner, case["name"], seg))
        n += 1
    assert n == 61 + 57 + 93 + 91


def test_multibyte_pins(cpins):
    """in-module tests of src/charwise.rs / src/charwise/iter.rs with multi-byte text"""
    for ka in cpins["multibyte_zero_length"] + cpins["known_answers"]:
        if ka["api"] not in API_MODE:
            continue
        _, p = _pair(ka["patterns"], ka["kind"])
        for seg in (0, 16):
            p.set_option("seg_bytes", seg)
            assert _sev(p.scan(API_MODE[ka["api"]], ka["haystack"])) == [tuple(t) for t in ka["matches_sev"]], ka["cite"]


def _words(rng, n, alphabet, max_chars):
    return ["".join(alphabet[i] for i in rng.integers(0, len(alphabet), size=int(rng.integers(1, max_chars + 1)))) for _ in range(n)]


@pytest.mark.parametrize("chain", [1, 0])
@pytest.mark.parametrize("alpha", sorted(ALPHABETS))
def test_fuzz_multibyte_all_iterators(alpha, chain):
    """random dictionaries and text; 16- and 48-byte lanes cut characters, halos start inside characters;
    the restart iterators both ways (chain = 1: speculate / reconcile / emit, 0: sync-point scanners)"""
    rng = np.random.default_rng(len(alpha) * 7 + 1)
    A = ALPHABETS[alpha]
    for trial in range(10):
        pats = _words(rng, int(rng.integers(1, 60)), A, 5)
        text = "".join(A[i] for i in rng.integers(0, len(A), size=int(rng.integers(0, 3000))))
        for kind in (0, 1, 2):
            o, p = _pair(pats, kind)
            p.set_option("restart_chain", chain)
            for api in APIS_OF_KIND[kind]:
                for seg in (0, 16, 48):
                    p.set_option("seg_bytes", seg)
                    assert _check_all_forms(o, p, api, text, (alpha, trial, kind, api, seg))


def test_fuzz_empty_pattern_in_the_set():
    """"" matches at every character boundary (find*, iter.rs:115-131, 178-190) and, under leftmost kinds, wherever no
    longer match starts (iter.rs:311-318).  Single-width alphabets: with mixed widths the reference advances by the
    width of the character that ended the walk (iter.rs:347) and leaves the character grid."""
    rng = np.random.default_rng(77)
    ran = unsupported = 0
    for alpha in ("ascii", "kana"):
        A = ALPHABETS[alpha]
        for trial in range(12):
            pats = _words(rng, int(rng.integers(0, 12)), A, 4)
            pats.insert(int(rng.integers(0, len(pats) + 1)), "")
            text = "".join(A[i] for i in rng.integers(0, len(A), size=int(rng.integers(0, 400))))
            for kind in (0, 1, 2):
                o, p = _pair(pats, kind)
                for api in APIS_OF_KIND[kind]:
                    for seg in (0, 16):
                        p.set_option("seg_bytes", seg)
                        ok = _check_all_forms(o, p, api, text, (alpha, trial, kind, api, seg))
                        ran += ok
                        unsupported += not ok
    assert ran > 200 and unsupported < ran


def test_unmapped_characters_and_device_haystacks():
    import torch
    pats = ["世界", "界中", "全世界", "に", "ab", "b"]
    text = "全世界中にzzab世界☃界中\U0001F600bに" * 50
    raw = text.encode()
    for kind in (0, 1):
        o, p = _pair(pats, kind)
        for api in APIS_OF_KIND[kind]:
            want = getattr(o, api)(text)
            for shift in (0, 1, 5):  # unaligned device address
                buf = torch.zeros(len(raw) + 32, dtype=torch.uint8, device="cuda")
                buf[shift:shift + len(raw)] = torch.frombuffer(bytearray(raw), dtype=torch.uint8).cuda()
                view = buf[shift:shift + len(raw)]
                assert _sev(p.scan(API_MODE[api], view)) == _sev(want), (kind, api, shift)
                assert p.scan_count(API_MODE[api], view) == (len(want), orc.matches_checksum(want))


def _dictionary(rng, n):
    base = [chr(c) for c in range(0x3041, 0x3097)] + [chr(c) for c in range(0x4E00, 0x4E00 + 400)] + list("abcdefgh")
    freq = 1.0 / np.arange(1, len(base) + 1)
    freq /= freq.sum()
    words = set()
    while len(words) < n:
        k = int(rng.integers(1, 7))
        words.add("".join(base[i] for i in rng.choice(len(base), size=k, p=freq)))
    return sorted(words), base, freq


@pytest.mark.parametrize("map_lds", [0, 1])
def test_dictionary_scale_text(map_lds):
    """config-5-shaped case: a 20 k-word dictionary, 2 MB of text made of dictionary words and noise
    (map_lds = 1: the chain scanners with the code mapper staged in LDS; read when the automaton is uploaded)"""
    rng = np.random.default_rng(5)
    words, base, freq = _dictionary(rng, 20000)
    parts = []
    size = 0
    while size < 2_000_000:
        w = words[int(rng.integers(0, len(words)))] if rng.random() < 0.7 else "".join(base[i] for i in rng.choice(len(base), size=3, p=freq))
        parts.append(w)
        size += len(w) * 3
    text = "".join(parts)
    for kind, apis in ((0, ["find_overlapping_iter", "find_iter"]), (1, ["leftmost_find_iter"])):
        o, p = _pair(words, kind)
        p.set_option("char_map_lds", map_lds)   # (read at upload)
        for api in apis:
            want = getattr(o, api)(text)
            assert len(want) > 100000
            got = p.scan(API_MODE[api], text)
            assert np.array_equal(got["start"], want["start"]) and np.array_equal(got["end"], want["end"]) and \
                np.array_equal(got["value"], want["value"]), api
            assert p.scan_count(API_MODE[api], text) == (len(want), orc.matches_checksum(want))


def test_lazy_windows_and_shard_tails_cut_characters():
    rng = np.random.default_rng(11)
    A = ALPHABETS["mixed"]
    pats = _words(rng, 80, A, 4)
    text = "".join(A[i] for i in rng.integers(0, len(A), size=40000))
    raw = text.encode()
    for kind in (0, 1):
        o, p = _pair(pats, kind)
        p.set_option("iter_window", 4096)
        for api in APIS_OF_KIND[kind]:
            want = _sev(getattr(o, api)(text))
            assert [(m.start(), m.end(), m.value()) for m in getattr(p, api)(text)] == want, api
    # shards of the overlapping scan: every byte offset is a legal cut, also inside a character
    o, p = _pair(pats, 0)
    want = o.find_overlapping_iter(text)
    cuts = [0, 1, 2, 3, 1000, 1001, 1002, 20001, len(raw) - 1, len(raw)]
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        part = want[(want["end"] > lo) & (want["end"] <= hi)]
        if lo == 0:
            part = want[want["end"] <= hi]
        got = p.scan_count(ScanMode.FindOverlapping, raw[:hi], begin=lo)
        assert got == (len(part), orc.matches_checksum(part)), (lo, hi)


def test_engine_choice_is_checked():
    _, p = _pair(["ab"], 0)
    for eng in (da.Engine.Tiered, da.Engine.Gram):
        with pytest.raises(da.DaachorseError) as ei:
            p.scan(ScanMode.FindOverlapping, "abab", engine=eng)
        assert ei.value.code == 6
    assert _sev(p.scan(ScanMode.FindOverlapping, "abab", engine=da.Engine.DArray)) == [(0, 2, 0), (2, 4, 0)]

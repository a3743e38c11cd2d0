"""Pins the CPU oracle (oracle/daac_oracle.c) against the reference's own golden vectors.

Vectors: tests/golden/aho_corasick_vectors.json (from tests/aho_corasick_crate_test.rs:63-382)
and tests/golden/bytewise_pins.json (in-module layout pins and doc known answers)."""
import itertools

import numpy as np
import pytest

from conftest import iter_vector_runs
from oracle import oracle as orc


def _run(pma, api, hay):
    return getattr(pma, api)(hay)


def test_all_vector_runs(vectors):
    n = 0
    for runner, case in iter_vector_runs(vectors):
        pma = orc.OraclePma.build(case["patterns"], kind=runner["kind"])
        got = _run(pma, runner["api"], case["haystack"])
        got_vse = [(int(m["value"]), int(m["start"]), int(m["end"])) for m in got]
        want = [tuple(t) for t in case["matches"]]
        assert got_vse == want, (runner, case["name"])
        n += 1
    # 4 collections: 61/57/93/91 cases; 2 standard collections are run twice (iter + stepper)
    assert n == 61 * 2 + 57 * 2 + 93 + 91


def test_known_answers(pins):
    for ka in pins["known_answers"]:
        if "patvals" in ka:
            pats = [p for p, _ in ka["patvals"]]
            vals = [v for _, v in ka["patvals"]]
        else:
            pats, vals = ka["patterns"], None
        pma = orc.OraclePma.build(pats, values=vals, kind=ka["kind"])
        got = orc.triples_sev(_run(pma, ka["api"], ka["haystack"]))
        assert got == [tuple(t) for t in ka["matches_sev"]], ka["cite"]


def test_double_array_layout(pins):
    da = pins["double_array"]
    pma = orc.OraclePma.build([bytes(p) for p in da["patterns_bytes"]])
    st = pma.states()[:11]
    assert [int(b) or None for b in st[:, 0]] == da["base"]
    assert [int(x) & 0xFF for x in st[:, 2]] == da["check"]
    assert [int(x) for x in st[:, 1]] == da["fail"]


def test_num_states_heap_bytes(pins):
    for e in pins["num_states"]:
        assert orc.OraclePma.build(e["patterns"]).num_states == e["num_states"], e["cite"]
    for e in pins["heap_bytes"]:
        assert orc.OraclePma.build(e["patterns"]).heap_bytes() == e["heap_bytes"], e["cite"]


def test_input_order(pins):
    io = pins["input_order"]
    a = orc.OraclePma.build([p for p, _ in io["sorted"]], values=[v for _, v in io["sorted"]])
    b = orc.OraclePma.build([p for p, _ in io["unsorted"]], values=[v for _, v in io["unsorted"]])
    assert np.array_equal(a.states(), b.states())
    assert np.array_equal(a.outputs(), b.outputs())


def _gen_patterns(gen):
    pats = []
    for g in gen:
        for lo, hi in g["ranges"]:
            for i in range(lo, hi + 1):
                pats.append(bytes(g["prefix"] + [i]))
    return pats


def test_n_blocks(pins):
    for e in pins["n_blocks"]:
        pma = orc.OraclePma.build(_gen_patterns(e["gen"]))
        st = pma.states()
        assert pma.num_states == e["num_states"], e["name"]
        assert len(st) == e["states_len"], e["name"]
        for idx, base in e["base_of"].items():
            assert int(st[int(idx), 0]) == base, e["name"]


def test_empty_pattern_set():
    pma = orc.OraclePma.build([])
    for a in range(256):
        assert len(pma.find_overlapping_iter(bytes([a]))) == 0
    for a, b in itertools.product(range(0, 256, 5), range(256)):
        assert len(pma.find_overlapping_iter(bytes([a, b]))) == 0


def test_serialize_roundtrip_and_invalid(pins):
    for e in pins["serialize_roundtrip"]:
        pma = orc.OraclePma.build(e["patterns"], kind=e["kind"])
        blob = pma.serialize()
        other = orc.OraclePma.deserialize(blob)
        assert other.consumed == len(blob)
        assert np.array_equal(pma.states(), other.states())
        assert np.array_equal(pma.leftmost_states(), other.leftmost_states())
        assert np.array_equal(pma.fails(), other.fails())
        assert np.array_equal(pma.outputs(), other.outputs())
        assert pma.match_kind == other.match_kind and pma.num_states == other.num_states
        assert other.serialize() == blob
    with pytest.raises(orc.OracleError) as ei:
        orc.OraclePma.deserialize(bytes(pins["invalid_blob"]["blob"]))
    assert ei.value.code == 4


def test_matchkind_mismatch_and_invalid_option(pins):
    for e in pins["matchkind_mismatch"]["must_fail"]:
        pma = orc.OraclePma.build(["a"], kind=e["kind"])
        with pytest.raises(orc.OracleError) as ei:
            _run(pma, e["api"], "")
        assert ei.value.code == 5
    io = pins["invalid_option"]
    with pytest.raises(orc.OracleError):
        orc.OraclePma.build(io["patterns"], num_free_blocks=io["num_free_blocks"])


def test_threaded_count_matches_iterator():
    rng = np.random.default_rng(7)
    pats = ["ab", "bca", "c", "abcab", "bb"]
    pma = orc.OraclePma.build(pats)
    hay = rng.integers(97, 100, size=200_000, dtype=np.uint8)
    m = pma.find_overlapping_iter(hay)
    c1 = pma.overlapping_count(hay, threads=1)
    c4 = pma.overlapping_count(hay, threads=4)
    assert c1 == c4 == (len(m), orc.matches_checksum(m))

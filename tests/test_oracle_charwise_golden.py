"""Pins the charwise CPU oracle (oracle/daac_oracle_charwise.c) against the reference's own vectors:
the six charwise runners of tests/aho_corasick_crate_test.rs:592-645 over the shared tables, and the
in-module pins of src/charwise.rs, src/charwise/iter.rs, src/charwise/mapper.rs (tests/golden/charwise_pins.json)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, iter_vector_runs
from oracle import oracle as orc


@pytest.fixture(scope="module")
def cpins():
    with open(os.path.join(GOLDEN, "charwise_pins.json")) as f:
        return json.load(f)


def test_all_charwise_vector_runs(vectors):
    """same tables, same expected triples: byte offsets (all vectors are ASCII)"""
    n = 0
    for runner, case in iter_vector_runs(vectors):
        pma = orc.OracleCharwisePma.build(case["patterns"], kind=runner["kind"])
        got = getattr(pma, runner["api"])(case["haystack"])
        assert [(int(m["value"]), int(m["start"]), int(m["end"])) for m in got] == [tuple(t) for t in case["matches"]], (runner, case["name"])
        n += 1
    assert n == 61 * 2 + 57 * 2 + 93 + 91


def test_charwise_layout_pins(cpins):
    da = cpins["double_array"]
    st = orc.OracleCharwisePma.build(da["patterns"]).states()[:11]
    assert [int(b) or None for b in st[:, 0]] == da["base"]
    assert [int(x) for x in st[:, 1]] == da["check"]
    assert [int(x) for x in st[:, 2]] == da["fail"]
    for e in cpins["num_states"]:
        assert orc.OracleCharwisePma.build(e["patterns"]).num_states == e["num_states"], e["cite"]
    for e in cpins["num_elements"]:
        assert orc.OracleCharwisePma.build(e["patterns"]).num_elements() == e["num_elements"], e["cite"]
    for e in cpins["heap_bytes"]:
        assert orc.OracleCharwisePma.build(e["patterns"]).heap_bytes() == e["heap_bytes"], e["cite"]
    io = cpins["input_order"]
    a = orc.OracleCharwisePma.build([p for p, _ in io["sorted"]], values=[v for _, v in io["sorted"]])
    b = orc.OracleCharwisePma.build([p for p, _ in io["unsorted"]], values=[v for _, v in io["unsorted"]])
    assert np.array_equal(a.states(), b.states()) and np.array_equal(a.outputs(), b.outputs())
    for e in cpins["n_blocks"]:
        pats = ["".join(chr(c) for c in g["prefix"] + [i]) for g in e["gen"] for i in range(g["range"][0], g["range"][1] + 1)]
        pma = orc.OracleCharwisePma.build(pats)
        st = pma.states()
        assert pma.num_states == e["num_states"] and len(st) == e["states_len"], e["name"]
        for idx, base in e["base_of"].items():
            assert int(st[int(idx), 0]) == base, e["name"]


def test_charwise_known_answers(cpins):
    for ka in cpins["multibyte_zero_length"] + cpins["known_answers"]:
        pma = orc.OracleCharwisePma.build(ka["patterns"], kind=ka["kind"])
        got = orc.triples_sev(getattr(pma, ka["api"])(ka["haystack"]))
        assert got == [tuple(t) for t in ka["matches_sev"]], ka["cite"]


def test_decoder_and_mapper(cpins):
    d = cpins["decoder"]
    text = "".join(chr(c) for c in d["code_points"])
    # one single-character pattern per code point: every character is found, ending at the pinned offsets
    pma = orc.OracleCharwisePma.build([chr(c) for c in d["code_points"]])
    m = pma.find_overlapping_iter(text)
    assert [int(x["end"]) for x in m] == d["end_offsets"]
    assert [int(x["value"]) for x in m] == list(range(len(d["code_points"])))
    # mapper: frequency descending, code point ascending (patterns chosen to realise the pinned freqs)
    mp = cpins["mapper"]
    pats = []
    for c, f in enumerate(mp["freqs"]):
        pats += [chr(c)] * f
    table = orc.OracleCharwisePma.build(pats).table()
    got = [None if int(x) == 0xFFFFFFFF else int(x) for x in table]
    assert got == mp["codes"]


def test_charwise_serialize_roundtrip():
    for kind in ("Standard", "LeftmostLongest"):
        pma = orc.OracleCharwisePma.build(["全世界", "世界", "に", "abc"], kind=kind)
        blob = pma.serialize()
        other = orc.OracleCharwisePma.deserialize(blob)
        assert other.consumed == len(blob) and other.serialize() == blob
        assert np.array_equal(pma.states(), other.states())
    with pytest.raises(orc.OracleError):
        orc.OracleCharwisePma.deserialize(bytes(21))

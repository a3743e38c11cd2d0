"""The product's C++ host builder (csrc/builder.cpp) against the oracle's C restatement and the
reference's layout pins: byte-identical serialize() blobs for every match kind."""
import numpy as np
import pytest

from oracle import oracle as orc

import daachorse_amd as da
from daachorse_amd import MatchKind, synth

KINDS = [MatchKind.Standard, MatchKind.LeftmostLongest, MatchKind.LeftmostFirst]


def _blob(patterns, kind=MatchKind.Standard, nfb=16, values=None):
    b = da.DoubleArrayAhoCorasickBuilder().match_kind(kind).num_free_blocks(nfb)
    pma = b.build(patterns) if values is None else b.build_with_values(list(zip(patterns, values)))
    return pma.serialize()


def test_golden_pattern_sets_build_identically(vectors):
    seen = set()
    for table in vectors["tables"].values():
        for case in table:
            key = tuple(case["patterns"])
            if key in seen:
                continue
            seen.add(key)
            for kind in KINDS:
                assert _blob(case["patterns"], kind) == orc.OraclePma.build(case["patterns"], kind=int(kind)).serialize(), (key, kind)


def test_layout_pins(pins):
    da_pin = pins["double_array"]
    pma = da.DoubleArrayAhoCorasick.new([bytes(p) for p in da_pin["patterns_bytes"]])
    o = orc.OraclePma.deserialize(pma.serialize())
    st = o.states()[:11]
    assert [int(b) or None for b in st[:, 0]] == da_pin["base"]
    assert [int(x) & 0xFF for x in st[:, 2]] == da_pin["check"]
    assert [int(x) for x in st[:, 1]] == da_pin["fail"]
    for e in pins["heap_bytes"]:
        assert da.DoubleArrayAhoCorasick.new(e["patterns"]).heap_bytes() == e["heap_bytes"]
    for e in pins["num_states"]:
        assert da.DoubleArrayAhoCorasick.new(e["patterns"]).num_states() == e["num_states"]
    io = pins["input_order"]
    a = da.DoubleArrayAhoCorasick.with_values([(p, v) for p, v in io["sorted"]]).serialize()
    b = da.DoubleArrayAhoCorasick.with_values([(p, v) for p, v in io["unsorted"]]).serialize()
    assert a == b
    for e in pins["n_blocks"]:
        pats = [bytes(g["prefix"] + [i]) for g in e["gen"] for lo, hi in g["ranges"] for i in range(lo, hi + 1)]
        pma = da.DoubleArrayAhoCorasick.new(pats)
        info = pma.info()
        assert info.num_states == e["num_states"] and info.states_len == e["states_len"], e["name"]
        st = orc.OraclePma.deserialize(pma.serialize()).states()
        for idx, base in e["base_of"].items():
            assert int(st[int(idx), 0]) == base, e["name"]


def test_invalid_option(pins):
    io = pins["invalid_option"]
    with pytest.raises(da.DaachorseError) as ei:
        da.DoubleArrayAhoCorasickBuilder().num_free_blocks(io["num_free_blocks"]).build(io["patterns"])
    assert ei.value.code == 2


def test_tuning_options_are_known_without_a_gpu():
    """daac_set_option: the options the documentation names (include/daachorse_amd.h) are accepted, an unknown one is refused with
    DAAC_ERR_INVALID_ARGUMENT — host logic, no device call."""
    for name, value, back in (("find3", 2, 1), ("left3", 0, 1), ("select_emit", 0, 1), ("find3_window", 1 << 20, 1 << 30), ("workspace_keep", 0, 8 << 30),
                              ("pfx_probe", 1, 16384), ("emit", 0, 1), ("iter_window", 1 << 20, 64 << 20)):
        da.set_option(name, value)
        da.set_option(name, back)
    with pytest.raises(da.DaachorseError) as ei:
        da.set_option("no_such_option", 1)
    assert ei.value.code == 1
    # daac_pma_set_option: the same names per handle (what the -m gpu tests use: nothing process-wide is left behind); the device's
    # allocator is nobody's handle
    p = da.DoubleArrayAhoCorasick.new([b"ab", b"bc"])
    p.set_option("find3", 2).set_option("gram_tail", 1).set_option("find3")
    for name in ("no_such_option", "pool", "pool_keep"):
        with pytest.raises(da.DaachorseError) as ei:
            p.set_option(name, 1)
        assert ei.value.code == 1, name


@pytest.mark.parametrize("nfb", [1, 2, 16])
def test_random_sets_build_identically(nfb):
    rng = np.random.default_rng(100 + nfb)
    for it in range(40):
        alpha = int(rng.integers(2, 40))
        pats = [bytes(rng.integers(0, alpha, size=int(rng.integers(0, 9))).astype(np.uint8)) for _ in range(int(rng.integers(0, 400)))]
        vals = [int(v) for v in rng.integers(0, 2**32, size=len(pats))] if it % 3 == 0 else None
        for kind in KINDS:
            want = orc.OraclePma.build(pats, values=vals, kind=int(kind), num_free_blocks=nfb).serialize()
            assert _blob(pats, kind, nfb, vals) == want, (it, kind)


def test_dictionaries_build_identically():
    for pats in (synth.patterns_cfg2(), synth.patterns_cfg3(30000)):
        for kind in KINDS:
            assert _blob(pats, kind) == orc.OraclePma.build(pats, kind=int(kind)).serialize()
    # wide alphabet, many blocks, window of one block
    rng = np.random.default_rng(3)
    pats = [bytes(rng.integers(0, 256, size=int(rng.integers(1, 6))).astype(np.uint8)) for _ in range(20000)]
    for nfb in (1, 16):
        assert _blob(pats, MatchKind.Standard, nfb) == orc.OraclePma.build(pats, num_free_blocks=nfb).serialize()

"""Host logic of the charwise path (no GPU): the C++ builder and container against the CPU oracle
(byte-identical serialize() blobs, same accept/reject decisions), the device tables derived on the
host, and the boundary's argument checks."""
import json
import os
import subprocess

import numpy as np
import pytest

import daachorse_amd as da
from conftest import GOLDEN, ROOT, iter_vector_runs
from oracle import oracle as orc


@pytest.fixture(scope="module")
def cpins():
    with open(os.path.join(GOLDEN, "charwise_pins.json")) as f:
        return json.load(f)


def _build(pats, kind=0, values=None, nfb=16):
    b = da.CharwiseDoubleArrayAhoCorasickBuilder().match_kind(kind).num_free_blocks(nfb)
    return b.build_with_values(zip(pats, values)) if values is not None else b.build(pats)


def _random_patterns(rng, n, alphabet, max_chars):
    return ["".join(alphabet[i] for i in rng.integers(0, len(alphabet), size=int(rng.integers(1, max_chars + 1)))) for _ in range(n)]


ALPHABETS = {
    "ascii": [chr(c) for c in range(0x61, 0x67)],
    "mixed": list("abcé¢ßд") + ["全", "世", "界", "中", "に", "\U0001F600", "\U00010348"],
    "kana": [chr(c) for c in range(0x3041, 0x3097)] + [chr(c) for c in range(0x4E00, 0x4E80)],
}


def test_builder_blobs_equal_the_oracles_on_every_golden_pattern_set(vectors):
    seen = set()
    for runner, case in iter_vector_runs(vectors):
        key = (runner["kind"], tuple(case["patterns"]))
        if key in seen:
            continue
        seen.add(key)
        o = orc.OracleCharwisePma.build(case["patterns"], kind=runner["kind"])
        p = _build(case["patterns"], orc.KIND[runner["kind"]])
        assert p.serialize() == o.serialize(), (runner["kind"], case["name"])
        assert p.num_states() == o.num_states and p.heap_bytes() == o.heap_bytes()
    assert len(seen) > 100


@pytest.mark.parametrize("alpha", sorted(ALPHABETS))
def test_builder_fuzz_multibyte(alpha):
    rng = np.random.default_rng(hash(alpha) % 1000)
    for trial in range(12):
        pats = _random_patterns(rng, int(rng.integers(1, 400)), ALPHABETS[alpha], 6)
        vals = [int(v) for v in rng.integers(0, 2**32, size=len(pats), dtype=np.uint64)]
        for kind in (0, 1, 2):
            nfb = int(rng.integers(1, 20))
            o = orc.OracleCharwisePma.build(pats, values=vals, kind=kind, num_free_blocks=nfb)
            p = _build(pats, kind, vals, nfb)
            assert p.serialize() == o.serialize(), (alpha, trial, kind)
            assert p.info().charwise == 1 and p.info().alphabet_size == len(set("".join(pats)))


def test_layout_pins_through_the_product_builder(cpins):
    for e in cpins["num_states"]:
        assert _build(e["patterns"]).num_states() == e["num_states"], e["cite"]
    for e in cpins["heap_bytes"]:
        assert _build(e["patterns"]).heap_bytes() == e["heap_bytes"], e["cite"]
    for e in cpins["n_blocks"]:
        pats = ["".join(chr(c) for c in g["prefix"] + [i]) for g in e["gen"] for i in range(g["range"][0], g["range"][1] + 1)]
        p = _build(pats)
        assert p.num_states() == e["num_states"] and p.info().states_len == e["states_len"], e["name"]


def test_serialize_roundtrip_and_trailing_bytes():
    pats = ["全世界", "世界", "に", "abc", ""]
    blob = _build(pats).serialize()
    p2, rest = da.CharwiseDoubleArrayAhoCorasick.deserialize(blob + b"tail")
    assert rest == b"tail" and p2.serialize() == blob
    assert p2.match_kind() == da.MatchKind.Standard and p2.alphabet_size() == 7
    # the two blob formats are not interchangeable: each loader rejects the other's
    with pytest.raises(da.DaachorseError):
        da.DoubleArrayAhoCorasick.deserialize(blob)


def test_corrupt_blobs_are_rejected_like_the_oracle_rejects_them():
    rng = np.random.default_rng(5)
    blob = bytearray(_build(["ab", "bc", "全世界", "世界"]).serialize())
    n_rej = 0
    for trial in range(400):
        b = bytearray(blob)
        if trial % 4 == 0:
            b = b[:int(rng.integers(0, len(b)))]
        else:
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        try:
            orc.OracleCharwisePma.deserialize(bytes(b))
            o_ok = True
        except orc.OracleError:
            o_ok = False
        try:
            da.CharwiseDoubleArrayAhoCorasick.deserialize(bytes(b))
            p_ok = True
        except da.DaachorseError as e:
            assert e.code == 4
            p_ok = False
        # the boundary is allowed to be stricter (failure links that never reach ROOT would hang a GPU), never laxer
        assert not (p_ok and not o_ok), trial
        n_rej += not p_ok
    assert n_rej > 100


def test_cyclic_failure_links_are_rejected():
    blob = bytearray(_build(["ab", "b"]).serialize())
    n = int.from_bytes(blob[0:4], "little")
    states = np.frombuffer(bytes(blob[4:4 + 16 * n]), dtype="<u4").reshape(n, 4).copy()
    real = [i for i in range(2, n) if states[i, 1] != 1 or states[i, 0] != 0]
    states[real[0], 2] = real[0]  # fail -> itself
    blob[4:4 + 16 * n] = states.tobytes()
    with pytest.raises(da.DaachorseError) as ei:
        da.CharwiseDoubleArrayAhoCorasick.deserialize(bytes(blob))
    assert ei.value.code == 4 and "failure links" in str(ei.value)


def test_builder_argument_errors():
    with pytest.raises(da.DaachorseError) as ei:  # not UTF-8: the reference takes &str, the boundary checks
        da.CharwiseDoubleArrayAhoCorasickBuilder().build([b"\xff\xfe"])
    assert ei.value.code == 1
    with pytest.raises(da.DaachorseError) as ei:
        da.CharwiseDoubleArrayAhoCorasickBuilder().build([b"\xe4\xb8"])  # truncated sequence
    assert ei.value.code == 1
    with pytest.raises(da.DaachorseError) as ei:
        da.CharwiseDoubleArrayAhoCorasickBuilder().build_with_values([("a", 2**32)])
    assert ei.value.code == 3


def test_matchkind_mismatch_needs_no_device():
    """charwise.rs:104-107, 163-166, 227-230, 309-312: wrong MatchKind panics before the haystack is looked at"""
    std = _build(["a"], 0)
    for kind in (1, 2):
        lm = _build(["a"], kind)
        for api in ("find_iter", "find_overlapping_iter", "find_overlapping_no_suffix_iter"):
            with pytest.raises(da.DaachorseError) as ei:
                list(getattr(lm, api)(""))
            assert ei.value.code == 5
    with pytest.raises(da.DaachorseError) as ei:
        list(std.leftmost_find_iter(""))
    assert ei.value.code == 5


@pytest.fixture(scope="module")
def char_tables_check(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("native") / "char_tables_check")
    csrc = os.path.join(ROOT, "daachorse_amd", "csrc")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "native", "char_tables_check.cpp"),
                           os.path.join(csrc, "pma.cpp"), os.path.join(csrc, "repack.cpp"), os.path.join(csrc, "charwise.cpp")])
    return exe


def test_device_tables_classic_links(char_tables_check, tmp_path):
    """classic failure links recomputed for leftmost automata == the Standard automaton's own links"""
    rng = np.random.default_rng(9)
    cases = [["ab", "abcd", "bcd", "cd", "d", "全世界", "世界", "界"]]
    for alpha in sorted(ALPHABETS):
        cases.append(_random_patterns(rng, 500, ALPHABETS[alpha], 7))
    for pats in cases:
        a, b = tmp_path / "lm.blob", tmp_path / "st.blob"
        a.write_bytes(orc.OracleCharwisePma.build(pats, kind=1).serialize())
        b.write_bytes(orc.OracleCharwisePma.build(pats, kind=0).serialize())
        out = subprocess.check_output([char_tables_check, str(a), str(b)]).decode()
        assert out.startswith("OK"), out
        assert int(out.split()[1]) == orc.OracleCharwisePma.build(pats, kind=1).num_states

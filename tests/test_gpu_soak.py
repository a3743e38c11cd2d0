"""A short slice of the randomised soak (tools/stress.py) inside the GPU suite: random automata, haystacks, segment /
window / chain / chunk settings and launch geometry against the CPU oracle — every run draws the same, fixed NUMBER of cases from fixed
seeds (count-boxed, not time-boxed), longer runs with other seeds are for spare GPU minutes."""
import importlib.util
import os

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _stress():
    spec = importlib.util.spec_from_file_location("daac_stress", os.path.join(ROOT, "tools", "stress.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_iterators_and_steppers_soak():
    _stress().iter_soak(0.0, 2024, max_cases=150)


def test_count_engines_soak():
    _stress().gram_soak(0.0, 2025, max_cases=40)


def test_round3_engines_soak():
    """gram3 bodies and launch shapes, PFX count and count + checksum, the emitter's two tuple formats: random alphabets of 2 .. 256 byte
    values, duplicate patterns, one-byte patterns, patterns of up to 60 bytes"""
    _stress().engines_soak(0.0, 2026, max_cases=40)


def test_table_sets_never_exceed_the_lds_of_a_workgroup():
    """29 byte classes with K = 3: the first GRAM table set plus its hit rings is more than the 160 KB a workgroup can have (found by the
    round-3 soak, seed 2007: the launch failed with `invalid argument`, and the error it left behind failed the next, healthy call too)"""
    import numpy as np
    import torch
    import daachorse_amd as da
    from daachorse_amd import Engine, ScanMode
    from oracle import oracle as orc
    rng = np.random.default_rng(3)
    syms = np.arange(40, 68, dtype=np.uint8)  # 28 distinct bytes
    pats = [bytes(syms[:10]), bytes(syms[10:20]), bytes(syms[18:28]) + bytes(syms[:7])]
    hay = syms[rng.integers(0, 28, size=600_000)].copy()
    for at in rng.integers(0, len(hay) - 20, size=3000):
        w = np.frombuffer(pats[int(at) % 3], dtype=np.uint8)
        hay[at:at + len(w)] = w
    o = orc.OraclePma.build(pats)
    want = o.overlapping_count(hay, threads=4)
    dev = torch.from_numpy(hay).cuda()
    for budget in (158 * 1024, 200 * 1024, 40 * 1024):
        p, _ = da.DoubleArrayAhoCorasick.deserialize(o.serialize())
        p.set_option("pfx", 2).set_option("gram_lds_budget", budget)   # (this handle's own settings, read at upload)
        info = p.upload().info()
        assert info.num_classes == 29
        assert not info.gram_available or info.gram_lds_bytes <= 160 * 1024, (budget, info.gram_lds_bytes)
        for eng in (Engine.Auto, Engine.Gram, Engine.Pfx, Engine.DArray):
            try:
                assert p.scan_count(ScanMode.FindOverlapping, dev, engine=eng) == want, (budget, eng)
                assert p.count(ScanMode.FindOverlapping, dev, engine=eng) == want[0], (budget, eng)
            except da.DaachorseError as e:
                assert e.code == 6, (budget, eng, str(e))  # "this engine does not serve the request" is the only refusal allowed

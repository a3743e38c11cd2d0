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

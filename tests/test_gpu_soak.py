"""A short slice of the randomised soak (tools/stress.py) inside the GPU suite: random automata, haystacks, segment /
window / chain / chunk settings and launch geometry against the CPU oracle — every run draws the same cases (fixed
seeds), longer runs with other seeds are for spare GPU minutes."""
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
    _stress().iter_soak(8.0, 2024)


def test_count_engines_soak():
    _stress().gram_soak(8.0, 2025)

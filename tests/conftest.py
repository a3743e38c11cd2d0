import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def vectors():
    with open(os.path.join(GOLDEN, "aho_corasick_vectors.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def pins():
    with open(os.path.join(GOLDEN, "bytewise_pins.json")) as f:
        return json.load(f)


def iter_vector_runs(vectors):
    """Yields (runner, case) for every vector-run of tests/aho_corasick_crate_test.rs:537-589."""
    for runner in vectors["runners"]:
        for table in vectors["collections"][runner["collection"]]:
            for case in vectors["tables"][table]:
                yield runner, case

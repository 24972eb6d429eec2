import gzip
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


def load_golden(name):
    with gzip.open(os.path.join(GOLDEN, name), "rt") as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def golden_integration():
    return load_golden("integration.json.gz")


@pytest.fixture(scope="session")
def golden_synthetic():
    return load_golden("synthetic.json.gz")


@pytest.fixture(scope="session")
def golden_kmeans():
    return load_golden("kmeans.json.gz")

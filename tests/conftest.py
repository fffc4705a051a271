import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def kat():
    with open(os.path.join(GOLDEN, "reference_kat.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def seeded():
    return dict(np.load(os.path.join(GOLDEN, "oracle_seeded.npz")))


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle

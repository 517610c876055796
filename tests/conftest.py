import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json
    return json.load(open(os.path.join(HERE, "golden", "golden.json")))


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(HERE, "golden")


@pytest.fixture(scope="session")
def golden_fox():
    """real-content fixtures made by tests/golden/make_golden_foxlogo.py from the reference's own foxlogo assets"""
    import json
    return json.load(open(os.path.join(HERE, "golden", "golden_foxlogo.json")))


@pytest.fixture(scope="session")
def foxlogo():
    import numpy as np
    z = np.load(os.path.join(HERE, "golden", "foxlogo.npz"))
    return {"frames": z["frames"], "p0": z["p0"], "p1": z["p1"]}

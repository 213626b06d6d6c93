import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def O():
    from oracle import oracle

    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def golden():
    import json

    with open(os.path.join(ROOT, "tests", "golden", "reconstruction_golden.json")) as f:
        return json.load(f)

import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def fixtures():
    with open(os.path.join(ROOT, "tests", "golden", "reference_fixtures.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    from oracle import load_oracle
    return load_oracle()


@pytest.fixture(scope="session")
def lib():
    """The product C-ABI library, loaded through ctypes (fails loudly if it was not built)."""
    from fnft_amd import capi
    return capi.load()

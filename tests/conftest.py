import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle as O  # oracle/oracle.py (test infrastructure)
    O.lib()
    return O


@pytest.fixture(scope="session")
def cc():
    import cp_cals_amd
    return cp_cals_amd


@pytest.fixture(scope="session")
def inputs():
    from cp_cals_amd import inputs as I
    return I

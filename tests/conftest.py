"""pytest configuration: `-m gpu` tests need a real MI355X, everything else runs on CPU."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu`)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): built on demand with gcc."""
    from oracle import cpu_oracle
    cpu_oracle.build()
    return cpu_oracle


@pytest.fixture(scope="session")
def np_oracle():
    from oracle import np_oracle
    return np_oracle


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")

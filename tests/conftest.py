import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")
    config.addinivalue_line("markers", "slow: soak tests; the default form is short (seconds), "
                                       "TS_SOAK_PROOFS=1500 gives the long one")


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle_py

    oracle_py.build()
    return oracle_py

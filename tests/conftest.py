import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure; builds oracle/liboracle.so with gcc on first use)."""
    from oracle import oracle

    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def ofdm():
    """The product package; loading the HIP library must work (fails loudly otherwise)."""
    import ofdm_amd

    ofdm_amd.load()
    return ofdm_amd

import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    """The CPU oracle (test infrastructure), built on demand from oracle/."""
    from tests import _oracle
    return _oracle.load()


@pytest.fixture(scope="session")
def hip_lib():
    from mpc_benchmark_amd import _capi
    return _capi.load_hip_library()

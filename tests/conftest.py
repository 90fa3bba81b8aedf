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


# Under `-x` a failure stops the run: the files that compare the HIP path with the oracle come first, the long robustness walks (whole schedules of the
# benchmarked ensemble, no oracle involved) last, so that a late timeout or a lost instance there cannot hide the parity verdict.
_PARITY_FIRST = ("test_abi_library", "test_gpu_fulldynamic", "test_gpu_kinodynamic", "test_gpu_centroidal", "test_gpu_fixed_dims", "test_gpu_legs", "test_gpu_edge_cases",
                 "test_gpu_free_running", "test_gpu_refine", "test_gpu_corrector", "test_gpu_walk", "test_gpu_walking_loop", "test_gpu_qp", "test_pipeline", "test_walk_generator",
                 "test_checkpoint", "test_dropin_schedule", "test_dropin_fixtures", "test_gpu_shards", "test_instance_params")
_WALKS_LAST = ("test_whole_schedule", "whole_schedule", "test_config4", "test_pipeline_walks_the_whole_schedule")


def pytest_collection_modifyitems(config, items):
    def key(item):
        mod = item.module.__name__.rsplit(".", 1)[-1]
        late = any(w in item.name for w in _WALKS_LAST)
        rank = _PARITY_FIRST.index(mod) if mod in _PARITY_FIRST else len(_PARITY_FIRST)
        return (1 if late else 0, rank)
    items.sort(key=key)   # (stable: the order inside a file is kept)

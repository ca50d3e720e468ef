import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """the multi-rank rehearsals of bench.py run LAST: two of them kill a rank on purpose (a hung exchange form, a failing
    extra), and whatever such a process leaves behind on the box (RCCL's shared-memory segments, a busy device) must not meet
    the RCCL tests of the other files"""
    last = [it for it in items if "test_gpu_bench_ranks" in it.nodeid]
    if last:
        items[:] = [it for it in items if "test_gpu_bench_ranks" not in it.nodeid] + last


@pytest.fixture(scope="session")
def oracle():
    from oracle import pfpn_oracle
    return pfpn_oracle


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")

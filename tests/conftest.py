import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


_HEAVY_ORACLE_MODULES = ("test_fullsize_gpu", "test_loss_block_gpu")
_ALL_THREADS = None


def pytest_runtest_setup(item):
    """CPU threads of the oracle: the GPU box's host exposes every core of the node (256); for the small shapes most tests
    feed the oracle a pool that size is pure overhead (a toy training loop ran 20x slower with it than with two threads).
    The full-size parity tests keep the whole pool, everything else gets eight."""
    global _ALL_THREADS
    import torch
    if _ALL_THREADS is None:
        _ALL_THREADS = torch.get_num_threads()
    heavy = any(m in item.module.__name__ for m in _HEAVY_ORACLE_MODULES)
    torch.set_num_threads(_ALL_THREADS if heavy else min(8, _ALL_THREADS))

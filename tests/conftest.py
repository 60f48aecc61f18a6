import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "neuralnet-tracker-traincode_amd")
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _fixed_cpu_threads():
    """The CPU oracle's fp32 rounding depends on how torch splits reductions over threads; with tiny batches one ReLU decision on the other
    side moves whole gradients by 1e-3 (tests/test_backbone_gpu.py).  A fixed thread count makes the oracle's values the same on every
    box with at least 8 cores; the large-batch tests raise it themselves for speed."""
    import torch

    torch.set_num_threads(min(8, os.cpu_count() or 1))
    yield

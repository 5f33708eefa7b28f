import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture
def reference_ops():
    """the module tree / trainer on the fp32 PyTorch statement of the ops (oracle/nn_ref.py) instead of the HIP kernels:
    CPU tests mark themselves with `pytestmark = pytest.mark.usefixtures("reference_ops")`"""
    from oracle import nn_ref
    with nn_ref.reference_ops():
        yield


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)

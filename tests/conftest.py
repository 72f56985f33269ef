import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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


@pytest.fixture(scope="session")
def vae_golden():
    return np.load(os.path.join(HERE, "golden", "vae_golden.npz"))


@pytest.fixture(scope="session")
def vae_golden_big():
    """The benchmarked 8192-frame batch captured from the reference (make_golden.py --big)."""
    return np.load(os.path.join(HERE, "golden", "vae_golden_b8192.npz"))


@pytest.fixture(scope="session")
def stft_golden():
    return np.load(os.path.join(HERE, "golden", "stft_ref_fixture.npz"))

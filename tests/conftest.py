import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    from oracle import pyoracle

    pyoracle.build()


def _have_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """plain `pytest` on a box without a HIP device skips the gpu-marked tests instead of failing them with
    "no usable HIP device"; `-m gpu` (the GPU box) runs them regardless and lets a missing device fail loudly"""
    if "gpu" in (config.getoption("-m") or "") or _have_gpu():
        return
    skip = pytest.mark.skip(reason="needs a MI355X (no HIP device here); run with -m gpu on the GPU box")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)

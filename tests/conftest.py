"""pytest configuration: registers the ``gpu`` marker, puts the repo root on
sys.path and exposes the package (whose directory name ``mg-gcn_amd`` is not a
Python identifier) and the oracle as fixtures."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # A gpu test collected without a GPU is skipped (the driver selects by marker;
    # this only protects a plain `pytest tests/` in the CPU container).
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    orc.lib()
    return orc


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as ge
    return ge.load_package()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly rather than skip: the
    # product has no CPU path. Plain runs simply deselect via `-m "not gpu"`.
    pass


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle

    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def hip():
    """The product binding. The shared object is git-ignored, so a fresh checkout
    has to compile it first (hipcc cross-compiles gfx950 without a GPU)."""
    import subprocess

    from drone_amd import binding

    if not os.path.exists(binding.LIB_PATH):
        subprocess.run(["make", "-C", os.path.join(ROOT, "drone_amd", "csrc")], check=True, capture_output=True)
    binding.load()
    return binding

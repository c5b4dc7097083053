import os
import sys

import pytest

try:    # PyTorch-ROCm bundles its own libamdhip64 (same soname as /opt/rocm's): whichever is loaded
    import torch  # noqa: F401  first serves the whole process, and torch cannot initialise on top of
except Exception:  # the system runtime — so torch goes first ("Sharing a process with PyTorch")
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than a few seconds on CPU")


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def lib():
    from clfacedetection_amd import load_library
    return load_library()


@pytest.fixture(scope="session")
def env():
    """One HIP environment for the whole GPU session (fails loudly without a GPU)."""
    from clfacedetection_amd import Environment
    e = Environment(0)
    yield e
    e.close()


_CASC = {}


@pytest.fixture(scope="session")
def cascades():
    """name -> (product Cascade, oracle CascadeArrays) for the shipped .vjc files."""
    from clfacedetection_amd import Cascade
    from clfacedetection_amd.api import DATA_DIR
    from oracle.oracle import load_vjc

    def get(name):
        if name not in _CASC:
            _CASC[name] = (Cascade.load(name), load_vjc(os.path.join(DATA_DIR, f"haarcascade_{name}.vjc")))
        return _CASC[name]
    return get

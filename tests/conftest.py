import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): oracle/oracle.py over oracle/_build/*.so."""
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def smcmc():
    """The product package (HIP library behind the C ABI)."""
    from smcmc_amd_loader import load_package
    return load_package()


def _gpu_present():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu(smcmc):
    if not _gpu_present():
        pytest.fail("a -m gpu test ran without a visible GPU: the HIP path has no CPU fallback")
    smcmc.load()
    return smcmc

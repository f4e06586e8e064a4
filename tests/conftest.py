import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_NUM_THREADS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def gsmcal_mod():
    import gsmcal
    return gsmcal


@pytest.fixture(scope="session")
def ctx(gsmcal_mod):
    """A GPU context; fails loudly (never skips to a CPU path) when the HIP library or GPU is missing."""
    return gsmcal_mod.default_context(0)


@pytest.fixture(scope="session")
def g_mod(gsmcal_mod, ctx):
    return gsmcal_mod

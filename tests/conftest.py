import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _ensure_built():
    from pathtrace_amd import _lib
    from oracle import orc
    if not (os.path.exists(_lib.LIB_PATH) and os.path.exists(orc.LIB_PATH)):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session", autouse=True)
def built():
    _ensure_built()


@pytest.fixture(scope="session")
def pt():
    import pathtrace_amd
    return pathtrace_amd


@pytest.fixture(scope="session")
def orc():
    from oracle import orc as o
    return o


@pytest.fixture(scope="session")
def gpu_ctx(pt):
    """One context for the whole GPU session (fails loudly if the HIP library or device is missing)."""
    ctx = pt.Context(0)
    yield ctx
    ctx.close()


def load_luminance_csv(path):
    """Reader of the reference's luminance.csv format (src/world.rs:344-369) -> float64[h,w,3]."""
    import numpy as np
    data = np.loadtxt(path, delimiter=",", skiprows=1)
    w = int(data[:, 0].max()) + 1
    h = int(data[:, 1].max()) + 1
    img = np.zeros((h, w, 3))
    img[data[:, 1].astype(int), data[:, 0].astype(int)] = data[:, 2:5]
    return img

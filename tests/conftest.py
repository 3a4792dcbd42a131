import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def L():
    return importlib.import_module("old-vpic_amd.layout")


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "kernels.npz"))


@pytest.fixture(scope="session")
def orc():
    from oracle import pyorc
    pyorc.lib()
    return pyorc


def bits_equal(a, b):
    """Bit-for-bit equality of two structured/plain arrays (padding bytes excluded)."""
    if a.dtype.names:
        return all(bits_equal(a[n], b[n]) for n in a.dtype.names if not n.startswith("_"))
    if a.dtype.kind == "f":
        return np.array_equal(a.view(f"u{a.dtype.itemsize}"), b.view(f"u{b.dtype.itemsize}"))
    return np.array_equal(a, b)

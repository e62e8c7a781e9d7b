import os
import sys

import pytest

# Some CPU tests import the reference's Python modules from /root/reference (authoring container): the interpreter must not leave
# bytecode caches in that read-only tree.
sys.dont_write_bytecode = True

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun / the driver's GPU tier)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def aveg(golden_dir):
    import numpy as np
    z = np.load(os.path.join(golden_dir, "aveg_mesh.npz"))
    return z["rest"], z["tets"]

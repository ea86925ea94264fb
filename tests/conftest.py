import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    path = os.path.join(ROOT, "tests", "golden", "reference_outputs.npz")
    return dict(np.load(path))


@pytest.fixture(scope="session")
def headdim_golden():
    import numpy as np
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "proc_headdim_golden.npz")))


@pytest.fixture(scope="session")
def paras_golden():
    import numpy as np
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "paras_golden.npz")))

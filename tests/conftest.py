import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_update():
    import numpy as np
    from psim_testlib import GOLDEN_DIR
    return np.load(os.path.join(GOLDEN_DIR, "update_cases.npz"))


@pytest.fixture(scope="session")
def golden_generator():
    import numpy as np
    from psim_testlib import GOLDEN_DIR
    return np.load(os.path.join(GOLDEN_DIR, "generator_vectors.npz"))

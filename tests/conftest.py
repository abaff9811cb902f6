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
def oracle():
    """The CPU oracle (test infrastructure). Built on demand with gcc."""
    from oracle import oracle as orc
    orc.build()
    return orc


def random_floats(rng, *shape):
    """src/test_helper.rs:3-6: uniform in [-0.5, 0.5)."""
    return (rng.random(shape, dtype=np.float32) - np.float32(0.5)).astype(np.float32)

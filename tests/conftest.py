import os
import sys

import numpy as np
import pytest

# The oracle (oracle/granne_oracle.c) opens one OpenMP region per build batch. The tests build thousands of tiny
# batches: on a 256-thread host an all-cores team that spin-waits between regions turns seconds into minutes
# (measured on the GPU box: 16 s vs 285 s for the same tests). A small, sleeping team is plenty here; bench.py's
# CPU baseline runs in its own process and is not affected.
os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(8, os.cpu_count() or 1))))
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")

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


def assert_counters(st, octr, exact):
    """The walk's counters [n_dist, n_expand, n_adj] against the oracle's. Expansions and adjacency entries are the
    reference's in every mode. n_dist is the reference's count of distinct evaluated nodes when the walker keeps an
    exact visited set (GRANNE_HIP_OPT_VISITED16 1..3, the general and the exact walker); the default register walkers
    keep none and evaluate a revisited neighbor again (granne_amd/csrc/wave_prims.h, VisitedNone): their n_dist lies
    between the reference's and one evaluation per adjacency entry read plus one entry point per layer."""
    st, octr = np.asarray(st).astype(np.int64), np.asarray(octr).astype(np.int64)
    assert (st[:, 1:] == octr[:, 1:]).all()
    if exact:
        assert (st[:, 0] == octr[:, 0]).all()
    else:
        assert (st[:, 0] >= octr[:, 0]).all()
        assert (st[:, 0] <= octr[:, 2] + 64).all()

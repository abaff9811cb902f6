"""The committed golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py) must
be reproduced by the oracle on every box: same layers from the deterministic build, same ids,
bit-identical distances, same counters."""
import glob
import os

import numpy as np
import pytest

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "*.npz")))
GOLDEN = [p for p in GOLDEN if not os.path.basename(p).startswith(("reorder_", "build_"))]


def load_case(path):
    z = np.load(path)
    layers = [z["layer%d" % l] for l in range(int(z["n_layers"]))]
    searches = sorted((int(k.split("_")[1]), int(k.split("_")[2])) for k in z.files if k.startswith("ids_"))
    return z, layers, searches


def test_fixtures_exist():
    assert len(GOLDEN) == 5


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_oracle_reproduces_golden(oracle, path):
    z, layers, searches = load_case(path)
    ix = oracle.Index(z["elements"], layers)
    for ms, k in searches:
        ids, ds, cnt, ctr = ix.search_batch(z["queries"], ms, k, n_threads=2)
        assert (ids == z["ids_%d_%d" % (ms, k)]).all()
        assert ds.tobytes() == z["dists_%d_%d" % (ms, k)].tobytes()
        assert (cnt == z["counts_%d_%d" % (ms, k)]).all()
        assert (ctr == z["stats_%d_%d" % (ms, k)]).all()


def test_golden_build_is_reproduced(oracle):
    z, layers, _ = load_case([p for p in GOLDEN if p.endswith("f32_d28.npz")][0])
    ix = oracle.build_index(z["elements"], num_neighbors=20, max_search=20, n_threads=1)
    assert len(ix.layers) == len(layers)
    assert all((a == b).all() for a, b in zip(ix.layers, layers))


def test_oracle_reproduces_golden_reorder(oracle):
    """Granne::reorder's permutation for every fixture index (tests/golden/reorder_orders.npz)."""
    orders = np.load(os.path.join(os.path.dirname(GOLDEN[0]), "reorder_orders.npz"))
    assert sorted(orders.files) == sorted(os.path.basename(p)[:-4] for p in GOLDEN)
    for path in GOLDEN:
        z, layers, _ = load_case(path)
        got = oracle.Index(z["elements"], layers).compute_order(n_threads=2)
        assert got.tolist() == orders[os.path.basename(path)[:-4]].tolist()


def test_oracle_reproduces_golden_batched_build(oracle):
    """The batched insertion schedule (what the GPU builder runs) on the f32_d28 rows: tests/golden/build_batched_f32_d28.npz."""
    here = os.path.dirname(GOLDEN[0])
    z = np.load(os.path.join(here, "build_batched_f32_d28.npz"))
    el = np.load(os.path.join(here, "f32_d28.npz"))["elements"]
    for threads in (1, 3):  # deterministic for any thread count
        ix = oracle.build_index(el, num_neighbors=20, max_search=20, batch_max=64, batch_div=8, n_threads=threads)
        assert len(ix.layers) == int(z["n_layers"])
        assert all((a == z["layer%d" % l]).all() for l, a in enumerate(ix.layers))

"""Granne::reorder / reorder_by_keys on the GPU (src/index/reorder.rs) against the oracle's restatement:
the permutation, the rewritten layers and the permuted elements, bit for bit -- and the reference's own
test (reorder.rs:299-322): search results are equal modulo the permutation."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import pyref  # noqa: E402
from tests.conftest import random_floats  # noqa: E402


@pytest.fixture(scope="module")
def ga():
    import granne_amd
    return granne_amd


def prep(oracle, raw, int8):
    return oracle.quantize(raw) if int8 else oracle.normalize_f32(raw)


def gpu_index(ga, oix):
    et = "angular" if oix.elements.dtype == np.float32 else "angular_int"
    return ga.Granne(et, oix.elements, oix.layers)


def assert_same_index(gix, oix):
    assert len(gix) == len(oix) and gix.num_layers() == len(oix.layers)
    for l, rows in enumerate(oix.layers):
        assert gix.layer_len(l) == rows.shape[0]
        for i in range(rows.shape[0]):
            assert gix.get_neighbors(i, l) == pyref.get_neighbors(rows, i), (l, i)
    for i in range(0, len(oix), max(1, len(oix) // 97)):
        assert gix.get_element(i).tobytes() == oix.elements[i].tobytes()


CASES = [
    # n, dim, int8, num_neighbors, max_search, layer_multiplier
    (5000, 5, False, 30, 5, 5.0),     # the reference's reorder_index test
    (3000, 100, False, 30, 40, 6.0),
    (2000, 100, True, 20, 20, 5.0),
    (1500, 28, False, 12, 20, 3.0),   # 7 layers
    (2200, 8, False, 8, 10, 2.0),     # > 8 layers: the trail is capped at NUM_LAYERS
    (900, 200, False, 16, 20, 15.0),
]


@pytest.mark.parametrize("n,dim,int8,nn,ms,mult", CASES)
def test_reorder_equals_oracle(ga, oracle, n, dim, int8, nn, ms, mult):
    rng = np.random.default_rng(n + dim)
    el = prep(oracle, random_floats(rng, n, dim), int8)
    oix = oracle.build_index(el, num_neighbors=nn, max_search=ms, layer_multiplier=mult)
    assert len(oix.layers) >= 2
    gix = gpu_index(ga, oix)
    want = oix.compute_order()
    got = gix.reorder()
    assert got.dtype == np.uint64 and got.tolist() == want.tolist()
    ore = oix.reordered(want)
    assert_same_index(gix, ore)
    # reorder.rs:311-320
    q = np.concatenate([el[[0, 10, 123, 99, 499]], prep(oracle, random_floats(rng, 32, dim), int8)])
    before = oix.search_batch(q, 10, 10)
    ids, ds, cnt = gix.search_batch(q, 10, 10)
    for i in range(q.shape[0]):
        c = int(cnt[i])
        assert c == int(before[2][i])
        assert [int(got[j]) for j in ids[i, :c]] == before[0][i, :c].tolist()
        assert ds[i, :c].tobytes() == before[1][i, :c].tobytes()
    # and like the oracle's reordered index, including tie-breaks on the new ids
    oi, od, oc, _ = ore.search_batch(q, 25, 10)
    ids, ds, cnt = gix.search_batch(q, 25, 10)
    assert (ids == oi).all() and ds.tobytes() == od.tobytes() and (cnt == oc).all()


def test_reorder_twice_and_save(ga, oracle, tmp_path):
    """A reordered index is an ordinary index: it can be reordered again and written in granne's format."""
    from oracle import fileformat
    rng = np.random.default_rng(5)
    el = prep(oracle, random_floats(rng, 2500, 32), False)
    oix = oracle.build_index(el, num_neighbors=16, max_search=20, layer_multiplier=5.0)
    gix = gpu_index(ga, oix)
    o1 = gix.reorder()
    ore = oix.reordered(oix.compute_order())
    o2 = gix.reorder()
    assert o2.tolist() == ore.compute_order().tolist()
    ore2 = ore.reordered(o2)
    assert_same_index(gix, ore2)
    assert sorted(o1.tolist()) == list(range(2500))
    gix.save_index(str(tmp_path / "i.granne"))
    gix.save_elements(str(tmp_path / "e.bin"))
    _meta, layers = fileformat.read_index((tmp_path / "i.granne").read_bytes())
    for l, rows in enumerate(ore2.layers):
        for i in range(0, rows.shape[0], 7):
            assert list(layers[l][i]) == pyref.get_neighbors(rows, i)
    back = ga.Granne.from_files(str(tmp_path / "i.granne"), "angular", str(tmp_path / "e.bin"))
    q = prep(oracle, random_floats(rng, 8, 32), False)
    a, b = back.search_batch(q, 30, 10), gix.search_batch(q, 30, 10)
    assert (a[0] == b[0]).all() and a[1].tobytes() == b[1].tobytes()


def test_reorder_by_keys(ga, oracle):
    rng = np.random.default_rng(6)
    el = prep(oracle, random_floats(rng, 4000, 16), False)
    oix = oracle.build_index(el, num_neighbors=12, max_search=20, layer_multiplier=7.0)
    keys = rng.integers(0, 60, 4000).astype(np.uint64)
    keys[::5] += np.uint64(1) << np.uint64(40)  # the upper key half matters too
    gix = gpu_index(ga, oix)
    got = gix.reorder_by_keys(keys)
    want = oix.order_by_keys(keys)
    assert got.tolist() == want.tolist()
    assert_same_index(gix, oix.reordered(want))
    with pytest.raises(ValueError):
        gix.reorder_by_keys(keys[:-1])


def test_reorder_rejects_what_the_reference_panics_on(ga, oracle):
    rng = np.random.default_rng(7)
    el = prep(oracle, random_floats(rng, 12, 8), False)
    one = oracle.build_index(el)
    assert len(one.layers) == 1
    with pytest.raises(RuntimeError, match="two layers"):
        gpu_index(ga, one).reorder()
    # len() != number of elements (build_partial): the reference's permute asserts
    big = prep(oracle, random_floats(rng, 600, 8), False)
    part = oracle.build_index(big, num_elements=400, layer_multiplier=5.0, max_search=10)
    with pytest.raises(RuntimeError, match="number of elements"):
        ga.Granne("angular", part.elements, part.layers).reorder()
    # the trail walks survive the slow path too
    full = oracle.build_index(big, layer_multiplier=5.0, max_search=10)
    g = gpu_index(ga, full)
    from granne_amd import _lib
    g.set_option(_lib.OPT_FORCE_SLOW, 1)
    assert g.reorder().tolist() == full.compute_order().tolist()


def test_gpu_reorder_reproduces_golden(ga):
    """The committed permutations (tests/golden/reorder_orders.npz, made by tests/golden/make_golden.py)."""
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    orders = np.load(os.path.join(here, "reorder_orders.npz"))
    for name in orders.files:
        z = np.load(os.path.join(here, name + ".npz"))
        layers = [z["layer%d" % l] for l in range(int(z["n_layers"]))]
        et = "angular" if z["elements"].dtype == np.float32 else "angular_int"
        gix = ga.Granne(et, z["elements"], layers)
        assert gix.reorder().tolist() == orders[name].tolist(), name

"""Granne::reorder (src/index/reorder.rs) -- the oracle's restatement pinned by the reference's own
test (reorder.rs:299-322), its reverse-mapping test (:324-334), and the independent Python restatement."""
import numpy as np
import pytest

from oracle import pyref


def _index(oracle, n, dim, seed, int8=False, **kw):
    rng = np.random.default_rng(seed)
    raw = (rng.random((n, dim), dtype=np.float32) - 0.5).astype(np.float32)
    el = np.stack([oracle.quantize(r) for r in raw]) if int8 else np.stack([oracle.normalize_f32(r) for r in raw])
    return oracle.build_index(el, **kw)


def test_reorder_index_reference_test(oracle):
    """reorder.rs:299-322: 5000 x 5-d, max_search(5), layer_multiplier(5.0); results equal modulo the permutation."""
    ix = _index(oracle, 5000, 5, 1, max_search=5, layer_multiplier=5.0)
    perm = ix.compute_order()
    assert sorted(perm.tolist()) == list(range(5000))
    re = ix.reordered(perm)
    for idx in [0, 10, 123, 99, 499]:
        element = ix.elements[idx]
        exp = ix.search(element, 10, 10)
        res = re.search(element, 10, 10)
        assert len(exp) == len(res) == 10
        for i in range(10):
            assert exp[i][0] == perm[res[i][0]]
            assert exp[i][1] == res[i][1]


def test_order_is_layer_preserving(oracle):
    ix = _index(oracle, 3000, 8, 2, max_search=20, layer_multiplier=6.0)
    perm = ix.compute_order()
    lens = [0] + [l.shape[0] for l in ix.layers]
    assert len(lens) >= 4
    assert perm[: lens[1]].tolist() == list(range(lens[1]))  # layer 0 keeps its order (reorder.rs:136)
    # layer 1's keys all map through the zero-initialised order_inv -> sorted by idx (reorder.rs:137,159)
    assert perm[lens[1]: lens[2]].tolist() == list(range(lens[1], lens[2]))
    for a, b in zip(lens[:-1], lens[1:]):
        assert sorted(perm[a:b].tolist()) == list(range(a, b))
    assert perm[lens[2]:].tolist() != list(range(lens[2], lens[-1]))  # deeper layers do move


@pytest.mark.parametrize("int8", [False, True])
def test_compute_order_matches_python_restatement(oracle, int8):
    ix = _index(oracle, 700, 12, 3 + int8, int8=int8, max_search=10, layer_multiplier=4.0, num_neighbors=8)
    assert len(ix.layers) >= 4
    want = pyref.compute_order(ix.layers, ix.elements)
    got = ix.compute_order()
    assert got.tolist() == want
    re = ix.reordered(got)
    want_layers = pyref.reorder_layers(ix.layers, want)
    for l, rows in zip(re.layers, want_layers):
        for i, row in enumerate(rows):
            assert pyref.get_neighbors(l, i) == row
            assert (l[i, len(row):] == 0xFFFFFFFF).all()
    assert (re.elements == ix.elements[np.asarray(want)]).all()


def test_reverse_mapping_reference_test(oracle):
    """reorder.rs:324-334 through gro_reorder_layer: i == rev[mapping[i]]."""
    n = 105
    mapping = np.arange(n, dtype=np.uint64)[::-1].copy()
    rows = np.arange(n, dtype=np.uint32).reshape(n, 1)  # node i -> neighbor i
    ix = oracle.Index(np.zeros((n, 2), np.float32), [rows])
    out = ix.reordered(mapping).layers[0]
    # row i holds rev[mapping[i]] == i
    assert out[:, 0].tolist() == list(range(n))


def test_reorder_needs_two_layers(oracle):
    ix = _index(oracle, 10, 4, 5)
    assert len(ix.layers) == 1
    with pytest.raises(RuntimeError):
        ix.compute_order()


def test_order_by_keys(oracle):
    """reorder_by_keys (reorder.rs:88-110): a layer-preserving sort by (key, idx)."""
    ix = _index(oracle, 2000, 6, 6, max_search=10, layer_multiplier=7.0)
    rng = np.random.default_rng(0)
    keys = rng.integers(0, 50, 2000).astype(np.uint64)  # many ties
    got = ix.order_by_keys(keys)
    lens = [0] + [l.shape[0] for l in ix.layers]
    want = []
    for a, b in zip(lens[:-1], lens[1:]):
        want += sorted(range(a, b), key=lambda l: (int(keys[l]), l))
    assert got.tolist() == want
    re = ix.reordered(got)
    element = ix.elements[77]
    for (i, d), (j, e) in zip(ix.search(element, 20, 10), re.search(element, 20, 10)):
        assert i == got[j] and d == e

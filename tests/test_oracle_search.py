"""Oracle pinning, part 2: src/index/mod.rs (search + build) and the adjacency codec.

Known-answer tests of the reference are restated verbatim; its property tests (unseeded RNG
upstream) run here on seeded draws of the same distribution; and the C oracle's search is
diffed id-for-id / bit-for-bit against the independent Python restatement oracle/pyref.py.
"""
import numpy as np
import pytest

from oracle import pyref
from tests.conftest import random_floats


def random_vectors(oracle, rng, n, dim, int8=False):
    """test_helper::random_vectors (src/test_helper.rs:12-19): each row through Vector::from."""
    raw = random_floats(rng, n, dim)
    return oracle.quantize(raw) if int8 else oracle.normalize_f32(raw)


# ---- known-answer tests ---------------------------------------------------------------------
KAT_LAYERS = [  # src/index/tests.rs:314-334
    (1000, 10.0, [10, 100, 1000]),
    (32, 2.0, [1, 2, 4, 8, 16, 32]),
    (10_000, 10.0, [1, 10, 100, 1000, 10_000, 10_000]),
    (20, 1.9, [2, 3, 6, 11, 20, 20]),
    (1_000_000_000, 20.0, [16, 313, 6250, 125_000, 2_500_000, 50_000_000, 1_000_000_000, 1_000_000_000]),
    (50, 100.0, [50]),
    (133689866, 15.0, [12, 177, 2641, 39612, 594178, 8912658, 133689866]),
]


@pytest.mark.parametrize("total,mult,expected", KAT_LAYERS)
def test_num_elements_in_layer_kat(oracle, total, mult, expected):
    assert [oracle.num_elements_in_layer(total, mult, l) for l in range(len(expected))] == expected
    assert [pyref.compute_num_elements_in_layer(total, mult, l) for l in range(len(expected))] == expected


def test_layer_sizes_of_the_benchmark_configs(oracle):
    """SURVEY.md 8 / BASELINE.md 3: the layer pyramids the configs imply."""
    assert [oracle.num_elements_in_layer(400_000, 15.0, l) for l in range(5)] == [8, 119, 1778, 26667, 400000]
    assert [oracle.num_elements_in_layer(10_000_000, 15.0, l) for l in range(6)] == \
        [14, 198, 2963, 44445, 666667, 10000000]


def test_delta_encode_kat(oracle):
    """src/slice_vector/set_vector.rs:231-237 and :239-248."""
    assert oracle.delta_encode([1, 2, 2, 4]).tolist() == [1, 1, 0, 2]
    data = [123, 345, 555, 555, 6999, 7000]
    assert oracle.delta_decode(oracle.delta_encode(data)).tolist() == data


@pytest.mark.parametrize("ids", [
    list(range(10)),            # set_vector.rs:250-260 push_and_get
    [],                         # :262-270 push_and_get_empty
    [37717, 660380],            # :272-283 push_and_get_4_bytes_per_number
    [5],                        # :285-293
    [5, 5],                     # :295-303 duplicates
    [0, 1, 2**32 - 2],
    list(range(0, 30 * 400_000, 400_000)),
])
def test_set_codec_roundtrip(oracle, ids):
    enc = oracle.set_encode(sorted(ids))
    assert enc[0] == len(ids)
    assert oracle.set_decode(enc).tolist() == sorted(ids)


def test_set_codec_sizes(oracle):
    """'4 bytes per number': 1 control byte + 2+3+1+1 data bytes == 2*4, so the raw form is
    kept (set_vector.rs:137-143, comment at :272-274)."""
    assert len(oracle.set_encode([37717, 660380])) == 1 + 8
    assert oracle.set_encode([]) == b"\x00"
    # ten small deltas: 1 count byte + 3 control bytes + 10 data bytes
    assert len(oracle.set_encode(list(range(10)))) == 1 + 3 + 10


# ---- property tests of the reference ---------------------------------------------------------
def verify_search(index, precision, max_search):
    """src/index/tests.rs:50-62."""
    found = 0
    for i in range(len(index)):
        if index.search(index.elements[i], max_search, 1)[0][0] == i:
            found += 1
    p1 = found / len(index)
    assert precision < p1, p1


def test_build_and_search_float(oracle):
    """src/index/tests.rs:41-48,114-121: 1500 x 28-d, num_neighbors 20, max_search 20."""
    rng = np.random.default_rng(10)
    ix = oracle.build_index(random_vectors(oracle, rng, 1500, 28), num_neighbors=20, max_search=20)
    assert len(ix) == 1500
    verify_search(ix, 0.95, 10)


def test_build_and_search_int8(oracle):
    """src/index/tests.rs:123-132: 500 x 32-d int8."""
    rng = np.random.default_rng(11)
    ix = oracle.build_index(random_vectors(oracle, rng, 500, 32, int8=True), num_neighbors=20, max_search=20)
    verify_search(ix, 0.95, 10)


def test_with_borrowed_elements_config(oracle):
    """src/index/tests.rs:64-82: 500 x 25-d, max_search 5, no reinsertion, searched at 40."""
    rng = np.random.default_rng(12)
    ix = oracle.build_index(random_vectors(oracle, rng, 500, 25), max_search=5, reinsert_elements=False)
    assert len(ix) == 500
    verify_search(ix, 0.95, 40)


def test_parallel_build_meets_the_same_bar(oracle):
    """rayon par_iter build (src/index/mod.rs:771-782) restated with OpenMP + per-node locks."""
    rng = np.random.default_rng(13)
    ix = oracle.build_index(random_vectors(oracle, rng, 1500, 28), num_neighbors=20, max_search=20, n_threads=4)
    verify_search(ix, 0.95, 10)


def test_select_neighbors(oracle):
    """src/index/tests.rs:11-39."""
    rng = np.random.default_rng(14)
    element = oracle.normalize_f32(random_floats(rng, 50))
    others = random_vectors(oracle, rng, 50, 50)
    cands = sorted(((oracle.dist(others[i], element), i) for i in range(50)))
    ids = [i for _, i in cands]
    ds = [d for d, _ in cands]
    nb = oracle.select_neighbors(others, ids, ds, 10)
    assert 0 < len(nb) <= 10
    assert all(nb[i - 1][1] <= nb[i][1] for i in range(1, len(nb)))
    nb = oracle.select_neighbors(others, ids, ds, 60)
    assert len(nb) == 50
    assert all(nb[i - 1][1] <= nb[i][1] for i in range(1, len(nb)))


def test_layer_structure(oracle):
    """Prefix-nested layers, width = num_neighbors everywhere, half degree above the bottom
    (src/index/mod.rs:393-398, 634-643, 665-668)."""
    rng = np.random.default_rng(15)
    ix = oracle.build_index(random_vectors(oracle, rng, 2000, 16), num_neighbors=10, max_search=30)
    sizes = [l.shape[0] for l in ix.layers]
    assert sizes == [oracle.num_elements_in_layer(2000, 15.0, l) for l in range(len(sizes))]
    assert sizes[-1] == 2000
    for l, layer in enumerate(ix.layers):
        assert layer.shape[1] == 10
        deg = (layer != oracle.UNUSED).sum(axis=1)
        assert deg.max() <= (10 if l == len(sizes) - 1 else 5)
        # valid ids form a prefix of each row and stay inside the layer
        for row, d in zip(layer, deg):
            assert (row[:d] != oracle.UNUSED).all() and (row[d:] == oracle.UNUSED).all()
            assert (row[:d] < layer.shape[0]).all()


def test_singlethreaded_build_is_deterministic(oracle):
    rng = np.random.default_rng(16)
    e = random_vectors(oracle, rng, 600, 20)
    a = oracle.build_index(e, num_neighbors=12, max_search=25)
    b = oracle.build_index(e, num_neighbors=12, max_search=25)
    assert all((x == y).all() for x, y in zip(a.layers, b.layers))


def test_empty_and_tiny_indexes(oracle):
    e = np.zeros((0, 8), np.float32)
    ix = oracle.build_index(e)
    assert len(ix.layers) == 0 and ix.search(np.zeros(8, np.float32), 10, 5) == []  # mod.rs:978-980
    rng = np.random.default_rng(17)
    e = random_vectors(oracle, rng, 2, 8)
    ix = oracle.build_index(e)
    assert len(ix) == 2
    r = ix.search(e[1], 5, 5)
    assert [i for i, _ in r][0] == 1 and len(r) == 2


def test_max_search_zero_is_an_error(oracle):
    rng = np.random.default_rng(18)
    ix = oracle.build_index(random_vectors(oracle, rng, 50, 8))
    with pytest.raises(RuntimeError):
        ix.search(ix.elements[0], 0, 1)


def test_zero_vectors_are_not_indexed(oracle):
    """src/index/mod.rs:813-815: zero vectors keep an all-UNUSED row and are never returned
    unless they are the entry point."""
    rng = np.random.default_rng(19)
    e = random_vectors(oracle, rng, 300, 16)
    e[100] = 0
    ix = oracle.build_index(e, num_neighbors=10, max_search=20)
    assert (ix.layers[-1][100] == oracle.UNUSED).all()
    assert not (ix.layers[-1] == 100).any()


# ---- C oracle vs independent Python restatement -----------------------------------------------
@pytest.mark.parametrize("int8", [False, True])
@pytest.mark.parametrize("max_search,k", [(1, 1), (5, 3), (20, 10), (50, 10), (7, 20)])
def test_search_matches_pyref(oracle, int8, max_search, k):
    rng = np.random.default_rng(20 + max_search + 100 * int8)
    e = random_vectors(oracle, rng, 400, 24, int8=int8)
    ix = oracle.build_index(e, num_neighbors=8, max_search=20, reinsert_elements=False)
    assert len(ix.layers) >= 2
    queries = random_vectors(oracle, rng, 8, 24, int8=int8)
    for q in queries:
        got, ctr = ix.search(q, max_search, k, counters=True)
        c = {"n_dist": 0, "n_expand": 0, "n_adj": 0}
        want = pyref.search(ix.layers, ix.elements, q, max_search, k, c)
        assert [i for i, _ in got] == [i for i, _ in want]
        assert np.array([d for _, d in got], np.float32).tobytes() == \
            np.array([d for _, d in want], np.float32).tobytes()
        assert ctr == (c["n_dist"], c["n_expand"], c["n_adj"])


def test_search_with_many_exact_ties_matches_pyref(oracle):
    """Duplicated int8 rows give exactly equal distances: exercises the asymmetric comparisons
    (break on `>`, enqueue on `<`, replace on tuple `<`)."""
    rng = np.random.default_rng(30)
    base = random_vectors(oracle, rng, 40, 16, int8=True)
    e = np.ascontiguousarray(base[rng.integers(0, 40, 300)])
    # hand-made graph (the builder would drop duplicates as dead nodes): random 6-regular rows
    layer = np.full((300, 8), oracle.UNUSED, np.uint32)
    for i in range(300):
        layer[i, :6] = rng.choice(300, 6, replace=False)
    top = np.full((10, 8), oracle.UNUSED, np.uint32)
    for i in range(10):
        top[i, :3] = rng.choice(10, 3, replace=False)
    ix = oracle.Index(e, [top, layer])
    for q in base[:10]:
        for ms in (1, 4, 16):
            got = ix.search(q, ms, ms)
            want = pyref.search(ix.layers, ix.elements, q, ms, ms)
            assert got == [(i, float(d)) for i, d in want]


def test_result_is_invariant_to_neighbor_order(oracle):
    """FixWidth (insertion order) vs Compressed (sorted order) must search identically
    (implied by src/index/tests.rs:337-451)."""
    rng = np.random.default_rng(31)
    e = random_vectors(oracle, rng, 800, 20)
    ix = oracle.build_index(e, num_neighbors=10, max_search=20)
    layers2 = []
    for layer in ix.layers:
        l2 = layer.copy()
        for row in l2:
            d = int((row != oracle.UNUSED).sum())
            row[:d] = np.sort(row[:d])
        layers2.append(l2)
    ix2 = oracle.Index(e, layers2)
    for q in random_vectors(oracle, rng, 20, 20):
        assert ix.search(q, 15, 10) == ix2.search(q, 15, 10)


def test_search_batch_equals_single_searches(oracle):
    rng = np.random.default_rng(32)
    e = random_vectors(oracle, rng, 1000, 20)
    ix = oracle.build_index(e, num_neighbors=10, max_search=20)
    q = random_vectors(oracle, rng, 33, 20)
    ids, ds, cnt, ctr = ix.search_batch(q, 12, 5, n_threads=3)
    for i in range(33):
        r, c = ix.search(q[i], 12, 5, counters=True)
        assert cnt[i] == len(r)
        assert ids[i, :cnt[i]].tolist() == [a for a, _ in r]
        assert ds[i, :cnt[i]].tolist() == [b for _, b in r]
        assert tuple(int(x) for x in ctr[i]) == c


def test_synth_rows_are_reproducible_and_in_range(oracle):
    a = oracle.synth_rows(0x6772616E6E65, 0, 64, 100)
    b = oracle.synth_rows(0x6772616E6E65, 32, 32, 100)
    assert (a[32:] == b).all()
    assert a.min() >= -0.5 and a.max() < 0.5
    assert abs(float(a.mean())) < 0.02


# ---- the batched insertion schedule (what the GPU builder implements) ---------------------------
@pytest.mark.parametrize("int8", [False, True])
def test_batched_build_meets_the_reference_bar(oracle, int8):
    """Same quality bar as the reference's build tests (src/index/tests.rs:41-62) under the
    batched schedule; deterministic whatever the thread count."""
    rng = np.random.default_rng(40 + int8)
    e = random_vectors(oracle, rng, 1500 if not int8 else 500, 28 if not int8 else 32, int8=int8)
    a = oracle.build_index(e, num_neighbors=20, max_search=20, batch_max=64, n_threads=1)
    b = oracle.build_index(e, num_neighbors=20, max_search=20, batch_max=64, n_threads=4)
    assert all((x == y).all() for x, y in zip(a.layers, b.layers))
    verify_search(a, 0.95, 10)
    sizes = [l.shape[0] for l in a.layers]
    assert sizes == [oracle.num_elements_in_layer(len(e), 15.0, l) for l in range(len(sizes))]


def test_batched_build_with_batch_of_one_is_the_sequential_build(oracle):
    rng = np.random.default_rng(42)
    e = random_vectors(oracle, rng, 700, 20)
    a = oracle.build_index(e, num_neighbors=10, max_search=20, n_threads=1)
    b = oracle.build_index(e, num_neighbors=10, max_search=20, batch_max=1)
    assert all((x == y).all() for x, y in zip(a.layers, b.layers))


def _verify_search(ix, num, max_search):
    """verify_search (src/index/tests.rs:50-62): fraction of elements that find themselves first."""
    n = len(ix)
    hits = sum(1 for i in range(0, n, max(1, n // num)) if ix.search(ix.elements[i], max_search, 1)[0][0] == i)
    return hits / len(range(0, n, max(1, n // num)))


def test_incremental_build_0(oracle):
    """src/index/tests.rs:134-168: layer counts after build_partial(12), (102) and build()."""
    rng = np.random.default_rng(41)
    b = oracle.Builder(random_vectors(oracle, rng, 1000, 5), layer_multiplier=10.0, num_neighbors=20, max_search=5)
    b.build_partial(12)
    assert b.layer_lens() == [10, 12]
    b.build_partial(102)
    assert b.layer_lens() == [10, 100, 102]
    b.build()
    assert b.layer_lens() == [10, 100, 1000]
    assert _verify_search(b.get_index(), 200, 5) > 0.95


def test_incremental_build_1(oracle):
    """src/index/tests.rs:170-192: ten chunks through build_partial on int8 vectors."""
    rng = np.random.default_rng(42)
    el = np.stack([oracle.quantize(r) for r in (rng.random((1000, 5), dtype=np.float32) - 0.5)])
    b = oracle.Builder(el, max_search=50)
    for i in range(1, 11):
        b.build_partial(i * 100)
        assert len(b) == i * 100
    ix = b.get_index()
    # 5-d int8 rows collide: a hit is the element itself or an exact duplicate at distance 0
    ok = sum(1 for i in range(0, 1000, 5) if ix.search(el[i], 5, 1)[0][0] == i or ix.search(el[i], 5, 1)[0][1] <= 1e-6)
    assert ok / 200 > 0.95


def test_empty_build(oracle):
    """src/index/tests.rs:293-302: build_partial(0) on a fresh builder does nothing."""
    rng = np.random.default_rng(43)
    b = oracle.Builder(random_vectors(oracle, rng, 100, 25))
    b.build_partial(0)
    assert b.layer_lens() == [] and len(b) == 0


def test_append_elements_with_expected_num_elements(oracle):
    """src/index/tests.rs:502-566: expected_num_elements(1000) fixes the layer sizes while only half of the
    elements have been pushed; the rest is indexed by a later build()."""
    rng = np.random.default_rng(44)
    el = random_vectors(oracle, rng, 1000, 50)
    b = oracle.Builder(el, expected_num_elements=1000, layer_multiplier=10.0, num_neighbors=20, max_search=50)
    b.build_partial(500)  # "insert half of the elements ... builder.build()"
    assert b.layer_lens() == [10, 100, 500]
    assert b.get_index().search(el[123], 50, 1)[0][0] == 123
    b.build()
    assert b.layer_lens() == [10, 100, 1000]
    ix = b.get_index()
    assert ix.search(el[123], 50, 1)[0][0] == 123 and ix.search(el[500 + 123], 50, 1)[0][0] == 500 + 123
    # without the hint the first half would have been laid out as a 500-element index
    c = oracle.Builder(el[:500], layer_multiplier=10.0, num_neighbors=20, max_search=50)
    c.build()
    assert c.layer_lens() == [5, 50, 500]


@pytest.mark.parametrize("n,dim,int8,nn,ms,mult,reinsert", [
    (220, 8, False, 10, 20, 6.0, True),
    (180, 16, True, 8, 15, 5.0, True),
    (200, 5, False, 6, 10, 4.0, False),
])
def test_sequential_build_matches_python_restatement(oracle, n, dim, int8, nn, ms, mult, reinsert):
    """The whole builder (src/index/mod.rs:364-402, 645-960) restated a second time in oracle/pyref.py::Builder:
    the graphs must be identical, layer by layer, id by id."""
    from oracle import pyref
    rng = np.random.default_rng(n + dim)
    el = random_vectors(oracle, rng, n, dim, int8)
    pb = pyref.Builder(el, num_neighbors=nn, max_search=ms, layer_multiplier=mult, reinsert_elements=reinsert)
    pb.build_partial(n)
    ob = oracle.build_index(el, num_neighbors=nn, max_search=ms, layer_multiplier=mult, reinsert_elements=reinsert,
                            n_threads=1)
    assert [len(l) for l in pb.layers] == [l.shape[0] for l in ob.layers]
    for got, want in zip(pb.rows(), ob.layers):
        assert (got == want).all()


def test_partial_builds_match_python_restatement(oracle):
    """build_partial in steps and expected_num_elements, C oracle vs oracle/pyref.py::Builder."""
    from oracle import pyref
    rng = np.random.default_rng(77)
    el = random_vectors(oracle, rng, 240, 6)
    kw = dict(num_neighbors=8, max_search=12, layer_multiplier=5.0)
    pb = pyref.Builder(el, expected_num_elements=400, **kw)
    ob = oracle.Builder(el, expected_num_elements=400, **kw)
    for step in (7, 60, 61, 240):
        pb.build_partial(step)
        ob.build_partial(step)
        assert [len(l) for l in pb.layers] == ob.layer_lens()
        for got, want in zip(pb.rows(), ob.get_index().layers):
            assert (got == want).all(), step


@pytest.mark.parametrize("n,dim,int8,nn,ms,mult,reinsert,bmax,bdiv", [
    (220, 8, False, 10, 20, 6.0, True, 16, 8),
    (180, 16, True, 8, 15, 5.0, True, 64, 4),
    (200, 5, False, 6, 10, 4.0, False, 7, 2),
])
def test_batched_build_matches_python_restatement(oracle, n, dim, int8, nn, ms, mult, reinsert, bmax, bdiv):
    """The batched insertion schedule (what the GPU builder runs; gro_build_config.batch_max) restated in
    oracle/pyref.py::Builder: same graphs from the C oracle for any thread count, also across build_partial steps."""
    from oracle import pyref
    rng = np.random.default_rng(n + dim + bmax)
    el = random_vectors(oracle, rng, n, dim, int8)
    kw = dict(num_neighbors=nn, max_search=ms, layer_multiplier=mult, reinsert_elements=reinsert, batch_max=bmax,
              batch_div=bdiv)
    pb = pyref.Builder(el, **kw)
    ob = oracle.Builder(el, n_threads=3, **kw)
    for step in (n // 2, n):
        pb.build_partial(step)
        ob.build_partial(step)
    want = ob.get_index().layers
    assert [len(l) for l in pb.layers] == [l.shape[0] for l in want]
    for got, w in zip(pb.rows(), want):
        assert (got == w).all()

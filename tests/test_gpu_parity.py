"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs. Bar: neighbor ids bit-exact, distances bit-exact (the f32 kernel keeps the
reference's operation order), result order ascending by (dist, id), counters equal.

Run on the MI355X box: python -m pytest tests -m gpu
"""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests.conftest import assert_counters, random_floats  # noqa: E402

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "*.npz")))
GOLDEN = [p for p in GOLDEN if not os.path.basename(p).startswith(("reorder_", "build_"))]


@pytest.fixture(scope="module")
def ga():
    import granne_amd
    from granne_amd import _lib, build
    assert os.path.exists(build.LIB_PATH), "libgranne_hip.so must be built in-tree (python -m granne_amd.build)"
    _lib.lib()
    return granne_amd


def prep(oracle, raw, int8):
    return oracle.quantize(raw) if int8 else oracle.normalize_f32(raw)


def assert_same(oracle_index, gpu_index, queries, max_search, k, check_stats=True, has_exact_set=True):
    """ids, distance bits, counts and counters against the oracle. An index whose visited-set option is on auto is
    walked in both forms: without a visited set (the default of the register walkers: n_dist counts evaluations,
    conftest.assert_counters) and with the exact tables (every counter the reference's)."""
    from granne_amd import _lib
    oi, od, oc, octr = oracle_index.search_batch(queries, max_search, k)
    mode = gpu_index.get_option(_lib.OPT_VISITED16)
    for m in ([0, 3] if mode == 0 else [mode]):
        gpu_index.set_option(_lib.OPT_VISITED16, m)
        try:
            ids, ds, cnt, st = gpu_index.search_batch(queries, max_search, k, stats=True)
        finally:
            gpu_index.set_option(_lib.OPT_VISITED16, mode)
        assert (cnt == oc).all(), (cnt, oc)
        for i in range(len(queries)):
            c = int(cnt[i])
            assert ids[i, :c].tolist() == oi[i, :c].tolist(), (m, i, ids[i, :c], oi[i, :c])
            assert ds[i, :c].tobytes() == od[i, :c].tobytes(), (m, i, ds[i, :c], od[i, :c])
            assert (ids[i, c:] == np.iinfo(np.uint64).max).all() and np.isinf(ds[i, c:]).all()
        if check_stats:
            # (the two-level lists -- max_search 1025..8192 -- keep no visited set whatever the option asks for)
            # (and has_exact_set=False: graphs of 64-id layers, walked without a set whatever the option asks for)
            assert_counters(st, octr, exact=(m in (1, 2, 3) and not 1024 < max_search <= 8192 and has_exact_set))
    return ids, ds, cnt


# ---- element preparation and the Dist operator ---------------------------------------------------
@pytest.mark.parametrize("dim", [1, 3, 25, 32, 100, 128, 200, 333])
def test_normalize_bit_exact(ga, oracle, dim):
    rng = np.random.default_rng(dim)
    raw = random_floats(rng, 777, dim)
    raw[5] = 0  # zero row stays zero (src/math.rs:134)
    assert ga.normalize(raw).tobytes() == oracle.normalize_f32(raw).tobytes()


@pytest.mark.parametrize("dim", [1, 3, 32, 100, 200, 333])
def test_quantize_bit_exact(ga, oracle, dim):
    rng = np.random.default_rng(1000 + dim)
    raw = random_floats(rng, 500, dim)
    raw[7] = 0
    assert (ga.quantize(raw) == oracle.quantize(raw)).all()


@pytest.mark.parametrize("int8", [False, True])
@pytest.mark.parametrize("dim", [3, 28, 100, 200, 257])
def test_dist_operator_bit_exact(ga, oracle, int8, dim):
    rng = np.random.default_rng(dim + 7 * int8)
    el = prep(oracle, random_floats(rng, 300, dim), int8)
    q = prep(oracle, random_floats(rng, 20, dim), int8)
    ix = ga.Granne("angular_int" if int8 else "angular", el, [])
    qi = rng.integers(0, 20, 2000)
    ei = rng.integers(0, 300, 2000)
    got = ix.dists(q, qi, ei)
    want = np.array([oracle.dist(el[e], q[a]) for a, e in zip(qi, ei)], np.float32)
    assert got.tobytes() == want.tobytes()


@pytest.mark.parametrize("int8", [False, True])
@pytest.mark.parametrize("dim", [1, 31, 32, 33, 100, 128, 200, 1500])
def test_dists_batched_bit_exact(ga, oracle, int8, dim):
    """ElementContainer::dists (src/elements/mod.rs:35-39): one element against many indices, batched."""
    rng = np.random.default_rng(11 * dim + int8)
    n, nq, m = 777, 13, 37
    el = prep(oracle, random_floats(rng, n, dim), int8)
    q = prep(oracle, random_floats(rng, nq, dim), int8)
    ix = ga.Granne("angular_int" if int8 else "angular", el, [])
    ids = rng.integers(0, n, (nq, m)).astype(np.uint32)
    ids[3, 5] = n            # out of range -> +inf
    ids[12, 36] = 0xFFFFFFFF  # UNUSED
    got = ix.dists_many(q, ids)
    want = np.array([[oracle.dist(el[e], q[a]) if e < n else np.inf for e in ids[a]] for a in range(nq)], np.float32)
    assert got.tobytes() == want.tobytes()
    assert ix.dists_many(q, np.zeros((nq, 0), np.uint32)).shape == (nq, 0)


def test_synthetic_rows_match_oracle(ga, oracle):
    import torch
    from granne_amd import _lib
    import ctypes as C
    t = torch.empty((1000, 100), dtype=torch.float32, device="cuda")
    _lib.check(_lib.lib().granne_hip_synth_rows_device(C.c_void_p(t.data_ptr()), 0x6772616E6E65, 17, 1000, 100, 0,
                                                       C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    assert t.cpu().numpy().tobytes() == oracle.synth_rows(0x6772616E6E65, 17, 1000, 100).tobytes()


def test_compute_distance(ga, oracle):
    """py/src/lib.rs:58-86."""
    rng = np.random.default_rng(3)
    a, b = random_floats(rng, 100), random_floats(rng, 100)
    assert ga.compute_distance("angular", a, b) == oracle.dist(oracle.normalize_f32(a), oracle.normalize_f32(b))
    assert ga.compute_distance("angular_int", a, b) == oracle.dist(oracle.quantize(a), oracle.quantize(b))
    with pytest.raises(ValueError):
        ga.compute_distance("euclid", a, b)


# ---- golden fixtures --------------------------------------------------------------------------------
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_gpu_reproduces_golden(ga, path):
    from granne_amd import _lib
    z = np.load(path)
    layers = [z["layer%d" % l] for l in range(int(z["n_layers"]))]
    et = "angular" if z["elements"].dtype == np.float32 else "angular_int"
    ix = ga.Granne(et, z["elements"], layers)
    assert len(ix) == layers[-1].shape[0] and ix.num_layers() == len(layers)
    for key in [k for k in z.files if k.startswith("ids_")]:
        ms, k = int(key.split("_")[1]), int(key.split("_")[2])
        for mode in (0, 3):  # as shipped: no visited set (n_dist counts evaluations); with the exact tables
            ix.set_option(_lib.OPT_VISITED16, mode)
            ids, ds, cnt, st = ix.search_batch(z["queries"], ms, k, stats=True)
            assert (cnt == z["counts_%d_%d" % (ms, k)]).all()
            want_ids, want_ds = z[key], z["dists_%d_%d" % (ms, k)]
            for i in range(len(cnt)):
                c = int(cnt[i])
                assert ids[i, :c].tolist() == want_ids[i, :c].tolist()
                assert ds[i, :c].tobytes() == want_ds[i, :c].tobytes()
            assert_counters(st, z["stats_%d_%d" % (ms, k)], exact=(mode == 3))
        ix.set_option(_lib.OPT_VISITED16, 0)


# ---- search parity on seeded random indexes ---------------------------------------------------------
CASES = [
    # n, dim, int8, num_neighbors, build max_search
    (2000, 100, False, 30, 30),
    (1200, 200, False, 30, 30),
    (1500, 28, False, 20, 20),
    (800, 24, False, 8, 20),
    (600, 3, False, 10, 20),
    (2000, 100, True, 30, 30),
    (500, 32, True, 20, 20),
    (700, 130, True, 12, 20),
    # int8 rows of 256 and 512 bytes on the register walker (the reference benches 200- and 300-d int8 rows,
    # benches/distance_computation.rs:29-39); 513+ dims take the general walker
    (700, 200, True, 20, 20),
    (600, 300, True, 20, 20),
    (500, 512, True, 16, 20),
    (300, 600, True, 16, 20),
    # f32 dims that take the streamed run-time-dim walker (walk_fast.h, DIM = 0): one chunk exactly, chunks + tails
    # of every length class, not a multiple of 4 (zero-padded tail unit), more than one group of three chunks
    (900, 32, False, 30, 30),
    (900, 50, False, 30, 30),
    (900, 96, False, 30, 30),
    (900, 97, False, 30, 30),
    (700, 128, False, 20, 20),
    (600, 300, False, 30, 30),
    (400, 768, False, 30, 20),
    # dims below 32 on the same walker: no full chunk, the row is its tail (the reference benches 3-d rows)
    (500, 1, False, 8, 10),
    (500, 2, False, 8, 10),
    (500, 5, False, 12, 20),
    (600, 16, False, 16, 20),
    (600, 31, False, 16, 20),
]


@pytest.mark.parametrize("n,dim,int8,nn,ms", CASES)
def test_search_parity(ga, oracle, n, dim, int8, nn, ms):
    rng = np.random.default_rng(n + dim)
    el = prep(oracle, random_floats(rng, n, dim), int8)
    oix = oracle.build_index(el, num_neighbors=nn, max_search=ms, n_threads=4)
    gix = ga.Granne("angular_int" if int8 else "angular", el, oix.layers)
    q = prep(oracle, random_floats(rng, 64, dim), int8)
    for max_search, k in [(1, 1), (7, 3), (50, 10), (64, 64), (65, 10), (128, 128), (200, 10), (256, 40)]:
        assert_same(oix, gix, q, max_search, k)
    # the elements themselves as queries (verify_search, src/index/tests.rs:50-62)
    ids, _, _ = assert_same(oix, gix, el[:200], 20, 1)
    if dim >= 3:  # (in 1 or 2 dimensions normalised rows coincide: the smallest id among equals wins, for both sides)
        assert (ids[:, 0] == np.arange(200)).mean() > 0.95


def test_get_element_and_neighbors(ga, oracle):
    rng = np.random.default_rng(5)
    el = prep(oracle, random_floats(rng, 300, 20), False)
    oix = oracle.build_index(el, num_neighbors=10, max_search=20)
    gix = ga.Granne("angular", el, oix.layers)
    assert len(gix) == 300 and gix.num_layers() == len(oix.layers)
    for l, layer in enumerate(oix.layers):
        assert gix.layer_len(l) == layer.shape[0]
        for i in (0, layer.shape[0] - 1):
            d = int((layer[i] != oracle.UNUSED).sum())
            assert gix.get_neighbors(i, l) == layer[i, :d].tolist()
    assert gix.get_neighbors(7) == oix.layers[-1][7, :int((oix.layers[-1][7] != oracle.UNUSED).sum())].tolist()
    assert (gix.get_element(17) == el[17]).all()


def test_raw_query_goes_through_vector_from(ga, oracle):
    """py/src/variants/index.rs:8-17: the binding normalises/quantises the query first."""
    rng = np.random.default_rng(6)
    raw = random_floats(rng, 500, 32)
    for int8 in (False, True):
        el = prep(oracle, raw, int8)
        oix = oracle.build_index(el, num_neighbors=10, max_search=20)
        gix = ga.Granne("angular_int" if int8 else "angular", raw, oix.layers, prepared=False)
        rq = random_floats(rng, 32)
        assert gix.search(rq, 30, 5, prepared=False) == oix.search(prep(oracle, rq, int8), 30, 5)


# ---- edge cases -------------------------------------------------------------------------------------
def test_empty_index(ga):
    ix = ga.Granne("angular", np.zeros((0, 8), np.float32), [])
    assert len(ix) == 0 and ix.num_layers() == 0
    assert ix.search(np.zeros(8, np.float32), 10, 5) == []  # src/index/mod.rs:978-980


def test_one_and_two_elements(ga, oracle):
    rng = np.random.default_rng(7)
    for n in (1, 2, 3):
        el = prep(oracle, random_floats(rng, n, 8), False)
        oix = oracle.build_index(el)
        gix = ga.Granne("angular", el, oix.layers)
        q = prep(oracle, random_floats(rng, 4, 8), False)
        assert_same(oix, gix, q, 5, 5)
        assert_same(oix, gix, q, 1, 1)


def test_max_search_zero_is_an_error(ga, oracle):
    rng = np.random.default_rng(8)
    el = prep(oracle, random_floats(rng, 50, 8), False)
    gix = ga.Granne("angular", el, oracle.build_index(el).layers)
    with pytest.raises(ga.GranneHipError) as e:
        gix.search(el[0], 0, 1)
    assert e.value.code == -1 and "max_search" in str(e.value)


def test_k_larger_than_max_search_and_than_index(ga, oracle):
    rng = np.random.default_rng(9)
    el = prep(oracle, random_floats(rng, 40, 16), False)
    oix = oracle.build_index(el, num_neighbors=6, max_search=10)
    gix = ga.Granne("angular", el, oix.layers)
    q = prep(oracle, random_floats(rng, 8, 16), False)
    ids, ds, cnt = assert_same(oix, gix, q, 5, 20)   # count = min(k, max_search)
    assert (cnt == 5).all()
    ids, ds, cnt = assert_same(oix, gix, q, 100, 100)  # count = reachable nodes
    assert (cnt <= 40).all()


def test_unindexed_zero_rows_and_partial_index(ga, oracle):
    """Zero vectors keep all-UNUSED rows (src/index/mod.rs:813-815); Granne::len() may be smaller
    than elements.len() (:76-83)."""
    rng = np.random.default_rng(10)
    el = prep(oracle, random_floats(rng, 400, 16), False)
    el[[0, 50, 399]] = 0
    oix = oracle.build_index(el, num_neighbors=8, max_search=20, num_elements=300)
    assert len(oix) == 300
    gix = ga.Granne("angular", el, oix.layers)
    assert len(gix) == 300
    q = prep(oracle, random_floats(rng, 16, 16), False)
    assert_same(oix, gix, q, 20, 10)  # entry point 0 is itself a zero vector here


def test_wide_rows_and_large_degree(ga, oracle):
    """num_neighbors up to 254 is legal (u8 count in the file format): rows wider than a wave."""
    rng = np.random.default_rng(11)
    el = prep(oracle, random_floats(rng, 600, 12), False)
    layer = np.full((600, 100), oracle.UNUSED, np.uint32)
    for i in range(600):
        d = int(rng.integers(0, 101))
        layer[i, :d] = rng.choice(600, d, replace=False)
    layer[0, :100] = rng.choice(600, 100, replace=False)
    top = np.full((12, 100), oracle.UNUSED, np.uint32)
    for i in range(12):
        top[i, :4] = rng.choice(12, 4, replace=False)
    assert int((layer != oracle.UNUSED).sum(axis=1).max()) > 64
    oix = oracle.Index(el, [top, layer])
    gix = ga.Granne("angular", el, [top, layer])
    q = prep(oracle, random_floats(rng, 16, 12), False)
    assert_same(oix, gix, q, 30, 10)
    assert_same(oix, gix, q, 3, 3)


@pytest.mark.parametrize("case", ["f32_100", "i8_100", "f32_200", "f32_gen48"])
@pytest.mark.parametrize("nn", [40, 63])
def test_layers_of_up_to_64_ids_stay_on_the_register_walker(ga, oracle, case, nn):
    """BuildConfig::num_neighbors beyond 32 (src/index/mod.rs:242-250; the GPU builder takes up to 63): rows of 64 ids on
    the device, walked by the register walker in two passes per expansion for max_search up to 1024."""
    from granne_amd import _lib
    int8 = case.startswith("i8")
    dim = {"f32_100": 100, "i8_100": 100, "f32_200": 200, "f32_gen48": 48}[case]
    rng = np.random.default_rng(640 + nn)
    el = prep(oracle, random_floats(rng, 5000, dim), int8)
    oix = oracle.build_index(el, num_neighbors=nn, max_search=60, n_threads=4)
    assert max(int((l != oracle.UNUSED).sum(axis=1).max()) for l in oix.layers) > 32
    gix = ga.Granne("angular_int" if int8 else "angular", el, oix.layers)
    q = prep(oracle, random_floats(rng, 64, dim), int8)
    for ms in (1, 50, 100, 200, 252, 300, 600, 1024):  # (round 5: lists of up to 17 x 64 keys for these graphs too)
        assert_same(oix, gix, q, ms, 10, has_exact_set=False)
        assert gix.get_option(_lib.OPT_LAST_WALKER) == _lib.WALKER_REGISTER_WIDE, ms
        if not int8:
            assert gix.last_slow_count() == 0
    assert_same(oix, gix, q[:8], 1100, 10)  # longer lists of such graphs: the exact walker
    assert gix.get_option(_lib.OPT_LAST_WALKER) == _lib.WALKER_EXACT


def test_rows_of_64_ids_with_every_prefix_length(ga, oracle):
    """Hand-made rows of 0..64 ids -- exactly 31, 32, 33 and 64 among them, a row that names a node in both halves, one
    that names it twice in its second half -- on the two-pass register walker (12-d rows: the streamed f32 shape)."""
    from granne_amd import _lib
    rng = np.random.default_rng(641)
    n = 700
    el = prep(oracle, random_floats(rng, n, 12), False)
    layer = np.full((n, 64), oracle.UNUSED, np.uint32)
    for i in range(n):
        d = [31, 32, 33, 64, 0, 1][i] if i < 6 else int(rng.integers(0, 65))
        layer[i, :d] = rng.choice(n, d, replace=False)
    layer[6, :40] = rng.choice(n, 40, replace=False)
    layer[6, 39] = layer[6, 3]           # the same node in both halves of the row
    layer[7, :40] = rng.choice(n, 40, replace=False)
    layer[7, 38] = layer[7, 35]          # twice in the second half
    top = np.full((12, 64), oracle.UNUSED, np.uint32)
    for i in range(12):
        top[i, :5] = rng.choice(12, 5, replace=False)
    oix = oracle.Index(el, [top, layer])
    gix = ga.Granne("angular", el, [top, layer])
    q = prep(oracle, random_floats(rng, 32, 12), False)
    for ms in (30, 3, 120):
        assert_same(oix, gix, q, ms, 10, has_exact_set=False)
        assert gix.get_option(_lib.OPT_LAST_WALKER) == _lib.WALKER_REGISTER_WIDE
    assert_same(oix, gix, el[:8], 60, 10, has_exact_set=False)  # member queries: the walk starts among the hand-made rows


def test_csr_layers_equal_fixed_width_layers(ga, oracle):
    """Layers::Compressed (sorted ids, src/slice_vector/set_vector.rs:40-46) and Layers::FixWidth
    search identically."""
    rng = np.random.default_rng(12)
    el = prep(oracle, random_floats(rng, 900, 20), False)
    oix = oracle.build_index(el, num_neighbors=10, max_search=20)
    offsets, ids = [], []
    for layer in oix.layers:
        deg = (layer != oracle.UNUSED).sum(axis=1)
        offsets.append(np.concatenate([[0], np.cumsum(deg)]).astype(np.uint64))
        ids.append(np.concatenate([np.sort(r[:d]) for r, d in zip(layer, deg)]).astype(np.uint32))
    gix = ga.Granne.from_csr("angular", el, offsets, ids)
    q = prep(oracle, random_floats(rng, 32, 20), False)
    assert_same(oix, gix, q, 25, 10)


# ---- the two hand-over conditions and the exact global-memory walker -------------------------------
def test_slow_path_equals_fast_path(ga, oracle):
    from granne_amd import _lib
    rng = np.random.default_rng(13)
    for int8 in (False, True):
        el = prep(oracle, random_floats(rng, 1500, 100), int8)
        oix = oracle.build_index(el, num_neighbors=20, max_search=20, n_threads=4)
        gix = ga.Granne("angular_int" if int8 else "angular", el, oix.layers)
        q = prep(oracle, random_floats(rng, 40, 100), int8)
        assert_same(oix, gix, q, 50, 10)
        assert gix.last_slow_count() == 0
        gix.set_option(_lib.OPT_FORCE_SLOW, 1)
        assert_same(oix, gix, q, 50, 10)
        assert gix.last_slow_count() == 40
        assert_same(oix, gix, q, 300, 50)
        gix.set_option(_lib.OPT_FORCE_SLOW, 0)
        assert_same(oix, gix, q, 300, 50)  # the register walker takes max_search up to 8192 (walk_fast.h)
        assert gix.last_slow_count() == 0
        assert_same(oix, gix, q, 1100, 50)
        assert gix.last_slow_count() == 0
        assert_same(oix, gix, q, 4200, 50)
        assert gix.last_slow_count() == 0
        assert_same(oix, gix, q, 8300, 50)  # beyond that: always the exact global-memory walker
        assert gix.last_slow_count() == 40


@pytest.mark.parametrize("int8", [False, True])
@pytest.mark.parametrize("ms", [61, 64, 125, 128, 253, 300, 512, 1000, 1024, 1025, 1500, 2048, 2049, 3000, 4096, 4097, 6000, 8192])
def test_large_max_search_stays_on_the_register_walker(ga, oracle, int8, ms):
    """The reference takes any max_search (src/index/mod.rs:1006-1010). Up to 8192 the walk stays in
    registers/LDS (lists of 64..8256 keys); only distance ties at the list's end may hand a walk over.
    The two-level lists (max_search beyond 1024: M in LDS, F in registers) are walked on more points than their lists hold."""
    rng = np.random.default_rng(1000 + ms)
    el = prep(oracle, random_floats(rng, 4000 if ms <= 1024 else 9000 if ms <= 4096 else 14000, 100), int8)
    oix = oracle.build_index(el, num_neighbors=30, max_search=40, n_threads=4)
    gix = ga.Granne("angular_int" if int8 else "angular", el, oix.layers)
    q = prep(oracle, random_floats(rng, 48 if ms <= 1024 else 24, 100), int8)
    assert_same(oix, gix, q, ms, 20)
    assert_same(oix, gix, q, ms, ms)
    if not int8:
        assert gix.last_slow_count() == 0


@pytest.mark.parametrize("case", ["f32_96", "f32_300", "f32_768", "i8_200", "i8_300"])
@pytest.mark.parametrize("ms", [300, 600, 1024])
def test_every_shape_of_the_register_walker_takes_max_search_1024(ga, oracle, case, ms):
    """The dims the reference benches besides 100 (benches/distance_computation.rs:29-39: 50 / 300; real embeddings: 96,
    300, 384, 768) walk on the streamed-dim walker, int8 rows of 129..512 dims on the 256- / 512-byte walkers: round 4 took
    them to max_search 508 / 252 and dropped to the exact walker beyond (a 20x cliff); they now hold lists of up to
    17 x 64 keys like the unrolled shapes."""
    from granne_amd import _lib
    int8 = case.startswith("i8")
    dim = int(case.split("_")[1])
    rng = np.random.default_rng(5100 + dim + ms)
    el = prep(oracle, random_floats(rng, 3000, dim), int8)
    oix = oracle.build_index(el, num_neighbors=20, max_search=30, n_threads=4)
    gix = ga.Granne("angular_int" if int8 else "angular", el, oix.layers)
    q = prep(oracle, random_floats(rng, 24, dim), int8)
    assert_same(oix, gix, q, ms, 10)
    assert gix.get_option(_lib.OPT_LAST_WALKER) == _lib.WALKER_REGISTER
    if not int8:
        assert gix.last_slow_count() == 0


def test_long_lists_on_200d_rows_and_with_an_exact_set_requested(ga, oracle):
    """max_search 1500 / 4096 on 800-byte rows; the long lists keep no visited set whatever OPT_VISITED16 asks for."""
    from granne_amd import _lib
    rng = np.random.default_rng(77)
    el = prep(oracle, random_floats(rng, 6000, 200), False)
    oix = oracle.build_index(el, num_neighbors=20, max_search=30, n_threads=4)
    gix = ga.Granne("angular", el, oix.layers)
    q = prep(oracle, random_floats(rng, 16, 200), False)
    for ms in (1500, 4096):
        assert_same(oix, gix, q, ms, 10, check_stats=False)
        assert gix.last_slow_count() == 0
    gix.set_option(_lib.OPT_VISITED16, 3)
    oi, od, oc, _ = oix.search_batch(q, 2000, 10)
    ids, ds, cnt = gix.search_batch(q, 2000, 10)
    assert (ids == oi).all() and ds.tobytes() == od.tobytes() and gix.last_slow_count() == 0


def test_max_search_beyond_the_register_lists(ga, oracle):
    """max_search above 8192 (above 1024 for wide int8 rows and streamed f32 dims) is the exact global-memory
    walker's as a whole batch -- its own launch, one block per query up to 32 x OPT_SLOW_BLOCKS. Same results."""
    rng = np.random.default_rng(41)
    for int8, dim, ms in [(False, 100, 8500), (True, 100, 8193), (True, 200, 1100), (False, 50, 1025)]:
        el = prep(oracle, random_floats(rng, 5000, dim), int8)
        oix = oracle.build_index(el, num_neighbors=20, max_search=20, n_threads=4)
        gix = ga.Granne("angular_int" if int8 else "angular", el, oix.layers)
        q = prep(oracle, random_floats(rng, 200, dim), int8)
        assert_same(oix, gix, q, ms, 10)
        assert gix.last_slow_count() == 200
        assert_same(oix, gix, q[:3], ms, 10)


def test_visited_table_overflow_hands_over(ga, oracle):
    from granne_amd import _lib
    rng = np.random.default_rng(14)
    el = prep(oracle, random_floats(rng, 3000, 32), False)
    oix = oracle.build_index(el, num_neighbors=20, max_search=20, n_threads=4)
    gix = ga.Granne("angular", el, oix.layers)
    q = prep(oracle, random_floats(rng, 64, 32), False)
    gix.set_option(_lib.OPT_VISITED_SLOTS, 256)  # far too small for max_search=100
    gix.set_option(_lib.OPT_OVERFLOW_SLOTS, 1)   # no overflow table: the walks are handed over
    assert_same(oix, gix, q, 100, 10)
    assert gix.last_slow_count() > 0
    gix.set_option(_lib.OPT_VISITED_SLOTS, 0)
    gix.set_option(_lib.OPT_OVERFLOW_SLOTS, 0)
    assert_same(oix, gix, q, 100, 10)
    assert gix.last_slow_count() == 0


@pytest.mark.parametrize("int8", [False, True])
@pytest.mark.parametrize("ef,lds_slots,ovf_slots", [(100, 256, 0), (100, 256, 4096), (250, 1024, 0), (40, 256, 2048),
                                                    (200, 512, 512), (100, 384, 0), (60, 768, 0), (100, 1536, 0)])
def test_visited_set_spills_to_global_overflow(ga, oracle, int8, ef, lds_slots, ovf_slots):
    """A full LDS visited table freezes and the walk continues with a global overflow table: same results,
    no hand-over -- unless the overflow table is too small as well (512 slots at ef=200), which hands over."""
    import torch
    from granne_amd import _lib
    rng = np.random.default_rng(140 + ef + int8)
    el = prep(oracle, random_floats(rng, 4000, 100), int8)
    oix = oracle.build_index(el, num_neighbors=20, max_search=20, n_threads=4)
    gix = ga.Granne("angular_int" if int8 else "angular", el, oix.layers)
    q = prep(oracle, random_floats(rng, 96, 100), int8)
    gix.set_option(_lib.OPT_VISITED_SLOTS, lds_slots)
    gix.set_option(_lib.OPT_OVERFLOW_SLOTS, ovf_slots)
    assert_same(oix, gix, q, ef, 10)
    slow = gix.last_slow_count()
    assert (slow > 0) == (ef == 200 and ovf_slots == 512)  # (tables of 3 * 2^k slots: 384, 768, 1536)
    # the device-side status words: [1] slow-path queries, [2] walks that spilled
    tq = torch.from_numpy(q.view(np.uint8).reshape(96, -1)).cuda()
    ids = torch.empty((96, 10), dtype=torch.int64, device="cuda")
    ds = torch.empty((96, 10), dtype=torch.float32, device="cuda")
    cnt = torch.empty(96, dtype=torch.int32, device="cuda")
    status = torch.zeros(4, dtype=torch.int32, device="cuda")
    gix.search_batch_device(tq.data_ptr(), 96, ef, 10, ids.data_ptr(), ds.data_ptr(), cnt.data_ptr(), 0,
                            status.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    st = status.cpu().numpy()
    assert st[0] == 0 and st[1] == slow and st[2] > 0
    want = oix.search_batch(q, ef, 10)
    assert (ids.cpu().numpy().astype(np.uint64) == want[0]).all()


@pytest.mark.parametrize("int8", [False, True])
@pytest.mark.parametrize("mode,ovf", [(0, 0), (4, 0), (1, 0), (3, 0), (3, 2048), (2, 1)])
def test_visited_set_modes(ga, oracle, int8, mode, ovf):
    """The register walkers' visited set: none (mode 0 = 4, the default: the list is searched by id) or the exact 32-bit
    table (modes 1..3; rounds 3a-3 had three forms of it), with the default overflow pool, a small one, and none (walks
    whose table fills are handed to the exact walker). Same ids and distance bits in every mode, the reference's counters
    in every mode that keeps a set."""
    from granne_amd import _lib
    rng = np.random.default_rng(160 + 3 * mode + int8)
    el = prep(oracle, random_floats(rng, 6000, 100), int8)
    oix = oracle.build_index(el, num_neighbors=20, max_search=20, n_threads=4)
    gix = ga.Granne("angular_int" if int8 else "angular", el, oix.layers)
    q = prep(oracle, random_floats(rng, 128, 100), int8)
    gix.set_option(_lib.OPT_VISITED16, mode)
    gix.set_option(_lib.OPT_VISITED16_LG, 7)  # retired: accepted, ignored
    gix.set_option(_lib.OPT_OVERFLOW_SLOTS, ovf)
    for ef, k in [(1, 1), (50, 10), (100, 10), (200, 10), (250, 30)]:
        assert_same(oix, gix, q, ef, k)
        if ovf == 0:
            assert gix.last_slow_count() == 0, ef
    # members as queries, duplicates of one query in a batch, a batch of one
    assert_same(oix, gix, el[:64], 30, 5)
    assert_same(oix, gix, np.repeat(q[:1], 5, axis=0), 60, 10)
    assert_same(oix, gix, q[:1], 60, 10)


def test_visited16_duplicate_neighbor_ids_in_a_row(ga, oracle):
    """A row that lists the same neighbor twice: the reference's HashSet::insert is false the second time
    (src/index/mod.rs:1026); the pair of lanes that comes second must see the first pair's entry."""
    from granne_amd import _lib
    rng = np.random.default_rng(77)
    el = prep(oracle, random_floats(rng, 3000, 100), False)
    oix = oracle.build_index(el, num_neighbors=20, max_search=20, n_threads=4)
    layers = [l.copy() for l in oix.layers]
    bottom = layers[-1]
    for r in range(0, len(bottom), 3):  # duplicate the first neighbor into the last used place (or the next free one)
        row = bottom[r]
        used = int((row != 0xFFFFFFFF).sum())
        if used >= 2:
            row[min(used, len(row) - 1)] = row[0]
    dup = oracle.Index(el, layers)
    q = prep(oracle, random_floats(rng, 64, 100), False)
    for mode in (0, 1, 2, 3, 4):
        gix = ga.Granne("angular", el, layers)
        gix.set_option(_lib.OPT_VISITED16, mode)
        assert_same(dup, gix, q, 50, 10)
        assert_same(dup, gix, q, 150, 10)


def test_slow_scratch_exhaustion_is_reported(ga, oracle):
    from granne_amd import _lib
    rng = np.random.default_rng(15)
    el = prep(oracle, random_floats(rng, 3000, 16), False)
    oix = oracle.build_index(el, num_neighbors=20, max_search=20, n_threads=4)
    gix = ga.Granne("angular", el, oix.layers)
    gix.set_option(_lib.OPT_FORCE_SLOW, 1)
    gix.set_option(_lib.OPT_SLOW_SLOTS, 256)
    with pytest.raises(ga.GranneHipError) as e:
        gix.search_batch(el[:8], 200, 10)
    assert e.value.code == _lib.ERR_OVERFLOW
    gix.set_option(_lib.OPT_SLOW_SLOTS, 1 << 16)
    assert_same(oix, gix, el[:8], 200, 10)


def test_exact_ties_from_duplicate_vectors(ga, oracle):
    """Duplicated int8 rows produce exactly equal distances. The comparisons are asymmetric
    (break on `>`, enqueue on `<`, replace on tuple `<`): every query must
    still agree with the oracle, whichever walker ends up serving it."""
    rng = np.random.default_rng(16)
    base = oracle.quantize(random_floats(rng, 30, 16))
    el = np.ascontiguousarray(base[rng.integers(0, 30, 2000)])
    layer = np.full((2000, 32), oracle.UNUSED, np.uint32)
    for i in range(2000):
        layer[i, :24] = rng.choice(2000, 24, replace=False)
    top = np.full((20, 32), oracle.UNUSED, np.uint32)
    for i in range(20):
        top[i, :5] = rng.choice(20, 5, replace=False)
    oix = oracle.Index(el, [top, layer])
    gix = ga.Granne("angular_int", el, [top, layer])
    for ms, k in [(1, 1), (4, 4), (16, 16), (50, 10), (64, 64), (100, 100)]:
        assert_same(oix, gix, base, ms, k)
    # the same with float rows (exact duplicates -> exact ties)
    fb = oracle.normalize_f32(random_floats(rng, 30, 100))
    fel = np.ascontiguousarray(fb[rng.integers(0, 30, 2000)])
    oix = oracle.Index(fel, [top, layer])
    gix = ga.Granne("angular", fel, [top, layer])
    for ms, k in [(1, 1), (16, 16), (50, 10), (100, 100)]:
        assert_same(oix, gix, fb, ms, k)


def test_large_batch_and_repeatability(ga, oracle):
    rng = np.random.default_rng(17)
    el = prep(oracle, random_floats(rng, 20000, 100), False)
    oix = oracle.build_index(el, num_neighbors=30, max_search=40, reinsert_elements=False, n_threads=0)
    gix = ga.Granne("angular", el, oix.layers)
    q = prep(oracle, random_floats(rng, 4096, 100), False)
    a = assert_same(oix, gix, q, 50, 10)
    b = gix.search_batch(q, 50, 10)
    assert (a[0] == b[0]).all() and a[1].tobytes() == b[1].tobytes()


def test_device_resident_api(ga, oracle):
    """granne_hip_search_batch_device with torch-owned buffers on torch's current stream."""
    import torch
    rng = np.random.default_rng(18)
    el = prep(oracle, random_floats(rng, 5000, 100), False)
    oix = oracle.build_index(el, num_neighbors=20, max_search=20, reinsert_elements=False, n_threads=0)
    gix = ga.Granne("angular", el, oix.layers)
    q = prep(oracle, random_floats(rng, 256, 100), False)
    dq = torch.from_numpy(q).cuda()
    ids = torch.empty((256, 10), dtype=torch.int64, device="cuda")
    ds = torch.empty((256, 10), dtype=torch.float32, device="cuda")
    cnt = torch.empty(256, dtype=torch.int32, device="cuda")
    st = torch.empty((256, 3), dtype=torch.int64, device="cuda")
    status = torch.zeros(4, dtype=torch.int32, device="cuda")
    gix.search_batch_device(dq.data_ptr(), 256, 50, 10, ids.data_ptr(), ds.data_ptr(), cnt.data_ptr(),
                            st.data_ptr(), status.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    oi, od, oc, octr = oix.search_batch(q, 50, 10)
    assert (ids.cpu().numpy().astype(np.uint64) == oi).all()
    assert ds.cpu().numpy().tobytes() == od.tobytes()
    assert (cnt.cpu().numpy().astype(np.uint32) == oc).all()
    assert_counters(st.cpu().numpy(), octr, exact=False)
    assert status.tolist() == [0, 0, 0, 0]


@pytest.mark.parametrize("case", ["f32_100", "i8_100", "f32_gen33", "i8_wide600", "forced_slow", "ties_hand_over"])
@pytest.mark.parametrize("n_batches", [1, 3, 35])
def test_several_batches_in_one_launch(ga, oracle, case, n_batches):
    """granne_hip_search_batches_device: a grid of n_batches x nq walkers (35 batches: two launches). Results, counts and
    counters are those of n_batches separate searches -- on the register walker, the general walker (int8 rows of 640
    bytes), the exact walker (every query handed over: the tail blocks address batches through the same table) and with
    tied distances (hand-overs inside a batched launch)."""
    import torch
    from granne_amd import _lib
    rng = np.random.default_rng(2600 + n_batches)
    int8 = case.startswith("i8")
    dim = {"f32_100": 100, "i8_100": 100, "f32_gen33": 33, "i8_wide600": 600, "forced_slow": 100, "ties_hand_over": 28}[case]
    n, nq, ef, k = 3000, 40, 30, 7
    raw = random_floats(rng, n, dim)
    if case == "ties_hand_over":
        raw[n // 2:] = raw[: n - n // 2]  # every vector twice: exact distance ties
    el = prep(oracle, raw, int8)
    oix = oracle.build_index(el, num_neighbors=12, max_search=20, reinsert_elements=False, n_threads=0)
    gix = ga.Granne("angular_int" if int8 else "angular", el, oix.layers)
    if case == "forced_slow":
        gix.set_option(_lib.OPT_FORCE_SLOW, 1)
    q = prep(oracle, random_floats(rng, n_batches * nq, dim), int8)
    tdt = torch.int8 if int8 else torch.float32
    # batches deliberately NOT contiguous with each other: separate allocations, one per batch
    dq = [torch.from_numpy(q[b * nq:(b + 1) * nq].copy()).cuda() for b in range(n_batches)]
    ids = [torch.full((nq, k), -7, dtype=torch.int64, device="cuda") for _ in range(n_batches)]
    ds = [torch.zeros((nq, k), dtype=torch.float32, device="cuda") for _ in range(n_batches)]
    cnt = [torch.full((nq,), 99, dtype=torch.int32, device="cuda") for _ in range(n_batches)]
    st = [torch.zeros((nq, 3), dtype=torch.int64, device="cuda") for _ in range(n_batches)]
    assert dq[0].dtype == tdt
    status = torch.zeros(4, dtype=torch.int32, device="cuda")
    ptrs = lambda ts: [t.data_ptr() for t in ts]  # noqa: E731
    gix.search_batches_device(ptrs(dq), nq, ef, k, ptrs(ids), ptrs(ds), ptrs(cnt), ptrs(st), status.data_ptr(),
                              torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    oi, od, oc, octr = oix.search_batch(q, ef, k)
    for b in range(n_batches):
        sl = slice(b * nq, (b + 1) * nq)
        assert (ids[b].cpu().numpy().astype(np.uint64) == oi[sl]).all(), (case, b)
        assert ds[b].cpu().numpy().tobytes() == od[sl].tobytes(), (case, b)
        assert (cnt[b].cpu().numpy().astype(np.uint32) == oc[sl]).all()
        exact = case in ("forced_slow", "i8_wide600")  # walkers that keep an exact visited set
        assert_counters(st[b].cpu().numpy(), octr[sl], exact=exact)
    assert status[0].item() == 0
    if case == "forced_slow":
        assert status[1].item() == n_batches * nq
    # without statistics, and with num_neighbors 0 (.take(0)): counts only
    for t in ids:
        t.fill_(-7)
    gix.search_batches_device(ptrs(dq), nq, ef, k, ptrs(ids), ptrs(ds), ptrs(cnt), None, 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert all((ids[b].cpu().numpy().astype(np.uint64) == oi[b * nq:(b + 1) * nq]).all() for b in range(n_batches))
    gix.search_batches_device(ptrs(dq), nq, ef, 0, ptrs(ids), ptrs(ds), ptrs(cnt), None, 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert all(int(c.abs().sum().item()) == 0 for c in cnt)


def test_begin_end_keeps_batches_in_flight_beside_one_stream(ga, oracle):
    """granne_hip_search_begin_device / _end_device: up to GRANNE_HIP_SEARCH_DEPTH batches begun and not ended; the
    caller's stream is ordered after each batch by its end; results are those of plain calls; tickets are checked."""
    import torch
    from granne_amd import _lib
    rng = np.random.default_rng(2800)
    el = prep(oracle, random_floats(rng, 4000, 100), True)
    oix = oracle.build_index(el, num_neighbors=16, max_search=20, reinsert_elements=False, n_threads=0)
    gix = ga.Granne("angular_int", el, oix.layers)
    nb, nq, D = 8, 96, _lib.SEARCH_DEPTH
    q = prep(oracle, random_floats(rng, nb * nq, 100), True)
    oi, od, oc, _ = oix.search_batch(q, 40, 10)
    dq = torch.from_numpy(q).cuda()
    ids = torch.zeros((nb, nq, 10), dtype=torch.int64, device="cuda")
    ds = torch.zeros((nb, nq, 10), dtype=torch.float32, device="cuda")
    cnt = torch.zeros((nb, nq), dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    tickets = []
    for b in range(nb):
        tickets.append(gix.search_begin_device(dq[b * nq:(b + 1) * nq].data_ptr(), nq, 40, 10, ids[b].data_ptr(), ds[b].data_ptr(),
                                               cnt[b].data_ptr(), 0, 0, s))
        if b >= D - 1:
            gix.search_end_device(tickets[b - D + 1], s)
    with pytest.raises(ga.GranneHipError):
        gix.search_end_device(tickets[0], s)  # ended already
    for t in tickets[nb - D + 1:]:
        gix.search_end_device(t, s)
    got = ids.clone()  # on the caller's stream: ordered after every batch by its end
    torch.cuda.synchronize()
    assert (got.cpu().numpy().reshape(-1, 10).astype(np.uint64) == oi).all()
    assert ds.cpu().numpy().reshape(-1, 10).tobytes() == od.tobytes()
    assert (cnt.cpu().numpy().reshape(-1).astype(np.uint32) == oc).all()
    for _ in range(D):
        tickets.append(gix.search_begin_device(dq.data_ptr(), nq, 40, 10, ids[0].data_ptr(), ds[0].data_ptr(), cnt[0].data_ptr(), 0, 0, s))
    with pytest.raises(ga.GranneHipError) as e:  # one more than the depth
        gix.search_begin_device(dq.data_ptr(), nq, 40, 10, ids[1].data_ptr(), ds[1].data_ptr(), cnt[1].data_ptr(), 0, 0, s)
    assert e.value.code == _lib.ERR_INVALID
    for t in tickets[-D:]:
        gix.search_end_device(t, s)
    torch.cuda.synchronize()


def test_search_depth_is_a_run_time_option_and_any_free_place_is_taken(ga, oracle):
    """GRANNE_HIP_OPT_SEARCH_DEPTH (1..16, default 3): begin takes ANY free place -- begin 0, 1, 2, end 1, begin again works
    (round 4 went round-robin and refused it) -- the depth cannot change while a batch is in flight, and eight batches in
    flight return what eight plain calls return."""
    import torch
    from granne_amd import _lib
    rng = np.random.default_rng(2810)
    el = prep(oracle, random_floats(rng, 3000, 100), True)
    oix = oracle.build_index(el, num_neighbors=16, max_search=20, reinsert_elements=False, n_threads=0)
    gix = ga.Granne("angular_int", el, oix.layers)
    assert gix.get_option(_lib.OPT_SEARCH_DEPTH) == _lib.SEARCH_DEPTH
    nb, nq = 8, 64
    q = prep(oracle, random_floats(rng, nb * nq, 100), True)
    oi, od, oc, _ = oix.search_batch(q, 30, 10)
    dq = torch.from_numpy(q).cuda()
    ids = torch.zeros((nb, nq, 10), dtype=torch.int64, device="cuda")
    ds = torch.zeros((nb, nq, 10), dtype=torch.float32, device="cuda")
    cnt = torch.zeros((nb, nq), dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream

    def begin(b):
        return gix.search_begin_device(dq[b * nq:(b + 1) * nq].data_ptr(), nq, 30, 10, ids[b].data_ptr(), ds[b].data_ptr(),
                                       cnt[b].data_ptr(), 0, 0, s)
    t = [begin(0), begin(1), begin(2)]
    gix.search_end_device(t[1], s)
    t.append(begin(3))  # the place of batch 1
    with pytest.raises(ga.GranneHipError):
        gix.set_option(_lib.OPT_SEARCH_DEPTH, 8)  # not while batches are in flight
    for k in (0, 2, 3):
        gix.search_end_device(t[k], s)
    with pytest.raises(ga.GranneHipError):
        gix.set_option(_lib.OPT_SEARCH_DEPTH, 0)
    with pytest.raises(ga.GranneHipError):
        gix.set_option(_lib.OPT_SEARCH_DEPTH, _lib.SEARCH_DEPTH_MAX + 1)
    gix.set_option(_lib.OPT_SEARCH_DEPTH, 8)
    assert gix.get_option(_lib.OPT_SEARCH_DEPTH) == 8
    ids.zero_()
    tickets = [begin(b) for b in range(nb)]
    with pytest.raises(ga.GranneHipError):
        begin(0)  # a ninth
    for tk in reversed(tickets):
        gix.search_end_device(tk, s)
    got = ids.clone()
    torch.cuda.synchronize()
    assert (got.cpu().numpy().reshape(-1, 10).astype(np.uint64) == oi).all()
    assert ds.cpu().numpy().reshape(-1, 10).tobytes() == od.tobytes()
    assert (cnt.cpu().numpy().reshape(-1).astype(np.uint32) == oc).all()


def test_many_streams_take_transient_scratch_blocks(ga, oracle):
    """An index caches one scratch block per stream for 64 streams; further streams search with a block of the
    stream-ordered allocator (no device-wide synchronisation, nothing discarded) and return the same results."""
    import torch
    rng = np.random.default_rng(2700)
    el = prep(oracle, random_floats(rng, 3000, 100), False)
    oix = oracle.build_index(el, num_neighbors=12, max_search=20, reinsert_elements=False, n_threads=0)
    gix = ga.Granne("angular", el, oix.layers)
    q = prep(oracle, random_floats(rng, 64, 100), False)
    oi, od, oc, _ = oix.search_batch(q, 40, 10)
    dq = torch.from_numpy(q).cuda()
    streams = [torch.cuda.Stream() for _ in range(70)]
    outs = []
    torch.cuda.synchronize()
    for rep in range(2):
        for s_ in streams:
            ids = torch.empty((64, 10), dtype=torch.int64, device="cuda")
            ds = torch.empty((64, 10), dtype=torch.float32, device="cuda")
            cnt = torch.empty(64, dtype=torch.int32, device="cuda")
            gix.search_batch_device(dq.data_ptr(), 64, 40, 10, ids.data_ptr(), ds.data_ptr(), cnt.data_ptr(), 0, 0, s_.cuda_stream)
            outs.append((ids, ds, cnt))
    torch.cuda.synchronize()
    for ids, ds, cnt in outs:
        assert (ids.cpu().numpy().astype(np.uint64) == oi).all() and ds.cpu().numpy().tobytes() == od.tobytes()
        assert (cnt.cpu().numpy().astype(np.uint32) == oc).all()


def test_concurrent_searches_on_a_shared_index(ga, oracle):
    """Granne::search takes &self and is re-entrant; granne_hip_search_batch is
    documented thread-safe on a shared handle: four host threads, different batches."""
    import threading
    rng = np.random.default_rng(19)
    el = prep(oracle, random_floats(rng, 6000, 100), False)
    oix = oracle.build_index(el, num_neighbors=20, max_search=20, reinsert_elements=False, n_threads=0)
    gix = ga.Granne("angular", el, oix.layers)
    qs = [prep(oracle, random_floats(rng, 300, 100), False) for _ in range(4)]
    want = [oix.search_batch(q, 40, 10) for q in qs]
    got = [None] * 4

    def work(i):
        for _ in range(5):
            got[i] = gix.search_batch(qs[i], 40, 10)

    th = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    [t.start() for t in th]
    [t.join() for t in th]
    for i in range(4):
        assert (got[i][0] == want[i][0]).all() and got[i][1].tobytes() == want[i][1].tobytes()


def test_timed_search_records_events_around_the_kernel(ga, oracle):
    """granne_hip_search_batch_device_timed: same results, and the two events bracket the search kernel."""
    import ctypes as C
    import torch
    from granne_amd import _lib
    rng = np.random.default_rng(77)
    el = prep(oracle, random_floats(rng, 3000, 100), False)
    oix = oracle.build_index(el, num_neighbors=20, max_search=20, n_threads=4)
    gix = ga.Granne("angular", el, oix.layers)
    q = prep(oracle, random_floats(rng, 128, 100), False)
    tq = torch.from_numpy(q).cuda()
    ids = torch.empty((128, 10), dtype=torch.int64, device="cuda")
    ds = torch.empty((128, 10), dtype=torch.float32, device="cuda")
    cnt = torch.empty(128, dtype=torch.int32, device="cuda")
    lib = _lib.lib()
    e0, e1 = C.c_void_p(), C.c_void_p()
    _lib.check(lib.granne_hip_event_create(C.byref(e0)))
    _lib.check(lib.granne_hip_event_create(C.byref(e1)))
    gix.search_batch_device_timed(tq.data_ptr(), 128, 50, 10, ids.data_ptr(), ds.data_ptr(), cnt.data_ptr(), 0, 0,
                                  torch.cuda.current_stream().cuda_stream, e0.value, e1.value)
    torch.cuda.synchronize()
    ms = C.c_float(-1.0)
    _lib.check(lib.granne_hip_event_elapsed_ms(e0, e1, C.byref(ms)))
    assert 0.0 < ms.value < 1000.0
    lib.granne_hip_event_destroy(e0)
    lib.granne_hip_event_destroy(e1)
    want = oix.search_batch(q, 50, 10)
    assert (ids.cpu().numpy().astype(np.uint64) == want[0]).all() and ds.cpu().numpy().tobytes() == want[1].tobytes()


def test_num_neighbors_zero_is_an_empty_result(ga, oracle):
    """`.take(0)` (src/index/mod.rs:974-977): no results, no error."""
    rng = np.random.default_rng(31)
    el = prep(oracle, random_floats(rng, 300, 100), False)
    oix = oracle.build_index(el, num_neighbors=10, max_search=20, n_threads=2)
    gix = ga.Granne("angular", el, oix.layers)
    ids, ds, cnt = gix.search_batch(el[:5], 20, 0)
    assert ids.shape == (5, 0) and ds.shape == (5, 0) and (cnt == 0).all()
    assert gix.search(el[0], 20, 0) == []


def test_neighbor_ids_outside_their_layer_are_rejected(ga, oracle):
    from granne_amd import _lib
    rng = np.random.default_rng(32)
    el = prep(oracle, random_floats(rng, 300, 100), False)
    oix = oracle.build_index(el, num_neighbors=10, max_search=20, n_threads=2)
    layers = [l.copy() for l in oix.layers]
    layers[0][0, 0] = layers[0].shape[0]  # first id past the top layer
    with pytest.raises(_lib.GranneHipError) as e:
        ga.Granne("angular", el, layers)
    assert e.value.code == _lib.ERR_INVALID


# ---- round 6: rows on lines, the neighbors' tails next to the ids (LayerDev::adjx) --------------------------------------
@pytest.mark.parametrize("dim", [100, 200])
@pytest.mark.parametrize("max_search", [1, 50, 200, 600, 1500, 3000])
def test_inline_tails_on_and_off_are_the_same_walk(ga, oracle, dim, max_search):
    """GRANNE_HIP_OPT_INLINE_TAILS: 100-d / 200-d f32 indexes keep a copy of every layer in which a node's ids are followed
    by the tails (the dim % 32 last floats, added after the ordered sum: src/math.rs:32-39) of its neighbors' rows. On (the
    default) or off, every list length of the register walker returns the oracle's ids, distance bits and counters; rows
    shorter than 32 ids (and empty ones) read no tail of their own; the option can be flipped on a live index."""
    if os.environ.get("GRANNE_HIP_INLINE_TAILS") is not None:
        pytest.skip("GRANNE_HIP_INLINE_TAILS is set: the experiment knob overrides the option this test switches")
    from granne_amd import _lib
    rng = np.random.default_rng(dim * 7 + max_search)
    n, nq = 6000, 96
    el = prep(oracle, random_floats(rng, n, dim), False)
    el[17] = el[4000]  # exact distance ties
    el[18] = 0         # a zero vector: never indexed, its row stays empty (src/index/mod.rs:813-815)
    q = prep(oracle, random_floats(rng, nq, dim), False)
    oix = oracle.build_index(el, num_neighbors=30, max_search=40, reinsert_elements=True, n_threads=0)
    gix = ga.Granne("angular", el, oix.layers)
    assert gix.get_option(_lib.OPT_INLINE_TAILS) == 1
    with_tails = gix.hbm_bytes()
    a = assert_same(oix, gix, q, max_search, 10)
    gix.set_option(_lib.OPT_INLINE_TAILS, 0)
    assert gix.get_option(_lib.OPT_INLINE_TAILS) == 0
    per_node = 128 + 32 * 4 * (dim % 32)
    assert with_tails - gix.hbm_bytes() == per_node * sum(len(l) for l in oix.layers)
    b = assert_same(oix, gix, q, max_search, 10)
    gix.set_option(_lib.OPT_INLINE_TAILS, 1)
    c = assert_same(oix, gix, q, max_search, 10)
    for x, y in ((a, b), (a, c)):
        assert (x[0] == y[0]).all() and x[1].tobytes() == y[1].tobytes()
    assert gix.hbm_bytes() == with_tails
    # one query per call (the rows-touched-ahead form of the walker touches the record's tail lines too)
    for i in range(4):
        res = gix.search(q[i], max_search, 10)
        oi, od, oc, _ = oix.search_batch(q[i:i + 1], max_search, 10)
        assert [r[0] for r in res] == oi[0, :int(oc[0])].tolist()


def test_inline_tails_follow_reorder_and_builder_output(ga, oracle):
    """The copy is made whenever an index's layers are final: from host layers, from a GPU builder's get_index, after
    Granne::reorder (ids and elements have moved: the tails must be the NEW neighbors')."""
    if os.environ.get("GRANNE_HIP_INLINE_TAILS") is not None:
        pytest.skip("GRANNE_HIP_INLINE_TAILS is set: the experiment knob overrides the option this test reads")
    from granne_amd import _lib
    rng = np.random.default_rng(99)
    n, dim, nq = 5000, 100, 64
    el = prep(oracle, random_floats(rng, n, dim), False)
    q = prep(oracle, random_floats(rng, nq, dim), False)
    b = ga.GranneBuilder("angular", el, num_neighbors=30, max_search=40, reinsert_elements=True, batch_max=256, batch_div=8)
    b.build()
    gix = b.get_index()
    assert gix.get_option(_lib.OPT_INLINE_TAILS) == 1
    oix = oracle.Index(el, b.layers())
    before = assert_same(oix, gix, q, 50, 10)
    order = gix.reorder()
    assert gix.get_option(_lib.OPT_INLINE_TAILS) == 1
    after = gix.search_batch(q, 50, 10)
    assert after[1].tobytes() == before[1].tobytes()
    distinct = (np.diff(before[1], axis=1) > 0).all(axis=1)
    assert np.array_equal(order[after[0][distinct].astype(np.int64)], before[0][distinct])
    # other shapes have no such copy: int8 rows are one line already, streamed dims read their tails from the rows
    for et, d in (("angular_int", 100), ("angular", 96), ("angular", 300)):
        e2 = prep(oracle, random_floats(rng, 500, d), et == "angular_int")
        o2 = oracle.build_index(e2, num_neighbors=10, max_search=20, n_threads=0)
        g2 = ga.Granne(et, e2, o2.layers)
        assert g2.get_option(_lib.OPT_INLINE_TAILS) == 0
        assert_same(o2, g2, prep(oracle, random_floats(rng, 16, d), et == "angular_int"), 30, 5)


@pytest.mark.parametrize("dim", [64, 65, 100, 127, 160, 200, 300])
def test_f32_rows_start_on_lines_and_every_operator_reads_them(ga, oracle, dim):
    """f32 rows of 256 bytes and more start on a 128-byte line on the device (row stride != row bytes for most dims): search,
    dists, get_element, the exact scan and the GPU builder all read the same rows."""
    rng = np.random.default_rng(dim)
    n = 3000
    el = prep(oracle, random_floats(rng, n, dim), False)
    q = prep(oracle, random_floats(rng, 32, dim), False)
    b = ga.GranneBuilder("angular", el, num_neighbors=20, max_search=30, reinsert_elements=False, batch_max=256, batch_div=8)
    b.build()
    gix = b.get_index()
    oix = oracle.Index(el, b.layers())
    ob = oracle.build_index(el, num_neighbors=20, max_search=30, reinsert_elements=False, batch_max=256, batch_div=8, n_threads=0)
    for lg, lo in zip(b.layers(), ob.layers):
        assert np.array_equal(lg, lo)  # the GPU build is the oracle's batched build
    assert_same(oix, gix, q, 40, 10)
    for i in (0, 1, n - 1):
        assert gix.get_element(i).tobytes() == el[i].tobytes()
    ids = rng.integers(0, n, (32, 9)).astype(np.uint32)
    want = np.array([[oracle.dist(el[e], q[a]) for e in ids[a]] for a in range(32)], np.float32)
    assert gix.dists_many(q, ids).tobytes() == want.tobytes()
    if dim <= 256:
        bi, bd, bc = gix.brute_force(q, 10)
        d_all = np.array([[oracle.dist(el[e], q[a]) for e in range(n)] for a in range(4)], np.float32)
        for a in range(4):
            order = np.lexsort((np.arange(n), d_all[a]))[:10]
            assert bd[a].tobytes() == d_all[a][order].tobytes()


def test_index_bytes_are_the_index_file(ga, oracle, tmp_path):
    """granne_hip_index_encode (Index::write_index into a writer, src/index/mod.rs:67-70) = the bytes save_index writes."""
    rng = np.random.default_rng(5)
    el = prep(oracle, random_floats(rng, 2000, 40), False)
    oix = oracle.build_index(el, num_neighbors=12, max_search=20, n_threads=0)
    gix = ga.Granne("angular", el, oix.layers)
    p = tmp_path / "ix.granne"
    gix.save_index(str(p))
    blob = gix.index_bytes()
    assert blob == p.read_bytes() and blob[:6] == b"granne" and len(blob) > 1024


@pytest.mark.parametrize("int8", [False, True])
def test_two_level_lists_on_ties_twins_and_short_graphs(ga, oracle, int8):
    """The two-level list of max_search beyond 1024 (walk_fast.h, search_layer_long; tools/model_twolevel.py): its flush,
    its theta by a split of two sorted arrays, its tie path -- on data made of ties (every vector five times: distances
    repeat, ids decide), on rows that name a neighbor twice, on a graph shorter than the list (everything is expanded, the
    queue runs empty), with every max_search class (33 / 65 / 129 windows) and k up to max_search."""
    rng = np.random.default_rng(4242 + int8)
    base = prep(oracle, random_floats(rng, 1800, 100), int8)
    el = np.concatenate([base] * 5)[rng.permutation(9000)]
    oix = oracle.build_index(el, num_neighbors=30, max_search=40, n_threads=4)
    layers = [l.copy() for l in oix.layers]
    bottom = layers[-1]
    for i in range(0, len(bottom), 7):  # rows that name their first neighbor twice
        row = bottom[i]
        nv = int((row != 0xFFFFFFFF).sum())
        if nv >= 3:
            row[nv - 1] = row[0]
    oix2 = oracle.Index(el, layers)
    gix = ga.Granne("angular_int" if int8 else "angular", el, layers)
    q = np.concatenate([prep(oracle, random_floats(rng, 12, 100), int8), el[:4]])
    for ms in (1100, 2048, 2500, 4096, 5000):
        assert_same(oix2, gix, q, ms, 10, check_stats=True)
    assert_same(oix2, gix, q[:6], 3000, 3000)
    # a graph shorter than the list: 700 points, max_search 1500
    small = prep(oracle, random_floats(rng, 700, 100), int8)
    o3 = oracle.build_index(small, num_neighbors=30, max_search=40, n_threads=4)
    g3 = ga.Granne("angular_int" if int8 else "angular", small, o3.layers)
    assert_same(o3, g3, q[:12], 1500, 700)
    assert_same(o3, g3, q[:12], 8000, 10)


@pytest.mark.parametrize("dim", [100, 200, 96, 300])
def test_revisits_skipped_before_their_rows_are_fetched(ga, oracle, dim):
    """GRANNE_HIP_OPT_SEEN_MIN (walk_fast.h, SEEN): launches of many f32 walks consult a cache of the ids a walk has
    evaluated BEFORE they fetch a neighbor's row and skip a hit -- the reference's `!visited.insert(n)` (mod.rs:1026) for the
    recent part of the visited set; a miss means nothing. Forced on for every launch here (the default asks for 2048 walks):
    ids, distance bits, expansions and adjacency counts are the oracle's on clustered data (most neighbors are revisits), on
    rows that name a neighbor twice, with and without the walkers' copy of the layers (100-d / 200-d: the unrolled walkers; 96-d
    / 300-d: the streamed one, whose revisits follow the first new neighbor's row); the evaluations counted lie between
    the oracle's distinct nodes and what the walker without the cache evaluates."""
    if os.environ.get("GRANNE_HIP_SEEN_MIN") is not None:
        pytest.skip("GRANNE_HIP_SEEN_MIN is set: the experiment knob overrides the option this test switches")
    from granne_amd import _lib
    rng = np.random.default_rng(900 + dim)
    centers = random_floats(rng, 40, dim)
    raw = centers[rng.integers(0, 40, 6000)] + 0.05 * random_floats(rng, 6000, dim)
    el = prep(oracle, raw.astype(np.float32), False)
    q = prep(oracle, (centers[rng.integers(0, 40, 200)] + 0.05 * random_floats(rng, 200, dim)).astype(np.float32), False)
    oix = oracle.build_index(el, num_neighbors=30, max_search=40, n_threads=4)
    layers = [l.copy() for l in oix.layers]
    for i in range(0, len(layers[-1]), 9):  # rows that name their first neighbor twice
        row = layers[-1][i]
        nv = int((row != 0xFFFFFFFF).sum())
        if nv >= 3:
            row[nv - 1] = row[0]
    oix = oracle.Index(el, layers)
    gix = ga.Granne("angular", el, layers)
    assert gix.get_option(_lib.OPT_SEEN_MIN) == 2048
    for tails in (1, 0):
        gix.set_option(_lib.OPT_INLINE_TAILS, tails)
        for ms in (1, 30, 60, 61, 124, 200, 252, 300):  # (one, two and four slots skip revisits; longer lists walk as before)
            gix.set_option(_lib.OPT_SEEN_MIN, 0xFFFFFFFF)
            ids0, ds0, cnt0, st0 = gix.search_batch(q, ms, 10, stats=True)
            gix.set_option(_lib.OPT_SEEN_MIN, 0)
            ids, ds, cnt, st = gix.search_batch(q, ms, 10, stats=True)
            oi, od, oc, octr = oix.search_batch(q, ms, 10)
            assert (cnt == oc).all() and (ids == oi).all() and ds.tobytes() == od.tobytes(), (tails, ms)
            assert (ids0 == oi).all() and ds0.tobytes() == od.tobytes()
            assert_counters(st, octr, exact=False)
            assert (st[:, 0] <= st0[:, 0]).all()
            if 1 < ms <= 252:
                assert st[:, 0].sum() < 0.9 * st0[:, 0].sum(), (ms, st[:, 0].sum(), st0[:, 0].sum())  # the cache does skip rows here
    gix.set_option(_lib.OPT_SEEN_MIN, 2048)

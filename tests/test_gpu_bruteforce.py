"""granne_hip_brute_force_device: the exact k nearest elements by a scan of all elements on the matrix cores
(granne_amd/csrc/brute_force.h; ElementContainer::dists for every index, src/elements/mod.rs:35-39).
Tolerance mode, as its header says: the returned DISTANCES are the reference's, bit for bit, for the returned ids
(dists_kernel recomputes them); the id set may differ from a scalar scan only between elements whose distances to the
query lie within the MFMA's rounding of each other at the k-th place. The tests hold it to that."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests.conftest import random_floats  # noqa: E402

TOL = 2e-6  # int8 rows (exact integer dots; an f32 dot over <= 256 terms of magnitude <= 1): |distance(returned j-th) - distance(true j-th)|
TOL_F32 = 4e-5  # f32 rows are scored as two bf16 pieces each (three matrix instructions per product, brute_force.h): what is dropped is below 2^-15 of |x||q| per product


def exact_topk(oracle, el, q, k):
    """the reference's distances for every (query, element) pair, ranked by (dist, id)"""
    out_i, out_d = [], []
    for qi in range(len(q)):
        d = np.array([oracle.dist(el[e], q[qi]) for e in range(len(el))], np.float32)
        order = np.lexsort((np.arange(len(el)), d))[:k]
        out_i.append(order)
        out_d.append(d[order])
    return np.array(out_i), np.array(out_d)


@pytest.mark.parametrize("int8,dim,n", [(False, 100, 5000), (False, 200, 3000), (False, 32, 700), (False, 256, 900), (False, 3, 300),
                                        (False, 300, 1300), (False, 768, 900), (False, 1000, 333),  # rows beyond 256 dims: the vector in chunks
                                        (True, 200, 2000), (True, 300, 1500), (True, 1000, 500), (True, 129, 700),  # int8 rows beyond 128 bytes
                                        (True, 100, 5000), (True, 128, 1500), (True, 17, 400)])
def test_brute_force_matches_a_scalar_scan(oracle, int8, dim, n):
    import granne_amd
    rng = np.random.default_rng(dim + n)
    raw = random_floats(rng, n, dim)
    el = oracle.quantize(raw) if int8 else oracle.normalize_f32(raw)
    rq = random_floats(rng, 70, dim)
    q = oracle.quantize(rq) if int8 else oracle.normalize_f32(rq)
    q[3] = el[11]  # a member as a query: distance 0 (or the clamp)
    ix = granne_amd.Granne("angular_int" if int8 else "angular", el, [])
    for k in (1, 10, 16):
        ids, ds, cnt = ix.brute_force(q, k)
        want_i, want_d = exact_topk(oracle, el, q, k)
        assert (cnt == min(k, n)).all()
        # distances are the reference's for the returned ids, ascending by (dist, id)
        for qi in range(len(q)):
            got = [oracle.dist(el[int(e)], q[qi]) for e in ids[qi]]
            assert np.array(got, np.float32).tobytes() == ds[qi].tobytes()
            keys = list(zip(ds[qi].tolist(), ids[qi].tolist()))
            assert keys == sorted(keys) and len(set(ids[qi].tolist())) == k
        # and they are the k smallest up to the MFMA's rounding
        tol = TOL if int8 else TOL_F32
        assert np.abs(ds - want_d).max() <= tol
        same = (ids == want_i.astype(np.uint64))
        assert same.mean() > 0.98  # ties / near-ties only
        for qi, j in zip(*np.nonzero(~same)):
            assert abs(float(ds[qi, j]) - float(want_d[qi, j])) <= tol


@pytest.mark.parametrize("int8,dim,n,nq", [(True, 100, 300_000, 300), (False, 100, 300_000, 70), (False, 200, 150_000, 40),
                                           (True, 64, 1_000_000, 520),
                                           # 128-byte int8 rows (the LDS-DMA ring, brute_force.h bf_i8_ring_kernel): four query tiles
                                           # of 512 and 64 ranges; one tile and 128 ranges merged in two steps; a last tile of 33 rows
                                           (True, 128, 70_049, 1600), (True, 100, 150_000, 513),
                                           (False, 384, 60_000, 300),   # f32 rows beyond 256 dims, with the priming pass
                                           (True, 256, 100_000, 300)])  # int8 rows beyond 128 bytes, priming + shared threshold
def test_primed_scan_with_the_shared_threshold_matches_the_scalar_scan(oracle, int8, dim, n, nq):
    """Sets large enough for the priming pass and the per-query threshold that the ranges share (brute_force.h, BfShare):
    the result is the oracle's scan whatever order the ranges publish in -- run twice, identical."""
    import granne_amd
    rng = np.random.default_rng(n + dim)
    raw = random_floats(rng, n, dim)
    el = oracle.quantize(raw) if int8 else oracle.normalize_f32(raw)
    rq = random_floats(rng, nq, dim)
    q = oracle.quantize(rq) if int8 else oracle.normalize_f32(rq)
    q[5] = el[n - 3]
    q[6] = 0  # a zero query: every distance the clamp's
    ix = granne_amd.Granne("angular_int" if int8 else "angular", el, [])
    oix = oracle.Index(el, [])
    for k in (10, 16, 1):
        ids, ds, cnt = ix.brute_force(q, k)
        ids2, ds2, cnt2 = ix.brute_force(q, k)
        assert (ids == ids2).all() and ds.tobytes() == ds2.tobytes() and (cnt == cnt2).all()
        _, want_i, want_d = oix.scan_topk(q, k)
        assert (cnt == k).all()
        live = np.ones(nq, bool)
        live[6] = False
        tol = TOL if int8 else TOL_F32
        assert np.abs(ds[live] - want_d[live]).max() <= tol
        same = ids[live] == want_i[live]
        assert same.mean() > 0.98
        for qi, j in zip(*np.nonzero(~same)):
            assert abs(float(ds[live][qi, j]) - float(want_d[live][qi, j])) <= tol
        for qi in (0, 5, nq - 1):  # the reference's distances for the returned ids
            got = np.array([oracle.dist(el[int(e)], q[qi]) for e in ids[qi]], np.float32)
            assert got.tobytes() == ds[qi].tobytes()


def test_brute_force_small_and_ragged(oracle):
    import granne_amd
    from granne_amd import GranneHipError
    rng = np.random.default_rng(5)
    el = oracle.normalize_f32(random_floats(rng, 7, 100))
    q = oracle.normalize_f32(random_floats(rng, 3, 100))
    ix = granne_amd.Granne("angular", el, [])
    ids, ds, cnt = ix.brute_force(q, 10)  # fewer elements than k
    assert (cnt == 7).all()
    assert (ids[:, 7:] == np.iinfo(np.uint64).max).all() and np.isinf(ds[:, 7:]).all()
    want_i, want_d = exact_topk(oracle, el, q, 7)
    assert (ids[:, :7] == want_i.astype(np.uint64)).all() and ds[:, :7].tobytes() == want_d.tobytes()
    with pytest.raises(GranneHipError):
        ix.brute_force(q, 17)
    # int8 rows of 128 bytes, fewer than one tile of them, a zero row among them (its distance is the clamp's 1)
    el8 = oracle.quantize(random_floats(rng, 9, 100))
    el8[4] = 0
    q8 = oracle.quantize(random_floats(rng, 5, 100))
    ix8 = granne_amd.Granne("angular_int", el8, [])
    ids, ds, cnt = ix8.brute_force(q8, 16)
    assert (cnt == 9).all() and (ids[:, 9:] == np.iinfo(np.uint64).max).all() and np.isinf(ds[:, 9:]).all()
    want_i, want_d = exact_topk(oracle, el8, q8, 9)
    assert ds[:, :9].tobytes() == want_d.tobytes()
    for qi in range(5):
        assert sorted(ids[qi, :9].tolist()) == list(range(9))
        keys = list(zip(ds[qi, :9].tolist(), ids[qi, :9].tolist()))
        assert keys == sorted(keys)
    with pytest.raises(GranneHipError):
        ix8.brute_force(q8, 0)


def test_brute_force_is_the_recall_ground_truth_of_a_walk(oracle):
    """the walk's results are a subset-quality approximation of the scan's: recall@10 of max_search 200 on 4000 points"""
    import granne_amd
    rng = np.random.default_rng(9)
    el = oracle.normalize_f32(random_floats(rng, 4000, 32))
    q = oracle.normalize_f32(random_floats(rng, 64, 32))
    oix = oracle.build_index(el, num_neighbors=20, max_search=50, n_threads=4)
    ix = granne_amd.Granne("angular", el, oix.layers)
    truth, _, _ = ix.brute_force(q, 10)
    got, _, _ = ix.search_batch(q, 200, 10)
    recall = np.mean([len(set(truth[i]) & set(got[i])) / 10 for i in range(len(q))])
    assert recall > 0.9

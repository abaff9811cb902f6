"""Partitioned mode on the GPU (one device stands in for the ranks: several shards per device):
per-shard HIP searches into packed buffers, the HIP merge kernel, checked against per-shard oracle
searches + the numpy merge -- through granne_amd.sharded (the one-process-per-GPU path, world 1)
and through granne_hip_sharded_* (the one-host-process path of include/granne_hip.h)."""
import ctypes as C

import numpy as np
import pytest

from oracle.merge import merge_topk_numpy

pytestmark = pytest.mark.gpu

from tests.conftest import random_floats  # noqa: E402


def _shards(oracle, int8, n_shards, seed):
    import granne_amd
    from granne_amd import sharded
    rng = np.random.default_rng(seed)
    raw = random_floats(rng, 4000, 32)
    el = oracle.quantize(raw) if int8 else oracle.normalize_f32(raw)
    q = oracle.quantize(random_floats(rng, 50, 32)) if int8 else oracle.normalize_f32(random_floats(rng, 50, 32))
    bounds = sharded.shard_bounds(len(el), n_shards)
    gixs, oixs = [], []
    for lo, hi in bounds:
        part = np.ascontiguousarray(el[lo:hi])
        oix = oracle.build_index(part, num_neighbors=10, max_search=20, n_threads=0)
        gixs.append(granne_amd.Granne("angular_int" if int8 else "angular", part, oix.layers))
        oixs.append(oix)
    return q, bounds, gixs, oixs


def _want(oixs, q, ef, k, offsets):
    res = [o.search_batch(q, ef, k) for o in oixs]
    return merge_topk_numpy(np.stack([r[0] for r in res]), np.stack([r[1] for r in res]), np.stack([r[2] for r in res]),
                            offsets, k)


@pytest.mark.parametrize("int8", [False, True])
@pytest.mark.parametrize("shards,k", [(2, 10), (8, 10), (3, 1), (8, 64)])
def test_sharded_search_and_merge(oracle, int8, shards, k):
    import torch
    from granne_amd import sharded
    q, bounds, gixs, oixs = _shards(oracle, int8, shards, shards * 100 + k)
    offsets = [b[0] for b in bounds]
    sg = sharded.ShardedGranne(gixs, offsets)  # world 1: all shards are local
    m_ids, m_ds, m_cnt = sg.search_batch(q, 70, k, timed=True)
    torch.cuda.synchronize()
    want = _want(oixs, q, 70, k, offsets)
    assert (m_cnt.cpu().numpy().astype(np.uint32) == want[2]).all()
    assert (m_ids.cpu().numpy().astype(np.uint64) == want[0]).all()
    assert m_ds.cpu().numpy().tobytes() == want[1].tobytes()
    assert set(sg.timings) == {"search_ms", "exchange_ms", "merge_ms"}


@pytest.mark.parametrize("int8", [False, True])
def test_sharded_c_abi(oracle, int8):
    """granne_hip_sharded_create / _search_batch / _search: what a Rust host calls."""
    from granne_amd import _lib
    lib = _lib.lib()
    shards, k, ef = 5, 7, 40
    q, bounds, gixs, oixs = _shards(oracle, int8, shards, 77)
    offsets = [b[0] for b in bounds]
    handles = (C.c_void_p * shards)(*[g._h for g in gixs])
    offs = (C.c_uint64 * shards)(*offsets)
    sh = C.c_void_p()
    _lib.check(lib.granne_hip_sharded_create(C.byref(sh), handles, offs, shards))
    try:
        assert lib.granne_hip_sharded_num_shards(sh) == shards
        assert lib.granne_hip_sharded_len(sh) == sum(len(g) for g in gixs)
        nq = len(q)
        ids = np.empty((nq, k), np.uint64)
        ds = np.empty((nq, k), np.float32)
        cnt = np.zeros(nq, np.uint32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        for _ in range(2):  # the handle's buffers are reused
            _lib.check(lib.granne_hip_sharded_search_batch(sh, p(q), nq, ef, k, p(ids), p(ds), p(cnt)))
            want = _want(oixs, q, ef, k, offsets)
            assert (cnt == want[2]).all() and (ids == want[0]).all() and ds.tobytes() == want[1].tobytes()
        one = C.c_uint32()
        _lib.check(lib.granne_hip_sharded_search(sh, p(q[3:4]), ef, k, p(ids), p(ds), C.byref(one)))
        assert one.value == want[2][3] and (ids[0] == want[0][3]).all()
        # num_neighbors == 0: empty results; max_search == 0: the reference panics -> error code
        _lib.check(lib.granne_hip_sharded_search_batch(sh, p(q), nq, ef, 0, p(ids), p(ds), p(cnt)))
        assert (cnt == 0).all()
        assert lib.granne_hip_sharded_search_batch(sh, p(q), nq, 0, k, p(ids), p(ds), p(cnt)) == _lib.ERR_INVALID
    finally:
        lib.granne_hip_sharded_destroy(sh)

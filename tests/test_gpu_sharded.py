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


def _batches(oracle, int8, n_batches, nq, seed):
    rng = np.random.default_rng(seed)
    raw = random_floats(rng, n_batches * nq, 32)
    q = oracle.quantize(raw) if int8 else oracle.normalize_f32(raw)
    return q.reshape(n_batches, nq, 32)


@pytest.mark.parametrize("int8", [False, True])
@pytest.mark.parametrize("exchange", ["peer", "rccl"])
def test_sharded_device_entries_pipelined(oracle, int8, exchange):
    """The stream-ordered entries of the one-process handle: search_batch_device, begin / end with two batches in flight,
    the pipelined host-pointer search_batches -- with peer copies and with the RCCL all-gather as the exchange step (one
    device here: a one-rank communicator; librccl comes in by dlopen) -- against per-shard oracle searches + numpy merge."""
    import torch
    from granne_amd import _lib, sharded
    shards, k, ef, nb, nq = 5, 6, 40, 6, 37
    _, bounds, gixs, oixs = _shards(oracle, int8, shards, 91)
    offsets = [b[0] for b in bounds]
    sh = sharded.ShardedHost(gixs, offsets)
    if exchange == "rccl":
        sh.set_option(_lib.SHARDED_OPT_EXCHANGE, _lib.SHARDED_EXCHANGE_RCCL)
        assert sh.get_option(_lib.SHARDED_OPT_EXCHANGE) == _lib.SHARDED_EXCHANGE_RCCL
    assert sh.get_option(_lib.SHARDED_OPT_DEPTH) == 2 and len(sh) == sum(len(g) for g in gixs)
    q = _batches(oracle, int8, nb, nq, 5)
    want = [_want(oixs, q[b], ef, k, offsets) for b in range(nb)]
    dq = torch.from_numpy(q).cuda()
    ids = torch.zeros((nb, nq, k), dtype=torch.int64, device="cuda")
    ds = torch.zeros((nb, nq, k), dtype=torch.float32, device="cuda")
    cnt = torch.zeros((nb, nq), dtype=torch.int32, device="cuda")
    status = torch.zeros(4, dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream

    def check():
        torch.cuda.synchronize()
        for b in range(nb):
            assert (cnt[b].cpu().numpy().astype(np.uint32) == want[b][2]).all()
            assert (ids[b].cpu().numpy().astype(np.uint64) == want[b][0]).all()
            assert ds[b].cpu().numpy().tobytes() == want[b][1].tobytes()
        ids.zero_(), ds.zero_(), cnt.zero_()

    for b in range(nb):  # one batch at a time, stream-ordered, no host synchronisation in between
        sh.search_batch_device(dq[b].data_ptr(), nq, ef, k, ids[b].data_ptr(), ds[b].data_ptr(), cnt[b].data_ptr(),
                               status.data_ptr(), s)
    check()
    assert status.tolist() == [0, 0, 0, 0]
    tickets = []
    for b in range(nb):  # two in flight: batch b is begun before batch b - 1 is ended
        tickets.append(sh.begin_device(dq[b].data_ptr(), nq, ef, k, ids[b].data_ptr(), ds[b].data_ptr(), cnt[b].data_ptr(), 0, s))
        if b >= 1:
            sh.end_device(tickets[b - 1], s)
    sh.end_device(tickets[-1], s)
    check()
    h_ids, h_ds, h_cnt = sh.search_batches(q, ef, k)  # host buffers, pipelined inside the library
    for b in range(nb):
        assert (h_cnt[b] == want[b][2]).all() and (h_ids[b] == want[b][0]).all() and h_ds[b].tobytes() == want[b][1].tobytes()
    # another batch size on the same handle regrows the slots' buffers
    q2 = _batches(oracle, int8, 3, 9, 6)
    h2 = sh.search_batches(q2, ef, k)
    for b in range(3):
        w = _want(oixs, q2[b], ef, k, offsets)
        assert (h2[0][b] == w[0]).all() and h2[1][b].tobytes() == w[1].tobytes()
    sh.close()


def test_sharded_depth_and_tickets(oracle):
    import torch
    from granne_amd import _lib, sharded
    _, bounds, gixs, oixs = _shards(oracle, False, 3, 92)
    sh = sharded.ShardedHost(gixs, [b[0] for b in bounds], depth=3)
    assert sh.get_option(_lib.SHARDED_OPT_DEPTH) == 3
    q = torch.from_numpy(_batches(oracle, False, 1, 8, 7)[0]).cuda()
    outs = [(torch.empty((8, 4), dtype=torch.int64, device="cuda"), torch.empty((8, 4), dtype=torch.float32, device="cuda"),
             torch.empty(8, dtype=torch.int32, device="cuda")) for _ in range(4)]
    s = torch.cuda.current_stream().cuda_stream
    t = [sh.begin_device(q.data_ptr(), 8, 30, 4, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), 0, s) for o in outs[:3]]
    with pytest.raises(_lib.GranneHipError) as e:  # a fourth batch with three in flight
        sh.begin_device(q.data_ptr(), 8, 30, 4, outs[3][0].data_ptr(), outs[3][1].data_ptr(), outs[3][2].data_ptr(), 0, s)
    assert e.value.code == _lib.ERR_INVALID
    with pytest.raises(_lib.GranneHipError):
        sh.set_option(_lib.SHARDED_OPT_DEPTH, 2)  # not while batches are in flight
    with pytest.raises(_lib.GranneHipError):
        sh.end_device(t[0] + (1 << 8), s)  # not a ticket of a batch in flight
    for x in t:
        sh.end_device(x, s)
    with pytest.raises(_lib.GranneHipError):
        sh.end_device(t[0], s)  # ended already
    torch.cuda.synchronize()
    want = _want(oixs, q.cpu().numpy(), 30, 4, [b[0] for b in bounds])
    for o in outs[:3]:
        assert (o[0].cpu().numpy().astype(np.uint64) == want[0]).all()
    sh.close()


def test_sharded_status_words_are_folded_on_the_device(oracle):
    """A shard whose exact-search scratch runs out: the device entry reports it in d_status[0] (and counts the hand-overs
    in [1]), the host entry returns GRANNE_HIP_ERR_OVERFLOW -- and the handle stays usable."""
    import torch
    from granne_amd import _lib, sharded
    _, bounds, gixs, oixs = _shards(oracle, False, 4, 93)
    offsets = [b[0] for b in bounds]
    sh = sharded.ShardedHost(gixs, offsets)
    q = _batches(oracle, False, 2, 16, 8)
    gixs[2].set_option(_lib.OPT_FORCE_SLOW, 1)
    dq = torch.from_numpy(q[0]).cuda()
    o = (torch.empty((16, 5), dtype=torch.int64, device="cuda"), torch.empty((16, 5), dtype=torch.float32, device="cuda"),
         torch.empty(16, dtype=torch.int32, device="cuda"))
    status = torch.zeros(4, dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    sh.search_batch_device(dq.data_ptr(), 16, 60, 5, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), status.data_ptr(), s)
    torch.cuda.synchronize()
    assert status.tolist()[:2] == [0, 16]  # one shard served its 16 queries with the exact walker; results unchanged
    want = _want(oixs, q[0], 60, 5, offsets)
    assert (o[0].cpu().numpy().astype(np.uint64) == want[0]).all() and o[1].cpu().numpy().tobytes() == want[1].tobytes()
    gixs[2].set_option(_lib.OPT_SLOW_SLOTS, 256)  # far too few for a walk of max_search 60
    status.zero_()
    sh.search_batch_device(dq.data_ptr(), 16, 60, 5, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), status.data_ptr(), s)
    torch.cuda.synchronize()
    assert status[0].item() == 1
    with pytest.raises(_lib.GranneHipError) as e:
        sh.search_batches(q, 60, 5)
    assert e.value.code == _lib.ERR_OVERFLOW
    gixs[2].set_option(_lib.OPT_FORCE_SLOW, 0)
    h = sh.search_batches(q, 60, 5)
    for b in range(2):
        w = _want(oixs, q[b], 60, 5, offsets)
        assert (h[0][b] == w[0]).all() and h[1][b].tobytes() == w[1].tobytes()
    sh.close()


@pytest.mark.parametrize("int8", [False, True])
def test_sharded_build_from_the_whole_element_set(oracle, int8):
    """granne_hip_sharded_build (index_create with device_ids / n_devices / partitioned): the element set is
    split into id ranges (src/elements/embeddings/parsing.rs:72-98), every shard built by the GPU builder. The shard graphs
    equal the oracle's batched builds of the same ranges; searches equal per-shard oracle searches + the numpy merge."""
    from granne_amd import sharded
    rng = np.random.default_rng(94)
    n, shards, k, ef = 3100, 3, 5, 40  # (3100 = 1034 + 1034 + 1032: the last range is shorter)
    raw = random_floats(rng, n, 32)
    el = oracle.quantize(raw) if int8 else oracle.normalize_f32(raw)
    q = oracle.quantize(random_floats(rng, 30, 32)) if int8 else oracle.normalize_f32(random_floats(rng, 30, 32))
    sh = sharded.ShardedHost.build("angular_int" if int8 else "angular", el, shards, devices=(0,), num_neighbors=12, max_search=30)
    bounds = sharded.shard_bounds(n, shards)
    assert sh.num_shards() == shards and len(sh) == n
    assert [sh.shard_offset(s) for s in range(shards)] == [b[0] for b in bounds]
    oixs = [oracle.build_index(np.ascontiguousarray(el[lo:hi]), num_neighbors=12, max_search=30, n_threads=0, batch_max=65536, batch_div=8)
            for lo, hi in bounds]
    bottom = len(oixs[1].layers) - 1
    got = sh.shard_layer(1, bottom)
    want = oixs[1].layers[bottom]
    assert (got[:, :want.shape[1]] == want).all() and (got[:, want.shape[1]:] == 0xFFFFFFFF).all()
    ids, ds, cnt = sh.search_batch(q, ef, k)
    w = _want(oixs, q, ef, k, [b[0] for b in bounds])
    assert (cnt == w[2]).all() and (ids == w[0]).all() and ds.tobytes() == w[1].tobytes()
    sh.close()


@pytest.mark.parametrize("int8", [False, True])
@pytest.mark.parametrize("groups", [[0, 0, 1, 1], [0, 1, 2, 3], [0, 1, 1, 1, 2], [0, 0, 0, 1, 1, 2, 2, 2]])
def test_exchange_groups_walk_the_multi_device_branches_on_one_gpu(oracle, int8, groups):
    """granne_hip_sharded_create_grouped: every exchange group is what the exchange step treats as a device of its own. With
    several groups on the ONE GPU of this box the branches a multi-GPU node takes all run -- the first shard of a remote
    group fetches the batch's queries for its group (the others wait for them), remote results reach the merge device's
    gather buffer by a copy of their own, a group's buffers are released by its own event -- with two batches in flight,
    uniform and non-uniform layouts; results against per-shard oracle searches + the numpy merge. (The copies are
    device-local here; the all-gather exchange is refused for such a layout: RCCL takes every device once.)"""
    import torch
    from granne_amd import _lib, sharded
    G, k, ef, nb, nq = len(groups), 6, 40, 5, 33
    _, bounds, gixs, oixs = _shards(oracle, int8, G, 95 + G)
    offsets = [b[0] for b in bounds]
    sh = sharded.ShardedHost(gixs, offsets, groups=groups)
    if len(set(groups)) > 1:
        with pytest.raises(_lib.GranneHipError):
            sh.set_option(_lib.SHARDED_OPT_EXCHANGE, _lib.SHARDED_EXCHANGE_RCCL)
    q = _batches(oracle, int8, nb, nq, 9)
    want = [_want(oixs, q[b], ef, k, offsets) for b in range(nb)]
    dq = torch.from_numpy(q).cuda()
    ids = torch.zeros((nb, nq, k), dtype=torch.int64, device="cuda")
    ds = torch.zeros((nb, nq, k), dtype=torch.float32, device="cuda")
    cnt = torch.zeros((nb, nq), dtype=torch.int32, device="cuda")
    status = torch.zeros(4, dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream

    def check():
        torch.cuda.synchronize()
        for b in range(nb):
            assert (cnt[b].cpu().numpy().astype(np.uint32) == want[b][2]).all()
            assert (ids[b].cpu().numpy().astype(np.uint64) == want[b][0]).all()
            assert ds[b].cpu().numpy().tobytes() == want[b][1].tobytes()
        ids.zero_(), ds.zero_(), cnt.zero_()

    for b in range(nb):
        sh.search_batch_device(dq[b].data_ptr(), nq, ef, k, ids[b].data_ptr(), ds[b].data_ptr(), cnt[b].data_ptr(), status.data_ptr(), s)
    check()
    assert status.tolist() == [0, 0, 0, 0]
    tickets = []
    for b in range(nb):  # two in flight: the slots' buffers and events of every group are reused under pipelining
        tickets.append(sh.begin_device(dq[b].data_ptr(), nq, ef, k, ids[b].data_ptr(), ds[b].data_ptr(), cnt[b].data_ptr(), 0, s))
        if b >= 1:
            sh.end_device(tickets[b - 1], s)
    sh.end_device(tickets[-1], s)
    check()
    h_ids, h_ds, h_cnt = sh.search_batches(q, ef, k)
    for b in range(nb):
        assert (h_cnt[b] == want[b][2]).all() and (h_ids[b] == want[b][0]).all() and h_ds[b].tobytes() == want[b][1].tobytes()
    sh.close()


def test_sharded_build_builds_the_groups_at_the_same_time(oracle):
    """granne_hip_sharded_build with several entries in device_ids: one host thread per entry builds that entry's shards
    (on eight GPUs the eight shards of configs[4] build in the time of one); two entries that name the same device are two
    exchange groups on it. The shard graphs equal the oracle's builds of the same ranges whatever the interleaving."""
    from granne_amd import _lib, sharded
    rng = np.random.default_rng(96)
    n, shards, k, ef = 4100, 4, 5, 40
    el = oracle.normalize_f32(random_floats(rng, n, 32))
    q = oracle.normalize_f32(random_floats(rng, 30, 32))
    sh = sharded.ShardedHost.build("angular", el, shards, devices=(0, 0), num_neighbors=12, max_search=30)
    bounds = sharded.shard_bounds(n, shards)
    assert sh.num_shards() == shards and [sh.shard_offset(s) for s in range(shards)] == [b[0] for b in bounds]
    oixs = [oracle.build_index(np.ascontiguousarray(el[lo:hi]), num_neighbors=12, max_search=30, n_threads=0, batch_max=65536, batch_div=8)
            for lo, hi in bounds]
    for s_ in range(shards):
        bottom = len(oixs[s_].layers) - 1
        got, want = sh.shard_layer(s_, bottom), oixs[s_].layers[bottom]
        assert (got[:, :want.shape[1]] == want).all()
    with pytest.raises(_lib.GranneHipError):  # two groups on one device: no all-gather
        sh.set_option(_lib.SHARDED_OPT_EXCHANGE, _lib.SHARDED_EXCHANGE_RCCL)
    ids, ds, cnt = sh.search_batch(q, ef, k)
    w = _want(oixs, q, ef, k, [b[0] for b in bounds])
    assert (cnt == w[2]).all() and (ids == w[0]).all() and ds.tobytes() == w[1].tobytes()
    sh.close()


@pytest.mark.parametrize("shards,k", [(4, 10), (64, 16), (7, 1), (3, 33)])
def test_merge_topk_accepts_lists_in_any_order(shards, k):
    """ADVICE r5 (medium), util_kernels.h merge_topk_kernel: the k-way merge is only right for lists ascending by (dist, id);
    the public entry promises the k best by (dist, global id) of whatever it is given (include/granne_hip.h). Lists out of
    order -- shuffled, ties on distance, ragged counts -- must merge like the numpy reference; sorted ones stay the fast case."""
    import torch
    from granne_amd import _lib
    rng = np.random.default_rng(shards * 1000 + k)
    nq = 37
    ids = rng.integers(0, 5000, (shards, nq, k)).astype(np.uint64)
    ds = (rng.integers(0, 12, (shards, nq, k)) / 16.0).astype(np.float32)  # many ties on distance
    cnt = rng.integers(0, k + 1, (shards, nq)).astype(np.uint32)
    cnt[0, :] = k
    offsets = (np.arange(shards, dtype=np.uint64) * 5000)
    want = merge_topk_numpy(ids, ds, cnt, offsets, k)
    d_ids, d_ds, d_cnt = (torch.from_numpy(ids.view(np.int64)).cuda(), torch.from_numpy(ds).cuda(), torch.from_numpy(cnt.view(np.int32)).cuda())
    o_ids = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    o_ds = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    o_cnt = torch.empty(nq, dtype=torch.int32, device="cuda")
    off = (C.c_uint64 * shards)(*[int(x) for x in offsets])
    _lib.check(_lib.lib().granne_hip_merge_topk_device(
        C.c_void_p(d_ids.data_ptr()), C.c_void_p(d_ds.data_ptr()), C.c_void_p(d_cnt.data_ptr()), off, shards, nq, k,
        C.c_void_p(o_ids.data_ptr()), C.c_void_p(o_ds.data_ptr()), C.c_void_p(o_cnt.data_ptr()), 0,
        C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    assert (o_cnt.cpu().numpy().astype(np.uint32) == want[2]).all()
    got_ids, got_d = o_ids.cpu().numpy().view(np.uint64), o_ds.cpu().numpy()
    for q in range(nq):
        c = int(want[2][q])
        assert got_d[q, :c].tobytes() == want[1][q, :c].tobytes() and (got_ids[q, :c] == want[0][q, :c]).all(), q

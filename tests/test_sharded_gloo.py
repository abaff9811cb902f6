"""The N > 1 path on CPU: two processes, gloo backend, world_size 2.

Partitioned mode: each rank owns id ranges and their indexes (two shards per rank
here); every rank answers the same query batch; ONE all-gather of the packed per-shard top-k (ids,
dists and counts in one buffer); merge by (dist, global id). The local
search and the merge are injected here (the CPU oracle and a numpy merge) because there is no GPU:
what is under test is granne_amd.sharded's exchange logic -- offsets, all-gather layout, ordering,
identical results on every rank -- against a single-process recomputation.
Also covers the replica-mode work split bench.py uses (disjoint query rows per rank).
"""
import os
import socket
import sys

import numpy as np
import pytest

from oracle.merge import merge_topk_numpy, pack_topk, unpack_topk
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOCAL = 2  # shards per rank


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data():
    from oracle import oracle as orc
    rng = np.random.default_rng(123)
    n, dim = 900, 16
    el = orc.normalize_f32((rng.random((n, dim), dtype=np.float32) - np.float32(0.5)))
    q = orc.normalize_f32((rng.random((24, dim), dtype=np.float32) - np.float32(0.5)))
    return el, q


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as orc
    from granne_amd import sharded
    el, q = _data()
    bounds = sharded.shard_bounds(len(el), world * LOCAL)
    mine = bounds[rank * LOCAL:(rank + 1) * LOCAL]
    local = [orc.build_index(np.ascontiguousarray(el[lo:hi]), num_neighbors=8, max_search=20) for lo, hi in mine]
    calls = {"all_gather": 0}
    real = dist.all_gather_into_tensor

    def counting(*a, **kw):
        calls["all_gather"] += 1
        return real(*a, **kw)
    dist.all_gather_into_tensor = counting

    def local_search(queries, max_search, k, out):  # CPU stand-in for the HIP searches: same packed outputs
        for i, ix in enumerate(local):
            ids, ds, cnt, _ = ix.search_batch(np.asarray(queries), max_search, k)
            packed = torch.from_numpy(pack_topk(ids, ds, cnt))
            out[i][:packed.numel()].copy_(packed)  # (the shard's status words follow: zero)

    def merge(gathered, offsets, nq, k):
        parts = [unpack_topk(gathered[g].numpy(), nq, k) for g in range(gathered.shape[0])]
        i, d, c = merge_topk_numpy(np.stack([p[0] for p in parts]), np.stack([p[1] for p in parts]),
                                   np.stack([p[2] for p in parts]), offsets, k)
        return torch.from_numpy(i.astype(np.int64)), torch.from_numpy(d), torch.from_numpy(c.astype(np.int32))

    sg = sharded.ShardedGranne([None] * LOCAL, [b[0] for b in bounds], local_search=local_search, merge=merge)
    ids, ds, cnt = sg.search_batch(q, 20, 5)
    assert calls["all_gather"] == 1, "the exchange is ONE collective per batch"
    assert sg.exchange_bytes_per_rank(len(q), 5) == LOCAL * (((len(q) * 5 * 12 + len(q) * 4 + 15) & ~15) + 16)
    # pipelined: five batches, two in flight; still one collective per batch, same results as one at a time
    batches = [q[:8], q[8:16], q[16:], q[4:12], q]
    calls["all_gather"] = 0
    piped = sg.search_batches(batches, 20, 5, depth=2)
    assert calls["all_gather"] == len(batches)
    for b, (pi, pd, pc) in zip(batches, piped):
        si, sd, sc = sg.search_batch(b, 20, 5)
        assert (pi == si).all() and pd.numpy().tobytes() == sd.numpy().tobytes() and (pc == sc).all()
    # a shard that reports exhausted scratch is seen by EVERY rank (its status words ride in the all-gather)
    if True:
        real_ls = sg._local_search

        def failing(queries, max_search, k, out):
            real_ls(queries, max_search, k, out)
            if rank == 1:
                out[0][sharded.packed_bytes(len(queries), k)] = 1  # shard 2 of the job: status word 0
        sg._local_search = failing
        from granne_amd._lib import GranneHipError
        try:
            sg.search_batch(q, 20, 5)
            raised = False
        except GranneHipError:
            raised = True
        assert raised, "rank %d did not see the other rank's exhausted shard" % rank
        sg._local_search = real_ls
    # a launch that fails in the middle of a pipelined run (on every rank alike): the exception surfaces, no slot stays
    # marked in flight, and the same object answers the next call as if nothing had happened
    if True:
        real_ls = sg._local_search
        n_calls = {"n": 0}

        def flaky(queries, max_search, k, out):
            n_calls["n"] += 1
            if n_calls["n"] == 3:
                raise RuntimeError("launch failed")
            real_ls(queries, max_search, k, out)
        sg._local_search = flaky
        try:
            sg.search_batches(batches, 20, 5, depth=2)
            raised = False
        except RuntimeError:
            raised = True
        assert raised and not any(sl.busy for sl in sg._slots)
        sg._local_search = real_ls
        again = sg.search_batches(batches, 20, 5, depth=2)
        for (pi, pd, pc), (ai, ad, ac) in zip(piped, again):
            assert (pi == ai).all() and pd.numpy().tobytes() == ad.numpy().tobytes() and (pc == ac).all()
        si, sd, sc = sg.search_batch(q, 20, 5)
        assert (si == ids).all()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), ids=ids.numpy(), ds=ds.numpy(), cnt=cnt.numpy(),
             offsets=np.array(sg.offsets))
    # replica mode: disjoint query rows, all rows covered
    r0, per = sharded.replica_query_rows(rank, world, 3, 8)
    t = torch.tensor([r0, per], dtype=torch.int64)
    allv = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(allv, t)
    starts = sorted(int(v[0]) for v in allv)
    assert starts == [g * 24 for g in range(world)] and all(int(v[1]) == 24 for v in allv)
    dist.barrier()
    dist.destroy_process_group()


def test_partitioned_search_two_ranks_gloo(tmp_path, oracle):
    world = 2
    mp.start_processes(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    r = [np.load(os.path.join(tmp_path, "rank%d.npz" % i)) for i in range(world)]
    # identical on every rank
    assert (r[0]["ids"] == r[1]["ids"]).all() and r[0]["ds"].tobytes() == r[1]["ds"].tobytes()
    assert (r[0]["cnt"] == r[1]["cnt"]).all()
    # equals a single-process recomputation: per-shard CPU search + merge
    from granne_amd import sharded
    el, q = _data()
    bounds = sharded.shard_bounds(len(el), world * LOCAL)
    assert r[0]["offsets"].tolist() == [b[0] for b in bounds]
    per_shard = []
    for lo, hi in bounds:
        ix = oracle.build_index(np.ascontiguousarray(el[lo:hi]), num_neighbors=8, max_search=20)
        per_shard.append(ix.search_batch(q, 20, 5))
    ids = np.stack([p[0] for p in per_shard])
    ds = np.stack([p[1] for p in per_shard])
    cnt = np.stack([p[2] for p in per_shard])
    want = merge_topk_numpy(ids, ds, cnt, [b[0] for b in bounds], 5)
    assert (r[0]["ids"].astype(np.uint64) == want[0]).all()
    assert r[0]["ds"].tobytes() == want[1].tobytes()
    # global ids really point at the right elements: distances recomputed from the full set
    for qi in range(len(q)):
        for j in range(int(want[2][qi])):
            assert oracle.dist(el[int(want[0][qi, j])], q[qi]) == float(want[1][qi, j])
    # and the merged list is at least as good as any single shard's
    assert (want[1][:, 0] <= ds[:, :, 0].min(axis=0)).all()


def test_shard_bounds():
    from granne_amd import sharded
    assert sharded.shard_bounds(10, 4) == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert sharded.shard_bounds(100_000_000, 8)[7] == (87_500_000, 100_000_000)
    assert sharded.shard_bounds(3, 8)[5] == (3, 3)


def _wait_worker(rank, world, port, out_dir):
    """bench.py's N > 1 tail: rank 0 works (recall, the CPU baseline) while the others block on the rendezvous store --
    not inside a collective, which would spin on the host cores the baseline is timed on."""
    import time
    import types
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    B = types.SimpleNamespace(rank=rank, world=world)
    t0 = time.time()
    if rank == 0:
        time.sleep(1.5)  # "measuring"
    bench.wait_for_rank0(B, "granne_bench_line_out", timeout_s=60)
    waited = time.time() - t0
    with open(os.path.join(out_dir, "wait%d.txt" % rank), "w") as f:
        f.write("%.3f" % waited)
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_wait_for_rank0_on_the_store(tmp_path):
    world = 2
    mp.start_processes(_wait_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    w = [float(open(os.path.join(tmp_path, "wait%d.txt" % i)).read()) for i in range(world)]
    assert w[0] >= 1.4 and w[1] >= 1.0  # rank 1 left only after rank 0 posted the key

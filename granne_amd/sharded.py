"""Partitioned (sharded) search, one process per GPU.

The element set is split into independent indexes (the reference's own shard helper splits
elements the same way, src/elements/embeddings/parsing.rs:63-100); a rank holds one or more
shards on its GPU. Every rank answers the same query batch on its shards, writing each shard's
top-k into ONE packed buffer ([nq*k u64 local ids][nq*k f32 dists][nq u32 counts]:
granne_hip_packed_topk_bytes); ONE all-gather of those buffers is the exchange step
(124 KB per shard at nq = 1024, k = 10 -- latency-, not bandwidth-bound over xGMI), then the merge
kernel ranks the n_shards*k candidates of each query by (dist, global id). SURVEY.md 8e.

torch.distributed supplies the collective (backend "nccl" = RCCL on ROCm; "gloo" in the CPU
tests). The local search and the merge are the HIP kernels behind the C ABI; they are constructor
parameters only so that tests/test_sharded_gloo.py can drive the exchange logic on a GPU-less box
with stand-ins from oracle/ -- this module itself contains no CPU implementation. A host that
drives several GPUs from one process uses granne_hip_sharded_* (include/granne_hip.h) instead.
"""
import ctypes as C

import numpy as np


def shard_bounds(n_elements, n_shards):
    """Shard g owns ids [g*ceil(n/G), min(n, (g+1)*ceil(n/G)))  (SURVEY.md 8e)."""
    per = -(-n_elements // n_shards)
    return [(min(n_elements, g * per), min(n_elements, (g + 1) * per)) for g in range(n_shards)]


def packed_bytes(nq, k):
    return (nq * k * 12 + nq * 4 + 15) & ~15


class ShardedGranne:
    """`local_indexes`: this rank's granne_amd.Granne objects (local ids), in global shard order
    rank*len(local_indexes) + i. `all_offsets`: the first global id of EVERY shard of the job (known
    to every rank: shard_bounds is deterministic), world*len(local_indexes) entries.
    Collective: every rank must call search_batch with the same queries."""

    def __init__(self, local_indexes, all_offsets, group=None, local_search=None, merge=None):
        import torch.distributed as dist
        self.dist = dist
        self.indexes = list(local_indexes) if isinstance(local_indexes, (list, tuple)) else [local_indexes]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.local = len(self.indexes)
        self.offsets = [int(o) for o in all_offsets]
        if len(self.offsets) != self.world * self.local:
            raise ValueError("need one offset per shard of the job: %d ranks x %d shards" % (self.world, self.local))
        self._on_gpu = local_search is None
        self._local_search = local_search or self._gpu_local_search
        self._merge = merge or self._gpu_merge
        self._streams = None
        self._status = None
        self.timings = None  # set by search_batch(timed=True): HIP-event ms of search / exchange / merge

    # ---- defaults: HIP kernels through the C ABI, tensors on this rank's GPU ---------------------
    def _gpu_local_search(self, queries, max_search, k, out):
        """Every local shard searches the batch on a stream of its own; results land in out[i] (packed)."""
        import torch
        from ._lib import check, lib
        q = queries if torch.is_tensor(queries) else torch.from_numpy(np.ascontiguousarray(queries))
        dev = torch.device("cuda", self.indexes[0].device)
        q = q.to(dev).contiguous()
        nq = q.shape[0]
        if self._streams is None:
            self._streams = [torch.cuda.Stream(device=dev) for _ in self.indexes]
            self._status = torch.zeros((self.local, 4), dtype=torch.int32, device=dev)
        cur = torch.cuda.current_stream(dev)
        self._status.zero_()
        for i, ix in enumerate(self.indexes):
            s = self._streams[i]
            s.wait_stream(cur)  # the queries (and `out`) are ready on the caller's stream
            check(lib().granne_hip_search_batch_packed_device(ix._h, C.c_void_p(q.data_ptr()), nq, int(max_search), int(k),
                                                              C.c_void_p(out[i].data_ptr()),
                                                              C.c_void_p(self._status[i].data_ptr()),
                                                              C.c_void_p(s.cuda_stream)))
        for s in self._streams:
            cur.wait_stream(s)
        return q

    def _gpu_merge(self, gathered, offsets, nq, k):
        import torch
        from ._lib import check, lib
        G = len(offsets)
        dev = gathered.device
        out_ids = torch.empty((nq, k), dtype=torch.int64, device=dev)
        out_d = torch.empty((nq, k), dtype=torch.float32, device=dev)
        out_c = torch.empty(nq, dtype=torch.int32, device=dev)
        off = (C.c_uint64 * G)(*offsets)
        check(lib().granne_hip_merge_topk_packed_device(C.c_void_p(gathered.data_ptr()), off, G, nq, k,
                                                        C.c_void_p(out_ids.data_ptr()), C.c_void_p(out_d.data_ptr()),
                                                        C.c_void_p(out_c.data_ptr()), dev.index or 0,
                                                        C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        return out_ids, out_d, out_c

    # ---- search + the one exchange step + merge ---------------------------------------------------------
    def search_batch(self, queries, max_search, k, check_status=True, timed=False):
        """Returns (ids [nq,k] global, dists [nq,k], counts [nq]) -- identical on every rank."""
        import torch
        nq = int(queries.shape[0])
        pb = packed_bytes(nq, k)
        on_gpu = self._on_gpu
        dev = torch.device("cuda", self.indexes[0].device) if on_gpu else torch.device("cpu")
        mine = torch.empty((self.local, pb), dtype=torch.uint8, device=dev)  # this rank's shards write here
        gathered = torch.empty((self.world, self.local, pb), dtype=torch.uint8, device=dev) if self.world > 1 else mine[None]
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if (timed and on_gpu) else None
        if ev:
            ev[0].record()
        self._local_search(queries, max_search, k, mine)
        if ev:
            ev[1].record()
        if self.world > 1:
            # ONE collective: every rank's packed results (ids, dists and counts together)
            self.dist.all_gather_into_tensor(gathered.view(-1), mine.reshape(-1), group=self.group)
        if ev:
            ev[2].record()
        out = self._merge(gathered.view(self.world * self.local, pb), self.offsets, nq, k)
        if ev:
            ev[3].record()
            torch.cuda.synchronize(dev)
            self.timings = {"search_ms": ev[0].elapsed_time(ev[1]), "exchange_ms": ev[1].elapsed_time(ev[2]),
                            "merge_ms": ev[2].elapsed_time(ev[3])}
        if check_status and on_gpu and self._status is not None:
            # a shard whose exact-search scratch ran out wrote empty results: report, never merge silently
            if int(self._status[:, 0].sum().item()) != 0:
                from ._lib import ERR_OVERFLOW, GranneHipError
                raise GranneHipError(ERR_OVERFLOW, "a shard's exact-search scratch is exhausted (raise OPT_SLOW_SLOTS)")
        return out

    def exchange_bytes_per_rank(self, nq, k):
        return self.local * packed_bytes(nq, k)


def replica_query_rows(rank, world_size, n_batches, batch):
    """Replica mode (bench.py --gpus N): rank r searches rows [r*n_batches*batch, (r+1)*...) of the
    query stream -- disjoint work, no collective on the data path."""
    per = n_batches * batch
    return rank * per, per

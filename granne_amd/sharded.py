"""Partitioned (sharded) search: the element set is split into independent indexes, one per
rank / GPU; every rank answers the same query batch on its shard, the per-shard top-k lists are
exchanged with ONE all-gather (nq*k*(8+4) bytes per rank: 120 KB at nq=1024, k=10 -- latency, not
bandwidth, over xGMI) and merged by (dist, global id). SURVEY.md 8e; the reference's own
sharding helper splits elements the same way (src/elements/embeddings/parsing.rs:63-100).

One process per GPU with torch.distributed (backend "nccl" = RCCL on ROCm; "gloo" in the CPU
tests). The local search and the merge default to the HIP kernels behind the C ABI; they are
constructor parameters only so that tests/test_sharded_gloo.py can drive the exchange logic on a
GPU-less box with stand-ins from oracle/ -- this module itself contains no CPU implementation.
"""
import ctypes as C

import numpy as np


def shard_bounds(n_elements, world_size):
    """Shard g owns ids [g*ceil(n/G), min(n, (g+1)*ceil(n/G)))  (SURVEY.md 8e)."""
    per = -(-n_elements // world_size)
    return [(min(n_elements, g * per), min(n_elements, (g + 1) * per)) for g in range(world_size)]


class ShardedGranne:
    """`local_index`: this rank's granne_amd.Granne over its shard (local ids). `offset`: the
    shard's first global id. Collective: every rank must call search_batch with the same queries."""

    def __init__(self, local_index, offset, group=None, local_search=None, merge=None):
        import torch.distributed as dist
        self.dist = dist
        self.index = local_index
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.offset = int(offset)
        self._local_search = local_search or self._gpu_local_search
        self._merge = merge or self._gpu_merge
        self._offsets = None

    # ---- defaults: HIP kernels through the C ABI, tensors on this rank's GPU ---------------------
    def _gpu_local_search(self, queries, max_search, k):
        import torch
        q = queries if torch.is_tensor(queries) else torch.from_numpy(np.ascontiguousarray(queries))
        q = q.cuda().contiguous()
        nq = q.shape[0]
        ids = torch.empty((nq, k), dtype=torch.int64, device="cuda")
        ds = torch.empty((nq, k), dtype=torch.float32, device="cuda")
        cnt = torch.empty(nq, dtype=torch.int32, device="cuda")
        self.index.search_batch_device(q.data_ptr(), nq, max_search, k, ids.data_ptr(), ds.data_ptr(), cnt.data_ptr(),
                                       0, 0, torch.cuda.current_stream().cuda_stream)
        return ids, ds, cnt

    def _gpu_merge(self, ids, ds, cnt, offsets, k):
        import torch
        from ._lib import check, lib
        G, nq, _ = ids.shape
        out_ids = torch.empty((nq, k), dtype=torch.int64, device=ids.device)
        out_d = torch.empty((nq, k), dtype=torch.float32, device=ids.device)
        out_c = torch.empty(nq, dtype=torch.int32, device=ids.device)
        off = (C.c_uint64 * G)(*[int(o) for o in offsets])
        check(lib().granne_hip_merge_topk_device(C.c_void_p(ids.data_ptr()), C.c_void_p(ds.data_ptr()),
                                                 C.c_void_p(cnt.data_ptr()), off, G, nq, k,
                                                 C.c_void_p(out_ids.data_ptr()), C.c_void_p(out_d.data_ptr()),
                                                 C.c_void_p(out_c.data_ptr()), ids.device.index or 0,
                                                 C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return out_ids, out_d, out_c

    # ---- the exchange --------------------------------------------------------------------------------
    def _all_offsets(self, device):
        import torch
        if self._offsets is None:
            mine = torch.tensor([self.offset], dtype=torch.int64, device=device)
            if self.world > 1:
                allv = [torch.empty_like(mine) for _ in range(self.world)]
                self.dist.all_gather(allv, mine, group=self.group)
                self._offsets = [int(t.item()) for t in allv]
            else:
                self._offsets = [self.offset]
        return self._offsets

    def search_batch(self, queries, max_search, k):
        """Returns (ids [nq,k] global, dists [nq,k], counts [nq]) -- identical on every rank."""
        import torch
        ids, ds, cnt = self._local_search(queries, max_search, k)
        offsets = self._all_offsets(ids.device)
        if self.world > 1:
            g_ids = torch.empty((self.world,) + tuple(ids.shape), dtype=ids.dtype, device=ids.device)
            g_ds = torch.empty((self.world,) + tuple(ds.shape), dtype=ds.dtype, device=ds.device)
            g_cnt = torch.empty((self.world,) + tuple(cnt.shape), dtype=cnt.dtype, device=cnt.device)
            # one exchange step: all-gather of the per-shard top-k (ids, dists, counts)
            # (views of one contiguous [world][...] buffer: the same layout on RCCL and on gloo)
            self.dist.all_gather(list(g_ids.unbind(0)), ids.contiguous(), group=self.group)
            self.dist.all_gather(list(g_ds.unbind(0)), ds.contiguous(), group=self.group)
            self.dist.all_gather(list(g_cnt.unbind(0)), cnt.contiguous(), group=self.group)
        else:
            g_ids, g_ds, g_cnt = ids[None], ds[None], cnt[None]
        return self._merge(g_ids, g_ds, g_cnt, offsets, k)


def replica_query_rows(rank, world_size, n_batches, batch):
    """Replica mode (bench.py --gpus N): rank r searches rows [r*n_batches*batch, (r+1)*...) of the
    query stream -- disjoint work, no collective on the data path."""
    per = n_batches * batch
    return rank * per, per

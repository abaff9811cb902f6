"""Partitioned (sharded) search, one process per GPU.

The element set is split into independent indexes (the reference's own shard helper splits
elements the same way, src/elements/embeddings/parsing.rs:63-100); a rank holds one or more
shards on its GPU. Every rank answers the same query batch on its shards, writing each shard's
top-k into ONE packed buffer ([nq*k u64 local ids][nq*k f32 dists][nq u32 counts]:
granne_hip_packed_topk_bytes) followed by the shard's four status words; ONE all-gather of those
buffers is the exchange step (124 KB per shard at nq = 1024, k = 10 -- latency-, not bandwidth-bound
over xGMI), then the merge kernel ranks the n_shards*k candidates of each query by (dist, global id).
The status words travel in the same collective, so a shard whose exact-search scratch ran out is seen
by EVERY rank: all ranks raise together instead of diverging. SURVEY.md 8e.

Batches are pipelined (`search_batches`): a batch's all-gather and merge run on a side stream while
the next batch is searched on the shard streams -- the exchange costs one collective latency per
batch, and that latency is hidden behind the next batch's search. `search_batch` is the one-batch
form (and reports the phases when `timed`).

torch.distributed supplies the collective (backend "nccl" = RCCL on ROCm; "gloo" in the CPU
tests). The local search and the merge are the HIP kernels behind the C ABI; they are constructor
parameters only so that tests/test_sharded_gloo.py can drive the exchange logic on a GPU-less box
with stand-ins from oracle/ -- this module itself contains no CPU implementation. A host that
drives several GPUs from one process uses granne_hip_sharded_* (include/granne_hip.h) instead.
"""
import ctypes as C

import numpy as np

STATUS_BYTES = 16  # u32[4] after each shard's packed top-k: [0] exact-search scratch exhausted, [1] hand-overs, [2] spills


def shard_bounds(n_elements, n_shards):
    """Shard g owns ids [g*ceil(n/G), min(n, (g+1)*ceil(n/G)))  (SURVEY.md 8e)."""
    per = -(-n_elements // n_shards)
    return [(min(n_elements, g * per), min(n_elements, (g + 1) * per)) for g in range(n_shards)]


def packed_bytes(nq, k):
    return (nq * k * 12 + nq * 4 + 15) & ~15


class _Slot:
    """One batch in flight: this rank's packed results, the gathered ones, and what marks the batch done."""

    def __init__(self):
        self.key = None
        self.mine = self.gathered = None
        self.out = None
        self.done = None   # GPU: event recorded after the merge; CPU: the collective's work handle
        self.busy = False
        self.nq = self.k = 0


class ShardedGranne:
    """`local_indexes`: this rank's granne_amd.Granne objects (local ids), in global shard order
    rank*len(local_indexes) + i. `all_offsets`: the first global id of EVERY shard of the job (known
    to every rank: shard_bounds is deterministic), world*len(local_indexes) entries.
    Collective: every rank must call search_batch / search_batches with the same queries."""

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
        self._streams = None      # one per local shard
        self._xstream = None      # exchange + merge
        self._slots = []
        self._exhausted = None    # device flag: a shard of the job reported exhausted scratch (checked lazily)
        self.timings = None  # set by search_batch(timed=True): HIP-event ms of search / exchange / merge

    # ---- defaults: HIP kernels through the C ABI, tensors on this rank's GPU ---------------------
    def _device(self):
        import torch
        return torch.device("cuda", self.indexes[0].device) if self._on_gpu else torch.device("cpu")

    def _gpu_streams(self):
        import torch
        if self._streams is None:
            dev = self._device()
            self._streams = [torch.cuda.Stream(device=dev) for _ in self.indexes]
            self._xstream = torch.cuda.Stream(device=dev)
        return self._streams

    def _gpu_local_search(self, queries, max_search, k, out, wait_for=None):
        """Every local shard searches the batch on a stream of its own; shard i's packed results and status words
        land in out[i]. Returns the events recorded after each shard's search."""
        import torch
        from ._lib import check, lib
        dev = self._device()
        q = queries if torch.is_tensor(queries) else torch.from_numpy(np.ascontiguousarray(queries))
        q = q.to(dev).contiguous()
        nq = q.shape[0]
        pb = packed_bytes(nq, k)
        cur = torch.cuda.current_stream(dev)
        if wait_for is not None:
            cur.wait_event(wait_for)  # the slot's previous batch has been gathered and merged
        out[:, pb:].zero_()  # the shards' status words
        ready = torch.cuda.Event()
        ready.record(cur)
        events = []
        for i, ix in enumerate(self.indexes):
            s = self._gpu_streams()[i]
            s.wait_event(ready)  # the queries and `out` are ready on the caller's stream
            if wait_for is not None:
                s.wait_event(wait_for)
            q.record_stream(s)
            check(lib().granne_hip_search_batch_packed_device(ix._h, C.c_void_p(q.data_ptr()), nq, int(max_search), int(k),
                                                              C.c_void_p(out[i].data_ptr()),
                                                              C.c_void_p(out[i].data_ptr() + pb),
                                                              C.c_void_p(s.cuda_stream)))
            e = torch.cuda.Event()
            e.record(s)
            events.append(e)
        return events

    def _gpu_merge(self, gathered, offsets, nq, k):
        """gathered: [n_shards, packed_bytes + STATUS_BYTES] on the GPU; runs on the current stream."""
        import torch
        from ._lib import check, lib
        G = len(offsets)
        dev = gathered.device
        out_ids = torch.empty((nq, k), dtype=torch.int64, device=dev)
        out_d = torch.empty((nq, k), dtype=torch.float32, device=dev)
        out_c = torch.empty(nq, dtype=torch.int32, device=dev)
        off = (C.c_uint64 * G)(*offsets)
        check(lib().granne_hip_merge_topk_packed_strided_device(
            C.c_void_p(gathered.data_ptr()), gathered.stride(0), off, G, nq, k, C.c_void_p(out_ids.data_ptr()),
            C.c_void_p(out_d.data_ptr()), C.c_void_p(out_c.data_ptr()), dev.index or 0,
            C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        return out_ids, out_d, out_c

    # ---- one batch through search, the one exchange step and the merge ------------------------------------------
    def _slot(self, i, nq, k):
        import torch
        while len(self._slots) <= i:
            self._slots.append(_Slot())
        sl = self._slots[i]
        if sl.key != (nq, k):
            dev = self._device()
            pbs = packed_bytes(nq, k) + STATUS_BYTES
            sl.mine = torch.zeros((self.local, pbs), dtype=torch.uint8, device=dev)  # this rank's shards write here
            sl.gathered = (torch.zeros((self.world, self.local, pbs), dtype=torch.uint8, device=dev)
                           if self.world > 1 else sl.mine[None])
            sl.key, sl.nq, sl.k = (nq, k), nq, k
            sl.done = None
        return sl

    def _start(self, sl, queries, max_search, k, ev=None):
        """Enqueue one batch: shard searches, then (side stream) all-gather + merge."""
        import torch
        nq = sl.nq
        if self._on_gpu:
            dev = self._device()
            cur = torch.cuda.current_stream(dev)
            self._gpu_streams()
            if ev:
                ev[0].record(cur)
            # the slot's buffers are free once its previous batch has been merged
            searched = self._local_search(queries, max_search, k, sl.mine, wait_for=sl.done)
            x = self._xstream
            for e in searched:
                x.wait_event(e)
            with torch.cuda.stream(x):
                if ev:
                    ev[1].record(x)
                if self.world > 1:
                    # ONE collective: every rank's packed results (ids, dists, counts and status words together)
                    self.dist.all_gather_into_tensor(sl.gathered.view(-1), sl.mine.reshape(-1), group=self.group)
                if ev:
                    ev[2].record(x)
                sl.out = self._merge(sl.gathered.view(self.world * self.local, -1), self.offsets, nq, k)
                if ev:
                    ev[3].record(x)
                sl.done = torch.cuda.Event()
                sl.done.record(x)
        else:
            self._local_search(queries, max_search, k, sl.mine)
            sl.done = None
            if self.world > 1:
                sl.done = self.dist.all_gather_into_tensor(sl.gathered.view(-1), sl.mine.reshape(-1), group=self.group,
                                                           async_op=True)
        sl.busy = True

    def _finish(self, sl, check_status):
        """The batch's results, usable on the caller's stream."""
        import torch
        nq, k = sl.nq, sl.k
        if self._on_gpu:
            cur = torch.cuda.current_stream(self._device())
            cur.wait_event(sl.done)
            for t in sl.out:
                t.record_stream(cur)
            out = sl.out
        else:
            if sl.done is not None:
                sl.done.wait()
            out = self._merge(sl.gathered.view(self.world * self.local, -1), self.offsets, nq, k)
        sl.busy = False
        if check_status:
            # the status words of EVERY shard of the job came with the all-gather: a shard whose exact-search scratch
            # ran out wrote empty results -- every rank sees it (and raises in _raise_if_exhausted), nobody merges silently
            pb = packed_bytes(nq, k)
            bad = sl.gathered.view(self.world * self.local, -1)[:, pb:pb + 4].any()
            self._exhausted = bad if self._exhausted is None else (self._exhausted | bad)
        return out

    def _raise_if_exhausted(self):
        bad, self._exhausted = self._exhausted, None
        if bad is not None and bool(bad.item()):  # one synchronisation
            from ._lib import ERR_OVERFLOW, GranneHipError
            raise GranneHipError(ERR_OVERFLOW, "a shard's exact-search scratch is exhausted (raise OPT_SLOW_SLOTS)")

    def search_batch(self, queries, max_search, k, check_status=True, timed=False):
        """Returns (ids [nq,k] global, dists [nq,k], counts [nq]) -- identical on every rank."""
        import torch
        nq = int(queries.shape[0])
        sl = self._slot(0, nq, k)
        if sl.busy:
            raise RuntimeError("search_batch while batches of search_batches are in flight")
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if (timed and self._on_gpu) else None
        try:
            self._start(sl, queries, max_search, k, ev)
            out = self._finish(sl, check_status)
        except BaseException:
            self._drain()
            raise
        if check_status:
            self._raise_if_exhausted()
        if ev:
            torch.cuda.synchronize(self._device())
            self.timings = {"search_ms": ev[0].elapsed_time(ev[1]), "exchange_ms": ev[1].elapsed_time(ev[2]),
                            "merge_ms": ev[2].elapsed_time(ev[3])}
        return out

    def search_batches(self, batches, max_search, k, depth=2, check_status=True):
        """Pipelined: up to `depth` batches in flight -- batch i's all-gather + merge overlap batch i+1's search.
        `batches`: a sequence of [nq, dim] query batches (the same on every rank). Returns the list of
        (ids, dists, counts), in order; results are identical to search_batch's."""
        depth = max(1, int(depth))
        results = [None] * len(batches)
        pending = []  # (batch index, slot)
        try:
            for b, q in enumerate(batches):
                if b % depth < len(self._slots) and self._slots[b % depth].busy:  # the slot's previous batch first
                    pb_, psl = pending.pop(0)
                    assert psl is self._slots[b % depth]
                    results[pb_] = self._finish(psl, check_status)
                sl = self._slot(b % depth, int(q.shape[0]), k)
                self._start(sl, q, max_search, k)
                pending.append((b, sl))
            while pending:
                pb_, psl = pending.pop(0)
                results[pb_] = self._finish(psl, check_status)
        except BaseException:
            self._drain()  # a failed launch or collective: no slot stays marked in flight, the object stays usable
            raise
        if check_status:
            self._raise_if_exhausted()  # once for the whole run of batches: every rank raises together
        return results

    def _drain(self):
        """After a failure inside the pipeline: wait for whatever the slots still have running (best effort) and
        forget it -- the next call starts from free slots on every rank that got here."""
        for sl in self._slots:
            if not sl.busy:
                continue
            try:
                if self._on_gpu:
                    if sl.done is not None:
                        sl.done.synchronize()
                elif sl.done is not None:
                    sl.done.wait()
            except Exception:
                pass
            sl.busy = False
            sl.done = None
        self._exhausted = None

    def status_of_last_batch(self, slot=0):
        """[n_shards, 4] int32: the status words of every shard of the job for the last batch of that slot."""
        import torch
        sl = self._slots[slot]
        pb = packed_bytes(sl.nq, sl.k)
        return sl.gathered.view(self.world * self.local, -1)[:, pb:pb + STATUS_BYTES].contiguous().view(torch.int32)

    def exchange_bytes_per_rank(self, nq, k):
        return self.local * (packed_bytes(nq, k) + STATUS_BYTES)


class ShardedHost:
    """granne_hip_sharded_* (include/granne_hip.h): the partitioned index as ONE host process drives it -- what a Rust
    host binds. `indexes`: granne_amd.Granne objects (local ids), any devices; `offsets`: the first
    global id of each; `groups`: the exchange group of each (None: one per device). Device calls take raw device pointers (int) on `self.device` (= the first shard's)."""

    def __init__(self, indexes, offsets, depth=None, exchange=None, groups=None):
        from ._lib import SHARDED_OPT_DEPTH, SHARDED_OPT_EXCHANGE, check, lib
        self._lib, self._check = lib(), check
        self.indexes = list(indexes)  # borrowed by the handle: kept alive here
        G = len(self.indexes)
        handles = (C.c_void_p * G)(*[ix._h for ix in self.indexes])
        offs = (C.c_uint64 * G)(*[int(o) for o in offsets])
        self._h = C.c_void_p()
        if groups is None:
            check(self._lib.granne_hip_sharded_create(C.byref(self._h), handles, offs, G))
        else:  # the exchange group of every shard (several on one device: the multi-device exchange on a single GPU)
            gr = (C.c_uint32 * G)(*[int(g) for g in groups])
            check(self._lib.granne_hip_sharded_create_grouped(C.byref(self._h), handles, offs, G, gr))
        self.device = int(self._lib.granne_hip_sharded_device(self._h))
        self.dim, self.np_dtype = self.indexes[0].dim, self.indexes[0].np_dtype
        if depth is not None:
            self.set_option(SHARDED_OPT_DEPTH, depth)
        if exchange is not None:
            self.set_option(SHARDED_OPT_EXCHANGE, exchange)

    @classmethod
    def build(cls, element_type, elements, n_shards, devices=(0,), depth=None, exchange=None, **config):
        """granne_hip_sharded_build: the whole element set ([n, dim], prepared) split into n_shards id ranges, every shard
        built with the GPU builder under `config` (GranneBuilder's keyword arguments: num_neighbors, max_search,
        reinsert_elements, layer_multiplier, batch_max, batch_div) on devices[s // ceil(n_shards / len(devices))].
        The handle owns its shard indexes."""
        from ._lib import BuildConfig, SHARDED_OPT_DEPTH, SHARDED_OPT_EXCHANGE, check, lib
        from .index import _ELEMENT_TYPES
        et = element_type.lower()
        if et not in _ELEMENT_TYPES:
            raise ValueError("Invalid element type")
        code, np_dtype = _ELEMENT_TYPES[et]
        el = np.ascontiguousarray(elements, dtype=np_dtype)
        if el.ndim != 2:
            raise ValueError("elements must be [n, dim]")
        cfg = BuildConfig()
        lib().granne_hip_build_config_default(C.byref(cfg))
        for key, value in config.items():
            if not hasattr(cfg, key):
                raise TypeError("unknown build option %r" % key)
            setattr(cfg, key, type(getattr(cfg, key))(value))
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        self = cls.__new__(cls)
        self._lib, self._check = lib(), check
        self.indexes = []  # the handle owns its shards
        self._h = C.c_void_p()
        check(lib().granne_hip_sharded_build(C.byref(self._h), C.byref(cfg), el.ctypes.data_as(C.c_void_p), el.shape[0],
                                             el.shape[1], code, int(n_shards), devs, len(devices)))
        self.device = int(lib().granne_hip_sharded_device(self._h))
        self.dim, self.np_dtype = el.shape[1], np_dtype
        if depth is not None:
            self.set_option(SHARDED_OPT_DEPTH, depth)
        if exchange is not None:
            self.set_option(SHARDED_OPT_EXCHANGE, exchange)
        return self

    def num_shards(self):
        return int(self._lib.granne_hip_sharded_num_shards(self._h))

    def shard_offset(self, shard):
        return int(self._lib.granne_hip_sharded_shard_offset(self._h, int(shard)))

    def shard_layer(self, shard, layer):
        """Layer `layer` of shard `shard` as a host [layer_len, 32 or 64] uint32 matrix (UNUSED padded): for inspection."""
        ix = self._lib.granne_hip_sharded_shard(self._h, int(shard))
        n = int(self._lib.granne_hip_index_layer_len(ix, int(layer)))
        rows = np.full((n, 64), 0xFFFFFFFF, np.uint32)
        cnt = C.c_uint32()
        for i in range(n):
            self._check(self._lib.granne_hip_index_get_neighbors(ix, i, int(layer), rows[i].ctypes.data_as(C.c_void_p), 64, C.byref(cnt)))
        return rows

    def close(self):
        if getattr(self, "_h", None):
            self._lib.granne_hip_sharded_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return int(self._lib.granne_hip_sharded_len(self._h))

    def set_option(self, option, value):
        self._check(self._lib.granne_hip_sharded_set_option(self._h, int(option), int(value)))

    def get_option(self, option):
        v = C.c_uint64(0)
        self._check(self._lib.granne_hip_sharded_get_option(self._h, int(option), C.byref(v)))
        return int(v.value)

    def search_batch_device(self, d_queries, nq, max_search, k, d_ids, d_dists, d_counts, d_status=0, stream=0):
        self._check(self._lib.granne_hip_sharded_search_batch_device(
            self._h, C.c_void_p(d_queries), int(nq), int(max_search), int(k), C.c_void_p(d_ids), C.c_void_p(d_dists),
            C.c_void_p(d_counts), C.c_void_p(d_status), C.c_void_p(stream)))

    def begin_device(self, d_queries, nq, max_search, k, d_ids, d_dists, d_counts, d_status=0, stream=0):
        t = C.c_uint64(0)
        self._check(self._lib.granne_hip_sharded_begin_device(
            self._h, C.c_void_p(d_queries), int(nq), int(max_search), int(k), C.c_void_p(d_ids), C.c_void_p(d_dists),
            C.c_void_p(d_counts), C.c_void_p(d_status), C.c_void_p(stream), C.byref(t)))
        return int(t.value)

    def end_device(self, ticket, stream=0):
        self._check(self._lib.granne_hip_sharded_end_device(self._h, C.c_uint64(ticket), C.c_void_p(stream)))

    def search_batches(self, queries, max_search, k):
        """queries: host [n_batches, nq, dim] (prepared). Returns ids [n_batches, nq, k] u64, dists f32, counts
        [n_batches, nq] u32 -- batches pipelined `depth` deep inside the library."""
        q = np.ascontiguousarray(queries, dtype=self.np_dtype)
        if q.ndim != 3 or q.shape[2] != self.dim:
            raise ValueError("queries must be [n_batches, nq, %d]" % self.dim)
        nb, nq = q.shape[0], q.shape[1]
        ids = np.empty((nb, nq, max(k, 0)), np.uint64)
        ds = np.empty((nb, nq, max(k, 0)), np.float32)
        cnt = np.zeros((nb, nq), np.uint32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        self._check(self._lib.granne_hip_sharded_search_batches(self._h, p(q), nb, nq, int(max_search), int(k), p(ids), p(ds),
                                                               p(cnt)))
        return ids, ds, cnt

    def search_batch(self, queries, max_search, k):
        q = np.ascontiguousarray(queries, dtype=self.np_dtype)
        ids, ds, cnt = self.search_batches(q[None], max_search, k)
        return ids[0], ds[0], cnt[0]


def replica_query_rows(rank, world_size, n_batches, batch):
    """Replica mode (bench.py --gpus N): rank r searches rows [r*n_batches*batch, (r+1)*...) of the
    query stream -- disjoint work, no collective on the data path."""
    per = n_batches * batch
    return rank * per, per

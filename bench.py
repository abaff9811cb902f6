#!/usr/bin/env python
"""bench.py -- queries/sec of the MI355X Granne::search path on BASELINE.json's workload.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], "C2"): 10M synthetic 100-d f32 angular vectors (uniform
[-0.5,0.5) per component, then angular::Vector::from), HNSW graph with granne's structure built
on the GPU (GranneBuilder mirror, BuildConfig::default()), batches of 1024 fresh queries,
max_search (ef) = 50, k = 10. A "step" is ONE batch of 1024 queries through Granne::search on one
GPU (one kernel launch); every step uses a different batch; queries, elements and graph are
resident in HBM before the timed region.

--mode replica (default): with N > 1 every rank holds a replica of the index on its own GPU and
searches its own batches (the path shards by query: no data-path collective; scaling = weak).
--mode partitioned: the element set is split into N id ranges (src/elements/embeddings/parsing.rs:63-100),
rank g builds and searches shard g; every rank searches the SAME batch, one all-gather of the packed
per-shard top-k (RCCL), then the merge kernel; a step = one batch through search + exchange + merge.

Rank 0 prints ONE JSON line. Besides the contract fields it carries
  roofline      HBM roofline of the dominant kernel: algorithmic bytes per launch (SURVEY.md 8d:
                n_dist*d*s + 4*n_adj + d*s + 8*k per query, n_dist = the reference's count of distinct nodes) /
                mean launch duration from HIP events recorded around the kernel on its stream
  cpu_baseline  the CPU oracle (restatement of the reference's search, OpenMP over queries = the
                caller-side rayon par_iter) timed on this box's host cores on a bounded sample of
                the same batches, with the GPU results checked against it (ids + distance bits)
  ef_sweep      recall@10 and queries/sec at max_search 50..800 on the same index
  int8          the same workload on angular_int (BASELINE.json configs[2]) as a sub-record
  secondary     a second synthetic workload on which recall@10 >= 0.95 is reachable (the headline
                data is i.i.d. uniform in 100-d, where it is not): QPS at the smallest such ef
  brute_force   the exact scan on the matrix cores (granne_hip_brute_force_device: the recall ground truth): its rate
                against the f32 MFMA peak, queries/s at recall 1.0, the oracle's scan as check and CPU baseline --
                on i.i.d. uniform 100-d data no max_search reaches recall 0.95, the scan does
  c4_shard      one shard of BASELINE.json configs[3] (12.5M x 200-d f32, batch 4096, ef 50) and
  c5_shard      one shard of configs[4] (125M x 100-d int8, batch 4096, ef 200), measured the same way
  partitioned   (WORLD_SIZE > 1 only) C2's 10M points split into WORLD_SIZE id ranges, one per rank: every rank
                searches the SAME batches, ONE all_gather_into_tensor of the packed per-shard top-k (RCCL) per
                batch, merge kernel; pipelined two batches deep
  latency_nq1   one query per call through the host-pointer API (the reference's own call shape)
  steady        the same K steps repeated back to back for >= 0.5 s (the contract's K-step window is a few ms)
"""
import argparse
import ctypes as C
import hashlib
import json
import math
import os
import sys
import time

# Batches in flight run on HIP streams of their own; HIP maps streams onto at most GPU_MAX_HW_QUEUES hardware
# queues (default 4, one of them the null stream's), and two streams that share a queue run one after the other
# (tools/inflight_probe.py: four streams on the default setting run like two; n streams on n queues fall back too --
# one queue is shared -- so there are more queues than streams). Must be set before HIP starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
# the CPU baseline's OpenMP team: one thread per core, spread over both sockets (read when libgomp loads)
if int(os.environ.get("WORLD_SIZE", "1")) <= 1:  # (several ranks on one host would all bind to the same cores)
    os.environ.setdefault("OMP_PROC_BIND", "spread")
    os.environ.setdefault("OMP_PLACES", "cores")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SEED = 0x6772616E6E65  # "granne"; queries use SEED + 1
HBM_PEAK_GBPS = 8000.0  # MI355X spec (MI355X_MICROARCH.md); 6290 measured copy ceiling


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", default="replica", choices=["replica", "partitioned"])
    ap.add_argument("--shards-per-gpu", type=int, default=1, help="partitioned mode: shards held by each rank")
    ap.add_argument("--elements", "--n", dest="n", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=100)
    ap.add_argument("--dtype", default="f32", choices=["f32", "i8"])
    ap.add_argument("--data", default="uniform", choices=["uniform", "latent"],
                    help="uniform: BASELINE.json's generator; latent: the secondary workload's")
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--ef", type=int, default=50)
    ap.add_argument("--k", type=int, default=10)
    # graph: BuildConfig::default() of the reference (src/index/mod.rs:220-231)
    ap.add_argument("--build-max-search", type=int, default=200)
    ap.add_argument("--build-reinsert", type=int, default=1)
    ap.add_argument("--num-neighbors", type=int, default=30)
    ap.add_argument("--batch-max", type=int, default=65536)
    ap.add_argument("--inflight", type=int, default=0,
                    help="batches in flight: step i is enqueued on HIP stream i %% inflight (1 = strictly sequential; "
                         "0 = per element type: up to 6 for f32, 12 for int8 -- int8 walks move a quarter of the bytes -- "
                         "the nearest count that divides --steps)")
    ap.add_argument("--cpu-batches", type=int, default=16, help="batches of the CPU baseline sample (0 = skip)")
    ap.add_argument("--cpu-seconds", type=float, default=2.0, help="wall seconds the CPU baseline is timed over (repeats its sample)")
    ap.add_argument("--c5-elements", type=int, default=125_000_000, help="elements of the c5_shard sub-record (0 = skip)")
    ap.add_argument("--c4-elements", type=int, default=12_500_000, help="elements of the c4_shard sub-record (0 = skip)")
    ap.add_argument("--no-partitioned", action="store_true", help="WORLD_SIZE > 1: skip the partitioned sub-record")
    ap.add_argument("--force-partitioned", action="store_true", help="take the partitioned sub-record with one rank too (tests)")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--sweep-ef", default="50,100,200,400,800", help="ef values of ef_sweep ('' = skip)")
    ap.add_argument("--no-recall", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the int8 / secondary / latency sub-records")
    ap.add_argument("--visited-slots", type=int, default=0,
                    help="GRANNE_HIP_OPT_VISITED_SLOTS: LDS visited-table slots per walker (0 = auto)")
    ap.add_argument("--reorder", action="store_true",
                    help="apply Granne::reorder (src/index/reorder.rs) to the built index before searching")
    return ap.parse_args()


def csrc_sha():
    """Hash of the walker's sources + build flags: PMC traffic figures in profiles/pmc_traffic.json are
    only quoted for the kernel they were measured on (the files that define its memory behaviour)."""
    from granne_amd import build as gbuild
    h = hashlib.sha256()
    for f in ("walk_fast.h", "wave_prims.h", "dist.h"):
        h.update(f.encode())
        h.update(open(os.path.join(gbuild.CSRC, f), "rb").read())
    h.update(" ".join(gbuild.FLAGS).encode())
    return h.hexdigest()[:16]


def auto_inflight(args, dtype, steps):
    """Batches in flight (step i goes to stream i % inflight). Where more stop paying (profiles/r3_inflight_novis.txt):
    f32 walks fill the memory system with six (6.2-6.3 M queries/s from six to twelve), int8 walks -- a quarter of the
    bytes, eight waves per SIMD -- with twelve. The K timed steps are few (the driver times 20): a count that divides K
    keeps the last round as full as the others, so the nearest divisor of K not below three quarters of that is taken."""
    if args.inflight:
        return max(1, args.inflight)
    best = 6 if dtype == "f32" else 12
    for d in range(best, (3 * best + 3) // 4 - 1, -1):
        if steps % d == 0:
            return d
    return best


def workload_label(n, dim, dtype, data, nq, ef, k):
    """Names what actually ran. BASELINE.json's configs get their tag only for their exact shape."""
    comp = "i.i.d. uniform components" if data == "uniform" else "16-d latent cube through a fixed random linear map"
    tag = "custom"
    if data == "uniform" and dim == 100 and n == 10_000_000 and nq == 1024 and ef == 50:
        tag = "C2 (BASELINE.json configs[1])" if dtype == "f32" else "C3 (BASELINE.json configs[2])"
    elif data == "uniform" and dtype == "f32" and dim == 200 and n == 12_500_000 and nq == 4096:
        tag = "C4 shard (one of the 8 shards of BASELINE.json configs[3]: 100M x 200-d f32)"
    elif data == "uniform" and dtype == "i8" and dim == 100 and nq == 4096 and ef == 200:
        tag = ("C5 shard (one of the 8 shards of BASELINE.json configs[4]: 1B x 100-d int8)" if n == 125_000_000
               else "C5-shaped shard at reduced size (BASELINE.json configs[4] has 125M per shard)")
    return "%s: %d x %d-d %s angular, %s, batch=%d, ef_search=%d, k=%d" % (tag, n, dim, dtype, comp, nq, ef, k)


class Bench:
    """One process = one GPU. Holds the library handles and the measurement helpers."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.args, self.torch, self.dist = args, torch, dist
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        # GRANNE_BENCH_FORCE_DIST=1 exercises the RCCL path (init, barrier, all-reduce) with one rank
        self.use_dist = self.world > 1 or bool(os.environ.get("GRANNE_BENCH_FORCE_DIST"))
        if self.use_dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            os.environ.setdefault("RANK", str(self.rank))
            os.environ.setdefault("WORLD_SIZE", str(self.world))
            torch.cuda.set_device(self.local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", self.local_rank))
        if args.gpus != self.world and self.rank == 0:
            log("note: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, self.world))
        torch.cuda.set_device(self.local_rank)
        self.dev = self.local_rank
        import granne_amd
        from granne_amd import _lib
        self.ga, self._lib, self.lib = granne_amd, _lib, _lib.lib()
        self.stream = torch.cuda.current_stream().cuda_stream
        self.sp = C.c_void_p(self.stream)
        # the in-flight streams, created ONCE: HIP assigns hardware queues at stream creation, so every measurement of
        # a run sees the same stream-to-queue mapping
        self.streams = [torch.cuda.Stream() for _ in range(16)]

    # ---- synthetic rows, generated and prepared on the device ---------------------------------------
    def synth_raw(self, seed, row0, rows, dim):
        raw = self.torch.empty((rows, dim), dtype=self.torch.float32, device="cuda")
        self._lib.check(self.lib.granne_hip_synth_rows_device(C.c_void_p(raw.data_ptr()), seed, row0, rows, dim,
                                                              self.dev, self.sp))
        return raw

    def prepare(self, raw, dtype):
        rows, dim = raw.shape
        if dtype == "f32":
            self._lib.check(self.lib.granne_hip_normalize_f32_device(C.c_void_p(raw.data_ptr()), rows, dim, self.dev, self.sp))
            return raw
        q = self.torch.empty((rows, dim), dtype=self.torch.int8, device="cuda")
        self._lib.check(self.lib.granne_hip_quantize_f32_device(C.c_void_p(raw.data_ptr()), C.c_void_p(q.data_ptr()),
                                                                rows, dim, self.dev, self.sp))
        return q

    def rows(self, data, seed, row0, rows, dim, dtype):
        """uniform: the reference's generator (src/test_helper.rs:3-6). latent: points of a LATENT-d
        uniform cube pushed through a fixed random LATENT x dim linear map (low intrinsic dimension:
        a graph index can reach recall 0.95 on it), then the same Vector::from."""
        if data == "uniform":
            if dtype == "i8" and rows > 20_000_000:  # the f32 staging of 125M rows is 50 GB: in pieces
                out = self.torch.empty((rows, dim), dtype=self.torch.int8, device="cuda")
                step = 12_500_000
                for r0 in range(0, rows, step):
                    r1 = min(rows, r0 + step)
                    out[r0:r1] = self.prepare(self.synth_raw(seed, row0 + r0, r1 - r0, dim), dtype)
                return out
            return self.prepare(self.synth_raw(seed, row0, rows, dim), dtype)
        LATENT = 16
        proj = self.synth_raw(SEED + 7, 0, LATENT, dim)
        out = self.torch.empty((rows, dim), dtype=self.torch.float32, device="cuda")
        step = 2_000_000
        for r0 in range(0, rows, step):
            r1 = min(rows, r0 + step)
            z = self.synth_raw(seed, row0 + r0, r1 - r0, LATENT)
            self.torch.matmul(z, proj, out=out[r0:r1])
        return self.prepare(out, dtype)

    def build_index(self, elements, dtype):
        a = self.args
        et = "angular" if dtype == "f32" else "angular_int"
        t0 = time.time()
        n, dim = elements.shape
        builder = self.ga.GranneBuilder.from_device(
            et, elements.data_ptr(), n, dim, device=self.dev, stream=self.stream, num_neighbors=a.num_neighbors,
            max_search=a.build_max_search, reinsert_elements=bool(a.build_reinsert), batch_max=a.batch_max,
            show_progress=False)
        builder.build()
        index = builder.get_index()
        self.torch.cuda.synchronize()
        return builder, index, time.time() - t0

    def barrier(self):
        if self.use_dist:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    # ---- one workload on one index: the timed K steps + per-launch events + counters ----------------
    def measure(self, index, queries, dim, esize, nq, ef, k, steps, warmup, inflight, contract=False, steady_s=0.0):
        torch = self.torch
        n_batches = warmup + steps
        ids = torch.empty((n_batches, nq, k), dtype=torch.int64, device="cuda")
        dists = torch.empty((n_batches, nq, k), dtype=torch.float32, device="cuda")
        counts = torch.empty((n_batches, nq), dtype=torch.int32, device="cuda")
        stats = torch.zeros((n_batches, nq, 3), dtype=torch.int64, device="cuda")
        status = torch.zeros(4, dtype=torch.int32, device="cuda")

        # raw pointers of every batch, taken once: a step is one library call (slicing five tensors per step costs more
        # host time than the call, and the K timed steps start from idle queues)
        q_ptr = [queries[b * nq:(b + 1) * nq].data_ptr() for b in range(n_batches)]
        i_ptr = [ids[b].data_ptr() for b in range(n_batches)]
        d_ptr = [dists[b].data_ptr() for b in range(n_batches)]
        c_ptr = [counts[b].data_ptr() for b in range(n_batches)]
        s_ptr = [stats[b].data_ptr() for b in range(n_batches)]
        st_ptr = status.data_ptr()

        def step(b, on):
            index.search_batch_device(q_ptr[b], nq, ef, k, i_ptr[b], d_ptr[b], c_ptr[b], s_ptr[b], st_ptr, on)

        # Step i is enqueued on stream i % inflight: a batch starts while the previous ones drain (one
        # batch of 1024 one-wave walkers fills one wave slot per SIMD). Every step is still one batch
        # of `nq` queries through one kernel launch; nothing is skipped or cached.
        inflight = min(inflight, len(self.streams))
        streams = self.streams[:inflight] if inflight > 1 else [torch.cuda.current_stream()]
        on_ = [x.cuda_stream for x in streams]
        for b in range(max(warmup, inflight)):  # (every stream has searched once: its scratch block exists)
            step(b % n_batches, on_[b % inflight])
        torch.cuda.synchronize()
        status.zero_()
        if contract:
            self.barrier()
        else:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(warmup + i, on_[i % inflight])
        if contract:
            self.barrier()
        else:
            torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if contract and self.use_dist:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)  # MAX over ranks
            elapsed = float(t.item())
        if int(status[0].item()) != 0:
            raise RuntimeError("exact-search scratch exhausted during the timed steps")

        # the same K steps again and again, back to back, until `steady_s` seconds have passed: the K-step window above
        # is a few milliseconds, this one is long enough to trust (same batches, same streams, one synchronisation)
        steady = None
        if steady_s > 0:
            rounds = max(2, int(math.ceil(steady_s / max(elapsed, 1e-6))))
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            for r in range(rounds):
                for i in range(steps):
                    step(warmup + i, on_[(r * steps + i) % inflight])
            torch.cuda.synchronize()
            dt = time.perf_counter() - t2
            steady = {"value": round(rounds * steps * nq / dt, 1), "unit": "queries/s", "steps": rounds * steps,
                      "seconds": round(dt, 3), "note": "the same K steps repeated back to back (rank-local)"}

        # the same K steps strictly one after the other on ONE stream, twice: first bare (the wall clock of K back-to-back
        # launches: what one batch at a time costs, launch gaps included), then with one pair of HIP events per step
        # recorded inside the library immediately around the walker's dispatch = what rocprofv3 reports per kernel
        # (a search is one kernel: walk_fast.h / slow_kernel.h). The events themselves are work on the stream, which is
        # why the two loops are separate.
        glib, _glib = self.lib, self._lib

        def hip_event():
            e = C.c_void_p()
            _glib.check(glib.granne_hip_event_create(C.byref(e)))
            return e

        step(0, self.stream)  # (the launch stream's scratch block exists)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(steps):
            step(warmup + i, self.stream)
        torch.cuda.synchronize()
        seq_elapsed = time.perf_counter() - t1
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        kev = [(hip_event(), hip_event()) for _ in range(steps)]
        for i in range(steps):
            b = warmup + i
            ev[i][0].record()
            index.search_batch_device_timed(queries[b * nq:(b + 1) * nq].data_ptr(), nq, ef, k, ids[b].data_ptr(),
                                            dists[b].data_ptr(), counts[b].data_ptr(), stats[b].data_ptr(),
                                            status.data_ptr(), self.stream, kev[i][0].value, kev[i][1].value)
            ev[i][1].record()
        torch.cuda.synchronize()
        call_ms = [a.elapsed_time(b) for a, b in ev]
        step_ms = []
        for a, b in kev:
            ms = C.c_float()
            _glib.check(glib.granne_hip_event_elapsed_ms(a, b, C.byref(ms)))
            step_ms.append(float(ms.value))
            glib.granne_hip_event_destroy(a)
            glib.granne_hip_event_destroy(b)

        # The algorithmic bytes are the REFERENCE algorithm's: its search evaluates every distinct node once (HashSet,
        # mod.rs:1026). The timed walkers keep no visited set and evaluate a revisited neighbor again (~3 % more rows
        # on this data; wave_prims.h VisitedNone) -- so the counters come from one more, untimed pass over the same K
        # batches with the exact bucket tables switched on, whose results must be the timed pass's, bit for bit.
        slow_n, spill_n = int(status[1].item()), int(status[2].item())  # (of the timed forms of the walk, not of the counting pass)
        evaluated = float(stats[warmup:, :, 0].sum().item())
        ids_x = torch.empty((nq, k), dtype=torch.int64, device="cuda")
        dists_x = torch.empty((nq, k), dtype=torch.float32, device="cuda")
        counts_x = torch.empty((nq,), dtype=torch.int32, device="cuda")
        stats_x = torch.zeros((steps, nq, 3), dtype=torch.int64, device="cuda")
        mode = index.get_option(_glib.OPT_VISITED16)
        index.set_option(_glib.OPT_VISITED16, 3)
        same = True
        try:
            for i in range(steps):
                b = warmup + i
                index.search_batch_device(queries[b * nq:(b + 1) * nq].data_ptr(), nq, ef, k, ids_x.data_ptr(),
                                          dists_x.data_ptr(), counts_x.data_ptr(), stats_x[i].data_ptr(),
                                          status.data_ptr(), self.stream)
                torch.cuda.synchronize()
                same = same and bool((ids_x == ids[b]).all().item()) and bool((counts_x == counts[b]).all().item()) \
                    and bool((dists_x.view(torch.int32) == dists[b].view(torch.int32)).all().item())
        finally:
            index.set_option(_glib.OPT_VISITED16, mode)
        if not same:
            raise RuntimeError("the walk without a visited set and the walk with the exact tables returned different results")
        if not bool((stats_x[:, :, 1:] == stats[warmup:, :, 1:]).all().item()):
            raise RuntimeError("expansion / adjacency counters differ between the two forms of the walk")
        st = stats_x.sum(dim=(0, 1)).cpu().numpy().astype(np.float64)  # n_dist, n_expand, n_adj (the reference's counts)
        alg_total = st[0] * dim * esize + st[2] * 4 + steps * nq * (dim * esize + k * 8)
        alg_per_launch = alg_total / steps
        mean_ms = float(np.mean(step_ms))
        achieved = alg_per_launch / (mean_ms * 1e-3) / 1e9
        return {
            "elapsed": elapsed, "value_local": steps * nq / elapsed, "seq_elapsed": seq_elapsed, "steady": steady,
            "ids": ids, "dists": dists, "counts": counts, "status": status, "inflight": inflight,
            "slow": slow_n, "spill": spill_n,
            "alg_per_launch": alg_per_launch, "achieved": achieved, "launch_ms_mean": mean_ms,
            "launch_ms_min": float(np.min(step_ms)), "call_ms_mean": float(np.mean(call_ms)),
            "per_query": {"n_dist": round(st[0] / (steps * nq), 1), "n_expand": round(st[1] / (steps * nq), 1),
                          "n_adj": round(st[2] / (steps * nq), 1), "rows_evaluated": round(evaluated / (steps * nq), 1)},
            "same_as_exact_set_walk": {"queries": steps * nq, "ids_dists_counts_bit_exact": True},
        }

    def roofline(self, m, traffic_key, value_per_gpu, nq):
        traffic, note = None, None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                ent = json.load(f).get(traffic_key)
            if ent:
                if ent.get("csrc_sha") == csrc_sha():
                    traffic = ent.get("hbm_bytes_per_launch")
                else:
                    note = "PMC traffic in profiles/pmc_traffic.json was measured on other kernel sources (%s): not quoted" % ent.get("csrc_sha")
        except Exception:
            pass
        r = {
            "bound": "hbm", "kernel": "fast_kernel (walk_fast.h)", "achieved": round(m["achieved"], 1), "peak": HBM_PEAK_GBPS,
            "unit": "GB/s", "frac": round(m["achieved"] / HBM_PEAK_GBPS, 4), "traffic": traffic,
            "aggregate_achieved_with_inflight": round(m["alg_per_launch"] * (value_per_gpu / nq) / 1e9, 1),
            "aggregate_frac_with_inflight": round(m["alg_per_launch"] * (value_per_gpu / nq) / 1e9 / HBM_PEAK_GBPS, 4),
            "alg_bytes_per_launch": int(m["alg_per_launch"]), "launch_ms_mean": round(m["launch_ms_mean"], 4),
            "launch_ms_min": round(m["launch_ms_min"], 4), "call_ms_mean": round(m["call_ms_mean"], 4),
            "per_query": m["per_query"], "same_as_exact_set_walk": m["same_as_exact_set_walk"],
        }
        if note:
            r["traffic_note"] = note
        return r

    # ---- ground truth / recall -----------------------------------------------------------------------
    def ground_truth(self, index, q0, k, dtype, timing=None, n=None):
        """exact k nearest elements by the library's scan on the matrix cores (granne_hip_brute_force_device,
        granne_amd/csrc/brute_force.h): ids [nq, k]. `timing`: a dict that receives the scan's rate."""
        torch = self.torch
        nq = q0.shape[0]
        ids = torch.empty((nq, k), dtype=torch.int64, device="cuda")
        ds = torch.empty((nq, k), dtype=torch.float32, device="cuda")
        cnt = torch.empty(nq, dtype=torch.int32, device="cuda")
        q0 = q0.contiguous()

        def run():
            index.brute_force_device(q0.data_ptr(), nq, k, ids.data_ptr(), ds.data_ptr(), cnt.data_ptr(), self.stream)
        run()
        torch.cuda.synchronize()
        if timing is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            n, dim = (n or len(index)), index.dim
            if dtype == "f32":
                flops = 2.0 * nq * n * dim
                timing.update({"kernel": "bf_f32_kernel (v_mfma_f32_32x32x2_f32) + merge + exact re-ranking", "ms": round(ms, 3),
                               "value": round(nq / (ms * 1e-3), 1), "value_unit": "queries/s at recall 1.0 (exact scan)",
                               "bound": "mfma", "achieved": round(flops / ms / 1e9, 1), "peak": 157.3, "unit": "TFLOP/s",
                               "frac": round(flops / ms / 1e9 / 157.3, 4), "queries": nq, "elements": n, "dim": dim,
                               "note": "2 * nq * n * dim flops / wall of the whole operator (HIP events); peak = dense f32 MFMA"})
            else:
                tiles = (nq + 255) // 256
                byts = float(tiles) * n * 128
                timing.update({"kernel": "bf_i8_kernel (v_mfma_i32_32x32x16_i8) + merge + exact re-ranking", "ms": round(ms, 3),
                               "value": round(nq / (ms * 1e-3), 1), "value_unit": "queries/s at recall 1.0 (exact scan)",
                               "bound": "hbm", "achieved": round(byts / ms / 1e6, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                               "frac": round(byts / ms / 1e6 / HBM_PEAK_GBPS, 4), "queries": nq, "elements": n, "dim": dim,
                               "note": "every tile of 256 queries streams the n 128-byte rows once: %d passes" % tiles})
        self._gt_dists = ds.cpu().numpy()
        return ids.cpu().numpy()

    def check_scan(self, oix, h_q, gt_ids, k, bf, n_check=32):
        """The exact scan against the oracle's scan (the reference's Dist for every (query, element) pair, OpenMP over
        the elements) on the first queries of the batch: its check and its CPU baseline. Adds to the brute_force record."""
        from oracle import oracle as orc
        m = min(n_check, h_q.shape[0])
        sec, o_ids, o_d = oix.scan_topk(h_q[:m], k, n_threads=0)
        g_ids, g_d = gt_ids[:m].astype(np.uint64), self._gt_dists[:m]
        bf["cpu_baseline"] = {"value": round(m / sec, 1), "unit": "queries/s", "cores": orc.lib().gro_max_threads(), "kind": "port",
                              "sample": "%d queries against all %d elements: the reference's Dist per pair (oracle gro_scan_topk, "
                                        "OpenMP over the elements, every row evaluated against all queries while it is in L1), "
                                        "%.2f s" % (m, len(oix.elements), sec)}
        bf["speedup_vs_cpu"] = round(bf["value"] / bf["cpu_baseline"]["value"], 1)
        bf["matches_oracle_scan"] = {"ids_equal_fraction": round(float((g_ids == o_ids).mean()), 5),
                                     "dists_bit_exact": bool(g_d.tobytes() == o_d.tobytes()),
                                     "max_abs_dist_diff": float(np.abs(g_d - o_d).max()), "queries_checked": int(m),
                                     "note": "tolerance mode: candidates are selected by the MFMA score, distances recomputed "
                                             "in the reference's arithmetic (granne_amd/csrc/brute_force.h)"}

    @staticmethod
    def recall(gt, got, k):
        got = got.cpu().numpy()
        return float(np.mean([len(set(gt[i]) & set(got[i])) / k for i in range(gt.shape[0])]))

    def launch_scaling(self, index, data, dim, dtype, ef, k, batches=(4096, 16384)):
        """The same kernel with more walks per launch, one launch at a time: finished walks' SIMDs are refilled from
        the same grid, so the per-launch roofline fraction shows what the batch of 1024 leaves idle.
        Not the headline: BASELINE.json fixes the batch."""
        esize = 4 if dtype == "f32" else 1
        out = []
        for nq in batches:
            steps, warmup = 20, 1
            q = self.rows(data, SEED + 3, 0, (steps + warmup) * nq, dim, dtype)
            m = self.measure(index, q, dim, esize, nq, ef, k, steps, warmup, 1)
            out.append({"batch": nq, "steps": steps, "launch_ms_mean": round(m["launch_ms_mean"], 4),
                        "frac": round(m["achieved"] / HBM_PEAK_GBPS, 4),
                        "qps_one_launch_at_a_time": round(steps * nq / m["seq_elapsed"], 1),
                        "slow_path_queries": m["slow"]})
            del m, q
            self.torch.cuda.empty_cache()
        return out

    def timed_window(self, run, n_distinct, min_s=0.05):
        """calls/s of run(j) over a window of at least min_s seconds (sized from a short probe)."""
        torch = self.torch
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for j in range(4):
            run(j % n_distinct)
        torch.cuda.synchronize()
        per = (time.perf_counter() - t0) / 4
        reps = max(8, int(math.ceil(min_s / max(per, 1e-7))))
        t0 = time.perf_counter()
        for j in range(reps):
            run(j % n_distinct)
        torch.cuda.synchronize()
        return reps / (time.perf_counter() - t0), reps

    def ef_sweep(self, index, queries, gt, nq, k, efs, steps, warmup, inflight, stop_at=None):
        """recall@10 (first timed batch) and queries/sec over windows of >= 50 ms, one batch at a time and in flight."""
        torch = self.torch
        out = []
        n_b = queries.shape[0] // nq
        o = (torch.empty((nq, k), dtype=torch.int64, device="cuda"), torch.empty((nq, k), dtype=torch.float32, device="cuda"),
             torch.empty((nq,), dtype=torch.int32, device="cuda"))
        streams = self.streams[:inflight]
        for e_ in efs:
            def run(b, on):
                index.search_batch_device(queries[b * nq:(b + 1) * nq].data_ptr(), nq, e_, k, o[0].data_ptr(), o[1].data_ptr(),
                                          o[2].data_ptr(), 0, 0, on)
            run(warmup, self.stream)
            torch.cuda.synchronize()
            rec = self.recall(gt, o[0], k)
            nd = n_b - warmup
            r_seq, reps = self.timed_window(lambda j: run(warmup + j, self.stream), nd)
            cnt = [0]

            def infl(j):  # outputs overwrite each other: timing only
                run(warmup + j, streams[cnt[0] % len(streams)].cuda_stream)
                cnt[0] += 1
            r_inf, reps_i = self.timed_window(infl, nd)
            out.append({"ef": e_, "recall_at_10": round(rec, 4), "qps": round(r_inf * nq, 1),
                        "qps_one_batch_at_a_time": round(r_seq * nq, 1), "batches_timed": reps_i})
            if stop_at is not None and rec >= stop_at:
                break
        return out

    # ---- the CPU oracle beside it ---------------------------------------------------------------------
    def host_index(self, elements, builder, order=None):
        """the oracle's view of the index: host copies of the elements (parallel first touch: the pages end up spread
        over the NUMA nodes of the threads that wrote them instead of all on one socket) and of the layers"""
        from concurrent.futures import ThreadPoolExecutor
        from oracle import oracle as orc
        orc.build()
        n = elements.shape[0]
        np_dt = np.float32 if elements.dtype == self.torch.float32 else np.int8
        # the rows the walks gather at random live in transparent huge pages when the kernel grants them (4 GB in 4 KB
        # pages is a TLB miss per row): the baseline should not lose to page walks what a tuned host would not
        nbytes = int(np.prod(elements.shape)) * np.dtype(np_dt).itemsize
        try:
            import mmap
            buf = mmap.mmap(-1, max(nbytes, 1))
            buf.madvise(mmap.MADV_HUGEPAGE)
            h_el = np.frombuffer(buf, dtype=np_dt, count=int(np.prod(elements.shape))).reshape(tuple(elements.shape))
            self._host_pages = "transparent huge pages requested (MADV_HUGEPAGE)"
        except Exception:
            h_el = np.empty(tuple(elements.shape), np_dt)
            self._host_pages = "default pages"
        parts = 64
        bounds = [n * i // parts for i in range(parts + 1)]

        def cp(i):
            h_el[bounds[i]:bounds[i + 1]] = elements[bounds[i]:bounds[i + 1]].cpu().numpy()
        with ThreadPoolExecutor(8) as ex:
            list(ex.map(cp, range(parts)))
        oix = orc.Index(h_el, builder.layers())
        if order is not None:
            oix = oix.reordered(order)
        return oix

    def cpu_baseline(self, oix, h_q, ef, k, g_ids, g_d, single_thread_queries=256):
        """the ONLY use of oracle/ in this file: the CPU baseline + parity check (the index view comes from host_index)."""
        from oracle import oracle as orc
        a = self.args
        nqs = h_q.shape[0]
        # thread count: the best of {OpenMP default, all logical CPUs} unless given (a cgroup quota below
        # the logical CPU count makes oversubscription much slower)
        cands = [a.cpu_threads] if a.cpu_threads else sorted({orc.lib().gro_max_threads(), os.cpu_count() or 1})
        best = None
        for th in cands:
            # one untimed pass inside the same parallel region, then as many timed passes as fill cpu_seconds
            probe, _, _, _ = oix.search_batch_timed(h_q, ef, k, n_threads=th, repeats=1)
            reps = max(1, int(math.ceil(a.cpu_seconds / max(probe, 1e-6))))
            sec, o_ids, o_d, o_c = oix.search_batch_timed(h_q, ef, k, n_threads=th, repeats=reps)
            rate = reps * nqs / sec
            if best is None or rate > best[0]:
                best = (rate, th, sec, reps, o_ids, o_d)
        rate, threads, cpu_s, reps, o_ids, o_d = best
        # one thread, on queries it has not just walked (a repeated pass over a few hundred queries runs out of L3)
        m1 = min(single_thread_queries, max(1, nqs // 2))
        oix.search_batch(h_q[:8], ef, k, n_threads=1)
        t1 = time.perf_counter()
        oix.search_batch(h_q[nqs - m1:], ef, k, n_threads=1)
        single = m1 / (time.perf_counter() - t1)
        ids_ok = bool((g_ids == o_ids).all())
        d_ok = g_d.tobytes() == o_d.tobytes()
        phys = (os.cpu_count() or 2) // 2
        return {
            "value": round(rate, 1), "unit": "queries/s", "cores": threads, "kind": "port",
            "single_thread": {"value": round(single, 1), "unit": "queries/s", "queries": int(m1)},
            "parallel_efficiency": round(rate / (min(threads, phys) * single), 3),
            "sample": "%d queries of the timed workload, same index, %d passes = %.2f s wall after one untimed pass "
                      "(gro_search_batch_timed: one OpenMP region, dynamic schedule over queries, threads and their scratch kept "
                      "between passes; OMP_PROC_BIND=%s OMP_PLACES=%s); oracle/granne_oracle.c (C restatement of the reference's "
                      "search; Rust toolchain absent); thread counts tried %s, best reported; elements first-touched by 8 threads, %s; "
                      "parallel_efficiency = value / (min(threads, %d physical cores) x single_thread)"
                      % (nqs, reps, cpu_s, os.environ.get("OMP_PROC_BIND"), os.environ.get("OMP_PLACES"), cands,
                         getattr(self, "_host_pages", "default pages"), phys),
            "gpu_matches_oracle": {"ids_bit_exact": ids_ok, "dists_bit_exact": bool(d_ok), "queries_checked": int(nqs)},
        }

    def latency_nq1(self, index, queries, dim, ef, k, reps=300):
        """one query per call, host pointers in and out (Granne::search's call shape, src/index/mod.rs:140-150)"""
        q = queries[:reps].cpu().numpy()
        ids = np.empty((1, k), np.uint64)
        ds = np.empty((1, k), np.float32)
        cnt = np.zeros(1, np.uint32)
        lat = []
        fn, h = self.lib.granne_hip_search_batch, index._h
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        for i in range(reps):
            qi = np.ascontiguousarray(q[i:i + 1])
            t0 = time.perf_counter()
            rc = fn(h, p(qi), 1, ef, k, p(ids), p(ds), p(cnt), None)
            lat.append(time.perf_counter() - t0)
            if rc:
                self._lib.check(rc)
        lat = np.array(lat[20:]) * 1e6
        return {"unit": "us", "median": round(float(np.median(lat)), 1), "p99": round(float(np.percentile(lat, 99)), 1),
                "calls": int(lat.size), "note": "granne_hip_search_batch with nq = 1: host buffers in and out (PCIe both ways) "
                "and one stream synchronisation per call"}


def run_replica(B, args):
    torch = B.torch
    world, rank = B.world, B.rank
    n, dim, nq, ef, k = args.n, args.dim, args.batch, args.ef, args.k
    esize = 4 if args.dtype == "f32" else 1
    n_batches = args.warmup + args.steps
    inflight = auto_inflight(args, args.dtype, args.steps)

    t0 = time.time()
    elements = B.rows(args.data, SEED, 0, n, dim, args.dtype)
    # every rank searches its own batches: rows [rank*n_batches*nq, ...) of the query stream
    queries = B.rows(args.data, SEED + 1, rank * n_batches * nq, n_batches * nq, dim, args.dtype)
    torch.cuda.synchronize()
    t_gen = time.time() - t0
    builder, index, t_build = B.build_index(elements, args.dtype)
    if args.visited_slots:
        index.set_option(B._lib.OPT_VISITED_SLOTS, args.visited_slots)
    order, t_reorder = None, 0.0
    if args.reorder:
        t0 = time.time()
        order = index.reorder()  # order[new id] = old id
        t_reorder = time.time() - t0
    layer_sizes = [builder.layer_len(l) for l in range(builder.num_layers())]
    if rank == 0:
        log("gen %.1fs, gpu build %.1fs, layers %s, index %.2f GB HBM" % (t_gen, t_build, layer_sizes, index.hbm_bytes() / 1e9))

    m = B.measure(index, queries, dim, esize, nq, ef, k, args.steps, args.warmup, inflight, contract=True, steady_s=0.5)
    value = world * args.steps * nq / m["elapsed"]
    wl_key = "%d|%d|%s|%s|nq%d|ef%d|k%d|nn%d|ms%d|re%d" % (n, dim, args.dtype, args.data, nq, ef, k, args.num_neighbors,
                                                        args.build_max_search, args.build_reinsert)
    if args.reorder:
        wl_key += "|reordered"
    out = {
        "metric": "queries/sec (recall@10 alongside), 10M x 100-d angular, batch=1024",
        "value": round(value, 1), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(m["elapsed"] / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "inflight_batches": m["inflight"], "slow_path_queries": m["slow"], "visited_spill_walks": m["spill"],
        "sequential": {"value": round(args.steps * nq / m["seq_elapsed"], 1),
                       "ms_per_step": round(m["seq_elapsed"] / args.steps * 1e3, 4),
                       "note": "same K steps, one stream, one batch at a time (rank-local)"},
        "steady": m["steady"],
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {
            "workload": workload_label(n, dim, args.dtype, args.data, nq, ef, k),
            "n_elements": n, "dim": dim, "batch": nq, "ef_search": ef, "k": k, "layers": layer_sizes,
            "graph": {"builder": "gpu-batched", "num_neighbors": args.num_neighbors,
                      "max_search": args.build_max_search, "reinsert": bool(args.build_reinsert),
                      "layer_multiplier": 15.0, "batch_max": args.batch_max, "build_s": round(t_build, 1),
                      "reordered": bool(args.reorder), "reorder_s": round(t_reorder, 2)},
            "parallelism": "replica x%d (one process per GPU, no data-path collective); %d batches in flight per GPU, "
                           "GPU_MAX_HW_QUEUES=%s" % (world, m["inflight"], os.environ.get("GPU_MAX_HW_QUEUES")),
        },
        "roofline": B.roofline(m, wl_key, value / world, nq),
        "kernel_sources_sha": csrc_sha(),
    }

    # ---- N > 1: the partitioned exchange over RCCL, as a sub-record (all ranks take part) ---------------
    if (world > 1 or args.force_partitioned) and not args.no_partitioned:
        out["partitioned"] = partitioned_record(B, args, n, dim, args.dtype, nq, ef, k, args.steps, args.warmup, 1,
                                                seed_base=SEED + 100)

    # ---- rank 0, N = 1: recall, sweep, CPU baseline, sub-records --------------------------------------
    if rank == 0:
        b0 = args.warmup
        gt = None
        if not args.no_recall:
            bf = {}
            gt = B.ground_truth(index, queries[b0 * nq:(b0 + 1) * nq], k, args.dtype, timing=bf)
            out["brute_force"] = bf
            got = m["ids"][b0]
            if order is not None:  # ground truth is in build ids, results in reordered ids
                got = torch.from_numpy(order.astype(np.int64)).cuda()[got.clamp_min(0)]
            out["recall_at_10"] = round(B.recall(gt, got, k), 4)
            efs = [int(x) for x in args.sweep_ef.split(",") if x]
            if efs and order is None:
                out["ef_sweep"] = B.ef_sweep(index, queries, gt, nq, k, efs, args.steps, args.warmup, inflight, stop_at=0.95)
        if world == 1 and args.cpu_batches > 0:
            nb = min(args.cpu_batches, args.steps)
            h_q = queries[b0 * nq:(b0 + nb) * nq].cpu().numpy()
            g_ids = m["ids"][b0:b0 + nb].reshape(-1, k).cpu().numpy().astype(np.uint64)
            g_d = m["dists"][b0:b0 + nb].reshape(-1, k).cpu().numpy()
            oix = B.host_index(elements, builder, order)
            out["cpu_baseline"] = B.cpu_baseline(oix, h_q, ef, k, g_ids, g_d)
            out["speedup_vs_cpu"] = round(value / out["cpu_baseline"]["value"], 2)
            if gt is not None and order is None and out.get("brute_force"):
                B.check_scan(oix, h_q[:nq], gt, k, out["brute_force"])
            del oix
        if world == 1 and not args.no_extras:
            out["latency_nq1"] = B.latency_nq1(index, queries, dim, ef, k)
            if order is None:
                out["launch_scaling"] = B.launch_scaling(index, args.data, dim, args.dtype, ef, k)
        del m
        if world == 1 and not args.no_extras and args.dtype == "f32" and args.data == "uniform" and order is None:
            del index, builder, elements, queries
            torch.cuda.empty_cache()
            out["int8"] = sub_record(B, args, "i8", "uniform", n, dim, nq, args.ef, args.steps, args.warmup,
                                     cpu_batches=args.cpu_batches, scaling=True)
            out["secondary"] = sub_record(B, args, "f32", "latent", n, dim, nq, args.ef, args.steps, args.warmup,
                                          cpu_batches=args.cpu_batches, find_ef=True)
            if args.c4_elements:
                out["c4_shard"] = sub_record(B, args, "f32", "uniform", args.c4_elements, 200, 4096, 50, 10, 3, cpu_batches=1,
                                             recall_queries=1024)
            if args.c5_elements:
                out["c5_shard"] = sub_record(B, args, "i8", "uniform", args.c5_elements, 100, 4096, 200, 10, 2, cpu_batches=1,
                                             recall_queries=1024)
    return out


def sub_record(B, args, dtype, data, n, dim, nq, ef, steps, warmup, cpu_batches=1, scaling=False, find_ef=False,
               recall_queries=None):
    """The same measurement on another element type / data distribution / shape, as a sub-record of the line."""
    torch = B.torch
    k = args.k
    esize = 4 if dtype == "f32" else 1
    t_sub = time.time()
    elements = B.rows(data, SEED, 0, n, dim, dtype)
    queries = B.rows(data, SEED + 1, 0, (warmup + steps) * nq, dim, dtype)
    builder, index, t_build = B.build_index(elements, dtype)
    rq = min(nq, recall_queries or nq)
    bf = {}
    gt = B.ground_truth(index, queries[warmup * nq:warmup * nq + rq], k, dtype, timing=bf)
    gt_dists = B._gt_dists
    inflight = auto_inflight(args, dtype, steps)
    layer_sizes = [builder.layer_len(l) for l in range(builder.num_layers())]
    rec = {"workload": workload_label(n, dim, dtype, data, nq, ef, k), "dtype": dtype, "data": "synthetic",
           "n_elements": n, "dim": dim, "layers": layer_sizes, "build_s": round(t_build, 1),
           "index_hbm_gb": round(index.hbm_bytes() / 1e9, 2), "brute_force": bf}
    if find_ef:
        sweep = B.ef_sweep(index, queries, gt, nq, k, [20, 30, 50, 70, 100, 140, 200, 300, 400, 600, 800], steps, warmup,
                           inflight, stop_at=0.95)
        rec["ef_sweep"] = sweep
        ok = [s for s in sweep if s["recall_at_10"] >= 0.95]
        ef = ok[0]["ef"] if ok else sweep[-1]["ef"]
        rec["smallest_ef_with_recall_0.95"] = ef if ok else None
        rec["workload"] = workload_label(n, dim, dtype, data, nq, ef, k)
    m = B.measure(index, queries, dim, esize, nq, ef, k, steps, warmup, inflight, steady_s=0.3)
    wl_key = "%d|%d|%s|%s|nq%d|ef%d|k%d|nn%d|ms%d|re%d" % (n, dim, dtype, data, nq, ef, k, args.num_neighbors,
                                                        args.build_max_search, args.build_reinsert)
    rec.update({
        "value": round(m["value_local"], 1), "unit": "queries/s", "ef_search": ef, "batch": nq, "k": k, "steps": steps,
        "inflight_batches": m["inflight"], "ms_per_step": round(m["elapsed"] / steps * 1e3, 4),
        "sequential": {"value": round(steps * nq / m["seq_elapsed"], 1), "ms_per_step": round(m["seq_elapsed"] / steps * 1e3, 4)},
        "steady": m["steady"],
        "slow_path_queries": m["slow"], "visited_spill_walks": m["spill"],
        "recall_at_10": round(B.recall(gt, m["ids"][warmup][:rq], k), 4),
        "roofline": B.roofline(m, wl_key, m["value_local"], nq),
    })
    if scaling:
        rec["launch_scaling"] = B.launch_scaling(index, data, dim, dtype, ef, k)
    if cpu_batches > 0:
        nb = min(cpu_batches, steps)  # a bounded sample, repeated until cpu_seconds have passed
        h_q = queries[warmup * nq:(warmup + nb) * nq].cpu().numpy()
        g_ids = m["ids"][warmup:warmup + nb].reshape(-1, k).cpu().numpy().astype(np.uint64)
        g_d = m["dists"][warmup:warmup + nb].reshape(-1, k).cpu().numpy()
        oix = B.host_index(elements, builder)
        rec["cpu_baseline"] = B.cpu_baseline(oix, h_q, ef, k, g_ids, g_d, single_thread_queries=128)
        rec["speedup_vs_cpu"] = round(rec["value"] / rec["cpu_baseline"]["value"], 2)
        if n <= 20_000_000:  # (the oracle's scan of 125M rows is not worth its minute)
            B._gt_dists = gt_dists
            B.check_scan(oix, h_q[:rq], gt, k, bf, n_check=16)
        del oix
    del m, index, builder, elements, queries
    torch.cuda.empty_cache()
    rec["wall_s"] = round(time.time() - t_sub, 1)
    return rec


def partitioned_record(B, args, n, dim, dtype, nq, ef, k, steps, warmup, spg, seed_base, depth=2, with_cpu=True):
    """The element set split into world * spg id ranges, one independent index per range
    (src/elements/embeddings/parsing.rs:63-100). A step = one batch through every shard's search + ONE all-gather of the
    packed per-shard top-k + the merge kernel; steps are pipelined `depth` deep (granne_amd/sharded.py). Collective:
    every rank calls this with the same arguments. Returns the record (the same on every rank up to rank-local timings)."""
    torch, dist = B.torch, B.dist
    from granne_amd import sharded
    world, rank = B.world, B.rank
    G = world * spg
    esize = 4 if dtype == "f32" else 1
    n_batches = warmup + steps
    bounds = sharded.shard_bounds(n, G)
    offsets = [b[0] for b in bounds]
    mine = list(range(rank * spg, (rank + 1) * spg))
    t0 = time.time()
    # per-shard seed, clear of the query stream's (SEED + 1): shard g is rows 0.. of stream seed_base + g
    elements = [B.rows(args.data, seed_base + g, 0, bounds[g][1] - bounds[g][0], dim, dtype) for g in mine]
    queries = B.rows(args.data, SEED + 1, 0, n_batches * nq, dim, dtype)  # the SAME batches on every rank
    torch.cuda.synchronize()
    t_gen = time.time() - t0
    builders, indexes, t_build = [], [], 0.0
    for e in elements:
        b, ix, tb = B.build_index(e, dtype)
        builders.append(b)
        indexes.append(ix)
        t_build += tb
    layer_sizes = [builders[0].layer_len(l) for l in range(builders[0].num_layers())]
    if rank == 0:
        log("partitioned: gen %.1fs, gpu build %.1fs (%d local shards), shard layers %s" % (t_gen, t_build, spg, layer_sizes))
    sg = sharded.ShardedGranne(indexes, offsets)
    batches = [queries[b * nq:(b + 1) * nq] for b in range(n_batches)]

    sg.search_batches(batches[:max(warmup, depth)], ef, k, depth=depth)
    B.barrier()
    t0 = time.perf_counter()
    res = sg.search_batches(batches[warmup:], ef, k, depth=depth, check_status=False)
    B.barrier()
    elapsed = time.perf_counter() - t0
    if B.use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if bool(sg.status_of_last_batch(0)[:, 0].any().item()):
        raise RuntimeError("exact-search scratch exhausted during the timed steps")
    value = steps * nq / elapsed  # every rank answers the same queries: the job's rate, not a sum over ranks
    out_ids = torch.stack([r[0] for r in res])
    out_d = torch.stack([r[1] for r in res])

    # the same steps strictly one batch at a time, and its phases (HIP events; synchronised, so not the pipelined rate)
    B.barrier()
    t0 = time.perf_counter()
    for b in range(warmup, n_batches):
        sg.search_batch(batches[b], ef, k, check_status=False)
    B.barrier()
    seq_elapsed = time.perf_counter() - t0
    ph = {"search_ms": [], "exchange_ms": [], "merge_ms": []}
    for i in range(min(steps, 10)):
        sg.search_batch(batches[warmup + i], ef, k, check_status=False, timed=True)
        for key in ph:
            ph[key].append(sg.timings[key])
    phases = {key: round(float(np.mean(v)), 4) for key, v in ph.items()}

    # roofline of the dominant kernel: shard 0 of this rank, one launch at a time
    m = B.measure(indexes[0], queries, dim, esize, nq, ef, k, steps, warmup, 1)
    out = {
        "metric": "queries/sec, partitioned index (every rank searches every batch), batch=%d" % nq,
        "value": round(value, 1), "unit": "queries/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": round(elapsed / steps * 1e3, 4), "higher_is_better": True, "scaling": "partitioned",
        "scaling_note": "the element set is split over the ranks and every rank answers the SAME queries: value is the job's "
                        "rate (not a sum over ranks), per-GPU work shrinks as ranks are added",
        "pipeline_depth": depth,
        "sequential": {"value": round(steps * nq / seq_elapsed, 1), "ms_per_step": round(seq_elapsed / steps * 1e3, 4),
                       "note": "one batch at a time: search, all-gather, merge, then the next batch"},
        "dtype": dtype, "data": "synthetic",
        "config": {
            "workload": "%d x %d-d %s angular in %d shards of %d, batch=%d, ef_search=%d, k=%d"
                        % (n, dim, dtype, G, bounds[0][1] - bounds[0][0], nq, ef, k),
            "n_elements": n, "shards": G, "shards_per_gpu": spg, "shard_elements": bounds[0][1] - bounds[0][0], "dim": dim,
            "batch": nq, "ef_search": ef, "k": k, "shard_layers": layer_sizes,
            "graph": {"builder": "gpu-batched", "num_neighbors": args.num_neighbors, "max_search": args.build_max_search,
                      "reinsert": bool(args.build_reinsert), "layer_multiplier": 15.0, "batch_max": args.batch_max,
                      "build_s": round(t_build, 1)},
            "parallelism": "partitioned x%d (%d ranks x %d shards; one all-gather of the packed per-shard top-k + status words "
                           "per batch, then merge_topk_kernel; %d batches pipelined)" % (G, world, spg, depth),
        },
        "exchange": {"collective": "all_gather_into_tensor (RCCL)" if world > 1 else "none (one rank)",
                     "collectives_per_batch": 1 if world > 1 else 0,
                     "bytes_per_rank_per_batch": sg.exchange_bytes_per_rank(nq, k), "ranks": world},
        "phases_ms": phases,
        "roofline": B.roofline(m, "partitioned|%d|%d|%s|nq%d|ef%d" % (bounds[0][1] - bounds[0][0], dim, dtype, nq, ef),
                               m["value_local"], nq),
    }
    out["roofline"]["note"] = "search kernel of ONE shard (this rank's first), one launch at a time"

    # ---- recall against exact brute force over ALL shards --------------------------------------------
    b0 = warmup
    if not args.no_recall:
        q0 = queries[b0 * nq:(b0 + 1) * nq]
        loc_v, loc_i = [], []
        for j, g in enumerate(mine):
            gt = torch.from_numpy(B.ground_truth(indexes[j], q0, k, dtype)).cuda()
            e = elements[j].float()
            if dtype == "i8":
                e = e / e.norm(dim=1, keepdim=True).clamp_min(1e-30)
            sims = (q0.float()[:, None, :] * e[gt]).sum(-1)
            loc_v.append(sims)
            loc_i.append(gt + offsets[g])
        lv, li = torch.cat(loc_v, 1), torch.cat(loc_i, 1)
        if world > 1:
            av = [torch.empty_like(lv) for _ in range(world)]
            ai = [torch.empty_like(li) for _ in range(world)]
            dist.all_gather(av, lv)
            dist.all_gather(ai, li)
            lv, li = torch.cat(av, 1), torch.cat(ai, 1)
        top = lv.topk(k, dim=1).indices
        gt_all = li.gather(1, top).cpu().numpy()
        out["recall_at_10"] = round(B.recall(gt_all, out_ids[0], k), 4)

    # ---- parity: this rank's first shard against the CPU oracle; the merge against the numpy merge ----
    if with_cpu and args.cpu_batches > 0:
        from oracle.merge import merge_topk_numpy, unpack_topk  # the checker, as in replica mode
        q1 = queries[b0 * nq:(b0 + 1) * nq]
        sg.search_batch(q1, ef, k)  # slot 0 now holds this batch's gathered per-shard results
        torch.cuda.synchronize()
        gathered = sg._slots[0].gathered.view(G, -1).cpu().numpy()
        parts = [unpack_topk(gathered[g], nq, k) for g in range(G)]
        w_ids, w_d, w_c = merge_topk_numpy(np.stack([p_[0] for p_ in parts]), np.stack([p_[1] for p_ in parts]),
                                           np.stack([p_[2] for p_ in parts]), offsets, k)
        merge_ok = bool((out_ids[0].cpu().numpy().astype(np.uint64) == w_ids).all()
                        and out_d[0].cpu().numpy().tobytes() == w_d.tobytes())
        oix = B.host_index(elements[0], builders[0])
        mi, md, mc = parts[mine[0]]
        saved = (args.cpu_threads, args.cpu_seconds)
        if world > 1:  # every rank runs this on the same host: share the cores, keep it short
            args.cpu_threads = max(1, (os.cpu_count() or 2) // (2 * world))
            args.cpu_seconds = min(args.cpu_seconds, 0.5)
        cb = B.cpu_baseline(oix, q1.cpu().numpy(), ef, k, mi, md, single_thread_queries=64)
        args.cpu_threads, args.cpu_seconds = saved
        shard_ok = bool(cb["gpu_matches_oracle"]["ids_bit_exact"] and cb["gpu_matches_oracle"]["dists_bit_exact"])
        ok = torch.tensor([int(merge_ok), int(shard_ok)], dtype=torch.int32, device="cuda")
        if B.use_dist:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        shard_rate = cb["value"]
        out["cpu_baseline"] = {
            "value": round(shard_rate / max(1, G), 1), "unit": "queries/s", "cores": cb["cores"], "kind": "port",
            "one_shard": {"value": shard_rate, "single_thread": cb["single_thread"], "parallel_efficiency": cb["parallel_efficiency"]},
            "sample": "search only (the host index view is built outside the clock): " + cb["sample"] + "; measured on ONE shard of "
                      "%d -- a CPU host answering the partitioned index searches all %d shards per query, so the job rate is that "
                      "shard rate / %d" % (bounds[0][1] - bounds[0][0], G, G),
            "gpu_matches_oracle": {"every_rank_first_shard_bit_exact": bool(int(ok[1].item())),
                                   "merged_equals_numpy_merge_of_shard_results": bool(int(ok[0].item())),
                                   "queries_checked": int(nq)},
        }
        del oix
    del m, sg, indexes, builders, elements, queries
    torch.cuda.empty_cache()
    return out


def run_partitioned(B, args):
    """BASELINE.json configs[3]/[4] shape: bench.py --mode partitioned [--shards-per-gpu S]."""
    out = partitioned_record(B, args, args.n, args.dim, args.dtype, args.batch, args.ef, args.k, args.steps, args.warmup,
                             max(1, args.shards_per_gpu), seed_base=SEED + 100)
    out["vs_baseline"] = None
    out["kernel_sources_sha"] = csrc_sha()
    return out


def main():
    args = parse()
    # stdout carries exactly ONE line (the JSON): anything libraries print meanwhile (RCCL's version
    # banner, progress output) is routed to stderr by pointing fd 1 at fd 2 until the very end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    B = Bench(args)
    out = run_partitioned(B, args) if args.mode == "partitioned" else run_replica(B, args)
    if B.rank == 0:
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if B.use_dist:
        B.dist.barrier()
        B.dist.destroy_process_group()


if __name__ == "__main__":
    main()

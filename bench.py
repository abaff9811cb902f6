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
                batch, merge kernel; pipelined two batches deep. Taken AFTER the contract line is printed (a collective
                one rank fails to reach must not cost the run its headline): bench_extras.json and stderr carry it
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

# The timed steps go out through granne_hip_search_batches_device: K batches of 1024 queries = ONE launch of K x 1024
# walkers on ONE stream, default HIP settings (round 3 kept 5-10 batches in flight on as many streams and had to raise
# GPU_MAX_HW_QUEUES for it). `--inflight N` (N > 1) still measures the stream form; it wants more hardware queues than
# streams (HIP maps streams onto 4 by default and streams that share a queue run one after the other), which must be set
# before HIP starts -- only then, never for the default line.
if any(a.startswith("--inflight") for a in sys.argv[1:]):
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
# the CPU baseline's OpenMP team: one thread per core, spread over both sockets (read when libgomp loads)
if int(os.environ.get("WORLD_SIZE", "1")) <= 1:  # (several ranks on one host would all bind to the same cores)
    os.environ.setdefault("OMP_PROC_BIND", "spread")
    os.environ.setdefault("OMP_PLACES", "cores")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# the process's CPU affinity as it was started (libgomp binds the master thread to its place at the first parallel region)
AFFINITY_AT_START = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None

SEED = 0x6772616E6E65  # "granne"; queries use SEED + 1
MIX_CENTERS, MIX_SIGMA = 4096, 0.30  # the 'mixture' generator (Bench.mixture_raw)
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
    ap.add_argument("--data", default="uniform", choices=["uniform", "latent", "mixture"],
                    help="uniform: BASELINE.json's generator; latent / mixture: the secondary workloads' (Bench.rows)")
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--ef", type=int, default=50)
    ap.add_argument("--k", type=int, default=10)
    # graph: BuildConfig::default() of the reference (src/index/mod.rs:220-231)
    ap.add_argument("--build-max-search", type=int, default=200)
    ap.add_argument("--build-reinsert", type=int, default=1)
    ap.add_argument("--num-neighbors", type=int, default=30)
    ap.add_argument("--batch-max", type=int, default=65536)
    ap.add_argument("--batches-per-call", type=int, default=0,
                    help="timed steps handed to the library per call (granne_hip_search_batches_device: one launch of that many "
                         "batches); 0 = all K steps when K <= 32, else the largest divisor of K up to 32; 1 = one batch per call")
    ap.add_argument("--inflight", type=int, default=1,
                    help="> 1: the round-3 form instead -- step i enqueued alone on HIP stream i %% inflight "
                         "(sets GPU_MAX_HW_QUEUES=24 unless the environment has it)")
    ap.add_argument("--driver", default="cabi", choices=["cabi", "torch"],
                    help="partitioned mode: whose pipelined rate is `value` -- cabi = granne_hip_sharded_begin/end_device (what a "
                         "Rust host binds), torch = granne_amd/sharded.py (one process per GPU; the only one with WORLD_SIZE > 1). "
                         "Both are measured when one process holds all shards")
    ap.add_argument("--parity-all-shards", action="store_true",
                    help="partitioned mode: every shard's first timed batch against the CPU oracle (host copy of each shard), "
                         "and the merged result against the numpy merge of the ORACLE's per-shard results")
    ap.add_argument("--profile-run", action="store_true",
                    help="only the warmup + timed + steady launches of the headline workload (for rocprofv3: every launch of "
                         "the walker in the trace is then a timed-shape launch); prints a reduced line")
    ap.add_argument("--cpu-batches", type=int, default=16, help="batches of the CPU baseline sample (0 = skip)")
    ap.add_argument("--cpu-seconds", type=float, default=2.0, help="wall seconds the CPU baseline is timed over (repeats its sample)")
    ap.add_argument("--c5-elements", type=int, default=125_000_000,
                    help="elements of the c5_shard sub-record: one shard of BASELINE's configs[4] (125M x 100-d int8, max_search 200, "
                         "batch 4096; adds ~110 s, 80 of them the shard's build); 0 = skip")
    ap.add_argument("--c4-elements", type=int, default=12_500_000, help="elements of the c4_shard sub-record (0 = skip)")
    ap.add_argument("--no-partitioned", action="store_true", help="WORLD_SIZE > 1: skip the partitioned sub-record")
    ap.add_argument("--partitioned-timeout", type=int, default=240,
                    help="WORLD_SIZE > 1: seconds the partitioned sub-record (taken after the line is printed) may take per rank")
    ap.add_argument("--force-partitioned", action="store_true", help="take the partitioned sub-record with one rank too (tests)")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--sweep-ef", default="50,100,200,400,800,1600,2048,4096,8192", help="ef values of ef_sweep ('' = skip)")
    ap.add_argument("--no-recall", action="store_true")
    ap.add_argument("--full-line", action="store_true", help="print the full record on stdout (default: the compact line; the "
                    "full record goes to bench_extras.json and stderr)")
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


def auto_group(args, steps):
    """Batches handed to the library per call (one launch each, up to 32 batches): all K timed steps when they fit one
    launch; else the largest divisor of K in [16, 32], so that every call of the timed region has the same shape; else 32
    (the last call then carries the remainder)."""
    if args.batches_per_call:
        return max(1, min(args.batches_per_call, steps))
    if steps <= 32:
        return steps
    for d in range(32, 15, -1):
        if steps % d == 0:
            return d
    return 32


def workload_label(n, dim, dtype, data, nq, ef, k, default_graph=True):
    """Names what actually ran. BASELINE.json's configs get their tag only for their exact shape (and the reference's
    default graph: BuildConfig::default(), src/index/mod.rs:220-231)."""
    comp = {"uniform": "i.i.d. uniform components", "latent": "16-d latent cube through a fixed random linear map",
            "mixture": "mixture of %d Gaussians (centers i.i.d. uniform, component j's sigma %.2f / sqrt(1 + j))" % (MIX_CENTERS, MIX_SIGMA)}[data]
    tag = "custom"
    if not default_graph:
        tag = "custom (non-default graph)"
    elif data == "uniform" and dim == 100 and n == 10_000_000 and nq == 1024 and ef == 50:
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
        # GRANNE_BENCH_BACKEND=gloo: the ranks of an N > 1 run SHARE the devices that exist (rank r on device r % count) and
        # talk over gloo -- a LOGIC test of the N-rank control flow on a box with fewer GPUs (rendezvous, barriers, the maximum
        # over ranks, rank 0's extras while the others wait, the line); RCCL refuses two ranks on one device. Never a measurement:
        # the line says so (`shared_devices`).
        self.backend = os.environ.get("GRANNE_BENCH_BACKEND", "nccl")
        self.shared_devices = self.backend != "nccl" and self.world > torch.cuda.device_count()
        if self.backend != "nccl":
            self.local_rank %= max(1, torch.cuda.device_count())
        if self.use_dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            os.environ.setdefault("RANK", str(self.rank))
            os.environ.setdefault("WORLD_SIZE", str(self.world))
            torch.cuda.set_device(self.local_rank)
            if self.backend == "nccl":
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", self.local_rank))
            else:
                dist.init_process_group(backend=self.backend)
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
        if data == "mixture":
            return self.prepare(self.mixture_raw(seed, row0, rows, dim), dtype)
        LATENT = 16
        proj = self.synth_raw(SEED + 7, 0, LATENT, dim)
        out = self.torch.empty((rows, dim), dtype=self.torch.float32, device="cuda")
        step = 2_000_000
        for r0 in range(0, rows, step):
            r1 = min(rows, r0 + step)
            z = self.synth_raw(seed, row0 + r0, r1 - r0, LATENT)
            self.torch.matmul(z, proj, out=out[r0:r1])
        return self.prepare(out, dtype)

    def mixture_raw(self, seed, row0, rows, dim):
        """A dim-d mixture of MIX_CENTERS Gaussians, every quantity from the counter-based uniform generator (so any rows of
        the stream can be regenerated anywhere): centers = rows of that generator (i.i.d. uniform components, the
        reference's src/test_helper.rs:3-6); row r belongs to center floor(u_r * MIX_CENTERS); its offset from the center
        is Gaussian (Box-Muller over two uniform streams) with a decaying spectrum, component j's sigma = MIX_SIGMA /
        sqrt(1 + j) -- the power-law spectrum real embeddings show (effective dimension of a cluster ~ 16), where an
        isotropic 100-d cloud would again be the uniform case within each cluster."""
        torch = self.torch
        centers = self.synth_raw(SEED + 9, 0, MIX_CENTERS, dim)
        scale = (MIX_SIGMA / torch.sqrt(1.0 + torch.arange(dim, dtype=torch.float32, device="cuda")))[None, :]
        out = torch.empty((rows, dim), dtype=torch.float32, device="cuda")
        step = 2_000_000
        for r0 in range(0, rows, step):
            r1 = min(rows, r0 + step)
            u1 = (self.synth_raw(seed + 0x1000, row0 + r0, r1 - r0, dim) + 0.5).clamp_(1e-7, 1.0)
            u2 = self.synth_raw(seed + 0x2000, row0 + r0, r1 - r0, dim) + 0.5
            which = ((self.synth_raw(seed + 0x3000, row0 + r0, r1 - r0, 1)[:, 0] + 0.5) * MIX_CENTERS).long().clamp_(0, MIX_CENTERS - 1)
            z = torch.sqrt(-2.0 * torch.log(u1)) * torch.cos(2.0 * math.pi * u2)
            out[r0:r1] = centers[which] + z * scale
            del u1, u2, z, which
        return out

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
    def measure(self, index, queries, dim, esize, nq, ef, k, steps, warmup, group, inflight=1, contract=False, steady_s=0.0,
                profile_only=False):
        """The K timed steps (K batches of nq fresh queries) handed to the library `group` batches per call
        (granne_hip_search_batches_device: one launch per call) on torch's current stream; or, inflight > 1, one batch per
        call on stream i % inflight (round 3's form). Then, untimed: the grouped launches again between HIP events (the
        dominant kernel's duration), the same steps one batch per call (`sequential`, with the library's own events
        around every launch), and a counting pass with an exact visited set (the reference's n_dist)."""
        torch = self.torch
        from granne_amd.index import pointer_array
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

        def call_args(b0, nb):  # the pointer arrays of one grouped call, built outside the clock
            sl = slice(b0, b0 + nb)
            return tuple(pointer_array(v[sl]) for v in (q_ptr, i_ptr, d_ptr, c_ptr, s_ptr))

        def grouped(a, on):
            index.search_batches_device(a[0], nq, ef, k, a[1], a[2], a[3], a[4], st_ptr, on)

        group = max(1, min(group, steps))
        inflight = max(1, min(inflight, len(self.streams)))
        streams = self.streams[:inflight] if inflight > 1 else [torch.cuda.current_stream()]
        on_ = [x.cuda_stream for x in streams]
        timed_calls = [call_args(warmup + g0, min(group, steps - g0)) for g0 in range(0, steps, group)]
        if inflight > 1:
            for b in range(max(warmup, inflight)):  # (every stream has searched once: its scratch block exists)
                step(b % n_batches, on_[b % inflight])
        else:
            for g0 in range(0, warmup, group):  # the W warmup steps, in calls of the timed shape
                grouped(call_args(g0, min(group, warmup - g0)), self.stream)
            grouped(timed_calls[0], self.stream)  # (the scratch block has the timed shape's size: no allocation under the clock)
        torch.cuda.synchronize()
        status.zero_()
        if contract:
            self.barrier()
        else:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        if inflight > 1:
            for i in range(steps):
                step(warmup + i, on_[i % inflight])
        else:
            for a in timed_calls:
                grouped(a, self.stream)
        if contract:
            self.barrier()
        else:
            torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if contract and self.use_dist:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if self.backend == "nccl" else "cpu")
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)  # MAX over ranks
            elapsed = float(t.item())
        if int(status[0].item()) != 0:
            raise RuntimeError("exact-search scratch exhausted during the timed steps")

        # the same K steps again and again, back to back, until `steady_s` seconds have passed: the K-step window above
        # is a few milliseconds, this one is long enough to trust (same batches, same calls, one synchronisation)
        steady = None
        if steady_s > 0:
            rounds = max(2, int(math.ceil(steady_s / max(elapsed, 1e-6))))
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            for r in range(rounds):
                if inflight > 1:
                    for i in range(steps):
                        step(warmup + i, on_[(r * steps + i) % inflight])
                else:
                    for a in timed_calls:
                        grouped(a, self.stream)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t2
            steady = {"value": round(rounds * steps * nq / dt, 1), "unit": "queries/s", "steps": rounds * steps,
                      "seconds": round(dt, 3), "note": "the same K steps repeated back to back (rank-local)"}

        # the timed shape's launches between HIP events on the launch stream (a grouped call is ONE kernel and nothing else:
        # walk_fast.h / slow_kernel.h), one pair per call, each call after the previous one has drained
        grp_ms = []
        if inflight == 1:
            for rep in range(3):
                for a in timed_calls:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    grouped(a, self.stream)
                    e1.record()
                    torch.cuda.synchronize()
                    grp_ms.append(e0.elapsed_time(e1))
        slow_n, spill_n = int(status[1].item()), int(status[2].item())  # (of the timed forms of the walk, not of the counting pass)
        if profile_only:
            return {"elapsed": elapsed, "value_local": steps * nq / elapsed, "steady": steady, "group": group,
                    "calls": len(timed_calls), "inflight": inflight, "group_launch_ms": grp_ms, "slow": slow_n, "spill": spill_n}

        # the same K steps strictly one after the other, ONE batch per call, twice: first bare (the wall clock of K
        # back-to-back launches: what one batch at a time costs, launch gaps included), then with one pair of HIP events
        # per step recorded inside the library immediately around the walker's dispatch = what rocprofv3 reports per
        # kernel. The events themselves are work on the stream, which is why the two loops are separate.
        glib, _glib = self.lib, self._lib

        def hip_event():
            e = C.c_void_p()
            _glib.check(glib.granne_hip_event_create(C.byref(e)))
            return e

        step(0, self.stream)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(steps):
            step(warmup + i, self.stream)
        torch.cuda.synchronize()
        seq_elapsed = time.perf_counter() - t1
        # one batch per call again, with up to GRANNE_HIP_SEARCH_DEPTH calls in flight beside the ONE caller stream
        # (granne_hip_search_begin_device / _end_device: the index's own streams; default hardware queues)
        def begin_end_run(depth_):
            index.set_option(_glib.OPT_SEARCH_DEPTH, depth_)

            def begin_end(first, count):
                tickets = []
                for i in range(count):
                    b = first + i
                    tickets.append(index.search_begin_device(q_ptr[b], nq, ef, k, i_ptr[b], d_ptr[b], c_ptr[b], s_ptr[b], st_ptr, self.stream))
                    if i >= depth_ - 1:
                        index.search_end_device(tickets[i - depth_ + 1], self.stream)
                for tk in tickets[max(0, count - depth_ + 1):]:
                    index.search_end_device(tk, self.stream)

            begin_end(0, min(n_batches, 2 * depth_))  # (the index's streams have searched once: their scratch blocks exist)
            torch.cuda.synchronize()
            t1_ = time.perf_counter()
            begin_end(warmup, steps)
            torch.cuda.synchronize()
            return time.perf_counter() - t1_

        be_elapsed = begin_end_run(_glib.SEARCH_DEPTH)  # the default depth (3: with the caller's stream, HIP's four hardware queues)
        # GRANNE_HIP_OPT_SEARCH_DEPTH = 8: short walks (int8) need more batches in flight to fill the chip (pays fully with
        # GPU_MAX_HW_QUEUES above HIP's default of 4, which this process does not touch)
        be8_elapsed = begin_end_run(8)
        index.set_option(_glib.OPT_SEARCH_DEPTH, _glib.SEARCH_DEPTH)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        kev = [(hip_event(), hip_event()) for _ in range(steps)]
        for i in range(steps):
            b = warmup + i
            ev[i][0].record()
            index.search_batch_device_timed(q_ptr[b], nq, ef, k, i_ptr[b], d_ptr[b], c_ptr[b], s_ptr[b], st_ptr, self.stream,
                                            kev[i][0].value, kev[i][1].value)
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
                index.search_batch_device(q_ptr[b], nq, ef, k, ids_x.data_ptr(), dists_x.data_ptr(), counts_x.data_ptr(),
                                          stats_x[i].data_ptr(), st_ptr, self.stream)
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
        # (graphs of 64-id layers and lists beyond 1024 keys walk without a visited set whatever the option asks for: their
        #  n_dist counts evaluated rows, a few percent above the reference's distinct nodes -- said in the record)
        counted_exactly = index.get_option(_glib.OPT_LAST_WALKER) != _glib.WALKER_REGISTER_WIDE and ef <= 1024
        alg_total = st[0] * dim * esize + st[2] * 4 + steps * nq * (dim * esize + k * 8)
        alg_per_batch = alg_total / steps
        mean_ms = float(np.mean(step_ms))
        # the timed shape: a launch carries `group` batches (the last call of the region may carry fewer: weight by batches)
        if grp_ms:
            per_call_batches = [min(group, steps - g0) for g0 in range(0, steps, group)] * 3
            grp_total_ms, grp_batches = float(np.sum(grp_ms)), float(np.sum(per_call_batches))
            launch_ms = grp_total_ms / len(grp_ms)
            alg_per_launch = alg_per_batch * grp_batches / len(grp_ms)
        else:  # the stream form launches one batch at a time
            launch_ms, alg_per_launch = mean_ms, alg_per_batch
        return {
            "elapsed": elapsed, "value_local": steps * nq / elapsed, "seq_elapsed": seq_elapsed, "steady": steady,
            "begin_end": {"value": round(steps * nq / be_elapsed, 1), "ms_per_step": round(be_elapsed / steps * 1e3, 4),
                          "depth": _glib.SEARCH_DEPTH,
                          "depth8": {"value": round(steps * nq / be8_elapsed, 1), "ms_per_step": round(be8_elapsed / steps * 1e3, 4),
                                     "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES", "unset (HIP default: 4)")},
                          "note": "one batch per call, granne_hip_search_begin_device / _end_device: up to `depth` calls in flight "
                                  "beside one caller stream (rank-local); depth8 = GRANNE_HIP_OPT_SEARCH_DEPTH 8"},
            "ids": ids, "dists": dists, "counts": counts, "status": status, "inflight": inflight, "group": group,
            "calls": len(timed_calls), "slow": slow_n, "spill": spill_n,
            "alg_per_launch": alg_per_launch, "achieved": alg_per_launch / (launch_ms * 1e-3) / 1e9, "launch_ms_mean": launch_ms,
            "launch_ms_min": float(np.min(grp_ms)) if grp_ms else float(np.min(step_ms)),
            "one": {"alg_per_launch": alg_per_batch, "achieved": alg_per_batch / (mean_ms * 1e-3) / 1e9, "launch_ms_mean": mean_ms,
                    "launch_ms_min": float(np.min(step_ms)), "call_ms_mean": float(np.mean(call_ms))},
            "per_query": {"n_dist": round(st[0] / (steps * nq), 1), "n_expand": round(st[1] / (steps * nq), 1),
                          "n_adj": round(st[2] / (steps * nq), 1), "rows_evaluated": round(evaluated / (steps * nq), 1)},
            "same_as_exact_set_walk": {"queries": steps * nq, "ids_dists_counts_bit_exact": True,
                                       "n_dist_is_the_references_count": bool(counted_exactly)},
        }

    def roofline(self, m, traffic_key, value_per_gpu, nq):
        """HBM roofline of the dominant kernel in the shape the timed region launches it: `group` batches per launch.
        achieved = the reference algorithm's bytes for the launch's queries / the launch's mean duration (HIP events on the
        launch stream). `one_batch_per_launch` is the same kernel launched with a single batch (round 3's headline shape)."""
        traffic, note = None, None
        key = traffic_key + ("|g%d" % m["group"] if m["inflight"] == 1 else "")
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                ent = json.load(f).get(key)
            if ent:
                if ent.get("csrc_sha") == csrc_sha():
                    traffic = ent.get("hbm_bytes_per_launch")
                else:
                    note = "PMC traffic in profiles/pmc_traffic.json was measured on other kernel sources (%s): not quoted" % ent.get("csrc_sha")
        except Exception:
            pass
        one = m["one"]
        r = {
            "bound": "hbm", "kernel": "fast_kernel (walk_fast.h)", "achieved": round(m["achieved"], 1), "peak": HBM_PEAK_GBPS,
            "unit": "GB/s", "frac": round(m["achieved"] / HBM_PEAK_GBPS, 4), "traffic": traffic,
            "launch": {"batches": m["group"] if m["inflight"] == 1 else 1,
                       "queries": (m["group"] if m["inflight"] == 1 else 1) * nq, "ms_mean": round(m["launch_ms_mean"], 4),
                       "ms_min": round(m["launch_ms_min"], 4), "alg_bytes": int(m["alg_per_launch"])},
            "alg_bytes_per_launch": int(m["alg_per_launch"]), "launch_ms_mean": round(m["launch_ms_mean"], 4),
            "whole_timed_region_achieved": round(one["alg_per_launch"] * (value_per_gpu / nq) / 1e9, 1),
            "whole_timed_region_frac": round(one["alg_per_launch"] * (value_per_gpu / nq) / 1e9 / HBM_PEAK_GBPS, 4),
            "one_batch_per_launch": {"achieved": round(one["achieved"], 1), "frac": round(one["achieved"] / HBM_PEAK_GBPS, 4),
                                     "alg_bytes_per_launch": int(one["alg_per_launch"]),
                                     "launch_ms_mean": round(one["launch_ms_mean"], 4), "launch_ms_min": round(one["launch_ms_min"], 4),
                                     "call_ms_mean": round(one["call_ms_mean"], 4)},
            "per_query": m["per_query"], "same_as_exact_set_walk": m["same_as_exact_set_walk"],
            "traffic_key": key,
        }
        if traffic:
            r["traffic_over_algorithmic"] = round(traffic / max(1.0, m["alg_per_launch"]), 3)
            r["traffic_note"] = ("HBM bytes per launch from the rocprofv3 PMC passes over this shape (profiles/pmc_traffic.json, taken on "
                                 "the same kernel sources %s; FETCH_SIZE corrected as MI355X_MICROARCH.md prescribes)" % csrc_sha())
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
            if dtype == "f32" and os.environ.get("GRANNE_HIP_BF_B16", "1") != "0":
                kpad = 112 if dim <= 112 else 208 if dim <= 208 else 256 if dim <= 256 else (dim + 127) // 128 * 128
                flops = 3.0 * 2.0 * nq * n * kpad  # what the matrix cores execute: three bf16 instructions per product, K padded
                timing.update({"kernel": ("bf_b16_kernel" if dim <= 256 else "bf_b16_chunked_kernel (the vector in chunks of 128 components)") +
                                         " (f32 rows as two bf16 pieces each: 3 x v_mfma_f32_32x32x16_bf16 per 16 components) "
                                         "+ merge + exact re-ranking", "ms": round(ms, 3),
                               "value": round(nq / (ms * 1e-3), 1), "value_unit": "queries/s at recall 1.0 (exact scan)",
                               "bound": "mfma", "achieved": round(flops / ms / 1e9, 1), "peak": 2500.0, "unit": "TFLOP/s",
                               "frac": round(flops / ms / 1e9 / 2500.0, 4), "queries": nq, "elements": n, "dim": dim,
                               "useful_f32_equivalent_tflops": round(2.0 * nq * n * dim / ms / 1e9, 1),
                               "note": "3 * 2 * nq * n * K flops the matrix cores execute (K = dim padded to %d) / wall of the whole "
                                       "operator (HIP events); peak = dense bf16 MFMA (MI355X_MICROARCH.md); round 5 scored on the f32 "
                                       "matrix path: 17.2 ms = 0.76 of ITS 157.3 TFLOP/s peak. Measured beside it (tools/mfma_lds_loop_b16.hip): "
                                       "the scan's inner loop alone takes 3.75-5.1 ms per 1024 x 10M x 112 -- the chip halves its shader "
                                       "clock under dense matrix load and sustains 1.3-1.8 PFLOP/s on this loop, not the peak's 2.5" % kpad})
            elif dtype == "f32":
                flops = 2.0 * nq * n * dim
                timing.update({"kernel": "bf_f32_kernel (v_mfma_f32_32x32x2_f32) + merge + exact re-ranking", "ms": round(ms, 3),
                               "value": round(nq / (ms * 1e-3), 1), "value_unit": "queries/s at recall 1.0 (exact scan)",
                               "bound": "mfma", "achieved": round(flops / ms / 1e9, 1), "peak": 157.3, "unit": "TFLOP/s",
                               "frac": round(flops / ms / 1e9 / 157.3, 4), "queries": nq, "elements": n, "dim": dim,
                               "note": "2 * nq * n * dim flops / wall of the whole operator (HIP events); peak = dense f32 MFMA"})
            else:
                ring = 64 < index.dim <= 128 and os.environ.get("GRANNE_HIP_BF_RING", "1") != "0"  # 128-byte rows: bf_i8_ring_kernel
                rowb = 128 if index.dim <= 128 else 256 if index.dim <= 256 else 512 if index.dim <= 512 else (index.dim + 1023) // 1024 * 1024
                tiles = (nq + 511) // 512 if ring else (nq + 255) // 256
                byts = float(n) * rowb  # the rows ONCE: the query tiles of a range run side by side and share them in L2
                ops = 2.0 * nq * n * rowb  # integer multiply-adds the matrix cores execute (rows zero padded to whole 128-byte blocks)
                timing.update({"kernel": ("bf_i8_ring_kernel (tiles by LDS-DMA into a ring, v_mfma_i32_32x32x32_i8, 64 queries per wave)"
                                          if ring else "bf_i8_chunked_kernel (rows in chunks of 128 bytes, v_mfma_i32_32x32x32_i8)" if index.dim > 128
                                          else "bf_i8_kernel (v_mfma_i32_32x32x32_i8)") + " + merge + exact re-ranking", "ms": round(ms, 3),
                               "value": round(nq / (ms * 1e-3), 1), "value_unit": "queries/s at recall 1.0 (exact scan)",
                               "bound": "mfma", "achieved": round(ops / ms / 1e9, 1), "peak": 3944.0, "unit": "TOP/s",
                               "frac": round(ops / ms / 1e9 / 3944.0, 4), "queries": nq, "elements": n, "dim": dim,
                               "hbm": {"achieved": round(byts / ms / 1e6, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                       "frac": round(byts / ms / 1e6 / HBM_PEAK_GBPS, 4)},
                               "note": "2 * nq * n * row bytes operations / wall of the whole operator (HIP events); peak = the guide's dense int8 "
                                       "figure (MI355X_MICROARCH.md). Measured beside it (tools/mfma_lds_loop.hip): the scan's inner loop "
                                       "alone on random int8 data sustains 2.7e15 operations/s -- the shader clock drops to 0.9-1.1 GHz "
                                       "under it -- so ~0.95 ms is this scan's floor at 1024 x 10M x 128 bytes whatever its structure; hbm = the n "
                                       "rows once (%d query tiles share them in L2)" % tiles})
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

    def ef_sweep(self, index, queries, gt, nq, k, efs, steps, warmup, group, stop_at=None):
        """recall@10 (first timed batch) and queries/sec over windows of >= 50 ms: `group` batches per call (one stream),
        and one batch per call."""
        torch = self.torch
        from granne_amd.index import pointer_array
        out = []
        n_b = queries.shape[0] // nq
        nd = n_b - warmup
        g = max(1, min(group, nd))
        o = (torch.empty((g, nq, k), dtype=torch.int64, device="cuda"), torch.empty((g, nq, k), dtype=torch.float32, device="cuda"),
             torch.empty((g, nq), dtype=torch.int32, device="cuda"))
        po = [pointer_array([o[j][b].data_ptr() for b in range(g)]) for j in range(3)]
        starts = list(range(0, nd - g + 1, g)) or [0]
        pq = [pointer_array([queries[(warmup + b0 + b) * nq:(warmup + b0 + b + 1) * nq].data_ptr() for b in range(g)]) for b0 in starts]
        for e_ in efs:
            def run(b, on):
                index.search_batch_device(queries[b * nq:(b + 1) * nq].data_ptr(), nq, e_, k, o[0][0].data_ptr(), o[1][0].data_ptr(),
                                          o[2][0].data_ptr(), 0, 0, on)
            run(warmup, self.stream)
            torch.cuda.synchronize()
            rec = self.recall(gt, o[0][0], k)
            slow0 = index.last_slow_count()
            r_seq, reps = self.timed_window(lambda j: run(warmup + j, self.stream), nd)

            def grp(j):  # outputs overwrite each other: timing only
                index.search_batches_device(pq[j % len(pq)], nq, e_, k, po[0], po[1], po[2], None, 0, self.stream)
            r_grp, reps_g = self.timed_window(grp, len(pq))
            out.append({"ef": e_, "recall_at_10": round(rec, 4), "qps": round(r_grp * g * nq, 1),
                        "qps_one_batch_at_a_time": round(r_seq * nq, 1), "batches_per_call": g, "calls_timed": reps_g})
            del slow0
            if stop_at is not None and rec >= stop_at:
                break
        return out

    # ---- the CPU oracle beside it ---------------------------------------------------------------------
    def host_info(self):
        """What bounds a CPU baseline on this box besides the cores' count: cgroup CPU quota, affinity mask, NUMA nodes,
        transparent huge pages (granted or not is read off the mapping after it is filled: host_index)."""
        def rd(path):
            try:
                with open(path) as f:
                    return f.read().strip()
            except Exception:
                return None
        info = {"logical_cpus": os.cpu_count(), "affinity_cpus": AFFINITY_AT_START,
                "cgroup_cpu_max": rd("/sys/fs/cgroup/cpu.max"), "thp_enabled": rd("/sys/kernel/mm/transparent_hugepage/enabled")}
        if info["cgroup_cpu_max"] is None:
            q, per = rd("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), rd("/sys/fs/cgroup/cpu/cpu.cfs_period_us")
            info["cgroup_cpu_max"] = "%s %s (v1)" % (q, per) if q else None
        try:
            info["numa_nodes"] = len([d for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit()])
        except Exception:
            info["numa_nodes"] = None
        try:
            with open("/proc/cpuinfo") as f:
                names = [l.split(":", 1)[1].strip() for l in f if l.startswith("model name")]
            info["cpu_model"] = names[0] if names else None
        except Exception:
            info["cpu_model"] = None
        quota = None
        cm = info["cgroup_cpu_max"]
        if cm and not cm.startswith("max") and not cm.startswith("-1"):
            try:
                a, b = cm.split()[:2]
                quota = float(a) / float(b)
            except Exception:
                quota = None
        info["cgroup_quota_cpus"] = quota
        return info

    def host_index(self, elements, builder, order=None):
        """the oracle's view of the index: host copies of the elements and of the layers. The element rows -- what the
        walks gather at random -- live in a mapping that asks for transparent huge pages (4 GB in 4 KB pages is a TLB miss
        per row) and are FIRST TOUCHED by the OpenMP team that will read them (gro_parallel_copy: static 2 MB chunks over
        the team, so the pages spread over the NUMA nodes the team spans instead of sitting on the node of one copier)."""
        from oracle import oracle as orc
        orc.build()
        n = elements.shape[0]
        np_dt = np.float32 if elements.dtype == self.torch.float32 else np.int8
        count = int(np.prod(elements.shape))
        nbytes = count * np.dtype(np_dt).itemsize
        try:
            import mmap
            buf = mmap.mmap(-1, max(nbytes, 1))
            buf.madvise(mmap.MADV_HUGEPAGE)
            h_el = np.frombuffer(buf, dtype=np_dt, count=count).reshape(tuple(elements.shape))
            self._host_pages = "transparent huge pages requested (MADV_HUGEPAGE)"
        except Exception:
            h_el = np.empty(tuple(elements.shape), np_dt)
            self._host_pages = "default pages"
        rows_per = max(1, (1 << 30) // max(1, elements.shape[1] * np.dtype(np_dt).itemsize))
        team = self.args.cpu_threads or 0
        for r0 in range(0, n, rows_per):
            r1 = min(n, r0 + rows_per)
            tmp = np.ascontiguousarray(elements[r0:r1].cpu().numpy())
            orc.lib().gro_parallel_copy(h_el[r0:r1].ctypes.data_as(C.c_void_p), tmp.ctypes.data_as(C.c_void_p), tmp.nbytes, team)
        try:  # how much of the process's anonymous memory sits in huge pages now (the element mapping dominates it)
            with open("/proc/self/smaps_rollup") as f:
                kb = {l.split(":")[0]: int(l.split()[1]) for l in f if l.startswith(("AnonHugePages", "Anonymous"))}
            self._host_pages += "; AnonHugePages %.1f GB of %.1f GB anonymous" % (kb.get("AnonHugePages", 0) / 1e6, kb.get("Anonymous", 0) / 1e6)
        except Exception:
            pass
        oix = orc.Index(h_el, builder.layers())
        if order is not None:
            oix = oix.reordered(order)
        return oix

    def cpu_baseline(self, oix, h_q, ef, k, g_ids, g_d, single_thread_queries=256, sweep=True):
        """the ONLY use of oracle/ in this file: the CPU baseline + parity check (the index view comes from host_index).
        sweep: thread counts {8, 16, ..., logical CPUs} each timed over cpu_seconds after an untimed pass; the best is the
        baseline and the whole sweep is reported. Without: the count the last sweep chose and one thread per physical core."""
        from oracle import oracle as orc
        a = self.args
        nqs = h_q.shape[0]
        logical = os.cpu_count() or 1
        phys = max(1, logical // 2)
        host = self.host_info()
        quota = host.get("cgroup_quota_cpus")
        if a.cpu_threads:
            cands = [a.cpu_threads]
        elif sweep or not getattr(self, "_cpu_best_threads", None):
            full = (8, 16, 32, 64, 128, 256, phys, logical)
            if quota:  # a container with a CPU quota: threads far beyond it only burn the quota early in every period
                q = max(1, int(round(quota)))
                full = (max(1, q // 2), q, 2 * q, 4 * q)
            cands = sorted({t for t in full if t <= logical})
        else:
            cands = sorted({self._cpu_best_threads})
        best, tried = None, []
        for th in cands:
            # an untimed pass inside the same parallel region precedes every timed run; the number of timed passes comes
            # from a probe of >= 0.3 s (a shorter one ends before a cgroup quota starts throttling the team)
            probe, _, _, _ = oix.search_batch_timed(h_q, ef, k, n_threads=th, repeats=1)
            r2 = max(1, int(math.ceil(0.3 / max(probe, 1e-6))))
            probe2, _, _, _ = oix.search_batch_timed(h_q, ef, k, n_threads=th, repeats=r2)
            reps = max(1, int(math.ceil(a.cpu_seconds / max(probe2 / r2, 1e-6))))
            sec, o_ids, o_d, o_c = oix.search_batch_timed(h_q, ef, k, n_threads=th, repeats=reps)
            rate = reps * nqs / sec
            tried.append({"threads": th, "value": round(rate, 1), "passes": reps, "seconds": round(sec, 2)})
            if best is None or rate > best[0]:
                best = (rate, th, sec, reps, o_ids, o_d)
        rate, threads, cpu_s, reps, o_ids, o_d = best
        if sweep:
            self._cpu_best_threads = threads
        # one thread, on queries it has not just walked (a repeated pass over a few hundred queries runs out of L3)
        m1 = min(single_thread_queries, max(1, nqs // 2))
        oix.search_batch(h_q[:8], ef, k, n_threads=1)
        t1 = time.perf_counter()
        oix.search_batch(h_q[nqs - m1:], ef, k, n_threads=1)
        single = m1 / (time.perf_counter() - t1)
        ids_ok = bool((g_ids == o_ids).all())
        d_ok = g_d.tobytes() == o_d.tobytes()
        usable = min(threads, phys)
        if quota:
            usable = min(usable, quota)
        if host.get("affinity_cpus") and host["affinity_cpus"] < logical:
            usable = min(usable, host["affinity_cpus"])
        return {
            "value": round(rate, 1), "unit": "queries/s", "cores": threads, "kind": "port",
            "single_thread": {"value": round(single, 1), "unit": "queries/s", "queries": int(m1)},
            "parallel_efficiency": round(rate / (usable * single), 3),
            "thread_sweep": tried, "host": host,
            "sample": "%d queries of the timed workload, same index, %d passes = %.2f s wall after one untimed pass "
                      "(gro_search_batch_timed: one OpenMP region, dynamic schedule over queries, threads and their scratch kept "
                      "between passes; OMP_PROC_BIND=%s OMP_PLACES=%s); oracle/granne_oracle.c (C restatement of the reference's "
                      "search; Rust toolchain absent); thread counts tried %s, best reported; elements first-touched by the OpenMP "
                      "team (gro_parallel_copy), %s; parallel_efficiency = value / (usable cores x single_thread), usable = "
                      "min(threads, %d physical cores, cgroup quota, affinity) = %.1f"
                      % (nqs, reps, cpu_s, os.environ.get("OMP_PROC_BIND"), os.environ.get("OMP_PLACES"), cands,
                         getattr(self, "_host_pages", "default pages"), phys, usable),
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
    group = auto_group(args, args.steps)

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

    m = B.measure(index, queries, dim, esize, nq, ef, k, args.steps, args.warmup, group, inflight=args.inflight, contract=True,
                  steady_s=0.5, profile_only=args.profile_run)
    value = world * args.steps * nq / m["elapsed"]
    if args.profile_run:  # rocprofv3 runs: the trace holds launches of the timed shape only
        return {"metric": "queries/sec (profile run: timed launches only)", "value": round(value, 1), "unit": "queries/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(m["elapsed"] / args.steps * 1e3, 4),
                "batches_per_call": m["group"], "calls": m["calls"], "inflight_streams": m["inflight"], "steady": m["steady"],
                "launch_ms_hip_events": [round(x, 4) for x in m["group_launch_ms"]], "kernel_sources_sha": csrc_sha(),
                "config": {"workload": workload_label(n, dim, args.dtype, args.data, nq, ef, k)}}
    wl_key = "%d|%d|%s|%s|nq%d|ef%d|k%d|nn%d|ms%d|re%d" % (n, dim, args.dtype, args.data, nq, ef, k, args.num_neighbors,
                                                        args.build_max_search, args.build_reinsert)
    if args.reorder:
        wl_key += "|reordered"
    out = {
        "metric": "queries/sec (recall@10 alongside), 10M x 100-d angular, batch=1024",
        "value": round(value, 1), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(m["elapsed"] / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "inflight_batches": m["inflight"], "batches_per_call": m["group"] if m["inflight"] == 1 else 1, "library_calls": m["calls"],
        "slow_path_queries": m["slow"], "visited_spill_walks": m["spill"],
        "sequential": {"value": round(args.steps * nq / m["seq_elapsed"], 1),
                       "ms_per_step": round(m["seq_elapsed"] / args.steps * 1e3, 4),
                       "note": "same K steps, one stream, one batch PER CALL (rank-local)"},
        "one_batch_calls_in_flight": m["begin_end"],
        "steady": m["steady"],
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {
            "workload": workload_label(n, dim, args.dtype, args.data, nq, ef, k,
                                       default_graph=(args.num_neighbors == 30 and args.build_max_search == 200 and bool(args.build_reinsert))),
            "n_elements": n, "dim": dim, "batch": nq, "ef_search": ef, "k": k, "layers": layer_sizes,
            "graph": {"builder": "gpu-batched", "num_neighbors": args.num_neighbors,
                      "max_search": args.build_max_search, "reinsert": bool(args.build_reinsert),
                      "layer_multiplier": 15.0, "batch_max": args.batch_max, "build_s": round(t_build, 1),
                      "reordered": bool(args.reorder), "reorder_s": round(t_reorder, 2)},
            "parallelism": ("replica x%d (one process per GPU, no data-path collective); the K timed steps in %d call(s) of "
                            "granne_hip_search_batches_device (%d batches = one launch each) on ONE stream, GPU_MAX_HW_QUEUES=%s"
                            % (world, m["calls"], m["group"], os.environ.get("GPU_MAX_HW_QUEUES", "unset (HIP default)")))
            if m["inflight"] == 1 else
                           ("replica x%d; one batch per call, %d batches in flight on HIP streams of the caller, "
                            "GPU_MAX_HW_QUEUES=%s" % (world, m["inflight"], os.environ.get("GPU_MAX_HW_QUEUES"))),
        },
        "roofline": B.roofline(m, wl_key, value / world, nq),
        "kernel_sources_sha": csrc_sha(),
    }
    if getattr(B, "shared_devices", False):
        out["shared_devices"] = ("LOGIC TEST, not a measurement: %d ranks share %d device(s) over %s (GRANNE_BENCH_BACKEND)"
                                 % (world, torch.cuda.device_count(), B.backend))

    # ---- the partitioned exchange over RCCL as a sub-record: with ONE rank (--force-partitioned) here, in the line; with
    # N > 1 ranks only after the line is out (main, partitioned_after_the_line) -- a collective that one rank fails to
    # reach must not cost the run its headline
    if world == 1 and args.force_partitioned and not args.no_partitioned:
        out["partitioned"] = partitioned_record(B, args, n, dim, args.dtype, nq, ef, k, args.steps, args.warmup, 1,
                                                seed_base=SEED + 100)

    rank0_after_the_timed_region(B, args, out, index, builder, elements, queries, m, order, value, group)
    return out


def rank0_plan(world, args):
    """What rank 0 measures after the timed region. Every N: recall@10 of the timed batch and the CPU baseline (the other
    ranks wait meanwhile, without spinning: main) -- the line of an N-GPU run carries `roofline` AND `cpu_baseline` like the
    one-GPU line. N = 1 only: the max_search sweep, the one-query latency, the other configs as sub-records."""
    return {"recall": not args.no_recall, "ef_sweep": world == 1 and not args.no_recall and bool(args.sweep_ef),
            "cpu_baseline": args.cpu_batches > 0, "cpu_thread_sweep": world == 1,
            "extras": world == 1 and not args.no_extras}


def rank0_after_the_timed_region(B, args, out, index, builder, elements, queries, m, order, value, group):
    torch = B.torch
    world, rank = B.world, B.rank
    n, dim, nq, ef, k = args.n, args.dim, args.batch, args.ef, args.k
    plan = rank0_plan(world, args)
    # ---- rank 0: recall, CPU baseline (every N); sweep, latency, sub-records (N = 1) ----------------------------------
    if rank == 0:
        b0 = args.warmup
        gt = None
        if plan["recall"]:
            bf = {}
            gt = B.ground_truth(index, queries[b0 * nq:(b0 + 1) * nq], k, args.dtype, timing=bf)
            out["brute_force"] = bf
            got = m["ids"][b0]  # (after --reorder both the scan's and the walk's ids are the reordered index's)
            out["recall_at_10"] = round(B.recall(gt, got, k), 4)
            efs = [int(x) for x in args.sweep_ef.split(",") if x]
            if plan["ef_sweep"] and efs and order is None:
                out["ef_sweep"] = B.ef_sweep(index, queries, gt, nq, k, efs, args.steps, args.warmup, group, stop_at=0.95)
        if plan["cpu_baseline"]:
            nb = min(args.cpu_batches, args.steps)
            h_q = queries[b0 * nq:(b0 + nb) * nq].cpu().numpy()
            g_ids = m["ids"][b0:b0 + nb].reshape(-1, k).cpu().numpy().astype(np.uint64)
            g_d = m["dists"][b0:b0 + nb].reshape(-1, k).cpu().numpy()
            oix = B.host_index(elements, builder, order)
            out["cpu_baseline"] = B.cpu_baseline(oix, h_q, ef, k, g_ids, g_d, sweep=plan["cpu_thread_sweep"])
            # (N > 1: rank 0's host cores against the whole job -- the CPU side does not grow with the GPUs)
            out["speedup_vs_cpu"] = round(value / out["cpu_baseline"]["value"], 2)
            if gt is not None and order is None and out.get("brute_force"):
                B.check_scan(oix, h_q[:nq], gt, k, out["brute_force"])
            if plan["extras"] and gt is not None and order is None and args.data == "uniform":
                out["recall_target"] = recall_target_record(B, args, out, oix, h_q[:nq], gt, k)
            del oix
        if plan["extras"]:
            out["latency_nq1"] = B.latency_nq1(index, queries, dim, ef, k)
            if order is None:
                out["launch_scaling"] = B.launch_scaling(index, args.data, dim, args.dtype, ef, k)
        del m
        if plan["extras"] and args.dtype == "f32" and args.data == "uniform" and order is None:
            del index, builder, elements, queries
            torch.cuda.empty_cache()
            def guarded(name, fn, *a, **kw):
                """A sub-record that fails (a parity check that raises, an allocation that does not fit) says so IN the line
                -- {"error": ...}, and on stderr -- and does not take the headline and the other sub-records with it."""
                try:
                    out[name] = fn(*a, **kw)
                except Exception as e:  # noqa: BLE001
                    import traceback
                    log("sub-record %s FAILED: %s" % (name, traceback.format_exc()))
                    out[name] = {"workload": name, "error": _short(repr(e), 300)}
                    torch.cuda.empty_cache()

            guarded("c1", c1_record, B, args)
            guarded("int8", sub_record, B, args, "i8", "uniform", n, dim, nq, args.ef, args.steps, args.warmup,
                    cpu_batches=args.cpu_batches, scaling=True)
            # the north star's bar -- >= 10x the CPU at recall@10 >= 0.95 -- on data a graph index CAN reach 0.95 on: two
            # documented generators (Bench.rows), each at the smallest max_search with recall >= 0.95, CPU beside it
            guarded("secondary", sub_record, B, args, "f32", "latent", n, dim, nq, args.ef, args.steps, args.warmup,
                    cpu_batches=args.cpu_batches, find_ef=True)
            guarded("secondary_mixture", sub_record, B, args, "f32", "mixture", n, dim, nq, args.ef, args.steps, args.warmup,
                    cpu_batches=args.cpu_batches, find_ef=True)
            if args.c4_elements:
                guarded("c4_shard", sub_record, B, args, "f32", "uniform", args.c4_elements, 200, 4096, 50, 10, 3, cpu_batches=1,
                        recall_queries=1024)
            if args.c5_elements:
                guarded("c5_shard", sub_record, B, args, "i8", "uniform", args.c5_elements, 100, 4096, 200, 10, 2, cpu_batches=1,
                        recall_queries=1024)


def recall_target_record(B, args, out, oix, h_q, gt, k):
    """What recall@10 >= 0.95 costs on BASELINE's own data (the north star's bar): the smallest max_search of the sweep
    that reaches it (none does up to 4096 on 10M i.i.d.-uniform 100-d points: intrinsic dimension ~ 100), what the walk
    reaches at the largest one -- on the GPU and, the SAME walk, on the CPU -- and the exact scan of both sides, which is
    then the only answer at that recall. The CPU walk is timed on a bounded sample at the sweep's largest max_search."""
    sweep = out.get("ef_sweep") or []
    ok = [s_ for s_ in sweep if s_["recall_at_10"] >= 0.95]
    rec = {"recall_at_10_target": 0.95,
           "smallest_max_search_reaching_it": ok[0]["ef"] if ok else None,
           "note": ("no max_search up to %d reaches it on this data: at recall >= 0.95 both sides answer with their exact scan"
                    % sweep[-1]["ef"]) if (sweep and not ok) else None}
    if sweep:
        last = ok[0] if ok else sweep[-1]
        nqs = min(64, h_q.shape[0])
        oix.search_batch(h_q[:8], last["ef"], k, n_threads=0)
        t0 = time.perf_counter()
        oi, _, _, _ = oix.search_batch(h_q[:nqs], last["ef"], k, n_threads=0)
        sec = time.perf_counter() - t0
        cpu_recall = float(np.mean([len(set(gt[i]) & set(oi[i].tolist())) / k for i in range(nqs)]))
        rec["walk_at_max_search"] = {"max_search": last["ef"], "recall_at_10": last["recall_at_10"], "gpu_qps": last["qps"],
                                     "cpu_qps": round(nqs / sec, 1), "cpu_recall_at_10_same_walk": round(cpu_recall, 4),
                                     "cpu_sample": "%d queries, all OpenMP threads, %.2f s" % (nqs, sec)}
    bf = out.get("brute_force") or {}
    if bf:
        rec["exact_scan"] = {"recall_at_10": 1.0, "gpu_qps": bf.get("value"),
                             "cpu_qps": (bf.get("cpu_baseline") or {}).get("value"), "speedup_vs_cpu": bf.get("speedup_vs_cpu")}
        if (bf.get("cpu_baseline") or {}).get("value") and bf.get("value"):
            cpu_best = bf["cpu_baseline"]["value"]
            rec["gpu_over_cpu_at_target"] = round(bf["value"] / cpu_best, 1) if not ok else None
    return rec


def c1_record(B, args):
    """BASELINE.json configs[0] = examples/glove.rs:46-60: ~400k x 100-d f32 angular vectors, index built with
    BuildConfig::default().max_search(10), then `index.search(&index.get_element(i), 200, 10)` for i in 0, 134, 5555,
    37000 -- member queries, one per call. GloVe itself is not in this image (no network): 400k rows of BASELINE's
    synthetic generator stand in. The four searches go through granne_hip_search (host pointers, the reference's call
    shape) and are compared with the CPU oracle on the same graph, bit for bit; a member's nearest neighbor must be itself
    at distance max(0, 1 - x.x). Beside it: 1024 member queries in one call against the oracle's rate."""
    torch = B.torch
    n, dim, ef, k = 400_000, 100, 200, 10
    t0 = time.time()
    el = B.rows("uniform", SEED + 50, 0, n, dim, "f32")
    builder = B.ga.GranneBuilder.from_device("angular", el.data_ptr(), n, dim, device=B.dev, stream=B.stream, num_neighbors=30,
                                             max_search=10, reinsert_elements=True, batch_max=args.batch_max, show_progress=False)
    builder.build()
    index = builder.get_index()
    torch.cuda.synchronize()
    t_build = time.time() - t0
    members = [0, 134, 5555, 37000]
    oix = B.host_index(el, builder)
    singles, ok, self_first = [], True, True
    for i in members:
        x = index.get_element(i)
        t1 = time.perf_counter()
        res = index.search(x, ef, k)
        dt = time.perf_counter() - t1
        oi, od, oc, _ = oix.search_batch(x[None], ef, k, n_threads=1)
        want = [(int(oi[0, j]), float(od[0, j])) for j in range(int(oc[0]))]
        ok = ok and [r[0] for r in res] == [w[0] for w in want] and \
            np.array([r[1] for r in res], np.float32).tobytes() == np.array([w[1] for w in want], np.float32).tobytes()
        self_first = self_first and res[0][0] == i and res[0][1] <= 1e-6
        singles.append({"member": i, "first": [res[0][0], res[0][1]], "call_us": round(dt * 1e6, 1)})
    # 1024 members, one call (device-resident), against the oracle on all threads
    mem = torch.arange(0, n, n // 1024, device="cuda")[:1024]
    q = el[mem].contiguous()
    ids = torch.empty((1024, k), dtype=torch.int64, device="cuda")
    ds = torch.empty((1024, k), dtype=torch.float32, device="cuda")
    cnt = torch.empty(1024, dtype=torch.int32, device="cuda")

    def run(_j):
        index.search_batch_device(q.data_ptr(), 1024, ef, k, ids.data_ptr(), ds.data_ptr(), cnt.data_ptr(), 0, 0, B.stream)
    rate, reps = B.timed_window(run, 1)
    h_q = q.cpu().numpy()
    cb = B.cpu_baseline(oix, h_q, ef, k, ids.cpu().numpy().astype(np.uint64), ds.cpu().numpy(), single_thread_queries=128, sweep=False)
    # (a graph built with max_search 10 does not lead every member back to itself: the fraction is reported, the CPU
    #  oracle on the same graph returns the same ids either way -- checked below)
    self_frac = float((ids[:, 0] == mem).float().mean().item())
    rec = {"workload": "C1 (BASELINE.json configs[0], examples/glove.rs:46-60): %d x %d-d f32 angular (synthetic stand-in for GloVe-100), "
                       "build max_search 10, member queries at (max_search %d, k %d)" % (n, dim, ef, k),
           "layers": [builder.layer_len(l) for l in range(builder.num_layers())], "build_s": round(t_build, 2),
           "four_searches": singles, "four_searches_equal_oracle_bit_for_bit": bool(ok),
           "the_four_members_are_their_own_nearest_at_distance_0": bool(self_first),
           "members_1024": {"value": round(rate * 1024, 1), "unit": "queries/s", "calls_timed": reps,
                            "fraction_that_found_itself_first": round(self_frac, 4),
                            "cpu_baseline": cb, "speedup_vs_cpu": round(rate * 1024 / cb["value"], 1)}}
    if not (ok and self_first and cb["gpu_matches_oracle"]["ids_bit_exact"] and cb["gpu_matches_oracle"]["dists_bit_exact"]):
        raise RuntimeError("C1 parity failed: %s" % json.dumps(rec)[:600])
    del oix, index, builder, el
    torch.cuda.empty_cache()
    return rec


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
    group = auto_group(args, steps)
    layer_sizes = [builder.layer_len(l) for l in range(builder.num_layers())]
    rec = {"workload": workload_label(n, dim, dtype, data, nq, ef, k), "dtype": dtype, "data": "synthetic",
           "n_elements": n, "dim": dim, "layers": layer_sizes, "build_s": round(t_build, 1),
           "index_hbm_gb": round(index.hbm_bytes() / 1e9, 2), "brute_force": bf}
    if find_ef:
        sweep = B.ef_sweep(index, queries, gt, nq, k, [20, 30, 50, 70, 100, 140, 200, 300, 400, 600, 800], steps, warmup,
                           group, stop_at=0.95)
        rec["ef_sweep"] = sweep
        ok = [s for s in sweep if s["recall_at_10"] >= 0.95]
        ef = ok[0]["ef"] if ok else sweep[-1]["ef"]
        rec["smallest_ef_with_recall_0.95"] = ef if ok else None
        rec["workload"] = workload_label(n, dim, dtype, data, nq, ef, k)
        # graph-quality guard, the reference's own (verify_search, src/index/tests.rs:50-62): members of the set, searched at
        # (max_search, 1), find themselves -- bar 0.95 as there. On data with structure a build that went wrong shows here.
        mem = torch.arange(0, n, max(1, n // 2048), device="cuda")[:2048]
        mq = elements[mem].contiguous()
        mids = torch.empty((mem.numel(), 1), dtype=torch.int64, device="cuda")
        mds = torch.empty((mem.numel(), 1), dtype=torch.float32, device="cuda")
        mcnt = torch.empty(mem.numel(), dtype=torch.int32, device="cuda")
        index.search_batch_device(mq.data_ptr(), mem.numel(), ef, 1, mids.data_ptr(), mds.data_ptr(), mcnt.data_ptr(), 0, 0, B.stream)
        torch.cuda.synchronize()
        rec["self_recall_at_1"] = round(float(((mids[:, 0] == mem) | (mds[:, 0] <= 1e-6)).float().mean().item()), 4)
        if rec["self_recall_at_1"] <= 0.95:
            raise RuntimeError("graph quality: only %.3f of %d members find themselves at max_search %d on '%s' data (bar 0.95, "
                               "src/index/tests.rs:50-62)" % (rec["self_recall_at_1"], mem.numel(), ef, data))
    m = B.measure(index, queries, dim, esize, nq, ef, k, steps, warmup, group, inflight=args.inflight, steady_s=0.3)
    wl_key = "%d|%d|%s|%s|nq%d|ef%d|k%d|nn%d|ms%d|re%d" % (n, dim, dtype, data, nq, ef, k, args.num_neighbors,
                                                        args.build_max_search, args.build_reinsert)
    rec.update({
        "value": round(m["value_local"], 1), "unit": "queries/s", "ef_search": ef, "batch": nq, "k": k, "steps": steps,
        "inflight_batches": m["inflight"], "batches_per_call": m["group"] if m["inflight"] == 1 else 1,
        "ms_per_step": round(m["elapsed"] / steps * 1e3, 4),
        "sequential": {"value": round(steps * nq / m["seq_elapsed"], 1), "ms_per_step": round(m["seq_elapsed"] / steps * 1e3, 4)},
        "one_batch_calls_in_flight": m["begin_end"],
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
        rec["cpu_baseline"] = B.cpu_baseline(oix, h_q, ef, k, g_ids, g_d, single_thread_queries=128, sweep=False)
        rec["speedup_vs_cpu"] = round(rec["value"] / rec["cpu_baseline"]["value"], 2)
        if n <= 20_000_000:  # (the oracle's scan of 125M rows is not worth its minute)
            B._gt_dists = gt_dists
            B.check_scan(oix, h_q[:rq], gt, k, bf, n_check=16)
        del oix
    del m, index, builder, elements, queries
    torch.cuda.empty_cache()
    rec["wall_s"] = round(time.time() - t_sub, 1)
    return rec


def _merge_rows_numpy(dists, ids, k):
    """[nq, m] distances and global ids -> the k smallest by (distance, id) per row (ground truth over all shards)."""
    order = np.lexsort((ids, dists), axis=1)[:, :k]
    return np.take_along_axis(ids, order, 1), np.take_along_axis(dists, order, 1)


def partitioned_record(B, args, n, dim, dtype, nq, ef, k, steps, warmup, spg, seed_base, depth=2, with_cpu=True):
    """The element set split into world * spg id ranges, one independent index per range
    (src/elements/embeddings/parsing.rs:63-100). A step = one batch through every shard's search + the ONE exchange step
    of the packed per-shard top-k + the merge kernel; steps are pipelined `depth` deep. Two drivers over the same kernels:
      torch  granne_amd/sharded.py: one process per GPU, torch streams, all_gather_into_tensor (RCCL) between ranks;
      cabi   granne_hip_sharded_begin_device / _end_device: ONE host process holds every shard (what a Rust host binds),
             peer copies or an in-library ncclAllGather as the exchange -- measured when world == 1.
    Shards are built one after the other and only the searchable index is kept (elements and builder of a shard are
    released before the next is generated): 8 x 12.5M x 200-d f32 or 4 x 125M x 100-d int8 fit one MI355X that way.
    Collective: every rank calls this with the same arguments. Returns the record (rank-local timings aside, the same
    on every rank)."""
    torch, dist = B.torch, B.dist
    from granne_amd import _lib as glib
    from granne_amd import sharded
    from oracle.merge import merge_topk_numpy, unpack_topk  # the checker, as in replica mode
    world, rank = B.world, B.rank
    G = world * spg
    esize = 4 if dtype == "f32" else 1
    n_batches = warmup + steps
    bounds = sharded.shard_bounds(n, G)
    offsets = [b[0] for b in bounds]
    mine = list(range(rank * spg, (rank + 1) * spg))
    b0 = warmup  # the batch parity and recall are taken on
    queries = B.rows(args.data, SEED + 1, 0, n_batches * nq, dim, dtype)  # the SAME batches on every rank
    q1 = queries[b0 * nq:(b0 + 1) * nq]
    h_q1 = q1.cpu().numpy()
    want_cpu = with_cpu and args.cpu_batches > 0
    indexes, layer_sizes, t_gen, t_build = [], None, 0.0, 0.0
    oracle_parts, cb, gt_parts = {}, None, []
    for j, g in enumerate(mine):
        t0 = time.time()
        # per-shard seed, clear of the query stream's (SEED + 1): shard g is rows 0.. of stream seed_base + g
        el = B.rows(args.data, seed_base + g, 0, bounds[g][1] - bounds[g][0], dim, dtype)
        torch.cuda.synchronize()
        t_gen += time.time() - t0
        builder, ix, tb = B.build_index(el, dtype)
        t_build += tb
        if layer_sizes is None:
            layer_sizes = [builder.layer_len(l) for l in range(builder.num_layers())]
        if not args.no_recall:  # exact k nearest of this shard (the scan on the matrix cores), global ids
            gi = B.ground_truth(ix, q1[:min(nq, 1024)], k, dtype)
            gt_parts.append((gi.astype(np.int64) + offsets[g], B._gt_dists.copy()))
        if want_cpu and (j == 0 or args.parity_all_shards):
            oix = B.host_index(el, builder)
            if j == 0:  # the CPU baseline: this rank's first shard, timed (the others are searched once, for parity)
                saved = (args.cpu_threads, args.cpu_seconds)
                if world > 1:  # every rank runs this on the same host: share the cores, keep it short
                    args.cpu_threads = max(1, (os.cpu_count() or 2) // (2 * world))
                    args.cpu_seconds = min(args.cpu_seconds, 0.5)
                o_ids, o_d, o_c, _ = oix.search_batch(h_q1, ef, k, n_threads=args.cpu_threads or 0)
                cb = B.cpu_baseline(oix, h_q1, ef, k, o_ids, o_d, single_thread_queries=64, sweep=(world == 1 and spg == 1))
                args.cpu_threads, args.cpu_seconds = saved
            else:
                o_ids, o_d, o_c, _ = oix.search_batch(h_q1, ef, k, n_threads=0)
            oracle_parts[g] = (o_ids, o_d, o_c)
            del oix
        del builder, el
        torch.cuda.empty_cache()
        indexes.append(ix)
        if rank == 0:
            log("partitioned: shard %d/%d built (%.1fs so far), %.1f GB of HBM in indexes"
                % (j + 1, spg, t_build, sum(i.hbm_bytes() for i in indexes) / 1e9))
    batches = [queries[b * nq:(b + 1) * nq] for b in range(n_batches)]
    drivers = {}

    # ---- driver "torch": granne_amd/sharded.py -----------------------------------------------------------------------
    sg = sharded.ShardedGranne(indexes, offsets)
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
    t_ids = torch.stack([r[0] for r in res])
    t_d = torch.stack([r[1] for r in res])
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
    drivers["torch"] = {"value": round(steps * nq / elapsed, 1), "ms_per_step": round(elapsed / steps * 1e3, 4),
                        "sequential": {"value": round(steps * nq / seq_elapsed, 1), "ms_per_step": round(seq_elapsed / steps * 1e3, 4)},
                        "phases_ms": phases, "exchange": "all_gather_into_tensor (RCCL)" if world > 1 else "none (one rank: shards write the gather buffer)",
                        "what": "granne_amd/sharded.py: torch streams, %d batches pipelined" % depth}
    # this batch's per-shard results as the ranks exchanged them (slot 0 holds the gathered buffers of the last search_batch)
    sg.search_batch(q1, ef, k)
    torch.cuda.synchronize()
    gathered = sg._slots[0].gathered.view(G, -1).cpu().numpy()
    parts = [unpack_topk(gathered[g], nq, k) for g in range(G)]
    del sg

    # ---- driver "cabi": the one-process handle of include/granne_hip.h ---------------------------------------------------
    c_ids = c_d = None
    if world == 1:
        o_ids = torch.empty((n_batches, nq, k), dtype=torch.int64, device="cuda")
        o_d = torch.empty((n_batches, nq, k), dtype=torch.float32, device="cuda")
        o_c = torch.empty((n_batches, nq), dtype=torch.int32, device="cuda")
        status = torch.zeros(4, dtype=torch.int32, device="cuda")
        s_ = B.stream
        qp = [batches[b].data_ptr() for b in range(n_batches)]
        ip, dp, cp = [o_ids[b].data_ptr() for b in range(n_batches)], [o_d[b].data_ptr() for b in range(n_batches)], \
            [o_c[b].data_ptr() for b in range(n_batches)]
        for name, exch in (("cabi", glib.SHARDED_EXCHANGE_PEER), ("cabi_rccl", glib.SHARDED_EXCHANGE_RCCL)):
            sh = sharded.ShardedHost(indexes, offsets, depth=depth)
            try:
                if exch != glib.SHARDED_EXCHANGE_PEER:
                    sh.set_option(glib.SHARDED_OPT_EXCHANGE, exch)
            except glib.GranneHipError as e:  # no librccl to load: say so, measure the rest
                drivers[name] = {"skipped": str(e)}
                sh.close()
                continue

            def pipelined(first, count):
                tickets = []
                for i in range(count):
                    b = first + i
                    tickets.append(sh.begin_device(qp[b], nq, ef, k, ip[b], dp[b], cp[b], status.data_ptr(), s_))
                    if i >= depth - 1:
                        sh.end_device(tickets[i - depth + 1], s_)
                for tk in tickets[max(0, count - depth + 1):]:
                    sh.end_device(tk, s_)

            pipelined(0, max(warmup, depth))
            torch.cuda.synchronize()
            status.zero_()
            t0 = time.perf_counter()
            pipelined(warmup, steps)
            torch.cuda.synchronize()
            el_c = time.perf_counter() - t0
            if int(status[0].item()):
                raise RuntimeError("exact-search scratch exhausted during the timed steps (cabi)")
            t0 = time.perf_counter()
            for b in range(warmup, n_batches):  # one batch at a time: stream-ordered, nothing overlaps, no host synchronisation
                sh.search_batch_device(qp[b], nq, ef, k, ip[b], dp[b], cp[b], 0, s_)
            torch.cuda.synchronize()
            seq_c = time.perf_counter() - t0
            # host buffers in and out (pinned staging + PCIe inside the clock), pipelined inside the library
            h_q = queries[warmup * nq:].cpu().numpy().reshape(steps, nq, dim)
            sh.search_batches(h_q[:depth], ef, k)
            t0 = time.perf_counter()
            h_res = sh.search_batches(h_q, ef, k)
            host_c = time.perf_counter() - t0
            drivers[name] = {"value": round(steps * nq / el_c, 1), "ms_per_step": round(el_c / steps * 1e3, 4),
                             "sequential": {"value": round(steps * nq / seq_c, 1), "ms_per_step": round(seq_c / steps * 1e3, 4)},
                             "host_pointers": {"value": round(steps * nq / host_c, 1), "ms_per_step": round(host_c / steps * 1e3, 4),
                                               "note": "granne_hip_sharded_search_batches: queries from and results to host memory, PCIe inside the clock"},
                             "exchange": "shards write the merge device's gather buffer; peer copies for remote shards" if exch == glib.SHARDED_EXCHANGE_PEER
                                         else "one in-place ncclAllGather (librccl by dlopen) over %d device(s)" % 1,
                             "what": "granne_hip_sharded_begin_device / _end_device on one stream, %d batches in flight" % depth}
            same = bool((o_ids[warmup:] == t_ids).all().item()) and bool((o_d[warmup:].view(torch.int32) == t_d.view(torch.int32)).all().item()) \
                and bool((torch.from_numpy(h_res[0].astype(np.int64)).cuda() == t_ids).all().item())
            drivers[name]["same_results_as_torch_driver"] = same
            if not same:
                raise RuntimeError("the %s driver and the torch driver returned different results" % name)
            if name == "cabi":
                c_ids, c_d = o_ids[warmup:].clone(), o_d[warmup:].clone()
            sh.close()

    pick = args.driver if (args.driver in drivers and "value" in drivers[args.driver]) else "torch"
    value, ms_step = drivers[pick]["value"], drivers[pick]["ms_per_step"]  # every rank answers the same queries: the job's rate, not a sum over ranks
    out_ids, out_d = (c_ids, c_d) if (pick == "cabi" and c_ids is not None) else (t_ids, t_d)

    # roofline of the dominant kernel: shard 0 of this rank, one launch at a time
    m = B.measure(indexes[0], queries, dim, esize, nq, ef, k, steps, warmup, 1)
    out = {
        "metric": "queries/sec, partitioned index (every rank searches every batch), batch=%d" % nq,
        "value": value, "unit": "queries/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": ms_step, "higher_is_better": True, "scaling": "partitioned",
        "scaling_note": "the element set is split over the shards and every shard answers the SAME queries: value is the job's "
                        "rate (not a sum over ranks), per-GPU work shrinks as ranks are added",
        "driver": pick, "drivers": drivers, "pipeline_depth": depth,
        "sequential": drivers[pick]["sequential"],
        "dtype": dtype, "data": "synthetic",
        "config": {
            "workload": "%d x %d-d %s angular in %d shards of %d, batch=%d, ef_search=%d, k=%d"
                        % (n, dim, dtype, G, bounds[0][1] - bounds[0][0], nq, ef, k),
            "n_elements": n, "shards": G, "shards_per_gpu": spg, "shard_elements": bounds[0][1] - bounds[0][0], "dim": dim,
            "batch": nq, "ef_search": ef, "k": k, "shard_layers": layer_sizes,
            "index_hbm_gb": round(sum(i.hbm_bytes() for i in indexes) / 1e9, 2),
            "graph": {"builder": "gpu-batched", "num_neighbors": args.num_neighbors, "max_search": args.build_max_search,
                      "reinsert": bool(args.build_reinsert), "layer_multiplier": 15.0, "batch_max": args.batch_max,
                      "build_s": round(t_build, 1), "gen_s": round(t_gen, 1)},
            "parallelism": "partitioned x%d (%d ranks x %d shards; the packed per-shard top-k + status words exchanged once per batch, "
                           "then merge_topk_kernel; %d batches pipelined)" % (G, world, spg, depth),
        },
        "exchange": {"collectives_per_batch": 1 if world > 1 else 0,
                     "bytes_per_shard_per_batch": int(sharded.packed_bytes(nq, k) + sharded.STATUS_BYTES), "ranks": world},
        "phases_ms": phases,
        "roofline": B.roofline(m, "partitioned|%d|%d|%s|nq%d|ef%d" % (bounds[0][1] - bounds[0][0], dim, dtype, nq, ef),
                               m["value_local"], nq),
    }
    out["roofline"]["note"] = "search kernel of ONE shard (this rank's first), one batch per launch"

    # ---- recall against the exact scan over ALL shards ----------------------------------------------------------------
    if not args.no_recall:
        li = torch.from_numpy(np.concatenate([p_[0] for p_ in gt_parts], 1)).cuda()
        lv = torch.from_numpy(np.concatenate([p_[1] for p_ in gt_parts], 1)).cuda()
        if world > 1:
            av = [torch.empty_like(lv) for _ in range(world)]
            ai = [torch.empty_like(li) for _ in range(world)]
            dist.all_gather(av, lv)
            dist.all_gather(ai, li)
            lv, li = torch.cat(av, 1), torch.cat(ai, 1)
        gt_all, _ = _merge_rows_numpy(lv.cpu().numpy(), li.cpu().numpy(), k)
        rq = gt_all.shape[0]
        out["recall_at_10"] = round(B.recall(gt_all, out_ids[0][:rq], k), 4)

    # ---- parity: shard results against the CPU oracle; the merged result against the numpy merge -----------------------
    if want_cpu:
        w_ids, w_d, w_c = merge_topk_numpy(np.stack([p_[0] for p_ in parts]), np.stack([p_[1] for p_ in parts]),
                                           np.stack([p_[2] for p_ in parts]), offsets, k)
        merge_ok = bool((out_ids[0].cpu().numpy().astype(np.uint64) == w_ids).all()
                        and out_d[0].cpu().numpy().tobytes() == w_d.tobytes())
        shard_ok, checked = True, 0
        for g, (oi, od, oc) in oracle_parts.items():
            gi, gd, gc = parts[g]
            shard_ok = shard_ok and bool((gi == oi).all()) and gd.tobytes() == od.tobytes() and bool((gc == oc).all())
            checked += 1
        oracle_merge_ok = None
        if world == 1 and len(oracle_parts) == G:  # every shard came from the oracle: the whole job against the CPU restatement
            x_ids, x_d, x_c = merge_topk_numpy(np.stack([oracle_parts[g][0] for g in range(G)]), np.stack([oracle_parts[g][1] for g in range(G)]),
                                               np.stack([oracle_parts[g][2] for g in range(G)]), offsets, k)
            oracle_merge_ok = bool((out_ids[0].cpu().numpy().astype(np.uint64) == x_ids).all()
                                   and out_d[0].cpu().numpy().tobytes() == x_d.tobytes())
        ok = torch.tensor([int(merge_ok), int(shard_ok)], dtype=torch.int32, device="cuda")
        if B.use_dist:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        shard_rate = cb["value"]
        out["cpu_baseline"] = {
            "value": round(shard_rate / max(1, G), 1), "unit": "queries/s", "cores": cb["cores"], "kind": "port",
            "one_shard": {"value": shard_rate, "single_thread": cb["single_thread"], "parallel_efficiency": cb["parallel_efficiency"],
                          "thread_sweep": cb.get("thread_sweep")},
            "host": cb.get("host"),
            "sample": "search only (the host index view is built outside the clock): " + cb["sample"] + "; measured on ONE shard of "
                      "%d -- a CPU host answering the partitioned index searches all %d shards per query, so the job rate is that "
                      "shard rate / %d" % (bounds[0][1] - bounds[0][0], G, G),
            "gpu_matches_oracle": {"shards_checked_against_oracle_per_rank": checked, "those_shards_bit_exact_on_every_rank": bool(int(ok[1].item())),
                                   "merged_equals_numpy_merge_of_shard_results": bool(int(ok[0].item())),
                                   "merged_equals_numpy_merge_of_ORACLE_shard_results": oracle_merge_ok,
                                   "queries_checked": int(nq)},
        }
        out["speedup_vs_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
        if not (merge_ok and shard_ok and oracle_merge_ok is not False):
            raise RuntimeError("partitioned parity failed: %s" % out["cpu_baseline"]["gpu_matches_oracle"])
    del m, indexes, queries
    torch.cuda.empty_cache()
    return out


def run_partitioned(B, args):
    """BASELINE.json configs[3]/[4] shape: bench.py --mode partitioned [--shards-per-gpu S]."""
    out = partitioned_record(B, args, args.n, args.dim, args.dtype, args.batch, args.ef, args.k, args.steps, args.warmup,
                             max(1, args.shards_per_gpu), seed_base=SEED + 100)
    out["vs_baseline"] = None
    out["kernel_sources_sha"] = csrc_sha()
    return out


LINE_LIMIT = 7000  # bytes of the stdout line (the driver keeps the last 8 KB of stdout)


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def _short(s, n=160):
    s = str(s)
    return s if len(s) <= n else s[:n - 3] + "..."


def _finite(o):
    """NaN / Infinity are not JSON: a strict parser on the other side must read the line."""
    if isinstance(o, float):
        return o if math.isfinite(o) else None
    if isinstance(o, dict):
        return {k: _finite(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_finite(v) for v in o]
    return o


def _oracle_match(g):
    """bit-exactness from a record's `gpu_matches_oracle`, in either of its two forms: one index (ids and distance bits) or a
    partitioned one (every shard against the oracle + the merged result against the numpy merge of the ORACLE's results)"""
    g = g or {}
    if "ids_bit_exact" in g:
        return bool(g.get("ids_bit_exact") and g.get("dists_bit_exact"))
    merged = g.get("merged_equals_numpy_merge_of_ORACLE_shard_results")
    if merged is None:  # N > 1 ranks: no rank's oracle holds every shard; the shards' GPU results ARE the oracle's (checked on every rank)
        merged = g.get("merged_equals_numpy_merge_of_shard_results")
    return bool(g.get("those_shards_bit_exact_on_every_rank") and merged)


def _compact_roofline(r):
    if not isinstance(r, dict):
        return r
    c = _pick(r, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic",
                  "alg_bytes_per_launch", "launch_ms_mean", "whole_timed_region_frac", "per_query"))
    if "launch" in r:
        c["launch"] = _pick(r["launch"], ("batches", "queries"))
    if "one_batch_per_launch" in r:
        c["one_batch_per_launch"] = _pick(r["one_batch_per_launch"], ("achieved", "frac", "alg_bytes_per_launch", "launch_ms_mean"))
    if r.get("traffic") is None and "traffic_note" in r:
        c["traffic_note"] = _short(r["traffic_note"], 120)
    return c


def _compact_cpu(cb):
    if not isinstance(cb, dict):
        return cb
    c = _pick(cb, ("value", "unit", "cores", "kind"))
    host = cb.get("host") or {}
    if "cgroup_quota_cpus" in host:
        c["quota_cpus"] = host["cgroup_quota_cpus"]
    if "logical_cpus" in host:
        c["host_logical_cpus"] = host["logical_cpus"]
    if "single_thread" in cb:
        c["single_thread"] = cb["single_thread"].get("value")
    if "thread_sweep" in cb:
        c["thread_sweep"] = {str(t["threads"]): t["value"] for t in cb["thread_sweep"]}
    c["sample"] = _short(cb.get("sample", ""), 200)
    g = cb.get("gpu_matches_oracle") or {}
    c["gpu_matches_oracle"] = {"bit_exact": _oracle_match(g), "queries": g.get("queries_checked")}
    return c


def _compact_sub(rec):
    """A sub-record on the line: {workload, value, frac, cpu, bit_exact} and the one-batch forms; the rest is in the side file."""
    if not isinstance(rec, dict):
        return rec
    c = {"workload": _short(rec.get("workload", ""), 96)}
    if "error" in rec:
        c["error"] = _short(rec["error"], 200)
        return c
    c.update(_pick(rec, ("value", "recall_at_10", "ef_search", "self_recall_at_1")))
    if rec.get("slow_path_queries"):
        c["slow_path_queries"] = rec["slow_path_queries"]
    if isinstance(rec.get("roofline"), dict):
        c["frac"] = rec["roofline"].get("frac")
        if rec["roofline"].get("traffic_over_algorithmic") is not None:
            c["traffic_over_algorithmic"] = rec["roofline"].get("traffic_over_algorithmic")
        ob = rec["roofline"].get("one_batch_per_launch") or {}
        c["frac_one_batch_per_launch"] = ob.get("frac")
    for k in ("sequential", "one_batch_calls_in_flight"):
        if isinstance(rec.get(k), dict):
            c[k] = rec[k].get("value")
    cb = rec.get("cpu_baseline")
    if isinstance(cb, dict):
        g = cb.get("gpu_matches_oracle") or {}
        c["cpu"] = cb.get("value")
        c["cpu_cores"] = cb.get("cores")
        c["bit_exact"] = _oracle_match(g)
        c["checked"] = g.get("queries_checked")
    return c


def compact_line(out, extras_path=None):
    """The ONE stdout line: the contract's keys, `config`, `roofline`, `cpu_baseline` and a few figures per sub-record,
    under LINE_LIMIT bytes whatever the full record holds (which goes to bench_extras.json and stderr)."""
    keep = ("metric", "value", "unit", "n_gpus", "shared_devices", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "batches_per_call", "library_calls", "inflight_batches", "slow_path_queries", "recall_at_10",
            "speedup_vs_cpu", "kernel_sources_sha")
    line = _pick(out, keep)
    cfg = dict(out.get("config") or {})
    if "parallelism" in cfg:
        cfg["parallelism"] = _short(cfg["parallelism"], 170)
    if "graph" in cfg:
        cfg["graph"] = _pick(cfg["graph"], ("builder", "num_neighbors", "max_search", "reinsert", "build_s", "reordered"))
    line["config"] = cfg
    for k in ("sequential", "one_batch_calls_in_flight", "steady"):
        if isinstance(out.get(k), dict):
            line[k] = _pick(out[k], ("value", "ms_per_step", "depth"))
            if isinstance(out[k].get("depth8"), dict):
                line[k]["depth8"] = out[k]["depth8"].get("value")
    if "roofline" in out:
        line["roofline"] = _compact_roofline(out["roofline"])
    if "cpu_baseline" in out:
        line["cpu_baseline"] = _compact_cpu(out["cpu_baseline"])
    bf = out.get("brute_force")
    if isinstance(bf, dict) and bf:
        line["brute_force"] = _pick(bf, ("ms", "value", "frac", "bound", "peak", "unit"))
    if isinstance(out.get("ef_sweep"), list):
        line["ef_sweep"] = [[s.get("ef"), s.get("recall_at_10"), s.get("qps")] for s in out["ef_sweep"]]
    if isinstance(out.get("latency_nq1"), dict):
        line["latency_nq1_us"] = _pick(out["latency_nq1"], ("median", "p99"))
    c1 = out.get("c1")
    if isinstance(c1, dict) and "error" in c1:
        line["c1"] = _pick(c1, ("workload", "error"))
    elif isinstance(c1, dict):
        m1 = c1.get("members_1024") or {}
        line["c1"] = {"workload": _short(c1.get("workload", ""), 110), "value": m1.get("value"),
                      "cpu": (m1.get("cpu_baseline") or {}).get("value"),
                      "bit_exact": bool(c1.get("four_searches_equal_oracle_bit_for_bit")),
                      "call_us": [s.get("call_us") for s in c1.get("four_searches", [])]}
    for k in ("int8", "secondary", "secondary_mixture", "c4_shard", "c5_shard", "partitioned"):
        if k in out:
            line[k] = _compact_sub(out[k])
    if isinstance(out.get("recall_target"), dict):
        rt = out["recall_target"]
        line["recall_target"] = _pick(rt, ("recall_at_10_target", "smallest_max_search_reaching_it", "gpu_over_cpu_at_target"))
        if isinstance(rt.get("walk_at_max_search"), dict):
            line["recall_target"]["walk"] = _pick(rt["walk_at_max_search"], ("max_search", "recall_at_10", "gpu_qps", "cpu_qps"))
        if isinstance(rt.get("exact_scan"), dict):
            line["recall_target"]["exact_scan"] = _pick(rt["exact_scan"], ("gpu_qps", "cpu_qps"))
    for k in ("drivers",):  # (--mode partitioned)
        if k in out:
            line[k] = _short(json.dumps(out[k]), 300)
    if extras_path:
        line["full_record"] = extras_path
    line = _finite(line)
    # whatever a later edit adds: the line stays under the limit (drop the widest optional parts first)
    for k in ("ef_sweep", "brute_force", "c1", "secondary_mixture", "c5_shard", "c4_shard", "secondary", "int8", "partitioned",
              "latency_nq1_us", "recall_target", "steady"):
        if len(json.dumps(line, allow_nan=False)) <= LINE_LIMIT:
            break
        line.pop(k, None)
    return json.dumps(line, allow_nan=False)


def write_extras(out):
    """The full record: next to the script, under gpurun_out/ when that exists (it travels back from a GPU box), and on stderr."""
    text = json.dumps(_finite(out), allow_nan=False)
    here = os.path.dirname(os.path.abspath(__file__))
    wrote = None
    for d in (here, os.path.join(here, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, "bench_extras.json"), "w") as f:
                    f.write(text + "\n")
                wrote = wrote or "bench_extras.json"
            except OSError:
                pass
    log("full record: " + text)
    return wrote


def self_launch(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks here (one process per GPU over
    RCCL), exactly as the driver's torch.distributed.run line would."""
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus:
        sys.stderr.write("bench.py: --gpus %d but this node shows %d GPU(s): refusing to report an %d-GPU number from fewer devices\n"
                         % (args.gpus, have, args.gpus))
        sys.exit(2)
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    log("launching %d ranks: %s" % (args.gpus, " ".join(cmd)))
    sys.stdout.flush()
    os.execvpe(cmd[0], cmd, env)


def partitioned_after_the_line(B, args, out):
    """N > 1, replica mode: the partitioned sub-record (every rank one id range of the 10M points, one exchange per batch),
    taken AFTER rank 0 has printed the contract line. It goes to bench_extras.json and stderr. Every rank arms a watchdog: a
    rank that fails leaves the others inside a collective, and those leave with their own alarm instead of an RCCL timeout."""
    import signal

    def give_up(signum, frame):
        log("partitioned sub-record: no result after %d s on rank %d, leaving it out" % (args.partitioned_timeout, B.rank))
        os._exit(0)

    signal.signal(signal.SIGALRM, give_up)
    signal.alarm(int(args.partitioned_timeout))
    try:
        rec = partitioned_record(B, args, args.n, args.dim, args.dtype, args.batch, args.ef, args.k, args.steps, args.warmup, 1,
                                 seed_base=SEED + 100)
    except Exception as e:  # noqa: BLE001 -- whatever it is, the line is out already
        log("partitioned sub-record failed on rank %d: %r" % (B.rank, e))
        sys.stderr.flush()
        os._exit(0)
    if B.rank == 0:
        out["partitioned"] = rec
        write_extras(out)
        log("partitioned: " + json.dumps(_finite(_compact_sub(rec)), allow_nan=False))
    # The alarm stays armed through the final barrier (main): a rank that left through the paths above never reaches it,
    # and the ranks waiting there must leave by their own watchdog, not by the process group's timeout.
    signal.alarm(max(30, int(args.partitioned_timeout) // 4))


def wait_for_rank0(B, key, timeout_s=1800):
    """Ranks other than 0 wait here while rank 0 measures recall and the CPU baseline -- blocked on the rendezvous store's
    socket, not inside a collective: a rank spinning in an RCCL barrier burns a host core of the very CPUs the baseline is
    timed on (the container's cgroup quota is 16). Rank 0 posts the key when it is through."""
    import datetime
    from torch.distributed import distributed_c10d
    store = distributed_c10d._get_default_store()
    if B.rank == 0:
        store.set(key, "1")
    else:
        store.wait([key], datetime.timedelta(seconds=timeout_s))


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args, sys.argv[1:])
    # stdout carries exactly ONE line (the JSON): anything libraries print meanwhile (RCCL's version
    # banner, progress output) is routed to stderr by pointing fd 1 at fd 2 until the very end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    B = Bench(args)
    out = run_partitioned(B, args) if args.mode == "partitioned" else run_replica(B, args)
    if B.rank == 0:
        extras = write_extras(out)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(_finite(out), allow_nan=False) if args.full_line else compact_line(out, extras), flush=True)
        os.dup2(2, 1)
    if B.use_dist and B.world > 1:
        wait_for_rank0(B, "granne_bench_line_out")  # (rank 0 took recall + the CPU baseline after the timed region)
    if args.mode == "replica" and B.world > 1 and not args.no_partitioned:
        partitioned_after_the_line(B, args, out)
    if B.use_dist:
        B.dist.barrier()
        B.dist.destroy_process_group()
        import signal
        signal.alarm(0)


if __name__ == "__main__":
    main()

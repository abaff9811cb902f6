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
                n_dist*d*s + 4*n_adj + d*s + 8*k per query, from the kernel's own exact counters) /
                mean launch duration from HIP events recorded around the kernel on its stream
  cpu_baseline  the CPU oracle (restatement of the reference's search, OpenMP over queries = the
                caller-side rayon par_iter) timed on this box's host cores on a bounded sample of
                the same batches, with the GPU results checked against it (ids + distance bits)
  ef_sweep      recall@10 and queries/sec at max_search 50..800 on the same index
  int8          the same workload on angular_int (BASELINE.json configs[2]) as a sub-record
  secondary     a second synthetic workload on which recall@10 >= 0.95 is reachable (the headline
                data is i.i.d. uniform in 100-d, where it is not): QPS at the smallest such ef
  latency_nq1   one query per call through the host-pointer API (the reference's own call shape)
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

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
    ap.add_argument("--inflight", type=int, default=3,
                    help="batches in flight: step i is enqueued on HIP stream i %% inflight (1 = strictly sequential)")
    ap.add_argument("--cpu-batches", type=int, default=16, help="batches of the CPU baseline sample (0 = skip)")
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
    def measure(self, index, queries, dim, esize, nq, ef, k, steps, warmup, inflight, contract=False):
        torch = self.torch
        n_batches = warmup + steps
        ids = torch.empty((n_batches, nq, k), dtype=torch.int64, device="cuda")
        dists = torch.empty((n_batches, nq, k), dtype=torch.float32, device="cuda")
        counts = torch.empty((n_batches, nq), dtype=torch.int32, device="cuda")
        stats = torch.zeros((n_batches, nq, 3), dtype=torch.int64, device="cuda")
        status = torch.zeros(4, dtype=torch.int32, device="cuda")

        def step(b, on):
            index.search_batch_device(queries[b * nq:(b + 1) * nq].data_ptr(), nq, ef, k, ids[b].data_ptr(),
                                      dists[b].data_ptr(), counts[b].data_ptr(), stats[b].data_ptr(),
                                      status.data_ptr(), on)

        # Step i is enqueued on stream i % inflight: a batch starts while the previous ones drain (one
        # batch of 1024 one-wave walkers fills one wave slot per SIMD). Every step is still one batch
        # of `nq` queries through one kernel launch; nothing is skipped or cached.
        streams = [torch.cuda.Stream() for _ in range(inflight)] if inflight > 1 else [torch.cuda.current_stream()]
        for b in range(warmup):
            step(b, streams[b % inflight].cuda_stream)
        torch.cuda.synchronize()
        status.zero_()
        if contract:
            self.barrier()
        else:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(warmup + i, streams[i % inflight].cuda_stream)
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

        # the same K steps strictly one after the other on ONE stream. Two pairs of HIP events per step, all
        # on the launch stream: around the whole call (scratch memset + walker + the exact walker's launch)
        # and, inside the library, immediately around the walker's dispatch = what rocprofv3 reports per kernel
        glib, _glib = self.lib, self._lib

        def hip_event():
            e = C.c_void_p()
            _glib.check(glib.granne_hip_event_create(C.byref(e)))
            return e

        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        kev = [(hip_event(), hip_event()) for _ in range(steps)]
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(steps):
            b = warmup + i
            ev[i][0].record()
            index.search_batch_device_timed(queries[b * nq:(b + 1) * nq].data_ptr(), nq, ef, k, ids[b].data_ptr(),
                                            dists[b].data_ptr(), counts[b].data_ptr(), stats[b].data_ptr(),
                                            status.data_ptr(), self.stream, kev[i][0].value, kev[i][1].value)
            ev[i][1].record()
        torch.cuda.synchronize()
        seq_elapsed = time.perf_counter() - t1
        call_ms = [a.elapsed_time(b) for a, b in ev]
        step_ms = []
        for a, b in kev:
            ms = C.c_float()
            _glib.check(glib.granne_hip_event_elapsed_ms(a, b, C.byref(ms)))
            step_ms.append(float(ms.value))
            glib.granne_hip_event_destroy(a)
            glib.granne_hip_event_destroy(b)

        st = stats[warmup:].sum(dim=(0, 1)).cpu().numpy().astype(np.float64)  # n_dist, n_expand, n_adj
        alg_total = st[0] * dim * esize + st[2] * 4 + steps * nq * (dim * esize + k * 8)
        alg_per_launch = alg_total / steps
        mean_ms = float(np.mean(step_ms))
        achieved = alg_per_launch / (mean_ms * 1e-3) / 1e9
        return {
            "elapsed": elapsed, "value_local": steps * nq / elapsed, "seq_elapsed": seq_elapsed,
            "ids": ids, "dists": dists, "counts": counts, "status": status,
            "slow": int(status[1].item()), "spill": int(status[2].item()),
            "alg_per_launch": alg_per_launch, "achieved": achieved, "launch_ms_mean": mean_ms,
            "launch_ms_min": float(np.min(step_ms)), "call_ms_mean": float(np.mean(call_ms)),
            "per_query": {"n_dist": round(st[0] / (steps * nq), 1), "n_expand": round(st[1] / (steps * nq), 1),
                          "n_adj": round(st[2] / (steps * nq), 1)},
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
            "alg_bytes_per_launch": int(m["alg_per_launch"]), "launch_ms_mean": round(m["launch_ms_mean"], 4),
            "launch_ms_min": round(m["launch_ms_min"], 4), "call_ms_mean": round(m["call_ms_mean"], 4),
            "per_query": m["per_query"],
        }
        if note:
            r["traffic_note"] = note
        return r

    # ---- ground truth / recall -----------------------------------------------------------------------
    def ground_truth(self, elements, q0, k, dtype):
        torch = self.torch
        n = elements.shape[0]
        nq = q0.shape[0]
        q0 = q0.float()
        best_v = torch.full((nq, k), -3.0e38, device="cuda")
        best_i = torch.zeros((nq, k), dtype=torch.int64, device="cuda")
        chunk = 1_000_000
        for c0 in range(0, n, chunk):
            e = elements[c0:c0 + chunk].float()
            if dtype == "i8":  # cosine on the quantised rows
                e = e / e.norm(dim=1, keepdim=True).clamp_min(1e-30)
            sim = q0 @ e.T
            v, i = sim.topk(k, dim=1)
            cat_v = torch.cat([best_v, v], 1)
            cat_i = torch.cat([best_i, i + c0], 1)
            best_v, sel = cat_v.topk(k, dim=1)
            best_i = cat_i.gather(1, sel)
        return best_i.cpu().numpy()

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
            steps, warmup = 4, 1
            q = self.rows(data, SEED + 3, 0, (steps + warmup) * nq, dim, dtype)
            m = self.measure(index, q, dim, esize, nq, ef, k, steps, warmup, 1)
            out.append({"batch": nq, "launch_ms_mean": round(m["launch_ms_mean"], 4),
                        "frac": round(m["achieved"] / HBM_PEAK_GBPS, 4),
                        "qps_one_launch_at_a_time": round(steps * nq / m["seq_elapsed"], 1),
                        "slow_path_queries": m["slow"]})
            del m, q
            self.torch.cuda.empty_cache()
        return out

    def ef_sweep(self, index, queries, gt, nq, k, efs, steps, warmup, stop_at=None):
        """recall@10 (first timed batch) and queries/sec (K batches, one at a time and three in flight)."""
        torch = self.torch
        out = []
        n_b = queries.shape[0] // nq
        o = (torch.empty((nq, k), dtype=torch.int64, device="cuda"), torch.empty((nq, k), dtype=torch.float32, device="cuda"),
             torch.empty((nq,), dtype=torch.int32, device="cuda"))
        streams = [torch.cuda.Stream() for _ in range(3)]
        for e_ in efs:
            def run(b, on):
                index.search_batch_device(queries[b * nq:(b + 1) * nq].data_ptr(), nq, e_, k, o[0].data_ptr(), o[1].data_ptr(),
                                          o[2].data_ptr(), 0, 0, on)
            run(warmup, self.stream)
            torch.cuda.synchronize()
            rec = self.recall(gt, o[0], k)
            reps = max(4, min(steps, n_b - warmup))
            t0 = time.perf_counter()
            for j in range(reps):
                run(warmup + j % (n_b - warmup), self.stream)
            torch.cuda.synchronize()
            t_seq = time.perf_counter() - t0
            t0 = time.perf_counter()
            for j in range(reps):
                run(warmup + j % (n_b - warmup), streams[j % 3].cuda_stream)  # outputs overwrite each other: timing only
            torch.cuda.synchronize()
            t_inf = time.perf_counter() - t0
            out.append({"ef": e_, "recall_at_10": round(rec, 4), "qps": round(reps * nq / t_inf, 1),
                        "qps_one_batch_at_a_time": round(reps * nq / t_seq, 1)})
            if stop_at is not None and rec >= stop_at:
                break
        return out

    # ---- the CPU oracle beside it ---------------------------------------------------------------------
    def cpu_baseline(self, elements, builder, queries_host_batches, ef, k, g_ids, g_d, order=None, single_thread_queries=256):
        """the ONLY use of oracle/ in this file: the CPU baseline + parity check."""
        from concurrent.futures import ThreadPoolExecutor
        from oracle import oracle as orc
        orc.build()
        a = self.args
        # host copy of the elements with parallel first touch: the pages end up spread over the NUMA
        # nodes of the threads that wrote them instead of all on one socket
        n = elements.shape[0]
        h_el = np.empty(tuple(elements.shape), np.float32 if elements.dtype == self.torch.float32 else np.int8)
        parts = 32
        bounds = [n * i // parts for i in range(parts + 1)]

        def cp(i):
            h_el[bounds[i]:bounds[i + 1]] = elements[bounds[i]:bounds[i + 1]].cpu().numpy()
        with ThreadPoolExecutor(8) as ex:
            list(ex.map(cp, range(parts)))
        oix = orc.Index(h_el, builder.layers())
        if order is not None:
            oix = oix.reordered(order)
        h_q = queries_host_batches
        nqs = h_q.shape[0]
        # thread count: the best of {OpenMP default, all logical CPUs} unless given (a cgroup quota below
        # the logical CPU count makes oversubscription much slower)
        cands = [a.cpu_threads] if a.cpu_threads else sorted({orc.lib().gro_max_threads(), os.cpu_count() or 1})
        best = None
        for th in cands:
            oix.search_batch(h_q[:min(nqs, 1024)], ef, k, n_threads=th)  # touch pages / spin up threads
            for _rep in range(3):  # best of three: the host is shared and noisy
                t1 = time.time()
                res = oix.search_batch(h_q, ef, k, n_threads=th)
                dt = time.time() - t1
                if best is None or dt < best[0]:
                    best = (dt, th, res)
        cpu_s, threads, (o_ids, o_d, o_c, o_ctr) = best
        m1 = min(single_thread_queries, nqs)
        t1 = time.time()
        oix.search_batch(h_q[:m1], ef, k, n_threads=1)
        single_s = time.time() - t1
        ids_ok = bool((g_ids == o_ids).all())
        d_ok = g_d.tobytes() == o_d.tobytes()
        return {
            "value": round(nqs / cpu_s, 1), "unit": "queries/s", "cores": threads, "kind": "port",
            "single_thread": {"value": round(m1 / single_s, 1), "unit": "queries/s", "queries": int(m1)},
            "sample": "%d queries of the timed workload, same index; oracle/granne_oracle.c (C restatement of the "
                      "reference's search; Rust toolchain absent), OpenMP dynamic over queries; %.2f s wall; thread counts "
                      "tried %s x 3 repeats, best reported; elements first-touched by 8 threads" % (nqs, cpu_s, cands),
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
    inflight = max(1, args.inflight)

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

    m = B.measure(index, queries, dim, esize, nq, ef, k, args.steps, args.warmup, inflight, contract=True)
    value = world * args.steps * nq / m["elapsed"]
    wl_key = "%d|%d|%s|%s|nq%d|ef%d|k%d|nn%d|ms%d|re%d" % (n, dim, args.dtype, args.data, nq, ef, k, args.num_neighbors,
                                                        args.build_max_search, args.build_reinsert)
    if args.reorder:
        wl_key += "|reordered"
    out = {
        "metric": "queries/sec (recall@10 alongside), 10M x 100-d angular, batch=1024",
        "value": round(value, 1), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(m["elapsed"] / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "inflight_batches": inflight, "slow_path_queries": m["slow"], "visited_spill_walks": m["spill"],
        "sequential": {"value": round(args.steps * nq / m["seq_elapsed"], 1),
                       "ms_per_step": round(m["seq_elapsed"] / args.steps * 1e3, 4),
                       "note": "same K steps, one stream, one batch at a time (rank-local)"},
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {
            "workload": "C2: %d x %d-d %s angular (BASELINE.json configs[1]), %s components, batch=%d, ef_search=%d, k=%d"
                        % (n, dim, args.dtype, "i.i.d. uniform" if args.data == "uniform" else "16-d latent", nq, ef, k),
            "n_elements": n, "dim": dim, "batch": nq, "ef_search": ef, "k": k, "layers": layer_sizes,
            "graph": {"builder": "gpu-batched", "num_neighbors": args.num_neighbors,
                      "max_search": args.build_max_search, "reinsert": bool(args.build_reinsert),
                      "layer_multiplier": 15.0, "batch_max": args.batch_max, "build_s": round(t_build, 1),
                      "reordered": bool(args.reorder), "reorder_s": round(t_reorder, 2)},
            "parallelism": "replica x%d (one process per GPU, no data-path collective); %d batches in flight per GPU"
                           % (world, inflight),
        },
        "roofline": B.roofline(m, wl_key, value / world, nq),
        "kernel_sources_sha": csrc_sha(),
    }

    # ---- rank 0, N = 1: recall, sweep, CPU baseline, sub-records --------------------------------------
    if rank == 0:
        b0 = args.warmup
        gt = None
        if not args.no_recall:
            gt = B.ground_truth(elements, queries[b0 * nq:(b0 + 1) * nq], k, args.dtype)
            got = m["ids"][b0]
            if order is not None:  # ground truth is in build ids, results in reordered ids
                got = torch.from_numpy(order.astype(np.int64)).cuda()[got.clamp_min(0)]
            out["recall_at_10"] = round(B.recall(gt, got, k), 4)
            efs = [int(x) for x in args.sweep_ef.split(",") if x]
            if efs and order is None:
                out["ef_sweep"] = B.ef_sweep(index, queries, gt, nq, k, efs, args.steps, args.warmup, stop_at=0.95)
        if world == 1 and args.cpu_batches > 0:
            nb = min(args.cpu_batches, args.steps)
            h_q = queries[b0 * nq:(b0 + nb) * nq].cpu().numpy()
            g_ids = m["ids"][b0:b0 + nb].reshape(-1, k).cpu().numpy().astype(np.uint64)
            g_d = m["dists"][b0:b0 + nb].reshape(-1, k).cpu().numpy()
            out["cpu_baseline"] = B.cpu_baseline(elements, builder, h_q, ef, k, g_ids, g_d, order)
            out["speedup_vs_cpu"] = round(value / out["cpu_baseline"]["value"], 2)
        if world == 1 and not args.no_extras:
            out["latency_nq1"] = B.latency_nq1(index, queries, dim, ef, k)
            if order is None:
                out["launch_scaling"] = B.launch_scaling(index, args.data, dim, args.dtype, ef, k)
        del m
        if world == 1 and not args.no_extras and args.dtype == "f32" and args.data == "uniform" and order is None:
            del index, builder, elements, queries
            torch.cuda.empty_cache()
            out["int8"] = sub_record(B, args, "i8", "uniform")
            out["secondary"] = sub_record(B, args, "f32", "latent")
    return out


def sub_record(B, args, dtype, data):
    """The same measurement on another element type / data distribution, as a sub-record of the line."""
    torch = B.torch
    n, dim, nq, k = args.n, args.dim, args.batch, args.k
    esize = 4 if dtype == "f32" else 1
    steps, warmup = args.steps, args.warmup
    elements = B.rows(data, SEED, 0, n, dim, dtype)
    queries = B.rows(data, SEED + 1, 0, (warmup + steps) * nq, dim, dtype)
    builder, index, t_build = B.build_index(elements, dtype)
    gt = B.ground_truth(elements, queries[warmup * nq:(warmup + 1) * nq], k, dtype)
    ef = args.ef
    rec = {"dtype": dtype, "data": "synthetic, %s" % ("i.i.d. uniform components (BASELINE.json configs[2])" if data == "uniform"
                                                       else "16-d latent uniform cube through a fixed random linear map into 100-d"),
           "build_s": round(t_build, 1)}
    if data == "latent":
        sweep = B.ef_sweep(index, queries, gt, nq, k, [20, 30, 50, 70, 100, 140, 200, 300, 400, 600, 800], steps, warmup, stop_at=0.95)
        rec["ef_sweep"] = sweep
        ok = [s for s in sweep if s["recall_at_10"] >= 0.95]
        ef = ok[0]["ef"] if ok else sweep[-1]["ef"]
        rec["smallest_ef_with_recall_0.95"] = ef if ok else None
    m = B.measure(index, queries, dim, esize, nq, ef, k, steps, warmup, max(1, args.inflight))
    wl_key = "%d|%d|%s|%s|nq%d|ef%d|k%d|nn%d|ms%d|re%d" % (n, dim, dtype, data, nq, ef, k, args.num_neighbors,
                                                        args.build_max_search, args.build_reinsert)
    rec.update({
        "value": round(m["value_local"], 1), "unit": "queries/s", "ef_search": ef, "batch": nq, "k": k,
        "inflight_batches": max(1, args.inflight), "ms_per_step": round(m["elapsed"] / steps * 1e3, 4),
        "sequential": {"value": round(steps * nq / m["seq_elapsed"], 1), "ms_per_step": round(m["seq_elapsed"] / steps * 1e3, 4)},
        "slow_path_queries": m["slow"], "visited_spill_walks": m["spill"],
        "recall_at_10": round(B.recall(gt, m["ids"][warmup], k), 4),
        "roofline": B.roofline(m, wl_key, m["value_local"], nq),
    })
    if data == "uniform":
        rec["launch_scaling"] = B.launch_scaling(index, data, dim, dtype, ef, k)
    if args.cpu_batches > 0:
        nb = min(args.cpu_batches, steps)  # the same bounded sample as the main record (a few thousand queries finish
        # in hundredths of a second on 128 threads: thread wake-up, not search, is what such a sample times)
        h_q = queries[warmup * nq:(warmup + nb) * nq].cpu().numpy()
        g_ids = m["ids"][warmup:warmup + nb].reshape(-1, k).cpu().numpy().astype(np.uint64)
        g_d = m["dists"][warmup:warmup + nb].reshape(-1, k).cpu().numpy()
        rec["cpu_baseline"] = B.cpu_baseline(elements, builder, h_q, ef, k, g_ids, g_d, single_thread_queries=128)
        rec["speedup_vs_cpu"] = round(rec["value"] / rec["cpu_baseline"]["value"], 2)
    del m, index, builder, elements, queries
    torch.cuda.empty_cache()
    return rec


def run_partitioned(B, args):
    """BASELINE.json configs[3]/[4]: the element set split into world * shards_per_gpu id ranges, one
    independent index per range (src/elements/embeddings/parsing.rs:63-100). A step = one batch through
    every shard's search + ONE all-gather of the packed per-shard top-k + the merge kernel."""
    torch, dist = B.torch, B.dist
    from granne_amd import sharded
    world, rank = B.world, B.rank
    spg = max(1, args.shards_per_gpu)
    G = world * spg
    n, dim, nq, ef, k = args.n, args.dim, args.batch, args.ef, args.k
    esize = 4 if args.dtype == "f32" else 1
    n_batches = args.warmup + args.steps
    bounds = sharded.shard_bounds(n, G)
    offsets = [b[0] for b in bounds]
    mine = list(range(rank * spg, (rank + 1) * spg))

    t0 = time.time()
    # per-shard seed, clear of the query stream's (SEED + 1): shard g is rows 0.. of stream SEED + 100 + g
    elements = [B.rows(args.data, SEED + 100 + g, 0, bounds[g][1] - bounds[g][0], dim, args.dtype) for g in mine]
    queries = B.rows(args.data, SEED + 1, 0, n_batches * nq, dim, args.dtype)  # the SAME batches on every rank
    torch.cuda.synchronize()
    t_gen = time.time() - t0
    builders, indexes, t_build = [], [], 0.0
    for e in elements:
        b, ix, tb = B.build_index(e, args.dtype)
        builders.append(b)
        indexes.append(ix)
        t_build += tb
    layer_sizes = [builders[0].layer_len(l) for l in range(builders[0].num_layers())]
    if rank == 0:
        log("gen %.1fs, gpu build %.1fs (%d local shards), shard layers %s" % (t_gen, t_build, spg, layer_sizes))
    sg = sharded.ShardedGranne(indexes, offsets)

    out_ids = torch.empty((n_batches, nq, k), dtype=torch.int64, device="cuda")
    out_d = torch.empty((n_batches, nq, k), dtype=torch.float32, device="cuda")

    def step(b, timed=False):
        i, d, c = sg.search_batch(queries[b * nq:(b + 1) * nq], ef, k, check_status=False, timed=timed)
        out_ids[b].copy_(i)
        out_d[b].copy_(d)

    for b in range(args.warmup):
        step(b)
    B.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    B.barrier()
    elapsed = time.perf_counter() - t0
    if B.use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if int(sg._status[:, 0].sum().item()) != 0:
        raise RuntimeError("exact-search scratch exhausted during the timed steps")
    value = args.steps * nq / elapsed  # every rank answers the same queries: the job's rate, not a sum over ranks

    # phases of a step (HIP events on the stream; synchronised, so not the pipelined rate)
    ph = {"search_ms": [], "exchange_ms": [], "merge_ms": []}
    for i in range(min(args.steps, 10)):
        step(args.warmup + i, timed=True)
        for key in ph:
            ph[key].append(sg.timings[key])
    phases = {key: round(float(np.mean(v)), 4) for key, v in ph.items()}

    # roofline of the dominant kernel: shard 0 of this rank, one launch at a time
    m = B.measure(indexes[0], queries, dim, esize, nq, ef, k, args.steps, args.warmup, 1)
    out = {
        "metric": "queries/sec (recall@10 alongside), partitioned index, batch=%d" % nq,
        "value": round(value, 1), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {
            "workload": "%d x %d-d %s angular in %d shards of %d (BASELINE.json configs[%d] shape), batch=%d, ef_search=%d, k=%d"
                        % (n, dim, args.dtype, G, bounds[0][1] - bounds[0][0], 3 if args.dtype == "f32" else 4, nq, ef, k),
            "n_elements": n, "shards": G, "shards_per_gpu": spg, "shard_elements": bounds[0][1] - bounds[0][0], "dim": dim,
            "batch": nq, "ef_search": ef, "k": k, "shard_layers": layer_sizes,
            "graph": {"builder": "gpu-batched", "num_neighbors": args.num_neighbors, "max_search": args.build_max_search,
                      "reinsert": bool(args.build_reinsert), "layer_multiplier": 15.0, "batch_max": args.batch_max,
                      "build_s": round(t_build, 1)},
            "parallelism": "partitioned x%d (%d ranks x %d shards; one all-gather of the packed per-shard top-k per batch, "
                           "then merge_topk_kernel)" % (G, world, spg),
        },
        "exchange": {"collective": "all_gather_into_tensor (RCCL)" if world > 1 else "none (one rank)",
                     "bytes_per_rank_per_batch": sg.exchange_bytes_per_rank(nq, k), "ranks": world},
        "phases_ms": phases,
        "roofline": B.roofline(m, "partitioned|%d|%d|%s|nq%d|ef%d" % (bounds[0][1] - bounds[0][0], dim, args.dtype, nq, ef),
                               m["value_local"], nq),
        "kernel_sources_sha": csrc_sha(),
    }
    out["roofline"]["note"] = "search kernel of ONE shard (this rank's first), one launch at a time"

    # ---- recall against exact brute force over ALL shards --------------------------------------------
    b0 = args.warmup
    if not args.no_recall:
        q0 = queries[b0 * nq:(b0 + 1) * nq]
        loc_v, loc_i = [], []
        for j, g in enumerate(mine):
            gt = torch.from_numpy(B.ground_truth(elements[j], q0, k, args.dtype)).cuda()
            e = elements[j].float()
            if args.dtype == "i8":
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
        out["recall_at_10"] = round(B.recall(gt_all, out_ids[b0], k), 4)

    # ---- parity: this rank's first shard against the CPU oracle; the merge against the numpy merge ----
    if args.cpu_batches > 0:
        from oracle import oracle as orc  # the checker, as in replica mode
        from oracle.merge import merge_topk_numpy, unpack_topk
        orc.build()
        q1 = queries[b0 * nq:(b0 + 1) * nq]
        pb = sharded.packed_bytes(nq, k)
        mine_buf = torch.empty((spg, pb), dtype=torch.uint8, device="cuda")
        sg._gpu_local_search(q1, ef, k, mine_buf)
        torch.cuda.synchronize()
        if world > 1:
            allb = torch.empty((world, spg, pb), dtype=torch.uint8, device="cuda")
            dist.all_gather_into_tensor(allb.view(-1), mine_buf.view(-1))
        else:
            allb = mine_buf[None]
        parts = [unpack_topk(allb.view(G, pb)[g].cpu().numpy(), nq, k) for g in range(G)]
        w_ids, w_d, w_c = merge_topk_numpy(np.stack([p_[0] for p_ in parts]), np.stack([p_[1] for p_ in parts]),
                                           np.stack([p_[2] for p_ in parts]), offsets, k)
        merge_ok = bool((out_ids[b0].cpu().numpy().astype(np.uint64) == w_ids).all()
                        and out_d[b0].cpu().numpy().tobytes() == w_d.tobytes())
        t1 = time.time()
        oix = orc.Index(elements[0].cpu().numpy(), builders[0].layers())
        o_ids, o_d, o_c, _ = oix.search_batch(q1.cpu().numpy(), ef, k, n_threads=0)
        cpu_s = time.time() - t1
        mi, md, mc = parts[mine[0]]
        shard_ok = bool((mi == o_ids).all() and md.tobytes() == o_d.tobytes() and (mc == o_c).all())
        ok = torch.tensor([int(merge_ok), int(shard_ok)], dtype=torch.int32, device="cuda")
        if B.use_dist:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        out["cpu_baseline"] = {
            "value": round(nq / cpu_s / max(1, G), 1), "unit": "queries/s", "cores": orc.lib().gro_max_threads(), "kind": "port",
            "sample": "%d queries on ONE shard of %d (oracle/granne_oracle.c, OpenMP over queries; includes building the host "
                      "index view): %.2f s; a CPU host answering the partitioned index searches all %d shards per query, "
                      "so the job rate is that shard rate / %d" % (nq, bounds[0][1] - bounds[0][0], cpu_s, G, G),
            "gpu_matches_oracle": {"every_rank_first_shard_bit_exact": bool(int(ok[1].item())),
                                   "merged_equals_numpy_merge_of_shard_results": bool(int(ok[0].item())),
                                   "queries_checked": int(nq)},
        }
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

#!/usr/bin/env python
"""BASELINE.json configs[4] through FILES: shards of 125M x 100-d int8 built by the GPU builder, written with
granne_hip_index_save (granne's own index / elements formats), dropped, loaded back with granne_hip_index_load_files (mmap)
and searched as one partitioned index (ef_search 200, batches of 4096).

  python tools/c5_files.py --shards 1            # the round trip of one shard: seconds, bytes, 4096 queries bit for bit
  python tools/c5_files.py --shards 8            # the whole 1B job on one GPU (265 GB of 288)

Results are compared with those of the same shard while it was the builder's in-memory index (whose equality with the CPU
oracle bench.py's c5_shard record checks). Appends one JSON object per stage to gpurun_out/r6_c5_files.jsonl."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--shards", type=int, default=1)
ap.add_argument("--n", type=int, default=125_000_000)
ap.add_argument("--dir", default="/dev/shm/granne_c5")
ap.add_argument("--nq", type=int, default=4096)
ap.add_argument("--ef", type=int, default=200)
ap.add_argument("--steps", type=int, default=6)
a = ap.parse_args()
sys.argv = [sys.argv[0]]
import bench  # noqa: E402

args = bench.parse()
args.dtype, args.n, args.dim = "i8", a.n, 100
B = bench.Bench(args)
torch = B.torch
import granne_amd  # noqa: E402
from granne_amd import sharded  # noqa: E402
from oracle.merge import merge_topk_numpy  # noqa: E402

os.makedirs(a.dir, exist_ok=True)
OUT = os.path.join(ROOT, "gpurun_out", "r6_c5_files.jsonl")
os.makedirs(os.path.dirname(OUT), exist_ok=True)


def emit(rec):
    rec["t"] = round(time.time() - T0, 1)
    with open(OUT, "a") as f:
        f.write(json.dumps(rec) + "\n")
    print(json.dumps(rec), flush=True)


def cgroup_gb():
    try:
        return int(open("/sys/fs/cgroup/memory.current").read()) / 1e9
    except OSError:
        return -1.0


T0 = time.time()
k, nq, ef = 10, a.nq, a.ef
n_batches = 1 + a.steps
queries = B.rows("uniform", bench.SEED + 1, 0, n_batches * nq, 100, "i8")
q0 = queries[:nq]
kept = []  # per shard: (ids, dists, counts) of batch 0 on the in-memory index
paths = []
for s in range(a.shards):
    t0 = time.time()
    el = B.rows("uniform", bench.SEED + 100 + s, 0, a.n, 100, "i8")
    torch.cuda.synchronize()
    t_gen = time.time() - t0
    builder, ix, t_build = B.build_index(el, "i8")
    ids = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    ds = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    cnt = torch.empty(nq, dtype=torch.int32, device="cuda")
    ix.search_batch_device(q0.data_ptr(), nq, ef, k, ids.data_ptr(), ds.data_ptr(), cnt.data_ptr(), 0, 0, B.stream)
    torch.cuda.synchronize()
    kept.append((ids.cpu().numpy().astype(np.uint64), ds.cpu().numpy(), cnt.cpu().numpy().astype(np.uint32)))
    hbm = ix.hbm_bytes()
    layers = [builder.layer_len(l) for l in range(builder.num_layers())]
    del builder, el
    torch.cuda.empty_cache()
    pi, pe = os.path.join(a.dir, "shard%d.granne" % s), os.path.join(a.dir, "shard%d.elements" % s)
    t0 = time.time()
    ix.save_index(pi)
    t_si = time.time() - t0
    t0 = time.time()
    ix.save_elements(pe)
    t_se = time.time() - t0
    del ix
    torch.cuda.empty_cache()
    paths.append((pi, pe))
    emit({"stage": "built+saved", "shard": s, "gen_s": round(t_gen, 1), "build_s": round(t_build, 1), "layers": layers,
          "index_hbm_gb": round(hbm / 1e9, 2), "save_index_s": round(t_si, 1), "save_elements_s": round(t_se, 1),
          "index_file_gb": round(os.path.getsize(pi) / 1e9, 2), "elements_file_gb": round(os.path.getsize(pe) / 1e9, 2),
          "cgroup_memory_gb": round(cgroup_gb(), 1)})
    if cgroup_gb() > 285:
        emit({"stage": "stop", "why": "host memory (tmpfs files + process) close to the cgroup limit"})
        sys.exit(3)

# ---- load every shard back from its files (mmap) -------------------------------------------------------------------
loaded = []
for s, (pi, pe) in enumerate(paths):
    t0 = time.time()
    ix = granne_amd.Granne.from_files(pi, "angular_int", pe, device=B.dev)
    torch.cuda.synchronize()
    t_load = time.time() - t0
    ids = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    ds = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    cnt = torch.empty(nq, dtype=torch.int32, device="cuda")
    ix.search_batch_device(q0.data_ptr(), nq, ef, k, ids.data_ptr(), ds.data_ptr(), cnt.data_ptr(), 0, 0, B.stream)
    torch.cuda.synchronize()
    same = bool((ids.cpu().numpy().astype(np.uint64) == kept[s][0]).all() and ds.cpu().numpy().tobytes() == kept[s][1].tobytes()
                and (cnt.cpu().numpy().astype(np.uint32) == kept[s][2]).all())
    loaded.append(ix)
    emit({"stage": "loaded", "shard": s, "load_files_s": round(t_load, 1), "index_hbm_gb": round(ix.hbm_bytes() / 1e9, 2),
          "hbm_in_indexes_gb": round(sum(i.hbm_bytes() for i in loaded) / 1e9, 1), "queries_checked": nq,
          "equal_to_the_in_memory_index_bit_for_bit": same, "slow_path_queries": int(ix.last_slow_count()),
          "cgroup_memory_gb": round(cgroup_gb(), 1)})
    if not same:
        sys.exit(4)

# ---- the partitioned job over the loaded shards --------------------------------------------------------------------
offsets = [s * a.n for s in range(a.shards)]
sh = sharded.ShardedHost(loaded, offsets)
out_ids = torch.empty((n_batches, nq, k), dtype=torch.int64, device="cuda")
out_d = torch.empty((n_batches, nq, k), dtype=torch.float32, device="cuda")
out_c = torch.empty((n_batches, nq), dtype=torch.int32, device="cuda")
status = torch.zeros(4, dtype=torch.int32, device="cuda")


def run(first, count):
    tickets = []
    for i in range(count):
        b = first + i
        tickets.append(sh.begin_device(queries[b * nq:(b + 1) * nq].data_ptr(), nq, ef, k, out_ids[b].data_ptr(), out_d[b].data_ptr(),
                                       out_c[b].data_ptr(), status.data_ptr(), B.stream))
        if i >= 1:
            sh.end_device(tickets[i - 1], B.stream)
    sh.end_device(tickets[-1], B.stream)


run(0, 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(1, a.steps)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
want = merge_topk_numpy(np.stack([x[0] for x in kept]), np.stack([x[1] for x in kept]), np.stack([x[2] for x in kept]), offsets, k)
merged_ok = bool((out_ids[0].cpu().numpy().astype(np.uint64) == want[0]).all() and out_d[0].cpu().numpy().tobytes() == want[1].tobytes())
emit({"stage": "partitioned", "workload": "%d shards x %d x 100-d int8 loaded from files, batch %d, ef_search %d, k %d (one GPU, one host process: "
      "granne_hip_sharded_*, two batches in flight)" % (a.shards, a.n, nq, ef, k),
      "value": round(a.steps * nq / dt, 1), "unit": "queries/s", "ms_per_batch": round(dt / a.steps * 1e3, 3), "steps": a.steps,
      "hbm_in_indexes_gb": round(sum(i.hbm_bytes() for i in loaded) / 1e9, 1), "status": status.tolist(),
      "merged_result_equals_numpy_merge_of_the_in_memory_shards_results": merged_ok, "cgroup_memory_gb": round(cgroup_gb(), 1)})
sh.close()
for pi, pe in paths:
    os.remove(pi)
    os.remove(pe)

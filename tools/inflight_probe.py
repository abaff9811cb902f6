#!/usr/bin/env python
"""How many batches in flight does the chip take, and does the answer depend on how HIP maps streams to hardware queues?
One build; the SAME streams for every configuration; every point repeated.  GPU_MAX_HW_QUEUES is set before HIP starts.

  python tools/inflight_probe.py --dtype i8 --hwq 8 --inflight 1,2,3,4,5,6,8 --reps 3
"""
import argparse
import os
import sys
import time

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="i8")
ap.add_argument("--hwq", type=int, default=0)
ap.add_argument("--inflight", default="1,2,3,4,5,6,8")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--nq", type=int, default=1024)
ap.add_argument("--ef", type=int, default=50)
ap.add_argument("--vs", default="0,4096")
ap.add_argument("--n", type=int, default=10_000_000)
ap.add_argument("--fast-build", action="store_true")
a = ap.parse_args()
if a.hwq:
    os.environ["GPU_MAX_HW_QUEUES"] = str(a.hwq)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]]
import bench  # noqa: E402

args = bench.parse()
args.dtype, args.n = a.dtype, a.n
if a.fast_build:
    args.build_max_search, args.build_reinsert = 50, 0
B = bench.Bench(args)
torch = B.torch
elements = B.rows("uniform", bench.SEED, 0, a.n, 100, a.dtype)
builder, index, tb = B.build_index(elements, a.dtype)
nq, ef, k = a.nq, a.ef, 10
nb = 32
queries = B.rows("uniform", bench.SEED + 1, 0, nb * nq, 100, a.dtype)
ids = torch.empty((nb, nq, k), dtype=torch.int64, device="cuda")
ds = torch.empty((nb, nq, k), dtype=torch.float32, device="cuda")
cnt = torch.empty((nb, nq), dtype=torch.int32, device="cuda")
status = torch.zeros(4, dtype=torch.int32, device="cuda")
streams = [torch.cuda.Stream() for _ in range(16)]
print("GPU_MAX_HW_QUEUES=%s dtype=%s build %.1fs" % (os.environ.get("GPU_MAX_HW_QUEUES"), a.dtype, tb), flush=True)


def run(n_inflight, steps):
    for i in range(steps):
        b = i % nb
        index.search_batch_device(queries[b * nq:(b + 1) * nq].data_ptr(), nq, ef, k, ids[b].data_ptr(), ds[b].data_ptr(),
                                  cnt[b].data_ptr(), 0, status.data_ptr(), streams[i % n_inflight].cuda_stream)


for vs in [int(x) for x in a.vs.split(",")]:
    index.set_option(B._lib.OPT_VISITED_SLOTS, vs)
    for infl in [int(x) for x in a.inflight.split(",")]:
        out = []
        for rep in range(a.reps):
            run(infl, 2 * infl)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(infl, a.steps)
            t_enq = time.perf_counter() - t0
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            out.append("%.2fM (enq %.0f us/step)" % (a.steps * nq / dt / 1e6, t_enq / a.steps * 1e6))
        print("vs=%-5d inflight=%d: %s" % (vs, infl, "  ".join(out)), flush=True)

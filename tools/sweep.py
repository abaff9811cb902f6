#!/usr/bin/env python
"""One build, many search configurations: queries/s with K batches in flight and the per-launch time,
for a matrix of (ef, batch, inflight, visited_slots).  Tuning tool; bench.py is the record.

  python tools/sweep.py --dtype i8 --n 10000000 --cfg ef=50,nq=1024,inflight=3,vs=0 --cfg ef=50,nq=1024,inflight=4,vs=2048
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")  # (configs with inflight > 1 want more queues than streams)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--n", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=100)
    ap.add_argument("--data", default="uniform")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--fast-build", action="store_true", help="max_search 50, no reinsertion (7 s instead of 25 s at 10M)")
    ap.add_argument("--cfg", action="append", default=[])
    ap.add_argument("--nn", type=int, default=30, help="num_neighbors of the build (33..63: layers of 64 ids on the device)")
    ap.add_argument("--latency", action="store_true", help="also time one query per call through the host-pointer API")
    a = ap.parse_args()
    sys.argv = [sys.argv[0]]
    args = bench.parse()
    args.dtype, args.n, args.dim, args.data = a.dtype, a.n, a.dim, a.data
    args.num_neighbors = a.nn
    if a.fast_build:
        args.build_max_search, args.build_reinsert = 50, 0
    B = bench.Bench(args)
    elements = B.rows(a.data, bench.SEED, 0, a.n, a.dim, a.dtype)
    builder, index, tb = B.build_index(elements, a.dtype)
    print("build %.1fs" % tb, flush=True)
    esize = 4 if a.dtype == "f32" else 1
    maxq = max(int(dict(kv.split("=") for kv in c.split(",")).get("nq", 1024)) for c in a.cfg) if a.cfg else 1024
    queries = B.rows(a.data, bench.SEED + 1, 0, (a.steps + a.warmup) * maxq, a.dim, a.dtype)
    if a.latency:
        for ef in (50, 200):
            print("nq=1 ef=%d" % ef, B.latency_nq1(index, queries, a.dim, ef, 10), flush=True)
    for c in a.cfg:
        kv = dict(x.split("=") for x in c.split(","))
        ef, nq, infl, vs = int(kv.get("ef", 50)), int(kv.get("nq", 1024)), int(kv.get("inflight", 1)), int(kv.get("vs", 0))
        group = int(kv.get("group", a.steps))  # batches per library call (round 4's timed shape); inflight > 1: round 3's streams
        index.set_option(B._lib.OPT_VISITED_SLOTS, vs)
        m = B.measure(index, queries, a.dim, esize, nq, ef, 10, a.steps, a.warmup, group, inflight=infl)
        print("%-40s value %9.0f q/s | one at a time %9.0f | launch %.4f ms (min %.4f) frac %.4f | slow %d spill %d"
              % (c, m["value_local"], a.steps * nq / m["seq_elapsed"], m["launch_ms_mean"], m["launch_ms_min"],
                 m["achieved"] / 8000.0, m["slow"], m["spill"]), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""The brute-force scan on its own (granne_hip_brute_force_device, granne_amd/csrc/brute_force.h): rate against the f32
MFMA / HBM roofline, and the command the MFMA-busy counter pass profiles (tools/gpu_prof.sh ... pmc5 with PROF_CMD).

  python tools/bf_bench.py [--n 10000000] [--nq 1024] [--dtype f32|i8] [--reps 3]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=10_000_000)
ap.add_argument("--dim", type=int, default=100)
ap.add_argument("--nq", type=int, default=1024)
ap.add_argument("--dtype", default="f32")
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
sys.argv = [sys.argv[0]]
import bench  # noqa: E402

args = bench.parse()
args.dtype, args.n, args.dim = a.dtype, a.n, a.dim
B = bench.Bench(args)
import granne_amd  # noqa: E402

el = B.rows("uniform", bench.SEED, 0, a.n, a.dim, a.dtype)
q = B.rows("uniform", bench.SEED + 1, 0, a.nq, a.dim, a.dtype)
et = "angular" if a.dtype == "f32" else "angular_int"
builder = granne_amd.GranneBuilder.from_device(et, el.data_ptr(), a.n, a.dim, device=B.dev, stream=B.stream)
index = builder.get_index()  # no layers: the scan needs the elements only
out = []
for _ in range(a.reps):
    t = {}
    B.ground_truth(index, q, 10, a.dtype, timing=t, n=a.n)
    out.append(t)
print(json.dumps(out[-1]))
print("all reps ms:", [t["ms"] for t in out], file=sys.stderr)

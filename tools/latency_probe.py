#!/usr/bin/env python
"""One query per call through the host-pointer API (granne_hip_search_batch, nq = 1: Granne::search's call shape,
src/index/mod.rs:140-150) on the benchmark's 10M x 100-d index: median / p99 of the call, per element type.
(Round 4 tried polling the stream with hipStreamQuery before the blocking hipStreamSynchronize at the end of a small call:
134.2 -> 136.3 us f32, 125.8 -> 129.4 us int8 -- HIP's wait already spins; not kept.)
usage: python tools/latency_probe.py [--n 10000000] [--reps 400]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=10_000_000)
ap.add_argument("--reps", type=int, default=400)
a = ap.parse_args()
sys.argv = [sys.argv[0]]  # bench.parse reads sys.argv: its defaults
args = bench.parse()
args.n = a.n
B = bench.Bench(args)
out = {}
for dtype in ("f32", "i8"):
    el = B.rows("uniform", bench.SEED, 0, a.n, 100, dtype)
    q = B.rows("uniform", bench.SEED + 1, 0, a.reps, 100, dtype)
    builder, index, tb = B.build_index(el, dtype)
    out[dtype] = B.latency_nq1(index, q, 100, 50, 10, reps=a.reps)
    del builder, index, el
    B.torch.cuda.empty_cache()
print(json.dumps(out))

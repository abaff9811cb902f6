#!/usr/bin/env python
"""Throughput of the stand-alone dists operator against the HBM roofline:
nq queries x m random element ids each over n synthetic rows; algorithmic bytes = pairs x row bytes."""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import granne_amd  # noqa: E402
from granne_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--elements", type=int, default=10_000_000)
ap.add_argument("--dim", type=int, default=100)
ap.add_argument("--dtype", default="f32", choices=["f32", "i8"])
ap.add_argument("--nq", type=int, default=1024)
ap.add_argument("--m", type=int, default=4096)
ap.add_argument("--steps", type=int, default=10)
a = ap.parse_args()
n, dim = a.elements, a.dim
lib = _lib.lib()
s = torch.cuda.current_stream().cuda_stream
rows = torch.empty((n, dim), dtype=torch.float32, device="cuda")
_lib.check(lib.granne_hip_synth_rows_device(C.c_void_p(rows.data_ptr()), 7, 0, n, dim, 0, C.c_void_p(s)))
q = torch.empty((a.nq, dim), dtype=torch.float32, device="cuda")
_lib.check(lib.granne_hip_synth_rows_device(C.c_void_p(q.data_ptr()), 8, 0, a.nq, dim, 0, C.c_void_p(s)))
if a.dtype == "f32":
    for t in (rows, q):
        _lib.check(lib.granne_hip_normalize_f32_device(C.c_void_p(t.data_ptr()), t.shape[0], dim, 0, C.c_void_p(s)))
    et, esize = "angular", 4
else:
    r8, q8 = torch.empty((n, dim), dtype=torch.int8, device="cuda"), torch.empty((a.nq, dim), dtype=torch.int8, device="cuda")
    for src, dst in ((rows, r8), (q, q8)):
        _lib.check(lib.granne_hip_quantize_f32_device(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), src.shape[0],
                                                      dim, 0, C.c_void_p(s)))
    rows, q, et, esize = r8, q8, "angular_int", 1
torch.cuda.synchronize()
ix = granne_amd.Granne.from_device(et, rows.data_ptr(), n, dim, [], [], [], device=0, stream=s)
torch.cuda.synchronize()
ids = torch.randint(0, n, (a.nq, a.m), dtype=torch.int32, device="cuda")
out = torch.empty((a.nq, a.m), dtype=torch.float32, device="cuda")
ix.dists_device(q.data_ptr(), a.nq, ids.data_ptr(), a.m, out.data_ptr(), 0, s)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.steps):
    ix.dists_device(q.data_ptr(), a.nq, ids.data_ptr(), a.m, out.data_ptr(), 0, s)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.steps
pairs = a.nq * a.m
alg = pairs * (dim * esize + 8)  # the row, its id and the result; queries stay in cache
print(json.dumps({"op": "dists", "dtype": a.dtype, "n": n, "dim": dim, "pairs_per_launch": pairs, "ms_per_launch": round(ms, 4),
                  "pairs_per_s": round(pairs / ms * 1e3, 1), "alg_GBps": round(alg / ms / 1e6, 1),
                  "hbm_frac_of_8TBps": round(alg / ms / 1e6 / 8000, 4), "sample_out": out[0, :3].tolist()}))

#!/usr/bin/env python
"""One query per call and one batch of 1024 per call on the 10M x 100-d f32 index under the library's environment knobs
(read once per process, so one process per setting): the shipped default, GRANNE_HIP_INLINE_TAILS=0 (rows read whole: the
layout before round 6) and GRANNE_HIP_TOUCH_MAX=0 (no rows touched ahead in small launches).
usage: python tools/r6_latency_ab.py            (spawns itself once per setting)"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SETTINGS = [("default", {}), ("inline_tails_off", {"GRANNE_HIP_INLINE_TAILS": "0"}), ("touch_off", {"GRANNE_HIP_TOUCH_MAX": "0"}),
            ("touch_1024", {"GRANNE_HIP_TOUCH_MAX": "1024"})]

if len(sys.argv) > 1 and sys.argv[1] == "--one":
    import bench
    sys.argv = [sys.argv[0]]
    args = bench.parse()
    args.build_max_search, args.build_reinsert = 50, 0  # (a quicker graph: the comparison is between settings)
    B = bench.Bench(args)
    torch = B.torch
    el = B.rows("uniform", bench.SEED, 0, args.n, 100, "f32")
    q = B.rows("uniform", bench.SEED + 1, 0, 24 * 1024, 100, "f32")
    builder, index, tb = B.build_index(el, "f32")
    lat = B.latency_nq1(index, q, 100, 50, 10, reps=600)
    ids = torch.empty((1024, 10), dtype=torch.int64, device="cuda")
    ds = torch.empty((1024, 10), dtype=torch.float32, device="cuda")
    cnt = torch.empty(1024, dtype=torch.int32, device="cuda")
    res = {"latency_nq1_us": lat["median"], "p99": lat["p99"]}
    for nq in (16, 256, 1024):
        def run(j):
            index.search_batch_device(q[j * 1024:j * 1024 + nq].data_ptr(), nq, 50, 10, ids.data_ptr(), ds.data_ptr(), cnt.data_ptr(), 0, 0, B.stream)
        rate, reps = B.timed_window(run, 24, min_s=0.2)
        res["one_call_of_%d_us" % nq] = round(1e6 / rate, 1)
    print("RESULT " + json.dumps(res))
else:
    for name, env in SETTINGS:
        e = dict(os.environ)
        e.update(env)
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=e, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")]
        print("%-18s %s" % (name, line[0][7:] if line else "FAILED: " + out.stderr[-400:]), flush=True)

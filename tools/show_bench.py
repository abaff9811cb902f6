#!/usr/bin/env python
"""Compact view of a bench.py JSON line.  usage: python tools/show_bench.py file.json"""
import json
import sys

d = json.load(open(sys.argv[1]))


def line(tag, r):
    rf = r.get("roofline", {})
    cb = r.get("cpu_baseline", {})
    print("%-10s value %10.0f q/s (%.4f ms/step) | one batch per call %10.0f | timed launch %.4f ms (min %.4f) frac %.4f traffic %s | ef %s recall %s | cpu %s (1t %s) x%s match %s slow %s spill %s"
          % (tag, r.get("value", 0), r.get("ms_per_step", 0), r.get("sequential", {}).get("value", 0), rf.get("launch_ms_mean", 0),
             rf.get("launch", {}).get("ms_min", rf.get("launch_ms_min", 0)), rf.get("frac", 0), rf.get("traffic"), r.get("ef_search", r.get("config", {}).get("ef_search")),
             r.get("recall_at_10"), cb.get("value"), cb.get("single_thread", {}).get("value"), r.get("speedup_vs_cpu"),
             cb.get("gpu_matches_oracle"), r.get("slow_path_queries"), r.get("visited_spill_walks")))


line(d.get("dtype", "main"), d)
for k in ("int8", "secondary", "c4_shard", "c5_shard"):
    if k in d:
        line(k, d[k])
for k in ("ef_sweep",):
    if k in d:
        print("ef_sweep:", " ".join("%d:%.3f/%.0fk/%.0fk" % (s["ef"], s["recall_at_10"], s["qps"] / 1e3, s["qps_one_batch_at_a_time"] / 1e3) for s in d[k]))
if "secondary" in d and "ef_sweep" in d["secondary"]:
    print("secondary sweep:", " ".join("%d:%.3f/%.0fk" % (s["ef"], s["recall_at_10"], s["qps"] / 1e3) for s in d["secondary"]["ef_sweep"]))
for tag, r in (("f32", d), ("int8", d.get("int8", {}))):
    if "launch_scaling" in r:
        print("launch_scaling %s:" % tag, " ".join("%d:%.4fms/frac %.3f/%.0fk" % (x["batch"], x["launch_ms_mean"], x["frac"], x["qps_one_launch_at_a_time"] / 1e3)
                                                    for x in r["launch_scaling"]))
for k in ("latency_nq1", "phases_ms", "exchange"):
    if k in d:
        print(k, d[k])

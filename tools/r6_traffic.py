#!/usr/bin/env python
"""profiles/pmc_traffic.json from the PMC passes of tools/r6_prof.sh (gpurun_out/prof_r6_*/summary.csv.traffic.json), stamped
with the hash of the walker's sources (bench.py quotes roofline.traffic only while that hash matches), and the summaries
copied to profiles/r6_*."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

sha = bench.csrc_sha()
p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
d = json.load(open(p))
KEYS = {"r6_f32": "10000000|100|f32|uniform|nq1024|ef50|k10|nn30|ms200|re1|g20",
        "r6_i8": "10000000|100|i8|uniform|nq1024|ef50|k10|nn30|ms200|re1|g20",
        "r6_c4shard": "12500000|200|f32|uniform|nq4096|ef50|k10|nn30|ms200|re1|g10",
        "r6_c5shard": "125000000|100|i8|uniform|nq4096|ef200|k10|nn30|ms200|re1|g10",
        "r6_latent": "10000000|100|f32|latent|nq1024|ef30|k10|nn30|ms200|re1|g20",
        "r6_latent_reordered": "10000000|100|f32|latent|nq1024|ef30|k10|nn30|ms200|re1|reordered|g20"}
for tag, key in KEYS.items():
    out = os.path.join(ROOT, "gpurun_out", "prof_%s" % tag)
    f = os.path.join(out, "summary.csv.traffic.json")
    for name, dst in (("summary.csv", "%s_rocprof_summary.csv"), ("kernel_stats_full.csv", "%s_kernel_stats_full.csv"),
                      ("bench_trace.json", "%s_bench_under_trace.json")):
        if os.path.exists(os.path.join(out, name)):
            shutil.copy(os.path.join(out, name), os.path.join(ROOT, "profiles", dst % tag))
    if not os.path.exists(f):
        continue
    t = json.load(open(f))
    t["csrc_sha"] = sha
    t["source"] = ("profiles/%s_rocprof_summary.csv: rocprofv3 --pmc FETCH_SIZE (/ --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum) passes over python "
                   "bench.py --profile-run (tools/r6_prof.sh): mean over the launches of the timed shape; FETCH_SIZE is in KB and doubled per "
                   "MI355X_MICROARCH.md (gfx950 counts 16-B/lane loads at half), cross-checked by TCC_MISS_sum x 128 B where that pass was taken" % tag)
    d[key] = t
    print(key, t.get("hbm_bytes_per_launch"), sha)
json.dump(d, open(p, "w"), indent=1)

#!/usr/bin/env python
"""profiles/pmc_traffic.json from the PMC passes of tools/r3_prof.sh (gpurun_out/prof_r3_{f32,i8}/summary.csv.traffic.json),
stamped with the hash of the walker's sources: bench.py quotes roofline.traffic only while that hash matches."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

sha = bench.csrc_sha()
p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
d = json.load(open(p))
for key, tag in (("10000000|100|f32|uniform|nq1024|ef50|k10|nn30|ms200|re1", "r3_f32"),
                 ("10000000|100|i8|uniform|nq1024|ef50|k10|nn30|ms200|re1", "r3_i8")):
    f = os.path.join(ROOT, "gpurun_out", "prof_%s" % tag, "summary.csv.traffic.json")
    if not os.path.exists(f):
        continue
    t = json.load(open(f))
    t["csrc_sha"] = sha
    t["source"] = ("profiles/%s_rocprof_summary.csv: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum passes over "
                   "'python bench.py --steps 10 --cpu-batches 0 --no-recall --no-extras --inflight 1%s' (tools/r3_prof.sh), mean over the "
                   "benchmark's fast_kernel launches; FETCH_SIZE is in KB and doubled per MI355X_MICROARCH.md (gfx950 counts 16-B/lane "
                   "loads at half), cross-checked by TCC_MISS_sum x 128 B" % (tag, " --dtype i8" if "i8" in tag else ""))
    d[key] = t
    print(key, t["hbm_bytes_per_launch"], sha)
json.dump(d, open(p, "w"), indent=1)

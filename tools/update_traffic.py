#!/usr/bin/env python
"""Record a PMC traffic measurement (tools/gpu_prof.sh -> summary.csv.traffic.json) in
profiles/pmc_traffic.json under a workload key, stamped with the hash of the kernel sources it was
measured on (bench.py quotes it only while that hash matches).
usage: python tools/update_traffic.py <workload key> <summary.csv.traffic.json> <source note>"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

key, src, note = sys.argv[1], sys.argv[2], sys.argv[3]
path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
db = json.load(open(path)) if os.path.exists(path) else {}
ent = json.load(open(src))
ent["csrc_sha"] = bench.csrc_sha()
ent["source"] = note
db[key] = ent
json.dump(db, open(path, "w"), indent=1)
print(key, ent)

#!/usr/bin/env python
"""How full is the chip during the timed steps?  Reads a rocprofv3 kernel trace CSV of a bench.py run and prints,
for the walker launches of the timed region (grid 1024 x 64): their durations, the time between the end of one
walker and the start of the next on the same queue (scratch memset + exact-walker launch + dispatch latency), and
the mean number of walkers running concurrently.   usage: python tools/trace_gaps.py <dir with *kernel_trace.csv>"""
import csv
import glob
import os
import statistics
import sys

f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
walk = [r for r in rows if "fast_kernel" in r["Kernel_Name"] and int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) == 65536]
walk.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 2 * K launches are: K with batches in flight, then K one at a time
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for name, sel in (("batches in flight", walk[-2 * K:-K]), ("one at a time", walk[-K:])):
    if not sel:
        continue
    t0 = min(int(r["Start_Timestamp"]) for r in sel)
    t1 = max(int(r["End_Timestamp"]) for r in sel)
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in sel]
    busy = sum(dur)
    byq = {}
    for r in sel:
        byq.setdefault(r.get("Queue_Id", "?"), []).append(r)
    gaps = []
    for q, rs in byq.items():
        rs.sort(key=lambda r: int(r["Start_Timestamp"]))
        for a, b in zip(rs, rs[1:]):
            gaps.append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3)
    print("%-18s %d launches on %d queues: window %.1f us, walker mean %.1f us (min %.1f max %.1f), mean concurrency %.2f, "
          "gap to the next walker on the same queue: mean %.1f us (min %.1f max %.1f)"
          % (name, len(sel), len(byq), (t1 - t0) / 1e3, statistics.mean(dur), min(dur), max(dur), busy / ((t1 - t0) / 1e3),
             statistics.mean(gaps) if gaps else 0, min(gaps) if gaps else 0, max(gaps) if gaps else 0))

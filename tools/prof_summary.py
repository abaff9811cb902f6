"""Summarise rocprofv3 CSV outputs (kernel stats + PMC counters per kernel) into profiles/."""
import csv, glob, os, sys, collections, statistics
root = sys.argv[1]; out = sys.argv[2]
lines = []
# bench.py also walks its timed batches once with the exact visited tables (the counting pass: fast_kernel<.., 1> / <.., 2>);
# the launches summarised here are the timed form's (fast_kernel<.., 3>: no visited set) whenever the trace holds any
import re
TIMED_RE = re.compile(r"(false|true), [345](, (false|true))?>")  # V16 = 3 / 4 / 5: no visited set (5: revisits skipped before their rows are fetched)
# the launches of interest by their walker count: PROF_WALKERS walker blocks + up to 64 tail blocks, 64 threads each
# (default: one batch of 1024; round 4's timed shape is --steps x 1024 walkers in one launch, shards 10 x 4096)
WALKERS = int(os.environ.get("PROF_WALKERS", "1024"))
GRID_LO, GRID_HI = WALKERS * 64, WALKERS * 64 + 64 * 64


def is_timed(n):
    return TIMED_RE.search(n) is not None


def timed_only(names):
    return any(is_timed(n) for n in names)
# kernel trace
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(list)
    bench = []
    only3 = timed_only(r["Kernel_Name"] for r in rows)
    for r in rows:
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        agg[r["Kernel_Name"]].append(dur)
        # the benchmark's launches: 1024 walker blocks + the tail blocks (slow_kernel.h), 64 threads each
        if ("search_kernel" in r["Kernel_Name"] or "fast_kernel" in r["Kernel_Name"]) and GRID_LO <= int(r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", 0)) <= GRID_HI:
            if only3 and not is_timed(r["Kernel_Name"]):
                continue
            bench.append(dur)
    tot = sum(sum(v) for v in agg.values())
    lines.append("# kernel trace: kernel, calls, total_ms, avg_us, min_us, max_us, pct")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        lines.append("%s,%d,%.3f,%.2f,%.2f,%.2f,%.2f" % (k[:100], len(v), sum(v) / 1e3, statistics.mean(v), min(v), max(v), 100 * sum(v) / tot))
    if bench:
        t = bench[3:] if len(bench) > 3 else bench
        lines.append("# walker launches of the timed shape (grid (%d + tail) x 64): n=%d mean %.1f us min %.1f us max %.1f us" % (WALKERS, len(t), statistics.mean(t), min(t), max(t)))
# pmc
for d in sorted(glob.glob(os.path.join(root, "pmc*"))):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        only3 = timed_only(r["Kernel_Name"] for r in rows)
        for r in rows:
            if r["Kernel_Name"].startswith("void granne_hip::bf_") or "::bf_" in r["Kernel_Name"][:40]:
                agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
                continue
            if "search_kernel" not in r["Kernel_Name"] and "fast_kernel" not in r["Kernel_Name"]:
                continue
            gs = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)
            if not (GRID_LO <= gs <= GRID_HI):
                continue
            if only3 and not is_timed(r["Kernel_Name"]):
                continue
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            lines.append("# PMC (%s) %s, per launch (mean over %d launches of the benchmark's batches / of the scan)" % (os.path.basename(d), k[:80], len(next(iter(cs.values())))))
            for c, v in sorted(cs.items()):
                lines.append("%s,%.1f" % (c, statistics.mean(v)))
open(out, "w").write("\n".join(lines) + "\n")
# machine-readable HBM traffic per launch for bench.py (FETCH_SIZE is in KB and, on gfx950, counts
# half the bytes of 16-byte-per-lane loads: MI355X_MICROARCH.md, HBM section)
vals = {}
for l in lines:
    for k in ("FETCH_SIZE", "WRITE_SIZE", "TCC_MISS_sum"):
        if l.startswith(k + ","):
            vals[k] = float(l.split(",")[1])
if "FETCH_SIZE" in vals:
    import json
    traffic = int(2 * vals["FETCH_SIZE"] * 1024 + vals.get("WRITE_SIZE", 0.0) * 1024)
    json.dump({"hbm_bytes_per_launch": traffic, "fetch_size_kb": vals["FETCH_SIZE"], "write_size_kb": vals.get("WRITE_SIZE"),
               "tcc_miss_x128": int(vals.get("TCC_MISS_sum", 0) * 128)}, open(out + ".traffic.json", "w"))
print("\n".join(lines))

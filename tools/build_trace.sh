#!/bin/bash
# kernel trace of one default 10M build (+ a few searches): where the build's seconds go
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
D=${1:-f32}
rm -rf /tmp/bt && mkdir -p /tmp/bt
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bt -o bt -- python $OLDPWD/tools/sweep.py --dtype $D --steps 2 --warmup 1 --cfg ef=50,nq=1024,inflight=1 ) > gpurun_out/build_trace_$D.log 2>&1
f=$(find /tmp/bt -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY' | tee gpurun_out/build_trace_$D.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.2f s" % (tot / 1e9))
for r in rows[:14]:
    print("%-90s calls %6s total %8.3f s avg %10.1f us" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e9, float(r["AverageNs"]) / 1e3))
PY
grep build gpurun_out/build_trace_$D.log | head -3

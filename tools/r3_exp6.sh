#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/pytest_gpu.log
python bench.py --mode partitioned --shards-per-gpu 2 --steps 20 --warmup 5 > gpurun_out/bench_part.json 2> gpurun_out/bench_part.err; echo "part rc $?"; tail -3 gpurun_out/bench_part.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_part.json'))
print({k:d[k] for k in ('value','ms_per_step','sequential','phases_ms','pipeline_depth','recall_at_10')})
print(d['cpu_baseline']['gpu_matches_oracle'], d['cpu_baseline']['value'], d['cpu_baseline']['one_shard'])
PY
for LG in 10 11; do GRANNE_HIP_V16_LG=$LG python tools/sweep.py --dtype f32 --steps 10 --cfg ef=200,nq=4096,inflight=1,vs=0 --cfg ef=200,nq=1024,inflight=4,vs=0 2>&1 | grep -v Warn | sed "s/^/lg=$LG /" | tail -3; done
S=$(date +%s); python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc $? wall $(( $(date +%s) - S )) s"; tail -3 gpurun_out/bench_default.err

#!/bin/bash
python bench.py --steps 20 --warmup 5 --c4-elements 0 --c5-elements 0 > gpurun_out/bench_noshards.json 2> gpurun_out/bench_noshards.err; echo "rc $?"; tail -3 gpurun_out/bench_noshards.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_noshards.json'))
print(json.dumps(d['brute_force'])[:1500]); print(json.dumps(d['int8']['brute_force'])[:900]); print(d['roofline']['traffic'], d['int8']['roofline']['traffic'])
PY

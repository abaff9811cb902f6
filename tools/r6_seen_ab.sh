#!/bin/bash
# A/B of the walkers' "seen" pre-filter (walk_fast.h, SEEN: revisits skipped before their rows are fetched): the same bench
# shapes with GRANNE_HIP_SEEN_MIN = 1e9 (off: no launch is that large) and = 2048 (launches of 2048 walks and more). usage: tools/r6_seen_ab.sh
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
CASES=("uniform f32 50 100" "latent f32 30 100" "mixture f32 50 100" "mixture f32 200 100")
# SEEN_AB_GEN=1: the streamed walker's dims instead (4M points)
if [ -n "$SEEN_AB_GEN" ]; then CASES=("uniform f32 50 96" "uniform f32 50 384" "mixture f32 50 128" "mixture f32 50 384"); export SEEN_AB_N=4000000; fi
for data in "${CASES[@]}"; do
  set -- $data
  for seen in off 2048; do
    if [ "$seen" = "off" ]; then export GRANNE_HIP_SEEN_MIN=1000000000; else export GRANNE_HIP_SEEN_MIN=$seen; fi
    python bench.py --data $1 --dtype $2 --ef $3 --dim ${4:-100} --elements ${SEEN_AB_N:-10000000} --steps 20 --warmup 3 --no-extras --cpu-batches 0 --no-recall --c5-elements 0 --build-max-search 50 --build-reinsert 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('%-8s %-4s dim ${4:-100} ef %-3s seen %-5s value %10.0f  frac %.3f  one batch %.3f  sequential %9.0f  rows/query %.0f (distinct %.0f)' % ('$1','$2','$3','$seen', d['value'], r['frac'], r['one_batch_per_launch']['frac'], d['sequential']['value'], r['per_query']['rows_evaluated'], r['per_query']['n_dist']))"
  done
done

#!/bin/bash
# A/B of the walkers' "seen" pre-filter (walk_fast.h, SEEN: revisits skipped before their rows are fetched): the same bench
# shapes with GRANNE_HIP_SEEN_MIN unset (off) and = 2048 (launches of 2048 walks and more). usage: tools/r6_seen_ab.sh
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for data in "uniform f32 50" "latent f32 30" "mixture f32 50" "mixture f32 200"; do
  set -- $data
  for seen in off 2048; do
    if [ "$seen" = "off" ]; then unset GRANNE_HIP_SEEN_MIN; else export GRANNE_HIP_SEEN_MIN=$seen; fi
    python bench.py --data $1 --dtype $2 --ef $3 --steps 20 --warmup 3 --no-extras --cpu-batches 0 --no-recall --c5-elements 0 --build-max-search 50 --build-reinsert 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('%-8s %-4s ef %-3s seen %-5s value %10.0f  frac %.3f  one batch %.3f  sequential %9.0f  rows/query %.0f (distinct %.0f)' % ('$1','$2','$3','$seen', d['value'], r['frac'], r['one_batch_per_launch']['frac'], d['sequential']['value'], r['per_query']['rows_evaluated'], r['per_query']['n_dist']))"
  done
done

#!/bin/bash
# Every shape the register walker serves, timed: streamed f32 dims, wide int8 rows, 64-id layers.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/r5_shapes.txt
CFG="--cfg ef=50,nq=1024,group=1 --cfg ef=50,nq=1024,group=20 --cfg ef=600,nq=1024,group=8 --cfg ef=1024,nq=1024,group=8"
{
echo "# tools/sweep.py: queries/s, launch ms, fraction of 8 TB/s (algorithmic bytes / launch), queries on the slow path"
for spec in "f32 96 4000000 30" "f32 300 2000000 30" "f32 768 1000000 30" "i8 200 4000000 30" "i8 300 4000000 30" "f32 100 4000000 40" "i8 100 4000000 63"; do
  set -- $spec
  echo "== $1 dim $2, $3 points, num_neighbors $4"
  timeout 600 python tools/sweep.py --dtype $1 --dim $2 --n $3 --nn $4 --fast-build --steps 16 --warmup 2 $CFG 2>&1 | grep -v "amdgpu.ids"
done
} > $OUT 2>&1
cat $OUT

#!/bin/bash
# rows touched ahead in launches of a few queries (walk_fast.h TOUCH): latency of one query per call and small batches, on / off
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
for d in f32 i8; do
  for t in 0 64; do
    echo "== $d GRANNE_HIP_TOUCH_MAX=$t"
    GRANNE_HIP_TOUCH_MAX=$t python tools/sweep.py --dtype $d --steps 40 --latency --cfg ef=50,nq=1,inflight=1 --cfg ef=50,nq=16,inflight=1 --cfg ef=50,nq=64,inflight=1 2>&1 | grep -v amdgpu.ids
  done
done 2>&1 | tee gpurun_out/touch.txt

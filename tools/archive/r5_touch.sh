#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
{
for dt in f32 i8; do
  for tm in 64 1024 4096; do
    echo "== $dt touch_max $tm"
    GRANNE_HIP_TOUCH_MAX=$tm timeout 300 python tools/sweep.py --dtype $dt --n 10000000 --fast-build --steps 20 \
      --cfg ef=50,nq=256,group=1 --cfg ef=50,nq=1024,group=1 --cfg ef=50,nq=1024,group=20 --cfg ef=50,nq=4096,group=1 2>&1 | grep -v "amdgpu.ids\|^build"
  done
done
} > gpurun_out/r5_touch.txt 2>&1
cat gpurun_out/r5_touch.txt

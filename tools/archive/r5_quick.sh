#!/bin/bash
# tools/r5_quick.sh <tag> [variants]: parity of the shipped library + the sweep of tools/r5_ab.sh without the phase clocks
cd "$(dirname "$0")/.."
TAG=${1:-q}; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/r5_$TAG.txt
{
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
for dt in f32 i8; do
  for v in ${VARIANTS:-ship}; do
    if [ "$v" = "ship" ]; then unset GRANNE_HIP_LIB; else export GRANNE_HIP_LIB=$PWD/granne_amd/lib/libgranne_hip_$v.so; fi
    echo "== $dt variant $v"
    timeout 300 python tools/sweep.py --dtype $dt --n 10000000 --fast-build --latency --steps 20 \
      --cfg ef=50,nq=1024,group=1 --cfg ef=50,nq=1024,group=20 --cfg ef=200,nq=4096,group=1 --cfg ef=200,nq=4096,group=10 2>&1 | grep -v "amdgpu.ids\|^build" | sed 's/, .note.*//'
  done
done
} > $OUT 2>&1
cat $OUT

#!/bin/bash
# Round 5 A/B on one GPU box: parity of the shipped library, then old vs new walkers (variants in granne_amd/lib),
# then the phase clocks of the new one.   tools/r5_ab.sh [tag] [pytest-target]
cd "$(dirname "$0")/.."
TAG=${1:-ab}; TESTS=${2:-tests/test_gpu_parity.py}
mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/r5_$TAG.txt
{
echo "== parity ($TESTS)"
timeout 600 python -m pytest $TESTS -m gpu -x -q 2>&1 | tail -8
for dt in f32 i8; do
  for v in ${VARIANTS:-old ship}; do
    if [ "$v" = "ship" ]; then unset GRANNE_HIP_LIB; else export GRANNE_HIP_LIB=$PWD/granne_amd/lib/libgranne_hip_$v.so; fi
    echo "== $dt variant $v"
    timeout 300 python tools/sweep.py --dtype $dt --n 10000000 --fast-build --latency --steps 20 \
      --cfg ef=50,nq=1024,group=1 --cfg ef=50,nq=1024,group=20 --cfg ef=200,nq=4096,group=1 --cfg ef=200,nq=4096,group=10 2>&1 | grep -v amdgpu.ids
  done
done
unset GRANNE_HIP_LIB
for dt in f32 i8; do
  echo "== phase clocks, new walker, $dt, nq=1"
  GRANNE_HIP_LIB=$PWD/granne_amd/lib/libgranne_hip_phase.so timeout 300 python tools/phase_probe.py --dtype $dt --nq 1 --fast-build 2>&1 | grep -v amdgpu.ids | head -24
done
} > $OUT 2>&1
tail -120 $OUT

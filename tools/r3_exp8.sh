#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/pytest_v20.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/pytest_v20.log
# 40M x 100-d int8: ids beyond the 16-bit tags at 512 / 1024 buckets -> 20-bit entries (vs=0) against the 32-bit table (vs=4096)
python tools/sweep.py --dtype i8 --n 40000000 --steps 20 --fast-build \
   --cfg ef=200,nq=4096,inflight=1,vs=0 --cfg ef=200,nq=4096,inflight=1,vs=4096 \
   --cfg ef=200,nq=4096,inflight=6,vs=0 --cfg ef=200,nq=4096,inflight=6,vs=4096 \
   --cfg ef=50,nq=1024,inflight=1,vs=0 --cfg ef=50,nq=1024,inflight=1,vs=4096 \
   --cfg ef=50,nq=1024,inflight=6,vs=0 --cfg ef=50,nq=1024,inflight=6,vs=4096 2>&1 | grep -v Warn | tail -9

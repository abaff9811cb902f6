#!/bin/bash
# round-3 A/B on the GPU box: parity tests, then 16-bit vs 32-bit visited tables at several in-flight depths
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc $?" ; tail -5 gpurun_out/pytest_gpu.log
CF=""
for infl in 1 3 4 5 6 8; do CF="$CF --cfg ef=50,nq=1024,inflight=$infl,vs=0 --cfg ef=50,nq=1024,inflight=$infl,vs=4096"; done
python tools/sweep.py --dtype i8 --steps 100 $CF --cfg ef=50,nq=16384,inflight=1,vs=0 --cfg ef=50,nq=16384,inflight=1,vs=4096 --latency > gpurun_out/ab_i8.txt 2>&1
CF=""
for infl in 1 3 4 5; do CF="$CF --cfg ef=50,nq=1024,inflight=$infl,vs=0 --cfg ef=50,nq=1024,inflight=$infl,vs=4096"; done
python tools/sweep.py --dtype f32 --steps 100 $CF --cfg ef=50,nq=16384,inflight=1,vs=0 --cfg ef=50,nq=16384,inflight=1,vs=4096 --cfg ef=200,nq=1024,inflight=3,vs=0 --cfg ef=200,nq=1024,inflight=3,vs=8192 --cfg ef=200,nq=4096,inflight=1,vs=0 --cfg ef=200,nq=4096,inflight=1,vs=4096 --latency > gpurun_out/ab_f32.txt 2>&1
cat gpurun_out/ab_i8.txt gpurun_out/ab_f32.txt | grep -v Warning

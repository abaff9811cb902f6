#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/pytest_v16.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/pytest_v16.log
python tools/sweep.py --dtype i8 --n 40000000 --steps 20 --fast-build \
   --cfg ef=200,nq=4096,inflight=1,vs=0 --cfg ef=200,nq=4096,inflight=1,vs=4096 \
   --cfg ef=200,nq=4096,inflight=4,vs=0 --cfg ef=200,nq=4096,inflight=4,vs=4096 \
   --cfg ef=50,nq=1024,inflight=1,vs=0 --cfg ef=50,nq=1024,inflight=1,vs=4096 \
   --cfg ef=50,nq=1024,inflight=6,vs=0 --cfg ef=50,nq=1024,inflight=6,vs=4096 > gpurun_out/exp3_i8_40m.txt 2>&1
python tools/sweep.py --dtype i8 --steps 100 --cfg ef=50,nq=1024,inflight=1,vs=0 --cfg ef=50,nq=1024,inflight=1,vs=4096 --cfg ef=50,nq=1024,inflight=6,vs=0 --cfg ef=50,nq=1024,inflight=6,vs=4096 --cfg ef=50,nq=16384,inflight=1,vs=0 > gpurun_out/exp3_i8_10m.txt 2>&1
GRANNE_HIP_LIB=granne_amd/lib/libgranne_hip_phase.so python tools/phase_probe.py --dtype i8 --n 40000000 --fast-build > gpurun_out/phase_i8_verify.txt 2>&1
grep -v Warn gpurun_out/exp3_i8_40m.txt gpurun_out/exp3_i8_10m.txt; cat gpurun_out/phase_i8_verify.txt | grep -v Warn | head -22

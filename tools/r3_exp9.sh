#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_builder.py tests/test_gpu_reorder.py tests/test_gpu_files.py tests/test_gpu_sharded.py -x -q > gpurun_out/pytest_r3b.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/pytest_r3b.log
python tools/sweep.py --dtype f32 --dim 3 --n 2000000 --steps 50 --fast-build --cfg ef=50,nq=1024,inflight=1,vs=0 --cfg ef=50,nq=1024,inflight=4,vs=0 2>&1 | grep -v Warn | tail -3
python tools/sweep.py --dtype f32 --dim 16 --n 2000000 --steps 50 --fast-build --cfg ef=50,nq=1024,inflight=1,vs=0 --cfg ef=50,nq=1024,inflight=4,vs=0 2>&1 | grep -v Warn | tail -3

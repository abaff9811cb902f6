#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_builder.py tests/test_gpu_sharded.py tests/test_gpu_files.py tests/test_gpu_reorder.py -x -q > gpurun_out/pytest_r3.log 2>&1; echo "pytest rc $?"; tail -6 gpurun_out/pytest_r3.log
# wide int8 rows: the register walker against the general one (vs != 0 keeps the register walker; 200-d int8, 2M rows)
python tools/sweep.py --dtype i8 --dim 200 --n 2000000 --steps 50 --fast-build --cfg ef=50,nq=1024,inflight=1,vs=0 --cfg ef=50,nq=1024,inflight=6,vs=0 --cfg ef=50,nq=4096,inflight=1,vs=0 2>&1 | grep -v Warn | tail -4
python tools/sweep.py --dtype i8 --dim 300 --n 2000000 --steps 50 --fast-build --cfg ef=50,nq=1024,inflight=1,vs=0 --cfg ef=50,nq=1024,inflight=6,vs=0 2>&1 | grep -v Warn | tail -3
# max_search beyond the register lists: the exact walker on many blocks
python tools/sweep.py --dtype f32 --n 2000000 --steps 3 --warmup 1 --fast-build --cfg ef=2048,nq=1024,inflight=1,vs=0 --cfg ef=4096,nq=1024,inflight=1,vs=0 2>&1 | grep -v Warn | tail -3

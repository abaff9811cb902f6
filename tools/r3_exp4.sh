#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_bruteforce.py -x -q > gpurun_out/pytest_bf.log 2>&1; echo "pytest rc $?"; tail -12 gpurun_out/pytest_bf.log
rocprofv3 -L 2>/dev/null | grep -i -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" | sort -u | head -30 > gpurun_out/mfma_counters.txt; cat gpurun_out/mfma_counters.txt
python tools/bf_bench.py --dtype f32 2>&1 | grep -v Warn | tail -3
python tools/bf_bench.py --dtype i8 2>&1 | grep -v Warn | tail -3
python tools/bf_bench.py --dtype f32 --dim 200 --n 12500000 2>&1 | grep -v Warn | tail -3

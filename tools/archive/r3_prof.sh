#!/bin/bash
# round-3 profile set: kernel trace + PMC passes of the benchmark's walker (f32, int8), MFMA activity of walker and scan
cd "$(dirname "$0")/.."
BENCH_ARGS="--inflight 1" bash tools/gpu_prof.sh r3_f32 "trace pmc1 pmc3 pmc4 pmc5" > gpurun_out/prof_r3_f32.log 2>&1
BENCH_ARGS="--inflight 1 --dtype i8" bash tools/gpu_prof.sh r3_i8 "trace pmc1 pmc3 pmc4 pmc5" > gpurun_out/prof_r3_i8.log 2>&1
PROF_CMD="python $PWD/tools/bf_bench.py --dtype f32 --reps 2" MFMA_COUNTERS="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES" bash tools/gpu_prof.sh r3_bf_f32 "pmc5" > gpurun_out/prof_r3_bf_f32.log 2>&1
PROF_CMD="python $PWD/tools/bf_bench.py --dtype i8 --reps 2" MFMA_COUNTERS="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_WAVE_CYCLES" bash tools/gpu_prof.sh r3_bf_i8 "pmc5" > gpurun_out/prof_r3_bf_i8.log 2>&1
tail -30 gpurun_out/prof_r3_f32.log; tail -12 gpurun_out/prof_r3_bf_f32.log; tail -8 gpurun_out/prof_r3_bf_i8.log

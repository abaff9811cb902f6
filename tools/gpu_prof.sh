#!/bin/bash
# rocprofv3 passes over a short bench run: kernel trace + stats, then PMC passes (each in its own run;
# never combined with sys/hip/hsa traces). Summaries (not the raw CSVs: they exceed the return limit)
# are left under gpurun_out/prof_<tag>/.   usage: tools/gpu_prof.sh <tag> ["trace pmc1 pmc2 pmc3 pmc4"]
cd "$(dirname "$0")/.."
ROOT=$PWD
TAG=${1:-r1}
PASSES=${2:-"trace pmc1 pmc3 pmc4"}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 10 --cpu-batches 0 --no-recall --no-extras ${BENCH_ARGS:-}"
cd /tmp
for P in $PASSES; do
  case $P in
    trace) rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $BENCH > $OUT/bench_trace.json 2> $OUT/trace.err ;;
    pmc1) rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $OUT/pmc1 -o p -- $BENCH > $OUT/bench_pmc1.json 2> $OUT/pmc1.err ;;
    pmc2) rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INST_CYCLES_VMEM -d $OUT/pmc2 -o p -- $BENCH > $OUT/bench_pmc2.json 2> $OUT/pmc2.err ;;
    pmc3) rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc3 -o p -- $BENCH > $OUT/bench_pmc3.json 2> $OUT/pmc3.err ;;
    pmc4) rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc4 -o p -- $BENCH > $OUT/bench_pmc4.json 2> $OUT/pmc4.err ;;
    # matrix-core activity: of the walker (the gather-dot is not a contraction: expected idle) and, with PROF_CMD set to
    # tools/bf_bench.py, of the brute-force scan (expected busy)
    pmc5) rocprofv3 --kernel-trace --output-format csv --pmc ${MFMA_COUNTERS:-SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU} -d $OUT/pmc5 -o p -- ${PROF_CMD:-$BENCH} > $OUT/bench_pmc5.json 2> $OUT/pmc5.err ;;
  esac
done
cd $ROOT
python tools/prof_summary.py $OUT $OUT/summary.csv | tail -45
cp $OUT/trace/t_kernel_stats.csv $OUT/kernel_stats_full.csv 2>/dev/null
rm -rf $OUT/trace $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4 $OUT/pmc5
ls -la $OUT

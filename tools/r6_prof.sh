#!/bin/bash
# round-6 profile set. Every run is `bench.py --profile-run`: its walker launches are the TIMED shape only (the K timed
# steps as one launch of K x batch walkers, warmup + timed + steady + three event-bracketed repeats), so the kernel-trace
# average IS the duration bench.py's roofline divides by. Passes: kernel trace + stats, FETCH_SIZE, WRITE_SIZE (+ L2 hit /
# miss) -- each in its own run, never with sys/hip/hsa traces. usage: tools/r6_prof.sh [f32 i8 c4 c5 streams]
cd "$(dirname "$0")/.."
ROOT=$PWD
export TMPDIR=/tmp
WHAT=${@:-"f32 i8 c4 c5"}
one() { # tag, walkers per timed launch, passes, bench args...
  local TAG=$1 W=$2 PASSES=$3; shift 3
  local OUT=$ROOT/gpurun_out/prof_$TAG
  mkdir -p $OUT
  local BENCH="python $ROOT/bench.py --profile-run --cpu-batches 0 --no-recall --no-extras $*"
  cd /tmp
  for P in $PASSES; do
    case $P in
      trace) rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $BENCH > $OUT/bench_trace.json 2> $OUT/trace.err ;;
      pmc1) rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $OUT/pmc1 -o p -- $BENCH > $OUT/bench_pmc1.json 2> $OUT/pmc1.err ;;
      pmc3) rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc3 -o p -- $BENCH > $OUT/bench_pmc3.json 2> $OUT/pmc3.err ;;
      pmc4) rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc4 -o p -- $BENCH > $OUT/bench_pmc4.json 2> $OUT/pmc4.err ;;
    esac
  done
  cd $ROOT
  PROF_WALKERS=$W python tools/prof_summary.py $OUT $OUT/summary.csv | tail -30
  cp $OUT/trace/t_kernel_stats.csv $OUT/kernel_stats_full.csv 2>/dev/null
  if [ "$TAG" = "r6_streams" ]; then python tools/trace_gaps.py $OUT/trace 20 > $OUT/trace_gaps.txt 2>&1; fi
  rm -rf $OUT/trace $OUT/pmc1 $OUT/pmc3 $OUT/pmc4
}
for w in $WHAT; do
  case $w in
    f32) one r6_f32 20480 "trace pmc1 pmc3 pmc4" --steps 20 --warmup 5 ;;
    i8)  one r6_i8 20480 "trace pmc1 pmc3 pmc4" --steps 20 --warmup 5 --dtype i8 ;;
    c4)  one r6_c4shard 40960 "trace pmc1 pmc3" --steps 10 --warmup 3 --elements 12500000 --dim 200 --batch 4096 --ef 50 ;;
    c5)  one r6_c5shard 40960 "trace pmc1 pmc3" --steps 10 --warmup 2 --elements 125000000 --dim 100 --dtype i8 --batch 4096 --ef 200 ;;
    # round 3's form for comparison: one batch per call, five streams in flight (kernel intervals overlap)
    streams) one r6_streams 1024 "trace" --steps 20 --warmup 5 --inflight 5 ;;
    # Granne::reorder on the one dataset with structure: the same shape before and after
    latent)   one r6_latent 20480 "trace pmc3 pmc4" --steps 20 --warmup 5 --data latent --ef 30 ;;
    latentre) one r6_latent_reordered 20480 "trace pmc3 pmc4" --steps 20 --warmup 5 --data latent --ef 30 --reorder ;;
  esac
done

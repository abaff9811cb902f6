#!/bin/bash
# final evidence of the round: profiles (trace + PMC), the default bench line, the partitioned line, the fixture harness on the GPU
mkdir -p gpurun_out
bash tools/r3_prof.sh > gpurun_out/r3_prof_tail.log 2>&1
S=$(date +%s); python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc $? wall $(( $(date +%s) - S )) s"
python bench.py --mode partitioned --shards-per-gpu 2 --steps 20 --warmup 5 > gpurun_out/bench_part.json 2> gpurun_out/bench_part.err; echo "part rc $?"
python bench.py --force-partitioned --no-extras --steps 20 --warmup 5 > gpurun_out/bench_forcepart.json 2> gpurun_out/bench_forcepart.err; echo "forcepart rc $?"; tail -2 gpurun_out/bench_forcepart.err
python oracle/ref_fixtures/emulate.py /tmp/fx > /dev/null 2>&1 && GRANNE_REF_FIXTURES=/tmp/fx python -m pytest tests/test_ref_fixtures.py -q 2>&1 | tail -2
python tools/show_bench.py gpurun_out/bench_default.json 2>&1 | head -12

#!/bin/bash
# final evidence of the round: the whole GPU suite, profiles (trace + PMC), the default bench line, the partitioned lines, the fixture harness
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/pytest_gpu.log
bash tools/r3_prof.sh > gpurun_out/r3_prof_tail.log 2>&1
python tools/prof_to_traffic.py
S=$(date +%s); python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc $? wall $(( $(date +%s) - S )) s"
python bench.py --mode partitioned --shards-per-gpu 2 --steps 20 --warmup 5 > gpurun_out/bench_part.json 2> gpurun_out/bench_part.err; echo "part rc $?"
python bench.py --force-partitioned --no-extras --steps 20 --warmup 5 > gpurun_out/bench_forcepart.json 2> gpurun_out/bench_forcepart.err; echo "forcepart rc $?"
python oracle/ref_fixtures/emulate.py /tmp/fx > /dev/null 2>&1 && GRANNE_REF_FIXTURES=/tmp/fx python -m pytest tests/test_ref_fixtures.py -q 2>&1 | tail -2
python tools/show_bench.py gpurun_out/bench_default.json 2>&1 | head -12
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json

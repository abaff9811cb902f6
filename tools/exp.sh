P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]), d["sequential"]["value"], d["roofline"]["launch_ms_mean"], d["roofline"]["frac"], d.get("recall_at_10"), d["config"]["graph"]["build_s"], d["roofline"]["per_query"], d.get("cpu_baseline",{}).get("value"))'
echo "== builder + fullsize + sharded tests"; timeout 900 python -m pytest tests/test_gpu_builder.py tests/test_gpu_sharded.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | tail -3
echo "== default build params (ms=200, reinsert), inflight 2"; GRANNE_BENCH_VERBOSE=1 python bench.py --steps 24 --build-max-search 200 --build-reinsert 1 --cpu-batches 8 2>gpurun_out/b_full.err | tee gpurun_out/b_full.json | python -c "$P"; tail -3 gpurun_out/b_full.err
echo "== ms=100 no reinsert"; python bench.py --steps 24 --build-max-search 100 --cpu-batches 0 2>/dev/null | python -c "$P"
echo "== inflight 3 again"; python bench.py --steps 30 --inflight 3 --cpu-batches 0 --no-recall 2>/dev/null | python -c "$P"
echo "== inflight 2 again"; python bench.py --steps 30 --inflight 2 --cpu-batches 0 --no-recall 2>/dev/null | python -c "$P"

P='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][0]); print(round(d["value"]), d["sequential"]["value"], d["slow_path_queries"], d["lds_retry_queries"], d["roofline"]["launch_ms_mean"], d["roofline"]["launch_ms_min"], d["roofline"]["frac"], d.get("cpu_baseline",{}).get("gpu_matches_oracle"))'
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3
echo "== ms=200 reinsert"; python bench.py --steps 40 --build-max-search 200 --build-reinsert 1 --cpu-batches 8 --no-recall 2>/dev/null | python -c "$P"
echo "== ms=50"; python bench.py --steps 40 --cpu-batches 0 --no-recall 2>/dev/null | python -c "$P"

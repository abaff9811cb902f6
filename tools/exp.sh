P='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][0]); print(round(d["value"]), d["sequential"]["value"], d["slow_path_queries"], d["roofline"]["launch_ms_mean"], d["roofline"]["launch_ms_min"], d["roofline"]["frac"], d.get("recall_at_10"), d["config"]["graph"]["build_s"], d["roofline"]["per_query"])'
echo "== ms=200 reinsert"; python bench.py --steps 40 --build-max-search 200 --build-reinsert 1 --cpu-batches 0 --no-recall 2>/dev/null | python -c "$P"
echo "== ms=50"; python bench.py --steps 40 --cpu-batches 0 --no-recall 2>/dev/null | python -c "$P"
echo "== ms=50 ef=100"; python bench.py --steps 40 --ef 100 --cpu-batches 0 --no-recall 2>/dev/null | python -c "$P"

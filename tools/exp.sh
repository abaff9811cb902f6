P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]), d["roofline"]["launch_ms_mean"], d["roofline"]["frac"], d.get("recall_at_10"), d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline",{}).get("gpu_matches_oracle"))'
echo "== new tests"; timeout 900 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_fullsize.py -m gpu -q -x -s 2>&1 | grep -v "amdgpu.ids" | tail -12
echo "== batch 1024"; python bench.py --steps 20 --cpu-batches 0 --no-recall 2>/dev/null | python -c "$P"
echo "== batch 4096"; python bench.py --steps 10 --batch 4096 --cpu-batches 0 --no-recall 2>/dev/null | python -c "$P"
echo "== batch 4096, 2 walkers/CU"; GRANNE_HIP_LDS_PAD=36000 python bench.py --steps 10 --batch 4096 --cpu-batches 0 --no-recall 2>/dev/null | python -c "$P"
echo "== batch 4096, 3 walkers/CU"; GRANNE_HIP_LDS_PAD=14000 python bench.py --steps 10 --batch 4096 --cpu-batches 0 --no-recall 2>/dev/null | python -c "$P"
echo "== i8 10M"; python bench.py --steps 20 --dtype i8 --cpu-batches 8 2>/dev/null | python -c "$P"

"""Host-side pieces of bench.py that need no GPU: the batches-per-call rule, the workload labels, the source stamp."""
import os
import sys
import types

import pytest  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_batches_per_call_divide_the_timed_steps():
    auto = types.SimpleNamespace(batches_per_call=0)
    # the driver's K = 20 (and the sub-records' K = 10): all steps in one call = one launch
    assert bench.auto_group(auto, 20) == 20
    assert bench.auto_group(auto, 10) == 10
    assert bench.auto_group(auto, 32) == 32
    # more steps than one launch carries (32 batches): the largest divisor of K that fits, so every call has one shape
    assert bench.auto_group(auto, 64) == 32
    assert bench.auto_group(auto, 100) == 25
    assert bench.auto_group(auto, 37) == 32  # no divisor in [16, 32]: calls of 32, the last one carries 5
    assert bench.auto_group(auto, 1000) == 25
    # an explicit --batches-per-call wins, bounded by K
    assert bench.auto_group(types.SimpleNamespace(batches_per_call=4), 20) == 4
    assert bench.auto_group(types.SimpleNamespace(batches_per_call=1), 20) == 1
    assert bench.auto_group(types.SimpleNamespace(batches_per_call=50), 20) == 20


def test_workload_label_names_baseline_configs_only_for_their_exact_shape():
    c2 = bench.workload_label(10_000_000, 100, "f32", "uniform", 1024, 50, 10)
    assert c2.startswith("C2 (BASELINE.json configs[1])")
    c3 = bench.workload_label(10_000_000, 100, "i8", "uniform", 1024, 50, 10)
    assert c3.startswith("C3 (BASELINE.json configs[2])")
    other = bench.workload_label(1_000_000, 100, "f32", "uniform", 1024, 50, 10)
    assert not other.startswith("C2") and "1000000 x 100-d f32" in other
    assert not bench.workload_label(10_000_000, 100, "f32", "uniform", 1024, 50, 10, default_graph=False).startswith("C2")
    latent = bench.workload_label(10_000_000, 100, "f32", "latent", 1024, 30, 10)
    assert "latent" in latent and not latent.startswith("C2")


def test_traffic_in_profiles_is_stamped_with_the_current_kernel_sources():
    """bench.py quotes roofline.traffic only while profiles/pmc_traffic.json carries the hash of the walker's sources:
    a commit that changes them without retaking the PMC passes shows up here."""
    import json
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
        d = json.load(f)
    sha = bench.csrc_sha()
    stamped = {k: v.get("csrc_sha") for k, v in d.items() if isinstance(v, dict)}
    # (the timed shape of round 4: K batches per launch, keys end in |g<K>; round 3's one-batch entries stay as history)
    headline = [k for k in stamped if k.startswith("10000000|100|") and "|g" in k]
    assert headline, stamped
    stale = [k for k in headline if stamped[k] != sha]
    if stale:  # not an error of the product: the bench line then carries traffic = null and says why
        import pytest
        pytest.skip("profiles/pmc_traffic.json was measured on other kernel sources (%s != %s): retake the PMC passes "
                    "(tools/r6_prof.sh + tools/r6_traffic.py)" % (stamped[stale[0]], sha))


def test_ground_truth_rows_merge_by_distance_then_id():
    """bench.py's recall over a partitioned index: the per-shard exact top-k lists ([nq, shards * k] distances and global
    ids) merged into the k smallest by (distance, id)."""
    import numpy as np
    rng = np.random.default_rng(5)
    nq, m, k = 40, 24, 10
    d = rng.integers(0, 6, (nq, m)).astype(np.float32) / 8  # many ties on distance
    ids = np.stack([rng.permutation(1000)[:m] for _ in range(nq)]).astype(np.int64)
    gi, gd = bench._merge_rows_numpy(d, ids, k)
    for q in range(nq):
        want = sorted(zip(d[q].tolist(), ids[q].tolist()))[:k]
        assert [(float(a), int(b)) for a, b in zip(gd[q], gi[q])] == want


def _fake_full_record():
    """A record with every part the r4 line carried (profiles/r4_bench_default.json, 26 KB), and worse: long notes, a NaN."""
    import json
    with open(os.path.join(ROOT, "profiles", "r4_bench_default.json")) as f:
        d = json.load(f)
    d["c5_shard"] = dict(d["c4_shard"])
    d["partitioned"] = dict(d["c4_shard"])
    d["cpu_baseline"]["sample"] = "x" * 4000
    d["config"]["parallelism"] = "y" * 3000
    d["roofline"]["whole_timed_region_frac"] = float("nan")
    d["steady"]["value"] = float("inf")
    return d


def test_stdout_line_stays_under_the_drivers_tail_and_parses_strictly():
    """The driver keeps the last 8 KB of stdout and parses ONE JSON line from it (round 4's 26 KB line came back
    `parsed: null`)."""
    import json
    line = bench.compact_line(_fake_full_record(), "bench_extras.json")
    assert len(line.encode()) < 8192 and len(line.encode()) <= bench.LINE_LIMIT and "\n" not in line

    def no_constants(x):
        raise ValueError("not JSON: " + x)
    d = json.loads(line, parse_constant=no_constants)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["config"]["workload"].startswith("C2")
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "one_batch_per_launch"):
        assert key in d["roofline"], key
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in d["cpu_baseline"], key
    assert d["roofline"]["whole_timed_region_frac"] is None and d["steady"]["value"] is None  # (NaN / inf never reach the line)
    for sub in ("int8", "secondary", "c4_shard", "c5_shard"):
        assert {"workload", "value", "frac", "cpu", "bit_exact"} <= set(d[sub]), sub
    # and a record far beyond anything bench.py produces still fits: optional parts are dropped, the contract stays
    big = _fake_full_record()
    big["ef_sweep"] = big["ef_sweep"] * 40
    big["config"]["layers"] = list(range(400))
    line = bench.compact_line(big, None)
    assert len(line.encode()) <= bench.LINE_LIMIT
    d = json.loads(line, parse_constant=no_constants)
    assert "roofline" in d and "cpu_baseline" in d and "value" in d


def test_gpus_n_without_a_launcher_never_measures_one_gpu_silently():
    """`python bench.py --gpus 8` with WORLD_SIZE unset starts the ranks itself (torch.distributed.run); on a node with
    fewer GPUs it refuses (exit 2) instead of reporting a one-GPU number under n_gpus 8."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.device_count() < 2:
        assert r.returncode == 2 and "refusing" in r.stderr and r.stdout.strip() == ""
    # the launch line itself
    seen = {}

    def fake_exec(file, argv, env):
        seen["argv"] = argv
        raise SystemExit(0)
    real, real_count = os.execvpe, torch.cuda.device_count
    os.execvpe, torch.cuda.device_count = fake_exec, lambda: 8
    try:
        with pytest.raises(SystemExit):
            bench.self_launch(types.SimpleNamespace(gpus=8), ["--gpus", "8", "--steps", "20", "--warmup", "5"])
    finally:
        os.execvpe, torch.cuda.device_count = real, real_count
    a = seen["argv"]
    assert a[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and a[a.index("--nproc-per-node") + 1] == "8"
    assert a[a.index("--master-addr") + 1] == "127.0.0.1" and a[-6:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]


def test_the_line_of_an_n_gpu_run_carries_roofline_and_cpu_baseline():
    """N > 1 (the driver's SCALE runs): rank 0 still takes recall and the CPU baseline after the timed region (the other
    ranks wait on the rendezvous store meanwhile), so the N-GPU line has `roofline` AND `cpu_baseline` like the one-GPU
    line; the sweep, the latency probe and the sub-records stay with N = 1. Assembled here with a stand-in for the GPU."""
    import json
    import numpy as np
    import torch
    saved = sys.argv
    sys.argv = ["bench.py", "--gpus", "2"]
    try:
        args = bench.parse()
    finally:
        sys.argv = saved
    plan = bench.rank0_plan(2, args)
    assert plan["recall"] and plan["cpu_baseline"] and not plan["ef_sweep"] and not plan["extras"] and not plan["cpu_thread_sweep"]
    one = bench.rank0_plan(1, args)
    assert one["recall"] and one["cpu_baseline"] and one["ef_sweep"] and one["extras"] and one["cpu_thread_sweep"]

    nq, k, dim, nb = args.batch, args.k, args.dim, args.warmup + args.steps
    calls = []

    class FakeBench:
        world, rank = 2, 0

        def __init__(self):
            self.torch = torch

        def ground_truth(self, index, q0, k, dtype, timing=None, n=None):
            calls.append("ground_truth")
            timing.update({"ms": 17.0, "value": 60000.0, "frac": 0.76, "bound": "mfma", "peak": 157.3, "unit": "TFLOP/s"})
            return np.zeros((q0.shape[0], k), np.int64)

        @staticmethod
        def recall(gt, got, k):
            return 0.02

        def host_index(self, elements, builder, order=None):
            calls.append("host_index")
            return object()

        def cpu_baseline(self, oix, h_q, ef, k, g_ids, g_d, single_thread_queries=256, sweep=True):
            calls.append(("cpu_baseline", sweep, h_q.shape[0]))
            return {"value": 45000.0, "unit": "queries/s", "cores": 16, "kind": "port", "sample": "stand-in",
                    "gpu_matches_oracle": {"ids_bit_exact": True, "dists_bit_exact": True, "queries_checked": int(h_q.shape[0])}}

        def check_scan(self, *a, **kw):
            calls.append("check_scan")

    m = {"ids": torch.zeros((nb, nq, k), dtype=torch.int64), "dists": torch.zeros((nb, nq, k), dtype=torch.float32)}
    queries = torch.zeros((nb * nq, dim), dtype=torch.float32)
    value = 2 * 7.6e6
    out = {"metric": "queries/sec (recall@10 alongside), 10M x 100-d angular, batch=1024", "value": value, "unit": "queries/s",
           "n_gpus": 2, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 0.134, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": bench.workload_label(args.n, dim, "f32", "uniform", nq, args.ef, k)},
           "roofline": {"bound": "hbm", "achieved": 6200.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.775, "traffic": None}}
    bench.rank0_after_the_timed_region(FakeBench(), args, out, None, None, None, queries, m, None, value, 20)
    assert "ground_truth" in calls and "host_index" in calls and ("cpu_baseline", False, min(args.cpu_batches, args.steps) * nq) in calls
    assert out["cpu_baseline"]["value"] == 45000.0 and out["recall_at_10"] == 0.02 and out["speedup_vs_cpu"] == round(value / 45000.0, 2)
    for absent in ("ef_sweep", "latency_nq1", "int8", "secondary", "c4_shard", "c5_shard", "recall_target"):
        assert absent not in out, absent
    line = json.loads(bench.compact_line(out, None))
    assert line["n_gpus"] == 2
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in line["roofline"], key
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in line["cpu_baseline"], key


def test_the_secondary_generators_are_named_in_their_labels():
    mix = bench.workload_label(10_000_000, 100, "f32", "mixture", 1024, 100, 10)
    assert "mixture of %d Gaussians" % bench.MIX_CENTERS in mix and not mix.startswith("C2")


def test_compact_line_reads_both_forms_of_the_parity_record():
    """`gpu_matches_oracle` of one index (ids + distance bits) and of a partitioned run (every shard + the merged result):
    the compact line's `bit_exact` must say true for both when they are (round 6: the partitioned form read as false)."""
    one = {"ids_bit_exact": True, "dists_bit_exact": True, "queries_checked": 16}
    part = {"shards_checked_against_oracle_per_rank": 4, "those_shards_bit_exact_on_every_rank": True,
            "merged_equals_numpy_merge_of_shard_results": True, "merged_equals_numpy_merge_of_ORACLE_shard_results": True,
            "queries_checked": 4096}
    assert bench._oracle_match(one) and bench._oracle_match(part)
    assert not bench._oracle_match(dict(one, dists_bit_exact=False))
    assert not bench._oracle_match(dict(part, merged_equals_numpy_merge_of_ORACLE_shard_results=False))
    assert not bench._oracle_match({})
    # N > 1 ranks: no rank's oracle holds every shard (the ORACLE merge is not made); every rank's shard equal to the oracle's and
    # the merged result equal to the numpy merge of the shards' results is the whole evidence there is
    many = dict(part, merged_equals_numpy_merge_of_ORACLE_shard_results=None)
    assert bench._oracle_match(many)
    assert not bench._oracle_match(dict(many, merged_equals_numpy_merge_of_shard_results=False))
    assert not bench._oracle_match(dict(many, those_shards_bit_exact_on_every_rank=False))
    cb = {"value": 1.0, "unit": "queries/s", "cores": 16, "kind": "port", "sample": "s", "gpu_matches_oracle": part}
    assert bench._compact_cpu(cb)["gpu_matches_oracle"] == {"bit_exact": True, "queries": 4096}
    assert bench._compact_sub({"workload": "w", "value": 1.0, "cpu_baseline": cb})["bit_exact"] is True

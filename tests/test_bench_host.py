"""Host-side pieces of bench.py that need no GPU: the in-flight rule, the workload labels, the source stamp."""
import os
import sys
import types

import pytest  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_batches_in_flight_divide_the_timed_steps():
    auto = types.SimpleNamespace(inflight=0)
    # the driver's K = 20: five f32 / ten int8 batches in flight (every round of the K steps is full)
    assert bench.auto_inflight(auto, "f32", 20) == 5
    assert bench.auto_inflight(auto, "i8", 20) == 10
    # K a multiple of the saturating count: that count; no divisor within a quarter below it: that count too
    assert bench.auto_inflight(auto, "f32", 12) == 6
    assert bench.auto_inflight(auto, "i8", 24) == 12
    assert bench.auto_inflight(auto, "f32", 7) == 6
    assert bench.auto_inflight(auto, "i8", 13) == 12
    # the sub-records' K = 10
    assert bench.auto_inflight(auto, "f32", 10) == 5
    assert bench.auto_inflight(auto, "i8", 10) == 10
    # an explicit --inflight wins
    assert bench.auto_inflight(types.SimpleNamespace(inflight=3), "i8", 20) == 3
    assert bench.auto_inflight(types.SimpleNamespace(inflight=1), "f32", 20) == 1


def test_workload_label_names_baseline_configs_only_for_their_exact_shape():
    c2 = bench.workload_label(10_000_000, 100, "f32", "uniform", 1024, 50, 10)
    assert c2.startswith("C2 (BASELINE.json configs[1])")
    c3 = bench.workload_label(10_000_000, 100, "i8", "uniform", 1024, 50, 10)
    assert c3.startswith("C3 (BASELINE.json configs[2])")
    other = bench.workload_label(1_000_000, 100, "f32", "uniform", 1024, 50, 10)
    assert not other.startswith("C2") and "1000000 x 100-d f32" in other
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
    headline = [k for k in stamped if k.startswith("10000000|100|")]
    assert headline, stamped
    stale = [k for k in headline if stamped[k] != sha]
    if stale:  # not an error of the product: the bench line then carries traffic = null and says why
        import pytest
        pytest.skip("profiles/pmc_traffic.json was measured on other kernel sources (%s != %s): retake the PMC passes "
                    "(tools/r3_prof.sh + tools/prof_to_traffic.py)" % (stamped[stale[0]], sha))

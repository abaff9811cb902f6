"""Partitioned mode on the GPU (one device stands in for the ranks): per-shard HIP searches, the
HIP merge kernel behind granne_hip_merge_topk_device, checked against per-shard oracle searches +
the numpy merge."""
import numpy as np
import pytest

from oracle.merge import merge_topk_numpy

pytestmark = pytest.mark.gpu

from tests.conftest import random_floats  # noqa: E402


@pytest.mark.parametrize("int8", [False, True])
@pytest.mark.parametrize("shards,k", [(2, 10), (8, 10), (3, 1), (8, 64)])
def test_sharded_search_and_merge(oracle, int8, shards, k):
    import torch
    import granne_amd
    from granne_amd import sharded
    rng = np.random.default_rng(shards * 100 + k)
    raw = random_floats(rng, 4000, 32)
    el = oracle.quantize(raw) if int8 else oracle.normalize_f32(raw)
    q = oracle.quantize(random_floats(rng, 50, 32)) if int8 else oracle.normalize_f32(random_floats(rng, 50, 32))
    bounds = sharded.shard_bounds(len(el), shards)
    g_ids, g_ds, g_cnt, o_ids, o_ds, o_cnt = [], [], [], [], [], []
    for lo, hi in bounds:
        part = np.ascontiguousarray(el[lo:hi])
        oix = oracle.build_index(part, num_neighbors=10, max_search=20, n_threads=0)
        gix = granne_amd.Granne("angular_int" if int8 else "angular", part, oix.layers)
        sg = sharded.ShardedGranne(gix, lo)
        i, d, c = sg._gpu_local_search(q, 70, k)
        g_ids.append(i); g_ds.append(d); g_cnt.append(c)
        oi, od, oc, _ = oix.search_batch(q, 70, k)
        o_ids.append(oi); o_ds.append(od); o_cnt.append(oc)
    offsets = [b[0] for b in bounds]
    sg = sharded.ShardedGranne(None, 0)
    m_ids, m_ds, m_cnt = sg._gpu_merge(torch.stack(g_ids), torch.stack(g_ds), torch.stack(g_cnt), offsets, k)
    torch.cuda.synchronize()
    want = merge_topk_numpy(np.stack(o_ids), np.stack(o_ds), np.stack(o_cnt), offsets, k)
    assert (m_cnt.cpu().numpy().astype(np.uint32) == want[2]).all()
    assert (m_ids.cpu().numpy().astype(np.uint64) == want[0]).all()
    assert m_ds.cpu().numpy().tobytes() == want[1].tobytes()

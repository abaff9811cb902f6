"""The GPU builder inserts in batches (a batch searches a frozen graph) where the
reference inserts one element at a time under per-node locks (src/index/mod.rs:757-782). The batched
schedule is deterministic and restated by the oracle (batch_max > 0); this test guards the property
that it does not cost graph quality: recall@10 of searches on the batched graph stays within a small
tolerance of the graph the reference's `singlethreaded` order builds (batch_max = 0)."""
import numpy as np

from tests.conftest import random_floats


def _recall(ix, queries, gt, ef, k=10):
    ids, _, cnt, _ = ix.search_batch(queries, ef, k)
    return float(np.mean([len(set(gt[i]) & set(ids[i, :cnt[i]].tolist())) / k for i in range(len(queries))]))


def test_batched_insertion_keeps_recall(oracle):
    rng = np.random.default_rng(77)
    n, dim, nq, k = 12000, 32, 200, 10
    el = oracle.normalize_f32(random_floats(rng, n, dim))
    q = oracle.normalize_f32(random_floats(rng, nq, dim))
    gt = np.argsort(-(q @ el.T), axis=1, kind="stable")[:, :k]
    seq = oracle.build_index(el, num_neighbors=20, max_search=60, n_threads=1, batch_max=0)
    # the GPU builder's defaults: batches of clamp(nodes_in_graph / 8, 1, 65536)
    bat = oracle.build_index(el, num_neighbors=20, max_search=60, n_threads=0, batch_max=65536, batch_div=8)
    for ef in (20, 60):
        rs, rb = _recall(seq, q, gt, ef), _recall(bat, q, gt, ef)
        assert rb >= rs - 0.03, (ef, rs, rb)
    assert _recall(bat, q, gt, 60) > 0.85

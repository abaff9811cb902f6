"""Generates the golden fixtures under tests/golden/ from the CPU oracle.

    python tests/golden/make_golden.py

The reference (Rust) cannot be built or imported in this environment, so these vectors are
outputs of oracle/granne_oracle.c (itself pinned by the reference's KATs/property tests and
diffed against oracle/pyref.py), not of the reference binary. They freeze today's answers so
that both the oracle and the HIP path are regression-checked against the same bytes.
Everything is derived from fixed seeds; rerunning this script must reproduce the files exactly.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SEED = 0x6772616E6E65  # "granne"

CASES = [
    # name, dtype, n, dim, num_neighbors, build max_search, nq, [(max_search, k)]
    ("f32_d100", "f32", 3000, 100, 30, 40, 48, [(1, 1), (10, 10), (50, 10), (64, 64), (100, 20)]),
    ("f32_d200", "f32", 1500, 200, 20, 30, 32, [(50, 10), (200, 10)]),
    ("f32_d28", "f32", 1500, 28, 20, 20, 32, [(10, 1), (50, 10)]),
    ("i8_d100", "i8", 3000, 100, 30, 40, 48, [(1, 1), (50, 10), (128, 30)]),
    ("i8_d32", "i8", 500, 32, 20, 20, 32, [(10, 1), (50, 10)]),
]


def make(name, dtype, n, dim, nn, ms, nq, searches):
    raw = orc.synth_rows(SEED, 0, n, dim)
    qraw = orc.synth_rows(SEED + 1, 0, nq, dim)
    if dtype == "f32":
        el, q = orc.normalize_f32(raw), orc.normalize_f32(qraw)
    else:
        el, q = orc.quantize(raw), orc.quantize(qraw)
    ix = orc.build_index(el, num_neighbors=nn, max_search=ms, n_threads=1)
    out = {"elements": el, "queries": q, "n_layers": np.int64(len(ix.layers))}
    for l, layer in enumerate(ix.layers):
        out["layer%d" % l] = layer
    for ms_, k in searches:
        ids, ds, cnt, ctr = ix.search_batch(q, ms_, k, n_threads=1)
        out["ids_%d_%d" % (ms_, k)] = ids
        out["dists_%d_%d" % (ms_, k)] = ds
        out["counts_%d_%d" % (ms_, k)] = cnt
        out["stats_%d_%d" % (ms_, k)] = ctr
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, [l.shape for l in ix.layers], os.path.getsize(os.path.join(HERE, name + ".npz")))
    return ix


def make_reorder(indexes):
    """Granne::reorder (src/index/reorder.rs): the permutation compute_order gives for each fixture index."""
    out = {name: ix.compute_order(n_threads=1).astype(np.uint32) for name, ix in indexes.items() if len(ix.layers) >= 2}
    np.savez_compressed(os.path.join(HERE, "reorder_orders.npz"), **out)
    print("reorder_orders", sorted(out), os.path.getsize(os.path.join(HERE, "reorder_orders.npz")))


def make_batched_build():
    """The graph of the GPU builder's schedule (gro_build_config.batch_max) on the f32_d28 rows."""
    el = orc.normalize_f32(orc.synth_rows(SEED, 0, 1500, 28))
    ix = orc.build_index(el, num_neighbors=20, max_search=20, batch_max=64, batch_div=8, n_threads=1)
    out = {"n_layers": np.int64(len(ix.layers))}
    for l, layer in enumerate(ix.layers):
        out["layer%d" % l] = layer
    np.savez_compressed(os.path.join(HERE, "build_batched_f32_d28.npz"), **out)
    print("build_batched_f32_d28", [l.shape for l in ix.layers], os.path.getsize(os.path.join(HERE, "build_batched_f32_d28.npz")))


if __name__ == "__main__":
    orc.build()
    make_reorder({c[0]: make(*c) for c in CASES})
    make_batched_build()

#!/usr/bin/env python
"""Writes fixtures in the format of oracle/ref_fixtures/src/main.rs -- but made by the ORACLE, not by
the reference. They pin nothing; they exist to exercise tests/test_ref_fixtures.py (its readers, its
comparisons) on a box without a Rust toolchain:

    python oracle/ref_fixtures/emulate.py /tmp/fx && GRANNE_REF_FIXTURES=/tmp/fx python -m pytest tests/test_ref_fixtures.py

Never point it at oracle/_ref/fixtures."""
import json
import os
import struct
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import fileformat as off  # noqa: E402
from oracle import oracle as orc  # noqa: E402

SEED = 0x6772616E6E65
# (name, n, dim, nq, num_neighbors, build max_search, searches, distinct rows, reorder) -- as in src/main.rs
CASES = [("f32_d100", 3000, 100, 64, 30, 50, [(1, 1), (50, 10), (200, 50), (300, 300), (1024, 10)], 3000, True),
         ("f32_d28", 700, 28, 32, 20, 30, [(5, 5), (40, 10)], 700, False),
         ("ties_d32", 1500, 32, 48, 20, 30, [(1, 1), (20, 10), (60, 60), (130, 20)], 500, False)]


def write_searches(d, prefix, ix, q, searches):
    files = []
    for m_, k in searches:
        fn = "%ssearch_ms%d_k%d.bin" % (prefix, m_, k)
        with open(os.path.join(d, fn), "wb") as f:
            for i in range(len(q)):
                res = ix.search(q[i], m_, k)
                f.write(struct.pack("<I", len(res)))
                for id_, dist in res:
                    f.write(struct.pack("<QI", id_, int(np.float32(dist).view(np.uint32))))
        files.append({"file": fn, "max_search": m_, "num_neighbors": k})
    return files


def main(out):
    assert "_ref" not in os.path.abspath(out), "emulated fixtures must not sit where the real ones go"
    orc.build()
    for name, n, dim, nq, nn, ms, searches, distinct, reorder in CASES:
        for i8 in (False, True):
            d = os.path.join(out, name + ("_i8" if i8 else ""))
            os.makedirs(d, exist_ok=True)
            prep = orc.quantize if i8 else orc.normalize_f32
            el = np.ascontiguousarray(prep(orc.synth_rows(SEED, 0, distinct, dim))[np.arange(n) % distinct])
            q = prep(orc.synth_rows(SEED + 1, 0, nq, dim))
            ix = orc.build_index(el, num_neighbors=nn, max_search=ms, reinsert_elements=True, n_threads=1, batch_max=0)
            open(os.path.join(d, "elements.bin"), "wb").write(off.write_elements(el))
            open(os.path.join(d, "index.granne"), "wb").write(off.write_index(ix.layers))
            q.tofile(os.path.join(d, "queries.bin"))
            files = write_searches(d, "", ix, q, searches)
            rfiles = []
            if reorder:
                order = ix.compute_order(n_threads=1)
                order.astype("<u8").tofile(os.path.join(d, "reorder_order.bin"))
                rix = ix.reordered(order)
                open(os.path.join(d, "reordered_elements.bin"), "wb").write(off.write_elements(rix.elements))
                rfiles = write_searches(d, "reordered_", rix, q, searches)
            np.array([np.float32(orc.dist(el[i], q[i])).view(np.uint32) for i in range(nq)], np.uint32).tofile(
                os.path.join(d, "dists.bin"))
            json.dump({"case": os.path.basename(d), "element_type": "angular_int" if i8 else "angular", "n": n, "dim": dim,
                       "nq": nq, "seed": SEED, "num_neighbors": nn, "build_max_search": ms, "reinsert_elements": True,
                       "layer_multiplier": 15.0, "feature": "EMULATED BY THE ORACLE -- not a reference run",
                       "layer_lens": [int(l.shape[0]) for l in ix.layers], "searches": files, "distinct": distinct,
                       "reordered_searches": rfiles},
                      open(os.path.join(d, "manifest.json"), "w"))
            print("wrote", d)


if __name__ == "__main__":
    main(sys.argv[1])

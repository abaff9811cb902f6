"""Reference merge of per-shard top-k lists (TEST INFRASTRUCTURE, like the rest of oracle/)."""
import numpy as np


def merge_topk_numpy(ids, dists, counts, offsets, k):
    """Reference merge (numpy): ids/dists [G][nq][k], counts [G][nq] -> top-k by (dist, global id)."""
    G, nq, _ = ids.shape
    out_ids = np.full((nq, k), np.iinfo(np.uint64).max, np.uint64)
    out_d = np.full((nq, k), np.inf, np.float32)
    out_c = np.zeros(nq, np.uint32)
    for q in range(nq):
        cand = []
        for g in range(G):
            for j in range(int(counts[g, q])):
                cand.append((float(dists[g, q, j]), int(ids[g, q, j]) + int(offsets[g])))
        cand.sort()
        cand = cand[:k]
        out_c[q] = len(cand)
        for j, (d, i) in enumerate(cand):
            out_ids[q, j] = i
            out_d[q, j] = d
    return out_ids, out_d, out_c

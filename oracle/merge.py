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


def packed_bytes(nq, k):
    """granne_hip_packed_topk_bytes: [nq*k u64 ids][nq*k f32 dists][nq u32 counts], padded to 16."""
    return (nq * k * 12 + nq * 4 + 15) & ~15


def pack_topk(ids, dists, counts):
    """One shard's results as the packed byte buffer the exchange step moves."""
    nq, k = ids.shape
    buf = np.zeros(packed_bytes(nq, k), np.uint8)
    buf[:nq * k * 8] = np.ascontiguousarray(ids, np.uint64).view(np.uint8).reshape(-1)
    buf[nq * k * 8:nq * k * 12] = np.ascontiguousarray(dists, np.float32).view(np.uint8).reshape(-1)
    buf[nq * k * 12:nq * k * 12 + nq * 4] = np.ascontiguousarray(counts, np.uint32).view(np.uint8).reshape(-1)
    return buf


def unpack_topk(buf, nq, k):
    buf = np.ascontiguousarray(buf, np.uint8).reshape(-1)
    ids = buf[:nq * k * 8].view(np.uint64).reshape(nq, k)
    dists = buf[nq * k * 8:nq * k * 12].view(np.float32).reshape(nq, k)
    counts = buf[nq * k * 12:nq * k * 12 + nq * 4].view(np.uint32)
    return ids, dists, counts

#!/usr/bin/env python
"""Lane-level CPU model of walk_fast.h's ranked merge (lists of up to 17 slots) against the reference's two heaps.

Mirrors FastWalker::search_layer / rank_candidates / merge_ranked / filter_mask operation for operation: 64 lanes, S slots
of 64-bit keys (dist bits | id << 1 | expanded), the LDS image of CAP + 32 keys, candidates in the odd lanes, passes of 32
neighbors (WIDE rows of up to 64 ids take two), the next node chosen before the scatter, the tie test on place CAP.
Distances are float32 bit patterns (non-negative floats order like their bits).  Not product code.
"""
import random
import struct
import sys

import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from model_unified import reference  # noqa: E402

KEY_INF = (1 << 64) - 1
NONE = 0xFFFFFFFF


def f32bits(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


def ranked(adj, dbits, ep, ef, S, wide=False):
    """-> (results [(dist bits, id)], (n_dist, n_expand, n_adj)) or (None, None) when the walk bails (tie at the boundary)."""
    CAP = 64 * S
    key = [[KEY_INF] * 64 for _ in range(S)]           # key[s][lane]
    img = [KEY_INF] * CAP + [0x123456789] * 32  # the image: CAP keys + what an expansion pushes off the end (stale between merges)

    def wkey(d, i):
        return (d << 32) | (i << 1)

    def first_unexpanded():
        for s in range(S):
            for l in range(64):
                if key[s][l] & 1 == 0:
                    return s * 64 + l
        return None

    def at(e):
        return key[e >> 6][e & 63]

    def count_closer(hi):
        return sum(1 for s in range(S) for l in range(64) if (key[s][l] >> 32) < hi)

    def nth_expanded(n):
        c = 0
        for s in range(S):
            for l in range(64):
                k = key[s][l]
                if (k & 1) and (k >> 32) != NONE:
                    c += 1
                    if c == n:
                        return s * 64 + l
        return None

    xkey = wkey(dbits[ep], ep) | 1
    key[0][0] = xkey
    img[0] = xkey
    twin_rows = any(len(set(r)) != len(r) for r in adj)  # LAYER_TWIN_ROWS: found at upload
    theta = (xkey >> 32) if ef == 1 else NONE
    xid = ep
    n_dist, n_expand, n_adj = 1, 0, 0
    while True:
        n_expand += 1
        row = adj[xid]
        halves = [row[:32], row[32:64]] if wide else [row[:32]]
        ypos, ykey, yid = NONE, KEY_INF, xid
        finished = have_next = False
        for half, ids in enumerate(halves):
            nvalid = len(ids)
            n_adj += nvalid
            if half == 0:
                p = first_unexpanded()
                if p is not None:
                    ypos, ykey = p, at(p)
                    yid = (ykey & 0xFFFFFFFF) >> 1
            # lane 2R+1 holds the candidate of neighbor R
            ck = [0] * 64
            cand = 0
            for R in range(32):
                nb = ids[R] if R < nvalid else (ids[-1] if nvalid else xid)
                k = wkey(dbits[nb], nb)
                ck[2 * R] = ck[2 * R + 1] = k          # (even lanes hold garbage distances in the kernel; unused)
                if R < nvalid:
                    cand |= 1 << (2 * R + 1)
            n_dist += nvalid
            # filter_mask
            passm = 0
            tiem = 0
            for l in range(64):
                if cand >> l & 1 and (ck[l] >> 32) <= theta:
                    passm |= 1 << l
                    if (ck[l] >> 32) == theta:
                        tiem |= 1 << l
            if tiem:
                w = nth_expanded(ef)
                if w is not None:
                    worst = at(w) >> 32
                    for l in range(64):
                        if tiem >> l & 1 and not (ck[l] >> 32) < worst:
                            passm &= ~(1 << l)
            # rank_candidates: every candidate ranked as if the list held none of them, then looked up in the image
            while True:
                shift = [[0] * 64 for _ in range(S)]
                rankv = [0] * 64
                below = [0] * 64
                it = passm
                while it:
                    j = (it & -it).bit_length() - 1
                    K = ck[j]
                    it &= it - 1
                    if twin_rows:
                        twins = 0
                        for l in range(64):
                            if (ck[l] & 0xFFFFFFFF) == (K & 0xFFFFFFFF) and it >> l & 1:
                                twins |= 1 << l
                        it &= ~twins
                        passm &= ~twins
                    above = 0
                    for s_ in range(S):
                        for l in range(64):
                            if key[s_][l] > K:
                                shift[s_][l] += 1
                                above += 1
                    rankv[j] = above
                    for l in range(64):
                        if ck[l] > K:
                            below[l] += 1
                known = 0
                for l in range(64):
                    rankv[l] = CAP - rankv[l]
                    if not passm >> l & 1:
                        continue
                    r0 = min(rankv[l], CAP - 1)
                    r1 = rankv[l] - 1 if rankv[l] else 0
                    me = (ck[l] & 0xFFFFFFFF) | 1
                    if (((img[r0] & 0xFFFFFFFF) | 1) == me and rankv[l] < CAP) or ((img[r1] & 0xFFFFFFFF) | 1) == me:
                        known |= 1 << l
                if not known:
                    break
                passm &= ~known
            m = bin(passm).count("1")
            if m:
                wm = [l for l in range(64) if passm >> l & 1 and below[l] == 0]
                assert len(wm) == 1, wm
                jw = wm[0]
                if rankv[jw] <= ypos:
                    ykey, ypos, yid = ck[jw], rankv[jw], (ck[jw] & 0xFFFFFFFF) >> 1
            last = (not wide) or half == 1 or nvalid < 32 or len(halves[1]) == 0
            if last:
                if ypos == NONE:
                    finished = True
                elif ypos >= ef and count_closer(ykey >> 32) >= ef:
                    finished = True
                else:
                    have_next = True
            if finished:
                break
            if m:
                # merge_ranked: scatter, read back, tie test on place CAP
                written = set()
                for s in range(S):
                    for l in range(64):
                        pos = s * 64 + l + shift[s][l]
                        assert pos not in written, "two keys for one place"
                        written.add(pos)
                        img[pos] = key[s][l]
                for l in range(64):
                    if passm >> l & 1:
                        pos = rankv[l] + below[l]
                        assert pos not in written, "two keys for one place"
                        written.add(pos)
                        img[pos] = ck[l]
                assert written == set(range(CAP + m)), "a permutation of the entries and the candidates"
                for s in range(S):
                    for l in range(64):
                        key[s][l] = img[s * 64 + l]
                flat = [key[s][l] for s in range(S) for l in range(64)]
                assert flat == sorted(flat), "the list must stay sorted"
                lost = img[CAP] >> 32
                theta = img[ef - 1] >> 32
                if lost == theta and theta != NONE:
                    return None, None
            if last:
                break
        if not have_next:
            break
        assert at(ypos) == ykey, "the node chosen before the merge must stand where it was predicted"
        assert first_unexpanded() == ypos, "and be the first unexpanded entry"
        key[ypos >> 6][ypos & 63] |= 1
        xid = yid
    flagged = [(k >> 32, (k & 0xFFFFFFFF) >> 1) for s in range(S) for k in key[s] if (k & 1) and (k >> 32) != NONE]
    return flagged[:ef], (n_dist, n_expand, n_adj)


def main(seed=1, rounds=1500):
    rnd = random.Random(seed)
    trials = bails = 0
    for it in range(rounds):
        n = rnd.choice([5, 20, 80, 300, 1000])
        wide = it % 3 == 0
        deg = rnd.choice([2, 4, 8, 15, 30, 32] + ([40, 63, 64] if wide else []))
        S = rnd.choice([1, 1, 2, 4])
        ef = rnd.choice([1, 1, 2, 5, 10, 50, 60] + ([100, 124] if S >= 2 else []) + ([200, 252] if S >= 4 else []))
        if ef > 64 * S - 4:
            ef = 64 * S - 4
        adj = [rnd.sample(range(n), min(deg, n)) for _ in range(n)]
        if it % 5 == 0:  # rows that name a neighbor twice
            for row in adj:
                if len(row) >= 2 and rnd.random() < 0.3:
                    row[-1] = row[0]
        mode = rnd.choice(["float", "int_small", "int_tiny", "dup"])
        if mode == "float":
            dv = [rnd.random() for _ in range(n)]
        elif mode == "int_small":
            dv = [rnd.randrange(50) / 50.0 for _ in range(n)]
        elif mode == "int_tiny":
            dv = [rnd.randrange(4) / 4.0 for _ in range(n)]
        else:
            base = [rnd.random() for _ in range(max(1, n // 4))]
            dv = [base[rnd.randrange(len(base))] for _ in range(n)]
        db = [f32bits(x) for x in dv]
        ep = rnd.randrange(n)
        r0, c0 = reference(adj, db.__getitem__, ep, ef)
        r1, c1 = ranked(adj, db, ep, ef, S, wide)
        trials += 1
        if r1 is None:
            bails += 1
            continue
        assert r0 == r1, (it, mode, n, deg, ef, S, wide, r0[:5], r1[:5])
        # expansions and adjacency entries are the reference's; n_dist counts evaluations (revisits included)
        assert c0[1:] == c1[1:] and c0[0] <= c1[0] <= c0[2] + 1, (it, mode, n, deg, ef, S, c0, c1)
    print("ok: %d walks equal, %d bailed (ties at the boundary)" % (trials - bails, bails))
    return trials - bails, bails


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 1)

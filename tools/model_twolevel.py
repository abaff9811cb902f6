#!/usr/bin/env python
"""CPU model of the walker's TWO-LEVEL list for max_search beyond 1024 (walk_fast.h, search_layer_long; round 6) against
the reference's two heaps (search_for_neighbors, src/index/mod.rs:999-1037).

Round 4's long lists kept ONE sorted array of 64 S keys (registers + an image in LDS) and merged every expansion's
candidates into it: O(S) work per expansion with a large constant (13.7 k of 25.4 k clocks per expansion at max_search
4096: profiles/r6_phase_f32_ef4096_before.txt). The two-level list keeps
  * M: the main sorted array, `cap` keys, in LDS only;
  * F: the "fresh" sorted list of at most `fcap` (63) keys in one register pair per lane -- the candidates of the last
       few expansions enter HERE, with the one-slot ranked merge of the short lists;
  * the logical list is M u F (keys are unique across both: a candidate is looked up in both before it enters).
Operations on the union:
  * theta  = distance of the union's entry max_search-1 (None while it is shorter): one merge-path split of the two sorted
             arrays, all 64 lanes at once on the device;
  * next   = the smaller of the two first unexpanded entries; flagged where it stands;
  * break  <=> theta < d_next  (#{entries with dist < d_x} >= max_search  <=>  entry max_search-1 is strictly closer);
  * filter = the reference's (mod.rs:1029): `worst` = the max_search-th EXPANDED entry of the union, looked at only when a
             candidate ties with theta;
  * flush  : when F cannot take an expansion's candidates (n_F + candidates > fcap) F is merged into M -- the O(S) step, once
             per several expansions. Entries beyond M's capacity fall off: dead, unless the closest of them ties with theta
             after the expansion's inserts -- then the walk is handed to the exact walker (bail), as with the short lists.
This script replays the model against the heaps on random graphs, distances full of ties included. Not product code."""
import bisect
import random
import sys

from model_unified import reference


def twolevel(adj, dist, ep, ef, cap, fcap):
    M = [[dist(ep), ep, True]]  # ascending (dist, id); third field = expanded. The entry point is popped at once
    F = []
    n_dist, n_expand, n_adj = 1, 1, len(adj[ep])
    x = ep
    lost = None
    flushes = 0

    def union():
        return sorted(M + F, key=lambda e: (e[0], e[1]))

    def flush():
        nonlocal M, F, lost
        U = union()
        for y in U[cap:]:
            lost = y[0] if lost is None else min(lost, y[0])
        M, F = U[:cap], []

    while True:
        U = union()
        theta = U[ef - 1][0] if len(U) >= ef else None
        row = adj[x]
        cands = []
        for n in row:
            dn = dist(n)
            n_dist += 1
            if theta is not None and dn > theta:
                continue  # dead: max_search entries are strictly closer
            cands.append((dn, n))
        if theta is not None and any(c[0] == theta for c in cands):
            # a tie with theta: res.peek() decides (mod.rs:1029); the device flushes first, so that the union is M
            flush()
            flushes += 1
            exp = [e for e in M if e[2]]
            worst = exp[ef - 1][0] if len(exp) >= ef else None
            cands = [c for c in cands if not (c[0] == theta and worst is not None and not c[0] < worst)]
        if len(F) + len(cands) > fcap:  # F must take them all: flush BEFORE they are ranked
            flush()
            flushes += 1
        seen = set()
        for dn, n in cands:
            if n in seen or any(e[1] == n for e in M) or any(e[1] == n for e in F):
                continue  # the union holds it already (a revisit), or the row names it twice
            seen.add(n)
            keys = [(e[0], e[1]) for e in F]
            F.insert(bisect.bisect_left(keys, (dn, n)), [dn, n, False])
        assert len(F) <= fcap
        U = union()
        theta = U[ef - 1][0] if len(U) >= ef else None
        if lost is not None:
            if theta is not None and lost == theta:
                return None, None, flushes  # the closest lost entry ties with entry max_search-1: not provably dead
            lost = None
        yM = next((e for e in M if not e[2]), None)
        yF = next((e for e in F if not e[2]), None)
        if yM is None and yF is None:
            break
        y = min([e for e in (yM, yF) if e is not None], key=lambda e: (e[0], e[1]))
        if theta is not None and theta < y[0]:
            break
        y[2] = True
        x = y[1]
        n_expand += 1
        n_adj += len(adj[x])
    exp = [(e[0], e[1]) for e in union() if e[2]]
    return exp[:ef], (n_dist, n_expand, n_adj), flushes


def main():
    rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    trials = bails = flushed = 0
    for it in range(3000):
        n = rnd.choice([5, 20, 80, 300, 1000, 3000])
        deg = rnd.choice([2, 4, 8, 15, 30])
        ef = rnd.choice([1, 2, 5, 10, 50, 100, 128])
        cap = rnd.choice([ef + 8, ef + 64, 2 * ef + 16, 192])
        cap = max(cap, ef + 1)
        fcap = rnd.choice([max(deg, 31), 63, 40])
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
        dist = dv.__getitem__
        ep = rnd.randrange(n)
        r0, c0 = reference(adj, dist, ep, ef)
        r1, c1, fl = twolevel(adj, dist, ep, ef, cap, fcap)
        trials += 1
        flushed += 1 if fl else 0
        if r1 is None:
            bails += 1
            continue
        assert r0 == r1, (it, mode, n, deg, ef, cap, fcap, r0[:5], r1[:5])
        # expansions and adjacency entries are the reference's; n_dist counts evaluations (no visited set)
        assert c0[1:] == c1[1:] and c0[0] <= c1[0] <= c0[2] + 1, (it, mode, n, deg, ef, c0, c1)
    print("ok: %d walks equal (%d of them flushed F into M at least once), %d bailed (ties at the boundary)"
          % (trials - bails, flushed, bails))
    return trials - bails, bails, flushed


if __name__ == "__main__":
    main()

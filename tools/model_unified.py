#!/usr/bin/env python
"""CPU model of the walker's single sorted list (walk_fast.h) against the reference's two heaps.

search_for_neighbors (src/index/mod.rs:999-1037) keeps `res` (the max_search best popped nodes)
and `pq` (the unbounded candidate queue). The device walker keeps ONE ascending list of `cap`
keys (dist, id) with an `expanded` flag per entry:
  * next node to expand  = first unexpanded entry x;
  * break                <=> #{entries with dist < d_x} >= max_search (they all precede x, so they
                             are all expanded: exactly `res.len() == max_search && d_x > res.peek().dist`);
  * enqueue filter       = the reference's, evaluated on the list: when >= max_search expanded
                           entries are in the list, `worst` is the max_search-th of them;
  * dead candidates      (>= max_search entries strictly closer) are not inserted;
  * an entry that falls off the end is provably dead unless it ties with entry max_search-1:
    then the walk is abandoned (handed to the exact global-memory walker). The kernel takes that
    test once per expansion (insert_sorted: the smallest lost distance against entry max_search-1
    AFTER the expansion's last insert -- that entry only moves closer while candidates go in, so a
    lost entry that does not tie with it then has max_search entries strictly closer); the model
    runs both forms (`deferred`).
This script replays both on random graphs -- including integer distances full of ties -- and
asserts equal results and counters whenever the model does not bail.  Not product code.
"""
import bisect
import random
import sys


def reference(adj, dist, ep, ef):
    import heapq
    res, pq, visited = [], [(dist(ep), ep)], {ep}
    n_dist, n_expand, n_adj = 1, 0, 0
    while pq:
        d, x = heapq.heappop(pq)
        if len(res) >= ef and d > -res[0][0]:
            break
        if len(res) < ef:
            heapq.heappush(res, (-d, -x))
        elif (d, x) < (-res[0][0], -res[0][1]):
            heapq.heapreplace(res, (-d, -x))
        n_expand += 1
        n_adj += len(adj[x])
        for n in adj[x]:
            if n not in visited:
                visited.add(n)
                dn = dist(n)
                n_dist += 1
                if len(res) < ef or dn < -res[0][0]:
                    heapq.heappush(pq, (dn, n))
    return sorted((-a, -b) for a, b in res), (n_dist, n_expand, n_adj)


def unified_passes(adj, dist, ep, ef, cap, pass_size):
    """The walker WITHOUT a visited set on rows wider than its lanes (walk_fast.h, WIDE: layers of up to 64 ids): an
    expansion takes the row in passes of `pass_size` neighbors -- distances, filter, look-up, insert and the tie test per
    pass. `worst` (res.peek(), mod.rs:1029) is the expansion's: res does not change while a row is processed; theta (the
    dead-candidate bound, entry max_search-1) is re-read after every pass's inserts, as the kernel does."""
    L = [[dist(ep), ep, True]]
    n_dist, n_expand, n_adj = 1, 1, len(adj[ep])
    x = ep
    while True:
        exp = [e for e in L if e[2]]
        worst = exp[ef - 1][0] if len(exp) >= ef else None
        row = adj[x]
        for p0 in range(0, max(len(row), 1), pass_size):
            theta = L[ef - 1][0] if len(L) >= ef else None
            lost = None
            for n in row[p0:p0 + pass_size]:
                dn = dist(n)
                n_dist += 1
                if worst is not None and not dn < worst:
                    continue
                if theta is not None and dn > theta:
                    continue
                if any(e[1] == n for e in L):
                    continue
                keys = [(e[0], e[1]) for e in L]
                L.insert(bisect.bisect_left(keys, (dn, n)), [dn, n, False])
                while len(L) > cap:
                    y = L.pop()
                    lost = y[0] if lost is None else min(lost, y[0])
            if lost is not None and lost == L[ef - 1][0]:
                return None, None
        ypos = next((i for i, e in enumerate(L) if not e[2]), None)
        if ypos is None:
            break
        if ypos >= ef and sum(1 for e in L if e[0] < L[ypos][0]) >= ef:
            break
        L[ypos][2] = True
        x = L[ypos][1]
        n_expand += 1
        n_adj += len(adj[x])
    exp = [(e[0], e[1]) for e in L if e[2]]
    return exp[:ef], (n_dist, n_expand, n_adj)


def unified(adj, dist, ep, ef, cap, deferred=True, novis=False):
    """The list walker with walk_fast.h's control flow: the next node is decided (and its break test
    taken) BEFORE the candidates of the current expansion are merged.
    novis: the walker WITHOUT a visited set (VisitedNone, wave_prims.h) -- every neighbor is evaluated; a candidate that
    passed the filter is looked up in the list (all `cap` places) in the next-node decision and right before its
    insert, and dropped if it is there."""
    L = [[dist(ep), ep, True]]  # ascending by (dist, id); third field = expanded. The entry point is popped at once
    visited = {ep}
    n_dist, n_expand, n_adj = 1, 1, len(adj[ep])
    x = ep
    while True:
        # expansion of x (already flagged): the filter is frozen (res does not change during an expansion)
        exp = [e for e in L if e[2]]
        worst = exp[ef - 1][0] if len(exp) >= ef else None
        theta = L[ef - 1][0] if len(L) >= ef else None
        cands = []
        for n in adj[x]:
            if novis or n not in visited:
                visited.add(n)
                dn = dist(n)
                n_dist += 1
                if worst is not None and not dn < worst:
                    continue
                if theta is not None and dn > theta:
                    continue  # dead: max_search entries are strictly closer
                cands.append((dn, n))
        # who is next? (before the merge)
        ypos = next((i for i, e in enumerate(L) if not e[2]), None)
        ykey = (L[ypos][0], L[ypos][1]) if ypos is not None else (float("inf"), 1 << 62)
        beat = [c for c in cands if c < ykey]
        if novis:  # the smallest candidate below y that the list does not hold (the held ones leave the candidates)
            while beat and any(e[1] == min(beat)[1] for e in L):
                known = min(beat)
                cands = [c for c in cands if c != known]
                beat = [c for c in cands if c < ykey]
        if beat:
            nxt = min(beat)[1]
        else:
            if ypos is None:
                break
            if ypos >= ef and sum(1 for e in L if e[0] < ykey[0]) >= ef:
                break
            nxt = ykey[1]
        lost = None
        for dn, n in cands:
            if novis and any(e[1] == n for e in L):
                continue  # in the list already: a revisit, or the row names the node twice
            keys = [(e[0], e[1]) for e in L]
            L.insert(bisect.bisect_left(keys, (dn, n)), [dn, n, False])
            while len(L) > cap:
                y = L.pop()
                if not deferred and y[0] == L[ef - 1][0]:
                    return None, None  # bail: not provably dead
                lost = y[0] if lost is None else min(lost, y[0])
        if deferred and lost is not None and lost == L[ef - 1][0]:
            return None, None  # bail: the closest lost entry ties with entry max_search-1
        p = next(i for i, e in enumerate(L) if not e[2])
        assert L[p][1] == nxt, "the decision taken before the merge must name the first unexpanded entry after it"
        L[p][2] = True
        x = nxt
        n_expand += 1
        n_adj += len(adj[x])
    exp = [(e[0], e[1]) for e in L if e[2]]
    return exp[:ef], (n_dist, n_expand, n_adj)


def main():
    rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    trials = bails = 0
    for it in range(4000):
        n = rnd.choice([5, 20, 80, 300, 1000])
        deg = rnd.choice([2, 4, 8, 15, 30])
        ef = rnd.choice([1, 1, 2, 5, 10, 50, 60, 64])
        cap = 64
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
        deferred = bool(it & 1)
        novis = bool(it & 2)
        r1, c1 = unified(adj, dist, ep, ef, cap, deferred, novis)
        trials += 1
        if r1 is None:
            bails += 1
            continue
        assert r0 == r1, (it, mode, deferred, novis, n, deg, ef, r0[:5], r1[:5])
        if novis:  # expansions and adjacency entries are the reference's; n_dist counts evaluations
            assert c0[1:] == c1[1:] and c0[0] <= c1[0] <= c0[2] + 1, (it, mode, deferred, n, deg, ef, c0, c1)
        else:
            assert c0 == c1, (it, mode, deferred, n, deg, ef, c0, c1)
    print("ok: %d walks equal, %d bailed (ties at the boundary)" % (trials - bails, bails))
    return trials - bails, bails


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Design-space simulation (CPU, numpy): how long is the dependent chain of a bottom-layer walk and
what would speculative multi-head expansion cost?  Not a test and not product code: it replays
search_for_neighbors (src/index/mod.rs:999-1037) on an oracle-built graph and counts

  head_miss   pops whose node was NOT the queue head when the previous expansion's prefetch was
              issued (its adjacency row needs a dependent load unless it travels with the vector)
  rounds(P)   HBM round trips when the top-P uncached queue entries are fetched per round and pops
              are served from the cache (P = 1: the plain walk)
  waste(P)    element rows fetched speculatively whose distance was never used

usage: python tools/sim_chain.py [n=200000] [ef=50] [nq=200]
"""
import heapq
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc  # noqa: E402


def walk(adj, el, q, ep, ef, P):
    UN = 0xFFFFFFFF
    dist = lambda i: max(0.0, 1.0 - float(el[i] @ q))  # noqa: E731
    visited = {ep}
    pq = [(dist(ep), ep)]
    res = []  # max-heap via negatives
    cache = {}  # node -> {nbr: dist} of neighbors unvisited at fetch time
    rounds = expansions = head_miss = fetched = used = 0
    prev_head = None
    while pq:
        d, x = pq[0]
        if len(res) >= ef and d > -res[0][0]:
            break
        if x not in cache:
            # one round trip: x and the next P-1 best uncached queue entries
            rounds += 1
            todo = [x]
            if P > 1:
                for dd, y in heapq.nsmallest(P + len(cache) + 1, pq):
                    if len(todo) >= P:
                        break
                    if y != x and y not in cache:
                        todo.append(y)
            for y in todo:
                row = adj[y]
                c = {}
                for nb in row:
                    if nb == UN:
                        break
                    if nb not in visited:
                        c[int(nb)] = dist(nb)
                fetched += len(c)
                cache[y] = c
        heapq.heappop(pq)
        if prev_head is not None and prev_head != x:
            head_miss += 1
        if len(res) < ef:
            heapq.heappush(res, (-d, -x))
        elif (d, x) < (-res[0][0], -res[0][1]):
            heapq.heapreplace(res, (-d, -x))
        expansions += 1
        prev_head = pq[0][1] if pq else None  # the head the prefetch of this expansion would target
        c = cache.pop(x)
        full = len(res) >= ef
        worst = -res[0][0]
        for nb in adj[x]:
            if nb == UN:
                break
            nb = int(nb)
            if nb in visited:
                continue
            visited.add(nb)
            used += 1
            dn = c[nb]
            if not full or dn < worst:
                heapq.heappush(pq, (dn, nb))
    for c in cache.values():
        pass
    return rounds, expansions, head_miss, fetched, used


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    ef = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    nq = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    dim = 100
    orc.build()
    el = orc.normalize_f32(orc.synth_rows(0x6772616E6E65, 0, n, dim))
    qs = orc.normalize_f32(orc.synth_rows(0x6772616E6E66, 0, nq, dim))
    ix = orc.build_index(el, n_threads=0, batch_max=65536)
    adj = ix.layers[-1]
    # entry points: greedy descent through the upper layers with the oracle
    for P in (1, 2, 3, 4):
        tot = np.zeros(5)
        for q in qs:
            ep = 0
            for l in range(len(ix.layers) - 1):
                ep = ix.search_for_neighbors(l, ep, q, 1)[0][0]
            tot += walk(adj, el, q, ep, ef, P)
        r, e, hm, f, u = tot / nq
        print("P=%d rounds %.1f expansions %.1f head_miss %.1f (%.0f%%) rows fetched %.0f used %.0f waste %.1f%%"
              % (P, r, e, hm, 100 * hm / e, f, u, 100 * (f - u) / max(u, 1)))


if __name__ == "__main__":
    main()

"""CPU models of device-side arguments, replayed against the reference's plain algorithms.

* tools/model_unified.py: the walker's ONE sorted list with expanded flags and its once-per-expansion tie test
  (walk_fast.h, insert_sorted) against the reference's two heaps (src/index/mod.rs:999-1037) on tie-heavy graphs.
  The same list WITHOUT a visited set (wave_prims.h VisitedNone: candidates are looked up in the list) against the same.
* tools/model_twolevel.py: the two-level list of max_search beyond 1024 (walk_fast.h, search_layer_long) against the two heaps,
  and its device steps (4-ary lower bound, flush in place, theta by a split of two sorted arrays, the insert into F) lane for lane.
* the builder's one-candidate form of add_and_limit_neighbors (builder_kernels.h, add_one_to_selected) against the
  full sort + select_neighbors pass (src/index/mod.rs:849-883, 923-959) over rows that fill, get limited, refill.
"""
import importlib.util
import os
import random

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sorted_list_with_deferred_tie_test_equals_two_heaps():
    spec = importlib.util.spec_from_file_location("model_unified", os.path.join(ROOT, "tools", "model_unified.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    rnd = random.Random(11)
    equal = bailed = 0
    for it in range(700):
        n = rnd.choice([5, 20, 80, 300])
        deg = rnd.choice([2, 4, 8, 15, 30])
        ef = rnd.choice([1, 2, 5, 10, 50, 60, 64])
        adj = [rnd.sample(range(n), min(deg, n)) for _ in range(n)]
        mode = rnd.choice(["float", "int_small", "int_tiny"])
        dv = [rnd.random() if mode == "float" else rnd.randrange(50 if mode == "int_small" else 4) / 50.0 for _ in range(n)]
        ep = rnd.randrange(n)
        r0, c0 = m.reference(adj, dv.__getitem__, ep, ef)
        r1, c1 = m.unified(adj, dv.__getitem__, ep, ef, 64, deferred=bool(it & 1))
        if r1 is None:
            bailed += 1
            continue
        assert r0 == r1 and c0 == c1, (it, mode, n, deg, ef)
        equal += 1
    assert equal > 500 and bailed > 0  # both outcomes are exercised


def test_walk_without_a_visited_set_equals_two_heaps():
    """The register walkers keep no visited set (granne_amd/csrc/wave_prims.h, VisitedNone): every neighbor is evaluated, a
    candidate that passed the filter is looked up in the list in the next-node decision and before its insert. Same
    results as the reference's HashSet walk on tie-heavy graphs, rows that name a neighbor twice and self-references
    included; expansions and adjacency entries equal, evaluations at least the reference's distinct nodes."""
    spec = importlib.util.spec_from_file_location("model_unified", os.path.join(ROOT, "tools", "model_unified.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    rnd = random.Random(23)
    equal = bailed = revisits = 0
    for it in range(900):
        n = rnd.choice([5, 20, 80, 300])
        deg = rnd.choice([2, 4, 8, 15, 30])
        ef = rnd.choice([1, 2, 5, 10, 50, 60, 64])
        adj = [rnd.sample(range(n), min(deg, n)) for _ in range(n)]
        if it % 3 == 0:
            for row in adj:
                if len(row) >= 2 and rnd.random() < 0.3:
                    row[-1] = row[0]
        mode = rnd.choice(["float", "int_small", "int_tiny"])
        dv = [rnd.random() if mode == "float" else rnd.randrange(50 if mode == "int_small" else 4) / 50.0 for _ in range(n)]
        ep = rnd.randrange(n)
        r0, c0 = m.reference(adj, dv.__getitem__, ep, ef)
        r1, c1 = m.unified(adj, dv.__getitem__, ep, ef, 64, deferred=True, novis=True)
        if r1 is None:
            bailed += 1
            continue
        assert r0 == r1, (it, mode, n, deg, ef)
        assert c0[1:] == c1[1:] and c0[0] <= c1[0] <= c0[2] + 1, (it, c0, c1)
        revisits += c1[0] - c0[0]
        equal += 1
    assert equal > 600 and bailed > 0 and revisits > 1000


def _dist(a, b):
    r = np.float32(1.0) - np.float32(np.dot(a, b))
    return float(r) if r > 0 else 0.0


def _select(el, cands, m):
    """select_neighbors, src/index/mod.rs:849-883: cands = [(id, d)] sorted ascending."""
    if len(cands) <= m:
        return list(cands)
    out = []
    for j, d in cands:
        if len(out) >= m:
            break
        if all(d <= _dist(el[n], el[j]) for n, _ in out):
            out.append((j, d))
    return out


def _full(el, node, row, x, cap):
    """add_and_limit_neighbors with one extra candidate, src/index/mod.rs:923-959 (ties by id)."""
    cands = [(j, _dist(el[node], el[j])) for j in row] + [(x, _dist(el[node], el[x]))]
    cands.sort(key=lambda c: (c[1], c[0]))
    return [j for j, _ in _select(el, cands, cap)]


def _one_candidate(el, node, row, x, cap):
    """builder_kernels.h, add_one_to_selected: `row` is full and the untouched output of select_neighbors."""
    c = len(row)
    d = [_dist(el[node], el[j]) for j in row]
    e = [_dist(el[j], el[x]) for j in row]
    keys = [(d[i], row[i]) for i in range(c)]
    assert keys == sorted(keys)
    dx = _dist(el[node], el[x])
    p = sum(1 for k in keys if k < (dx, x))
    if p >= cap or any(not (dx <= e[i]) for i in range(p)):
        return list(row)
    keep = [row[i] for i in range(c) if i < p or d[i] <= e[i]]
    return (keep[:p] + [x] + keep[p:])[:cap]


def _replay(el, nodes, cap, rng, pool=150, cands=None):
    """connect_nodes (src/index/mod.rs:898-921) for a stream of near neighbors per node; every time the row is full and
    is the untouched output of the pass before, the one-candidate form must give what the full pass gives."""
    took_short = changed = 0
    for node in nodes:
        # link candidates are near neighbors, as in a build
        near = np.asarray(cands[node]) if cands is not None else np.argsort(-(el @ el[node]))[:pool]
        row, selected = [], False
        for x in rng.permutation(near).tolist():
            if x == node or x in row:
                continue
            if len(row) < cap:  # a free place, :912-914
                row.append(x)
                selected = False
                continue
            want = _full(el, node, row, x, cap)
            if selected:
                got = _one_candidate(el, node, row, x, cap)
                assert got == want, (node, x, cap)
                took_short += 1
                changed += got != row
            row, selected = want, True
    return took_short, changed


def test_one_candidate_add_and_limit_equals_the_full_pass():
    rng = np.random.default_rng(5)
    short = changed = 0
    # random points: select_neighbors prunes hard, small rows stay full now and then
    for dim, cap in [(100, 4), (48, 6), (16, 5), (6, 3)]:
        raw = rng.random((500, dim), dtype=np.float32) - np.float32(0.5)
        el = (raw / np.linalg.norm(raw, axis=1, keepdims=True)).astype(np.float32)
        a, b = _replay(el, range(30), cap, rng)
        short, changed = short + a, changed + b
    # shells: the neighbors of a center sit at center + eps * (its own direction), so each is closer to the center than
    # to the others (rows of 30 stay full and the newcomer mostly just takes its sorted place) -- except for pairs
    # that share a direction, where the closer one displaces the other
    for dim in (64, 100):
        rows, cands = [], {}
        for c in range(12):
            center = rng.standard_normal(dim).astype(np.float32)
            center /= np.linalg.norm(center)
            g = rng.standard_normal((100, dim)).astype(np.float32)
            g[60:] = g[rng.integers(0, 60, 40)] + np.float32(0.15) * rng.standard_normal((40, dim)).astype(np.float32)
            g /= np.linalg.norm(g, axis=1, keepdims=True)
            pts = center + rng.uniform(0.2, 0.7, (100, 1)).astype(np.float32) * g
            cands[len(rows)] = list(range(len(rows) + 1, len(rows) + 101))
            rows.append(center)
            rows.extend(pts)
        raw = np.asarray(rows, np.float32)
        el = (raw / np.linalg.norm(raw, axis=1, keepdims=True)).astype(np.float32)
        a, b = _replay(el, list(cands), 30, rng, cands=cands)
        assert a > 200, a  # full rows of 30 did take the short form
        short, changed = short + a, changed + b
    assert short > 1000 and changed > 50, (short, changed)


def test_two_level_list_walk_equals_two_heaps():
    """walk_fast.h, max_search beyond 1024 (search_layer_long): the two-level list -- M sorted in LDS, F (up to 63 keys) in
    registers, theta = the union's entry max_search-1, break <=> theta < d_next, flush when F cannot take an expansion's
    candidates, the tie path of the filter after a flush, what falls off M's end dead unless it ties with theta
    (tools/model_twolevel.py) -- against the reference's two heaps on tie-heavy graphs, twin rows included."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    spec = importlib.util.spec_from_file_location("model_twolevel", os.path.join(ROOT, "tools", "model_twolevel.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    rnd = random.Random(66)
    equal = bailed = flushed = 0
    for it in range(700):
        n = rnd.choice([5, 20, 80, 300, 1000])
        deg = rnd.choice([2, 4, 8, 15, 30])
        ef = rnd.choice([1, 2, 5, 10, 50, 100])
        cap = max(rnd.choice([ef + 8, ef + 64, 2 * ef + 16]), ef + 1)
        fcap = rnd.choice([max(deg, 31), 63])
        adj = [rnd.sample(range(n), min(deg, n)) for _ in range(n)]
        if it % 4 == 0:
            for row in adj:
                if len(row) >= 2 and rnd.random() < 0.3:
                    row[-1] = row[0]
        mode = rnd.choice(["float", "int_small", "int_tiny"])
        dv = [rnd.random() if mode == "float" else rnd.randrange(50 if mode == "int_small" else 4) / 50.0 for _ in range(n)]
        ep = rnd.randrange(n)
        r0, c0 = m.reference(adj, dv.__getitem__, ep, ef)
        r1, c1, fl = m.twolevel(adj, dv.__getitem__, ep, ef, cap, fcap)
        flushed += 1 if fl else 0
        if r1 is None:
            bailed += 1
            continue
        assert r0 == r1 and c0[1:] == c1[1:] and c0[0] <= c1[0] <= c0[2] + 1, (it, mode, n, deg, ef, cap, fcap)
        equal += 1
    assert equal > 500 and bailed > 0 and flushed > 200


def test_two_level_list_lane_for_lane():
    """The device's steps on the two-level list, lane for lane (64 lanes, windows of 64 keys), against plain merges:
    * m_lower_bound: the 4-ary lower bound with its fixed number of steps (6 for 2112 keys, 7 for 4160 / 8256);
    * flush: every F entry's rank in M; M's windows moved up IN PLACE from the top down -- a window's shift is a_lo (the F
      entries ranked below the window) plus, per lane, the F entries ranked inside it at or below the lane's entry -- F's
      entries written last at rank + j; what lands beyond CAP is lost; the lower bound of the first unexpanded entry;
    * update_theta: lane j tries "j keys of F and ef - j of M": exactly one lane is right;
    * insert_fresh: ranks / shifts / `below` of the candidates in F, the look-up at rank and rank - 1, one scatter."""
    import bisect
    rnd = random.Random(67)
    INF = (1 << 64) - 1

    def key(d, i, f):
        return (d << 32) | (i << 1) | f

    def lower_bound_4ary(M, CAP, k, steps):
        lo, hi = 0, CAP
        for _ in range(steps):
            ln, last = hi - lo, (hi - 1 if hi else 0)
            p1, p2, p3 = min(last, lo + (ln >> 2)), min(last, lo + (ln >> 1)), min(last, lo + ((3 * ln) >> 2))
            go, l1, l2, l3 = lo < hi, M[p1] < k, M[p2] < k, M[p3] < k
            nlo = p3 + 1 if l3 else p2 + 1 if l2 else p1 + 1 if l1 else lo
            nhi = hi if l3 else p3 if l2 else p2 if l1 else p1
            if go:
                lo, hi = nlo, nhi
        assert lo == hi
        return lo

    for it in range(300):
        S, steps = rnd.choice([(33, 6), (65, 7), (129, 7), (3, 6)])
        CAP = 64 * S
        nM = rnd.choice([0, 1, 5, 64, 65, CAP // 2, CAP - 70, CAP - 1, CAP])
        nF = rnd.randrange(0, 64)
        few = it % 2  # few distinct distances: ties on distance, the ids decide
        ids = rnd.sample(range(1 << 22), nM + nF + 40)
        Mreal = sorted(key(rnd.randrange(6 if few else 1 << 20), ids[j], rnd.randrange(2)) for j in range(nM))
        Freal = sorted(key(rnd.randrange(6 if few else 1 << 20), ids[nM + j], rnd.randrange(2)) for j in range(nF))
        M = Mreal + [INF] * (CAP - nM)
        F = Freal + [INF] * (64 - nF)
        for k in Freal[:5] + [0, INF - 1]:
            assert lower_bound_4ary(M, CAP, k, steps) == bisect.bisect_left(M, k)
        # ---- update_theta
        ef = rnd.randrange(1, max(2, min(CAP - 64, nM + nF + 5)))
        union = sorted(Mreal + Freal)
        if nM + nF >= ef:
            ok = []
            for j in range(64):
                in_range = j <= nF and j <= ef and ef - j <= nM
                i = ef - j if in_range else 0
                Mi, Mi1 = M[min(i, CAP - 1)], M[i - 1 if i else 0]
                Fj, Fj1 = F[j], (F[j - 1] if j else 0)
                good = in_range and (j == 0 or Fj1 < Mi) and (i == 0 or Mi1 < Fj)
                last = Mi1 if j == 0 else Fj1 if i == 0 else max(Fj1, Mi1)
                if good:
                    ok.append(last)
            assert ok == [union[ef - 1]], (it, nM, nF, ef, len(ok))
        # ---- insert_fresh on F: up to 32 candidates, some already in F (revisits)
        room = 63 - nF
        cands = []
        for j in range(rnd.randrange(0, min(32, room) + 1)):
            if rnd.random() < 0.2 and nF:
                cands.append(Freal[rnd.randrange(nF)] & ~1)
            else:
                cands.append(key(rnd.randrange(6 if few else 1 << 20), ids[nM + nF + j], 0))
        cands = list(dict.fromkeys(cands))
        passm = list(range(len(cands)))
        while True:
            shift = [sum(1 for c in passm if F[l] > cands[c]) for l in range(64)]
            rankv = {c: 64 - sum(1 for l in range(64) if F[l] > cands[c]) for c in passm}
            below = {c: sum(1 for o in passm if cands[c] > cands[o]) for c in passm}
            known = [c for c in passm if (rankv[c] < 64 and (F[min(rankv[c], 63)] | 1) == (cands[c] | 1)) or
                     (F[rankv[c] - 1 if rankv[c] else 0] | 1) == (cands[c] | 1)]
            if not known:
                break
            passm = [c for c in passm if c not in known]
        img = [None] * 128
        for l in range(64):
            img[l + shift[l]] = F[l]
        for c in passm:
            assert img[rankv[c] + below[c]] is None
            img[rankv[c] + below[c]] = cands[c]
        fresh = sorted({K for K in cands if not any((e | 1) == (K | 1) for e in Freal)})
        assert img[:64] == (sorted(Freal + fresh) + [INF] * 64)[:64], (it, nF, len(cands))
        # ---- flush
        live = list(range(nF))
        r = [lower_bound_4ary(M, CAP, F[j], steps) for j in live]
        out = list(M)
        lost = []
        m_un = next((p for p in range(nM) if not (M[p] & 1)), CAP)
        if nM and nF:
            w_first, w_top = r[0] >> 6, (nM - 1) >> 6
            a_hi = nF
            for w in range(w_top, w_first - 1, -1):
                a_lo = sum(1 for j in live if r[j] < w * 64)
                window = [out[w * 64 + l] for l in range(64)]  # read before anything of this window is written
                for l in range(64):
                    e = w * 64 + l
                    c = a_lo + sum(1 for j in range(a_lo, a_hi) if e >= r[j])
                    if e < nM:
                        if e + c < CAP:
                            out[e + c] = window[l]
                        else:
                            lost.append(window[l])
                a_hi = a_lo
        dest = [r[j] + j for j in live]
        for j in live:
            if dest[j] < CAP:
                out[dest[j]] = F[j]
            else:
                lost.append(F[j])
        merged = sorted(Mreal + Freal)
        assert out[:min(CAP, nM + nF)] == merged[:CAP], (it, S, nM, nF)
        assert all(k == INF for k in out[min(CAP, nM + nF):])
        assert sorted(lost) == merged[CAP:]
        fun = [j for j in live if not (F[j] & 1)]
        lb = min([m_un if m_un < CAP else nM] + ([dest[fun[0]]] if fun else []))
        first = next((p for p in range(min(CAP, nM + nF)) if not (out[p] & 1)), CAP)
        assert lb <= first and all(k & 1 for k in out[:min(lb, CAP)])


def test_rows_taken_in_two_passes_equal_two_heaps():
    """walk_fast.h's WIDE walker (layers of up to 64 ids): a row is evaluated, filtered and inserted in two passes of 32
    neighbors. The reference's filter bound (res.peek()) belongs to the expansion, the list's own bound (entry
    max_search-1) is re-read between the passes -- results, expansions and adjacency entries equal the reference's HashSet
    walk on tie-heavy graphs with rows of up to 64 ids, duplicates within and across the halves included."""
    spec = importlib.util.spec_from_file_location("model_unified", os.path.join(ROOT, "tools", "model_unified.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    rnd = random.Random(64)
    equal = bailed = 0
    for it in range(500):
        n = rnd.choice([70, 120, 400])
        deg = rnd.choice([33, 40, 48, 63, 64])
        ef = rnd.choice([1, 2, 10, 50, 60, 64])
        adj = [rnd.sample(range(n), min(deg, n)) for _ in range(n)]
        if it % 3 == 0:
            for row in adj:
                if rnd.random() < 0.4:
                    row[-1] = row[rnd.randrange(len(row) - 1)]  # the same node in both halves (or twice in one)
        mode = rnd.choice(["float", "int_small", "int_tiny"])
        dv = [rnd.random() if mode == "float" else rnd.randrange(50 if mode == "int_small" else 4) / 50.0 for _ in range(n)]
        ep = rnd.randrange(n)
        r0, c0 = m.reference(adj, dv.__getitem__, ep, ef)
        r1, c1 = m.unified_passes(adj, dv.__getitem__, ep, ef, 64, 32)
        if r1 is None:
            bailed += 1
            continue
        assert r0 == r1, (it, mode, n, deg, ef)
        assert c0[1:] == c1[1:] and c1[0] >= c0[0], (it, c0, c1)
        equal += 1
    assert equal > 300


def test_ranked_merge_of_the_short_lists_equals_two_heaps():
    """walk_fast.h's lists of up to 17 slots (round 5): every candidate of an expansion ranked at once, one scatter through
    the list's LDS image, the next node chosen (and flagged) before the merge, revisits looked up in the image after the
    ranks. tools/model_ranked.py replays it lane for lane -- 64 lanes, S slots, candidates in the odd lanes, rows of up to
    64 ids in two passes, rows that name a neighbor twice, ties -- against the reference's two heaps."""
    spec = importlib.util.spec_from_file_location("model_ranked", os.path.join(ROOT, "tools", "model_ranked.py"))
    m = importlib.util.module_from_spec(spec)
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    spec.loader.exec_module(m)
    equal, bailed = m.main(seed=31, rounds=400)
    assert equal > 300 and bailed > 0


def test_threshold_shared_through_a_score_histogram_never_passes_the_kth_best():
    """brute_force.h BfShare: the 128 lists of a query (64 ranges x 2 lane halves) count their inserts per score bucket;
    whoever looks the histogram up takes the highest bucket edge that kk counted elements reach. Replayed in float32 with
    the device's bucket arithmetic, lists scanning in a random interleaving and polling on the doubling schedule: at every
    moment the threshold is at most the kk-th best of the whole set, so the union of the lists ends with the kk best -- in
    whatever order the lists ran. (The device takes the reciprocal approximately: the downward fix-up of the bucket is
    what the argument rests on, so the model perturbs the estimate by +-1.)"""
    f32 = np.float32
    BUCKETS = 32

    def edge(step, j):
        return f32(step * f32(64 + j))

    def next_below(t):
        return np.nextafter(f32(t), f32(-np.inf), dtype=f32)

    rnd = random.Random(5)
    for trial in range(60):
        kk = rnd.choice([1, 7, 16])
        n_lists = rnd.choice([2, 8, 128])
        per_list = rnd.choice([40, 300, 2000])
        tie_heavy = trial % 3 == 0
        rng = np.random.default_rng(trial)
        scores = rng.normal(0.0, 0.1, (n_lists, per_list)).astype(f32)
        if tie_heavy:
            scores = (np.round(scores * 40) / 40).astype(f32)
        flat = np.sort(scores.ravel())[::-1]
        kth = flat[kk - 1]
        # the primed threshold: any score that kk elements reach (here: a low quantile of the top), one ulp below
        base = next_below(flat[min(len(flat) - 1, kk * rnd.choice([1, 4, 40]))])
        if not base > 0:
            continue
        step = f32(base * f32(1.0 / 64.0))
        hist = np.zeros(BUCKETS, np.int64)
        tau = [f32(base)] * n_lists
        kept = [[] for _ in range(n_lists)]        # every list keeps its own 16 best, like the register lists
        pos = [0] * n_lists
        tiles = [0] * n_lists
        next_poll = [2] * n_lists
        TILE = 16
        live = list(range(n_lists))
        while live:
            s = rnd.choice(live)
            tiles[s] += 1
            if tiles[s] == next_poll[s]:           # BfShare::poll
                next_poll[s] = tiles[s] * 2
                total, top = 0, -1
                for b in range(BUCKETS - 1, -1, -1):
                    total += int(hist[b])
                    if total >= kk:
                        top = b
                        break
                if top > 0:
                    tau[s] = max(tau[s], next_below(edge(step, top)))
            assert tau[s] <= kth, (trial, tau[s], kth)
            for sc in scores[s, pos[s]:pos[s] + TILE]:
                if sc > tau[s]:
                    kept[s].append(sc)
                    kept[s] = sorted(kept[s], reverse=True)[:16]
                    if len(kept[s]) == 16:
                        tau[s] = max(tau[s], kept[s][-1])
                    j = int(f32(sc) * f32(1.0 / step)) - 64 + rnd.choice([-1, 0, 0, 1])  # BfShare::count
                    j = max(0, min(BUCKETS - 1, j))
                    while j > 0 and sc < edge(step, j):
                        j -= 1
                    assert sc >= edge(step, j) or j == 0
                    hist[j] += 1
            pos[s] += TILE
            if pos[s] >= per_list:
                live.remove(s)
        union = np.sort(np.array([x for l in kept for x in l], f32))[::-1]
        assert len(union) >= kk and (union[:kk] == flat[:kk]).all(), trial

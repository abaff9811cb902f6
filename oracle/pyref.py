"""Second, independent restatement of granne's search path in plain Python (TEST INFRASTRUCTURE).

Written separately from oracle/granne_oracle.c on purpose: the C oracle is diffed against this
file in tests/ (small sizes only -- pure-Python loops). It uses Python's own containers
(heapq, set) where the C oracle hand-rolls heaps and a hash table, so a logic slip in either
shows up as a mismatch. Scalars are numpy float32; the fused multiply-add is libm's fmaf.

Citations are relative to /root/reference (granne v0.5.2).
"""
import ctypes
import ctypes.util
import heapq
import math

import numpy as np

_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.fmaf.restype = ctypes.c_float
_libm.fmaf.argtypes = [ctypes.c_float, ctypes.c_float, ctypes.c_float]

f32 = np.float32
UNUSED = 0xFFFFFFFF


def fmaf(a, b, c):
    return f32(_libm.fmaf(float(a), float(b), float(c)))


def dot_product_f32(x, y):
    """src/math.rs:16-42."""
    chunk = [f32(0.0)] * 32
    n = len(x)
    full = (n // 32) * 32
    for base in range(0, full, 32):
        for i in range(32):
            chunk[i] = fmaf(x[base + i], y[base + i], chunk[i])
    r = f32(0.0)
    for i in range(32):
        r = f32(r + chunk[i])
    for i in range(full, n):
        r = fmaf(x[i], y[i], r)
    return r


def normalize_f32(x):
    """src/math.rs:131-141."""
    x = np.array(x, dtype=f32)
    norm = f32(np.sqrt(dot_product_f32(x, x)))
    if norm > 0:
        for i in range(len(x)):
            x[i] = f32(x[i] / norm)
    return x


def dist_f32(x, y):
    """src/elements/angular.rs:63-74."""
    d = f32(f32(1.0) - dot_product_f32(x, y))
    return d if f32(0.0) <= d else f32(0.0)


def quantize(s):
    """src/elements/angular_int.rs:27-45."""
    s = np.asarray(s, dtype=f32)
    mx = f32(127.0) if len(s) == 0 else f32(np.max(np.abs(s)))
    out = np.zeros(len(s), np.int8)
    for i, x in enumerate(s):
        with np.errstate(all="ignore"):
            vi = f32(f32(x * f32(127.0)) / mx)
        if np.isnan(vi):
            out[i] = 0
        else:
            out[i] = int(max(-128.0, min(127.0, math.trunc(float(vi)))))
    return out


def dist_i8(x, y):
    """src/elements/angular_int.rs:47-60 + src/math.rs:59-89."""
    xi = np.asarray(x, np.int64)
    yi = np.asarray(y, np.int64)
    r = f32(int(np.sum(xi * yi)))
    dx = f32(int(np.sum(xi * xi)))
    dy = f32(int(np.sum(yi * yi)))
    with np.errstate(all="ignore"):
        q = f32(r / f32(np.sqrt(dx) * np.sqrt(dy)))
    if np.isnan(q):
        q = f32(0.0)
    d = f32(f32(1.0) - q)
    return d if f32(0.0) <= d else f32(0.0)


def dist(x, y):
    return dist_f32(x, y) if np.asarray(x).dtype == np.float32 else dist_i8(x, y)


def compute_num_elements_in_layer(total, multiplier, layer_idx):
    """src/index/mod.rs:634-643 (multiplier is an f32 widened to f64)."""
    m = float(f32(multiplier))
    t = float(total)
    v = math.ceil(t / math.pow(m, math.floor(math.log(t) / math.log(m)) - layer_idx))
    return min(int(v), total)


def get_neighbors(layer, idx):
    """src/index/mod.rs:540-552."""
    out = []
    for x in layer[idx]:
        if int(x) == UNUSED:
            break
        out.append(int(x))
    return out


def search_for_neighbors(layer, entrypoint, elements, goal, max_search, counters=None):
    """src/index/mod.rs:999-1037 with src/max_size_heap.rs:5-45."""
    res = []  # max-heap via negated keys: entries (-d, -id)
    pq = []   # min-heap of (d, id)
    visited = set()

    def res_full():
        return len(res) >= max_search

    def res_peek():  # the largest (d, id)
        return (-res[0][0], -res[0][1])

    d0 = float(dist(elements[entrypoint], goal))
    if counters is not None:
        counters["n_dist"] += 1
    heapq.heappush(pq, (d0, entrypoint))
    visited.add(entrypoint)

    while pq:
        d, idx = heapq.heappop(pq)
        if res_full() and d > res_peek()[0]:
            break
        # MaxSizeHeap::push
        if not res_full():
            heapq.heappush(res, (-d, -idx))
        elif (d, idx) < res_peek():
            heapq.heapreplace(res, (-d, -idx))
        nbrs = get_neighbors(layer, idx)
        if counters is not None:
            counters["n_expand"] += 1
            counters["n_adj"] += len(nbrs)
        for n in nbrs:
            if n not in visited:
                visited.add(n)
                dn = float(dist(elements[n], goal))
                if counters is not None:
                    counters["n_dist"] += 1
                if (not res_full()) or dn < res_peek()[0]:
                    heapq.heappush(pq, (dn, n))
    out = sorted((-a, -b) for a, b in res)
    return [(i, d) for d, i in out]


def search(layers, elements, query, max_search, num_neighbors, counters=None):
    """src/index/mod.rs:140-150, 963-997."""
    if max_search == 0:
        raise RuntimeError("panic: res.peek().unwrap() on empty heap (src/index/mod.rs:1019)")
    if not layers:
        return []
    entrypoint = 0
    for layer in layers[:-1]:
        entrypoint = search_for_neighbors(layer, entrypoint, elements, query, 1, counters)[0][0]
    return search_for_neighbors(layers[-1], entrypoint, elements, query, max_search, counters)[:num_neighbors]


# ---- Granne::reorder (src/index/reorder.rs) ---------------------------------------------------------
NUM_LAYERS = 8  # reorder.rs:177


def find_entrypoint_trail(layers, elements, max_layer, element):
    """reorder.rs:180-208. eps[i] is read before it is assigned, so every walk starts at node 0."""
    eps = [0] * NUM_LAYERS
    for i, layer in list(enumerate(layers))[: min(NUM_LAYERS, max_layer)]:
        ep = 0 if i == 0 else eps[i]
        eps[i] = search_for_neighbors(layer, ep, elements, element, 1)[0][0]
    return eps


def compute_order(layers, elements):
    """reorder.rs:135-175."""
    order = list(range(len(layers[0])))
    order_inv = [0] * len(layers[len(layers) - 2])
    for layer in range(1, len(layers)):
        keyed = []
        for idx in range(len(layers[layer - 1]), len(layers[layer])):
            eps = find_entrypoint_trail(layers, elements, layer, elements[idx])
            keyed.append(([order_inv[i] for i in eps], idx))
        keyed.sort()
        order.extend(idx for _, idx in keyed)
        if layer < len(layers) - 1:
            for i in range(len(layers[layer - 1]), len(layers[layer])):
                order_inv[order[i]] = i
    return order


def reorder_layers(layers, order):
    """reorder.rs:210-292: rows follow `order`, ids go through the reverse mapping; MultiSetVector::push
    sorts each set (src/slice_vector/set_vector.rs:41-47)."""
    rev = [0] * len(order)
    for i, j in enumerate(order):
        rev[j] = i
    return [[sorted(rev[n] for n in get_neighbors(layer, order[i])) for i in range(len(layer))] for layer in layers]


# ---- GranneBuilder, `singlethreaded` order (src/index/mod.rs:364-402, 645-960) ----------------------
EPS100 = float(np.float32(100.0) * np.float32(1.1920929e-07))  # 100.0 * f32::EPSILON


class Builder:
    """GranneBuilder::new + build_partial, restated from the Rust source independently of
    oracle/granne_oracle.c. Layers are lists of rows (lists of ids, UNUSED padded to num_neighbors).
    Where the reference's sort is unstable (add_and_limit_neighbors sorts by distance only,
    mod.rs:943) ties are ordered by id, the convention the C oracle and the GPU builder share."""

    def __init__(self, elements, num_neighbors=30, max_search=200, layer_multiplier=15.0, reinsert_elements=True,
                 expected_num_elements=None, batch_max=0, batch_div=8):
        # batch_max > 0: the BATCHED schedule of the GPU builder (oracle/granne_oracle.h, gro_build_config):
        # a batch's members search and select against the graph frozen at the start of the batch, then
        # their link updates are applied in batch order
        self.batch_max, self.batch_div = batch_max, batch_div
        self.elements = elements
        self.num_neighbors = num_neighbors
        self.max_search = max_search
        self.layer_multiplier = layer_multiplier
        self.reinsert_elements = reinsert_elements
        self.expected_num_elements = expected_num_elements
        self.layers = []

    def __len__(self):
        return len(self.layers[-1]) if self.layers else 0

    # mod.rs:374-402
    def build_partial(self, num_elements):
        if num_elements == 0:
            return
        assert num_elements >= len(self) and num_elements <= len(self.elements)
        if self.layers:
            self._index_elements_in_last_layer(num_elements)
        while len(self) < num_elements:
            self.layers.append([list(r) for r in self.layers[-1]] if self.layers else [])
            self._index_elements_in_last_layer(num_elements)

    # mod.rs:646-713
    def _index_elements_in_last_layer(self, max_num_elements):
        total = self.expected_num_elements if self.expected_num_elements is not None else len(self.elements)
        ideal = compute_num_elements_in_layer(max(total, len(self.elements)), self.layer_multiplier, len(self.layers) - 1)
        if ideal <= len(self.layers[-1]):
            return
        num_in_layer = min(max_num_elements, ideal)
        nn, ms = self.num_neighbors, self.max_search
        if ideal < total:  # not the last layer: half num_neighbors
            nn = max(1, nn // 2)
        layer = self.layers.pop()
        prev_layers = self.layers  # get_index() over the layers below the one being built
        self._index_elements(nn, ms, num_in_layer, prev_layers, layer, False)
        if self.reinsert_elements:
            self._index_elements(nn, max(1, ms // 2), num_in_layer, prev_layers, layer, True)
        self.layers.append(layer)

    # mod.rs:716-802
    def _index_elements(self, nn, ms, num_elements, prev_layers, layer, reinsert):
        assert len(layer) <= num_elements
        already = len(layer)
        if reinsert:
            already = 0
        else:
            layer.extend([UNUSED] * self.num_neighbors for _ in range(num_elements - len(layer)))
        order = list(range(len(layer) - 1, -1, -1) if reinsert else range(already, len(layer)))
        if self.batch_max > 0:
            pos = 0
            while pos < len(order):
                n_in_graph = len(layer) if reinsert else already + pos
                batch = min(max(1, n_in_graph // max(1, self.batch_div)), self.batch_max, len(order) - pos)
                members = order[pos:pos + batch]
                chosen = [self._select(nn, ms, prev_layers, layer, idx) for idx in members]  # graph frozen
                for idx, neighbors in zip(members, chosen):
                    if neighbors is not None:
                        self._apply(layer, idx, neighbors)
                pos += batch
        else:
            for idx in order:
                self._index_element(nn, ms, prev_layers, layer, idx)
        for i in range(len(layer)):
            self._add_and_limit_neighbors(layer[i], i, [], nn)

    # mod.rs:805-846
    def _index_element(self, nn, ms, prev_layers, layer, idx):
        neighbors = self._select(nn, ms, prev_layers, layer, idx)
        if neighbors is not None:
            self._apply(layer, idx, neighbors)

    def _select(self, nn, ms, prev_layers, layer, idx):  # :812-832, reads the graph only
        el = self.elements
        if float(dist(el[idx], el[idx])) > EPS100:
            return None
        found = search(prev_layers, el, el[idx], 1, 1)
        entrypoint = found[0][0] if found else 0
        candidates = [(i, d) for i, d in search_for_neighbors(layer, entrypoint, el, el[idx], ms) if i != idx]
        neighbors = self._select_neighbors(candidates, nn)
        if nn // 2 < len(neighbors) and neighbors[nn // 2][1] < EPS100:
            return None
        return neighbors

    def _apply(self, layer, idx, neighbors):  # :834-845, the link updates
        if layer[idx][0] == UNUSED:
            for k, (j, _) in enumerate(neighbors[: len(layer[idx])]):  # initialize_node, :886-896
                layer[idx][k] = j
        else:
            for j, d in neighbors:
                self._connect_nodes(layer[idx], idx, j, d)
        for j, d in neighbors:
            self._connect_nodes(layer[j], j, idx, d)

    # mod.rs:849-883
    def _select_neighbors(self, candidates, max_neighbors):
        if len(candidates) <= max_neighbors:
            return list(candidates)
        el = self.elements
        out = []
        for j, d in candidates:
            if len(out) >= max_neighbors:
                break
            if all(d <= float(dist(el[n], el[j])) for n, _ in out):
                out.append((j, d))
        return out

    # mod.rs:898-921
    def _connect_nodes(self, node, i, j, d):
        if i == j:
            return
        for pos, x in enumerate(node):
            if x == UNUSED or x == j:
                node[pos] = j
                return
        self._add_and_limit_neighbors(node, i, [(j, d)], len(node))

    # mod.rs:923-959
    def _add_and_limit_neighbors(self, node, node_id, extra, num_neighbors):
        el = self.elements
        neighbors = []
        for x in node:
            if x == UNUSED:
                break
            neighbors.append(x)
        candidates = [(n, float(dist(el[node_id], el[n]))) for n in neighbors] + list(extra)
        candidates.sort(key=lambda c: (c[1], c[0]))
        kept = self._select_neighbors(candidates, num_neighbors)
        for k in range(len(node)):
            node[k] = kept[k][0] if k < len(kept) else UNUSED

    def rows(self):
        """Layers as UNUSED-padded uint32 matrices, like the oracle's."""
        return [np.array(l, np.uint32).reshape(len(l), self.num_neighbors) for l in self.layers]

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

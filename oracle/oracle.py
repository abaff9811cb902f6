"""ctypes binding of the CPU oracle (oracle/libgranne_oracle.so).

TEST INFRASTRUCTURE ONLY. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this module; the product package (granne_amd) never does. See granne_oracle.h for
the reference citations of every function and for the oracle's pinning status.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgranne_oracle.so")

F32, I8 = 0, 1
UNUSED = 0xFFFFFFFF


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "granne_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libgranne_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class _Index(C.Structure):
    _fields_ = [
        ("elements", C.c_void_p),
        ("n_elements", C.c_uint64),
        ("dim", C.c_uint32),
        ("dtype", C.c_int),
        ("n_layers", C.c_uint32),
        ("layer_len", C.POINTER(C.c_uint64)),
        ("layer_rows", C.POINTER(C.c_void_p)),
        ("layer_width", C.POINTER(C.c_uint32)),
    ]


class _BuildConfig(C.Structure):
    _fields_ = [
        ("layer_multiplier", C.c_float),
        ("expected_num_elements", C.c_uint64),
        ("num_neighbors", C.c_uint32),
        ("max_search", C.c_uint32),
        ("reinsert_elements", C.c_int),
        ("n_threads", C.c_int),
        ("batch_max", C.c_uint32),
        ("batch_div", C.c_uint32),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.gro_dot_f32.restype = C.c_float
        L.gro_dot_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.gro_dot_i8.restype = None
        L.gro_dot_i8.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t] + [C.POINTER(C.c_int32)] * 3
        L.gro_normalize_f32.restype = None
        L.gro_normalize_f32.argtypes = [C.c_void_p, C.c_size_t]
        L.gro_dist_f32.restype = C.c_float
        L.gro_dist_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.gro_reference_dist_f32.restype = C.c_float
        L.gro_reference_dist_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.gro_quantize.restype = None
        L.gro_quantize.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.gro_dist_i8.restype = C.c_float
        L.gro_dist_i8.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.gro_num_elements_in_layer.restype = C.c_uint64
        L.gro_num_elements_in_layer.argtypes = [C.c_uint64, C.c_float, C.c_uint64]
        L.gro_search_for_neighbors.restype = C.c_size_t
        L.gro_search_for_neighbors.argtypes = [C.POINTER(_Index), C.c_uint32, C.c_uint64, C.c_void_p,
                                               C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        L.gro_search.restype = C.c_size_t
        L.gro_search.argtypes = [C.POINTER(_Index), C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p,
                                 C.c_void_p, C.c_void_p]
        L.gro_search_batch.restype = C.c_int
        L.gro_search_batch.argtypes = [C.POINTER(_Index), C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.gro_search_batch_timed.restype = C.c_double
        L.gro_search_batch_timed.argtypes = [C.POINTER(_Index), C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.gro_scan_topk.restype = C.c_double
        L.gro_scan_topk.argtypes = [C.POINTER(_Index), C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]
        L.gro_build_config_default.restype = None
        L.gro_build_config_default.argtypes = [C.POINTER(_BuildConfig)]
        L.gro_builder_create.restype = C.c_void_p
        L.gro_builder_create.argtypes = [C.POINTER(_BuildConfig), C.c_void_p, C.c_uint64, C.c_uint32, C.c_int]
        L.gro_builder_build_partial.restype = None
        L.gro_builder_build_partial.argtypes = [C.c_void_p, C.c_uint64]
        L.gro_builder_load_layers.restype = C.c_int
        L.gro_builder_load_layers.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.gro_builder_num_layers.restype = C.c_uint32
        L.gro_builder_num_layers.argtypes = [C.c_void_p]
        L.gro_builder_layer_len.restype = C.c_uint64
        L.gro_builder_layer_len.argtypes = [C.c_void_p, C.c_uint32]
        L.gro_builder_layer_width.restype = C.c_uint32
        L.gro_builder_layer_width.argtypes = [C.c_void_p, C.c_uint32]
        L.gro_builder_layer_rows.restype = C.c_void_p
        L.gro_builder_layer_rows.argtypes = [C.c_void_p, C.c_uint32]
        L.gro_builder_destroy.restype = None
        L.gro_builder_destroy.argtypes = [C.c_void_p]
        L.gro_select_neighbors.restype = C.c_size_t
        L.gro_select_neighbors.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t,
                                           C.c_size_t, C.c_void_p, C.c_void_p]
        L.gro_delta_encode.restype = None
        L.gro_delta_encode.argtypes = [C.c_void_p, C.c_size_t]
        L.gro_delta_decode.restype = None
        L.gro_delta_decode.argtypes = [C.c_void_p, C.c_size_t]
        L.gro_set_encode.restype = C.c_size_t
        L.gro_set_encode.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.gro_set_decode.restype = C.c_size_t
        L.gro_set_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.gro_compute_order.restype = C.c_int
        L.gro_compute_order.argtypes = [C.POINTER(_Index), C.c_void_p, C.c_int]
        L.gro_order_by_keys.restype = C.c_int
        L.gro_order_by_keys.argtypes = [C.POINTER(_Index), C.c_void_p, C.c_void_p]
        L.gro_reorder_layer.restype = None
        L.gro_reorder_layer.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p]
        L.gro_synth_rows.restype = None
        L.gro_synth_rows.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p]
        L.gro_max_threads.restype = C.c_int
        L.gro_parallel_copy.restype = None
        L.gro_parallel_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _dtype_code(a):
    if a.dtype == np.float32:
        return F32
    if a.dtype == np.int8:
        return I8
    raise TypeError("elements must be float32 or int8, got %s" % a.dtype)


# ---- src/math.rs, src/elements/* ------------------------------------------------------------
def dot_f32(x, y):
    x = np.ascontiguousarray(x, np.float32); y = np.ascontiguousarray(y, np.float32)
    return float(lib().gro_dot_f32(_p(x), _p(y), x.size))


def dot_i8(x, y):
    x = np.ascontiguousarray(x, np.int8); y = np.ascontiguousarray(y, np.int8)
    r, dx, dy = C.c_int32(), C.c_int32(), C.c_int32()
    lib().gro_dot_i8(_p(x), _p(y), x.size, C.byref(r), C.byref(dx), C.byref(dy))
    return r.value, dx.value, dy.value


def normalize_f32(x):
    """angular::Vector::from(Vec<f32>) (angular.rs:55-61). Returns a new array (rows normalised)."""
    x = np.array(x, dtype=np.float32, order="C", copy=True)
    if x.ndim == 1:
        lib().gro_normalize_f32(_p(x), x.size)
    else:
        for r in x:
            lib().gro_normalize_f32(_p(r), r.size)
    return x


def quantize(x):
    """angular_int::Vector::from(Vec<f32>) (angular_int.rs:19-45)."""
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty(x.shape, np.int8)
    if x.ndim == 1:
        lib().gro_quantize(_p(x), x.size, _p(out))
    else:
        for i in range(x.shape[0]):
            lib().gro_quantize(_p(x[i]), x.shape[1], _p(out[i]))
    return out


def dist(x, y):
    x = np.ascontiguousarray(x); y = np.ascontiguousarray(y)
    if _dtype_code(x) == F32:
        return float(lib().gro_dist_f32(_p(x), _p(y.astype(np.float32, copy=False)), x.size))
    return float(lib().gro_dist_i8(_p(x), _p(y.astype(np.int8, copy=False)), x.size))


def reference_dist_f32(x, y):
    x = np.ascontiguousarray(x, np.float32); y = np.ascontiguousarray(y, np.float32)
    return float(lib().gro_reference_dist_f32(_p(x), _p(y), x.size))


def num_elements_in_layer(total, multiplier, layer_idx):
    return int(lib().gro_num_elements_in_layer(total, multiplier, layer_idx))


def synth_rows(seed, row0, n_rows, dim):
    out = np.empty((n_rows, dim), np.float32)
    lib().gro_synth_rows(seed, row0, n_rows, dim, _p(out))
    return out


# ---- index ------------------------------------------------------------------------------------
class Index:
    """A granne index in builder/FixWidth form: elements + prefix-nested fixed-width layers."""

    def __init__(self, elements, layers):
        self.elements = np.ascontiguousarray(elements)
        assert self.elements.ndim == 2
        self.layers = [np.ascontiguousarray(l, np.uint32) for l in layers]
        n = len(self.layers)
        self._len = (C.c_uint64 * max(n, 1))(*[l.shape[0] for l in self.layers])
        self._width = (C.c_uint32 * max(n, 1))(*[l.shape[1] for l in self.layers])
        self._rows = (C.c_void_p * max(n, 1))(*[l.ctypes.data for l in self.layers])
        self._c = _Index(self.elements.ctypes.data, self.elements.shape[0], self.elements.shape[1],
                         _dtype_code(self.elements), n, self._len, self._rows, self._width)

    def __len__(self):
        return self.layers[-1].shape[0] if self.layers else 0

    @property
    def dim(self):
        return self.elements.shape[1]

    def search(self, query, max_search, num_neighbors, counters=False):
        """Granne::search (src/index/mod.rs:140-150): [(id, dist)] ascending by (dist, id)."""
        q = np.ascontiguousarray(query, self.elements.dtype)
        ids = np.empty(max(num_neighbors, 1), np.uint64)
        ds = np.empty(max(num_neighbors, 1), np.float32)
        ctr = (C.c_uint64 * 3)(0, 0, 0)
        n = lib().gro_search(C.byref(self._c), _p(q), max_search, num_neighbors, _p(ids), _p(ds), ctr)
        if n == C.c_size_t(-1).value:
            raise RuntimeError("max_search == 0 (reference panics, src/index/mod.rs:1019)")
        res = [(int(ids[i]), float(ds[i])) for i in range(n)]
        return (res, tuple(ctr)) if counters else res

    def search_for_neighbors(self, layer, entrypoint, goal, max_search):
        g = np.ascontiguousarray(goal, self.elements.dtype)
        ids = np.empty(max_search, np.uint64)
        ds = np.empty(max_search, np.float32)
        n = lib().gro_search_for_neighbors(C.byref(self._c), layer, entrypoint, _p(g), max_search, _p(ids),
                                           _p(ds), None)
        return [(int(ids[i]), float(ds[i])) for i in range(n)]

    def search_batch(self, queries, max_search, num_neighbors, n_threads=0):
        """Caller-side parallel loop over Granne::search. Returns ids[nq,k] u64, dists[nq,k] f32,
        counts[nq] u32, counters[nq,3] u64 (n_dist, n_expand, n_adj)."""
        q = np.ascontiguousarray(queries, self.elements.dtype)
        nq = q.shape[0]
        ids = np.full((nq, num_neighbors), np.iinfo(np.uint64).max, np.uint64)
        ds = np.full((nq, num_neighbors), np.inf, np.float32)
        cnt = np.zeros(nq, np.uint32)
        ctr = np.zeros((nq, 3), np.uint64)
        rc = lib().gro_search_batch(C.byref(self._c), _p(q), nq, max_search, num_neighbors, _p(ids), _p(ds),
                                    _p(cnt), _p(ctr), n_threads)
        if rc != 0:
            raise RuntimeError("max_search == 0 (reference panics, src/index/mod.rs:1019)")
        return ids, ds, cnt, ctr

    def search_batch_timed(self, queries, max_search, num_neighbors, n_threads=0, repeats=1):
        """bench.py's CPU baseline: one warm-up pass, then `repeats` timed passes over `queries` with the threads and
        their scratch kept alive (gro_search_batch_timed). Returns (seconds of the timed passes, ids, dists, counts)."""
        q = np.ascontiguousarray(queries, self.elements.dtype)
        nq = q.shape[0]
        ids = np.full((nq, num_neighbors), np.iinfo(np.uint64).max, np.uint64)
        ds = np.full((nq, num_neighbors), np.inf, np.float32)
        cnt = np.zeros(nq, np.uint32)
        sec = lib().gro_search_batch_timed(C.byref(self._c), _p(q), nq, max_search, num_neighbors, _p(ids), _p(ds),
                                           _p(cnt), n_threads, repeats)
        if sec < 0:
            raise RuntimeError("max_search == 0 (reference panics, src/index/mod.rs:1019)")
        return sec, ids, ds, cnt

    def scan_topk(self, queries, k, n_threads=0):
        """Exact k nearest elements of every query by a scan of all elements, ascending by (distance, id):
        (seconds, ids [nq, k] u64, dists [nq, k] f32)."""
        q = np.ascontiguousarray(queries, self.elements.dtype)
        ids = np.empty((q.shape[0], k), np.uint64)
        ds = np.empty((q.shape[0], k), np.float32)
        sec = lib().gro_scan_topk(C.byref(self._c), _p(q), q.shape[0], k, _p(ids), _p(ds), n_threads)
        return sec, ids, ds

    # ---- Granne::reorder (src/index/reorder.rs) -------------------------------------------------
    def compute_order(self, n_threads=0):
        """compute_order (reorder.rs:135-175): order[i] = j <=> element j moves to position i."""
        order = np.empty(len(self), np.uint64)
        if lib().gro_compute_order(C.byref(self._c), _p(order), n_threads or lib().gro_max_threads()) != 0:
            raise RuntimeError("reorder needs at least two layers (the reference panics, reorder.rs:137)")
        return order

    def order_by_keys(self, keys):
        """The order reorder_by_keys computes (reorder.rs:88-110)."""
        k = np.ascontiguousarray(keys, np.uint64)
        assert k.shape == (len(self),)
        order = np.empty(len(self), np.uint64)
        lib().gro_order_by_keys(C.byref(self._c), _p(k), _p(order))
        return order

    def reordered(self, order):
        """reorder_layers + elements.permute (reorder.rs:59-85, 210-281): a new Index."""
        order = np.ascontiguousarray(order, np.uint64)
        layers = []
        for l in self.layers:
            out = np.empty_like(l)
            lib().gro_reorder_layer(_p(l), l.shape[0], l.shape[1], _p(order), order.size, _p(out))
            layers.append(out)
        return Index(self.elements[order.astype(np.int64)], layers)


def build_index(elements, num_neighbors=30, max_search=200, layer_multiplier=15.0, reinsert_elements=True,
                expected_num_elements=0, n_threads=1, num_elements=None, batch_max=0, batch_div=8, resume_from=None):
    """GranneBuilder::new(config, elements).build() -> get_index() (src/index/mod.rs:364-488).
    n_threads=1 is the reference's `singlethreaded` feature: deterministic insertion order.
    resume_from = layers of a written index (neighbor sets as UNUSED-padded matrices): the builder
    adopts them first, GranneBuilder::from_bytes (mod.rs:430-461)."""
    elements = np.ascontiguousarray(elements)
    cfg = _BuildConfig()
    lib().gro_build_config_default(C.byref(cfg))
    cfg.layer_multiplier = layer_multiplier
    cfg.expected_num_elements = expected_num_elements
    cfg.num_neighbors = num_neighbors
    cfg.max_search = max_search
    cfg.reinsert_elements = int(bool(reinsert_elements))
    cfg.n_threads = n_threads
    cfg.batch_max = batch_max
    cfg.batch_div = batch_div
    b = lib().gro_builder_create(C.byref(cfg), _p(elements), elements.shape[0], elements.shape[1],
                                 _dtype_code(elements))
    try:
        if resume_from is not None:
            sets = [np.ascontiguousarray(l, np.uint32) for l in resume_from]
            n = len(sets)
            lens = (C.c_uint64 * max(n, 1))(*[l.shape[0] for l in sets])
            widths = (C.c_uint32 * max(n, 1))(*[l.shape[1] for l in sets])
            ptrs = (C.c_void_p * max(n, 1))(*[l.ctypes.data for l in sets])
            assert lib().gro_builder_load_layers(b, n, lens, ptrs, widths) == 0
        lib().gro_builder_build_partial(b, elements.shape[0] if num_elements is None else num_elements)
        layers = []
        for l in range(lib().gro_builder_num_layers(b)):
            n = lib().gro_builder_layer_len(b, l)
            w = lib().gro_builder_layer_width(b, l)
            ptr = lib().gro_builder_layer_rows(b, l)
            if n == 0:
                layers.append(np.zeros((0, w), np.uint32))
            else:
                arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint32)), shape=(n, w))
                layers.append(arr.copy())
    finally:
        lib().gro_builder_destroy(b)
    return Index(elements, layers)


class Builder:
    """GranneBuilder with several build_partial calls on one builder (src/index/mod.rs:364-402)."""

    def __init__(self, elements, num_neighbors=30, max_search=200, layer_multiplier=15.0, reinsert_elements=True,
                 expected_num_elements=0, n_threads=1, batch_max=0, batch_div=8):
        self.elements = np.ascontiguousarray(elements)
        cfg = _BuildConfig()
        lib().gro_build_config_default(C.byref(cfg))
        cfg.layer_multiplier = layer_multiplier
        cfg.expected_num_elements = expected_num_elements
        cfg.num_neighbors = num_neighbors
        cfg.max_search = max_search
        cfg.reinsert_elements = int(bool(reinsert_elements))
        cfg.n_threads = n_threads
        cfg.batch_max = batch_max
        cfg.batch_div = batch_div
        self._b = lib().gro_builder_create(C.byref(cfg), _p(self.elements), self.elements.shape[0],
                                           self.elements.shape[1], _dtype_code(self.elements))

    def build_partial(self, num_elements):
        lib().gro_builder_build_partial(self._b, num_elements)

    def build(self):
        self.build_partial(self.elements.shape[0])

    def layer_lens(self):
        return [int(lib().gro_builder_layer_len(self._b, l)) for l in range(lib().gro_builder_num_layers(self._b))]

    def __len__(self):
        lens = self.layer_lens()
        return lens[-1] if lens else 0

    def get_index(self):
        layers = []
        for l in range(lib().gro_builder_num_layers(self._b)):
            n, w = lib().gro_builder_layer_len(self._b, l), lib().gro_builder_layer_width(self._b, l)
            ptr = lib().gro_builder_layer_rows(self._b, l)
            layers.append(np.zeros((0, w), np.uint32) if n == 0 else
                          np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint32)), shape=(n, w)).copy())
        return Index(self.elements, layers)

    def __del__(self):
        if getattr(self, "_b", None):
            lib().gro_builder_destroy(self._b)
            self._b = None


def select_neighbors(elements, cand_ids, cand_dists, max_neighbors):
    elements = np.ascontiguousarray(elements)
    ci = np.ascontiguousarray(cand_ids, np.uint64); cd = np.ascontiguousarray(cand_dists, np.float32)
    oi = np.empty(ci.size, np.uint64); od = np.empty(ci.size, np.float32)
    n = lib().gro_select_neighbors(_p(elements), elements.shape[1], _dtype_code(elements), _p(ci), _p(cd), ci.size,
                                   max_neighbors, _p(oi), _p(od))
    return [(int(oi[i]), float(od[i])) for i in range(n)]


# ---- adjacency codec --------------------------------------------------------------------------
def delta_encode(data):
    a = np.array(data, np.uint32)
    lib().gro_delta_encode(_p(a), a.size)
    return a


def delta_decode(data):
    a = np.array(data, np.uint32)
    lib().gro_delta_decode(_p(a), a.size)
    return a


def set_encode(sorted_ids):
    a = np.ascontiguousarray(sorted_ids, np.uint32)
    out = np.empty(1 + 5 * max(4, a.size) + 8, np.uint8)
    n = lib().gro_set_encode(_p(a), a.size, _p(out))
    return bytes(out[:n])


def set_decode(enc):
    e = np.frombuffer(enc, np.uint8)
    out = np.empty(max(4, int(e[0])) + 4, np.uint32)
    n = lib().gro_set_decode(_p(e), e.size, _p(out))
    return out[:n].copy()

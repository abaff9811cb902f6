"""Host-side mirror of the reference's search interface over libgranne_hip.so.

`Granne` follows the reference's Python class (py/src/lib.rs:149-344) -- search, get_element,
get_neighbors, __len__, num_layers, layer_len -- and the Rust `Granne` it wraps
(src/index/mod.rs:106-185), with `search_batch` added (the reference has no batch API; its
callers loop or par_iter over `search`). Elements are prepared like the reference does it:
`normalize` = angular::Vector::from (src/elements/angular.rs:55-61), `quantize` =
angular_int::Vector::from (src/elements/angular_int.rs:19-45), both executed on the device.
"""
import ctypes as C
import os

import numpy as np

from ._lib import F32, I8, check, lib

DEFAULT_MAX_SEARCH = 200   # py/src/lib.rs:14
DEFAULT_NUM_ELEMENTS = 10  # py/src/lib.rs:15

_ELEMENT_TYPES = {"angular": (F32, np.float32), "angular_int": (I8, np.int8)}


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def normalize(rows, device=0):
    """angular::Vector::from for every row of `rows` (float32). Returns a new array."""
    a = np.array(rows, dtype=np.float32, order="C", copy=True)
    flat = a.reshape(1, -1) if a.ndim == 1 else a
    if flat.size:
        check(lib().granne_hip_normalize_f32(_p(flat), flat.shape[0], flat.shape[1], device))
    return a


def quantize(rows, device=0):
    """angular_int::Vector::from for every row of `rows` (float32 in, int8 out)."""
    a = np.ascontiguousarray(rows, dtype=np.float32)
    flat = a.reshape(1, -1) if a.ndim == 1 else a
    out = np.empty(flat.shape, np.int8)
    if flat.size:
        check(lib().granne_hip_quantize_f32(_p(flat), _p(out), flat.shape[0], flat.shape[1], device))
    return out.reshape(a.shape)


class Granne:
    """An HNSW index resident in the HBM of one MI355X."""

    def __init__(self, element_type, elements, layers, device=0, prepared=True):
        """element_type: "angular" (f32) or "angular_int" (int8).
        elements: [n, dim] array. With prepared=True (default) rows are taken as stored in a
        Vectors file (already normalised / quantised); with prepared=False raw float rows go
        through Vector::from first.
        layers: list of [layer_len, width] uint32 arrays (UNUSED padded), top layer first --
        what GranneBuilder::get_index hands to Granne (src/index/mod.rs:483-488)."""
        et = element_type.lower()
        if et not in _ELEMENT_TYPES:
            raise ValueError("Invalid element type")  # the reference panics (py/src/lib.rs:210)
        self.element_type = et
        self.dtype_code, self.np_dtype = _ELEMENT_TYPES[et]
        self.device = device
        if not prepared:
            elements = normalize(elements, device) if et == "angular" else quantize(elements, device)
        el = np.ascontiguousarray(elements, dtype=self.np_dtype)
        if el.ndim != 2:
            raise ValueError("elements must be [n, dim]")
        layers = [np.ascontiguousarray(l, dtype=np.uint32) for l in layers]
        n = len(layers)
        lens = (C.c_uint64 * max(n, 1))(*[l.shape[0] for l in layers])
        widths = (C.c_uint32 * max(n, 1))(*[l.shape[1] for l in layers])
        rows = (C.c_void_p * max(n, 1))(*[l.ctypes.data for l in layers])
        h = C.c_void_p()
        check(lib().granne_hip_index_create(C.byref(h), _p(el), el.shape[0], el.shape[1], self.dtype_code, n, lens,
                                            rows, widths, device))
        self._h = h
        self.dim = el.shape[1]

    @classmethod
    def from_csr(cls, element_type, elements, offsets, ids, device=0):
        """From decoded on-disk layers: per layer (offsets[len+1] u64, ids u32)."""
        self = cls.__new__(cls)
        et = element_type.lower()
        if et not in _ELEMENT_TYPES:
            raise ValueError("Invalid element type")
        self.element_type = et
        self.dtype_code, self.np_dtype = _ELEMENT_TYPES[et]
        self.device = device
        el = np.ascontiguousarray(elements, dtype=self.np_dtype)
        offs = [np.ascontiguousarray(o, dtype=np.uint64) for o in offsets]
        idl = [np.ascontiguousarray(i, dtype=np.uint32) for i in ids]
        n = len(offs)
        lens = (C.c_uint64 * max(n, 1))(*[o.size - 1 for o in offs])
        po = (C.c_void_p * max(n, 1))(*[o.ctypes.data for o in offs])
        pi = (C.c_void_p * max(n, 1))(*[i.ctypes.data if i.size else None for i in idl])
        h = C.c_void_p()
        check(lib().granne_hip_index_create_csr(C.byref(h), _p(el), el.shape[0], el.shape[1], self.dtype_code, n,
                                                lens, po, pi, device))
        self._h = h
        self.dim = el.shape[1]
        return self

    @classmethod
    def from_files(cls, index_path, element_type, elements_path, device=0):
        """Granne(index_path, element_type, elements_path) of the reference's binding
        (py/src/lib.rs:177-214): an index file written by write_index + a Vectors file."""
        self = cls.__new__(cls)
        et = element_type.lower()
        if et not in _ELEMENT_TYPES:
            raise ValueError("Invalid element type")
        self.element_type = et
        self.dtype_code, self.np_dtype = _ELEMENT_TYPES[et]
        self.device = device
        h = C.c_void_p()
        check(lib().granne_hip_index_load_files(C.byref(h), os.fsencode(index_path), os.fsencode(elements_path),
                                                self.dtype_code, device))
        self._h = h
        self.dim = int(lib().granne_hip_index_dim(h))
        return self

    @classmethod
    def from_bytes(cls, index_bytes, element_type, elements_bytes, device=0):
        """Granne::from_bytes (src/index/mod.rs:106-113) over Vectors::from_bytes."""
        self = cls.__new__(cls)
        et = element_type.lower()
        if et not in _ELEMENT_TYPES:
            raise ValueError("Invalid element type")
        self.element_type = et
        self.dtype_code, self.np_dtype = _ELEMENT_TYPES[et]
        self.device = device
        ib = np.frombuffer(index_bytes, np.uint8)
        eb = np.frombuffer(elements_bytes, np.uint8)
        h = C.c_void_p()
        check(lib().granne_hip_index_load(C.byref(h), _p(ib), ib.size, _p(eb), eb.size, self.dtype_code, device))
        self._h = h
        self.dim = int(lib().granne_hip_index_dim(h))
        return self

    def save_index(self, path):
        """py/src/lib.rs:318-330."""
        check(lib().granne_hip_index_save(self._h, os.fsencode(path), None))

    def index_bytes(self):
        """Index::write_index into a buffer (src/index/mod.rs:67-70): the bytes of the index file."""
        out, n = C.c_void_p(), C.c_uint64(0)
        check(lib().granne_hip_index_encode(self._h, C.byref(out), C.byref(n)))
        try:
            return C.string_at(out, n.value)
        finally:
            lib().granne_hip_bytes_free(out)

    def save_elements(self, path):
        """py/src/lib.rs:332-343."""
        check(lib().granne_hip_index_save(self._h, None, os.fsencode(path)))

    @classmethod
    def from_device(cls, element_type, d_elements_ptr, n_elements, dim, layer_lens, d_layer_ptrs, layer_widths,
                    device=0, stream=0):
        """Elements and fixed-width layers already in device memory (raw pointers)."""
        self = cls.__new__(cls)
        et = element_type.lower()
        self.element_type = et
        self.dtype_code, self.np_dtype = _ELEMENT_TYPES[et]
        self.device = device
        n = len(layer_lens)
        lens = (C.c_uint64 * max(n, 1))(*layer_lens)
        widths = (C.c_uint32 * max(n, 1))(*layer_widths)
        rows = (C.c_void_p * max(n, 1))(*d_layer_ptrs)
        h = C.c_void_p()
        check(lib().granne_hip_index_create_device(C.byref(h), C.c_void_p(d_elements_ptr), n_elements, dim,
                                                   self.dtype_code, n, lens, rows, widths, device,
                                                   C.c_void_p(stream)))
        self._h = h
        self.dim = dim
        return self

    def close(self):
        if getattr(self, "_h", None):
            lib().granne_hip_index_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- Index trait (src/index/mod.rs:54-71) --------------------------------------------------
    def __len__(self):
        return int(lib().granne_hip_index_len(self._h))

    def num_layers(self):
        return int(lib().granne_hip_index_num_layers(self._h))

    def layer_len(self, layer):
        return int(lib().granne_hip_index_layer_len(self._h, layer))

    def get_neighbors(self, idx, layer=None):
        if layer is None:
            layer = self.num_layers() - 1
        buf = np.empty(512, np.uint32)
        n = C.c_uint32()
        check(lib().granne_hip_index_get_neighbors(self._h, idx, layer, _p(buf), buf.size, C.byref(n)))
        return buf[: n.value].astype(np.int64).tolist()

    def get_element(self, idx):
        out = np.empty(self.dim, self.np_dtype)
        check(lib().granne_hip_index_get_element(self._h, idx, _p(out)))
        return out

    def hbm_bytes(self):
        return int(lib().granne_hip_index_hbm_bytes(self._h))

    # ---- reorder (src/index/reorder.rs) --------------------------------------------------------------
    def reorder(self):
        """Granne::reorder: places similar elements closer together, in place. Returns the permutation
        (uint64): permutation[i] == j means the element with idx j has been moved to idx i."""
        order = np.empty(len(self), np.uint64)
        check(lib().granne_hip_index_reorder(self._h, _p(order)))
        return order

    def reorder_by_keys(self, keys):
        """Granne::reorder_by_keys: layer-preserving sort by (key, idx); keys are u64."""
        k = np.ascontiguousarray(keys, dtype=np.uint64)
        if k.shape != (len(self),):
            raise ValueError("need one key per element")  # assert_eq!(self.len(), keys.len()), reorder.rs:91
        order = np.empty(len(self), np.uint64)
        check(lib().granne_hip_index_reorder_by_keys(self._h, _p(k), _p(order)))
        return order

    # ---- options --------------------------------------------------------------------------------
    def set_option(self, option, value):
        check(lib().granne_hip_index_set_option(self._h, option, value))

    def get_option(self, option):
        import ctypes as C
        v = C.c_uint64(0)
        check(lib().granne_hip_index_get_option(self._h, option, C.byref(v)))
        return int(v.value)

    def last_slow_count(self):
        return int(lib().granne_hip_index_last_slow_count(self._h))

    # ---- search -----------------------------------------------------------------------------------
    def _prepare(self, element, prepared):
        if prepared:
            return np.ascontiguousarray(element, dtype=self.np_dtype)
        return normalize(element, self.device) if self.element_type == "angular" else quantize(element, self.device)

    def search(self, element, max_search=DEFAULT_MAX_SEARCH, num_elements=DEFAULT_NUM_ELEMENTS, prepared=True):
        """Granne.search (py/src/lib.rs:227-233): [(id, distance)] ascending by (distance, id).
        prepared=False applies Vector::from to `element` first, as the reference's binding does."""
        ids, dists, counts = self.search_batch(np.asarray(element).reshape(1, -1), max_search, num_elements, prepared)
        return [(int(ids[0, i]), float(dists[0, i])) for i in range(int(counts[0]))]

    def search_batch(self, elements, max_search=DEFAULT_MAX_SEARCH, num_elements=DEFAULT_NUM_ELEMENTS, prepared=True,
                     stats=False):
        """nq independent searches on the GPU. Returns ids [nq,k] uint64, dists [nq,k] float32,
        counts [nq] uint32 (and stats [nq,3] uint64 = n_dist, n_expand, n_adj when stats=True)."""
        q = self._prepare(elements, prepared)
        if q.ndim != 2 or q.shape[1] != self.dim:
            raise ValueError("queries must be [nq, %d]" % self.dim)
        nq, k = q.shape[0], int(num_elements)
        ids = np.empty((nq, max(k, 0)), np.uint64)
        dists = np.empty((nq, max(k, 0)), np.float32)
        counts = np.zeros(nq, np.uint32)
        st = np.zeros((nq, 3), np.uint64)
        check(lib().granne_hip_search_batch(self._h, _p(q), nq, int(max_search), k, _p(ids), _p(dists), _p(counts),
                                            _p(st)))
        return (ids, dists, counts, st) if stats else (ids, dists, counts)

    def search_batch_device(self, d_queries, nq, max_search, num_elements, d_ids, d_dists, d_counts, d_stats=0,
                            d_status=0, stream=0):
        """Device-resident, asynchronous variant: every argument is a raw device pointer (int)."""
        check(lib().granne_hip_search_batch_device(self._h, C.c_void_p(d_queries), nq, int(max_search),
                                                   int(num_elements), C.c_void_p(d_ids), C.c_void_p(d_dists),
                                                   C.c_void_p(d_counts), C.c_void_p(d_stats), C.c_void_p(d_status),
                                                   C.c_void_p(stream)))

    def search_batches_device(self, d_queries, nq, max_search, num_elements, d_ids, d_dists, d_counts, d_stats=None,
                              d_status=0, stream=0):
        """Several batches of nq queries through ONE launch (granne_hip_search_batches_device). d_queries, d_ids,
        d_dists, d_counts (and d_stats, optional): sequences of raw device pointers (int), one per batch -- or ctypes
        pointer arrays prepared once with `pointer_array`. Asynchronous on `stream`."""
        n = len(d_queries)
        arr = lambda v: v if isinstance(v, C.Array) else pointer_array(v)  # noqa: E731
        check(lib().granne_hip_search_batches_device(self._h, n, arr(d_queries), int(nq), int(max_search), int(num_elements),
                                                     arr(d_ids), arr(d_dists), arr(d_counts),
                                                     arr(d_stats) if d_stats is not None else None, C.c_void_p(d_status),
                                                     C.c_void_p(stream)))

    def search_begin_device(self, d_queries, nq, max_search, num_elements, d_ids, d_dists, d_counts, d_stats=0, d_status=0,
                            stream=0):
        """granne_hip_search_begin_device: the batch runs beside `stream` on a stream of the index; returns the ticket."""
        t = C.c_uint64(0)
        check(lib().granne_hip_search_begin_device(self._h, C.c_void_p(d_queries), int(nq), int(max_search), int(num_elements),
                                                   C.c_void_p(d_ids), C.c_void_p(d_dists), C.c_void_p(d_counts),
                                                   C.c_void_p(d_stats), C.c_void_p(d_status), C.c_void_p(stream), C.byref(t)))
        return int(t.value)

    def search_end_device(self, ticket, stream=0):
        """granne_hip_search_end_device: `stream` continues after the batch of `ticket`."""
        check(lib().granne_hip_search_end_device(self._h, C.c_uint64(ticket), C.c_void_p(stream)))

    def search_batch_device_timed(self, d_queries, nq, max_search, num_elements, d_ids, d_dists, d_counts, d_stats,
                                  d_status, stream, ev_before, ev_after):
        """search_batch_device plus two raw hipEvent_t recorded around the search kernel's dispatch."""
        check(lib().granne_hip_search_batch_device_timed(self._h, C.c_void_p(d_queries), nq, int(max_search),
                                                         int(num_elements), C.c_void_p(d_ids), C.c_void_p(d_dists),
                                                         C.c_void_p(d_counts), C.c_void_p(d_stats), C.c_void_p(d_status),
                                                         C.c_void_p(stream), C.c_void_p(ev_before), C.c_void_p(ev_after)))

    def brute_force_device(self, d_queries, nq, k, d_ids, d_dists, d_counts, stream=0):
        """Exact k nearest elements of every query by a scan of all elements on the matrix cores
        (granne_hip_brute_force_device). Raw device pointers (int), asynchronous on `stream`."""
        check(lib().granne_hip_brute_force_device(self._h, C.c_void_p(d_queries), int(nq), int(k), C.c_void_p(d_ids),
                                                  C.c_void_p(d_dists), C.c_void_p(d_counts), C.c_void_p(stream)))

    def brute_force(self, queries, k, prepared=True):
        """Host convenience over brute_force_device (torch moves the buffers): ids [nq, k] u64, dists [nq, k] f32,
        counts [nq] u32, ascending by (distance, id) -- the exact answer Granne::search approximates.
        prepared=False applies Vector::from to the queries first, as `search` does."""
        import torch
        q = self._prepare(queries, prepared)
        if q.ndim == 1:
            q = q[None]
        if q.ndim != 2 or q.shape[1] != self.dim:
            raise ValueError("queries must be [nq, %d]" % self.dim)
        dev = torch.device("cuda", self.device)
        tq = torch.from_numpy(q.view(np.uint8).reshape(q.shape[0], -1)).to(dev)
        ids = torch.empty((q.shape[0], k), dtype=torch.int64, device=dev)
        ds = torch.empty((q.shape[0], k), dtype=torch.float32, device=dev)
        cnt = torch.empty(q.shape[0], dtype=torch.int32, device=dev)
        self.brute_force_device(tq.data_ptr(), q.shape[0], k, ids.data_ptr(), ds.data_ptr(), cnt.data_ptr(),
                                torch.cuda.current_stream(dev).cuda_stream)
        torch.cuda.synchronize(dev)
        return ids.cpu().numpy().astype(np.uint64), ds.cpu().numpy(), cnt.cpu().numpy().astype(np.uint32)

    def dists_device(self, d_queries, nq, d_ids, m, d_out, d_status=0, stream=0):
        """ElementContainer::dists (src/elements/mod.rs:35-39) batched on device: out[q, j] =
        dist(element ids[q, j], query q). Raw device pointers (int), asynchronous on `stream`."""
        check(lib().granne_hip_dists_device(self._h, C.c_void_p(d_queries), int(nq), C.c_void_p(d_ids), int(m),
                                            C.c_void_p(d_out), C.c_void_p(d_status), C.c_void_p(stream)))

    def dists_many(self, queries, ids):
        """ElementContainer::dists for a batch: queries [nq, dim] (prepared), ids [nq, m] -> [nq, m] f32
        (+inf where an id is out of range)."""
        import torch
        q = np.ascontiguousarray(queries, dtype=self.np_dtype)
        ii = np.ascontiguousarray(ids, dtype=np.uint32)
        if q.ndim != 2 or q.shape[1] != self.dim or ii.ndim != 2 or ii.shape[0] != q.shape[0]:
            raise ValueError("queries must be [nq, %d] and ids [nq, m]" % self.dim)
        dev = torch.device("cuda", self.device)
        tq = torch.from_numpy(q.view(np.uint8).reshape(q.shape[0], -1)).to(dev)
        ti = torch.from_numpy(ii.view(np.int32)).to(dev)
        out = torch.empty(ii.shape, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            s = torch.cuda.current_stream().cuda_stream
            self.dists_device(tq.data_ptr(), q.shape[0], ti.data_ptr(), ii.shape[1], out.data_ptr(), 0, s)
            torch.cuda.synchronize()
        return out.cpu().numpy()

    def dists(self, queries, qidx, ids):
        """ElementContainer::dist_to_element for explicit (query, element) pairs."""
        q = np.ascontiguousarray(queries, dtype=self.np_dtype)
        qi = np.ascontiguousarray(qidx, dtype=np.uint32)
        ii = np.ascontiguousarray(ids, dtype=np.uint32)
        out = np.empty(qi.size, np.float32)
        check(lib().granne_hip_dist_pairs(self._h, _p(q), q.shape[0], _p(qi), _p(ii), qi.size, _p(out)))
        return out


def pointer_array(ptrs):
    """A ctypes array of device pointers (ints) for the multi-batch calls; build it once when the buffers are fixed."""
    return (C.c_void_p * len(ptrs))(*[int(x) for x in ptrs])


def compute_distance(element_type, a, b, device=0):
    """compute_distance (py/src/lib.rs:58-86): both vectors go through Vector::from."""
    et = element_type.lower()
    if et not in _ELEMENT_TYPES:
        raise ValueError("Unsupported element type")
    prep = normalize if et == "angular" else quantize
    pa = prep(np.asarray(a, np.float32), device).reshape(1, -1)
    pb = prep(np.asarray(b, np.float32), device).reshape(1, -1)
    ix = Granne(et, pa, [], device=device)
    try:
        return float(ix.dists(pb, [0], [0])[0])
    finally:
        ix.close()

"""GranneBuilder on the GPU: host-side mirror of the reference's builder interface
(py/src/lib.rs:346-579 over src/index/mod.rs:198-531) on top of granne_hip_builder_*."""
import ctypes as C

import numpy as np

from ._lib import BUILD_ALL, BuildConfig, check, lib
from .index import _ELEMENT_TYPES, Granne, _p, normalize, quantize


class GranneBuilder:
    def __init__(self, element_type, elements=None, layer_multiplier=None, expected_num_elements=None,
                 num_neighbors=None, max_search=None, reinsert_elements=True, show_progress=False, device=0,
                 prepared=True, batch_max=None, batch_div=None):
        """Keyword arguments follow the reference's GranneBuilder.__new__ (py/src/lib.rs:384-405);
        unset ones take BuildConfig::default() (src/index/mod.rs:220-231: multiplier 15,
        num_neighbors 30, max_search 200). batch_max / batch_div tune the GPU insertion schedule
        (include/granne_hip.h)."""
        et = element_type.lower()
        if et not in _ELEMENT_TYPES:
            raise ValueError("Invalid element type")
        self.element_type = et
        self.dtype_code, self.np_dtype = _ELEMENT_TYPES[et]
        self.device = device
        cfg = BuildConfig()
        lib().granne_hip_build_config_default(C.byref(cfg))
        if layer_multiplier is not None:
            cfg.layer_multiplier = layer_multiplier
        if expected_num_elements is not None:
            cfg.expected_num_elements = expected_num_elements
        if num_neighbors is not None:
            cfg.num_neighbors = num_neighbors
        if max_search is not None:
            cfg.max_search = max_search
        cfg.reinsert_elements = int(bool(reinsert_elements))
        cfg.show_progress = int(bool(show_progress))
        if batch_max is not None:
            cfg.batch_max = batch_max
        if batch_div is not None:
            cfg.batch_div = batch_div
        self.config = cfg
        self._pending = []
        self._prepared = prepared
        self._h = None
        self.dim = None
        if elements is not None:
            el = np.asarray(elements)
            if el.ndim != 2:
                raise ValueError("elements must be [n, dim]")
            self._pending.append(self._prep(el))
            self.dim = el.shape[1]

    def _prep(self, rows):
        if self._prepared:
            return np.ascontiguousarray(rows, dtype=self.np_dtype)
        return normalize(rows, self.device) if self.element_type == "angular" else quantize(rows, self.device)

    @classmethod
    def from_device(cls, element_type, d_elements_ptr, n_elements, dim, device=0, stream=0, **kwargs):
        """Elements already in device memory: dense [n][dim] rows, prepared."""
        self = cls(element_type, None, device=device, **kwargs)
        h = C.c_void_p()
        check(lib().granne_hip_builder_create_device(C.byref(h), C.byref(self.config), C.c_void_p(d_elements_ptr),
                                                     n_elements, dim, self.dtype_code, device, C.c_void_p(stream)))
        self._h = h
        self.dim = dim
        return self

    def append(self, element):
        """GranneBuilder.append (py/src/lib.rs:487-489): the element is indexed by the next build()."""
        row = np.asarray(element).reshape(1, -1)
        if self.dim is None:
            self.dim = row.shape[1]
        self._pending.append(self._prep(row))

    def _ensure(self):
        if self._h is not None and self._pending and int(lib().granne_hip_builder_num_elements(self._h)) == 0:
            self.close()  # an empty builder takes its dimension from the first element pushed
            self.dim = self._pending[0].shape[1]
        if self._h is None:
            if self._pending:
                el = np.ascontiguousarray(np.concatenate(self._pending, axis=0))
            else:
                el = np.zeros((0, self.dim or 1), self.np_dtype)
            self.dim = el.shape[1]
            h = C.c_void_p()
            check(lib().granne_hip_builder_create(C.byref(h), C.byref(self.config), _p(el), el.shape[0], el.shape[1],
                                                  self.dtype_code, self.device))
            self._h = h
            self._pending = []
        elif self._pending:  # elements appended after the builder went to the device
            el = np.ascontiguousarray(np.concatenate(self._pending, axis=0))
            if el.shape[1] != self.dim:
                raise ValueError("dimension mismatch")
            check(lib().granne_hip_builder_append(self._h, _p(el), el.shape[0]))
            self._pending = []

    def load_index(self, index):
        """GranneBuilder::from_bytes / from_file (src/index/mod.rs:430-469): adopt the layers of a written
        index (bytes or a path) before building on; lists are resized to num_neighbors."""
        self._ensure()
        data = index if isinstance(index, (bytes, bytearray, memoryview)) else open(index, "rb").read()
        buf = np.frombuffer(data, np.uint8)
        check(lib().granne_hip_builder_load_index(self._h, buf.ctypes.data_as(C.c_void_p), buf.size))

    def build(self, num_elements=None):
        """GranneBuilder.build (py/src/lib.rs:499-507): all elements, or the first num_elements."""
        self._ensure()
        check(lib().granne_hip_builder_build(self._h, BUILD_ALL if num_elements is None else int(num_elements)))

    def __len__(self):
        self._ensure()
        return int(lib().granne_hip_builder_len(self._h))

    def num_elements(self):
        self._ensure()
        return int(lib().granne_hip_builder_num_elements(self._h))

    def num_layers(self):
        self._ensure()
        return int(lib().granne_hip_builder_num_layers(self._h))

    def layer_len(self, layer):
        self._ensure()
        return int(lib().granne_hip_builder_layer_len(self._h, layer))

    def get_layer(self, layer):
        """The layer as the reference's builder holds it: [layer_len, num_neighbors] uint32."""
        self._ensure()
        out = np.empty((self.layer_len(layer), self.config.num_neighbors), np.uint32)
        if out.size:
            check(lib().granne_hip_builder_get_layer(self._h, layer, _p(out)))
        return out

    def layers(self):
        return [self.get_layer(l) for l in range(self.num_layers())]

    def get_neighbors(self, idx, layer=None):
        if layer is None:
            layer = self.num_layers() - 1
        row = self.get_layer(layer)[idx]
        return [int(x) for x in row if x != 0xFFFFFFFF]

    def get_index(self):
        """GranneBuilder::get_index (src/index/mod.rs:483-488): a searchable Granne on the same device."""
        self._ensure()
        h = C.c_void_p()
        check(lib().granne_hip_builder_get_index(self._h, C.byref(h)))
        ix = Granne.__new__(Granne)
        ix.element_type = self.element_type
        ix.dtype_code, ix.np_dtype = self.dtype_code, self.np_dtype
        ix.device = self.device
        ix._h = h
        ix.dim = self.dim
        return ix

    def save_index(self, path):
        """GranneBuilder.save_index (py/src/lib.rs:509-521): the compressed on-disk form."""
        import os
        layers = self.layers()
        n = len(layers)
        lens = (C.c_uint64 * max(n, 1))(*[l.shape[0] for l in layers])
        widths = (C.c_uint32 * max(n, 1))(*[l.shape[1] for l in layers])
        rows = (C.c_void_p * max(n, 1))(*[l.ctypes.data for l in layers])
        check(lib().granne_hip_write_index_file(os.fsencode(path), n, lens, rows, widths))

    def save_elements(self, path):
        """GranneBuilder.save_elements (py/src/lib.rs:523-535)."""
        ix = self.get_index()
        try:
            ix.save_elements(path)
        finally:
            ix.close()

    def close(self):
        if getattr(self, "_h", None):
            lib().granne_hip_builder_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

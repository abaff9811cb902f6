"""granne's index / elements files: the product's host-side codec (C ABI, no GPU needed) against
the independent Python restatement in oracle/fileformat.py, plus the reference's known answers
(src/slice_vector/set_vector.rs:250-303: push_and_get*, the '4 bytes per number' case)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from granne_amd import _lib, build
from oracle import fileformat as off
from tests.conftest import random_floats

UNUSED = 0xFFFFFFFF


@pytest.fixture(scope="module")
def lib():
    build.build_library()
    return _lib.lib()


def product_write_index(lib, path, layers):
    layers = [np.ascontiguousarray(l, np.uint32) for l in layers]
    n = len(layers)
    lens = (C.c_uint64 * max(n, 1))(*[l.shape[0] for l in layers])
    widths = (C.c_uint32 * max(n, 1))(*[l.shape[1] for l in layers])
    rows = (C.c_void_p * max(n, 1))(*[l.ctypes.data for l in layers])
    _lib.check(lib.granne_hip_write_index_file(os.fsencode(path), n, lens, rows, widths))
    return open(path, "rb").read()


def product_decode(lib, buf):
    b = np.frombuffer(buf, np.uint8)
    n = C.c_uint32()
    lens = (C.c_uint64 * 64)()
    nids = (C.c_uint64 * 64)()
    _lib.check(lib.granne_hip_index_file_info(b.ctypes.data_as(C.c_void_p), b.size, C.byref(n), lens, nids, 64))
    out = []
    for l in range(n.value):
        offs = np.zeros(lens[l] + 1, np.uint64)
        ids = np.zeros(max(nids[l], 1), np.uint32)
        _lib.check(lib.granne_hip_index_file_decode_layer(b.ctypes.data_as(C.c_void_p), b.size, l,
                                                          offs.ctypes.data_as(C.c_void_p), ids.ctypes.data_as(C.c_void_p)))
        out.append([ids[int(offs[i]):int(offs[i + 1])].tolist() for i in range(lens[l])])
    return out


def random_layers(rng, sizes, width, max_id_of):
    layers = []
    for n in sizes:
        l = np.full((n, width), UNUSED, np.uint32)
        for i in range(n):
            d = int(rng.integers(0, min(width, max_id_of(n)) + 1))
            l[i, :d] = rng.choice(max_id_of(n), d, replace=False)
        layers.append(l)
    return layers


@pytest.mark.parametrize("sizes,width", [([3, 40, 700], 30), ([59], 8), ([60], 8), ([61], 8), ([119, 120, 121], 5),
                                          ([1], 30), ([7, 100, 1500, 20000], 30)])
def test_index_files_written_by_product_and_by_oracle_are_byte_identical(lib, tmp_path, sizes, width):
    rng = np.random.default_rng(sum(sizes))
    layers = random_layers(rng, sizes, width, lambda n: n)
    got = product_write_index(lib, str(tmp_path / "a.granne"), layers)
    want = off.write_index(layers)
    assert got[:6] == b"granne" and len(got) >= 1024
    assert got == want
    meta = json.loads(got[6:1024].decode())
    assert meta["version"] == 2 and meta["compressed"] is True and meta["num_layers"] == len(sizes)
    assert meta["layer_counts"] == sizes and sum(meta["layer_sizes"]) == len(got) - 1024
    # and both readers recover the sorted neighbor sets (cf. write_and_load, src/index/tests.rs:337-394)
    sets = [[sorted(int(x) for x in r if x != UNUSED) for r in l] for l in layers]
    assert product_decode(lib, got) == sets
    assert off.read_index(got)[1] == sets


def test_large_ids_use_the_raw_form_and_still_roundtrip(lib, tmp_path):
    """Neighbor ids spread over 4e9: deltas need 4 bytes, stream-vbyte is not smaller, records fall
    back to raw u32 (src/slice_vector/set_vector.rs:137-143, test at :272-283)."""
    l = np.full((4, 6), UNUSED, np.uint32)
    l[0, :2] = [660380, 37717]
    l[1, :6] = [4000000000, 5, 3000000000, 2000000000, 1000000000, 70000]
    l[3, :1] = [5]
    got = product_write_index(lib, str(tmp_path / "b.granne"), [l])
    assert got == off.write_index([l])
    assert product_decode(lib, got) == [[[37717, 660380], [5, 70000, 1000000000, 2000000000, 3000000000, 4000000000], [], [5]]]


def test_elements_file(lib, tmp_path, oracle):
    rng = np.random.default_rng(3)
    for arr in (oracle.normalize_f32(random_floats(rng, 17, 25)), oracle.quantize(random_floats(rng, 9, 100))):
        path = str(tmp_path / "e.bin")
        _lib.check(lib.granne_hip_write_elements_file(os.fsencode(path), arr.ctypes.data_as(C.c_void_p), arr.shape[0],
                                                      arr.shape[1], 0 if arr.dtype == np.float32 else 1))
        buf = open(path, "rb").read()
        assert buf == off.write_elements(arr)
        assert (off.read_elements(buf, arr.dtype) == arr).all()


def test_malformed_files_are_rejected(lib):
    b = np.frombuffer(b"nope" + b" " * 2000, np.uint8)
    n = C.c_uint32()
    assert lib.granne_hip_index_file_info(b.ctypes.data_as(C.c_void_p), b.size, C.byref(n), None, None, 0) == _lib.ERR_IO
    assert b"Library string" in lib.granne_hip_last_error()
    good = off.write_index([np.full((5, 4), UNUSED, np.uint32)])
    trunc = np.frombuffer(good[:-3], np.uint8)
    assert lib.granne_hip_index_file_info(trunc.ctypes.data_as(C.c_void_p), trunc.size, C.byref(n), None, None, 0) == _lib.ERR_IO


def test_oracle_built_index_roundtrips_through_the_file(lib, tmp_path, oracle):
    rng = np.random.default_rng(4)
    el = oracle.normalize_f32(random_floats(rng, 800, 16))
    ix = oracle.build_index(el, num_neighbors=10, max_search=20)
    buf = product_write_index(lib, str(tmp_path / "c.granne"), ix.layers)
    dec = product_decode(lib, buf)
    for layer, nodes in zip(ix.layers, dec):
        for row, ids in zip(layer, nodes):
            assert sorted(int(x) for x in row if x != UNUSED) == ids


def test_corrupt_header_sizes_are_rejected_not_wrapped(lib, tmp_path):
    """A layer_sizes entry near 2^64 must not wrap the bounds check, and a digit string that overflows
    u64 must not be read modulo 2^64 (the reference panics safely on the slice bounds, io.rs:75-84)."""
    rng = np.random.default_rng(5)
    layers = random_layers(rng, [3, 40], 6, lambda n: n)
    buf = bytearray(product_write_index(lib, str(tmp_path / "ok.granne"), layers))
    head = bytes(buf[:1024]).decode()
    meta = json.loads(head[6:].strip())
    good = meta["layer_sizes"][1]

    def with_sizes(sizes_text):
        h = head.replace('"layer_sizes":[%d,%d]' % (meta["layer_sizes"][0], good), '"layer_sizes":%s' % sizes_text)
        assert h != head
        h = h.rstrip(" ")
        assert len(h) <= 1024
        return h.encode().ljust(1024, b" ") + bytes(buf[1024:])

    def info(b):
        a = np.frombuffer(b, np.uint8)
        n = C.c_uint32()
        lens = (C.c_uint64 * 64)()
        nids = (C.c_uint64 * 64)()
        return lib.granne_hip_index_file_info(a.ctypes.data_as(C.c_void_p), a.size, C.byref(n), lens, nids, 64)

    assert info(bytes(buf)) == 0
    first = meta["layer_sizes"][0]
    # start + size wraps to a small number in u64
    assert info(with_sizes("[%d,%d]" % (first, 2 ** 64 - 1024 - first + 8))) == _lib.ERR_IO
    assert info(with_sizes("[%d,%d]" % (first, 2 ** 64 - 1))) == _lib.ERR_IO
    # 2^64 + good reads as `good` modulo 2^64
    assert info(with_sizes("[%d,%d]" % (first, 2 ** 64 + good))) == _lib.ERR_IO
    assert info(with_sizes("[%d,%d]" % (first, good + 1))) == _lib.ERR_IO  # plain truncation


def test_layers_encoded_by_several_threads_equal_the_one_thread_file(lib, tmp_path):
    """Layers of 65536 nodes and more are encoded by up to 16 host threads over consecutive node ranges (a 125M-node layer
    took one thread 69 s): same bytes as the oracle's writer, which walks the nodes in order."""
    rng = np.random.default_rng(70000)
    n, width = 70_001, 6
    layer = np.full((n, width), UNUSED, np.uint32)
    deg = rng.integers(0, width + 1, n)
    vals = rng.integers(0, n, (n, width)).astype(np.uint32)
    for d in range(1, width + 1):  # rows of d distinct-enough ids (duplicates are legal in the file: the set form sorts them)
        m = deg >= d
        layer[m, d - 1] = vals[m, d - 1]
    small = np.full((5, width), UNUSED, np.uint32)
    small[:, 0] = np.arange(5, dtype=np.uint32)[::-1]
    got = product_write_index(lib, str(tmp_path / "big.granne"), [small, layer])
    assert got == off.write_index([small, layer])

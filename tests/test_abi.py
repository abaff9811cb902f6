"""CPU-side checks of the drop-in boundary: libgranne_hip.so builds for gfx950, loads, and exports
exactly the symbols include/granne_hip.h declares. No compute is attempted without a GPU, and
the library must refuse -- not fall back -- when no device is present."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from granne_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build_library()
    return _lib.lib()


def _declared():
    text = open(os.path.join(ROOT, "include", "granne_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(granne_hip_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree(lib):
    declared = _declared()
    assert len(declared) >= 25
    assert sorted(_lib.SIGNATURES) == declared


def test_every_declared_symbol_is_exported(lib):
    raw = C.CDLL(build.LIB_PATH)
    for name in _declared():
        assert hasattr(raw, name), name


RUST_GPU_RS = os.path.join(ROOT, "rust", "granne-hip", "src", "gpu.rs")


def test_rust_crate_is_on_disk_and_integration_md_points_at_it():
    """The Rust side is a crate a maintainer (or a box with cargo) can `cargo check`: Cargo.toml, build.rs (the link
    line, like the reference's build.rs:1-6), src/lib.rs, src/gpu.rs -- INTEGRATION.md refers to the file instead of
    embedding a second copy."""
    crate = os.path.join(ROOT, "rust", "granne-hip")
    for f in ("Cargo.toml", "build.rs", os.path.join("src", "lib.rs"), os.path.join("src", "gpu.rs")):
        assert os.path.getsize(os.path.join(crate, f)) > 0, f
    assert "cargo:rustc-link-lib=dylib=granne_hip" in open(os.path.join(crate, "build.rs")).read()
    assert re.search(r'granne\s*=\s*\{[^}]*path', open(os.path.join(crate, "Cargo.toml")).read())
    lib_rs = open(os.path.join(crate, "src", "lib.rs")).read()
    assert "pub mod gpu;" in lib_rs
    gpu = open(RUST_GPU_RS).read()
    for name in re.findall(r"use crate::\{([^}]*)\}", gpu)[0].split(","):
        assert re.search(r"pub use granne::\{[^}]*\b%s\b" % name.strip(), lib_rs), name  # what gpu.rs reaches through crate::
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert "rust/granne-hip/src/gpu.rs" in doc
    assert 'extern "C" {' not in doc  # no second copy of the binding to drift
    # the reference's Index trait (src/index/mod.rs:54-71) on the GPU type, method for method
    i = gpu.index("impl<E: GpuElements> Index for GpuGranne<E> {")
    body = gpu[i:gpu.index("\n}\n", i)]
    for m in ("fn len(", "fn num_layers(", "fn layer_len(", "fn get_neighbors(", "fn write_index<B: std::io::Write + std::io::Seek>("):
        assert m in body, m
    assert "granne_hip_index_encode(" in body and "granne_hip_bytes_free(" in body


def test_rust_binding_declares_the_whole_header():
    """rust/granne-hip/src/gpu.rs: every entry point of include/granne_hip.h is declared, with exactly the
    types the header has (tools/gen_rust_sys.py derives the Rust declaration from the C prototype)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_rust_sys", os.path.join(ROOT, "tools", "gen_rust_sys.py"))
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    doc = re.sub(r"\s+", "", open(RUST_GPU_RS).read())
    names = []
    for name, ret, params in g.protos():
        names.append(name)
        assert re.sub(r"\s+", "", g.decl(name, ret, params)) in doc, name
    assert sorted(names) == _declared()
    assert not set(re.findall(r"fn(granne_hip_[a-z0-9_]+)\(", doc)) - set(names)  # nothing bound that the header lacks


def test_abi_version(lib):
    assert lib.granne_hip_abi_version() == 3


def test_library_contains_gfx950_code_object():
    blob = open(build.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    assert b"search_kernel" in blob


def _has_gpu(lib):
    n = C.c_int(0)
    lib.granne_hip_device_count(C.byref(n))
    return n.value > 0


def test_no_silent_cpu_fallback(lib):
    """Without a GPU every entry point that would compute must fail with a device error."""
    if _has_gpu(lib):
        pytest.skip("a GPU is present; the failure path is for GPU-less hosts")
    el = np.zeros((4, 8), np.float32)
    h = C.c_void_p()
    rc = lib.granne_hip_index_create(C.byref(h), el.ctypes.data_as(C.c_void_p), 4, 8, 0, 0, None, None, None, 0)
    assert rc in (_lib.ERR_NO_DEVICE, _lib.ERR_HIP)
    assert not h.value
    assert lib.granne_hip_last_error()
    rc = lib.granne_hip_normalize_f32(el.ctypes.data_as(C.c_void_p), 4, 8, 0)
    assert rc in (_lib.ERR_NO_DEVICE, _lib.ERR_HIP)


def test_argument_validation_needs_no_device(lib):
    h = C.c_void_p()
    el = np.zeros((4, 8), np.float32)
    p = el.ctypes.data_as(C.c_void_p)
    assert lib.granne_hip_index_create(None, p, 4, 8, 0, 0, None, None, None, 0) == _lib.ERR_INVALID
    assert lib.granne_hip_index_create(C.byref(h), p, 4, 8, 7, 0, None, None, None, 0) == _lib.ERR_INVALID  # dtype
    assert lib.granne_hip_index_create(C.byref(h), p, 4, 0, 0, 0, None, None, None, 0) == _lib.ERR_INVALID  # dim
    lens = (C.c_uint64 * 2)(4, 2)  # not prefix-nested
    assert lib.granne_hip_index_create(C.byref(h), p, 4, 8, 0, 2, lens, None, None, 0) == _lib.ERR_INVALID
    assert b"prefix" in lib.granne_hip_last_error()
    assert lib.granne_hip_search_batch(None, p, 1, 10, 1, p, p, p, None) == _lib.ERR_INVALID
    # entries added for the operators around the path: null handles / buffers are rejected before any device call
    assert lib.granne_hip_index_reorder(None, None) == _lib.ERR_INVALID
    assert lib.granne_hip_index_reorder_by_keys(None, p, None) == _lib.ERR_INVALID
    assert lib.granne_hip_dists_device(None, p, 1, p, 1, p, None, None) == _lib.ERR_INVALID
    assert lib.granne_hip_dist_pairs_device(None, p, p, p, 1, p, None) == _lib.ERR_INVALID
    assert lib.granne_hip_search_batch_device_timed(None, p, 1, 10, 1, p, p, p, None, None, None, None, None) == _lib.ERR_INVALID
    assert lib.granne_hip_event_create(None) == _lib.ERR_INVALID
    assert lib.granne_hip_event_elapsed_ms(None, None, None) == _lib.ERR_INVALID
    lib.granne_hip_event_destroy(None)  # a no-op
    # round 4: several batches per launch; the partitioned handle's device-pointer and option entries
    assert lib.granne_hip_search_batches_device(None, 1, p, 1, 10, 1, p, p, p, None, None, None) == _lib.ERR_INVALID
    assert lib.granne_hip_search_begin_device(None, p, 1, 10, 1, p, p, p, None, None, None, None) == _lib.ERR_INVALID
    assert lib.granne_hip_search_end_device(None, 0, None) == _lib.ERR_INVALID
    assert lib.granne_hip_sharded_search_batch_device(None, p, 1, 10, 1, p, p, p, None, None) == _lib.ERR_INVALID
    assert lib.granne_hip_sharded_begin_device(None, p, 1, 10, 1, p, p, p, None, None, None) == _lib.ERR_INVALID
    assert lib.granne_hip_sharded_end_device(None, 0, None) == _lib.ERR_INVALID
    assert lib.granne_hip_sharded_search_batches(None, p, 1, 1, 10, 1, p, p, p) == _lib.ERR_INVALID
    assert lib.granne_hip_sharded_set_option(None, 1, 2) == _lib.ERR_INVALID
    assert lib.granne_hip_sharded_get_option(None, 1, None) == _lib.ERR_INVALID
    assert lib.granne_hip_sharded_device(None) == -1
    assert lib.granne_hip_sharded_build(None, None, p, 4, 8, 0, 2, None, 1) == _lib.ERR_INVALID
    assert not lib.granne_hip_sharded_shard(None, 0) and lib.granne_hip_sharded_shard_offset(None, 0) == 0


def test_rust_wrappers_call_the_entry_points_they_claim():
    """rust/granne-hip/src/gpu.rs: the safe wrappers of the device-resident entries (round 5) call exactly the prototype
    each one names, with the argument count the header declares -- a Rust host reaches search_batches_device / begin / end
    (and the partitioned equivalents) without writing `unsafe` itself."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_rust_sys", os.path.join(ROOT, "tools", "gen_rust_sys.py"))
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    arity = {name: len(params) for name, _ret, params in g.protos()}
    doc = open(RUST_GPU_RS).read()

    def body_of(owner, fn):
        i = doc.index("impl<E: GpuElements> %s<E> {" % owner) if owner else 0
        j = doc.index("pub fn %s" % fn, i)
        depth, k = 0, doc.index("{", j)
        for k in range(k, len(doc)):
            depth += doc[k] == "{"
            depth -= doc[k] == "}"
            if depth == 0:
                break
        return doc[j:k + 1]

    def call_args(body, name):
        i = body.index(name + "(") + len(name) + 1
        depth, parts, cur = 0, [], ""
        for ch in body[i:]:
            if ch in "([{":
                depth += 1
            if ch in ")]}":
                if depth == 0:
                    break
                depth -= 1
            if ch == "," and depth == 0:
                parts.append(cur)
                cur = ""
            else:
                cur += ch
        parts.append(cur)
        return [p for p in (x.strip() for x in parts) if p]

    claims = [("GpuGranne", "search_batch_device", "granne_hip_search_batch_device"),
              ("GpuGranne", "search_batches_device", "granne_hip_search_batches_device"),
              ("GpuGranne", "begin<'a>", "granne_hip_search_begin_device"),
              ("GpuGranne", "set_search_depth", "granne_hip_index_set_option"),
              ("GpuGranne", "set_inline_tails", "granne_hip_index_set_option"),
              ("GpuShardedGranne", "search_batch_device", "granne_hip_sharded_search_batch_device"),
              ("GpuShardedGranne", "begin<'a>", "granne_hip_sharded_begin_device")]
    for owner, fn, entry in claims:
        body = body_of(owner, fn)
        assert "unsafe { %s(" % entry in body, (owner, fn)
        assert len(call_args(body, entry)) == arity[entry], (owner, fn, call_args(body, entry), arity[entry])
    for ty, entry in (("InFlight", "granne_hip_search_end_device"), ("ShardedInFlight", "granne_hip_sharded_end_device")):
        i = doc.index("impl<'a, E: GpuElements> %s<'a, E> {" % ty)
        body = doc[i:doc.index("impl<'a, E: GpuElements> Drop for %s" % ty, i)]
        assert len(call_args(body, entry)) == arity[entry], ty
    # the public signatures of these wrappers hold no raw pointer and are not `unsafe fn`
    for owner, fn, _ in claims:
        sig = body_of(owner, fn).split("{", 1)[0]
        assert "*mut" not in sig and "*const" not in sig and "unsafe" not in sig, sig

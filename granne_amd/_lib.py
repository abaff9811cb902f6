"""ctypes loader for libgranne_hip.so (the C ABI declared in include/granne_hip.h).

There is no fallback: if the library is missing or the GPU is absent, calls raise."""
import ctypes as C
import os

from . import build as _build

F32, I8 = 0, 1
UNUSED = 0xFFFFFFFF
BUILD_ALL = 0xFFFFFFFFFFFFFFFF  # GRANNE_HIP_BUILD_ALL: Builder::build()

OK = 0
ERR_INVALID, ERR_HIP, ERR_NO_DEVICE, ERR_OVERFLOW, ERR_IO = -1, -2, -3, -4, -5

OPT_VISITED_SLOTS, OPT_FORCE_SLOW, OPT_SLOW_SLOTS, OPT_SLOW_BLOCKS, OPT_OVERFLOW_SLOTS = 1, 2, 3, 4, 5
OPT_VISITED16, OPT_VISITED16_LG, OPT_LAST_WALKER, OPT_SEARCH_DEPTH, OPT_INLINE_TAILS, OPT_SEEN_MIN = 6, 7, 8, 9, 10, 11
WALKER_NONE, WALKER_REGISTER, WALKER_REGISTER_WIDE, WALKER_GENERAL, WALKER_EXACT = 0, 1, 2, 3, 4
SEARCH_DEPTH = 3  # GRANNE_HIP_SEARCH_DEPTH (the default of OPT_SEARCH_DEPTH)
SEARCH_DEPTH_MAX = 16  # GRANNE_HIP_SEARCH_DEPTH_MAX
SHARDED_OPT_DEPTH, SHARDED_OPT_EXCHANGE = 1, 2
SHARDED_EXCHANGE_PEER, SHARDED_EXCHANGE_RCCL = 0, 1


class GranneHipError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("granne_hip error %d: %s" % (code, message))
        self.code = code


class BuildConfig(C.Structure):
    """granne_hip_build_config (include/granne_hip.h) = BuildConfig (src/index/mod.rs:198-231)."""
    _fields_ = [
        ("layer_multiplier", C.c_float),
        ("expected_num_elements", C.c_uint64),
        ("num_neighbors", C.c_uint32),
        ("max_search", C.c_uint32),
        ("reinsert_elements", C.c_int),
        ("show_progress", C.c_int),
        ("batch_max", C.c_uint32),
        ("batch_div", C.c_uint32),
    ]


_lib = None

vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int

# name -> (restype, argtypes): every symbol include/granne_hip.h declares
SIGNATURES = {
    "granne_hip_last_error": (C.c_char_p, []),
    "granne_hip_abi_version": (i32, []),
    "granne_hip_device_count": (i32, [C.POINTER(i32)]),
    "granne_hip_index_create": (i32, [C.POINTER(vp), vp, u64, u32, i32, u32, vp, vp, vp, i32]),
    "granne_hip_index_create_csr": (i32, [C.POINTER(vp), vp, u64, u32, i32, u32, vp, vp, vp, i32]),
    "granne_hip_index_create_device": (i32, [C.POINTER(vp), vp, u64, u32, i32, u32, vp, vp, vp, i32, vp]),
    "granne_hip_index_destroy": (None, [vp]),
    "granne_hip_index_len": (u64, [vp]),
    "granne_hip_index_num_layers": (u32, [vp]),
    "granne_hip_index_layer_len": (u64, [vp, u32]),
    "granne_hip_index_dim": (u32, [vp]),
    "granne_hip_index_dtype": (i32, [vp]),
    "granne_hip_index_device": (i32, [vp]),
    "granne_hip_index_hbm_bytes": (u64, [vp]),
    "granne_hip_index_get_neighbors": (i32, [vp, u64, u32, vp, u32, C.POINTER(u32)]),
    "granne_hip_index_get_element": (i32, [vp, u64, vp]),
    "granne_hip_search_batch": (i32, [vp, vp, u32, u32, u32, vp, vp, vp, vp]),
    "granne_hip_search_batch_device": (i32, [vp, vp, u32, u32, u32, vp, vp, vp, vp, vp, vp]),
    "granne_hip_search_batches_device": (i32, [vp, u32, vp, u32, u32, u32, vp, vp, vp, vp, vp, vp]),
    "granne_hip_search_begin_device": (i32, [vp, vp, u32, u32, u32, vp, vp, vp, vp, vp, vp, C.POINTER(u64)]),
    "granne_hip_search_end_device": (i32, [vp, u64, vp]),
    "granne_hip_device_malloc": (i32, [C.POINTER(vp), u64, i32]),
    "granne_hip_device_free": (i32, [vp, i32]),
    "granne_hip_copy_to_device": (i32, [vp, vp, u64, i32, vp]),
    "granne_hip_copy_to_host": (i32, [vp, vp, u64, i32, vp]),
    "granne_hip_stream_create": (i32, [C.POINTER(vp), i32]),
    "granne_hip_stream_destroy": (i32, [vp, i32]),
    "granne_hip_stream_synchronize": (i32, [vp, i32]),
    "granne_hip_event_create": (i32, [vp]),
    "granne_hip_event_destroy": (None, [vp]),
    "granne_hip_event_elapsed_ms": (i32, [vp, vp, vp]),
    "granne_hip_search_batch_device_timed": (i32, [vp, vp, u32, u32, u32, vp, vp, vp, vp, vp, vp, vp, vp]),
    "granne_hip_search": (i32, [vp, vp, u32, u32, vp, vp, C.POINTER(u32)]),
    "granne_hip_normalize_f32_device": (i32, [vp, u64, u32, i32, vp]),
    "granne_hip_quantize_f32_device": (i32, [vp, vp, u64, u32, i32, vp]),
    "granne_hip_dist_pairs_device": (i32, [vp, vp, vp, vp, u64, vp, vp]),
    "granne_hip_dists_device": (i32, [vp, vp, u32, vp, u32, vp, vp, vp]),
    "granne_hip_index_reorder": (i32, [vp, vp]),
    "granne_hip_index_reorder_by_keys": (i32, [vp, vp, vp]),
    "granne_hip_normalize_f32": (i32, [vp, u64, u32, i32]),
    "granne_hip_quantize_f32": (i32, [vp, vp, u64, u32, i32]),
    "granne_hip_dist_pairs": (i32, [vp, vp, u32, vp, vp, u64, vp]),
    "granne_hip_synth_rows_device": (i32, [vp, u64, u64, u64, u32, i32, vp]),
    "granne_hip_index_load": (i32, [C.POINTER(vp), vp, u64, vp, u64, i32, i32]),
    "granne_hip_index_load_files": (i32, [C.POINTER(vp), C.c_char_p, C.c_char_p, i32, i32]),
    "granne_hip_write_index_file": (i32, [C.c_char_p, u32, vp, vp, vp]),
    "granne_hip_write_elements_file": (i32, [C.c_char_p, vp, u64, u32, i32]),
    "granne_hip_index_save": (i32, [vp, C.c_char_p, C.c_char_p]),
    "granne_hip_index_encode": (i32, [vp, C.POINTER(vp), C.POINTER(u64)]),
    "granne_hip_bytes_free": (None, [vp]),
    "granne_hip_index_file_info": (i32, [vp, u64, C.POINTER(u32), vp, vp, u32]),
    "granne_hip_index_file_decode_layer": (i32, [vp, u64, u32, vp, vp]),
    "granne_hip_brute_force_device": (i32, [vp, vp, u32, u32, vp, vp, vp, vp]),
    "granne_hip_brute_force": (i32, [vp, vp, u32, u32, vp, vp, vp]),
    "granne_hip_merge_topk_device": (i32, [vp, vp, vp, vp, u32, u32, u32, vp, vp, vp, i32, vp]),
    "granne_hip_packed_topk_bytes": (u64, [u32, u32]),
    "granne_hip_search_batch_packed_device": (i32, [vp, vp, u32, u32, u32, vp, vp, vp]),
    "granne_hip_merge_topk_packed_device": (i32, [vp, vp, u32, u32, u32, vp, vp, vp, i32, vp]),
    "granne_hip_merge_topk_packed_strided_device": (i32, [vp, u64, vp, u32, u32, u32, vp, vp, vp, i32, vp]),
    "granne_hip_sharded_create": (i32, [C.POINTER(vp), vp, vp, u32]),
    "granne_hip_sharded_create_grouped": (i32, [C.POINTER(vp), vp, vp, u32, vp]),
    "granne_hip_sharded_destroy": (None, [vp]),
    "granne_hip_sharded_num_shards": (u32, [vp]),
    "granne_hip_sharded_len": (u64, [vp]),
    "granne_hip_sharded_device": (i32, [vp]),
    "granne_hip_sharded_shard": (vp, [vp, u32]),
    "granne_hip_sharded_shard_offset": (u64, [vp, u32]),
    "granne_hip_sharded_build": (i32, [C.POINTER(vp), vp, vp, u64, u32, i32, u32, vp, u32]),
    "granne_hip_sharded_set_option": (i32, [vp, i32, u64]),
    "granne_hip_sharded_get_option": (i32, [vp, i32, C.POINTER(u64)]),
    "granne_hip_sharded_search_batch_device": (i32, [vp, vp, u32, u32, u32, vp, vp, vp, vp, vp]),
    "granne_hip_sharded_begin_device": (i32, [vp, vp, u32, u32, u32, vp, vp, vp, vp, vp, C.POINTER(u64)]),
    "granne_hip_sharded_end_device": (i32, [vp, u64, vp]),
    "granne_hip_sharded_search_batches": (i32, [vp, vp, u32, u32, u32, u32, vp, vp, vp]),
    "granne_hip_sharded_search_batch": (i32, [vp, vp, u32, u32, u32, vp, vp, vp]),
    "granne_hip_sharded_search": (i32, [vp, vp, u32, u32, vp, vp, C.POINTER(u32)]),
    "granne_hip_build_config_default": (None, [vp]),
    "granne_hip_builder_create": (i32, [C.POINTER(vp), vp, vp, u64, u32, i32, i32]),
    "granne_hip_builder_create_device": (i32, [C.POINTER(vp), vp, vp, u64, u32, i32, i32, vp]),
    "granne_hip_builder_append": (i32, [vp, vp, u64]),
    "granne_hip_builder_load_index": (i32, [vp, vp, u64]),
    "granne_hip_builder_build": (i32, [vp, u64]),
    "granne_hip_builder_len": (u64, [vp]),
    "granne_hip_builder_num_elements": (u64, [vp]),
    "granne_hip_builder_num_layers": (u32, [vp]),
    "granne_hip_builder_layer_len": (u64, [vp, u32]),
    "granne_hip_builder_get_layer": (i32, [vp, u32, vp]),
    "granne_hip_builder_get_index": (i32, [vp, C.POINTER(vp)]),
    "granne_hip_builder_destroy": (None, [vp]),
    "granne_hip_index_set_option": (i32, [vp, i32, u64]),
    "granne_hip_index_get_option": (i32, [vp, i32, C.POINTER(u64)]),
    "granne_hip_index_last_slow_count": (u64, [vp]),
}


def lib():
    """The loaded library. Raises if it has not been built (python -m granne_amd.build)."""
    global _lib
    if _lib is None:
        path = os.environ.get("GRANNE_HIP_LIB") or _build.LIB_PATH  # GRANNE_HIP_LIB: kernel experiments (tools/sweep.py)
        if not os.path.exists(path):
            raise GranneHipError(ERR_NO_DEVICE, "libgranne_hip.so is not built: run `python -m granne_amd.build` "
                                 "(no CPU fallback exists)")
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64. If torch is
        # importable, load it first so that libgranne_hip.so binds to the same runtime (two
        # runtimes in one process leave the second without devices).
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        L = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != OK:
        raise GranneHipError(rc, lib().granne_hip_last_error().decode("utf-8", "replace"))
    return rc

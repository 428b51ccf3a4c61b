"""ctypes binding of libserenade_hip.so (include/serenade_hip.h).  Thin: every call maps 1:1 onto a C
entry point and raises SerenadeError with srn_last_error() on a negative return code."""
import ctypes as C
import os

import numpy as np

from . import build as _build

SRN_OK, SRN_EINVAL, SRN_ENOMEM, SRN_EHIP, SRN_ERANGE, SRN_EIO, SRN_ENODEV, SRN_ESTATE, SRN_ETIMEOUT = 0, -1, -2, -3, -4, -5, -6, -7, -8
ATTR_ADULT, ATTR_FOR_SALE, ATTR_NONE = 1, 2, 0xFF
FLAG_BUSINESS_LOGIC = 1
MAX_HOW_MANY, MAX_SESSION_LEN, MAX_K = 512, 255, 8192

_CODES = {SRN_EINVAL: "SRN_EINVAL", SRN_ENOMEM: "SRN_ENOMEM", SRN_EHIP: "SRN_EHIP", SRN_ERANGE: "SRN_ERANGE",
          SRN_EIO: "SRN_EIO", SRN_ENODEV: "SRN_ENODEV", SRN_ESTATE: "SRN_ESTATE", SRN_ETIMEOUT: "SRN_ETIMEOUT"}


class SerenadeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s: %s" % (_CODES.get(code, code), msg))
        self.code = code


class SessionsView(C.Structure):
    _fields_ = [("sess_off", C.c_void_p), ("items", C.c_void_p), ("max_ts", C.c_void_p), ("n_sessions", C.c_size_t)]


class IndexInfo(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_items", "n_sessions_total", "n_sessions_kept", "nnz_rows", "nnz_postings",
                                          "m_index", "max_session_len", "max_row_len", "device_bytes")] + \
               [("device", C.c_int32), ("offsets_64bit", C.c_int32), ("idf_weighting", C.c_double), ("incomplete_items", C.c_uint64)]


class ShardGroupStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_shards", "batches", "queries", "bytes_head", "bytes_counts", "bytes_lists", "bytes_results", "bytes_lists_max_rank")] + \
               [("transport", C.c_uint32), ("overlapped", C.c_uint32)] + [(n, C.c_uint64) for n in ("stage_batches", "bytes_stage_candidates", "bytes_stage_minpos", "neighbour_batches", "bytes_neighbours")]


# srn_shard_comm_t: the application's transport for a shard group (three collectives on device buffers)
ALL_REDUCE_MAX_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p)
ALL_GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p)
ALL_GATHER_V_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_void_p)


class ShardComm(C.Structure):
    _fields_ = [("user", C.c_void_p), ("all_reduce_max_i32", ALL_REDUCE_MAX_FN), ("all_gather", ALL_GATHER_FN), ("all_gather_v", ALL_GATHER_V_FN),
                ("all_reduce_min_i32", ALL_REDUCE_MAX_FN)]


SHARD_GROUP_ID_BYTES = 256
FLAG_INPUTS_RESIDENT = 2


class Limits(C.Structure):
    _fields_ = [("max_how_many", C.c_uint32), ("max_session_len", C.c_uint32), ("max_k", C.c_uint32), ("reserved", C.c_uint32)]


# every symbol include/serenade_hip.h and include/serenade_hip_internal.h (the srn_debug_* aids) declare: (restype, argtypes)
_vp, _sz, _u64, _i = C.c_void_p, C.c_size_t, C.c_uint64, C.c_int
SYMBOLS = {
    "srn_sessions_from_tsv": (_i, [C.c_char_p, C.POINTER(_vp)]),
    "srn_sessions_view": (_i, [_vp, C.POINTER(SessionsView)]),
    "srn_sessions_length_quantile": (_i, [_vp, C.c_double, C.POINTER(_u64)]),
    "srn_sessions_free": (None, [_vp]),
    "srn_index_build": (_i, [C.POINTER(SessionsView), _sz, _sz, C.c_double, _i, C.POINTER(_vp)]),
    "srn_index_new_from_avro": (_i, [C.c_char_p, _i, C.POINTER(_vp)]),
    "srn_index_build_gpu": (_i, [C.POINTER(SessionsView), _sz, _sz, C.c_double, _i, C.POINTER(_vp)]),
    "srn_index_new_from_csv": (_i, [C.c_char_p, _sz, C.c_double, _sz, _i, C.POINTER(_vp)]),
    "srn_index_save": (_i, [_vp, C.c_char_p]),
    "srn_index_load": (_i, [C.c_char_p, _i, C.POINTER(_vp)]),
    "srn_index_set_attributes": (_i, [_vp, _vp, _vp, _sz]),
    "srn_index_info": (_i, [_vp, C.POINTER(IndexInfo)]),
    "srn_index_postings": (_i, [_vp, _u64, _vp, _sz, C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "srn_index_items_for_session": (_i, [_vp, C.c_uint32, _vp, _sz, C.POINTER(_sz)]),
    "srn_index_find_attributes": (_i, [_vp, _u64, C.POINTER(C.c_uint8)]),
    "srn_index_session_recency": (_i, [_vp, _vp, _sz]),
    "srn_index_serve_start": (_i, [_vp, _sz, _sz, _sz, C.c_int, C.c_uint, C.c_uint, C.c_uint]),
    "srn_index_serve_stop": (_i, [_vp]),
    "srn_index_serve_stats": (_i, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    "srn_find_neighbors": (_i, [_vp, _vp, _sz, _sz, _sz, _vp, _vp, C.POINTER(_sz)]),
    "srn_index_free": (None, [_vp]),
    "srn_predict": (_i, [_vp, _vp, _sz, _sz, _sz, _sz, _i, _vp, _vp, C.POINTER(_sz)]),
    "srn_predict_stats": (_i, [_vp, C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_u64)]),
    "srn_predict_batch": (_i, [_vp, _vp, _vp, _sz, _sz, _sz, _sz, C.c_uint, _vp, _vp, _vp]),
    "srn_batcher_create": (_i, [_vp, _sz, C.c_uint, _sz, _sz, _sz, _i, C.POINTER(_vp)]),
    "srn_batcher_predict": (_i, [_vp, _vp, _sz, _vp, _vp, C.POINTER(_sz)]),
    "srn_batcher_stats": (_i, [_vp, C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_u64)]),
    "srn_batcher_free": (None, [_vp]),
    "srn_batcher_how_many": (_i, [_vp, C.POINTER(_sz)]),
    "srn_session_key": (_i, [C.c_char_p, _sz, C.POINTER(_u64), C.POINTER(_u64)]),
    "srn_session_store_create": (_i, [_u64, _u64, C.POINTER(_vp)]),
    "srn_session_store_free": (None, [_vp]),
    "srn_session_store_get": (_i, [_vp, _u64, _u64, _u64, _vp, _sz, C.POINTER(_sz)]),
    "srn_session_store_update": (_i, [_vp, _u64, _u64, _u64, _vp, _sz]),
    "srn_session_store_sweep": (_i, [_vp, _u64, C.POINTER(_u64)]),
    "srn_recommend": (_i, [_vp, _vp, C.c_char_p, _sz, _u64, _i, _sz, _u64, _vp, _vp, C.POINTER(_sz)]),
    "srn_predict_batch_device": (_i, [_vp, _vp, _vp, _sz, _sz, _sz, _sz, _sz, C.c_uint, _vp, _vp, _vp, _vp]),
    "srn_predict_batch_debug": (_i, [_vp, _vp, _vp, _sz, _sz, _sz, _sz, C.c_uint, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "srn_index_reserve": (_i, [_vp, _sz, _sz, _sz, _sz, _sz, C.c_uint, _vp]),
    "srn_last_kernel_ms": (_i, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint32)]),
    "srn_index_build_shard": (_i, [C.POINTER(SessionsView), _sz, _sz, C.c_double, C.c_uint32, C.c_uint32, _i, C.POINTER(_vp)]),
    "srn_index_shard": (_i, [_vp, C.c_uint32, C.c_uint32, _i, C.POINTER(_vp)]),
    "srn_index_build_shard_gpu": (_i, [C.POINTER(SessionsView), _sz, _sz, C.c_double, C.c_uint32, C.c_uint32, _i, C.POINTER(_vp)]),
    "srn_index_load_shard": (_i, [C.c_char_p, C.c_uint32, C.c_uint32, _i, C.POINTER(_vp)]),
    "srn_shard_group_unique_id": (_i, [_vp, _sz]),
    "srn_shard_group_create": (_i, [_vp, _vp, _i, _i, C.POINTER(_vp)]),
    "srn_shard_group_create_with_comm": (_i, [_vp, _i, _i, _vp, C.POINTER(_vp)]),
    "srn_shard_group_create_local": (_i, [C.POINTER(_vp), _i, C.POINTER(_vp)]),
    "srn_shard_group_predict_batch": (_i, [_vp, _vp, _vp, _sz, _sz, _sz, _sz, _sz, C.c_uint, _vp, _vp, _vp, _vp]),
    "srn_shard_group_stats": (_i, [_vp, _vp]),
    "srn_shard_group_set_overlap": (_i, [_vp, _i]),
    "srn_shard_group_wait": (_i, [_vp, _u64]),
    "srn_shard_group_set_postings": (_i, [_vp, _vp]),
    "srn_debug_shard_group_times": (_i, [_vp, _vp]),
    "srn_index_postings_view": (_i, [_vp, _i, C.POINTER(_vp)]),
    "srn_shard_group_free": (None, [_vp]),
    "srn_kernel_timing": (_i, [_vp, C.c_int]),
    "srn_kernel_times": (_i, [_vp, C.c_uint32, _vp, _vp, C.POINTER(C.c_uint32)]),
    "srn_kernel_times_detail": (_i, [_vp, C.c_uint32, _vp, _vp, _vp, _vp, C.POINTER(C.c_uint32)]),
    "srn_debug_phase_cycles": (_i, [_vp, _i, _vp]),
    "srn_debug_reload_knobs": (None, []),
    "srn_debug_last_mid_count": (_i, [_vp, C.POINTER(C.c_uint32)]),
    "srn_debug_serve_stamps": (_i, [_vp, _vp]),
    "srn_debug_shard_nb_positions_stride": (C.c_uint32, [_sz, _sz]),
    "srn_debug_last_big_count": (_i, [_vp, C.POINTER(C.c_uint32)]),
    "srn_debug_sback_launches": (_i, [_vp, C.POINTER(_u64)]),
    "srn_last_path_counts": (_i, [_vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "srn_device_count": (_i, [C.POINTER(_i)]),
    "srn_limits": (None, [C.POINTER(Limits)]),
    "srn_last_error": (C.c_char_p, []),
    "srn_version": (C.c_char_p, []),
}

_lib = None


def _preload_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's).  Two HIP runtimes
    in one process each open the KFD and the second one sees no GPU, so when torch is installed we load
    ITS runtime first; libserenade_hip.so's DT_NEEDED libamdhip64.so.7 then binds to it and torch
    (device memory, streams, torch.distributed/RCCL) shares one runtime with our kernels.
    SRN_HIP_RUNTIME=system skips this and uses /opt/rocm's runtime (no torch in the process then)."""
    if os.environ.get("SRN_HIP_RUNTIME", "") == "system":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec and spec.submodule_search_locations:
            p = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
            if os.path.exists(p):
                C.CDLL(p, mode=C.RTLD_GLOBAL)
            # and ONE RCCL: the shard group (srn_group.hip) looks for a librccl already in the process before it loads /opt/rocm's
            r = os.path.join(list(spec.submodule_search_locations)[0], "lib", "librccl.so")
            if os.path.exists(r) and os.environ.get("SRN_RCCL_LIB", "") == "":
                os.environ["SRN_RCCL_LIB"] = r
    except Exception:
        pass


def lib():
    """Load the in-tree HIP library.  Fails loudly if it is missing: there is no fallback path."""
    global _lib
    if _lib is None:
        path = os.environ.get("SRN_LIB_PATH") or _build.LIB          # (SRN_LIB_PATH: a kernel variant built by build.build_variant, experiments only)
        if not os.path.exists(path):
            raise ImportError("libserenade_hip.so is not built; run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(there is no CPU fallback for the predict path)")
        _preload_hip_runtime()
        L = C.CDLL(path)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def reload_knobs():
    """Re-read the SRN_* test knobs from the environment (the library reads them once, at first use)."""
    lib().srn_debug_reload_knobs()


def check(rc):
    if rc != 0:
        raise SerenadeError(rc, lib().srn_last_error().decode(errors="replace"))


def ptr(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def as_u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def as_u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def device_count():
    n = C.c_int()
    check(lib().srn_device_count(C.byref(n)))
    return n.value

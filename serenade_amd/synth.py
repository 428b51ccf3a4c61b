"""Synthetic workloads for BASELINE.json configs 2-5 -- wrapper of csrc/srn_synth.cpp."""
import ctypes as C
import os

import numpy as np

from . import build as _build

SEED = 0x5E4E4ADE
T0 = 1_500_000_000

# name -> (interactions, items, k, m, idf_weighting); m_index = m; last_items = 4; how_many = 21
CONFIGS = {
    "tiny": (200_000, 20_000, 100, 500, 1.0),          # tests / smoke
    "cfg2": (2_000_000, 100_000, 500, 1000, 1.0),      # "Synthetic 2M interactions / 100K items, k=500 m=1000"
    "cfg3": (60_000_000, 1_760_000, 1500, 2500, 2.0),  # "Synthetic 60M / 1.76M items, k=1500 m=2500 idf=2"
    "cfg4": (582_000_000, 6_500_000, 1500, 2500, 1.0),
    "cfg5": (2_300_000_000, 20_000_000, 1500, 2500, 1.0),
    "cfg5_8th": (287_500_000, 2_500_000, 1500, 2500, 1.0),   # config 5's shape at 1/8 scale: what one GPU of the 8 would see of the sessions if they were split too
}
ZIPF_ALPHA = 1.05
LAST_ITEMS = 4
HOW_MANY = 21

_lib = None


def lib():
    global _lib
    if _lib is None:
        try:
            _build.build_synth()          # no-op unless the source is newer than the .so
        except Exception:
            if not os.path.exists(_build.SYNTH_LIB):
                raise
        L = C.CDLL(_build.SYNTH_LIB)
        vp, u64 = C.c_void_p, C.c_uint64
        L.srn_synth_training.restype = vp
        L.srn_synth_training.argtypes = [u64, u64, u64, C.c_double, C.c_uint32, C.c_int]
        L.srn_synth_queries.restype = vp
        L.srn_synth_queries.argtypes = [u64, u64, u64, C.c_double, C.c_uint32, C.c_int]
        for n in ("srn_synth_n_sessions", "srn_synth_nnz", "srn_synth_n_queries", "srn_synth_q_nnz"):
            getattr(L, n).restype = u64
            getattr(L, n).argtypes = [vp]
        L.srn_synth_copy_training.argtypes = [vp, vp, vp, vp]
        L.srn_synth_copy_queries.argtypes = [vp, vp, vp]
        L.srn_synth_copy_next.argtypes = [vp, vp]
        L.srn_synth_free.argtypes = [vp]
        L.srn_synth_producer.restype = vp
        L.srn_synth_producer.argtypes = [vp, vp, vp, u64, u64, u64, C.c_double, C.c_int]
        L.srn_synth_producer_n_items.restype = u64
        L.srn_synth_producer_n_items.argtypes = [vp]
        L.srn_synth_producer_nnz.restype = u64
        L.srn_synth_producer_nnz.argtypes = [vp]
        L.srn_synth_producer_copy.argtypes = [vp, vp, vp, vp, vp]
        L.srn_synth_producer_free.argtypes = [vp]
        L.srn_synth_write_avro.restype = C.c_int
        L.srn_synth_write_avro.argtypes = [C.c_char_p, vp, vp, vp, vp, u64, C.c_int]
        _lib = L
    return _lib


def _threads():
    return max(1, min(32, os.cpu_count() or 1))


def training_sessions(n_interactions, n_items, seed=SEED, alpha=ZIPF_ALPHA, t0=T0):
    """-> (sess_off u64[n+1], items u64[nnz] ascending+dedup per session, max_ts u32[n] unique)."""
    L = lib()
    h = L.srn_synth_training(seed, int(n_interactions), int(n_items), float(alpha), int(t0), _threads())
    n, nnz = L.srn_synth_n_sessions(h), L.srn_synth_nnz(h)
    off, items, ts = np.empty(n + 1, np.uint64), np.empty(nnz, np.uint64), np.empty(n, np.uint32)
    L.srn_synth_copy_training(h, off.ctypes.data, items.ctypes.data, ts.ctypes.data)
    L.srn_synth_free(h)
    return off, items, ts


def queries(n_sessions, n_items, seed=SEED, alpha=ZIPF_ALPHA, max_items=LAST_ITEMS, with_next=False):
    """Evaluator-style query stream -> (items_flat u64, q_off u32[nq+1]); with_next: also next u64[nq], the held-out item that followed each prefix
    (src/bin/evaluator.rs:75 -- the item Mrr / HitRate score a recommendation list against)."""
    L = lib()
    h = L.srn_synth_queries(seed, int(n_sessions), int(n_items), float(alpha), int(max_items), _threads())
    nq, nnz = L.srn_synth_n_queries(h), L.srn_synth_q_nnz(h)
    items, off = np.empty(nnz, np.uint64), np.empty(nq + 1, np.uint32)
    L.srn_synth_copy_queries(h, items.ctypes.data, off.ctypes.data)
    nxt = None
    if with_next:
        nxt = np.empty(nq, np.uint64)
        L.srn_synth_copy_next(h, nxt.ctypes.data)
    L.srn_synth_free(h)
    return (items, off, nxt) if with_next else (items, off)


TIE_MODES = {"ours": 0, "reverse": 1, "mixed": 2, "per-item": 3}


def tie_timestamps(ts, per_second, t0=T0):
    """The unique synthetic timestamps at a coarser clock: ~per_second sessions share each value (production data has second resolution)."""
    return (t0 + (ts.astype(np.int64) - t0) // int(per_second)).astype(np.uint32)


def avro_index(base, off, items, ts, m_index, max_len, idf_weighting, tie_mode="reverse", files=4):
    """A stand-in for the reference's offline index producer (srn_synth.cpp): writes <base>/itemindex/*.avro + <base>/sessionindex/*.avro and returns what it
    wrote as arrays (item_ids ascending, list_off, list_sessions, idf) -- the lists AS GIVEN, for a checker."""
    L = lib()
    off, items, ts = np.ascontiguousarray(off, np.uint64), np.ascontiguousarray(items, np.uint64), np.ascontiguousarray(ts, np.uint32)
    n = len(ts)
    h = L.srn_synth_producer(off.ctypes.data, items.ctypes.data, ts.ctypes.data, n, int(m_index), int(max_len), float(idf_weighting), TIE_MODES[tie_mode])
    ni, nnz = L.srn_synth_producer_n_items(h), L.srn_synth_producer_nnz(h)
    ids, loff, lsess, idf = np.empty(ni, np.uint64), np.empty(ni + 1, np.uint64), np.empty(max(nnz, 1), np.uint32), np.empty(ni, np.float64)
    L.srn_synth_producer_copy(h, ids.ctypes.data, loff.ctypes.data, lsess.ctypes.data, idf.ctypes.data)
    os.makedirs(os.path.join(base, "itemindex"), exist_ok=True)
    os.makedirs(os.path.join(base, "sessionindex"), exist_ok=True)
    rc = L.srn_synth_write_avro(str(base).encode(), h, off.ctypes.data, items.ctypes.data, ts.ctypes.data, n, int(files))
    L.srn_synth_producer_free(h)
    if rc:
        raise IOError("could not write the Avro index under %s" % base)
    return ids, loff, lsess[:nnz], idf

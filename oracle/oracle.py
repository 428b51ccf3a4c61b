"""ctypes wrapper around oracle/libvmis_oracle.so  --  TEST INFRASTRUCTURE ONLY.

Only tests/, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import this
module; the product package ``serenade_amd`` never does.  See the header of ``vmis_oracle.cpp``.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libvmis_oracle.so")
_lib = None

u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")


def build(force=False):
    """Compile the oracle with g++ (no GPU, no reference sources involved)."""
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "vmis_oracle.cpp")):
        subprocess.check_call(["make", "-C", _HERE, "-s", "clean", "all"])


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    sz, vp = C.c_size_t, C.c_void_p
    L.orc_index_build.restype = vp
    L.orc_index_build.argtypes = [u64p, u64p, u32p, sz, sz, sz, C.c_double, C.c_int]
    L.orc_index_build_restricted.restype = vp
    L.orc_index_build_restricted.argtypes = [u64p, u64p, u32p, sz, sz, sz, C.c_double, u64p, sz, C.c_int, sz]
    L.orc_index_from_parts.restype = vp
    L.orc_index_from_parts.argtypes = [u64p, u64p, u32p, f64p, u8p, sz, u64p, u64p, u32p, sz, vp]
    L.orc_index_free.argtypes = [vp]
    L.orc_sessions_read_tsv.restype = vp
    L.orc_sessions_read_tsv.argtypes = [C.c_char_p]
    L.orc_sessions_count.restype = sz
    L.orc_sessions_count.argtypes = [vp]
    L.orc_sessions_nnz.restype = sz
    L.orc_sessions_nnz.argtypes = [vp]
    L.orc_sessions_copy.argtypes = [vp, u64p, u64p, u32p, u64p]
    L.orc_sessions_free.argtypes = [vp]
    L.orc_index_set_attributes.argtypes = [vp, u64p, u8p, sz]
    L.orc_index_num_items.restype = sz
    L.orc_index_num_items.argtypes = [vp]
    L.orc_index_total_pairs.restype = sz
    L.orc_index_total_pairs.argtypes = [vp]
    L.orc_index_postings.restype = C.c_long
    L.orc_index_postings.argtypes = [vp, C.c_uint64, u32p, sz, C.POINTER(C.c_double)]
    L.orc_predict_literal.argtypes = [vp, u64p, sz, sz, sz, sz, C.c_int, u64p, f64p, C.POINTER(sz)]
    L.orc_find_neighbors_literal.argtypes = [vp, u64p, sz, sz, sz, u32p, f64p, C.POINTER(sz)]
    L.orc_predict_canonical.argtypes = [vp, u64p, sz, sz, sz, sz, C.c_int, u64p, f64p, C.POINTER(sz), vp]
    L.orc_neighbors_canonical.argtypes = [vp, u64p, sz, sz, sz, u32p, u32p, C.POINTER(sz), C.POINTER(sz)]
    L.orc_scores_canonical.restype = C.c_long
    L.orc_scores_canonical.argtypes = [vp, u64p, sz, sz, sz, u64p, f64p, i64p, sz]
    L.orc_predict_batch.restype = C.c_double
    L.orc_predict_batch.argtypes = [vp, C.c_int, u64p, u32p, sz, sz, sz, sz, C.c_int, C.c_int,
                                    vp, vp, vp, vp, vp]
    L.orc_kat_itemscore_heap.restype = sz
    L.orc_kat_itemscore_heap.argtypes = [u64p, f64p, sz, sz, u64p]
    L.orc_kat_itemscore_sorted.restype = sz
    L.orc_kat_itemscore_sorted.argtypes = [u64p, f64p, sz, u64p]
    L.orc_kat_sessiontime_heap.restype = sz
    L.orc_kat_sessiontime_heap.argtypes = [u32p, u32p, sz, sz, u32p]
    _lib = L
    return L


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def read_tsv(path):
    """read_from_file restated (src/vmisknn/vmis_index.rs:591-752) -> (sess_off, items, ts, session_ids)."""
    L = lib()
    h = L.orc_sessions_read_tsv(path.encode())
    if not h:
        raise IOError(path)
    n, nnz = L.orc_sessions_count(h), L.orc_sessions_nnz(h)
    off = np.zeros(n + 1, np.uint64)
    items = np.zeros(nnz, np.uint64)
    ts = np.zeros(n, np.uint32)
    sids = np.zeros(n, np.uint64)
    L.orc_sessions_copy(h, off, items, ts, sids)
    L.orc_sessions_free(h)
    return off, items, ts, sids


class OracleIndex:
    """Restatement of VMISIndex (src/vmisknn/vmis_index.rs:28-35) built by prepare_hashmap (:422-528)."""

    def __init__(self, sess_off, items, ts, m_index, max_len, idf_weighting=1.0, fast=False, wanted=None, threads=1, items_hint=0):
        """wanted = item ids: the RESTRICTED index for a query sample (prepare_hashmap_restricted: posting lists only for these items, idf of all, built
        on `threads` threads; queries naming other known items are refused; items_hint = an upper bound of the number of distinct items, sizes the count table).  The session arrays are then borrowed, not copied."""
        self.L = lib()
        self.sess_off, self.items, self.ts = _u64(sess_off), _u64(items), np.ascontiguousarray(ts, np.uint32)
        if wanted is not None:
            w = _u64(np.unique(np.asarray(wanted, dtype=np.uint64)))
            self.h = self.L.orc_index_build_restricted(self.sess_off, self.items, self.ts, len(self.ts), int(m_index), int(max_len), float(idf_weighting),
                                                       w, len(w), int(threads), int(items_hint))
            if not self.h:
                raise MemoryError("restricted oracle index: item table overflow")
            return
        self.h = self.L.orc_index_build(self.sess_off, self.items, self.ts, len(self.ts), int(m_index),
                                        int(max_len), float(idf_weighting), int(bool(fast)))

    @classmethod
    def from_parts(cls, item_ids, lists, idf, flags, sess_off, sess_items, ts, tie_rank=None):
        """VMISIndex::new restated (vmis_index.rs:85-314): the index a PRE-BUILT (Avro) index's records make -- posting lists (`lists`: one sequence of session
        indices per item, in the producer's order), idf and flags (bit0 IsAdult, bit1 ForSale) AS GIVEN, session rows at their SessionIndex.  tie_rank [n_sessions]:
        the canonical form's order among sessions of EQUAL timestamp (larger = more recent; default: by session index) -- the reference leaves it open."""
        self = cls.__new__(cls)
        self.L = lib()
        self.sess_off, self.items, self.ts = _u64(sess_off), _u64(sess_items), np.ascontiguousarray(ts, np.uint32)
        self.tie = None if tie_rank is None else np.ascontiguousarray(tie_rank, np.uint32)
        if isinstance(lists, tuple):       # (list_off u64[n + 1], sessions u32[nnz]) already flat
            lo, flat = _u64(lists[0]), np.ascontiguousarray(lists[1], np.uint32)
        else:
            lo = np.zeros(len(item_ids) + 1, np.uint64)
            lo[1:] = np.cumsum([len(x) for x in lists])
            flat = np.ascontiguousarray(np.concatenate([np.asarray(x, np.uint32) for x in lists]) if len(lists) else np.zeros(0, np.uint32), np.uint32)
        if len(flat) == 0:
            flat = np.zeros(1, np.uint32)
        self.h = self.L.orc_index_from_parts(_u64(item_ids), lo, flat, np.ascontiguousarray(idf, np.float64), np.ascontiguousarray(flags, np.uint8), len(item_ids),
                                             self.sess_off, self.items, self.ts, len(self.ts), _ptr(self.tie))
        return self

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_index_free(self.h)
            self.h = None

    @property
    def num_items(self):
        return self.L.orc_index_num_items(self.h)

    @property
    def total_pairs(self):
        return self.L.orc_index_total_pairs(self.h)

    def set_attributes(self, ids, flags):
        self.L.orc_index_set_attributes(self.h, _u64(ids), np.ascontiguousarray(flags, np.uint8), len(ids))

    def postings(self, item, cap=1 << 20):
        out = np.zeros(cap, np.uint32)
        idf = C.c_double()
        n = self.L.orc_index_postings(self.h, int(item), out, cap, C.byref(idf))
        if n < 0:
            return None, None
        return out[:n].copy(), idf.value

    def predict_literal(self, session, k, m, how_many, business=False):
        s = _u64(session)
        ids, sc, n = np.zeros(max(how_many, 1), np.uint64), np.zeros(max(how_many, 1)), C.c_size_t()
        rc = self.L.orc_predict_literal(self.h, s, len(s), k, m, how_many, int(business), ids, sc, C.byref(n))
        if rc:
            raise RuntimeError("reference would panic on this input")
        return ids[:n.value].copy(), sc[:n.value].copy()

    def find_neighbors_literal(self, session, k, m):
        s = _u64(session)
        sid, sc, n = np.zeros(max(k, 1), np.uint32), np.zeros(max(k, 1)), C.c_size_t()
        self.L.orc_find_neighbors_literal(self.h, s, len(s), k, m, sid, sc, C.byref(n))
        return sid[:n.value].copy(), sc[:n.value].copy()

    def predict_canonical(self, session, k, m, how_many, business=False, stats=False):
        s = _u64(session)
        ids, sc, n = np.zeros(max(how_many, 1), np.uint64), np.zeros(max(how_many, 1)), C.c_size_t()
        st = np.zeros(7, np.uint64)
        rc = self.L.orc_predict_canonical(self.h, s, len(s), k, m, how_many, int(business), ids, sc, C.byref(n), _ptr(st))
        if rc:
            raise RuntimeError("reference would panic on this input")
        r = (ids[:n.value].copy(), sc[:n.value].copy())
        return r + (st,) if stats else r

    def neighbors_canonical(self, session, k, m):
        s = _u64(session)
        sid, num, n, U = np.zeros(max(k, 1), np.uint32), np.zeros(max(k, 1), np.uint32), C.c_size_t(), C.c_size_t()
        self.L.orc_neighbors_canonical(self.h, s, len(s), k, m, sid, num, C.byref(n), C.byref(U))
        return sid[:n.value].copy(), num[:n.value].copy(), U.value

    def scores_canonical(self, session, k, m, cap=1 << 20):
        s = _u64(session)
        ids, sc, acc = np.zeros(cap, np.uint64), np.zeros(cap), np.zeros(cap, np.int64)
        n = self.L.orc_scores_canonical(self.h, s, len(s), k, m, ids, sc, acc, cap)
        if n < 0:
            raise RuntimeError("reference would panic on this input")
        return ids[:n].copy(), sc[:n].copy(), acc[:n].copy()

    def predict_batch(self, which, items_flat, q_off, k, m, how_many, business=False, threads=1,
                      want_results=True, want_stats=False, want_latency=False):
        """which: 'literal' | 'canonical'.  Returns dict(elapsed, ids, scores, counts, stats, lat_us)."""
        items_flat, q_off = _u64(items_flat), np.ascontiguousarray(q_off, np.uint32)
        nq = len(q_off) - 1
        ids = np.zeros((nq, how_many), np.uint64) if want_results else None
        sc = np.zeros((nq, how_many)) if want_results else None
        cnt = np.zeros(nq, np.uint32) if want_results else None
        st = np.zeros((nq, 7), np.uint64) if want_stats else None
        lat = np.zeros(nq) if want_latency else None
        el = self.L.orc_predict_batch(self.h, 0 if which == "literal" else 1, items_flat, q_off, nq, k, m, how_many,
                                      int(business), int(threads), _ptr(ids), _ptr(sc), _ptr(cnt), _ptr(st), _ptr(lat))
        if el < 0:
            raise ValueError("a restricted oracle index was asked about an item outside its `wanted` set")
        return dict(elapsed=el, ids=ids, scores=sc, counts=cnt, stats=st, lat_us=lat)


def kat_itemscore_heap(ids, scores, how_many):
    out = np.zeros(len(ids), np.uint64)
    n = lib().orc_kat_itemscore_heap(_u64(ids), np.ascontiguousarray(scores, np.float64), len(ids), how_many, out)
    return out[:n].tolist()


def kat_itemscore_sorted(ids, scores):
    out = np.zeros(len(ids), np.uint64)
    n = lib().orc_kat_itemscore_sorted(_u64(ids), np.ascontiguousarray(scores, np.float64), len(ids), out)
    return out[:n].tolist()


def kat_sessiontime_heap(ids, times, how_many):
    out = np.zeros(len(ids), np.uint32)
    n = lib().orc_kat_sessiontime_heap(np.ascontiguousarray(ids, np.uint32), np.ascontiguousarray(times, np.uint32),
                                       len(ids), how_many, out)
    return out[:n].tolist()

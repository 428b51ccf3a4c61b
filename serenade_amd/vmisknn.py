"""Host-side mirror of the reference's interface for the VMIS-kNN hot path, over the C ABI.

Reference (bolcom/serenade, Rust)                      here
---------------------------------------------------   -------------------------------------------
vmis_index::VMISIndex::new_from_csv(path, m, idf_w)    VMISIndex.new_from_csv(path, m, idf_w)
  (src/vmisknn/vmis_index.rs:38-83)
vmisknn::predict(&index, &session, k, m, how_many,     predict(index, session, k, m, how_many,
                 enable_business_logic)                        enable_business_logic)
  -> BinaryHeap<ItemScore>, callers take               -> list[ItemScore] already in
     .into_sorted_vec()  (src/vmisknn/mod.rs:118-215)     into_sorted_vec() order (score-descending)
ItemScore { id: u64, score: f64 } (mod.rs:45-49)       ItemScore(id, score)

All compute happens in libserenade_hip.so on the GPU.  There is no Python or CPU implementation of
the path in this package; without the library or without a GPU the calls raise.
"""
import ctypes as C
from collections import namedtuple

import numpy as np

from . import capi
from .capi import SerenadeError  # noqa: F401  (re-export)

ItemScore = namedtuple("ItemScore", ["id", "score"])


class VMISIndex:
    """Owns an srn_index_t handle (flat CSR index in HBM)."""

    def __init__(self, handle):
        self._h = handle

    # ---- constructors ---------------------------------------------------------------------
    @classmethod
    def new_from_csv(cls, path_to_training, m_most_recent_sessions, idf_weighting, max_session_len=0, device=0):
        """VMISIndex::new_from_csv (vmis_index.rs:38-83).  max_session_len=0: exact p99.5 of session lengths
        (the reference uses a t-digest estimate of the same quantile, vmis_index.rs:689-716)."""
        h = C.c_void_p()
        capi.check(capi.lib().srn_index_new_from_csv(str(path_to_training).encode(), int(m_most_recent_sessions),
                                                     float(idf_weighting), int(max_session_len), int(device), C.byref(h)))
        return cls(h)

    @classmethod
    def from_sessions(cls, sess_off, items, max_ts, m_index, max_session_len, idf_weighting=1.0, device=0, builder="host"):
        """prepare_hashmap (vmis_index.rs:422-528) on sessions already in (ascending, de-duplicated) row form.
        builder="gpu" constructs the identical index with rocPRIM sorts on the device (needs device >= 0)."""
        sess_off, items, max_ts = capi.as_u64(sess_off), capi.as_u64(items), capi.as_u32(max_ts)
        if len(sess_off) != len(max_ts) + 1 or (len(sess_off) and int(sess_off[-1]) != len(items)):
            raise ValueError("inconsistent session CSR")
        v = capi.SessionsView(sess_off.ctypes.data, items.ctypes.data, max_ts.ctypes.data, len(max_ts))
        h = C.c_void_p()
        build = capi.lib().srn_index_build_gpu if builder == "gpu" else capi.lib().srn_index_build
        capi.check(build(C.byref(v), int(m_index), int(max_session_len), float(idf_weighting), int(device), C.byref(h)))
        return cls(h)

    @classmethod
    def new_from_avro(cls, base_path, device=0):
        """VMISIndex::new(base_path) (vmis_index.rs:85-314): pre-built index from <base>/itemindex and <base>/sessionindex Avro files."""
        h = C.c_void_p()
        capi.check(capi.lib().srn_index_new_from_avro(str(base_path).encode(), int(device), C.byref(h)))
        return cls(h)

    @classmethod
    def load(cls, path, device=0):
        h = C.c_void_p()
        capi.check(capi.lib().srn_index_load(str(path).encode(), int(device), C.byref(h)))
        return cls(h)

    def save(self, path):
        capi.check(capi.lib().srn_index_save(self._h, str(path).encode()))

    def close(self):
        if getattr(self, "_h", None) and capi is not None and getattr(capi, "lib", None) is not None:   # (None at interpreter shutdown)
            capi.lib().srn_index_free(self._h)
            self._h = None

    __del__ = close

    # ---- accessors ------------------------------------------------------------------------
    @property
    def info(self):
        out = capi.IndexInfo()
        capi.check(capi.lib().srn_index_info(self._h, C.byref(out)))
        return {n: getattr(out, n) for n, _ in capi.IndexInfo._fields_}

    def set_attributes(self, item_ids, flags):
        ids, fl = capi.as_u64(item_ids), np.ascontiguousarray(flags, np.uint8)
        capi.check(capi.lib().srn_index_set_attributes(self._h, capi.ptr(ids), capi.ptr(fl), len(ids)))

    def postings(self, item_id):
        """(reference session indices most-recent-first, idf) of one item, or (None, None) if unknown."""
        n, idf = C.c_int64(), C.c_double()
        capi.check(capi.lib().srn_index_postings(self._h, int(item_id), None, 0, C.byref(n), C.byref(idf)))
        if n.value < 0:
            return None, None
        out = np.zeros(max(n.value, 1), np.uint32)
        capi.check(capi.lib().srn_index_postings(self._h, int(item_id), capi.ptr(out), n.value, C.byref(n), C.byref(idf)))
        return out[:n.value], idf.value

    # the trait's other accessors (SimilarityComputationNew, src/vmisknn/similarity_indexed.rs:9-23)
    def idf(self, item_id):
        """idf(&u64) -> f64 (vmis_index.rs:321-323).  The reference panics on an unknown item; here: KeyError."""
        n, idf = C.c_int64(), C.c_double()
        capi.check(capi.lib().srn_index_postings(self._h, int(item_id), None, 0, C.byref(n), C.byref(idf)))
        if n.value < 0:
            raise KeyError(item_id)
        return idf.value

    def items_for_session(self, session):
        """items_for_session(&u32) -> &[u64] (vmis_index.rs:317-319): ascending item ids of a kept session, by its reference session index."""
        n = C.c_size_t()
        capi.check(capi.lib().srn_index_items_for_session(self._h, int(session), None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), np.uint64)
        capi.check(capi.lib().srn_index_items_for_session(self._h, int(session), capi.ptr(out), n.value, C.byref(n)))
        return out[:n.value]

    def serve_start(self, k, m, how_many, enable_business_logic=False, lanes=1, max_items_in_session=4, idle_ms=2000):
        """srn_index_serve_start: park `lanes` resident workgroups that answer predict() calls of exactly these parameters without a kernel launch."""
        capi.check(capi.lib().srn_index_serve_start(self._h, int(k), int(m), int(how_many), int(bool(enable_business_logic)), int(lanes), int(max_items_in_session), int(idle_ms)))

    def serve_stop(self):
        capi.check(capi.lib().srn_index_serve_stop(self._h))

    def serve_stats(self):
        """(answered by a resident workgroup, sent to the launch path, kernel launches, resident workgroups)"""
        a, b, c, n = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint32()
        capi.check(capi.lib().srn_index_serve_stats(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(n)))
        return a.value, b.value, c.value, n.value

    def session_recency(self):
        """Recency rank of every reference session (0 = oldest; 0xFFFFFFFF: not kept): the total order behind "most recent", ties among equal timestamps included."""
        n = int(self.info["n_sessions_total"])
        out = np.zeros(max(n, 1), np.uint32)
        capi.check(capi.lib().srn_index_session_recency(self._h, capi.ptr(out), len(out)))
        return out[:n]

    def find_attributes(self, item_id):
        """find_attributes(&u64) -> Option<&ProductAttributes> (vmis_index.rs:417-419): the SRN_ATTR_* bits, or None."""
        fl = C.c_uint8()
        capi.check(capi.lib().srn_index_find_attributes(self._h, int(item_id), C.byref(fl)))
        return None if fl.value == capi.ATTR_NONE else int(fl.value)

    def find_neighbors(self, evolving_session, k, m):
        """find_neighbors(&[u64], k, m) (vmis_index.rs:325-415) -> (reference session indices u32[K], similarities f64[K]), best first."""
        ev = capi.as_u64(evolving_session)
        ses, sc, n = np.zeros(max(int(k), 1), np.uint32), np.zeros(max(int(k), 1)), C.c_size_t()
        capi.check(capi.lib().srn_find_neighbors(self._h, capi.ptr(ev), len(ev), int(k), int(m), capi.ptr(ses), capi.ptr(sc), C.byref(n)))
        return ses[:n.value], sc[:n.value]

    def kernel_timing(self, enable=True):
        """srn_kernel_timing: per-kernel HIP events on every batch call (what last_kernel_ms / kernel_times* read); off by default -- each event idles the stream ~6 us."""
        capi.check(capi.lib().srn_kernel_timing(self._h, 1 if enable else 0))

    def last_kernel_ms(self):
        """(avg ms of the main predict kernel, ms of the retry pass, #queries retried) for the last call."""
        a, b, r = C.c_double(), C.c_double(), C.c_uint32()
        capi.check(capi.lib().srn_last_kernel_ms(self._h, C.byref(a), C.byref(b), C.byref(r)))
        return a.value, b.value, r.value


    def last_path_counts(self):
        """(queries of the last call, served by the general kernel, served through the global-table pass)."""
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        capi.check(capi.lib().srn_last_path_counts(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def last_mid_count(self):
        """Queries of the last call that the lean fast kernel listed for its MID instantiation (sessions of <= 10 items, 5..8 lists); measurement aid."""
        a = C.c_uint32()
        capi.check(capi.lib().srn_debug_last_mid_count(self._h, C.byref(a)))
        return a.value

    def last_big_count(self):
        """... and how many of those the MID instantiation listed for its BIG form (merged lists beyond the 53 KB layout's buffers); measurement aid."""
        a = C.c_uint32()
        capi.check(capi.lib().srn_debug_last_big_count(self._h, C.byref(a)))
        return a.value

    def kernel_times(self, max_n=64):
        """Per-call (main kernel ms, retry pass ms) of the most recent predict calls, oldest first (HIP events
        recorded on the launch stream around each launch)."""
        a, b, n = np.zeros(max_n), np.zeros(max_n), C.c_uint32()
        capi.check(capi.lib().srn_kernel_times(self._h, max_n, capi.ptr(a), capi.ptr(b), C.byref(n)))
        return a[:n.value].copy(), b[:n.value].copy()


    def kernel_times_detail(self, max_n=64):
        """Per-call ms of (prep kernel, fast kernel alone, all predict launches, global-table retry pass), oldest first."""
        a, b, c, d, n = np.zeros(max_n), np.zeros(max_n), np.zeros(max_n), np.zeros(max_n), C.c_uint32()
        capi.check(capi.lib().srn_kernel_times_detail(self._h, max_n, capi.ptr(a), capi.ptr(b), capi.ptr(c), capi.ptr(d), C.byref(n)))
        return tuple(x[:n.value].copy() for x in (a, b, c, d))

    def debug_phase_cycles(self, enable):
        """Profiling aid: fetch-and-clear the per-phase shader-cycle counters, then switch accounting on/off."""
        out = np.zeros(16, np.uint64)
        capi.check(capi.lib().srn_debug_phase_cycles(self._h, int(bool(enable)), capi.ptr(out)))
        return out


CSR = namedtuple("CSR", ["items_flat", "q_off"])
CSR.__doc__ = "Explicit CSR batch: CSR(items_flat u64[nnz], q_off int[nq + 1]).  Unambiguous, unlike a bare tuple of two arrays."


def _is_csr_pair(sessions):
    """A bare tuple (items_flat, q_off) is read as CSR only if it cannot be two evolving sessions of the same kind: two 1-D integer numpy arrays whose second one starts
    at 0, is non-decreasing and ends at len(first), AND whose dtypes tell them apart -- q_off is uint32 (what the C ABI takes: nobody keeps item ids in it), or
    items_flat is uint64 and q_off any other integer dtype (what np.cumsum hands back).  Two arrays of ONE dtype that also read as (items, offsets) are refused
    (ValueError): since round 4 such a pair -- e.g. int64 items with int64 np.cumsum offsets, accepted as CSR before -- must be passed as
    `CSR(items_flat, q_off)` (README, "Python mirror"); a wrong guess would answer a different question silently."""
    if isinstance(sessions, CSR):
        flat, off = np.asarray(sessions.items_flat), np.asarray(sessions.q_off)
        if flat.ndim != 1 or off.ndim != 1 or len(off) < 1 or int(off[0]) != 0 or int(off[-1]) != len(flat) or (len(off) > 1 and not bool((off[1:] >= off[:-1]).all())):
            raise ValueError("CSR(items_flat, q_off): q_off must start at 0, be non-decreasing and end at len(items_flat)")
        return True
    if not (isinstance(sessions, tuple) and len(sessions) == 2 and all(isinstance(a, np.ndarray) and a.ndim == 1 for a in sessions)):
        return False
    flat, off = sessions
    if not np.issubdtype(off.dtype, np.integer) or not np.issubdtype(flat.dtype, np.integer) or len(off) < 1:
        return False
    looks = int(off[0]) == 0 and int(off[-1]) == len(flat) and (len(off) < 2 or bool((off[1:] >= off[:-1]).all()))
    if not looks:
        return False
    if (off.dtype == np.uint32 and flat.dtype != np.uint32) or (flat.dtype == np.uint64 and off.dtype != np.uint64):
        return True
    # two arrays of one kind whose second one also reads as offsets of the first: refuse to guess
    raise ValueError("ambiguous batch: a tuple of two %s / %s arrays that could be (items_flat, q_off) or two evolving sessions -- "
                     "pass serenade_amd.CSR(items_flat, q_off), or q_off as uint32, or the two sessions as lists" % (flat.dtype, off.dtype))


def _flatten(sessions):
    # CSR input is CSR(items_flat, q_off) or a tuple of two numpy arrays of different kinds, see _is_csr_pair; a tuple of two evolving sessions (lists, or arrays
    # that are not such a pair) is two queries
    if _is_csr_pair(sessions):
        if len(sessions[0]) > 0xFFFFFFFF:
            raise ValueError("more than 2^32 - 1 items in one batch: q_off is 32-bit in the C ABI")
        return capi.as_u64(sessions[0]), capi.as_u32(sessions[1])
    off = np.zeros(len(sessions) + 1, np.uint32)
    off[1:] = np.cumsum([len(s) for s in sessions])
    flat = np.fromiter((x for s in sessions for x in s), dtype=np.uint64, count=int(off[-1]))
    return flat, off


def predict(index, evolving_session, k, m, how_many, enable_business_logic):
    """vmisknn::predict (src/vmisknn/mod.rs:118-125); returns list[ItemScore], score-descending."""
    ev = capi.as_u64(evolving_session)
    ids, sc, n = np.zeros(max(how_many, 1), np.uint64), np.zeros(max(how_many, 1)), C.c_size_t()
    capi.check(capi.lib().srn_predict(index._h, capi.ptr(ev), len(ev), int(k), int(m), int(how_many),
                                      int(bool(enable_business_logic)), capi.ptr(ids), capi.ptr(sc), C.byref(n)))
    return [ItemScore(int(i), float(s)) for i, s in zip(ids[:n.value], sc[:n.value])]


def predict_batch(index, sessions, k, m, how_many, enable_business_logic=False, out=None):
    """Many evolving sessions in one call (list of sequences, or (items_flat, q_off)).
    -> (ids u64[nq, how_many], scores f64[nq, how_many], counts u32[nq]).  out = (ids, scores, counts) of an earlier call of the same
    shape: the result buffers are reused (what a serving / evaluator host does) instead of freshly allocated."""
    flat, off = _flatten(sessions)
    nq = len(off) - 1
    if out is not None:
        ids, sc, cnt = out
        if ids.shape != (nq, how_many) or sc.shape != (nq, how_many) or cnt.shape != (nq,) or ids.dtype != np.uint64 or sc.dtype != np.float64 or cnt.dtype != np.uint32 \
                or not (ids.flags.c_contiguous and sc.flags.c_contiguous and cnt.flags.c_contiguous):
            raise ValueError("out must be (u64[nq, how_many], f64[nq, how_many], u32[nq]), C-contiguous")
    else:
        ids, sc, cnt = np.zeros((nq, how_many), np.uint64), np.zeros((nq, how_many)), np.zeros(nq, np.uint32)
    capi.check(capi.lib().srn_predict_batch(index._h, capi.ptr(flat), capi.ptr(off), nq, int(k), int(m), int(how_many),
                                            capi.FLAG_BUSINESS_LOGIC if enable_business_logic else 0,
                                            capi.ptr(ids), capi.ptr(sc), capi.ptr(cnt)))
    return ids, sc, cnt


def predict_batch_debug(index, sessions, k, m, how_many, enable_business_logic=False, neighbours=True):
    """predict_batch plus per-query stats [nq, 8] = (P, C, K, I, D, H, L, status) and, optionally, the selected
    neighbour sessions (reference session indices) with their integer similarity numerators."""
    flat, off = _flatten(sessions)
    nq = len(off) - 1
    ids, sc, cnt = np.zeros((nq, how_many), np.uint64), np.zeros((nq, how_many)), np.zeros(nq, np.uint32)
    stats = np.zeros((nq, 8), np.uint32)
    nb_s = np.zeros((nq, k), np.uint32) if neighbours else None
    nb_n = np.zeros((nq, k), np.uint32) if neighbours else None
    nb_c = np.zeros(nq, np.uint32) if neighbours else None
    capi.check(capi.lib().srn_predict_batch_debug(index._h, capi.ptr(flat), capi.ptr(off), nq, int(k), int(m), int(how_many),
                                                  capi.FLAG_BUSINESS_LOGIC if enable_business_logic else 0,
                                                  capi.ptr(ids), capi.ptr(sc), capi.ptr(cnt), capi.ptr(stats),
                                                  capi.ptr(nb_s), capi.ptr(nb_n), capi.ptr(nb_c)))
    return dict(ids=ids, scores=sc, counts=cnt, stats=stats, nb_sessions=nb_s, nb_num=nb_n, nb_counts=nb_c)


def reserve(index, nq, max_len, k, m, how_many, enable_business_logic=False, stream=0):
    """srn_index_reserve: size the workspace bound to `stream` so that predict_batch_device calls of up to nq queries allocate nothing."""
    capi.check(capi.lib().srn_index_reserve(index._h, int(nq), int(max_len), int(k), int(m), int(how_many),
                                            capi.FLAG_BUSINESS_LOGIC if enable_business_logic else 0, C.c_void_p(stream)))


def predict_batch_device(index, d_items_flat, d_q_off, nq, max_len, k, m, how_many, enable_business_logic,
                         d_out_ids, d_out_scores, d_out_counts, stream=0, resident=False):
    """Device-resident variant: arguments are raw device addresses (e.g. torch.Tensor.data_ptr()) on the
    index's GPU and a hipStream_t handle (e.g. torch.cuda.current_stream().cuda_stream); asynchronous.  resident=True (SRN_FLAG_INPUTS_RESIDENT):
    the query buffers are complete in device memory now, not pending on `stream`."""
    capi.check(capi.lib().srn_predict_batch_device(index._h, C.c_void_p(d_items_flat), C.c_void_p(d_q_off), int(nq),
                                                   int(max_len), int(k), int(m), int(how_many),
                                                   (capi.FLAG_BUSINESS_LOGIC if enable_business_logic else 0) | (capi.FLAG_INPUTS_RESIDENT if resident else 0),
                                                   C.c_void_p(d_out_ids), C.c_void_p(d_out_scores),
                                                   C.c_void_p(d_out_counts), C.c_void_p(stream)))

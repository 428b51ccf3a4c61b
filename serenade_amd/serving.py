"""Dynamic batching in front of the predict kernel (srn_batcher_*): the serving-side caller of the hot path.

The reference answers each /v1/recommend call with its own vmisknn::predict on an actix worker thread
(src/endpoints/recommend_resource.rs:56-62).  `Batcher.predict` has that call's shape and may be called from many threads
at once; the library folds the waiting calls into one kernel launch.  No HTTP and no session store here."""
import ctypes as C

import numpy as np

from . import capi
from .vmisknn import ItemScore


class Batcher:
    def __init__(self, index, k, m, how_many, enable_business_logic=False, max_batch=4096, max_wait_us=200):
        self._index = index           # keep the index alive
        self.how_many = int(how_many)
        h = C.c_void_p()
        capi.check(capi.lib().srn_batcher_create(index._h, int(max_batch), int(max_wait_us), int(k), int(m), int(how_many),
                                                 int(bool(enable_business_logic)), C.byref(h)))
        self._h = h

    def predict(self, evolving_session):
        """-> [ItemScore(id, score), ...] best first, like serenade_amd.predict; blocks until the batch it joined is done."""
        ev = capi.as_u64(evolving_session)
        ids, sc, n = np.zeros(self.how_many, np.uint64), np.zeros(self.how_many, np.float64), C.c_size_t()
        capi.check(capi.lib().srn_batcher_predict(self._h, capi.ptr(ev), len(ev), capi.ptr(ids), capi.ptr(sc), C.byref(n)))
        return [ItemScore(int(i), float(s)) for i, s in zip(ids[:n.value], sc[:n.value])]

    @property
    def stats(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        capi.check(capi.lib().srn_batcher_stats(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return dict(requests=a.value, batches=b.value, max_batch_seen=c.value)

    def close(self):
        if getattr(self, "_h", None):
            capi.lib().srn_batcher_free(self._h)
            self._h = None

    __del__ = close


def session_key(session_id):
    """u128 key of a session_id string, as the reference builds it (MD5 digest read big-endian, recommend_resource.rs:27-28)."""
    raw = session_id.encode() if isinstance(session_id, str) else bytes(session_id)
    hi, lo = C.c_uint64(), C.c_uint64()
    capi.check(capi.lib().srn_session_key(raw, len(raw), C.byref(hi), C.byref(lo)))
    return (hi.value << 64) | lo.value


class SessionStore:
    """In-memory stand-in for RocksDBSessionStore (src/sessions/mod.rs): same get / update calls, same idle and TTL clocks."""

    def __init__(self, ttl_secs=30 * 60, idle_secs=20 * 60):
        h = C.c_void_p()
        capi.check(capi.lib().srn_session_store_create(int(ttl_secs), int(idle_secs), C.byref(h)))
        self._h = h

    def get_session_items(self, key, now=0, cap=256):
        out, n = np.zeros(cap, np.uint64), C.c_size_t()
        capi.check(capi.lib().srn_session_store_get(self._h, key >> 64, key & (2**64 - 1), int(now), capi.ptr(out), cap, C.byref(n)))
        return [int(x) for x in out[:n.value]]

    def update_session_items(self, key, items, now=0):
        it = capi.as_u64(items)
        capi.check(capi.lib().srn_session_store_update(self._h, key >> 64, key & (2**64 - 1), int(now), capi.ptr(it), len(it)))

    def sweep(self, now=0):
        n = C.c_uint64()
        capi.check(capi.lib().srn_session_store_sweep(self._h, int(now), C.byref(n)))
        return n.value

    def close(self):
        if getattr(self, "_h", None):
            capi.lib().srn_session_store_free(self._h)
            self._h = None

    __del__ = close


def recommend(batcher, store, session_id, item_id, user_consent=True, max_items_in_session=2, now=0):
    """Body of the reference's /v1/recommend handler (recommend_resource.rs:20-65) -> recommended item ids, best first."""
    raw = session_id.encode() if isinstance(session_id, str) else bytes(session_id)
    ids, n = np.zeros(batcher.how_many, np.uint64), C.c_size_t()
    capi.check(capi.lib().srn_recommend(batcher._h, store._h if store is not None else None, raw, len(raw), int(item_id),
                                        int(bool(user_consent)), int(max_items_in_session), int(now), capi.ptr(ids), None, C.byref(n)))
    return [int(i) for i in ids[:n.value]]

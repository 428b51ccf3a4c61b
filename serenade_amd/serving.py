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

"""Dynamic batching (srn_batcher_*): many threads call predict() concurrently, the library folds them into shared launches;
every caller must get exactly what a call of its own would have returned."""
import threading

import numpy as np
import pytest

from helpers import flatten, random_queries, small_dataset

pytestmark = pytest.mark.gpu


def oracle_predictions(off, items, ts, m_index, max_len, idfw, queries, k, m, how_many, business=False):
    """What vmisknn::predict (mod.rs:118-215) returns for every query, from the CPU oracle's canonical form: [[(id, score), ...], ...], best first."""
    from oracle import oracle as O
    oix = O.OracleIndex(off, items, ts, m_index, max_len, idfw, fast=True)
    flat, qo = flatten(queries)
    ref = oix.predict_batch("canonical", flat, qo, k, m, how_many, business, threads=2)
    return [[(int(i), float(s)) for i, s in zip(ref["ids"][q, :c], ref["scores"][q, :c])] for q, c in enumerate(ref["counts"])]


def same_predictions(got, want):
    """ids and order identical, scores within 1e-12 relative (the bound of tests/test_gpu_parity.py)."""
    assert [i for i, _ in got] == [i for i, _ in want], (got, want)
    np.testing.assert_allclose([s for _, s in got], [s for _, s in want], rtol=1e-12, atol=0)


def test_batcher_matches_direct_predict_under_concurrency():
    import serenade_amd as sa
    from serenade_amd.serving import Batcher
    off, items, ts, ids = small_dataset(31, n_sessions=4000, n_items=500)
    gix = sa.VMISIndex.from_sessions(off, items, ts, 300, 12, 1.0)
    qs = random_queries(4, ids, 600, max_len=6)
    want = oracle_predictions(off, items, ts, 300, 12, 1.0, qs, 100, 300, 21)      # the ORACLE's answers, not another HIP call's
    b = Batcher(gix, 100, 300, 21, False, max_batch=64, max_wait_us=2000)
    got = [None] * len(qs)
    errs = []

    def worker(lo, hi):
        try:
            for i in range(lo, hi):
                got[i] = b.predict(qs[i])
        except Exception as e:   # noqa: BLE001
            errs.append(e)

    n_threads = 24
    step = (len(qs) + n_threads - 1) // n_threads
    threads = [threading.Thread(target=worker, args=(t * step, min(len(qs), (t + 1) * step))) for t in range(n_threads)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errs, errs
    for g, w in zip(got, want):
        same_predictions(g, w)
    st = b.stats
    assert st["requests"] == len(qs) and st["batches"] < len(qs) and 1 < st["max_batch_seen"] <= 64, st
    # a bad request fails alone (same code and message class as srn_predict), the batcher keeps serving
    with pytest.raises(sa.SerenadeError) as e:
        b.predict([])
    assert e.value.code == -1
    with pytest.raises(sa.SerenadeError) as e:
        b.predict([1] * 300)
    assert e.value.code == -4
    same_predictions(b.predict(qs[0]), want[0])
    b.close()


def test_batcher_rejects_bad_configuration():
    import serenade_amd as sa
    from serenade_amd.serving import Batcher
    off, items, ts, ids = small_dataset(32, n_sessions=200, n_items=50)
    gix = sa.VMISIndex.from_sessions(off, items, ts, 50, 12, 1.0)
    for kw in (dict(k=0, m=10, how_many=5), dict(k=10, m=10, how_many=100000), dict(k=10, m=10, how_many=5, max_batch=0)):
        with pytest.raises(sa.SerenadeError):
            Batcher(gix, kw["k"], kw["m"], kw["how_many"], max_batch=kw.get("max_batch", 16))


def test_recommend_follows_the_reference_handler():
    """srn_recommend == the body of v1_recommend (recommend_resource.rs:20-65) replayed in Python over the CPU ORACLE's predict:
    session read with the idle rule (sessions/mod.rs:37-72), append unless the click repeats the last item, drop the oldest beyond the
    limit (recommend_resource.rs:39-54), predict on the evolving session (:56), ids of into_sorted_vec() (:58-62)."""
    import serenade_amd as sa
    from oracle import oracle as O
    from serenade_amd.serving import Batcher, SessionStore, recommend, session_key
    off, items, ts, ids = small_dataset(33, n_sessions=3000, n_items=300)
    gix = sa.VMISIndex.from_sessions(off, items, ts, 200, 12, 1.0)
    oix = O.OracleIndex(off, items, ts, 200, 12, 1.0, fast=True)
    b = Batcher(gix, 50, 200, 21, False, max_batch=32, max_wait_us=100)
    store = SessionStore(ttl_secs=1800, idle_secs=1200)
    rng = np.random.default_rng(5)
    model, now, max_items = {}, 10_000, 3
    for step in range(400):
        now += int(rng.integers(0, 500))
        sid = "visitor-%d" % rng.integers(0, 12)
        item = int(ids[rng.integers(0, len(ids))]) if rng.random() < 0.9 else 999_999_999   # sometimes unknown to the index
        consent = rng.random() < 0.85
        if consent:
            sess, t = model.get(sid, ([], 0))
            if now - t > 1200:
                sess = []
            sess = list(sess)
            if not sess:
                sess.append(item)
            elif sess[-1] != item:
                sess.append(item)
                if len(sess) > max_items:
                    sess.pop(0)
            model[sid] = (sess, now)
        else:
            sess = [item]
        want = [int(i) for i in oix.predict_canonical(sess, 50, 200, 21, False)[0]]
        assert recommend(b, store, sid, item, consent, max_items, now=now) == want, (step, sid, sess)
        if consent:
            assert store.get_session_items(session_key(sid), now=now) == sess
    with pytest.raises(sa.SerenadeError):
        recommend(b, None, "x", 1, True, 3)           # consent needs a store
    assert recommend(b, None, "x", int(ids[0]), False, 3) == [int(i) for i in oix.predict_canonical([int(ids[0])], 50, 200, 21, False)[0]]
    b.close()
    store.close()

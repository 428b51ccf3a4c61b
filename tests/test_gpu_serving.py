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


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_persistent_latency_path_against_the_oracle():
    """srn_index_serve_start (round 6): resident workgroups answer srn_predict -- the reference's call shape, recommend_resource.rs:56 / evaluator.rs:58 -- without a kernel
    launch.  Every answer against the canonical oracle (a session the resident form cannot finish silently takes the launch path: same rows); calls with other
    parameters are not served by it; a call that frees device memory (a workspace that grows) makes the resident workgroups leave instead of waiting for them for ever,
    and they come back; they leave by themselves after idle_ms; concurrent callers share the lanes."""
    import threading
    import time
    import serenade_amd as sa
    from serenade_amd import synth
    from oracle import oracle as O
    inter, n_items, k, m, idfw = synth.CONFIGS["tiny"]
    off, items, ts = synth.training_sessions(inter, n_items)
    gix = sa.VMISIndex.from_sessions(off, items, ts, m, 34, idfw)
    oix = O.OracleIndex(off, items, ts, m, 34, idfw, fast=True)
    qi, qo = synth.queries(400, n_items, max_items=10)
    sessions = [qi[qo[q]:qo[q + 1]] for q in range(600)]
    want = [oix.predict_canonical(s, k, m, 21) for s in sessions]

    def check(q, got):
        ids, sc = want[q]
        assert [i for i, _ in got] == ids.tolist(), q
        np.testing.assert_allclose([x for _, x in got], sc, rtol=1e-12, atol=0)

    sa.predict(gix, sessions[0], k, m, 21, False)                          # (the launch path's workspace exists before anything is resident)
    gix.serve_start(k, m, 21, False, lanes=2, max_items_in_session=10, idle_ms=1500)
    assert gix.serve_stats()[2:] == (2, 4)                                  # 2 workgroups of the lean form + 2 of the form for 5..10 items: one launch per form
    for q in range(300):
        check(q, sa.predict(gix, sessions[q], k, m, 21, False))
    served, not_served, launches, lanes = gix.serve_stats()
    assert served + not_served == 300 and served >= 270 and 2 <= launches <= 12, (served, not_served, launches)   # (a session that took the launch path may have grown its workspace: a hipFree, before which the resident workgroups leave)
    # other parameters: the launch path, same answers as ever
    got = sa.predict(gix, sessions[5], k // 2, m, 21, False)
    ids, sc = oix.predict_canonical(sessions[5], k // 2, m, 21)
    assert [i for i, _ in got] == ids.tolist() and gix.serve_stats()[0] == served
    # a batch whose workspace has to grow frees device memory: the resident workgroups leave first (no deadlock) and are started again on demand
    bq, bo = synth.queries(3000, n_items)
    rows = sa.predict_batch(gix, (bq, bo), k, m, 21, False)
    ref = oix.predict_batch("canonical", bq, bo, k, m, 21, False, threads=4)
    assert np.array_equal(rows[0], ref["ids"]) and np.array_equal(rows[2], ref["counts"])
    for q in range(300, 340):
        check(q, sa.predict(gix, sessions[q], k, m, 21, False))
    s2 = gix.serve_stats()
    assert s2[0] > served and s2[2] > launches, s2                          # served again, after a restart
    # idle: they leave by themselves ...
    time.sleep(2.5)
    launches = gix.serve_stats()[2]
    for q in range(340, 360):
        check(q, sa.predict(gix, sessions[q], k, m, 21, False))
    assert gix.serve_stats()[2] > launches                                  # ... and come back
    # concurrent callers share the lanes; who finds them taken runs the launch path
    errs = []

    def worker(t):
        try:
            for q in range(360 + t * 60, 360 + (t + 1) * 60):
                check(q, sa.predict(gix, sessions[q], k, m, 21, False))
        except Exception as e:  # pragma: no cover
            errs.append(repr(e))
    th = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for t in th:
        t.start()
    for _ in range(3):          # stopped and started again UNDER the callers: a stop retires the resident state (a caller may still hold it), the calls fall through to the launch path
        time.sleep(0.01)
        gix.serve_stop()
        time.sleep(0.01)
        gix.serve_start(k, m, 21, False, lanes=2, max_items_in_session=10, idle_ms=1500)
    for t in th:
        t.join()
    assert not errs, errs
    gix.serve_stop()
    assert gix.serve_stats()[3] == 0
    check(1, sa.predict(gix, sessions[1], k, m, 21, False))
    gix.close()

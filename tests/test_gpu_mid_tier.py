"""The fast kernel's MID instantiation (srn_fast.hip, round 4): evolving sessions of up to 10 items with up to 8 posting lists and similarity numerators up to 63 --
the reference's whole grid of last_items_in_session (src/hyperparameter/hyperparamgrid.rs:93-139) -- served between the lean fast kernel and the general kernel.

Everything goes through the C ABI and is compared with the canonical CPU oracle (find_neighbors src/vmisknn/vmis_index.rs:325-415, predict src/vmisknn/mod.rs:118-215);
the same batches with SRN_NO_MID=1 (the launch sequence of round 3: what the lean kernel cannot take goes straight to the general kernel) must give the same bytes.
"""
import numpy as np
import pytest

from helpers import flatten, random_queries, small_dataset

pytestmark = pytest.mark.gpu

SCORE_RTOL = 1e-12


def _oracle():
    from oracle import oracle
    return oracle


def _against_oracle(gix, oix, queries, k, m, n, business=False):
    import serenade_amd as sa
    ids, scores, counts = sa.predict_batch(gix, queries, k, m, n, business)
    flat, off = flatten(queries)
    ref = oix.predict_batch("canonical", flat, off, k, m, n, business, threads=4)
    assert np.array_equal(counts, ref["counts"]), "result counts differ"
    for q in range(len(queries)):
        c = int(ref["counts"][q])
        assert np.array_equal(ids[q, :c], ref["ids"][q, :c]), (q, queries[q], ids[q, :c], ref["ids"][q, :c])
        np.testing.assert_allclose(scores[q, :c], ref["scores"][q, :c], rtol=SCORE_RTOL, atol=0)
        assert not ids[q, c:].any() and not scores[q, c:].any(), "row tails must read 0"
    return ids, scores, counts


@pytest.fixture
def no_mid(monkeypatch):
    from serenade_amd import capi

    def switch(off):
        if off:
            monkeypatch.setenv("SRN_NO_MID", "1")
        else:
            monkeypatch.delenv("SRN_NO_MID", raising=False)
        capi.reload_knobs()
    yield switch
    monkeypatch.undo()
    capi.reload_knobs()


def _long_queries(seed, ids, n, lo, hi, unknown_rate=0.05, dup_rate=0.1):
    rng = np.random.default_rng(seed)
    w = 1.0 / np.arange(1, len(ids) + 1) ** 0.9
    w /= w.sum()
    qs = []
    for _ in range(n):
        ln = int(rng.integers(lo, hi + 1))
        q = ids[rng.choice(len(ids), size=ln, p=w)].tolist()
        for j in range(ln):
            if rng.random() < unknown_rate:
                q[j] = int(7 + rng.integers(0, 1000))
            elif j and rng.random() < dup_rate:
                q[j] = q[int(rng.integers(0, j))]
        qs.append([int(x) for x in q])
    return qs


def test_sessions_of_five_to_ten_items_vs_oracle(no_mid):
    """Posting lists long enough for both cuts to bite with 5..8 lists merged; k-cut classes up to 55 (L = 10)."""
    import serenade_amd as sa
    O = _oracle()
    off, items, ts, ids = small_dataset(31, n_sessions=40000, n_items=150, max_len=12)
    gix = sa.VMISIndex.from_sessions(off, items, ts, 3000, 12, 1.0)
    oix = O.OracleIndex(off, items, ts, 3000, 12, 1.0)
    qs = _long_queries(13, ids, 500, 1, 10)
    for (k, m, n) in [(100, 500, 21), (1500, 2500, 21), (50, 2560, 24), (700, 1000, 5), (1, 1, 1)]:
        no_mid(False)
        got = _against_oracle(gix, oix, qs, k, m, n)
        nq, general, _glob = gix.last_path_counts()
        mid = gix.last_mid_count()
        assert mid >= len(qs) // 4, "sessions of 5..10 items should have been listed for the MID instantiation (%d of %d)" % (mid, nq)
        assert general <= mid // 2 + 8, "the MID instantiation should serve most of what it is handed (%d listed, %d reached the general kernel)" % (mid, general)   # (what it passes on: merged lists beyond the LDS buffers)
        no_mid(True)
        ref = sa.predict_batch(gix, qs, k, m, n, False)
        assert gix.last_mid_count() == 0
        for a, b in zip(got, ref):
            assert np.array_equal(a, b), "with and without the MID tier the results must be the same bytes"


def test_tenth_position_weight_zero(no_mid):
    """L = 10: linear_score(10) = 0 (mod.rs:110-116) -- neighbours that match only the oldest item add 0 to every item of their row, and such items are RETURNED
    (score 0) when the positive scores do not fill the top n: the MID instantiation hands exactly those queries to the general kernel."""
    import serenade_amd as sa
    O = _oracle()
    off, items, ts, ids = small_dataset(5, n_sessions=6000, n_items=200, max_len=8)
    gix = sa.VMISIndex.from_sessions(off, items, ts, 1000, 10, 1.0)
    oix = O.OracleIndex(off, items, ts, 1000, 10, 1.0)
    rng = np.random.default_rng(4)
    qs = []
    for i in range(400):
        old = int(ids[rng.integers(0, len(ids))])
        if i % 2:   # oldest item known, the nine recent ones unknown: every neighbour has weight 0
            qs.append([old] + [int(7 + rng.integers(0, 1000)) for _ in range(9)])
        else:       # oldest item known and rare ones behind it: a mix of zero and positive weights
            qs.append([old] + [int(x) for x in ids[rng.integers(len(ids) // 2, len(ids), size=9)]])
    no_mid(False)
    for (k, m, n) in [(100, 500, 21), (1500, 1000, 24), (20, 50, 5)]:
        ids_, scores, counts = _against_oracle(gix, oix, qs, k, m, n)
        assert (scores[np.arange(len(qs)) % 2 == 1][:, 0] == 0).all() and (counts[1::2] > 0).any(), "all-zero-weight queries return items of score 0"
        assert gix.last_mid_count() > 0


def test_many_lists_business_rules_and_ties(no_mid):
    """Business rules on (mod.rs:162-182) through the MID instantiation; tied timestamps (canonical order); idf weighting 2."""
    import serenade_amd as sa
    O = _oracle()
    off, items, ts, ids = small_dataset(8, n_sessions=20000, n_items=120, tied_timestamps=True, max_len=9)
    gix = sa.VMISIndex.from_sessions(off, items, ts, 2500, 9, 2.0)
    oix = O.OracleIndex(off, items, ts, 2500, 9, 2.0)
    rng = np.random.default_rng(4)
    known = np.unique(items)
    flags = rng.choice(np.array([0, 1, 2, 3, 0xFF], np.uint8), size=len(known), p=[0.1, 0.05, 0.55, 0.2, 0.1])
    gix.set_attributes(known, flags)
    oix.set_attributes(known, flags)
    business = True
    qs = _long_queries(3, ids, 400, 5, 9, unknown_rate=0.02, dup_rate=0.05)
    no_mid(False)
    for (k, m, n) in [(1500, 2500, 21), (300, 800, 10)]:
        _against_oracle(gix, oix, qs, k, m, n, business)
        assert gix.last_mid_count() >= len(qs) // 2
        if business:
            _against_oracle(gix, oix, qs, k, m, n, False)


def test_one_long_session_no_longer_sends_the_batch_to_the_general_kernel(no_mid):
    """Up to round 3 the fast kernel was chosen per LAUNCH on the batch's longest session: one session of nine items and 4 095 short ones all went through the general
    kernel.  The admission is per query now."""
    import serenade_amd as sa
    O = _oracle()
    off, items, ts, ids = small_dataset(12, n_sessions=8000, n_items=400, max_len=10)
    gix = sa.VMISIndex.from_sessions(off, items, ts, 500, 30, 1.0)
    oix = O.OracleIndex(off, items, ts, 500, 30, 1.0)
    qs = random_queries(6, ids, 2000, max_len=4)
    qs[17] = [int(x) for x in ids[:25]]            # 25 items: the general kernel's, whatever the tiers in front of it
    qs[900] = [int(x) for x in ids[5:14]]          # 9 items
    no_mid(False)
    _against_oracle(gix, oix, qs, 100, 500, 21)
    nq, general, _ = gix.last_path_counts()
    assert general < 100, "short sessions must stay on the fast kernel (%d of %d reached the general kernel)" % (general, nq)


def test_synthetic_tiny_long_sessions(no_mid):
    """The generator's tiny config (20 K items: scored items lie outside the direct-mapped range, so the per-chunk integer floors, the sketch filter, walk B and the exact
    table all run) with evaluator-style sessions of up to 10 items: oracle-exact, the same bytes with and without the tier, and the tier serves what it is handed."""
    import serenade_amd as sa
    from serenade_amd import synth
    O = _oracle()
    inter, n_items, k, m, idfw = synth.CONFIGS["tiny"]
    off, items, ts = synth.training_sessions(inter, n_items)
    gix = sa.VMISIndex.from_sessions(off, items, ts, m, 34, idfw)
    oix = O.OracleIndex(off, items, ts, m, 34, idfw, fast=True)
    for max_items in (5, 8, 10):
        qi, qo = synth.queries(1500, n_items, max_items=max_items)
        nq = min(len(qo) - 1, 3000)
        qs = [qi[qo[i]:qo[i + 1]].tolist() for i in range(nq)]
        no_mid(False)
        got = _against_oracle(gix, oix, qs, k, m, synth.HOW_MANY)
        nq_, general, _ = gix.last_path_counts()
        mid = gix.last_mid_count()
        assert mid > 0 and general <= mid // 4 + 8, "max_items %d: %d listed for the MID instantiation, %d reached the general kernel" % (max_items, mid, general)
        no_mid(True)
        ref = sa.predict_batch(gix, qs, k, m, synth.HOW_MANY, False)
        for a, b in zip(got, ref):
            assert np.array_equal(a, b), "with and without the MID tier the results must be the same bytes"


def test_latency_path_runs_the_fast_launch_sequence(monkeypatch):
    """srn_predict / host batches of <= 256 sessions (the zero-copy latency path): since round 4 the fast kernel's launch sequence (lean -> MID -> general -> finish) on the same pinned
    buffers for calls of <= 32 sessions and for batches with a session of > 8 items (SRN_TINY_FAST=2, the default; 3 = wherever the shape allows it); 0 = prep + general kernel as in rounds 1-3.  Same bytes either way, equal to the oracle."""
    import serenade_amd as sa
    from serenade_amd import synth, capi
    O = _oracle()
    inter, n_items, k, m, idfw = synth.CONFIGS["tiny"]
    off, items, ts = synth.training_sessions(inter, n_items)
    gix = sa.VMISIndex.from_sessions(off, items, ts, m, 34, idfw)
    oix = O.OracleIndex(off, items, ts, m, 34, idfw, fast=True)
    qi, qo = synth.queries(400, n_items, max_items=10)
    qs = [qi[qo[i]:qo[i + 1]].tolist() for i in range(min(len(qo) - 1, 600))]
    qs[3] = [int(x) for x in qi[:14]]   # 14 items: negative weights, the general kernel's whatever the path
    got = {}
    try:
        for mode in ("3", "0", "1", "2"):
            monkeypatch.setenv("SRN_TINY_FAST", mode); capi.reload_knobs()
            rows = []
            for lo in range(0, len(qs), 150):          # batches of <= 256: the latency path
                rows.append(_against_oracle(gix, oix, qs[lo:lo + 150], k, m, synth.HOW_MANY))
            single = [sa.predict(gix, q, k, m, synth.HOW_MANY, False) for q in qs[:40]]   # the reference's call shape
            got[mode] = (rows, [[(r.id, r.score) for r in recs] for recs in single])
    finally:
        monkeypatch.undo(); capi.reload_knobs()
    for mode in ("0", "1", "2"):
        for (a, b) in zip(got["3"][0], got[mode][0]):
            for x, y in zip(a, b):
                assert np.array_equal(x, y), "latency path: the fast launch sequence and the general kernel must give the same bytes"
        assert got["3"][1] == got[mode][1]


def test_big_form_takes_what_outgrows_the_53kb_layout(monkeypatch):
    """MID's BIG form (80 KB of LDS, two workgroups per CU): queries whose merged lists need more than the 53 KB layout's 12 032 words of merge buffers (n > 5 884 staged entries) used to go
    on to the general kernel; with dense lists and 5..10 of them per query they are common here.  Oracle-exact, and the same bytes with SRN_NO_BIG=1."""
    import serenade_amd as sa
    from serenade_amd import capi
    O = _oracle()
    off, items, ts, ids = small_dataset(31, n_sessions=40000, n_items=150, max_len=12)
    gix = sa.VMISIndex.from_sessions(off, items, ts, 3000, 12, 1.0)
    oix = O.OracleIndex(off, items, ts, 3000, 12, 1.0)
    qs = _long_queries(13, ids, 500, 4, 10)
    try:
        monkeypatch.delenv("SRN_NO_BIG", raising=False); capi.reload_knobs()
        got = _against_oracle(gix, oix, qs, 1500, 2500, 21)
        _nq, general, _ = gix.last_path_counts()
        mid, big = gix.last_mid_count(), gix.last_big_count()
        assert big > 0 and big <= mid, (mid, big)
        assert general < big, "the BIG form should serve most of what MID lists for it (%d listed for MID, %d of them for BIG, %d reached the general kernel)" % (mid, big, general)
        monkeypatch.setenv("SRN_NO_BIG", "1"); capi.reload_knobs()
        ref = sa.predict_batch(gix, qs, 1500, 2500, 21, False)
        assert gix.last_big_count() == 0 and gix.last_path_counts()[1] >= big
        for a, b in zip(got, ref):
            assert np.array_equal(a, b), "with and without the BIG form the results must be the same bytes"
    finally:
        monkeypatch.undo(); capi.reload_knobs()


# ---- the LONG instantiation (round 5): sessions of 11..20 items, negative weights from the eleventh position on ---------------------------------------------------

def _no_long(monkeypatch_env, off):
    import os
    from serenade_amd import capi
    if off:
        os.environ["SRN_NO_LONG"] = "1"
    else:
        os.environ.pop("SRN_NO_LONG", None)
    capi.reload_knobs()


def test_sessions_of_eleven_to_twenty_items_vs_oracle():
    """The reference's README lets last_items_in_session_range go to 20 (README.md:116) and linear_score is negative for positions 11..99 (mod.rs:110-116).  Sessions of
    1..20 items in one batch (device admission per query: lean / MID / BIG / LONG side by side), up to 20 posting lists merged, both cuts biting, numerators up to 210;
    against the canonical oracle -- ids, order, scores incl. what negative weights subtract -- and the same bytes as with SRN_NO_LONG=1 (the general kernel for those
    sessions, as until round 4); with the tier on, almost nothing of the batch may reach the general kernel."""
    import serenade_amd as sa
    O = _oracle()
    off, items, ts, ids = small_dataset(41, n_sessions=40000, n_items=400, max_len=14)
    gix = sa.VMISIndex.from_sessions(off, items, ts, 2500, 14, 1.0)
    oix = O.OracleIndex(off, items, ts, 2500, 14, 1.0)
    qs = _long_queries(21, ids, 700, 1, 20, unknown_rate=0.03, dup_rate=0.08)
    n_long = sum(1 for q in qs if len(q) > 10)
    try:
        for (k, m, n) in [(100, 500, 21), (1500, 2500, 21), (60, 2500, 24), (700, 1000, 5)]:
            _no_long(None, False)
            got = _against_oracle(gix, oix, qs, k, m, n)
            nq, general, _glob = gix.last_path_counts()
            assert general <= n_long // 3 + 8, "the LONG instantiation should serve most sessions of 11..20 items (%d such sessions, %d queries reached the general kernel; k %d m %d n %d, mid %d big %d)" % (n_long, general, k, m, n, gix.last_mid_count(), gix.last_big_count())
            _no_long(None, True)
            ref = sa.predict_batch(gix, qs, k, m, n, False)
            assert gix.last_path_counts()[1] >= n_long
            for a, b in zip(got, ref):
                assert np.array_equal(a, b), "with and without the LONG tier the results must be the same bytes"
    finally:
        _no_long(None, False)


def test_negative_and_zero_weights_positions_eleven_to_twenty():
    """Queries built so that negative weights matter: (a) recent items unknown, old ones (positions 11..20) known -- every neighbour has a negative or zero weight, every
    score is <= 0 and the positive scores cannot fill the top n: the general kernel answers (an item of score <= 0 IS returned, mod.rs:143-153); (b) popular recent items
    plus popular OLD items -- neighbours that match only an old item subtract from items the recent ones' neighbours add to; (c) the tenth position (weight 0) between them.
    Business rules on in one round.  Exact against the oracle."""
    import serenade_amd as sa
    O = _oracle()
    off, items, ts, ids = small_dataset(6, n_sessions=30000, n_items=160, max_len=10, tied_timestamps=True)
    gix = sa.VMISIndex.from_sessions(off, items, ts, 2000, 10, 2.0)
    oix = O.OracleIndex(off, items, ts, 2000, 10, 2.0)
    rng = np.random.default_rng(9)
    known = np.unique(items)
    flags = rng.choice(np.array([0, 1, 2, 3, 0xFF], np.uint8), size=len(known), p=[0.05, 0.05, 0.7, 0.15, 0.05])
    gix.set_attributes(known, flags)
    oix.set_attributes(known, flags)
    pop = ids[:30]
    qs = []
    for i in range(450):
        L = int(rng.integers(11, 21))
        unk = lambda c: [int(7 + rng.integers(0, 1000)) for _ in range(c)]
        if i % 3 == 0:      # (a) only the oldest L - 10 positions are known items
            q = [int(x) for x in pop[rng.integers(0, len(pop), size=L - 10)]] + unk(10)
        elif i % 3 == 1:    # (b) known popular items at both ends
            q = [int(x) for x in pop[rng.integers(0, len(pop), size=L - 9)]] + unk(5) + [int(x) for x in ids[rng.integers(0, len(ids), size=4)]]
        else:               # (c) one known item exactly at the tenth position from the end, known old ones behind it
            q = [int(x) for x in pop[rng.integers(0, len(pop), size=L - 10)]] + [int(pop[rng.integers(0, len(pop))])] + unk(9)
        qs.append(q[-L:])
    try:
        _no_long(None, False)
        for (k, m, n, business) in [(100, 500, 21, False), (1500, 2000, 21, True), (40, 2000, 24, False)]:
            ids_, scores, counts = _against_oracle(gix, oix, qs, k, m, n, business)
            assert (scores[0::3][counts[0::3] > 0][:, 0] <= 0).all(), "queries whose known items are all beyond the tenth position score <= 0 everywhere"
            assert (scores < 0).any(), "negative scores must occur (and be returned where nothing positive fills the list)"
    finally:
        _no_long(None, False)


@pytest.mark.gpu
@pytest.mark.parametrize("how_many", [25, 50, 64])
def test_more_than_24_recommendations_stay_on_the_fast_kernels(how_many, monkeypatch):
    """num_items_to_recommend is a free config value (src/config.rs:17, docs/CONFIG.md:23).  Until round 5 a caller asking for more than 24 items sent the whole batch to the
    general kernel (7x slower); the threshold sample now takes the ceil(n / 8)-th largest of every wave, so the fast kernels serve n <= 64.  Against the oracle on
    BASELINE configs[1]'s index (lean shape) and on sessions of up to 10 / 20 items (MID / BIG / LONG), and bit-identical to the general kernel's rows (SRN_FAST_HOW_MANY_MAX=24)."""
    import serenade_amd as sa
    from serenade_amd import capi, synth
    from oracle import oracle as O
    inter, n_items, k, m, idfw = synth.CONFIGS["cfg2"]
    off, items, ts = synth.training_sessions(inter, n_items)
    gix = sa.VMISIndex.from_sessions(off, items, ts, m, 34, idfw, builder="gpu")
    oix = O.OracleIndex(off, items, ts, m, 34, idfw, fast=True)
    # (n = 64: the sample's threshold is loose enough that a third of configs[1]'s queries collect more than the 160 candidates the layout holds -- those take the general kernel)
    for max_items, nq_s, min_fast in ((4, 3000, 0.97 if how_many <= 50 else 0.5), (10, 1500, 0.9 if how_many <= 50 else 0.4), (20, 1500, 0.8 if how_many <= 50 else 0.3)):
        qi, qo = synth.queries(nq_s, n_items, max_items=max_items)
        nq = len(qo) - 1
        got = sa.predict_batch(gix, (qi, qo), k, m, how_many, False)
        n_all, general, _ = gix.last_path_counts()
        assert n_all == nq and general <= (1.0 - min_fast) * nq, (max_items, general, nq)
        ref = oix.predict_batch("canonical", qi, qo, k, m, how_many, False, threads=8)
        assert np.array_equal(got[2], ref["counts"]) and np.array_equal(got[0], ref["ids"]), max_items
        np.testing.assert_allclose(got[1], ref["scores"], rtol=1e-12, atol=0)
        assert (got[2] == how_many).mean() > 0.5                         # most queries do fill the longer list
        monkeypatch.setenv("SRN_FAST_HOW_MANY_MAX", "24"); capi.reload_knobs()
        try:
            old = sa.predict_batch(gix, (qi, qo), k, m, how_many, False)
            assert gix.last_path_counts()[1] == nq
        finally:
            monkeypatch.undo(); capi.reload_knobs()
        for x, y in zip(got, old):
            assert np.array_equal(x, y)
    # a single-session call (srn_predict) with a long list
    q1 = qi[qo[5]:qo[6]]
    one = sa.predict(gix, q1, k, m, how_many, False)
    want = oix.predict_canonical(q1, k, m, how_many)
    assert [i for i, _ in one] == want[0].tolist()

"""GPU parity: the HIP path (through the C ABI) against the canonical CPU oracle on the same inputs.

Bar (BASELINE.json north_star): same ranked item lists, scores within 1e-5.  What is actually asserted
is stronger: item ids and order identical, neighbour (session, numerator) sets identical, per-query
counters identical, and scores equal to 1e-12 relative (they are one f64 multiply + divide of an exact
integer accumulator on both sides).
"""
import os

import numpy as np
import pytest

from helpers import flatten, random_queries, small_dataset

pytestmark = pytest.mark.gpu

SCORE_RTOL = 1e-12   # north_star tolerance is 1e-5; integer-exact accumulation lets us hold 1e-12


def _oracle():
    from oracle import oracle
    return oracle


def _check_batch(gix, oix, queries, k, m, how_many, business=False, check_neighbours=True):
    import serenade_amd as sa
    res = sa.predict_batch_debug(gix, queries, k, m, how_many, business, neighbours=check_neighbours)
    flat, off = flatten(queries)
    ref = oix.predict_batch("canonical", flat, off, k, m, how_many, business, threads=4, want_stats=True)
    assert np.array_equal(res["counts"], ref["counts"]), "result counts differ"
    for q in range(len(queries)):
        n = int(ref["counts"][q])
        assert np.array_equal(res["ids"][q, :n], ref["ids"][q, :n]), (q, queries[q], res["ids"][q, :n], ref["ids"][q, :n])
        np.testing.assert_allclose(res["scores"][q, :n], ref["scores"][q, :n], rtol=SCORE_RTOL, atol=0)
    assert np.array_equal(res["stats"][:, :7].astype(np.uint64), ref["stats"]), "P,C,K,I,D,H,L counters differ"
    # the product call: the debug counters switch the sketch pre-filter off (they need every scored item), this one has it on
    ids, scores, counts = sa.predict_batch(gix, queries, k, m, how_many, business)
    assert np.array_equal(counts, res["counts"]) and np.array_equal(ids, res["ids"]) and np.array_equal(scores, res["scores"]), \
        "filtered (product) path differs from the unfiltered (debug) path"
    if check_neighbours:
        for q in range(len(queries)):
            sid, num, _U = oix.neighbors_canonical(queries[q], k, m)
            kq = int(res["nb_counts"][q])
            got = sorted(zip(res["nb_sessions"][q, :kq].tolist(), res["nb_num"][q, :kq].tolist()))
            assert got == sorted(zip(sid.tolist(), num.tolist())), (q, queries[q])
    return res


def test_kat1_should_train_and_predict():
    """src/vmisknn/mod.rs:229-310: two sessions, query [920005] -> 4 results, 920004 first."""
    import serenade_amd as sa
    off = np.array([0, 3, 7], np.uint64)
    items = np.array([920004, 920005, 920006, 920002, 920003, 920004, 920005], np.uint64)
    ts = np.array([1, 1], np.uint32)
    gix = sa.VMISIndex.from_sessions(off, items, ts, 5, 5, 1.0)
    recs = sa.predict(gix, [920005], 500, 500, 20, False)
    assert len(recs) == 4
    assert recs[0].id == 920004
    assert recs[0].score == pytest.approx(2.2549733432916628, rel=1e-12)
    assert [r.id for r in recs[1:]] == [920002, 920003, 920006]          # 3-way tie, canonical order = id asc
    for r in recs[1:]:
        assert r.score == pytest.approx(1.751319134149782, rel=1e-12)


@pytest.fixture(params=["default", "no_fast", "no_mid", "fast_runs3", "no_masks", "no_hot", "hot64", "no_sketch", "sketch64", "no_merge", "dense"])
def kernel_path(request, monkeypatch):
    """The kernel picks code paths per launch: position-set slots (sessions <= 8 items) vs numerator slots + first-match
    pass; direct-mapped accumulators for popular items vs hash only; sketch pre-filter on / off / tiny; candidate sessions by
    merge tree vs session hash table.  The knobs force each combination."""
    if request.param == "no_masks":
        monkeypatch.setenv("SRN_NO_MASKS", "1")
    elif request.param == "no_hot":
        monkeypatch.setenv("SRN_HOT_SLOTS", "0")
    elif request.param == "hot64":
        monkeypatch.setenv("SRN_HOT_SLOTS", "64")
        monkeypatch.setenv("SRN_NO_MASKS", "1")
    elif request.param == "no_sketch":
        monkeypatch.setenv("SRN_SKETCH_SLOTS", "0")
    elif request.param == "dense":             # three workgroups per CU: the 80-VGPR build + small LDS geometry, second LDS tier behind it
        monkeypatch.setenv("SRN_DENSE", "1")
    elif request.param == "no_merge":          # candidate sessions through the LDS hash table + selects instead of the merge tree
        monkeypatch.setenv("SRN_NO_MERGE", "1")
    elif request.param == "sketch64":          # heavy collisions in the upper-bound words: the filter must stay exact
        monkeypatch.setenv("SRN_SKETCH_SLOTS", "64")
        monkeypatch.setenv("SRN_HOT_SLOTS", "32")
    elif request.param == "fast_runs3":        # the fast kernel's form for > 2^28 sessions: 29 rank bits + 3 list bits, queries with 4 lists handed over
        monkeypatch.setenv("SRN_FAST_RUNS", "3")
    elif request.param == "no_fast":           # the general kernel alone (the fast kernel hands it single queries otherwise)
        monkeypatch.setenv("SRN_NO_FAST", "1")
    elif request.param == "no_mid":            # round 3's launch sequence: no MID instantiation between the lean fast kernel and the general kernel ("default" has it since round 4:
        monkeypatch.setenv("SRN_NO_MID", "1")  # sessions of 5..10 items in these tests' batches go through it), and the latency path on prep + general kernel
        monkeypatch.setenv("SRN_TINY_FAST", "0")
    from serenade_amd import capi
    capi.reload_knobs()                        # the library reads its knobs once; tests switch paths between calls
    yield request.param
    monkeypatch.undo()
    capi.reload_knobs()


@pytest.mark.parametrize("tied", [False, True])
def test_small_random_vs_oracle(tied, kernel_path):
    import serenade_amd as sa
    O = _oracle()
    off, items, ts, ids = small_dataset(11 + tied, n_sessions=3000, n_items=400, tied_timestamps=tied)
    # idf_weighting = 0 switches every item to the un-weighted branch (idf <= 0: score += w * sim, mod.rs:146-152, Q5)
    for (m_index, max_len, idfw) in [(200, 12, 1.0), (40, 8, 2.0), (200, 12, 0.0)]:
        gix = sa.VMISIndex.from_sessions(off, items, ts, m_index, max_len, idfw)
        oix = O.OracleIndex(off, items, ts, m_index, max_len, idfw)
        qs = random_queries(5, ids, 300, max_len=6)
        for (k, m, n) in [(50, 200, 21), (10, 30, 5), (500, 500, 100), (1, 1, 1), (3, 1000, 512)]:
            _check_batch(gix, oix, qs, k, m, n)


def test_long_rows_and_large_cuts(kernel_path):
    """Rows of up to 34 items (the 64-byte row slots keep 14-15 inline, the rest lives in the overflow area and is walked in
    further rounds) and cuts that do not fit the dense phase-2 lists (m = 4000 > what region B holds next to the boundary-bin
    list: the radix selects over the session table take over), with posting lists long enough for both cuts to bite."""
    import serenade_amd as sa
    O = _oracle()
    off, items, ts, ids = small_dataset(21, n_sessions=30000, n_items=90, max_len=34)
    gix = sa.VMISIndex.from_sessions(off, items, ts, 5000, 34, 1.0)
    oix = O.OracleIndex(off, items, ts, 5000, 34, 1.0)
    qs = random_queries(8, ids, 80, max_len=5, unknown_rate=0.02, dup_rate=0.1)
    for (k, m, n) in [(700, 1500, 21), (2500, 4000, 21), (100, 4500, 40)]:
        res = _check_batch(gix, oix, qs, k, m, n, check_neighbours=(k <= 700))
        assert (res["stats"][:, 1] == m).any() and (res["stats"][:, 2] == k).any(), "both cuts should be exercised"
    # the ABI's limits: k = SRN_MAX_K neighbours (the neighbour list alone takes 32 KB of LDS: one workgroup per CU), how_many = 512
    res = _check_batch(gix, oix, qs[:30], 8192, 5000, 512, check_neighbours=False)
    assert (res["stats"][:, 2] > 4000).any()


def test_long_sessions_negative_weights_and_duplicates(kernel_path):
    """L up to 14: linear_score goes to 0 at position 10 and negative beyond (Q3); duplicates (Q1/Q2)."""
    import serenade_amd as sa
    O = _oracle()
    off, items, ts, ids = small_dataset(3, n_sessions=2500, n_items=120, max_len=10)
    gix = sa.VMISIndex.from_sessions(off, items, ts, 300, 10, 1.0)
    oix = O.OracleIndex(off, items, ts, 300, 10, 1.0)
    qs = random_queries(9, ids, 200, max_len=14, unknown_rate=0.1, dup_rate=0.3)
    _check_batch(gix, oix, qs, 100, 300, 50)
    # all-unknown recent items push every first match beyond position 10: weights <= 0 everywhere
    rng = np.random.default_rng(1)
    deep = [[int(x) for x in ids[rng.choice(len(ids), size=3)]] + [7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17] for _ in range(50)]
    res = _check_batch(gix, oix, deep, 100, 300, 512)
    assert (res["scores"] < 0).any() and (res["scores"] == 0).any(), "expected zero and negative scores (Q3)"


def test_very_long_sessions_keep_the_accumulators_exact():
    """Evolving sessions of 100+ items whose neighbours match only OLD positions: |10 * linear_score| reaches 89 at position 99
    (the direct-mapped accumulator's sum field must be sized for that, not for 9)."""
    import serenade_amd as sa
    O = _oracle()
    off, items, ts, ids = small_dataset(17, n_sessions=4000, n_items=60, max_len=6)
    gix = sa.VMISIndex.from_sessions(off, items, ts, 2000, 10, 1.0)
    oix = O.OracleIndex(off, items, ts, 2000, 10, 1.0)
    rng = np.random.default_rng(2)
    qs = []
    for _ in range(12):
        old = [int(x) for x in ids[rng.choice(len(ids), size=6)]]                  # known items, far back in the session
        pad = [int(5 + j) for j in range(int(rng.integers(60, 110)))]              # unknown recent items: first matches at 60..110
        qs.append(old + pad)
    res = _check_batch(gix, oix, qs, 1500, 2000, 40, check_neighbours=False)
    assert (res["scores"] < 0).any(), "expected negative scores (positions beyond 10)"


def test_accumulator_range_guard_at_the_abi_limits():
    """The exact accumulators are 32-bit: k = SRN_MAX_K with sessions whose first matches sit near position 90+ could push
    k * |10 * linear_score| * numerator past 2^31.  Such a call is refused with SRN_ERANGE instead of wrapping; the
    same k with sessions the bound allows still matches the oracle."""
    import serenade_amd as sa
    from serenade_amd import capi
    O = _oracle()
    off, items, ts, ids = small_dataset(19, n_sessions=3000, n_items=60, max_len=6)
    gix = sa.VMISIndex.from_sessions(off, items, ts, 3000, 10, 1.0)
    oix = O.OracleIndex(off, items, ts, 3000, 10, 1.0)
    rng = np.random.default_rng(3)
    long_q = [int(x) for x in ids[rng.choice(len(ids), size=6)]] + [int(5 + j) for j in range(249)]     # 255 items: the ABI limit
    with pytest.raises(sa.SerenadeError) as e:
        sa.predict_batch(gix, [long_q], capi.MAX_K, 3000, 21, False)
    assert e.value.code == capi.SRN_ERANGE
    with pytest.raises(sa.SerenadeError) as e:
        sa.predict(gix, long_q, capi.MAX_K, 3000, 21, False)
    assert e.value.code == capi.SRN_ERANGE
    # k * 1.11 M < 2^31 for k = 1900 at L = 255: allowed, and exact
    _check_batch(gix, oix, [long_q, long_q[3:200]], 1900, 3000, 40, check_neighbours=False)
    # k = 8192 with sessions of <= 60 items (8192 * 50 * 11 * 12 / 2 ... well inside): allowed, and exact
    qs = [[int(x) for x in ids[rng.choice(len(ids), size=5)]] + [int(5 + j) for j in range(int(rng.integers(20, 55)))] for _ in range(6)]
    _check_batch(gix, oix, qs, capi.MAX_K, 3000, 40, check_neighbours=False)


def test_unknown_items_and_single_predict():
    import serenade_amd as sa
    off, items, ts, ids = small_dataset(5)
    gix = sa.VMISIndex.from_sessions(off, items, ts, 100, 12, 1.0)
    assert sa.predict(gix, [1, 2, 3], 10, 10, 5, False) == []                     # unknown items -> empty, no error
    with pytest.raises(sa.SerenadeError) as e:
        sa.predict(gix, [], 10, 10, 5, False)                                     # reference panics (mod.rs:157)
    assert e.value.code == -1
    with pytest.raises(sa.SerenadeError):
        sa.predict(gix, [int(ids[0])], 10, 10, 513, False)                        # SRN_ERANGE
    with pytest.raises(sa.SerenadeError):
        sa.predict(gix, [int(ids[0])], 0, 10, 5, False)


def test_business_rules():
    import serenade_amd as sa
    O = _oracle()
    off, items, ts, ids = small_dataset(21, n_sessions=2000, n_items=200)
    gix = sa.VMISIndex.from_sessions(off, items, ts, 200, 12, 1.0)
    oix = O.OracleIndex(off, items, ts, 200, 12, 1.0)
    rng = np.random.default_rng(4)
    known = np.unique(items)
    flags = rng.choice(np.array([0, 1, 2, 3, 0xFF], np.uint8), size=len(known), p=[0.1, 0.05, 0.55, 0.2, 0.1])
    gix.set_attributes(known, flags)
    oix.set_attributes(known, flags)
    qs = random_queries(6, ids, 300, max_len=4, unknown_rate=0.0)
    _check_batch(gix, oix, qs, 60, 200, 21, business=True, check_neighbours=False)
    _check_batch(gix, oix, qs, 60, 200, 21, business=False, check_neighbours=False)


def test_example_golden_fixture():
    """tests/golden/example_golden.npz: the reference's own example data (assets/example), 931 evaluator
    queries, expected outputs produced by the pinned oracle (tests/golden/make_golden.py)."""
    import os
    import serenade_amd as sa
    from helpers import GOLDEN
    path = os.path.join(GOLDEN, "example_golden.npz")
    g = np.load(path)
    for tag in ("a", "b"):
        m, k, n, idfw, max_len = (int(x) for x in g["params_" + tag])
        gix = sa.VMISIndex.from_sessions(g["sess_off"], g["items"], g["ts"], m, max_len, float(idfw))
        ids, sc, cnt = sa.predict_batch(gix, sa.CSR(g["q_items_" + tag], g["q_off_" + tag]), k, m, n, True)
        assert np.array_equal(cnt, g["counts_" + tag])
        assert np.array_equal(ids, g["ids_" + tag])
        np.testing.assert_allclose(sc, g["scores_" + tag], rtol=SCORE_RTOL, atol=0)


def test_retry_path_with_global_tables():
    """Queries whose candidate set cannot fit the LDS session table go through the global-table pass."""
    import serenade_amd as sa
    O = _oracle()
    rng = np.random.default_rng(8)
    n_items, n_sessions = 60, 60000
    ids = (np.arange(n_items, dtype=np.uint64) + 1) * 1000
    off, items = [0], []
    for s in range(n_sessions):                      # short sessions over few items -> long, mostly disjoint lists
        row = np.unique(ids[rng.choice(n_items, size=int(rng.integers(1, 3)))])
        items.extend(row.tolist()); off.append(len(items))
    ts = (1 + rng.permutation(n_sessions)).astype(np.uint32)
    off, items = np.array(off, np.uint64), np.array(items, np.uint64)
    gix = sa.VMISIndex.from_sessions(off, items, ts, 2000, 5, 1.0)
    oix = O.OracleIndex(off, items, ts, 2000, 5, 1.0, fast=True)
    qs = [ids[rng.permutation(n_items)[:40]].tolist() for _ in range(6)] + [[int(ids[0])], [int(ids[1]), int(ids[2])]]
    res = _check_batch(gix, oix, qs, 1500, 40000, 21, check_neighbours=False)
    assert (res["stats"][:6, 7] == 1).all(), "long queries should have been served by the global-table pass"
    assert (res["stats"][6:, 7] == 0).all()


def test_global_table_pass_with_position_sets_and_sketch_filter():
    """The global-table pass in its production form: sessions of <= 8 items (position-set slots), the sketch pre-filter ON (no debug
    outputs) and most scored items outside the direct-mapped range.  Heavy items with lists of ~30 K sessions make L * m overflow the LDS
    session table; the pass must still feed the sketch words (found on config 5: it added 0 there and lost every non-popular item)."""
    import serenade_amd as sa
    O = _oracle()
    rng = np.random.default_rng(21)
    n_heavy, n_tail, n_sessions = 10, 30000, 200000
    heavy = (np.arange(n_heavy, dtype=np.uint64) + 1) * 7
    tail = (np.arange(n_tail, dtype=np.uint64) + 1) * 1000 + 3
    off, items = [0], []
    for s in range(n_sessions):
        row = np.unique(np.concatenate([heavy[rng.choice(n_heavy, size=int(rng.integers(1, 3)))], tail[rng.choice(n_tail, size=int(rng.integers(2, 6)))]]))
        items.extend(row.tolist()); off.append(len(items))
    ts = (1 + rng.permutation(n_sessions)).astype(np.uint32)
    off, items = np.array(off, np.uint64), np.array(items, np.uint64)
    m, k, n = 40000, 1500, 21
    gix = sa.VMISIndex.from_sessions(off, items, ts, m, 12, 1.0)
    oix = O.OracleIndex(off, items, ts, m, 12, 1.0, fast=True)
    qs = [heavy[rng.permutation(n_heavy)[:int(rng.integers(4, 9))]].tolist() for _ in range(12)] + [[int(heavy[0])], [int(tail[5]), int(heavy[2])]]
    flat, qoff = flatten(qs)
    ids, sc, cnt = sa.predict_batch(gix, (flat, qoff), k, m, n)            # 8 items at most: position sets; no debug outputs: the filter is on
    nq, general, global_pass = gix.last_path_counts()
    assert global_pass >= 8, "the heavy queries should have been served by the global-table pass"
    ref = oix.predict_batch("canonical", flat, qoff, k, m, n, threads=4)
    assert np.array_equal(cnt, ref["counts"])
    assert np.array_equal(ids, ref["ids"])
    np.testing.assert_allclose(sc, ref["scores"], rtol=SCORE_RTOL, atol=0)
    assert (np.isin(ids[:12], tail).sum(axis=1) > 0).all(), "the case needs non-popular items among the results"


def test_synthetic_tiny_config_matches_oracle(kernel_path):
    """The bench generator's `tiny` config end to end (u64 hashed ids, unique timestamps, k/m cuts hit)."""
    import serenade_amd as sa
    from serenade_amd import synth
    O = _oracle()
    inter, n_items, k, m, idfw = synth.CONFIGS["tiny"]
    off, items, ts = synth.training_sessions(inter, n_items)
    gix = sa.VMISIndex.from_sessions(off, items, ts, m, 34, idfw)
    oix = O.OracleIndex(off, items, ts, m, 34, idfw, fast=True)
    qi, qo = synth.queries(600, n_items)
    nq = len(qo) - 1
    qs = [qi[qo[i]:qo[i + 1]].tolist() for i in range(nq)]
    res = _check_batch(gix, oix, qs, k, m, synth.HOW_MANY, check_neighbours=False)
    st = res["stats"]
    assert (st[:, 1] == m).any() and (st[:, 2] == k).any(), "workload should hit both the m-cut and the k-cut"


def test_business_rules_synthetic_tiny(kernel_path):
    """Business rules (mod.rs:162-182) on the generator's tiny config: 20 K items, so that scored items lie outside the direct-mapped
    range as well -- the fast kernel applies the rules to its threshold sample, to the floor survivors and to the elements it lists."""
    import serenade_amd as sa
    from serenade_amd import synth
    O = _oracle()
    inter, n_items, k, m, idfw = synth.CONFIGS["tiny"]
    off, items, ts = synth.training_sessions(inter, n_items)
    gix = sa.VMISIndex.from_sessions(off, items, ts, m, 34, idfw)
    oix = O.OracleIndex(off, items, ts, m, 34, idfw, fast=True)
    rng = np.random.default_rng(44)
    known = np.unique(items)
    flags = rng.choice(np.array([0, 1, 2, 3, 0xFF], np.uint8), size=len(known), p=[0.15, 0.05, 0.5, 0.2, 0.1])
    gix.set_attributes(known, flags)
    oix.set_attributes(known, flags)
    qi, qo = synth.queries(500, n_items)
    nq = len(qo) - 1
    qs = [qi[qo[i]:qo[i + 1]].tolist() for i in range(nq)]
    _check_batch(gix, oix, qs, k, m, synth.HOW_MANY, business=True, check_neighbours=False)
    if kernel_path == "default":
        assert gix.last_path_counts()[1] < nq // 2, "with business rules on, most queries should still be served by the fast kernel"


def test_kernel_timing_is_a_switch_and_the_path_counts_do_not_need_it():
    """srn_kernel_timing: off by default (the per-kernel events idle the launch stream), on for a benchmark; the counters of the call -- queries handed to the
    general kernel, queries through the global-table pass -- arrive in pinned words written by the last finish kernel either way."""
    import serenade_amd as sa
    from serenade_amd import synth, capi
    inter, n_items, k, m, idfw = synth.CONFIGS["tiny"]
    off, items, ts = synth.training_sessions(inter, n_items)
    gix = sa.VMISIndex.from_sessions(off, items, ts, m, 34, idfw)
    flat, qoff = synth.queries(3000, n_items)
    nq = len(qoff) - 1
    a = sa.predict_batch(gix, (flat, qoff), k, m, 21, False)
    with pytest.raises(capi.SerenadeError) as e:
        gix.last_kernel_ms()
    assert e.value.code == capi.SRN_EINVAL
    assert len(gix.kernel_times(8)[0]) == 0
    counts_off = gix.last_path_counts()
    assert counts_off[0] == nq and counts_off[1] < nq // 4
    gix.kernel_timing(True)
    b = sa.predict_batch(gix, (flat, qoff), k, m, 21, False)
    ms, _, _ = gix.last_kernel_ms()
    assert ms > 0
    t_prep, t_fast, t_pred, t_retry = gix.kernel_times_detail(8)
    assert len(t_fast) >= 1 and (t_fast > 0).all() and (t_prep > 0).all()
    assert gix.last_path_counts() == counts_off
    gix.kernel_timing(False)
    c = sa.predict_batch(gix, (flat, qoff), k, m, 21, False)
    assert len(gix.kernel_times(8)[0]) == 0          # only the trailing run of timed calls is reported
    for x in (b, c):
        assert all(np.array_equal(u, v) for u, v in zip(a, x))


def test_device_pointer_entry_point_matches_host_entry_point():
    torch = pytest.importorskip("torch")
    import serenade_amd as sa
    off, items, ts, ids = small_dataset(31, n_sessions=3000, n_items=300)
    gix = sa.VMISIndex.from_sessions(off, items, ts, 200, 12, 1.0)
    qs = random_queries(2, ids, 500, max_len=5)
    flat, qoff = flatten(qs)
    k, m, n = 50, 200, 21
    ids_h, sc_h, cnt_h = sa.predict_batch(gix, (flat, qoff), k, m, n)
    dev = torch.device("cuda:0")
    d_flat = torch.from_numpy(flat.view(np.int64)).to(dev)
    d_off = torch.from_numpy(qoff.view(np.int32)).to(dev)
    d_ids = torch.zeros(len(qs) * n, dtype=torch.int64, device=dev)
    d_sc = torch.zeros(len(qs) * n, dtype=torch.float64, device=dev)
    d_cnt = torch.zeros(len(qs), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream()
    sa.reserve(gix, 4 * len(qs), 5, k, m, n, False, stream.cuda_stream)          # the workspace of this stream is sized up front: the call below allocates nothing
    sa.predict_batch_device(gix, d_flat.data_ptr(), d_off.data_ptr(), len(qs), 5, k, m, n, False,
                            d_ids.data_ptr(), d_sc.data_ptr(), d_cnt.data_ptr(), stream.cuda_stream)
    stream.synchronize()
    cnt_d = d_cnt.cpu().numpy().view(np.uint32)
    assert np.array_equal(cnt_d, cnt_h)
    ids_d = d_ids.cpu().numpy().view(np.uint64).reshape(len(qs), n)
    sc_d = d_sc.cpu().numpy().reshape(len(qs), n)
    for q in range(len(qs)):
        c = int(cnt_h[q])
        assert np.array_equal(ids_d[q, :c], ids_h[q, :c])
        assert np.array_equal(sc_d[q, :c], sc_h[q, :c])


def test_resident_inputs_flag_overlaps_prep_and_changes_nothing():
    """SRN_FLAG_INPUTS_RESIDENT on srn_predict_batch_device: the call's prep kernel runs on the library's own stream beside the previous call's kernels (two sets of prep
    records alternate).  Many calls back to back without a host synchronisation, three different batches in rotation, each into its own result buffers: every call's
    rows equal the flag-less call's."""
    torch = pytest.importorskip("torch")
    import serenade_amd as sa
    from serenade_amd import synth
    inter, n_items, k, m, idfw = synth.CONFIGS["tiny"]
    off, items, ts = synth.training_sessions(inter, n_items)
    gix = sa.VMISIndex.from_sessions(off, items, ts, m, 34, idfw)
    dev = torch.device("cuda:0")
    n = synth.HOW_MANY
    batches = []
    for b in range(3):
        qi, qo = synth.queries(2500, n_items, seed=synth.SEED + 101 * (b + 1))
        nq = len(qo) - 1
        ref = sa.predict_batch(gix, (qi, qo), k, m, n)
        batches.append((torch.from_numpy(qi.view(np.int64).copy()).to(dev), torch.from_numpy(qo.view(np.int32).copy()).to(dev), nq, ref))
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream()
    outs = []
    for rep in range(9):
        d_flat, d_off, nq, _ = batches[rep % 3]
        o = (torch.zeros(nq * n, dtype=torch.int64, device=dev), torch.zeros(nq * n, dtype=torch.float64, device=dev), torch.zeros(nq, dtype=torch.int32, device=dev))
        sa.predict_batch_device(gix, d_flat.data_ptr(), d_off.data_ptr(), nq, synth.LAST_ITEMS, k, m, n, False, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(),
                                stream.cuda_stream, resident=True)
        outs.append(o)
    stream.synchronize()
    for rep, o in enumerate(outs):
        nq, ref = batches[rep % 3][2], batches[rep % 3][3]
        assert np.array_equal(o[2].cpu().numpy().view(np.uint32), ref[2]), rep
        assert np.array_equal(o[0].cpu().numpy().view(np.uint64).reshape(nq, n), ref[0]) and np.array_equal(o[1].cpu().numpy().reshape(nq, n), ref[1]), rep


def test_resident_and_plain_calls_mixed_on_one_stream():
    """a resident call's side-stream prep must wait for the LAST call that read its record set, resident or not.  Calls with and without
    SRN_FLAG_INPUTS_RESIDENT alternate on one stream in every pattern of three, larger batches so that a kernel is still running when the next call is enqueued."""
    torch = pytest.importorskip("torch")
    import serenade_amd as sa
    from serenade_amd import synth
    inter, n_items, k, m, idfw = synth.CONFIGS["tiny"]
    off, items, ts = synth.training_sessions(inter, n_items)
    gix = sa.VMISIndex.from_sessions(off, items, ts, m, 34, idfw)
    dev = torch.device("cuda:0")
    n = synth.HOW_MANY
    batches = []
    for b in range(3):
        qi, qo = synth.queries(30000 + 5000 * b, n_items, seed=synth.SEED + 77 * (b + 1))
        nq = len(qo) - 1
        ref = sa.predict_batch(gix, (qi, qo), k, m, n)
        batches.append((torch.from_numpy(qi.view(np.int64).copy()).to(dev), torch.from_numpy(qo.view(np.int32).copy()).to(dev), nq, ref))
    torch.cuda.synchronize()
    stream = torch.cuda.Stream()
    for pattern in ((0, 1, 1), (1, 0, 1), (1, 1, 0), (0, 1, 0), (1, 0, 0)):
        outs = []
        with torch.cuda.stream(stream):
            for rep in range(12):
                d_flat, d_off, nq, _ = batches[rep % 3]
                o = (torch.zeros(nq * n, dtype=torch.int64, device=dev), torch.zeros(nq * n, dtype=torch.float64, device=dev), torch.zeros(nq, dtype=torch.int32, device=dev))
                stream.synchronize() if rep == 0 else None
                sa.predict_batch_device(gix, d_flat.data_ptr(), d_off.data_ptr(), nq, synth.LAST_ITEMS, k, m, n, False, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(),
                                        stream.cuda_stream, resident=bool(pattern[rep % 3] if rep < 9 else pattern[(rep + 1) % 3]))
                outs.append(o)
        stream.synchronize()
        for rep, o in enumerate(outs):
            nq, ref = batches[rep % 3][2], batches[rep % 3][3]
            assert np.array_equal(o[2].cpu().numpy().view(np.uint32), ref[2]), (pattern, rep)
            assert np.array_equal(o[0].cpu().numpy().view(np.uint64).reshape(nq, n), ref[0]) and np.array_equal(o[1].cpu().numpy().reshape(nq, n), ref[1]), (pattern, rep)


def test_evaluator_binary_on_reference_example(tmp_path):
    """The drop-in `evaluator <config.toml>` host program (mirror of src/bin/evaluator.rs) on the reference's example data,
    rebuilt from the golden fixture: 931 evaluations, HitRate@20 0.6402, the README's metric line (README.md:170-172)."""
    import os
    import subprocess
    from helpers import GOLDEN
    from serenade_amd import build
    g = np.load(os.path.join(GOLDEN, "example_golden.npz"))
    off, items, ts = g["sess_off"].astype(np.int64), g["items"], g["ts"]
    with open(tmp_path / "train.txt", "w") as f:
        f.write("SessionId\tItemId\tTime\n")
        for s in range(len(ts)):
            for it in items[off[s]:off[s + 1]]:
                f.write("%d\t%d\t%d.0\n" % (s + 1, it, ts[s]))
        f.write("%d\t1\t1.0\n" % (len(ts) + 1))          # the loader never adds the final row (vmis_index.rs:669)
    with open(tmp_path / "test.txt", "w") as f:
        f.write("SessionId\tItemId\tTime\n")
        for s, it, t in g["test_rows"]:
            f.write("%d\t%d\t%d.0\n" % (s, it, t))
    with open(tmp_path / "example.toml", "w") as f:
        f.write('config_type = "toml"\n[data]\ntraining_data_path="train.txt"\n[model]\nm_most_recent_sessions = 500\n'
                'neighborhood_size_k = 50\nmax_items_in_session = 2\nnum_items_to_recommend = 20\nidf_weighting = 1\n'
                '[logic]\nenable_business_logic = true\n[hyperparam]\ntest_data_path = "test.txt"\n')
    exe = build.build_evaluator()
    for extra in ([], ["--per-call"]):
        r = subprocess.run([exe, "example.toml"] + extra, cwd=tmp_path, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        lines = r.stdout.splitlines()
        assert "Qty test evaluations: 931" in lines
        hdr = lines.index("qty_evaluations,Mrr@20,Ndcg@20,HitRate@20,Popularity@20,Precision@20,Coverage@20,Recall@20,F1score@20")
        vals = [float(x) for x in lines[hdr + 1].split(",")]
        assert vals[0] == 931 and vals[3] == 0.6402
        for got, want, tol in zip(vals[1:], [0.3277, 0.3553, 0.6402, 0.0499, 0.0680, 0.2765, 0.4456, 0.1180],
                                  [0.005, 0.005, 0.0001, 0.002, 0.0005, 0.005, 0.003, 0.001]):
            assert abs(got - want) <= tol + 1e-9, vals
    assert any(l.startswith("p90 (microseconds)") for l in lines)


@pytest.mark.parametrize("config,n_check", [("cfg2", 6000), ("cfg3", 3000)])
def test_baseline_configs_full_size(config, n_check, monkeypatch):
    """BASELINE.json configs[1] (2 M interactions / 100 K items, k=500 m=1000) and configs[2] (60 M / 1.76 M, k=1500 m=2500,
    idf_weighting=2) at FULL size: a sample of the evaluator-style query stream against the canonical oracle (ids, order,
    counters exact; scores 1e-12), and -- size-independent property -- the two independent first-match implementations
    (position-set slots vs numerator slots + row pass) must agree bit for bit on a much larger sample."""
    import serenade_amd as sa
    from serenade_amd import synth
    O = _oracle()
    inter, n_items, k, m, idfw = synth.CONFIGS[config]
    off, items, ts = synth.training_sessions(inter, n_items)
    gix = sa.VMISIndex.from_sessions(off, items, ts, m, 34, idfw, builder="gpu")   # the index bench.py times: built on the GPU
    qi, qo = synth.queries(12000, n_items)
    nq = len(qo) - 1
    oix = O.OracleIndex(off, items, ts, m, 34, idfw, fast=True)
    res = sa.predict_batch_debug(gix, (qi[:qo[n_check]], qo[:n_check + 1]), k, m, synth.HOW_MANY, neighbours=False)
    ref = oix.predict_batch("canonical", qi[:qo[n_check]], qo[:n_check + 1], k, m, synth.HOW_MANY, threads=16, want_stats=True)
    assert np.array_equal(res["counts"], ref["counts"])
    assert np.array_equal(res["ids"], ref["ids"])
    np.testing.assert_allclose(res["scores"], ref["scores"], rtol=SCORE_RTOL, atol=0)
    assert np.array_equal(res["stats"][:, :7].astype(np.uint64), ref["stats"])
    assert (res["stats"][:, 1] == m).any() and (res["stats"][:, 2] == k).any()      # both cuts are exercised at this size
    a = sa.predict_batch(gix, (qi, qo), k, m, synth.HOW_MANY)     # product path: sketch pre-filter on
    assert np.array_equal(a[0][:n_check], res["ids"]) and np.array_equal(a[1][:n_check], res["scores"])
    from serenade_amd import capi
    try:
        monkeypatch.setenv("SRN_NO_FAST", "1")                     # the general kernel alone (merge tree, position sets, sketch filter)
        capi.reload_knobs()
        g = sa.predict_batch(gix, (qi, qo), k, m, synth.HOW_MANY)
        for x, y in zip(a, g):
            assert np.array_equal(x, y)
        monkeypatch.setenv("SRN_NO_MERGE", "1")                    # session hash table + selects instead of the merge tree
        capi.reload_knobs()
        c = sa.predict_batch(gix, (qi, qo), k, m, synth.HOW_MANY)
        for x, y in zip(a, c):
            assert np.array_equal(x, y)
        monkeypatch.setenv("SRN_NO_MASKS", "1")                    # no direct-mapped part => no threshold => nothing filtered
        monkeypatch.setenv("SRN_HOT_SLOTS", "0")
        capi.reload_knobs()
        b = sa.predict_batch(gix, (qi, qo), k, m, synth.HOW_MANY)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    finally:
        monkeypatch.undo()
        capi.reload_knobs()
    assert nq > 30000
    # Round 5: the launch the bench times is BIG -- 2^18 resident queries in one call here, served in the order of their most popular items (>= 131 072
    # queries) -- and the checked rows are drawn UNIFORMLY over it (seeded permutation + the launch's first and last 32 queries) from what that launch wrote: a defect
    # that depends on where a query sits in a large launch (a hand-off list that overflows late, the last workgroups, the serving order's last chunk) must not pass
    torch = pytest.importorskip("torch")
    B = 1 << 18
    bi, bo = synth.queries(int(B / 3.0) + 4096, n_items, seed=synth.SEED + 7919, max_items=synth.LAST_ITEMS)
    bo = bo[:B + 1]; bi = bi[:bo[-1]]
    dev = torch.device("cuda:0"); n = synth.HOW_MANY
    d_flat = torch.from_numpy(bi.view(np.int64).copy()).to(dev); d_off = torch.from_numpy(bo.view(np.int32).copy()).to(dev)
    o_ids = torch.zeros(B * n, dtype=torch.int64, device=dev); o_sc = torch.zeros(B * n, dtype=torch.float64, device=dev); o_cnt = torch.zeros(B, dtype=torch.int32, device=dev)
    sa.predict_batch_device(gix, d_flat.data_ptr(), d_off.data_ptr(), B, synth.LAST_ITEMS, k, m, n, False, o_ids.data_ptr(), o_sc.data_ptr(), o_cnt.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    pos = np.sort(np.unique(np.concatenate([np.arange(32), np.arange(B - 32, B), np.random.default_rng(0x5E4E4ADE).permutation(B)[:n_check]]))).astype(np.int64)
    bo64 = bo.astype(np.int64); lens = bo64[pos + 1] - bo64[pos]
    sub_off = np.zeros(len(pos) + 1, np.uint32); sub_off[1:] = np.cumsum(lens)
    take = np.repeat(bo64[pos] - sub_off[:-1].astype(np.int64), lens) + np.arange(int(sub_off[-1]), dtype=np.int64)
    ref2 = oix.predict_batch("canonical", np.ascontiguousarray(bi[take]), sub_off, k, m, n, threads=16)
    g_cnt = o_cnt.cpu().numpy().view(np.uint32)[pos]; g_ids = o_ids.cpu().numpy().view(np.uint64).reshape(B, n)[pos]; g_sc = o_sc.cpu().numpy().reshape(B, n)[pos]
    assert np.array_equal(g_cnt, ref2["counts"]), "uniform sample of the 2^18-query launch: counts differ from the oracle"
    mask = np.arange(n)[None, :] < ref2["counts"][:, None].astype(np.int64)
    assert np.array_equal(g_ids[mask], ref2["ids"][mask]), "uniform sample of the 2^18-query launch: ranked lists differ from the oracle"
    np.testing.assert_allclose(g_sc[mask], ref2["scores"][mask], rtol=SCORE_RTOL, atol=0)


def test_config4_full_size(monkeypatch):
    """BASELINE.json configs[3]: 582 M interactions / 6.5 M items (~10 GB of index in HBM), k=1500 m=2500, index built on the
    GPU.  A 1 000-query sample against the canonical oracle (ids, order, counters exact; scores 1e-12), and the path-equivalence
    properties on > 30 000 queries: fast kernel == general kernel (merge tree) == session hash table + selects.
    Needs ~60 GB of host memory for the generator's sessions and the oracle's index: a box without it FAILS the test (SRN_ALLOW_SKIP_BIG=1 skips on purpose)."""
    import os
    from helpers import big_config_unavailable
    try:
        avail_gb = next(int(l.split()[1]) for l in open("/proc/meminfo") if l.startswith("MemAvailable")) / 1e6
    except Exception:
        avail_gb = float("inf")
    if os.environ.get("SRN_SKIP_CFG4") or avail_gb < 60:
        big_config_unavailable("configs[3] (582 M interactions)", "SRN_SKIP_CFG4 set" if os.environ.get("SRN_SKIP_CFG4") else "%.0f GB of host memory available, ~60 needed" % avail_gb)
    print("covers BASELINE.json configs[3]: synthetic 582M interactions / 6.5M items, 1 GPU, k=1500")
    import serenade_amd as sa
    from serenade_amd import capi, synth
    O = _oracle()
    inter, n_items, k, m, idfw = synth.CONFIGS["cfg4"]
    off, items, ts = synth.training_sessions(inter, n_items)
    gix = sa.VMISIndex.from_sessions(off, items, ts, m, 34, idfw, builder="gpu")
    qi, qo = synth.queries(12000, n_items)
    nq = len(qo) - 1
    assert nq > 30000
    a = sa.predict_batch(gix, (qi, qo), k, m, synth.HOW_MANY)
    assert gix.last_path_counts()[1] < nq // 4, "most queries should have been served by the fast kernel"
    try:
        monkeypatch.setenv("SRN_NO_FAST", "1"); capi.reload_knobs()
        b = sa.predict_batch(gix, (qi, qo), k, m, synth.HOW_MANY)
        monkeypatch.setenv("SRN_NO_MERGE", "1"); capi.reload_knobs()
        c = sa.predict_batch(gix, (qi, qo), k, m, synth.HOW_MANY)
    finally:
        monkeypatch.undo(); capi.reload_knobs()
    for x, y, z in zip(a, b, c):
        assert np.array_equal(x, y) and np.array_equal(x, z)
    n_check = 1000
    res = sa.predict_batch_debug(gix, (qi[:qo[n_check]], qo[:n_check + 1]), k, m, synth.HOW_MANY, neighbours=False)
    assert np.array_equal(a[0][:n_check], res["ids"]) and np.array_equal(a[1][:n_check], res["scores"])
    del gix
    oix = O.OracleIndex(off, items, ts, m, 34, idfw, fast=True)
    ref = oix.predict_batch("canonical", qi[:qo[n_check]], qo[:n_check + 1], k, m, synth.HOW_MANY, threads=32, want_stats=True)
    assert np.array_equal(res["counts"], ref["counts"])
    assert np.array_equal(res["ids"], ref["ids"])
    np.testing.assert_allclose(res["scores"], ref["scores"], rtol=SCORE_RTOL, atol=0)
    assert np.array_equal(res["stats"][:, :7].astype(np.uint64), ref["stats"])


def test_concurrent_host_threads_share_one_index():
    """The handle is re-entrant like the reference's Arc<VMISIndex> shared by actix workers (src/bin/serving.rs:62-94):
    several host threads call srn_predict / srn_predict_batch on one index at the same time (ctypes releases the GIL)."""
    import threading
    import serenade_amd as sa
    O = _oracle()
    off, items, ts, ids = small_dataset(61, n_sessions=4000, n_items=500)
    gix = sa.VMISIndex.from_sessions(off, items, ts, 200, 12, 1.0)
    oix = O.OracleIndex(off, items, ts, 200, 12, 1.0)
    qs = random_queries(13, ids, 240, max_len=5, unknown_rate=0.02)
    flat, qoff = flatten(qs)
    ref = oix.predict_batch("canonical", flat, qoff, 50, 200, 21, threads=4)
    errors = []

    def worker(t):
        try:
            for rep in range(3):
                if t % 2 == 0:
                    ids_, sc_, cnt_ = sa.predict_batch(gix, (flat, qoff), 50, 200, 21)
                    assert np.array_equal(cnt_, ref["counts"]) and np.array_equal(ids_, ref["ids"])
                else:
                    for q in range(t, len(qs), 8):
                        recs = sa.predict(gix, qs[q], 50, 200, 21, False)
                        n = int(ref["counts"][q])
                        assert [r.id for r in recs] == ref["ids"][q, :n].tolist()
        except Exception as e:  # pragma: no cover
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


def test_sixty_four_host_threads_combine_into_rounds():
    """64 threads calling srn_predict on one handle (the reference's actix workers, src/bin/serving.rs:62-94) fell off a
    cliff in round 2 (64 streams, 64 spinning waits).  Concurrent calls now combine into rounds (srn_combine.cpp): every answer still equals the
    oracle's, and the rounds are fewer than the requests."""
    import threading
    import serenade_amd as sa
    from serenade_amd import capi
    import ctypes as C
    O = _oracle()
    off, items, ts, ids = small_dataset(67, n_sessions=4000, n_items=500)
    gix = sa.VMISIndex.from_sessions(off, items, ts, 200, 12, 1.0)
    oix = O.OracleIndex(off, items, ts, 200, 12, 1.0)
    qs = random_queries(17, ids, 512, max_len=5, unknown_rate=0.02)
    flat, qoff = flatten(qs)
    ref = oix.predict_batch("canonical", flat, qoff, 50, 200, 21, threads=4)
    errors = []

    def worker(t):
        try:
            for rep in range(6):
                for q in range(t, len(qs), 64):
                    recs = sa.predict(gix, qs[q], 50, 200, 21, False)
                    n = int(ref["counts"][q])
                    assert [r.id for r in recs] == ref["ids"][q, :n].tolist(), (q, qs[q])
                    np.testing.assert_allclose([r.score for r in recs], ref["scores"][q, :n], rtol=SCORE_RTOL, atol=0)
        except Exception as e:  # pragma: no cover
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(64)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:3]
    rounds, requests, biggest = C.c_uint64(), C.c_uint64(), C.c_uint64()
    capi.check(capi.lib().srn_predict_stats(gix._h, C.byref(rounds), C.byref(requests), C.byref(biggest)))
    assert requests.value == 6 * len(qs)
    assert rounds.value <= requests.value and biggest.value >= 1
    # an empty session fails alone; a call with other parameters gets its own round
    with pytest.raises(sa.SerenadeError):
        sa.predict(gix, [], 50, 200, 21, False)
    f1, o1 = flatten([qs[0]])
    ref1 = oix.predict_batch("canonical", f1, o1, 10, 30, 5)
    assert [r.id for r in sa.predict(gix, qs[0], 10, 30, 5, False)] == ref1["ids"][0, :int(ref1["counts"][0])].tolist()


def test_host_threads_sharing_one_stream_take_turns():
    """Device-pointer calls reuse the workspace bound to their stream in stream order, which holds only while a call's launches are contiguous in the stream: several host
    threads enqueueing on ONE stream (a host whose stream pool is smaller than its thread pool -- torch hands out 32) used to interleave their launch sequences over the
    same buffers: wrong rows, then a GPU memory fault (tools/fuzz_concurrency.py, 64 threads).  The library serialises them now."""
    import threading
    import torch
    import serenade_amd as sa
    from serenade_amd import synth
    inter, n_items, k, m, idfw = synth.CONFIGS["tiny"]
    off, items, ts = synth.training_sessions(inter, n_items)
    gix = sa.VMISIndex.from_sessions(off, items, ts, m, 34, idfw)
    flat, qoff = synth.queries(40000, n_items)
    nq_all = len(qoff) - 1
    ref = sa.predict_batch(gix, (flat, qoff), k, m, 21, False)
    dev = torch.device("cuda:0")
    d_flat = torch.from_numpy(flat.view(np.int64).copy()).to(dev)
    shared = torch.cuda.Stream(dev)
    bad = []

    def worker(tid):
        rng = np.random.default_rng(900 + tid)
        for _ in range(40):
            size = int(rng.choice([300, 3000, 20000])); lo = int(rng.integers(0, nq_all - size)); hi = lo + size
            d_o = torch.from_numpy((qoff[lo:hi + 1] - qoff[lo]).astype(np.int32)).to(dev)
            r = (torch.zeros(size * 21, dtype=torch.int64, device=dev), torch.zeros(size * 21, dtype=torch.float64, device=dev), torch.zeros(size, dtype=torch.int32, device=dev))
            torch.cuda.synchronize()                    # (the buffers above are complete: what follows runs on the shared stream only)
            sa.predict_batch_device(gix, d_flat[int(qoff[lo]):].data_ptr(), d_o.data_ptr(), size, 8, k, m, 21, False, r[0].data_ptr(), r[1].data_ptr(), r[2].data_ptr(), shared.cuda_stream)
            shared.synchronize()
            if not (np.array_equal(r[0].cpu().numpy().view(np.uint64).reshape(size, 21), ref[0][lo:hi]) and np.array_equal(r[2].cpu().numpy().view(np.uint32), ref[2][lo:hi])):
                bad.append((tid, lo, hi))

    ths = [threading.Thread(target=worker, args=(i,)) for i in range(6)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not bad, bad[:3]


def test_single_chunk_host_batches_download_in_pieces():
    """A host batch of one chunk with >= 1 MB of results comes back in pieces (two, copied by the calling thread, below 4 MB; up to eight of >= 2 MB through the copy
    threads above): the pieces cut across the ids | scores | counts layout of the staging buffer at 4 KB boundaries -- every byte must land where the one-piece path puts it."""
    import torch
    import serenade_amd as sa
    from serenade_amd import synth
    inter, n_items, k, m, idfw = synth.CONFIGS["tiny"]
    off, items, ts = synth.training_sessions(inter, n_items)
    gix = sa.VMISIndex.from_sessions(off, items, ts, m, 34, idfw)
    flat, qoff = synth.queries(60000, n_items)
    dev = torch.device("cuda:0")
    for nq, n in ((3500, 21), (13001, 21), (50000, 21), (50000, 5), (30011, 24)):      # 1.2 MB (2 pieces), 4.4 MB (2 through the pool), 17 MB (8), 4.2 MB, 11.6 MB (5)
        f, o = flat[:qoff[nq]], qoff[:nq + 1]
        got = sa.predict_batch(gix, (f, o), k, m, n, False)
        d_f = torch.from_numpy(f.view(np.int64).copy()).to(dev); d_o = torch.from_numpy(o.view(np.int32).copy()).to(dev)
        r_ids = torch.zeros(nq * n, dtype=torch.int64, device=dev); r_sc = torch.zeros(nq * n, dtype=torch.float64, device=dev); r_cnt = torch.zeros(nq, dtype=torch.int32, device=dev)
        sa.predict_batch_device(gix, d_f.data_ptr(), d_o.data_ptr(), nq, 8, k, m, n, False, r_ids.data_ptr(), r_sc.data_ptr(), r_cnt.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(got[0].reshape(-1), r_ids.cpu().numpy().view(np.uint64)), (nq, n)
        assert np.array_equal(got[1].reshape(-1), r_sc.cpu().numpy()), (nq, n)
        assert np.array_equal(got[2], r_cnt.cpu().numpy().view(np.uint32)), (nq, n)


@pytest.mark.parametrize("chunks", [0, 1, 3, 7])
def test_host_pointer_batches_through_the_chunked_pipeline(chunks, monkeypatch):
    """srn_predict_batch on host buffers: <= 256 sessions take the zero-copy latency path, larger batches are cut into chunks whose uploads,
    kernels and downloads overlap (srn_hostpipe.hip).  Every size -- chunk boundaries that do not divide the batch included -- gives the oracle's
    rows, and the reused-buffer form writes the same bytes."""
    import serenade_amd as sa
    from serenade_amd import capi
    O = _oracle()
    off, items, ts, ids = small_dataset(71, n_sessions=6000, n_items=700)
    gix = sa.VMISIndex.from_sessions(off, items, ts, 300, 12, 1.0)
    oix = O.OracleIndex(off, items, ts, 300, 12, 1.0)
    qs = random_queries(19, ids, 9000, max_len=6, unknown_rate=0.02)
    flat, qoff = flatten(qs)
    ref = oix.predict_batch("canonical", flat, qoff, 60, 300, 21, threads=4)
    try:
        if chunks:
            monkeypatch.setenv("SRN_HOST_CHUNKS", str(chunks))
        capi.reload_knobs()
        out = None
        for nq in (1, 16, 17, 256, 257, 1000, 4097, 9000):
            f, o = flat[:qoff[nq]], qoff[:nq + 1]
            got = sa.predict_batch(gix, (f, o), 60, 300, 21)
            mask = np.arange(21)[None, :] < ref["counts"][:nq, None].astype(np.int64)
            assert np.array_equal(got[2], ref["counts"][:nq]), nq
            assert np.array_equal(got[0][mask], ref["ids"][:nq][mask]), nq
            np.testing.assert_allclose(got[1][mask], ref["scores"][:nq][mask], rtol=SCORE_RTOL, atol=0)
            assert not got[0][~mask].any() and not got[1][~mask].any(), "the unused tail of a row reads as 0"
            if nq == 9000:
                out = sa.predict_batch(gix, (f, o), 60, 300, 21, out=got)          # same buffers again
                assert out[0] is got[0] and np.array_equal(out[0][mask], ref["ids"][:nq][mask])
    finally:
        monkeypatch.undo(); capi.reload_knobs()


def test_trait_accessors_find_neighbors_at_the_abi():
    """srn_find_neighbors -- the non-debug find_neighbors of SimilarityComputationNew (src/vmisknn/similarity_indexed.rs:9-23, vmis_index.rs:325-415) -- against the
    oracle's canonical neighbours: the same (session, numerator / U) pairs, best first (similarity desc, ties: the more recent session -- (timestamp, index) -- first);
    sessions of 1..12 items with repeats and unknown items, both cuts biting, tied timestamps."""
    import serenade_amd as sa
    O = _oracle()
    off, items, ts, ids = small_dataset(7, n_sessions=3000, n_items=120, max_len=12, tied_timestamps=True)
    gix = sa.VMISIndex.from_sessions(off, items, ts, 300, 12, 1.0)
    oix = O.OracleIndex(off, items, ts, 300, 12, 1.0)
    qs = random_queries(77, ids, 60, max_len=12, unknown_rate=0.05, dup_rate=0.15)
    for (k, m) in [(50, 300), (500, 120), (1500, 300)]:
        for q in qs:
            ses, sc = gix.find_neighbors(q, k, m)
            sid, num, U = oix.neighbors_canonical(q, k, m)
            assert len(ses) == len(sid) <= k
            assert sorted(zip(ses.tolist(), sc.tolist())) == sorted(zip(sid.tolist(), (num / float(max(U, 1))).tolist())), q
            key = [(-float(s), -int(ts[i]), -int(i)) for i, s in zip(ses.tolist(), sc.tolist())]
            assert key == sorted(key), "not best-first / most-recent-first at ties"
    with pytest.raises(sa.SerenadeError):
        gix.find_neighbors([], 10, 10)
    assert len(gix.find_neighbors([2 ** 60 + 1], 10, 10)[0]) == 0          # an unknown item alone: no neighbours (vmis_index.rs:350)


def test_serving_order_changes_no_row():
    """Round 5: batches of >= SRN_ORDER_MIN (131 072) queries are SERVED in the order of their most popular item -- keys from the prep kernel, one radix sort, the order
    dealt to the XCDs in chunks of 256 -- so that like queries share an L2.  Which workgroup serves a query must not change its row: with the ordering pass forced on
    (SRN_ORDER_MIN=1) batches of awkward sizes (below, at and above the chunk size and the 8-way deal; the host-pointer path's chunks) equal the oracle and, bit for
    bit, the same batch without the pass; sessions of up to 10 items in the batch (the MID instantiation's list is filled in serving order)."""
    import os
    import serenade_amd as sa
    from serenade_amd import capi
    O = _oracle()
    off, items, ts, ids = small_dataset(17, n_sessions=20000, n_items=1500, max_len=12)
    gix = sa.VMISIndex.from_sessions(off, items, ts, 400, 12, 1.0)
    oix = O.OracleIndex(off, items, ts, 400, 12, 1.0)
    try:
        for nq, max_len in ((257, 4), (2049, 4), (5000, 10)):
            qs = random_queries(300 + nq, ids, nq, max_len=max_len, unknown_rate=0.04, dup_rate=0.1)
            flat, qoff = flatten(qs)
            ref = oix.predict_batch("canonical", flat, qoff, 300, 400, 21, False, threads=4)
            got = {}
            for omin in ("0", "1"):
                os.environ["SRN_ORDER_MIN"] = omin
                capi.reload_knobs()
                got[omin] = sa.predict_batch(gix, (flat, qoff), 300, 400, 21, False)
            for a, b in zip(got["0"], got["1"]):
                assert np.array_equal(a, b), "the serving order changed a row (%d queries)" % nq
            ids_, sc_, cnt_ = got["1"]
            assert np.array_equal(cnt_, ref["counts"])
            mask = np.arange(21)[None, :] < ref["counts"][:, None].astype(np.int64)
            assert np.array_equal(ids_[mask], ref["ids"][mask])
            np.testing.assert_allclose(sc_[mask], ref["scores"][mask], rtol=SCORE_RTOL, atol=0)
    finally:
        os.environ.pop("SRN_ORDER_MIN", None)
        capi.reload_knobs()


@pytest.mark.parametrize("knobs", [{}, {"SRN_TINY_SPIN": "0"}, {"SRN_TINY_FUSED": "0"}, {"SRN_TINY_PHASES": "2", "SRN_TINY_FUSED": "0"}])
def test_single_session_calls_one_launch_against_the_oracle(knobs):
    """srn_predict -- the reference's call shape, one evolving session per call -- is ONE launch since round 5: the fast kernel's TINY instantiation writes the prep record
    itself, serves the query and finishes its row from registers; what it cannot finish (a session for the MID tier, > 63 entries, a shape for the general kernel) runs behind
    it for that call only.  Every call against the canonical oracle, on an index small enough that many queries have no threshold (> 63 entries: the second phase), with sessions
    of 1..10 items, unknown and repeated items, business rules; the same with the caller waiting for the stream, with the five-launch form, and with that form in two phases."""
    import serenade_amd as sa
    from serenade_amd import capi
    from oracle import oracle as O
    os.environ.update(knobs)
    capi.reload_knobs()
    try:
        for (n_sessions, n_items, m_index, k, m) in [(4000, 600, 500, 100, 500), (20000, 150, 2500, 1500, 2500)]:
            off, items, ts, ids = small_dataset(97 + n_items, n_sessions=n_sessions, n_items=n_items, max_len=20)
            gix = sa.VMISIndex.from_sessions(off, items, ts, m_index, 20, 1.0)
            oix = O.OracleIndex(off, items, ts, m_index, 20, 1.0)
            rng = np.random.default_rng(5)
            known = np.unique(items)
            flags = rng.choice(np.array([0, 1, 2, 3, 0xFF], np.uint8), size=len(known), p=[0.1, 0.05, 0.55, 0.2, 0.1])
            gix.set_attributes(known, flags)
            oix.set_attributes(known, flags)
            qs = random_queries(11, ids, 260, max_len=10, unknown_rate=0.05, dup_rate=0.15)
            flat, qoff = flatten(qs)
            for business in (False, True):
                for n in (21, 5):
                    ref = oix.predict_batch("canonical", flat, qoff, k, m, n, business, threads=4)
                    for qi, q in enumerate(qs):
                        recs = sa.predict(gix, q, k, m, n, business)
                        cnt = int(ref["counts"][qi])
                        assert [r.id for r in recs] == ref["ids"][qi, :cnt].tolist(), (knobs, n_items, business, n, qi, q)
                        np.testing.assert_allclose([r.score for r in recs], ref["scores"][qi, :cnt], rtol=1e-12, atol=0)
                    # ... and small host batches (a workgroup per session in the same launch; the last one to finish publishes): 2, 5, 32, 48 sessions per call, and 49 -- the first size beyond the fused launch
                    for nb in (2, 5, 32, 48, 49):
                        for q0 in range(0, len(qs) - nb, 37):
                            ids_, sc_, cnt_ = sa.predict_batch(gix, qs[q0:q0 + nb], k, m, n, business)
                            assert np.array_equal(cnt_, ref["counts"][q0:q0 + nb]), (knobs, nb, q0)
                            for j in range(nb):
                                c = int(cnt_[j])
                                assert np.array_equal(ids_[j, :c], ref["ids"][q0 + j, :c]), (knobs, nb, q0, j)
                                np.testing.assert_allclose(sc_[j, :c], ref["scores"][q0 + j, :c], rtol=1e-12, atol=0)
    finally:
        for kk in knobs:
            os.environ.pop(kk, None)
        capi.reload_knobs()

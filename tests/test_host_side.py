"""Host side of the product without a GPU: the C ABI loads and exports every declared symbol, the TSV loader
and the flat-index builder agree with the oracle's restatement of the reference, error codes are right.
No predict call is made here (that path needs the GPU and has no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import serenade_amd as sa
from serenade_amd import capi, synth
from helpers import ROOT, extract_example, have_reference_assets, small_dataset
from oracle import oracle as O


def test_library_exports_every_symbol_the_header_declares():
    """include/serenade_hip.h = the drop-in boundary; include/serenade_hip_internal.h = this repository's own measurement / test aids (srn_debug_* only).
    Together they declare exactly what the ctypes binding binds, and the library exports all of it; nothing stage-level is left in the boundary."""
    def names(fn):
        text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", fn)).read(), flags=re.S)      # (declarations, not the prose around them)
        out = set(re.findall(r"\b(srn_[a-z0-9_]+)\s*\(", text))
        return out - {n for n in out if n.endswith("_t")}
    public, internal = names("serenade_hip.h"), names("serenade_hip_internal.h")
    assert internal and all(n.startswith("srn_debug_") for n in internal), internal
    assert not [n for n in public if n.startswith("srn_debug_") or n.startswith("srn_shard_stage") or n.startswith("srn_shard_lists")], "scaffolding in the public header"
    declared = public | internal
    assert declared == set(capi.SYMBOLS), (declared ^ set(capi.SYMBOLS))
    L = capi.lib()
    for name in declared:
        assert getattr(L, name) is not None
    assert L.srn_version().startswith(b"serenade_hip")
    lim = capi.Limits()
    L.srn_limits(C.byref(lim))
    assert (lim.max_how_many, lim.max_session_len, lim.max_k) == (capi.MAX_HOW_MANY, capi.MAX_SESSION_LEN, capi.MAX_K)


def _write_tsv(path, rows):
    with open(path, "w") as f:
        f.write("SessionId\tItemId\tTime\n")
        for s, i, t in rows:
            f.write("%d\t%d\t%s\n" % (s, i, t))


def _product_sessions(path):
    h = C.c_void_p()
    capi.check(capi.lib().srn_sessions_from_tsv(str(path).encode(), C.byref(h)))
    v = capi.SessionsView()
    capi.check(capi.lib().srn_sessions_view(h, C.byref(v)))
    n = v.n_sessions
    off = np.ctypeslib.as_array(C.cast(v.sess_off, C.POINTER(C.c_uint64)), (n + 1,)).copy()
    items = np.ctypeslib.as_array(C.cast(v.items, C.POINTER(C.c_uint64)), (int(off[-1]),)).copy()
    ts = np.ctypeslib.as_array(C.cast(v.max_ts, C.POINTER(C.c_uint32)), (n,)).copy()
    q = C.c_uint64()
    capi.check(capi.lib().srn_sessions_length_quantile(h, 0.995, C.byref(q)))
    capi.lib().srn_sessions_free(h)
    return off, items, ts, q.value


def _third_loader(path):
    """A third, deliberately naive restatement of read_from_file (vmis_index.rs:591-686), written row by row from the reference's loop and sharing no code
    shape with the product loader (C++) or the oracle's (C): a check that the two agree because the semantics are right, not because one hand wrote both."""
    rows = []
    with open(path) as f:
        next(f)
        for line in f:
            a, b, c = line.rstrip("\n").split("\t")
            t = float(c)
            rows.append((int(a), int(b), int(np.floor(t + 0.5)) if t >= 0 else -int(np.floor(-t + 0.5))))     # f64::round: half away from zero
    rows.sort(key=lambda r: r[0])                                 # Python's sort is stable, as sort_by_key is
    sessions, stamps = [], []
    cur, cur_max = [rows[0][1]], rows[0][2]
    for i in range(1, len(rows)):
        same = rows[i][0] == rows[i - 1][0]
        if same and i != len(rows) - 1:
            if rows[i][1] not in cur:
                cur.append(rows[i][1])
                cur_max = max(cur_max, rows[i][2])
        else:                                                     # also taken by the file's LAST row, whatever its session: it opens a session nobody closes
            sessions.append(sorted(cur))
            stamps.append(cur_max & 0xFFFFFFFF)
            cur, cur_max = [rows[i][1]], rows[i][2]
    off = np.zeros(len(sessions) + 1, np.uint64)
    off[1:] = np.cumsum([len(s) for s in sessions])
    return off, np.array([x for s in sessions for x in s], np.uint64), np.array(stamps, np.uint32)


def test_tsv_loader_matches_read_from_file_quirks(tmp_path):
    """Q8: unsorted input, stable order inside a session, de-dup keeps first, max-ts only from non-duplicate rows,
    f64 timestamps rounded, last row never added and last session dropped."""
    rng = np.random.default_rng(0)
    rows = []
    for s in rng.permutation(200):
        for j in range(int(rng.integers(1, 9))):
            rows.append((int(s) * 3 + 1, int(rng.integers(1, 40)), "%.1f" % (1.59e9 + float(rng.integers(0, 10**6)) + 0.5 * (j % 2))))
    order = rng.permutation(len(rows))
    rows = [rows[i] for i in order]
    path = tmp_path / "train.tsv"
    _write_tsv(path, rows)
    off, items, ts, q995 = _product_sessions(path)
    o_off, o_items, o_ts, _ = O.read_tsv(str(path))
    assert np.array_equal(off, o_off) and np.array_equal(items, o_items) and np.array_equal(ts, o_ts)
    t_off, t_items, t_ts = _third_loader(path)
    assert np.array_equal(off, t_off) and np.array_equal(items, t_items) and np.array_equal(ts, t_ts)
    # hand-made file: session 9's final row is never added; a trailing single-row session is dropped entirely
    p2 = tmp_path / "tiny.tsv"
    _write_tsv(p2, [(7, 30, "10.4"), (9, 5, "20.0"), (7, 10, "11.6"), (7, 30, "99.0"), (9, 6, "21.0"), (9, 8, "22.0")])
    off2, items2, ts2, _ = _product_sessions(p2)
    assert off2.tolist() == [0, 2, 4] and items2.tolist() == [10, 30, 5, 6]
    assert ts2.tolist() == [12, 21]                         # 11.6 rounds to 12; the duplicate row (t=99) does not move the max
    _write_tsv(p2, [(7, 30, "10.0"), (7, 10, "11.0"), (9, 5, "20.0")])
    off3, items3, ts3, _ = _product_sessions(p2)
    assert off3.tolist() == [0, 2] and items3.tolist() == [10, 30] and ts3.tolist() == [11]
    lens = np.diff(off.astype(np.int64))
    assert q995 == int(round(float(np.quantile(lens, 0.995))))


@pytest.mark.skipif(not have_reference_assets(), reason="/root/reference assets not present (GPU box)")
def test_new_from_csv_on_reference_example(tmp_path):
    d = extract_example(tmp_path)
    path = os.path.join(d, "train.txt")
    off, items, ts, q995 = _product_sessions(path)
    o_off, o_items, o_ts, _ = O.read_tsv(path)
    assert np.array_equal(off, o_off) and np.array_equal(items, o_items) and np.array_equal(ts, o_ts)
    t_off, t_items, t_ts = _third_loader(path)
    assert np.array_equal(off, t_off) and np.array_equal(items, t_items) and np.array_equal(ts, t_ts)
    assert q995 == 15
    ix = sa.VMISIndex.new_from_csv(path, 500, 1.0, device=-1)     # VMISIndex::new_from_csv(path, m, idf_weighting)
    info = ix.info
    assert info["n_sessions_total"] == 23753 and info["max_session_len"] == 15 and info["device"] == -1
    oix = O.OracleIndex(o_off, o_items, o_ts, 500, 15, 1.0)
    assert info["n_items"] == oix.num_items and info["nnz_rows"] == oix.total_pairs
    for it in np.unique(o_items)[::5]:
        a, ia = ix.postings(int(it))
        b, ib = oix.postings(int(it))
        if b is None:
            assert a is None
            continue
        assert np.array_equal(a, b) and ia == ib


@pytest.mark.parametrize("tied", [False, True])
def test_flat_index_matches_prepare_hashmap(tied):
    """Posting order (timestamp desc, session index desc), truncation to m, idf and the p99.5 filter (Q5, Q7, Q9)."""
    off, items, ts, ids = small_dataset(40 + tied, n_sessions=4000, n_items=300, tied_timestamps=tied)
    for (m_index, max_len, idfw) in [(50, 7, 1.0), (1000, 12, 2.0), (1, 3, 0.0)]:
        ix = sa.VMISIndex.from_sessions(off, items, ts, m_index, max_len, idfw, device=-1)
        oix = O.OracleIndex(off, items, ts, m_index, max_len, idfw)
        info = ix.info
        lens = np.diff(off.astype(np.int64))
        assert info["n_sessions_kept"] == int((lens <= max_len).sum())
        assert info["n_items"] == oix.num_items and info["nnz_rows"] == oix.total_pairs
        for it in ids:
            a, ia = ix.postings(int(it))
            b, ib = oix.postings(int(it))
            if b is None:
                assert a is None
                continue
            assert np.array_equal(a, b), it
            assert ia == ib
        assert ix.postings(5)[0] is None


def test_save_load_roundtrip(tmp_path):
    off, items, ts, ids = small_dataset(3)
    ix = sa.VMISIndex.from_sessions(off, items, ts, 40, 12, 1.0, device=-1)
    p = tmp_path / "index.srn"
    ix.save(p)
    ix2 = sa.VMISIndex.load(p, device=-1)
    assert ix.info == ix2.info
    for it in ids[::3]:
        a, ia = ix.postings(int(it))
        b, ib = ix2.postings(int(it))
        assert (a is None and b is None) or (np.array_equal(a, b) and ia == ib)
    good = open(p, "rb").read()
    # a damaged file must come back as SRN_EIO -- never as out-of-range indices on the device or a probe loop that does not end:
    # bad magic, truncation, a length field that exceeds the file, and single corrupted words in every array
    rng = np.random.default_rng(5)
    damaged = [b"garbage!" + good[8:], good[:len(good) // 2], good[:120] + (2 ** 62).to_bytes(8, "little") + good[128:]]
    for _ in range(40):
        at = int(rng.integers(8 + 13 * 8 + 8, len(good) - 4)) & ~3
        damaged.append(good[:at] + b"\xff\xff\xff\x7f" + good[at + 4:])
    rejected = 0
    for blob in damaged:
        with open(p, "wb") as f:
            f.write(blob)
        try:
            ix3 = sa.VMISIndex.load(p, device=-1)
        except sa.SerenadeError as e:
            assert e.code == capi.SRN_EIO
            rejected += 1
            continue
        # what still loads (e.g. a changed idf double or attribute byte) must at least be structurally sound
        assert ix3.info["n_items"] == ix.info["n_items"]
    assert rejected >= 20, rejected


def test_predict_batch_takes_a_tuple_of_two_sessions_as_two_queries():
    """(items_flat, q_off) is recognised by its shape -- two numpy arrays, 32-bit offsets from 0 to len(items_flat) -- not by being
    a 2-tuple: a tuple of two evolving sessions is two queries."""
    from serenade_amd import vmisknn
    flat, off = vmisknn._flatten(([11, 12, 13], [14, 15]))
    assert flat.tolist() == [11, 12, 13, 14, 15] and off.tolist() == [0, 3, 5]
    flat, off = vmisknn._flatten((np.array([11, 12, 13], np.uint64), np.array([14, 15], np.uint64)))
    assert flat.tolist() == [11, 12, 13, 14, 15] and off.tolist() == [0, 3, 5]
    flat, off = vmisknn._flatten((np.array([11, 12, 13, 14, 15], np.uint64), np.array([0, 3, 5], np.uint32)))
    assert flat.tolist() == [11, 12, 13, 14, 15] and off.tolist() == [0, 3, 5]
    # offsets of an integer dtype other than the items' uint64 (np.cumsum returns int64) are offsets, not a second session ...
    for dt in (np.int64, np.uint64, np.int32):
        lens = np.array([3, 2], dt)
        q_off = np.concatenate([np.zeros(1, dt), np.cumsum(lens, dtype=dt)])
        pair = (np.array([11, 12, 13, 14, 15], np.uint64), q_off)
        flat, off = vmisknn._flatten(pair if dt != np.uint64 else vmisknn.CSR(*pair))
        assert flat.tolist() == [11, 12, 13, 14, 15] and off.tolist() == [0, 3, 5] and off.dtype == np.uint32
    # ... but two uint64 arrays are two evolving sessions even if the second looks like offsets (item id 0 is legal): CSR() says otherwise explicitly
    with pytest.raises(ValueError):      # ... a pair that reads both ways is refused, not guessed
        vmisknn._flatten((np.array([5, 7, 9], np.uint64), np.array([0, 3], np.uint64)))
    flat, off = vmisknn._flatten(([5, 7, 9], [0, 3]))
    assert flat.tolist() == [5, 7, 9, 0, 3] and off.tolist() == [0, 3, 5]
    flat, off = vmisknn._flatten(vmisknn.CSR(np.array([5, 7, 9], np.uint64), np.array([0, 3], np.uint64)))
    assert flat.tolist() == [5, 7, 9] and off.tolist() == [0, 3]
    with pytest.raises(ValueError):
        vmisknn._flatten(vmisknn.CSR(np.array([5, 7, 9], np.uint64), np.array([0, 2], np.uint64)))
    # a pair that is not offsets (does not start at 0 / end at len / decreases) stays two sessions
    flat, off = vmisknn._flatten((np.array([11, 12, 13], np.uint64), np.array([0, 2, 1, 3], np.int64)))
    assert off.tolist() == [0, 3, 7]
    # int64 items with int64 np.cumsum offsets (accepted as CSR until round 3) read both ways: refused with a message that names the two ways out --
    # uint32 offsets (the ABI's type: never item ids) or CSR(...) --, and both ways out work
    it64 = np.array([11, 12, 13, 14, 15], np.int64)
    with pytest.raises(ValueError) as e:
        vmisknn._flatten((it64, np.array([0, 3, 5], np.int64)))
    assert "CSR(" in str(e.value) and "uint32" in str(e.value)
    for ok in ((it64, np.array([0, 3, 5], np.uint32)), vmisknn.CSR(it64, np.array([0, 3, 5], np.int64))):
        flat, off = vmisknn._flatten(ok)
        assert flat.tolist() == [11, 12, 13, 14, 15] and flat.dtype == np.uint64 and off.tolist() == [0, 3, 5]
    flat, off = vmisknn._flatten((it64, np.array([14, 15], np.int64)))            # (two int64 sessions that do not read as offsets: two queries)
    assert off.tolist() == [0, 5, 7]


def test_error_codes_without_a_device():
    off, items, ts, ids = small_dataset(4, n_sessions=200, n_items=50)
    ix = sa.VMISIndex.from_sessions(off, items, ts, 10, 12, 1.0, device=-1)
    with pytest.raises(sa.SerenadeError) as e:
        sa.predict(ix, [int(ids[0])], 10, 10, 5, False)          # no device attached, and no CPU fallback
    assert e.value.code == capi.SRN_ENODEV
    with pytest.raises(sa.SerenadeError) as e:
        sa.predict_batch(ix, [[int(ids[0])]], 10, 10, 5)
    assert e.value.code == capi.SRN_ENODEV
    with pytest.raises(sa.SerenadeError) as e:
        sa.VMISIndex.from_sessions(off, items[::-1].copy(), ts, 10, 12, 1.0, device=-1)     # rows not ascending
    assert e.value.code == capi.SRN_EINVAL
    with pytest.raises(sa.SerenadeError) as e:
        sa.VMISIndex.from_sessions(off, items, ts, 0, 12, 1.0, device=-1)                    # m_index = 0
    assert e.value.code == capi.SRN_EINVAL
    with pytest.raises(sa.SerenadeError) as e:
        sa.VMISIndex.new_from_csv("/nonexistent/train.txt", 10, 1.0, device=-1)
    assert e.value.code == capi.SRN_EIO
    if capi.device_count() == 0:
        with pytest.raises(sa.SerenadeError) as e:
            sa.VMISIndex.from_sessions(off, items, ts, 10, 12, 1.0, device=0)                # no GPU here: loud failure
        assert e.value.code == capi.SRN_EHIP


def test_set_attributes_host_only():
    off, items, ts, ids = small_dataset(6, n_sessions=200, n_items=50)
    ix = sa.VMISIndex.from_sessions(off, items, ts, 10, 12, 1.0, device=-1)
    ix.set_attributes(ids[:5], np.array([0, 1, 2, 3, 0xFF], np.uint8))     # accepted; consulted only by GPU predict


def test_synthetic_generator_is_deterministic_and_shaped():
    a = synth.training_sessions(50_000, 5_000)
    b = synth.training_sessions(50_000, 5_000)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    off, items, ts = a
    lens = np.diff(off.astype(np.int64))
    assert lens.min() >= 1 and lens.max() <= 34
    pct = np.percentile(lens, [25, 50, 75, 90, 95, 99])                        # vmis_index.rs:116-126: 2 3 6 10 14 27
    assert pct[0] == 2 and pct[1] in (2, 3) and pct[2] in (5, 6) and pct[3] in (9, 10) and pct[4] in (13, 14) and 24 <= pct[5] <= 28
    assert len(np.unique(ts)) == len(ts)                                       # unique timestamps
    assert int(items.max()) < (1 << 48) and int(items.max()) > (1 << 32)       # u64 id path
    for s in range(0, len(ts), 997):
        row = items[off[s]:off[s + 1]]
        assert np.all(row[1:] > row[:-1])
    qi, qo = synth.queries(500, 5_000)
    ql = np.diff(qo.astype(np.int64))
    assert ql.min() >= 1 and ql.max() <= synth.LAST_ITEMS
    qi2, qo2 = synth.queries(500, 5_000, seed=synth.SEED + 1)
    assert not (len(qi) == len(qi2) and np.array_equal(qi, qi2))


def _evaluator():
    from serenade_amd import build
    return build.build_evaluator()


def test_evaluator_metric_known_answers():
    """serenade_amd/csrc/host/evaluator.cpp restates src/metrics/*.rs; its self-test runs the reference's metric KATs
    (mrr.rs:53-61, ndcg.rs:76-84, precision.rs:64-74, recall.rs:65-75, f1score.rs:51-64, hitrate.rs:55-68)."""
    import subprocess
    r = subprocess.run([_evaluator(), "--metrics-selftest"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count(" ok") == 8 and "MISMATCH" not in r.stdout


@pytest.mark.skipif(not have_reference_assets(), reason="/root/reference assets not present (GPU box)")
def test_evaluate_file_reproduces_readme_metrics(tmp_path):
    """evaluate_file.rs mirror on the oracle's predictions for the reference example (example.toml parameters, how_many=20):
    README.md:170-172 -> Mrr 0.3277 Ndcg 0.3553 HitRate 0.6402 Popularity 0.0499 Precision 0.0680 Coverage 0.2765
    Recall 0.4456 F1 0.1180 over 931 evaluations (rank metrics carry the reference's tie noise)."""
    import subprocess
    from helpers import evaluator_queries, flatten, read_test_data_evolving
    d = extract_example(tmp_path)
    off, items, ts, _ = O.read_tsv(os.path.join(d, "train.txt"))
    ix = O.OracleIndex(off, items, ts, 500, 15, 1.0, fast=True)
    qs = evaluator_queries(read_test_data_evolving(os.path.join(d, "test.txt")), 2)
    flat, qoff = flatten([q for q, _ in qs])
    r = ix.predict_batch("canonical", flat, qoff, 50, 500, 20, business=True, threads=4)
    pred = tmp_path / "pred.txt"
    with open(pred, "w") as f:
        for i, (_, nxt) in enumerate(qs):
            f.write(",".join(str(x) for x in r["ids"][i, :r["counts"][i]]) + ";" + ",".join(str(x) for x in nxt) + "\n")
    out = subprocess.run([_evaluator(), "--evaluate-file", os.path.join(d, "train.txt"), str(pred)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    assert lines[-2] == "qty_evaluations,Mrr@20,Ndcg@20,HitRate@20,Popularity@20,Precision@20,Coverage@20,Recall@20,F1score@20"
    vals = [float(x) for x in lines[-1].split(",")]
    assert vals[0] == 931
    readme = [0.3277, 0.3553, 0.6402, 0.0499, 0.0680, 0.2765, 0.4456, 0.1180]
    tol = [0.005, 0.005, 0.0001, 0.002, 0.0005, 0.005, 0.003, 0.001]
    for got, want, t in zip(vals[1:], readme, tol):
        assert abs(got - want) <= t + 1e-9, (vals, readme)


def test_trait_accessors_items_for_session_and_find_attributes():
    """srn_index_items_for_session / srn_index_find_attributes / idf through srn_index_postings: the accessors of SimilarityComputationNew
    (src/vmisknn/similarity_indexed.rs:9-23, vmis_index.rs:317-323, 417-419) on a host-only index, against the training sessions themselves."""
    off, items, ts, ids = small_dataset(9, n_sessions=400, n_items=60, max_len=20)
    max_len = 9                                                             # some sessions are longer: not kept (vmis_index.rs:452)
    ix = sa.VMISIndex.from_sessions(off, items, ts, 50, max_len, 1.0, device=-1)
    lens = np.diff(off.astype(np.int64))
    assert (lens > max_len).any() and (lens <= max_len).any()
    for s in range(len(ts)):
        row = items[int(off[s]):int(off[s + 1])]
        if lens[s] <= max_len:
            assert np.array_equal(ix.items_for_session(s), row)
        else:
            with pytest.raises(sa.SerenadeError) as e:
                ix.items_for_session(s)
            assert e.value.code == capi.SRN_ERANGE
    with pytest.raises(sa.SerenadeError) as e:
        ix.items_for_session(len(ts))
    assert e.value.code == capi.SRN_EINVAL
    known = np.unique(items[np.repeat(lens <= max_len, lens)])
    assert all(ix.find_attributes(int(i)) == capi.ATTR_FOR_SALE for i in known[:20])        # the CSV path's {for_sale, not adult} (vmis_index.rs:514-517)
    assert ix.find_attributes(2 ** 61 + 5) is None
    ix.set_attributes(known[:3], np.array([capi.ATTR_ADULT | capi.ATTR_FOR_SALE, 0, capi.ATTR_NONE], np.uint8))
    assert [ix.find_attributes(int(i)) for i in known[:3]] == [3, 0, None]
    total_pairs = int(lens[lens <= max_len].sum())
    it = int(known[0])
    n_with = sum(1 for s in range(len(ts)) if lens[s] <= max_len and it in items[int(off[s]):int(off[s + 1])])
    assert ix.idf(it) == pytest.approx(np.log(total_pairs / n_with), rel=1e-15)             # vmis_index.rs:509-512
    with pytest.raises(KeyError):
        ix.idf(2 ** 61 + 5)


def test_the_default_library_can_only_be_the_default_build(monkeypatch):
    """SRN_CFLAGS used to rebuild libserenade_hip.so IN PLACE -- the file that ships to the GPU box -- and the kernels keep "timing only, wrong
    results" macros.  build_hip() now refuses experiment flags outright and any other SRN_CFLAGS unless SRN_CFLAGS_INPLACE=1; variants go to serenade_amd/variants/."""
    from serenade_amd import build as B
    monkeypatch.setenv("SRN_CFLAGS", "-DSRN_FAST_EXP_ONELIST=1")
    with pytest.raises(RuntimeError, match="experiment-only"):
        B.build_hip()
    monkeypatch.setenv("SRN_CFLAGS", "-DSRN_FAST_STOP=3")
    monkeypatch.setenv("SRN_CFLAGS_INPLACE", "1")          # (not even then)
    with pytest.raises(RuntimeError, match="experiment-only"):
        B.build_hip()
    monkeypatch.delenv("SRN_CFLAGS_INPLACE")
    monkeypatch.setenv("SRN_CFLAGS", "-DSRN_MERGE_G=12")
    with pytest.raises(RuntimeError, match="in place"):
        B.build_hip()
    monkeypatch.delenv("SRN_CFLAGS")
    assert B.build_hip() == B.LIB                             # the default build: up to date, nothing to do

"""The reference's pre-built index format (Avro item / session index, vmis_index.rs:85-314) -> flat index.

CPU part: the files written by tests/avro_write.py (same schemas, codecs null and snappy) are parsed and the index contents are
what went in.  GPU part: an index loaded from Avro predicts exactly what the index built from the same sessions predicts when
the Avro files carry that index's own posting lists and idf (i.e. the offline producer and the TSV builder agree)."""
import numpy as np
import pytest

import avro_write as AW
from helpers import random_queries, small_dataset


def _write_index(base, off, items, ts, m_index, idf_of, attrs=None, codec="snappy", files=2, reorder_ties=False, compressor=None):
    """The producer side: per item its m_index most recent sessions (by (time, session index)), idf, flags; per session its row."""
    n = len(ts)
    order = np.lexsort((np.arange(n), ts))            # ascending (time, index) = recency rank
    rank = np.empty(n, np.int64); rank[order] = np.arange(n)
    per_item = {}
    for s in range(n):
        for it in items[off[s]:off[s + 1]].tolist():
            per_item.setdefault(it, []).append(s)
    item_recs = []
    for it, ss in sorted(per_item.items()):
        ss = sorted(ss, key=lambda s: -rank[s])[:m_index]
        if reorder_ties:
            ss = ss[::-1]                                # the loader re-orders by its own recency: any order in the file is fine
        a = attrs.get(it, (True, False)) if attrs else (True, False)
        item_recs.append(AW.enc_item(it, ss, idf_of[it], a[0], a[1]))
    sess_recs = [AW.enc_session(s, items[off[s]:off[s + 1]].tolist(), int(ts[s])) for s in range(n)]
    for part in range(files):
        AW.write_container(f"{base}/itemindex/part-{part}.avro", AW.ITEM_SCHEMA, item_recs[part::files], codec, block_records=97, compressor=compressor)
        AW.write_container(f"{base}/sessionindex/part-{part}.avro", AW.SESSION_SCHEMA, sess_recs[part::files], codec, block_records=211, compressor=compressor)
    return per_item


def _idf_like_builder(off, items, weighting):
    pairs = int(off[-1]); u, c = np.unique(items, return_counts=True)
    return {int(i): float(np.log(pairs / k) * weighting) for i, k in zip(u, c)}


def test_snappy_writer_uses_every_element_kind_and_decodes_independently():
    """The test-side compressor emits literals and copies with 1-, 2- and 4-byte offsets, overlapping ones included, and an
    independent pure-Python decoder of the format description gets the input back -- so the streams the C++ reader is fed are
    valid snappy by a second opinion, not only by its own."""
    rng = np.random.default_rng(3)
    base = bytes(rng.integers(0, 7, size=5000, dtype=np.uint8)) + b"\x00" * 300 + bytes(rng.integers(0, 255, size=70000, dtype=np.uint8))
    data = base + base[100:68000] + b"abcd" * 50          # a far repeat (offset > 65535 -> 4-byte form only), near repeats, runs
    st = {}
    z = AW.snappy_with_copies(data, st)
    assert AW.snappy_decode(z) == data
    assert st.get(1, 0) > 0 and st.get(2, 0) > 0 and st.get(3, 0) > 0 and st.get("overlap", 0) > 0, st
    assert AW.snappy_decode(AW.snappy_literal_only(data)) == data


@pytest.mark.parametrize("codec", ["null", "snappy", "snappy-copies"])
def test_avro_index_contents_on_host(tmp_path, codec):
    import serenade_amd as sa
    off, items, ts, ids = small_dataset(41, n_sessions=900, n_items=120, tied_timestamps=True)
    idf = _idf_like_builder(off, items, 1.0)
    attrs = {int(ids[0]): (False, True), int(ids[1]): (True, True)}
    compressor = AW.snappy_with_copies if codec == "snappy-copies" else None
    codec = "snappy" if codec == "snappy-copies" else codec
    per_item = _write_index(str(tmp_path), off, items, ts, 40, idf, attrs, codec, reorder_ties=True, compressor=compressor)
    ix = sa.VMISIndex.new_from_avro(tmp_path, device=-1)
    ref = sa.VMISIndex.from_sessions(off, items, ts, 40, 10**6, 1.0, device=-1)
    info, rinfo = ix.info, ref.info
    for key in ("n_items", "n_sessions_kept", "nnz_rows", "nnz_postings", "m_index"):
        assert info[key] == rinfo[key], key
    for it in list(per_item)[:60] + [int(ids[0]), int(ids[1])]:
        a, b = ix.postings(it), ref.postings(it)
        assert np.array_equal(a[0], b[0]) and a[1] == pytest.approx(b[1], rel=0, abs=0)      # same sessions, same order, the file's idf
    assert ix.postings(123456789)[0] is None


def test_avro_index_rejects_what_it_cannot_represent(tmp_path):
    import serenade_amd as sa
    off, items, ts, ids = small_dataset(42, n_sessions=300, n_items=40)
    idf = _idf_like_builder(off, items, 1.0)
    with pytest.raises(sa.SerenadeError) as e:
        sa.VMISIndex.new_from_avro(tmp_path / "nothing-here", device=-1)
    assert e.value.code == -5          # SRN_EIO
    n = len(ts)
    item_recs, sess_recs = _records(off, items, ts, idf)
    # malformed containers: a CRC that does not match, a wrong sync marker, a block cut in half
    for how in ("crc", "sync", "truncate"):
        base = tmp_path / how
        AW.write_container(f"{base}/itemindex/a.avro", AW.ITEM_SCHEMA, item_recs, "snappy", block_records=50, compressor=AW.snappy_with_copies, corrupt=how)
        AW.write_container(f"{base}/sessionindex/a.avro", AW.SESSION_SCHEMA, sess_recs, "snappy", block_records=50, compressor=AW.snappy_with_copies)
        with pytest.raises(sa.SerenadeError) as e:
            sa.VMISIndex.new_from_avro(base, device=-1)
        assert e.value.code == -1, how
        assert {"crc": "CRC", "sync": "sync", "truncate": "truncated"}[how] in str(e.value), (how, str(e.value))
    # a session row naming an item without an item-index record (the reference would panic when scoring it)
    AW.write_container(f"{tmp_path}/orphan/itemindex/a.avro", AW.ITEM_SCHEMA, item_recs[1:], "null")
    AW.write_container(f"{tmp_path}/orphan/sessionindex/a.avro", AW.SESSION_SCHEMA, sess_recs, "null")
    with pytest.raises(sa.SerenadeError) as e:
        sa.VMISIndex.new_from_avro(tmp_path / "orphan", device=-1)
    assert e.value.code == -1


def _records(off, items, ts, idf, m_index=10**9, tie_break=-1):
    """Item / session records as a producer with its own tie-break among equal timestamps would write them
    (tie_break = -1: larger session index first, as prepare_hashmap does; +1: smaller first)."""
    n = len(ts)
    per_item = {}
    for s in range(n):
        for it in items[off[s]:off[s + 1]].tolist():
            per_item.setdefault(it, []).append(s)
    item_recs = [AW.enc_item(it, sorted(ss, key=lambda s: (-int(ts[s]), tie_break * s))[:m_index], idf[it], True, False) for it, ss in sorted(per_item.items())]
    sess_recs = [AW.enc_session(s, items[off[s]:off[s + 1]].tolist(), int(ts[s])) for s in range(n)]
    return item_recs, sess_recs


def test_avro_schema_with_reordered_and_extra_fields(tmp_path):
    """avro-rs writes the fields in the order of the producer's struct: the reader takes them by NAME from the writer schema,
    skips what it does not know (here a string and a nullable union) and accepts logical-type wrappers."""
    import serenade_amd as sa
    off, items, ts, ids = small_dataset(44, n_sessions=200, n_items=30)
    idf = _idf_like_builder(off, items, 1.0)
    item_schema = {"type": "record", "name": "ItemIndex", "fields": [
        {"name": "note", "type": "string"},
        {"name": "idf", "type": "double"},
        {"name": "IsAdult", "type": "boolean"},
        {"name": "ItemId", "type": {"type": "long", "logicalType": "whatever"}},
        {"name": "maybe", "type": ["null", "long"]},
        {"name": "ForSale", "type": "boolean"},
        {"name": "session_indices_time_ordered", "type": {"type": "array", "items": "int"}}]}
    n = len(ts)
    per_item = {}
    for s in range(n):
        for it in items[off[s]:off[s + 1]].tolist():
            per_item.setdefault(it, []).append(s)
    recs = []
    for j, (it, ss) in enumerate(sorted(per_item.items())):
        ss = sorted(ss, key=lambda s: (-int(ts[s]), -s))
        recs.append(AW.zz(3) + b"abc" + __import__("struct").pack("<d", idf[it]) + bytes([0]) + AW.zz(it) + (AW.zz(1) + AW.zz(j) if j % 2 else AW.zz(0)) + bytes([1]) + AW.enc_array(ss))
    _, sess_recs = _records(off, items, ts, idf)
    AW.write_container(f"{tmp_path}/itemindex/a.avro", item_schema, recs, "snappy", compressor=AW.snappy_with_copies)
    AW.write_container(f"{tmp_path}/sessionindex/a.avro", AW.SESSION_SCHEMA, sess_recs, "null")
    ix = sa.VMISIndex.new_from_avro(tmp_path, device=-1)
    ref = sa.VMISIndex.from_sessions(off, items, ts, 10**6, 10**6, 1.0, device=-1)
    for it in list(per_item)[:30]:
        a, b = ix.postings(it), ref.postings(it)
        assert np.array_equal(a[0], b[0]) and a[1] == b[1]


def _canonical_over_given_lists(lists, rows, ts, idf, session, k, m, n):
    """DESIGN.md section 1 with the posting lists AS GIVEN (not rebuilt from the rows): what a pre-built index whose lists are not
    most-recent prefixes must still produce -- the reference uses the lists as they are (vmis_index.rs:201-228, 332-391) and tests
    the first match against the full row (mod.rs:133-138)."""
    L = len(session)
    recency = lambda s: (int(ts[s]), s)
    seen, num = set(), {}
    for pos in range(L):
        it = session[L - 1 - pos]
        if it in seen:
            continue
        seen.add(it)
        for s in lists.get(it, [])[:m]:
            num[s] = num.get(s, 0) + (L - pos)
    U = len(seen)
    cand = sorted(num, key=recency, reverse=True)[:m]
    nb = sorted(cand, key=lambda s: (num[s], recency(s)), reverse=True)[:k]
    acc = {}
    rev = session[::-1]
    for s in nb:
        p = next(i + 1 for i, it in enumerate(rev) if it in rows[s])
        w10 = 10 - p if p < 100 else 0
        for it in rows[s]:
            acc[it] = acc.get(it, 0) + w10 * num[s]
    acc.pop(session[-1], None)
    scored = sorted(((-(idf[it] if idf[it] > 0 else 1.0) * a / (10.0 * U), it) for it, a in acc.items()))[:n]
    return [it for _, it in scored], [-x for x, _ in scored]


@pytest.mark.gpu
def test_avro_index_with_a_different_tie_break_uses_the_lists_as_given(tmp_path):
    """Timestamps tie and the producer truncated its lists with ANOTHER tie-break than ours: the lists are not most-recent prefixes
    under (Time, SessionIndex), the load must not fail and the answers are those of the lists as given."""
    import serenade_amd as sa
    off, items, ts, ids = small_dataset(45, n_sessions=600, n_items=40, tied_timestamps=True)
    idf = _idf_like_builder(off, items, 1.0)
    m_index = 12
    item_recs, sess_recs = _records(off, items, ts, idf, m_index=m_index, tie_break=+1)
    AW.write_container(f"{tmp_path}/itemindex/a.avro", AW.ITEM_SCHEMA, item_recs, "snappy", compressor=AW.snappy_with_copies)
    AW.write_container(f"{tmp_path}/sessionindex/a.avro", AW.SESSION_SCHEMA, sess_recs, "snappy")
    gix = sa.VMISIndex.new_from_avro(tmp_path)
    n = len(ts)
    rows = [set(items[off[s]:off[s + 1]].tolist()) for s in range(n)]
    per_item = {}
    for s in range(n):
        for it in items[off[s]:off[s + 1]].tolist():
            per_item.setdefault(it, []).append(s)
    # the loader re-orders every list by (Time, SessionIndex) descending; the SET of sessions is the producer's
    lists = {it: sorted(sorted(ss, key=lambda s: (-int(ts[s]), s))[:m_index], key=lambda s: (int(ts[s]), s), reverse=True) for it, ss in per_item.items()}
    differs = sum(lists[it] != sorted(ss, key=lambda s: (int(ts[s]), s), reverse=True)[:m_index] for it, ss in per_item.items())
    assert differs > 0, "the fixture should contain lists that are not most-recent prefixes under our order"
    qs = random_queries(7, ids, 120, max_len=4, unknown_rate=0.0)
    k, m, nrec = 8, m_index, 10
    got_ids, got_sc, got_cnt = sa.predict_batch(gix, qs, k, m, nrec, False)
    for q, sess in enumerate(qs):
        want_ids, want_sc = _canonical_over_given_lists(lists, rows, ts, idf, sess, k, m, nrec)
        c = int(got_cnt[q])
        assert c == len(want_ids), (q, sess)
        np.testing.assert_allclose(got_sc[q, :c], want_sc, rtol=1e-12, atol=0)
        # equal scores are ordered by public id ascending on both sides (sorted() above: (-score, id))
        assert got_ids[q, :c].tolist() == want_ids, (q, sess)


@pytest.mark.gpu
def test_avro_index_predicts_like_the_sessions_built_index(tmp_path):
    import serenade_amd as sa
    off, items, ts, ids = small_dataset(43, n_sessions=5000, n_items=400, tied_timestamps=True)
    idf = _idf_like_builder(off, items, 2.0)
    attrs = {int(i): (bool(j % 3), bool(j % 5 == 0)) for j, i in enumerate(ids)}
    _write_index(str(tmp_path), off, items, ts, 150, idf, attrs, "snappy")
    a = sa.VMISIndex.new_from_avro(tmp_path)
    b = sa.VMISIndex.from_sessions(off, items, ts, 150, 10**6, 2.0)
    b.set_attributes(list(attrs), [(1 if ad else 0) | (2 if fs else 0) for fs, ad in attrs.values()])
    qs = random_queries(6, ids, 400, max_len=6)
    for business in (False, True):
        ra, rb = sa.predict_batch(a, qs, 80, 150, 21, business), sa.predict_batch(b, qs, 80, 150, 21, business)
        for x, y in zip(ra, rb):
            assert np.array_equal(x, y)

"""The reference's pre-built index format (Avro item / session index, vmis_index.rs:85-314) -> flat index.

CPU part: the files written by tests/avro_write.py (same schemas, codecs null and snappy) are parsed and the index contents are
what went in.  GPU part: an index loaded from Avro predicts exactly what the index built from the same sessions predicts when
the Avro files carry that index's own posting lists and idf (i.e. the offline producer and the TSV builder agree)."""
import numpy as np
import pytest

import avro_write as AW
from helpers import random_queries, small_dataset


def _write_index(base, off, items, ts, m_index, idf_of, attrs=None, codec="snappy", files=2, reorder_ties=False):
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
        AW.write_container(f"{base}/itemindex/part-{part}.avro", AW.ITEM_SCHEMA, item_recs[part::files], codec, block_records=97)
        AW.write_container(f"{base}/sessionindex/part-{part}.avro", AW.SESSION_SCHEMA, sess_recs[part::files], codec, block_records=211)
    return per_item


def _idf_like_builder(off, items, weighting):
    pairs = int(off[-1]); u, c = np.unique(items, return_counts=True)
    return {int(i): float(np.log(pairs / k) * weighting) for i, k in zip(u, c)}


@pytest.mark.parametrize("codec", ["null", "snappy"])
def test_avro_index_contents_on_host(tmp_path, codec):
    import serenade_amd as sa
    off, items, ts, ids = small_dataset(41, n_sessions=900, n_items=120, tied_timestamps=True)
    idf = _idf_like_builder(off, items, 1.0)
    attrs = {int(ids[0]): (False, True), int(ids[1]): (True, True)}
    per_item = _write_index(str(tmp_path), off, items, ts, 40, idf, attrs, codec, reorder_ties=True)
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
    # a session list that is not the item's most recent sessions
    n = len(ts)
    per_item = {}
    for s in range(n):
        for it in items[off[s]:off[s + 1]].tolist():
            per_item.setdefault(it, []).append(s)
    victim = max(per_item, key=lambda i: len(per_item[i]))
    recs = [AW.enc_item(it, (sorted(ss, key=lambda s: ts[s])[:3] if it == victim else sorted(ss, key=lambda s: -int(ts[s]))), idf[it], True, False) for it, ss in per_item.items()]
    AW.write_container(f"{tmp_path}/bad/itemindex/a.avro", AW.ITEM_SCHEMA, recs, "null")
    AW.write_container(f"{tmp_path}/bad/sessionindex/a.avro", AW.SESSION_SCHEMA, [AW.enc_session(s, items[off[s]:off[s + 1]].tolist(), int(ts[s])) for s in range(n)], "null")
    with pytest.raises(sa.SerenadeError) as e:
        sa.VMISIndex.new_from_avro(tmp_path / "bad", device=-1)
    assert e.value.code == -1 and "most recent" in str(e.value)


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

"""GPU index construction (srn_index_build_gpu, rocPRIM sorts) must give exactly the index the host builder gives."""
import filecmp

import numpy as np
import pytest

from helpers import small_dataset

pytestmark = pytest.mark.gpu


def _same_index(tmp_path, off, items, ts, m_index, max_len, idfw):
    import serenade_amd as sa
    a = sa.VMISIndex.from_sessions(off, items, ts, m_index, max_len, idfw, builder="host")
    b = sa.VMISIndex.from_sessions(off, items, ts, m_index, max_len, idfw, builder="gpu")
    ia, ib = a.info, b.info
    ia.pop("device_bytes"), ib.pop("device_bytes")
    assert ia == ib
    pa, pb = tmp_path / "host.srn", tmp_path / "gpu.srn"
    a.save(pa), b.save(pb)
    assert filecmp.cmp(pa, pb, shallow=False), "flat index files differ"      # every array, bit for bit (idf included)
    return a, b


@pytest.mark.parametrize("tied", [False, True])
def test_gpu_builder_equals_host_builder_small(tmp_path, tied):
    off, items, ts, ids = small_dataset(70 + tied, n_sessions=5000, n_items=600, tied_timestamps=tied)
    for (m_index, max_len, idfw) in [(50, 7, 1.0), (1000, 12, 2.0), (1, 3, 0.0)]:
        _same_index(tmp_path, off, items, ts, m_index, max_len, idfw)


def test_gpu_builder_equals_host_builder_synthetic_and_predicts(tmp_path):
    import serenade_amd as sa
    from serenade_amd import synth
    inter, n_items, k, m, idfw = synth.CONFIGS["cfg2"]
    off, items, ts = synth.training_sessions(inter, n_items)
    a, b = _same_index(tmp_path, off, items, ts, m, 34, idfw)
    qi, qo = synth.queries(2000, n_items)
    ra, rb = sa.predict_batch(a, (qi, qo), k, m, 21), sa.predict_batch(b, (qi, qo), k, m, 21)
    for x, y in zip(ra, rb):
        assert np.array_equal(x, y)


def test_gpu_builder_rejects_unsorted_rows():
    import serenade_amd as sa
    off, items, ts, ids = small_dataset(5, n_sessions=100, n_items=50)
    with pytest.raises(sa.SerenadeError) as e:
        sa.VMISIndex.from_sessions(off, items[::-1].copy(), ts, 10, 12, 1.0, builder="gpu")
    assert e.value.code == -1


@pytest.mark.parametrize("tied", [False, True])
def test_gpu_built_index_content_equals_the_oracles_literal_prepare_hashmap(tied):
    """A8 directly on the box that runs the benchmark: the index srn_index_build_gpu makes, item by item, against the ORACLE's literal restatement of
    prepare_hashmap (vmis_index.rs:422-528) -- posting lists in order (timestamp desc, session index desc), cut to m (:497-504), idf (:509-512), the session-length filter (:452) --
    not against the product's other builder."""
    import serenade_amd as sa
    from oracle import oracle as O
    off, items, ts, ids = small_dataset(40 + tied, n_sessions=4000, n_items=300, tied_timestamps=tied)
    for (m_index, max_len, idfw) in [(50, 7, 1.0), (1000, 12, 2.0), (1, 3, 0.0)]:
        ix = sa.VMISIndex.from_sessions(off, items, ts, m_index, max_len, idfw, builder="gpu")
        oix = O.OracleIndex(off, items, ts, m_index, max_len, idfw)          # fast=False: the literal loops
        info = ix.info
        lens = np.diff(off.astype(np.int64))
        assert info["n_sessions_kept"] == int((lens <= max_len).sum())
        assert info["n_items"] == oix.num_items and info["nnz_rows"] == oix.total_pairs
        n_lists = 0
        for it in ids:
            a, ia = ix.postings(int(it))
            b, ib = oix.postings(int(it))
            if b is None:
                assert a is None
                continue
            assert np.array_equal(a, b), it
            assert ia == ib                                                   # the same double
            n_lists += 1
        assert n_lists == info["n_items"] and ix.postings(5)[0] is None
        # and the rows the scoring loop reads (items_for_session, vmis_index.rs:317-319) for the kept sessions
        for s in np.flatnonzero(lens <= max_len)[:200]:
            assert ix.items_for_session(int(s)).tolist() == items[off[s]:off[s + 1]].tolist()

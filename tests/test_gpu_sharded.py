"""Item shards (srn_index_shard & co.) under the shard group: row FRAGMENTS of every length in both slot forms, every kernel path of the launch sequence that runs
over them (fast kernel, general kernel with the merge tree, general kernel with the session hash table), the shard's memory footprint, the three ways to build a
shard.  All predictions go through srn_shard_group_predict_batch (the stage-level entry points and the Python-driven pipelines of rounds 1-2 are gone) and are
compared with the CPU oracle and, bit for bit, with the unsharded HIP path.  The group's pipelines and transports themselves: tests/test_gpu_shard_group.py."""
import numpy as np
import pytest

from helpers import flatten, random_queries, small_dataset

pytestmark = pytest.mark.gpu
SCORE_RTOL = 1e-12


def _to_dev(flat, qoff):
    import torch
    dev = torch.device("cuda:0")
    return torch.from_numpy(flat.view(np.int64).copy()).to(dev), torch.from_numpy(qoff.view(np.int32).copy()).to(dev)


def _np(res):
    import torch
    torch.cuda.synchronize()
    return res[0].cpu().numpy().view(np.uint64), res[1].cpu().numpy(), res[2].cpu().numpy().view(np.uint32)


def _check(got, ref, unsharded):
    ids, sc, cnt = got
    n = ids.shape[1]
    assert np.array_equal(cnt, ref["counts"])
    mask = np.arange(n)[None, :] < ref["counts"][:, None].astype(np.int64)
    assert np.array_equal(ids[mask], ref["ids"][mask])
    np.testing.assert_allclose(sc[mask], ref["scores"][mask], rtol=SCORE_RTOL, atol=0)
    assert np.array_equal(ids, unsharded[0]) and np.array_equal(sc, unsharded[1]) and np.array_equal(cnt, unsharded[2])     # bit-identical: same integers, same f64 operations


def test_row_fragments_of_every_length_and_shard_row_memory():
    """A shard keeps each session's FRAGMENT (the items it owns) in a 16-byte slot, longer fragments continue in an overflow area: rows of up to 80 items over 2
    and 5 shards put fragments of 0..40+ items through every branch of the walks.  The shard's HBM footprint must be well under the unsharded index's 64-byte row
    slots, and a shard refuses plain predictions."""
    import serenade_amd as sa
    from serenade_amd import sharded, capi
    from oracle import oracle as O
    off, items, ts, ids = small_dataset(57, n_sessions=3000, n_items=500, max_len=80)
    qs = random_queries(17, ids, 300, max_len=8)
    flat, qoff = flatten(qs)
    d_flat, d_off = _to_dev(flat, qoff)
    full = sa.VMISIndex.from_sessions(off, items, ts, 300, 80, 1.0)
    oix = O.OracleIndex(off, items, ts, 300, 80, 1.0)
    for n_shards in (2, 5):
        shards = [sharded.ShardedVMISIndex(off, items, ts, 300, 80, 1.0, g, n_shards) for g in range(n_shards)]       # (built on the GPU, cut per shard)
        lens = np.diff(off.astype(np.int64))
        assert sum(s.info["n_items"] for s in shards) == len(np.unique(items[np.repeat(lens <= 80, lens)]))
        grp = sharded.ShardGroup.local(shards)
        for (k, m, n) in [(100, 300, 21), (500, 500, 50)]:
            ref = oix.predict_batch("canonical", flat, qoff, k, m, n, False, threads=4)
            _check(_np(grp.predict_batch(d_flat, d_off, len(qs), 8, k, m, n)), ref, sa.predict_batch(full, (flat, qoff), k, m, n, False))
        assert max(s.info["device_bytes"] for s in shards) < full.info["device_bytes"] * (0.7 if n_shards == 2 else 0.45), (n_shards, [s.info["device_bytes"] for s in shards], full.info["device_bytes"])
        with pytest.raises(capi.SerenadeError):
            out = np.zeros(21, np.uint64); sc = np.zeros(21); cnt = np.zeros(1, np.uint32)
            capi.check(capi.lib().srn_predict_batch(shards[0]._h, flat.ctypes.data, qoff.ctypes.data, 1, 100, 300, 21, 0,
                                                    out.ctypes.data, sc.ctypes.data, cnt.ctypes.data))
        grp.close()


@pytest.fixture(params=["fast", "no_fast", "no_merge"])
def lists_kernel_path(request, monkeypatch):
    """The lists / neighbours pipelines run the unsharded launch sequence over the fragments: through the fast kernel (default), the general kernel alone, and the
    general kernel with the session hash table instead of the merge tree."""
    from serenade_amd import capi
    if request.param == "no_fast":
        monkeypatch.setenv("SRN_NO_FAST", "1")
    elif request.param == "no_merge":
        monkeypatch.setenv("SRN_NO_FAST", "1")
        monkeypatch.setenv("SRN_NO_MERGE", "1")
    capi.reload_knobs()
    yield request.param
    monkeypatch.undo()
    capi.reload_knobs()


@pytest.mark.parametrize("n_shards", [1, 2, 5])
def test_every_kernel_path_over_fragments(n_shards, lists_kernel_path):
    """Rows of up to 80 items put fragments of 0..40+ items through the 16-byte slots and their overflow blocks, in the fast kernel's packed form and in the general
    one; m < the lists' lengths exercises the global cut x_lo.  With the fast kernel available the group is also run with replicated postings (the neighbours
    pipeline: front end + back end); without it that pipeline must stand aside (the lists pipeline answers)."""
    import serenade_amd as sa
    from serenade_amd import sharded
    from oracle import oracle as O
    off, items, ts, ids = small_dataset(58, n_sessions=6000, n_items=500, max_len=80)
    qs = random_queries(18, ids, 600, max_len=8)
    flat, qoff = flatten(qs)
    d_flat, d_off = _to_dev(flat, qoff)
    full = sa.VMISIndex.from_sessions(off, items, ts, 400, 80, 1.0)
    oix = O.OracleIndex(off, items, ts, 400, 80, 1.0)
    shards = [sharded.ShardedVMISIndex(off, items, ts, 400, 80, 1.0, g, n_shards) for g in range(n_shards)]
    grp = sharded.ShardGroup.local(shards)
    for with_postings in (False, True):
        grp.set_postings(full if with_postings else None)
        for (k, m, n) in [(100, 400, 21), (500, 300, 21), (30, 60, 5)]:
            ref = oix.predict_batch("canonical", flat, qoff, k, m, n, False, threads=4)
            _check(_np(grp.predict_batch(d_flat, d_off, len(qs), 8, k, m, n)), ref, sa.predict_batch(full, (flat, qoff), k, m, n, False))
    st = grp.stats
    assert st["stage_batches"] == 0                                                       # sessions of <= 8 items, m <= m_index: never the three-stage pipeline
    assert st["neighbour_batches"] == (3 if lists_kernel_path == "fast" and n_shards > 1 else 0)
    grp.close()


def test_gpu_built_shards_equal_host_built_shards_and_load_from_one_file(tmp_path):
    """The three ways to a shard give the same bytes: the host builder restricted to the shard's items (srn_index_build_shard), the
    unsharded index built on the GPU and cut (srn_index_build_shard_gpu), an unsharded index saved to disk and cut on load
    (srn_index_load_shard) -- the last is how an 8-GPU node would start: build or ship ONE index, every rank cuts its own part."""
    import serenade_amd as sa
    from serenade_amd import sharded
    off, items, ts, ids = small_dataset(54, n_sessions=6000, n_items=500, tied_timestamps=True)
    full = sa.VMISIndex.from_sessions(off, items, ts, 120, 10, 2.0, device=0, builder="gpu")
    full.save(tmp_path / "full.srn")
    for G in (2, 5):
        for g in range(G):
            a = sharded.ShardedVMISIndex(off, items, ts, 120, 10, 2.0, g, G, device=-1, builder="host")
            b = sharded.ShardedVMISIndex(off, items, ts, 120, 10, 2.0, g, G, device=0, builder="gpu")
            c = sharded.ShardedVMISIndex.load(tmp_path / "full.srn", g, G, device=-1)
            d = sharded.ShardedVMISIndex.from_full(full, g, G, device=-1)
            blobs = []
            for nm, ix in (("a", a), ("b", b), ("c", c), ("d", d)):
                ix.save(tmp_path / ("%s.srn" % nm))
                blobs.append(open(tmp_path / ("%s.srn" % nm), "rb").read())
            assert blobs[0] == blobs[1] == blobs[2] == blobs[3], (G, g)


def test_synthetic_shape_through_four_shards_both_pipelines():
    """The production-shaped generator (long lists, popular items, m-cut and k-cut both active; u64 hashed ids) through 4 shards: lists pipeline, then neighbours."""
    import serenade_amd as sa
    from serenade_amd import sharded, synth
    from oracle import oracle as O
    inter, n_items, k, m, idfw = synth.CONFIGS["tiny"]
    off, items, ts = synth.training_sessions(inter, n_items)
    flat, qoff = synth.queries(3000, n_items)
    d_flat, d_off = _to_dev(flat, qoff)
    full = sa.VMISIndex.from_sessions(off, items, ts, m, 34, idfw)
    oix = O.OracleIndex(off, items, ts, m, 34, idfw, fast=True)
    ref = oix.predict_batch("canonical", flat, qoff, k, m, 21, False, threads=4)
    u = sa.predict_batch(full, (flat, qoff), k, m, 21, False)
    shards = [sharded.ShardedVMISIndex(off, items, ts, m, 34, idfw, g, 4) for g in range(4)]
    grp = sharded.ShardGroup.local(shards)
    _check(_np(grp.predict_batch(d_flat, d_off, len(qoff) - 1, 4, k, m, 21)), ref, u)
    grp.set_postings(sharded.postings_view(full))
    _check(_np(grp.predict_batch(d_flat, d_off, len(qoff) - 1, 4, k, m, 21)), ref, u)
    assert grp.stats["neighbour_batches"] == 1 and grp.stats["bytes_lists"] > 0

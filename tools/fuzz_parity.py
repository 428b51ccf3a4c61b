"""Randomised parity soak (GPU box): random small indices, random (k, m, how_many, session lengths, idf weighting, business rules, shard counts), every entry point --
host batches on both sides of the latency-path limit, device-resident batches, the shard group -- against the canonical oracle, bit for bit on ids / counts and 1e-12
on scores.  usage: python tools/fuzz_parity.py [seconds] [seed] [max index rounds]   (prints the failing configuration and exits 1 on the first mismatch)
Round 3: its first minute found the three-stage sharded pipeline returning 0xFFFFFFFF for sessions whose candidate table outgrows LDS (20 items x 3 000 sessions per item) --
the stages have a global-table pass of their own since (device_shard_stage); 12 160 comparisons over 1 216 random small indices pass, and 18 K over 967 indices with the kernel-path knobs and larger indices mixed in (seeds 2-8)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import serenade_amd as sa
from serenade_amd import sharded, capi
from oracle import oracle as O
from helpers import flatten, random_queries, small_dataset

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
max_rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 30
dev = torch.device("cuda:0")
t_end = time.time() + budget
rounds = checks = 0

def check(got, ref, n, what, cfg):
    global checks
    ids, sc, cnt = got
    ok = np.array_equal(cnt, ref["counts"])
    if ok:
        mask = np.arange(n)[None, :] < ref["counts"][:, None].astype(np.int64)
        ok = np.array_equal(ids[mask], ref["ids"][mask]) and np.allclose(sc[mask], ref["scores"][mask], rtol=1e-12, atol=0) and not ids[~mask].any() and not sc[~mask].any()
    checks += 1
    if not ok:
        bad = np.nonzero(cnt != ref["counts"])[0][:5] if not np.array_equal(cnt, ref["counts"]) else np.nonzero((ids != ref["ids"]).any(axis=1))[0][:5]
        print("MISMATCH in %s: %r; first bad queries %s" % (what, cfg, bad.tolist())); sys.exit(1)

while time.time() < t_end and rounds < max_rounds:
    seed = seed0 * 100003 + rounds
    rng = np.random.default_rng(seed)
    # kernel-path knobs (read once by the library: reloaded per round)
    KNOBS = [{}, {}, {}, {"SRN_NO_FAST": "1"}, {"SRN_NO_MERGE": "1"}, {"SRN_NO_MASKS": "1"}, {"SRN_HOT_SLOTS": "64", "SRN_NO_MASKS": "1"}, {"SRN_SKETCH_SLOTS": "64", "SRN_HOT_SLOTS": "32"},
             {"SRN_FAST_RUNS": "3"}, {"SRN_NO_MID": "1"}, {"SRN_NO_BIG": "1"}, {}, {"SRN_DENSE": "1"}, {"SRN_TINY_MAX": "1"}, {"SRN_HOST_CHUNKS": "3"}, {"SRN_SKETCH_SLOTS": "0"}, {"SRN_HOT_SLOTS": "0"},
             # round 5: the LONG instantiation, the serving order, the shard group's wave-per-query back end
             {"SRN_NO_LONG": "1"}, {"SRN_ORDER_MIN": "1"}, {"SRN_ORDER_MIN": "1", "SRN_SBACK_MIN_SHARDS": "2"}, {"SRN_SBACK_MIN_SHARDS": "2"}, {"SRN_SBACK_MIN_SHARDS": "2", "SRN_SBACK_BITMAP": "1"}, {"SRN_NO_SBACK": "1"}, {"SRN_ORDER_MIN": "1"},
             # ... its streaming form (neighbours exchanged as posting positions), its second tier (the fast kernel's back-end form over a list) switched off
             {"SRN_SBACK_MIN_SHARDS": "2", "SRN_SBACK_STREAM": "1"}, {"SRN_SBACK_MIN_SHARDS": "2", "SRN_SBACK_STREAM": "1", "SRN_ORDER_MIN": "1"}, {"SRN_SBACK_MIN_SHARDS": "2", "SRN_NO_SBACK_SECOND": "1"},
             {"SRN_SBACK_MIN_SHARDS": "2"}, {"SRN_SBACK_MIN_SHARDS": "2", "SRN_ORDER_MIN": "1"},
             # ... presence bytes shipped by the fronting rank, rows finished by the serving wave; the latency path's older forms
             {"SRN_SBACK_MIN_SHARDS": "2", "SRN_SBACK_PBYTES": "1"}, {"SRN_SBACK_MIN_SHARDS": "2", "SRN_SBACK_PBYTES": "1", "SRN_ORDER_MIN": "1"}, {"SRN_SBACK_MIN_SHARDS": "2", "SRN_SBACK_FINISH": "1"},
             {"SRN_TINY_FUSED": "0"}, {"SRN_TINY_SPIN": "0"}, {"SRN_TINY_FUSED_MAX": "256", "SRN_TINY_FAST": "3"}, {}, {},
             # round 6: how_many beyond 24 on the general kernel only, pre-built indices without the per-item test / without the inferred tie order, the streaming exchange forced off
             {"SRN_FAST_HOW_MANY_MAX": "24"}, {"SRN_NO_VIOL": "1"}, {"SRN_AVRO_NO_TIE_INFERENCE": "1"}, {"SRN_AVRO_NO_TIE_INFERENCE": "1", "SRN_NO_VIOL": "1"}, {"SRN_SBACK_MIN_SHARDS": "2", "SRN_SBACK_STREAM": "0"}, {}, {}]
    for kk in ("SRN_FAST_HOW_MANY_MAX", "SRN_NO_VIOL", "SRN_AVRO_NO_TIE_INFERENCE", "SRN_SBACK_STREAM"):
        os.environ.pop(kk, None)
    for kk in ("SRN_NO_FAST", "SRN_NO_MID", "SRN_NO_BIG", "SRN_NO_MERGE", "SRN_NO_MASKS", "SRN_HOT_SLOTS", "SRN_SKETCH_SLOTS", "SRN_FAST_RUNS", "SRN_DENSE", "SRN_TINY_MAX", "SRN_HOST_CHUNKS", "SRN_NO_LONG", "SRN_ORDER_MIN", "SRN_SBACK_MIN_SHARDS", "SRN_SBACK_BITMAP", "SRN_NO_SBACK", "SRN_SBACK_STREAM", "SRN_NO_SBACK_SECOND", "SRN_SBACK_PBYTES", "SRN_SBACK_FINISH", "SRN_TINY_FUSED", "SRN_TINY_SPIN", "SRN_TINY_FUSED_MAX", "SRN_TINY_FAST"):
        os.environ.pop(kk, None)
    knobs = KNOBS[int(rng.integers(0, len(KNOBS)))]
    os.environ.update(knobs); capi.reload_knobs()
    big = rng.random() < 0.15   # now and then an index large enough for the cuts to bite on realistic list lengths
    n_sessions = int(rng.choice([300, 2000, 8000, 30000])) if not big else int(rng.choice([120000, 300000])); n_items = int(rng.choice([40, 300, 2500])) if not big else int(rng.choice([2500, 20000]))
    row_max = int(rng.choice([4, 12, 34, 80])); tied = bool(rng.random() < 0.3)
    m_index = int(rng.choice([5, 60, 500, 3000])); idfw = float(rng.choice([0.0, 1.0, 2.0, 5.0]))
    max_q = int(rng.choice([1, 3, 4, 6, 8, 9, 10, 10, 12, 15, 20, 20]))   # (5..10 items: the fast kernel's MID instantiation, round 4)
    off, items, ts, ids = small_dataset(seed, n_sessions=n_sessions, n_items=n_items, tied_timestamps=tied, max_len=row_max)
    avro_mode = None
    if rng.random() < 0.3:
        # round 6: the same sessions through the reference's production route -- a stand-in producer (its own order among equal timestamps, sessions beyond its length cut in the
        # session index only) writes the Avro index, srn_index_new_from_avro loads it; the checker is the oracle's restatement of VMISIndex::new over the lists AS GIVEN, ties
        # among equal timestamps in the order the index serves with
        import shutil, tempfile
        from serenade_amd import synth as _synth
        avro_mode = str(rng.choice(["ours", "reverse", "mixed", "per-item"])); cut = max(1, row_max - int(rng.integers(0, 3)))
        tmpd = tempfile.mkdtemp(prefix="srn_fuzz_avro_")
        try:
            p_ids, p_off, p_sess, p_idf = _synth.avro_index(tmpd, off, items, ts, m_index, cut, idfw, avro_mode, files=int(rng.integers(1, 4)))
            gix = sa.VMISIndex.new_from_avro(tmpd)
        finally:
            shutil.rmtree(tmpd, ignore_errors=True)
        oix = O.OracleIndex.from_parts(p_ids, (p_off, p_sess), p_idf, np.full(len(p_ids), 2, np.uint8), off, items, ts, tie_rank=gix.session_recency())
    else:
        gix = sa.VMISIndex.from_sessions(off, items, ts, m_index, row_max, idfw)
        oix = O.OracleIndex(off, items, ts, m_index, row_max, idfw)
    business = bool(rng.random() < 0.3)
    if business:
        known = (np.unique(items) if avro_mode is None else p_ids); flags = rng.choice(np.array([0, 1, 2, 3, 0xFF], np.uint8), size=len(known), p=[0.15, 0.05, 0.5, 0.2, 0.1])
        gix.set_attributes(known, flags); oix.set_attributes(known, flags)
    nq_all = int(rng.choice([40, 700, 3000]))
    qs = random_queries(seed + 1, ids, nq_all, max_len=max_q, unknown_rate=float(rng.choice([0.0, 0.05, 0.3])), dup_rate=float(rng.choice([0.0, 0.2])))
    flat, qoff = flatten(qs)
    for rep in range(3):
        k = int(rng.choice([1, 7, 100, 500, 1500, 4000])); m = int(rng.choice([1, 20, 300, 2500, 6000])); n = int(rng.choice([1, 5, 21, 24, 25, 40, 50, 64, 100, 512]))
        cfg = dict(avro=avro_mode, knobs=knobs, seed=seed, n_sessions=n_sessions, n_items=n_items, row_max=row_max, tied=tied, m_index=m_index, idfw=idfw, max_q=max_q, business=business, nq=nq_all, k=k, m=m, n=n)
        try:
            ref = oix.predict_batch("canonical", flat, qoff, k, m, n, business, threads=8)
            check(sa.predict_batch(gix, (flat, qoff), k, m, n, business), ref, n, "host batch", cfg)
            sub = min(nq_all, int(rng.choice([1, 16, 200, 256])))
            refs = {kk: v[:sub] for kk, v in ref.items() if kk in ("ids", "scores", "counts")}
            check(sa.predict_batch(gix, (flat[:qoff[sub]], qoff[:sub + 1]), k, m, n, business), refs, n, "host batch <= 256", cfg)
            resident = False
            if rng.random() < 0.35:   # round 6: the persistent latency path behind srn_predict (resident workgroups; what they cannot serve takes the launch path: same rows)
                try:
                    gix.serve_start(k, m, n, business, lanes=int(rng.integers(1, 4)), max_items_in_session=int(min(max_q, 10)), idle_ms=1500); resident = True
                except capi.SerenadeError:
                    resident = False
            for qi in rng.integers(0, nq_all, size=12 if resident else 3):   # srn_predict: the reference's call shape
                recs = sa.predict(gix, qs[int(qi)], k, m, n, business)
                cnt_ref = int(ref["counts"][qi]) if ref["counts"][qi] != 0xFFFFFFFF else 0
                if [r.id for r in recs] != ref["ids"][qi, :cnt_ref].tolist() or not np.allclose([r.score for r in recs], ref["scores"][qi, :cnt_ref], rtol=1e-12, atol=0):
                    print("MISMATCH in srn_predict, query %d: %r" % (qi, cfg)); sys.exit(1)
                checks += 1
            if resident:
                gix.serve_stop()
            d_f = torch.from_numpy(flat.view(np.int64).copy()).to(dev); d_o = torch.from_numpy(qoff.view(np.int32).copy()).to(dev)
            r_ids = torch.zeros(nq_all * n, dtype=torch.int64, device=dev); r_sc = torch.zeros(nq_all * n, dtype=torch.float64, device=dev); r_cnt = torch.zeros(nq_all, dtype=torch.int32, device=dev)
            sa.predict_batch_device(gix, d_f.data_ptr(), d_o.data_ptr(), nq_all, max_q, k, m, n, business, r_ids.data_ptr(), r_sc.data_ptr(), r_cnt.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            check((r_ids.cpu().numpy().view(np.uint64).reshape(nq_all, n), r_sc.cpu().numpy().reshape(nq_all, n), r_cnt.cpu().numpy().view(np.uint32)), ref, n, "device batch", cfg)
            if rep == 0:
                G = int(rng.choice([1, 2, 3, 5, 8]))
                shards = [sharded.ShardedVMISIndex.from_full(gix, g, G) for g in range(G)]
                if business:
                    for s_ in shards:
                        capi.check(capi.lib().srn_index_set_attributes(s_._h, capi.ptr(capi.as_u64(known)), capi.ptr(flags), len(known)))
                grp = sharded.ShardGroup.local(shards)
                res = grp.predict_batch(d_f, d_o, nq_all, max_q, k, m, n, business)
                torch.cuda.synchronize()
                check((res[0].cpu().numpy().view(np.uint64), res[1].cpu().numpy(), res[2].cpu().numpy().view(np.uint32)), ref, n, "shard group x%d (stage batches %d)" % (G, grp.stats["stage_batches"]), cfg)
                # round 4: the same group with replicated postings (the neighbours pipeline where the batch's shape allows it; the lists / three-stage pipelines otherwise)
                post = gix if rng.random() < 0.5 else sharded.postings_view(gix)
                if gix.info["incomplete_items"] == 0:
                    grp.set_postings(post)
                for rep2 in range(2 if gix.info["incomplete_items"] == 0 else 0):
                    res = grp.predict_batch(d_f, d_o, nq_all, max_q, k, m, n, business, resident=bool(rep2))
                    torch.cuda.synchronize()
                    check((res[0].cpu().numpy().view(np.uint64), res[1].cpu().numpy(), res[2].cpu().numpy().view(np.uint32)), ref, n, "shard group x%d with postings (neighbour batches %d)" % (G, grp.stats["neighbour_batches"]), cfg)
                grp.close()
        except capi.SerenadeError as e:
            if e.code != capi.SRN_ERANGE:
                print("ERROR %r in %r" % (e, cfg)); sys.exit(1)
    rounds += 1
print("fuzz ok: %d index rounds, %d comparisons" % (rounds, checks))

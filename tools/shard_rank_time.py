"""What ONE rank of a G-way item-sharded index does per batch, measured on one GPU, for the LISTS pipeline (first) and the NEIGHBOURS pipeline: all G shards of the config in this process (an
in-process group: the all-gathers degenerate), SRN_GROUP_TIMING events around shard 0's own launches -- prep records of the whole batch + the front end over its
1 / G of the queries | the back end over ALL queries on its row fragments (fast kernel back end + general kernel over the handed-over queries + finish kernels) | the
top-n merge.  The exchanges themselves are not in it (no second GPU here): their payload is printed.
usage: python tools/shard_nb_rank_time.py cfg3 8 [batch]"""
import ctypes as C, os, sys, time
os.environ["SRN_GROUP_TIMING"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import serenade_amd as sa
from serenade_amd import capi, sharded as SH, synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 17
inter, n_items, k, m, idfw = synth.CONFIGS[cfg]
off, items, ts = synth.training_sessions(inter, n_items)
full = sa.VMISIndex.from_sessions(off, items, ts, m, 34, idfw, device=0, builder="gpu")
t0 = time.time()
shards = [SH.ShardedVMISIndex.from_full(full, g, G) for g in range(G)]
post = SH.postings_view(full)
print("cut %d shards + the postings view in %.1f s; device bytes: full %.2f GB, shard 0 %.3f GB, replicated postings %.3f GB" % (
    G, time.time() - t0, full.info["device_bytes"] / 1e9, shards[0].info["device_bytes"] / 1e9, post.info["device_bytes"] / 1e9))
qi, qo = synth.queries(int(B / 3.0) + 4096, n_items, seed=synth.SEED + 7919, max_items=synth.LAST_ITEMS)
qo = qo[:B + 1]; qi = qi[:qo[-1]]
dev = torch.device("cuda:0")
d_flat = torch.from_numpy(qi.view(np.int64).copy()).to(dev); d_off = torch.from_numpy(qo.view(np.int32).copy()).to(dev)
L, n = synth.LAST_ITEMS, synth.HOW_MANY
grp = SH.ShardGroup.local(shards)
lres = []
for it in range(4):
    ref = grp.predict_batch(d_flat, d_off, B, L, k, m, n)          # the lists pipeline
    torch.cuda.synchronize()
    t3 = (C.c_double * 3)()
    capi.check(capi.lib().srn_debug_shard_group_times(grp._h, t3))
    lres.append(list(t3))
ref = [x.clone() for x in ref]
la = np.median(np.array(lres[1:]), axis=0)
stl = grp.stats
print("%s G=%d batch %d, LISTS pipeline, one rank: head + count + copy %.3f ms (all %d local shards' / %d) | prep + launch sequence over the gathered lists %.3f ms | merge %.3f ms -> %.3f ms per batch = %.2f M queries/s "
      "per batch stream without the exchanges; list prefixes exchanged: %.0f B/query sent per rank" % (cfg, G, B, la[0] / G, G, G, la[1], la[2], la[0] / G + la[1] + la[2], B / (la[0] / G + la[1] + la[2]) / 1e3,
                                                                                                      stl["bytes_lists"] / max(1, stl["queries"]) / G))
grp.set_postings(post)
capi.check(capi.lib().srn_kernel_timing(shards[0]._h, 1))
res = []
for it in range(5):
    out = grp.predict_batch(d_flat, d_off, B, L, k, m, n)
    torch.cuda.synchronize()
    t3 = (C.c_double * 3)()
    capi.check(capi.lib().srn_debug_shard_group_times(grp._h, t3))
    res.append(list(t3))
assert all(torch.equal(a, b) for a, b in zip(out, ref)), "neighbours pipeline != lists pipeline"
ta, tb, tc, td, nn = np.zeros(8), np.zeros(8), np.zeros(8), np.zeros(8), C.c_uint32()
capi.check(capi.lib().srn_kernel_times_detail(shards[0]._h, 8, capi.ptr(ta), capi.ptr(tb), capi.ptr(tc), capi.ptr(td), C.byref(nn)))
a_, b_, c_ = C.c_uint32(), C.c_uint32(), C.c_uint32()
capi.check(capi.lib().srn_last_path_counts(shards[0]._h, C.byref(a_), C.byref(b_), C.byref(c_)))
print("shard 0's back-end launches (HIP events, last %d calls): fast kernel (back end) %.3f ms, all predict launches %.3f ms (general kernel over %d handed-over queries + finish kernels: %.3f ms), global-table pass %.3f ms" % (
    nn.value, tb[:nn.value].mean(), tc[:nn.value].mean(), b_.value, (tc - tb)[:nn.value].mean(), td[:nn.value].mean()))
if os.environ.get("SRN_NB_PHASES"):      # per-phase shader cycles of shard 0's workgroups (front end over its slice + back end over the batch), one extra batch
    cyc = np.zeros(16, np.uint64)
    capi.check(capi.lib().srn_debug_phase_cycles(shards[0]._h, 1, capi.ptr(cyc)))
    grp.predict_batch(d_flat, d_off, B, L, k, m, n); torch.cuda.synchronize()
    capi.check(capi.lib().srn_debug_phase_cycles(shards[0]._h, 0, capi.ptr(cyc)))
    names = ["0 record+barrier", "1 stage", "2 cuts", "3 merge tree", "4 -", "5", "6", "7", "8 row requests+clears", "9 walk A", "10 phase 4a", "11 live check", "12 walk B+resolve", "13 hand-off", "14", "15"]
    print("phase cycles per query of the batch (front end phases: 1/%d of the queries):" % G, ", ".join("%s %.0f" % (nm, c / B) for nm, c in zip(names, cyc.astype(np.float64)) if c / B > 5))
    c15, c7 = int(cyc[15]), int(cyc[7])
    live = c7 & 0xFFFFFFFF
    print("wave-per-query back end: served %d, live (walk B ran) %d, listed elements / live query %.1f, candidates / served %.1f; handed over by cause: candidates %d, long-fragment queues %d, hit list %d, exact table %d" % (
        int(cyc[14]), live, cyc[5] / max(1.0, float(live)), cyc[6] / max(1.0, float(cyc[14])), c15 & 0xFFFFF, (c15 >> 20) & 0xFFFFF, (c15 >> 40) & 0xFFFFF, c7 >> 32))
a = np.median(np.array(res[1:]), axis=0)
st = grp.stats
print("%s G=%d batch %d, NEIGHBOURS pipeline, one rank: prep + front end (1/%d of the queries) %.3f ms | back end over all queries %.3f ms | merge %.3f ms -> %.3f ms per batch = %.2f M queries/s "
      "per batch stream without the exchanges; neighbour lists all-gathered: %.1f MB per rank sent (%.0f B/query of the batch), top-n gather %.1f MB per rank; results == lists pipeline" % (
          cfg, G, B, G, a[0], a[1], a[2], a.sum(), B / a.sum() / 1e3, (B / G) * (k + 1) * 4 / 1e6, (k + 1) * 4 / G, B * n * 16 / 1e6))

"""BASELINE configs[4] (2.3 B interactions / 20 M items, k=1500 m=2500) on ONE MI355X: the unsharded index in HBM (u64 slots: 477 M
sessions need 29 rank bits + 4 position bits), its throughput, parity of a query sample against the canonical oracle, and the
8-way item-sharded index cut from it: all 8 shards resident on the one GPU, the lists pipeline against the unsharded result,
one rank's kernel time.  Needs ~150 GB of host memory and ~200 GB of HBM.   usage: python tools/cfg5_one_gpu.py [cfg5] [--no-oracle]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import serenade_amd as sa
from serenade_amd import sharded as SH, synth

cfg = next((a for a in sys.argv[1:] if not a.startswith("--")), "cfg5")
want_oracle = "--no-oracle" not in sys.argv
def say(*a):
    print(*a); sys.stdout.flush()
try:
    say("cgroup memory.max:", open("/sys/fs/cgroup/memory.max").read().strip())
except Exception as e:
    say("cgroup memory.max: n/a", e)
inter, n_items, k, m, idfw = synth.CONFIGS[cfg]
t0 = time.time(); off, items, ts = synth.training_sessions(inter, n_items); t_gen = time.time() - t0
say("%s: %d interactions, %d sessions generated in %.1f s" % (cfg, len(items), len(ts), t_gen))
t0 = time.time(); full = sa.VMISIndex.from_sessions(off, items, ts, m, 34, idfw, device=0, builder="gpu"); t_build = time.time() - t0
info = full.info
say("index built on the GPU + attached in %.1f s: %d items, %d kept sessions, %.2f GB on device" % (t_build, info["n_items"], info["n_sessions_kept"], info["device_bytes"] / 1e9))
B = 1 << 18
qi, qo = synth.queries(int(B / 3.0) + 4096, n_items, seed=synth.SEED + 7919, max_items=synth.LAST_ITEMS)
qo = qo[:B + 1]; qi = qi[:qo[-1]]
dev = torch.device("cuda:0")
d_flat = torch.from_numpy(qi.view(np.int64).copy()).to(dev); d_off = torch.from_numpy(qo.view(np.int32).copy()).to(dev)
L, n = synth.LAST_ITEMS, synth.HOW_MANY
out_ids = torch.zeros(B * n, dtype=torch.int64, device=dev); out_sc = torch.zeros(B * n, dtype=torch.float64, device=dev); out_cnt = torch.zeros(B, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
def run():
    sa.predict_batch_device(full, d_flat.data_ptr(), d_off.data_ptr(), B, L, k, m, n, False, out_ids.data_ptr(), out_sc.data_ptr(), out_cnt.data_ptr(), st)
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
say("unsharded, %d queries per call, resident: %.2f ms per call = %.2f M queries/s; path counts (nq, general, global pass) %s" % (B, ms, B / ms / 1e3, full.last_path_counts()))
ref_ids = out_ids.cpu().numpy().view(np.uint64).reshape(B, n).copy(); ref_sc = out_sc.cpu().numpy().reshape(B, n).copy(); ref_cnt = out_cnt.cpu().numpy().view(np.uint32).copy()
# path equivalence: the general kernel alone (u64 slots at this size) must give the same bytes as fast kernel + hand-overs
from serenade_amd import capi
os.environ["SRN_NO_FAST"] = "1"; capi.reload_knobs()
run(); torch.cuda.synchronize()
e0.record(); run(); e1.record(); torch.cuda.synchronize()
same_paths = np.array_equal(out_ids.cpu().numpy().view(np.uint64).reshape(B, n), ref_ids) and np.array_equal(out_sc.cpu().numpy().reshape(B, n), ref_sc) and np.array_equal(out_cnt.cpu().numpy().view(np.uint32), ref_cnt)
say("general kernel alone: %.2f ms per call = %.2f M queries/s; identical to the fast path's results: %s" % (e0.elapsed_time(e1), B / e0.elapsed_time(e1) / 1e3, same_paths))
del os.environ["SRN_NO_FAST"]; capi.reload_knobs()
if not same_paths:
    gi = out_ids.cpu().numpy().view(np.uint64).reshape(B, n); gs = out_sc.cpu().numpy().reshape(B, n); gc = out_cnt.cpu().numpy().view(np.uint32)
    bad = np.nonzero((gi != ref_ids).any(1) | (gs != ref_sc).any(1) | (gc != ref_cnt))[0]
    say("differing queries: %d of %d; first: %s" % (len(bad), B, bad[:8]))
    for q in bad[:4]:
        say(" q", q, "L", qo[q + 1] - qo[q], "items", qi[qo[q]:qo[q + 1]], "counts fast/general", ref_cnt[q], gc[q])
        say("   fast   ", ref_ids[q][:8], ref_sc[q][:8])
        say("   general", gi[q][:8], gs[q][:8])
    if "--stop-on-diff" in sys.argv: sys.exit(1)

# ---- 8 shards, all on this GPU ----
G = 8
t0 = time.time()
shards = [SH.ShardedVMISIndex.from_full(full, g, G) for g in range(G)]
say("cut %d shards in %.1f s; shard 0: %.2f GB on device, sum %.2f GB" % (G, time.time() - t0, shards[0].info["device_bytes"] / 1e9, sum(s.info["device_bytes"] for s in shards) / 1e9))
NS = 1 << 16
so = qo[:NS + 1]; sf = qi[:so[-1]]
s_flat = torch.from_numpy(sf.view(np.int64).copy()).to(dev); s_off = torch.from_numpy(so.view(np.int32).copy()).to(dev)
ok_mode = SH.lists_supported(shards[0], L, k, m, n)
say("lists pipeline supported:", ok_mode)
t0 = time.time()
res = SH.predict_batch_sharded_lists_local(shards, s_flat, s_off, NS, L, k, m, n) if ok_mode else SH.predict_batch_sharded_local(shards, s_flat, s_off, NS, L, k, m, n)
torch.cuda.synchronize()
g_ids = res[0].cpu().numpy().view(np.uint64); g_sc = res[1].cpu().numpy(); g_cnt = res[2].cpu().numpy().view(np.uint32)
same = np.array_equal(g_cnt, ref_cnt[:NS]) and np.array_equal(g_ids, ref_ids[:NS]) and np.array_equal(g_sc, ref_sc[:NS])
say("8-way sharded pipeline == unsharded on %d queries: %s (%.1f s for all 8 ranks' work on one GPU)" % (NS, same, time.time() - t0))

if want_oracle:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from oracle import oracle as O
    for s in shards: s.close()
    t0 = time.time(); oix = O.OracleIndex(off, items, ts, m, 34, idfw, fast=True); say("oracle index in %.1f s" % (time.time() - t0))
    nc = 1000
    t0 = time.time(); ref = oix.predict_batch("canonical", qi[:qo[nc]], qo[:nc + 1], k, m, n, False, threads=16)
    mask = np.arange(n)[None, :] < ref["counts"][:, None].astype(np.int64)
    ok = np.array_equal(ref_cnt[:nc], ref["counts"]) and np.array_equal(ref_ids[:nc][mask], ref["ids"][mask]) and np.allclose(ref_sc[:nc][mask], ref["scores"][mask], rtol=1e-12, atol=0)
    say("canonical oracle on %d queries (%.1f s): ids / counts exact, scores 1e-12: %s" % (nc, time.time() - t0, ok))
    assert ok
assert same

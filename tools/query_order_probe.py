"""Does the ORDER in which a batch's queries are processed matter?  Queries that share their most popular evolving item share most of their posting-list entries and
neighbour rows; processed close together (in time, and on one XCD: block b runs on XCD b % 8, each XCD has its own 4 MB L2) the second one finds them in L2.
The probe: the same 2^20 (or given) queries through srn_predict_batch_device (a) in the generator's random order, (b) sorted on the HOST by the batch-frequency of
their most frequent item -- a stand-in for a device-side ordering pass --, fast kernel time from HIP events; and the same for one rank's back end of an 8-way
item-sharded index.  Results are row-for-row the same (checked).
usage: python tools/query_order_probe.py [cfg3] [batch] [G]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import serenade_amd as sa
from serenade_amd import capi, sharded as SH, synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
G = int(sys.argv[3]) if len(sys.argv) > 3 else 8
inter, n_items, k, m, idfw = synth.CONFIGS[cfg]
off, items, ts = synth.training_sessions(inter, n_items)
full = sa.VMISIndex.from_sessions(off, items, ts, m, 34, idfw, device=0, builder="gpu")
qi, qo = synth.queries(int(B / 3.0) + 4096, n_items, seed=synth.SEED + 7919, max_items=synth.LAST_ITEMS)
qo = qo[:B + 1].astype(np.int64); qi = qi[:qo[-1]]
lens = np.diff(qo)
# key of a query: its most frequent item in the batch (ties: the larger id), queries ordered by (frequency desc, item)
uniq, inv, cnt = np.unique(qi, return_inverse=True, return_counts=True)
fq = cnt[inv]                                                # frequency of every query item
seg = np.repeat(np.arange(B), lens)
best = np.zeros(B, np.int64); np.maximum.at(best, seg, fq * (1 << 22) + (inv % (1 << 22)))
order = np.argsort(-best, kind="stable")


def permuted(order):
    l2 = lens[order]
    o2 = np.zeros(B + 1, np.int64); o2[1:] = np.cumsum(l2)
    take = np.repeat(qo[order] - o2[:-1], l2) + np.arange(int(o2[-1]))
    return np.ascontiguousarray(qi[take]), o2.astype(np.uint32)


dev = torch.device("cuda:0")
L, n = synth.LAST_ITEMS, synth.HOW_MANY
out = (torch.zeros(B * n, dtype=torch.int64, device=dev), torch.zeros(B * n, dtype=torch.float64, device=dev), torch.zeros(B, dtype=torch.int32, device=dev))
stream = torch.cuda.current_stream()
res = {}
for name, (f_, o_), omin in (("random order, no ordering pass", (qi, qo.astype(np.uint32)), "0"), ("sorted on the host, no ordering pass", permuted(order), "0"),
                             ("random order, DEVICE ordering pass (default)", (qi, qo.astype(np.uint32)), None)):
    if omin is None:
        os.environ.pop("SRN_ORDER_MIN", None)
    else:
        os.environ["SRN_ORDER_MIN"] = omin
    capi.reload_knobs()
    d_flat = torch.from_numpy(f_.view(np.int64).copy()).to(dev); d_off = torch.from_numpy(o_.view(np.int32).copy()).to(dev)
    full.kernel_timing(True)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(8)]
    for it in range(8):
        ev[it][0].record(stream)
        sa.predict_batch_device(full, d_flat.data_ptr(), d_off.data_ptr(), B, L, k, m, n, False, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), stream.cuda_stream)
        ev[it][1].record(stream)
    torch.cuda.synchronize()
    tp, tf, tpr, tr = full.kernel_times_detail(4)
    res[name] = (out[0].cpu().numpy().reshape(B, n).copy(), out[2].cpu().numpy().copy())
    print("%s B=%d unsharded, %-45s: fast kernel %.3f ms, all predict launches %.3f ms, prep (+ sort) %.3f ms, whole call %.3f ms" % (
        cfg, B, name, tf.mean(), tpr.mean(), tp.mean(), np.median([a_.elapsed_time(b_) for a_, b_ in ev[3:]])))
names = list(res)
a, b, c = res[names[0]], res[names[1]], res[names[2]]
mask = np.arange(n)[None, :] < a[1][:, None]
assert np.array_equal(a[1][order], b[1]) and np.array_equal(np.where(mask, a[0], 0)[order], np.where(mask[order], b[0], 0)), "host-sorted batch: results differ"
assert np.array_equal(a[1], c[1]) and np.array_equal(np.where(mask, a[0], 0), np.where(mask, c[0], 0)), "device ordering pass: results differ"
print("rows identical in all three")
if G > 1:
    os.environ["SRN_GROUP_TIMING"] = "1"
    Bs = min(B, 1 << 17)
    shards = [SH.ShardedVMISIndex.from_full(full, g, G) for g in range(G)]
    grp = SH.ShardGroup.local(shards)
    grp.set_postings(SH.postings_view(full))
    for name, omin in (("no ordering pass", "0"), ("device ordering pass (default)", None)):
        if omin is None:
            os.environ.pop("SRN_ORDER_MIN", None)
        else:
            os.environ["SRN_ORDER_MIN"] = omin
        capi.reload_knobs()
        d_flat = torch.from_numpy(qi[:qo[Bs]].view(np.int64).copy()).to(dev); d_off = torch.from_numpy(qo[:Bs + 1].astype(np.uint32).view(np.int32).copy()).to(dev)
        ts_ = []
        for it in range(5):
            got = grp.predict_batch(d_flat, d_off, Bs, L, k, m, n); torch.cuda.synchronize()
            t3 = (C.c_double * 3)(); capi.check(capi.lib().srn_debug_shard_group_times(grp._h, t3)); ts_.append(list(t3))
        t = np.median(np.array(ts_[1:]), axis=0)
        cn = got[2].cpu().numpy()
        assert np.array_equal(cn.view(np.uint32), a[1][:Bs].view(np.uint32)), "sharded counts differ from the unsharded path"
        print("%s G=%d batch %d, one rank, %-32s: prep (+ sort) + front %.3f ms | back end %.3f ms | merge %.3f ms" % (cfg, G, Bs, name, t[0], t[1], t[2]))

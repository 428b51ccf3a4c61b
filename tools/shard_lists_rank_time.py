"""What ONE rank of a G-way item-sharded index does per batch in lists mode, measured on one GPU: all G shards of the config are
built in this process, the exchange is emulated with tensor ops (not timed), and shard 0's own kernels are timed with events:
head + count + copy (before the exchange) and the prep + unsharded launch sequence over its row fragments (after it).
usage: python tools/shard_lists_rank_time.py cfg3 8 [batch]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import serenade_amd as sa
from serenade_amd import sharded as SH, synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 17
inter, n_items, k, m, idfw = synth.CONFIGS[cfg]
off, items, ts = synth.training_sessions(inter, n_items)
full = sa.VMISIndex.from_sessions(off, items, ts, m, 34, idfw, device=0, builder="gpu")
t0 = time.time()
shards = [SH.ShardedVMISIndex.from_full(full, g, G) for g in range(G)]
print("cut %d shards in %.1f s; device bytes: full %.2f GB, shard 0 %.3f GB (sum of shards %.2f GB)" % (
    G, time.time() - t0, full.info["device_bytes"] / 1e9, shards[0].info["device_bytes"] / 1e9, sum(s.info["device_bytes"] for s in shards) / 1e9))
qi, qo = synth.queries(int(B / 3.0) + 4096, n_items, seed=synth.SEED + 7919, max_items=synth.LAST_ITEMS)
qo = qo[:B + 1]; qi = qi[:qo[-1]]
dev = torch.device("cuda:0")
d_flat = torch.from_numpy(qi.view(np.int64).copy()).to(dev); d_off = torch.from_numpy(qo.view(np.int32).copy()).to(dev)
L, n = synth.LAST_ITEMS, synth.HOW_MANY
stream = torch.cuda.current_stream().cuda_stream
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
res = []
for it in range(4):
    h = [SH._lists_head(ix, d_flat, d_off, B, L, m, stream) for ix in shards]
    head = torch.stack([x[1] for x in h]).max(dim=0).values.contiguous()
    c = [SH._lists_count(ix, d_off, B, L, x[0], head, stream) for ix, x in zip(shards, h)]
    stride = (max(1, max(int(x[2].item()) for x in c)) + 63) // 64 * 64
    flats = [SH._lists_copy(ix, B, L, x[0], y[0], y[1], stride, stream) for ix, x, y in zip(shards, h, c)]
    kept_g, off_g, lists_g = torch.stack([y[0] for y in c]).contiguous(), torch.stack([y[1] for y in c]).contiguous(), torch.stack(flats).contiguous()
    torch.cuda.synchronize()
    ev[0].record()
    h0 = SH._lists_head(shards[0], d_flat, d_off, B, L, m, stream)
    c0 = SH._lists_count(shards[0], d_off, B, L, h0[0], head, stream)
    f0 = SH._lists_copy(shards[0], B, L, h0[0], c0[0], c0[1], stride, stream)
    ev[1].record()
    r = SH._lists_predict(shards[0], d_flat, d_off, B, L, k, m, n, kept_g, off_g, lists_g, head, h[0][0], stream)
    ev[2].record()
    torch.cuda.synchronize()
    res.append((ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])))
    if it == 0:
        nq_l, general, glob = shards[0].last_path_counts() if hasattr(shards[0], "last_path_counts") else (0, 0, 0)
a, b = np.median([x[0] for x in res[1:]]), np.median([x[1] for x in res[1:]])
ex = (kept_g.numel() * 4 + off_g.numel() * 8 + lists_g.numel() * 4) / G
print("%s G=%d batch %d: export %.3f ms, predict %.3f ms per rank -> %.2f M queries/s per batch stream without the exchanges; exchange payload %.1f MB per rank (%.0f B/query), top-n gather %.1f MB" % (
    cfg, G, B, a, b, B / (a + b) / 1e3, ex / 1e6, ex / B, B * n * 16 / 1e6))

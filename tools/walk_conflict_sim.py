"""Walk A's LDS serialisation, simulated on the CPU over real neighbour lists of config 3 (round 3): the neighbour rows dealt to lanes as vmis_fast_kernel does,
every lane's accumulator word per item position, and the LDS-array cycles of one ds_add_u32 wave instruction under the model of MI355X_MICROARCH.md (section LDS):
two lane groups of 32, bank = (addr / 4) mod 32, a group takes as many cycles as its fullest bank has lanes (atomics to ONE address serialise like a conflict).
Configurations: (R, H) = R replicas of the H hottest items, the replica chosen by the row's recency rank.   usage: python tools/walk_conflict_sim.py [queries]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from serenade_amd import synth
from oracle import oracle as O
NQ = int(sys.argv[1]) if len(sys.argv) > 1 else 10
inter, n_items, k, m, idfw = synth.CONFIGS["cfg3"]
off, items, ts = synth.training_sessions(inter, n_items)
f = O.OracleIndex(off, items, ts, m, 34, idfw, fast=True)
uid, cnt = np.unique(items, return_counts=True)
order = np.lexsort((uid, -cnt)); rank_of = np.empty(len(uid), np.int64); rank_of[order] = np.arange(len(uid))
qi, qo = synth.queries(3000, n_items, seed=synth.SEED + 7919)
HOT, SK = 4096, 4096
CONFIGS = [(1, 0), (8, 16), (8, 32), (16, 16), (4, 64), (8, 64), (16, 32), (32, 16), ("ideal", 0)]
def words(idx, row, R, H):
    w = np.where(idx < HOT, idx, HOT + (idx & (SK - 1)))
    if R == "ideal":
        return w * 64 + np.arange(len(idx))   # (no two lanes ever share an address: what random banks alone cost)
    if R > 1:
        w = np.where(idx < H, HOT + SK + idx * R + (row % R), w)
    return w
def cost(w, act):
    c = 0
    for g in (slice(0, 32), slice(32, 64)):
        a = w[g][act[g]]
        if len(a):
            c += np.bincount(a % 32, minlength=32).max()
    return c
res = {c: [0, 0] for c in CONFIGS}
done = 0
for q in range(200, 1200):
    ev = qi[qo[q]:qo[q + 1]]
    sid, num, U = f.neighbors_canonical(ev, k, m)
    if len(sid) < 1000:
        continue
    done += 1
    K = len(sid)
    lens = (off[sid.astype(np.int64) + 1] - off[sid.astype(np.int64)]).astype(np.int64)
    rows = np.full((K, 14), -1, np.int64)
    for j, s in enumerate(sid):
        r = rank_of[np.searchsorted(uid, items[off[s]:off[s + 1]])][:14]
        rows[j, :len(r)] = r
    for w8 in range(8):
        for t in range(3):
            js = w8 * 64 + np.arange(64) + t * 512
            ok = js < K
            jj = np.minimum(js, K - 1)
            for p in range(14):
                idx = rows[jj, p]; act = ok & (idx >= 0)
                if not act.any():
                    continue
                for c in CONFIGS:
                    wd = words(np.maximum(idx, 0), sid[jj].astype(np.int64), c[0], c[1])
                    res[c][0] += cost(wd, act); res[c][1] += 1
    if done >= NQ:
        break
for c, (tot, n) in res.items():
    print("R=%-5s replicas of the %3d hottest items: %.2f LDS cycles per ds_add wave instruction (%d instructions per query)" % (c[0], c[1], tot / n, n / done))

"""Scratch analysis (CPU, oracle = test infrastructure): per-query shape statistics of a synthetic config, to size the
fast kernel's LDS buffers and walk rounds.  python tools/workload_stats.py cfg3 2000"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from serenade_amd import synth
from oracle import oracle as O
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
NQ = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
inter, n_items, k, m, idfw = synth.CONFIGS[cfg]
off, items, ts = synth.training_sessions(inter, n_items)
oix = O.OracleIndex(off, items, ts, m, 34, idfw, fast=True)
qi, qo = synth.queries(NQ // 3 + 2048, n_items, seed=synth.SEED + 7919)
lens = np.diff(off.astype(np.int64))
rows = []
for q in range(NQ):
    s = qi[qo[q]:qo[q + 1]]
    L = len(s)
    seen, lists = set(), []
    for pos in range(L):
        it = int(s[L - 1 - pos])
        if it in seen: continue
        seen.add(it)
        p, _ = oix.postings(it)
        if p is None or len(p) == 0: continue
        lists.append(ts[p[:m]])
    nruns = len(lists)
    P = sum(len(l) for l in lists)
    xlo = max([l[m - 1] for l in lists if len(l) >= m], default=0)
    kept = [int((l >= xlo).sum()) for l in lists]
    n = sum(kept)
    sid, num, U = oix.neighbors_canonical(s, k, m)
    rl = lens[sid] if len(sid) else np.zeros(0, np.int64)
    rows.append((L, nruns, P, n, len(sid), (rl > 6).sum(), (rl > 14).sum(), (rl > 30).sum(), rl.sum(), max(kept, default=0)))
a = np.array(rows, np.float64)
names = ["L", "nruns", "P", "n_staged", "K", "rows>6", "rows>14", "rows>30", "I", "max_kept"]
for i, nm in enumerate(names):
    print("%-9s mean %9.1f  p10 %7.0f p50 %7.0f p90 %7.0f p99 %7.0f max %7.0f" % ((nm, a[:, i].mean()) + tuple(np.percentile(a[:, i], [10, 50, 90, 99, 100]))))
print("nruns hist", np.bincount(a[:, 1].astype(int)))
print("frac n_staged > 8704:", (a[:, 3] > 8704).mean(), " > 6000:", (a[:, 3] > 6000).mean(), " >5000:", (a[:,3] > 5000).mean())
print("frac K < 1500:", (a[:, 4] < 1500).mean(), " K<200:", (a[:, 4] < 200).mean(), "K<24", (a[:,4] < 24).mean())

"""Scratch measurement (GPU box): single-query latency through srn_predict (host pointers, PCIe-inclusive) on a config."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import serenade_amd as sa
from serenade_amd import synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
max_items = int(sys.argv[2]) if len(sys.argv) > 2 else synth.LAST_ITEMS   # sessions keep their last max_items items (4 = the headline workload)
inter, n_items, k, m, idfw = synth.CONFIGS[cfg]
off, items, ts = synth.training_sessions(inter, n_items)
ix = sa.VMISIndex.from_sessions(off, items, ts, m, 34, idfw, device=0, builder="gpu")
qi, qo = synth.queries(12000, n_items, seed=synth.SEED + 7919, max_items=max_items)
qlen = np.diff(qo.astype(np.int64))
for nq in (1, 4, 16):
    lat = []
    for i in range(0, min(1200 * nq, len(qo) - 1 - nq), nq):
        f, o = qi[qo[i]:qo[i + nq]], (qo[i:i + nq + 1] - qo[i]).astype(np.uint32)
        t1 = time.perf_counter()
        if nq == 1:
            sa.predict(ix, f, k, m, 21, False)
        else:
            sa.predict_batch(ix, (f, o), k, m, 21, False)
        lat.append((time.perf_counter() - t1) * 1e6)
    lat = np.array(lat[200:])
    print("%s, max_items %d: %2d queries per call (host pointers): p50 %.1f us  p90 %.1f us  p99 %.1f us" % (cfg, max_items, nq, np.percentile(lat, 50), np.percentile(lat, 90), np.percentile(lat, 99)))
    if nq == 1 and max_items > 4:
        ln = qlen[200:200 + len(lat)]
        for lo, hi in ((1, 4), (5, 8), (9, 10)):
            sel = (ln >= lo) & (ln <= hi)
            if sel.any():
                print("      sessions of %d..%d items (%d): p50 %.1f us  p90 %.1f us" % (lo, hi, int(sel.sum()), np.percentile(lat[sel], 50), np.percentile(lat[sel], 90)))

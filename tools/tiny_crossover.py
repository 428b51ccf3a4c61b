"""Where does the latency path's fast launch sequence stop paying?  Host batches of 1..256 sessions (config 3, sessions of <= 4 items), SRN_TINY_FAST=0 (prep + general kernel) against 3 (the fast
kernel's sequence for every batch), alternating in one process (knobs re-read between the blocks).  usage: python tools/tiny_crossover.py [cfg3]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import serenade_amd as sa
from serenade_amd import synth, capi
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
inter, n_items, k, m, idfw = synth.CONFIGS[cfg]
off, items, ts = synth.training_sessions(inter, n_items)
ix = sa.VMISIndex.from_sessions(off, items, ts, m, 34, idfw, device=0, builder="gpu")
qi, qo = synth.queries(40000, n_items, seed=synth.SEED + 7919)
for nq in (1, 8, 16, 24, 32, 48, 64, 128, 256):
    res = {}
    for rep in range(2):
        for tf in ("0", "3"):
            os.environ["SRN_TINY_FAST"] = tf; capi.reload_knobs()
            lat = []
            out = None
            for i in range(0, min(500 * nq, len(qo) - 1 - nq), nq):
                f, o = qi[qo[i]:qo[i + nq]], (qo[i:i + nq + 1] - qo[i]).astype(np.uint32)
                t1 = time.perf_counter()
                out = sa.predict_batch(ix, (f, o), k, m, 21, False, out=out)
                lat.append((time.perf_counter() - t1) * 1e6)
            res.setdefault(tf, []).append(np.array(lat[50:]))
    a, b = np.concatenate(res["0"]), np.concatenate(res["3"])
    print("%3d sessions per call: prep + general kernel p50 %.1f us p90 %.1f   fast sequence p50 %.1f us p90 %.1f" % (nq, np.percentile(a, 50), np.percentile(a, 90), np.percentile(b, 50), np.percentile(b, 90)), flush=True)
os.environ.pop("SRN_TINY_FAST", None)

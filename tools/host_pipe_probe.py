"""srn_predict_batch on host pointers: where the time goes (round 3).  usage: python tools/host_pipe_probe.py [cfg3]   (needs a GPU)
Times the chunked pipeline of srn_hostpipe.hip at several batch sizes under different chunk counts / copy-thread settings; SRN_HOST_TRACE prints
each call's host-side timeline."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import serenade_amd as sa
from serenade_amd import capi, synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
inter, n_items, k, m, idfw = synth.CONFIGS[cfg]
off, items, ts = synth.training_sessions(inter, n_items)
ix = sa.VMISIndex.from_sessions(off, items, ts, m, 34, idfw, builder="gpu")
qi, qo = synth.queries(int((1 << 20) / 3.0) + 4096, n_items, seed=synth.SEED + 7919, max_items=synth.LAST_ITEMS)
n = synth.HOW_MANY
def run(nq, reps, label, **env):
    for kk, v in env.items():
        os.environ[kk] = str(v)
    capi.reload_knobs()
    f, o = qi[:qo[nq]], qo[:nq + 1]
    out = None; ms = []
    for r in range(reps + 2):
        t0 = time.perf_counter(); out = sa.predict_batch(ix, (f, o), k, m, n, False, out=out); ms.append((time.perf_counter() - t0) * 1e3)
    for kk in env:
        del os.environ[kk]
    capi.reload_knobs()
    ms = np.array(ms[2:])
    print("%-44s nq %8d  p50 %8.3f ms  min %8.3f ms  -> %6.2f M q/s" % (label, nq, np.median(ms), ms.min(), nq / np.median(ms) / 1e3)); sys.stdout.flush()
os.environ["SRN_HOST_TRACE"] = "1"
SIZES = [int(x) for x in os.environ.get("PROBE_SIZES", "1048576,65536,4096").split(",")]
QUICK = os.environ.get("PROBE_QUICK") is not None    # chunk counts only
for nq, reps in [(s, 4 if s >= (1 << 19) else 10) for s in SIZES]:
    run(nq, reps, "default")
    run(nq, reps, "no copy to the caller's buffers", SRN_HOST_NOCOPY=1)
    for ch in ((1, 2, 3, 4, 6) if QUICK else (1, 2, 4, 8, 16, 32)):
        if nq // ch >= 1024:
            run(nq, reps, "chunks=%d" % ch, SRN_HOST_CHUNKS=ch)
    for blk in (() if QUICK else (0, 16, 32, 128, 256)):
        run(nq, reps, "download kernel blocks=%d" % blk, SRN_D2H_BLOCKS=blk)

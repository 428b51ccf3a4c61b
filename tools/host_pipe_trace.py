"""A few srn_predict_batch calls on host pointers, to be run under `rocprofv3 --kernel-trace --memory-copy-trace` (timeline of the chunked pipeline).
usage: python tools/host_pipe_trace.py [nq] [calls]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import serenade_amd as sa
from serenade_amd import synth
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 3
inter, n_items, k, m, idfw = synth.CONFIGS["cfg3"]
off, items, ts = synth.training_sessions(inter, n_items)
ix = sa.VMISIndex.from_sessions(off, items, ts, m, 34, idfw, builder="gpu")
qi, qo = synth.queries(int(nq / 3.0) + 4096, n_items, seed=synth.SEED + 7919, max_items=synth.LAST_ITEMS)
f, o = qi[:qo[nq]], qo[:nq + 1]
out = None
for c in range(calls):
    t0 = time.perf_counter(); out = sa.predict_batch(ix, (f, o), k, m, synth.HOW_MANY, False, out=out); print("call %d: %.3f ms" % (c, (time.perf_counter() - t0) * 1e3))

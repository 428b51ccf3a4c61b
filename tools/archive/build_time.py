import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import serenade_amd as sa
from serenade_amd import synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
inter, n_items, k, m, idfw = synth.CONFIGS[cfg]
t0 = time.time(); off, items, ts = synth.training_sessions(inter, n_items); print("generate %.2f s" % (time.time() - t0), flush=True)
for b in ("gpu", "gpu", "host"):
    t0 = time.time(); ix = sa.VMISIndex.from_sessions(off, items, ts, m, 34, idfw, builder=b); dt = time.time() - t0
    print("%s builder: %.2f s  (%d items, %d sessions, %d pairs, %.2f GB on device)" % (b, dt, ix.info["n_items"], ix.info["n_sessions_kept"], ix.info["nnz_rows"], ix.info["device_bytes"] / 1e9), flush=True)
    ix.close()

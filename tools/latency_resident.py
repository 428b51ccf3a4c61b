"""srn_predict latency with and without the persistent latency path (round 6), C++ host (serenade_amd/bin/serve_bench), config 3 by default.
python tools/latency_resident.py [cfg] > profiles/r06_latency_resident_<cfg>.json   (needs a GPU)"""
import json, os, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import serenade_amd as sa
from serenade_amd import build, synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
inter, n_items, k, m, idfw = synth.CONFIGS[cfg]
exe = build.build_serve_bench()
off, items, ts = synth.training_sessions(inter, n_items)
ix = sa.VMISIndex.from_sessions(off, items, ts, m, 34, idfw, builder="gpu")
tmp = tempfile.mkdtemp()
ipath, qpath = os.path.join(tmp, "index.srn"), os.path.join(tmp, "queries.bin")
ix.save(ipath); ix.close()
qi, qo = synth.queries(20000, n_items, seed=synth.SEED + 7919)
with open(qpath, "wb") as f:
    f.write(np.uint64(len(qo) - 1).tobytes()); f.write(qo.astype(np.uint32).tobytes()); f.write(qi.astype(np.uint64).tobytes())
runs = []
secs = os.environ.get("SRN_SERVE_SECONDS", "4")
for threads, resident in ((1, 0), (1, 1), (4, 0), (4, 4), (16, 0), (16, 8), (16, 16), (64, 16)):
    out = subprocess.run([exe, ipath, qpath, str(threads), secs, str(k), str(m), str(synth.HOW_MANY), "4096", "100", "direct"] + (["resident:%d" % resident] if resident else []),
                         capture_output=True, text=True, timeout=120)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    runs.append(json.loads(line[-1]) if line else {"error": out.stderr[-400:], "threads": threads, "resident": resident})
    print(json.dumps(runs[-1]), file=sys.stderr)
print(json.dumps({"config": cfg, "k": k, "m": m, "how_many": synth.HOW_MANY,
                  "what": "closed loop, C++ host: every client thread calls srn_predict with its next evolving session when the previous answer is back; "
                          "resident_workgroups > 0: the persistent latency path (srn_index_serve_start) behind the same call", "runs": runs}, indent=1))

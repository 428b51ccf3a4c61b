"""Latency under concurrent load through the dynamic batcher (serenade_amd/bin/serve_bench), config 3 by default.
python tools/serve_bench.py [cfg] > profiles/rNN_serving_<cfg>.json   (needs a GPU)"""
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
lanes_list = [int(x) for x in os.environ.get("SRN_SERVE_LANES", "4").split(",")]     # SRN_PREDICT_LANES values for the direct mode (0 = no combining, round 2's behaviour)
secs = os.environ.get("SRN_SERVE_SECONDS", "4")
for mode in (["direct"], []):
    for lanes in (lanes_list if mode else [None]):
        for threads in (1, 16, 64, 256, 1024):
            if mode and threads > 256:
                continue
            env = dict(os.environ)
            if lanes is not None:
                env["SRN_PREDICT_LANES"] = str(lanes)
            out = subprocess.run([exe, ipath, qpath, str(threads), secs, str(k), str(m), str(synth.HOW_MANY), "4096", "100"] + mode,
                                 capture_output=True, text=True, env=env)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            runs.append(json.loads(line[-1]) if line else {"error": out.stderr[-400:], "threads": threads, "mode": mode})
            if lanes is not None:
                runs[-1]["predict_lanes"] = lanes
            print(json.dumps(runs[-1]), file=sys.stderr)
print(json.dumps({"config": cfg, "k": k, "m": m, "how_many": synth.HOW_MANY,
                  "what": "closed loop: every client thread sends its next evolving session when the previous answer is back (PCIe copies included)",
                  "reference_claim": "README.md:17 -- < 7 ms p90 at 1000+ requests/s on 2 vCPU (whole HTTP request)", "runs": runs}, indent=1))

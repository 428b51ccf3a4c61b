"""One query of config 5 through every kernel path (debugging aid): fast kernel, general kernel, general kernel without the sketch filter (debug outputs)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import serenade_amd as sa
from serenade_amd import synth, capi
cfg = sys.argv[1]; qsel = [int(x) for x in sys.argv[2].split(",")]
inter, n_items, k, m, idfw = synth.CONFIGS[cfg]
off, items, ts = synth.training_sessions(inter, n_items)
full = sa.VMISIndex.from_sessions(off, items, ts, m, 34, idfw, device=0, builder="gpu")
B = 1 << 18
qi, qo = synth.queries(int(B / 3.0) + 4096, n_items, seed=synth.SEED + 7919, max_items=synth.LAST_ITEMS)
n = synth.HOW_MANY
for q in qsel:
    f = qi[qo[q]:qo[q + 1]].copy(); o = np.array([0, len(f)], np.uint32)
    # a batch of 32 copies so that the batched (not the latency) path runs
    ff = np.tile(f, 32); oo = (np.arange(33) * len(f)).astype(np.uint32)
    print("query", q, f)
    for name, env in (("fast", {}), ("general", {"SRN_NO_FAST": "1"}), ("general, hash table", {"SRN_NO_FAST": "1", "SRN_NO_MERGE": "1"}),
                      ("hash, no sketch", {"SRN_NO_FAST": "1", "SRN_NO_MERGE": "1", "SRN_SKETCH_SLOTS": "0"}), ("hash, no hot", {"SRN_NO_FAST": "1", "SRN_NO_MERGE": "1", "SRN_HOT_SLOTS": "0"}),
                      ("merge, no sketch", {"SRN_NO_FAST": "1", "SRN_SKETCH_SLOTS": "0"})):
        for kk, vv in env.items(): os.environ[kk] = vv
        capi.reload_knobs()
        ids, sc, cnt = sa.predict_batch(full, (ff, oo), k, m, n)
        top = ids[:, 0]
        print(" %-22s" % name, full.last_path_counts(), cnt[0], ids[0][:3], np.round(sc[0][:3], 3), "copies whose top item is", top[0], ":", int((top == top[0]).sum()), "others:", np.unique(top[top != top[0]]))
        for kk in env: del os.environ[kk]
        capi.reload_knobs()
    dbg = sa.predict_batch_debug(full, (ff, oo), k, m, n, False, neighbours=True)
    print(" %-22s" % "debug (no filter)", dbg["counts"][0], dbg["ids"][0][:3], np.round(dbg["scores"][0][:3], 3), "stats", dbg["stats"][0], "stats[7] of the copies", dbg["stats"][:, 7], "tops", np.unique(dbg["ids"][:, 0]))
    os.environ["SRN_NO_MERGE"] = "1"; capi.reload_knobs()
    dbg = sa.predict_batch_debug(full, (ff, oo), k, m, n, False, neighbours=True)
    print(" %-22s" % "debug, hash table", dbg["counts"][0], dbg["ids"][0][:3], np.round(dbg["scores"][0][:3], 3), "stats", dbg["stats"][0], "stats[7] of the copies", dbg["stats"][:, 7], "tops", np.unique(dbg["ids"][:, 0]))
    del os.environ["SRN_NO_MERGE"]; capi.reload_knobs()
    for it in f:
        post = full.postings(int(it)) if hasattr(full, "postings") else None
        print("   item", it, "postings:", None if post is None else (len(post[0]), post[1]))

"""Device-resident batches of a few sizes, to be run under `rocprofv3 --kernel-trace` (the launch sequence of a mid-size batch: where the time between
the first kernel's start and the last kernel's end goes).  usage: python tools/batch_trace.py [sizes comma separated] [calls]
then: python tools/batch_trace.py --analyze <dir>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2 and sys.argv[1] == "--analyze":
    import csv, glob
    rows = []
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("srn::", "")[:44]) for r in rows)
    # iterations: separated by gaps > 150 us of no kernel activity after a marker-free heuristic: a new iteration starts at every vmis_prep_kernel
    its, cur = [], []
    for k in ks:
        if "prep" in k[2] and "shard" not in k[2] and cur: its.append(cur); cur = []
        cur.append(k)
    if cur: its.append(cur)
    by = {}
    for it in its: by.setdefault(len(it), []).append(it)
    print("%d iterations" % len(its))
    # group iterations by fast-kernel duration magnitude (= batch size); print the median one of every group
    groups = {}
    for it in its:
        f = [k for k in it if "fast" in k[2]]
        key = round((sum(k[1] - k[0] for k in f) / 1e3) ** 0.5) if f else -1
        groups.setdefault(key, []).append(it)
    for key, g in sorted(groups.items()):
        if len(g) < 5: continue
        g.sort(key=lambda it: it[-1][1] - it[0][0]); it = g[len(g) // 2]
        t0 = it[0][0]
        print("group of %d iterations, median span %.1f us (first start -> last end), kernel sum %.1f us:" % (len(g), (it[-1][1] - t0) / 1e3, sum(k[1] - k[0] for k in it) / 1e3))
        prev = t0
        for s, e, n in it:
            print("   +%8.1f us  gap %6.1f  dur %8.1f  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, n)); prev = max(prev, e)
    sys.exit(0)
import numpy as np, torch
import serenade_amd as sa
from serenade_amd import synth
sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "4096,65536").split(",")]
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 40
inter, n_items, k, m, idfw = synth.CONFIGS[os.environ.get("SRN_TRACE_CFG", "cfg3")]
off, items, ts = synth.training_sessions(inter, n_items)
ix = sa.VMISIndex.from_sessions(off, items, ts, m, 34, idfw, builder="gpu")
qi, qo = synth.queries(int(max(sizes) / 3.0) + 4096, n_items, seed=synth.SEED + 7919, max_items=synth.LAST_ITEMS)
dev = torch.device("cuda:0")
for nq in sizes:
    f, o = qi[:qo[nq]], qo[:nq + 1]
    d_f = torch.from_numpy(f.view(np.int64).copy()).to(dev); d_o = torch.from_numpy(o.astype(np.int32)).to(dev)
    ts_ = []
    o_ids = torch.zeros(nq * synth.HOW_MANY, dtype=torch.int64, device=dev); o_sc = torch.zeros(nq * synth.HOW_MANY, dtype=torch.float64, device=dev); o_cnt = torch.zeros(nq, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    if nq <= 256:   # the zero-copy latency path (host pointers)
        out = None
        for c in range(calls * 5):
            t0 = time.perf_counter(); out = sa.predict_batch(ix, (f, o), k, m, synth.HOW_MANY, False, out=out); ts_.append((time.perf_counter() - t0) * 1e3)
        ts_.sort(); print("batch %d (host pointers, latency path): p50 %.4f ms  min %.4f ms" % (nq, ts_[len(ts_) // 2], ts_[0]))
        continue
    for c in range(calls):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sa.predict_batch_device(ix, d_f.data_ptr(), d_o.data_ptr(), nq, synth.LAST_ITEMS, k, m, synth.HOW_MANY, False, o_ids.data_ptr(), o_sc.data_ptr(), o_cnt.data_ptr(), st)
        torch.cuda.synchronize(); ts_.append((time.perf_counter() - t0) * 1e3)
    ts_.sort(); print("batch %d: p50 %.3f ms  min %.3f ms" % (nq, ts_[len(ts_) // 2], ts_[0]))

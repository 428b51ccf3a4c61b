#!/bin/bash
# Where a single srn_predict call's ~50 us go: rocprofv3 --kernel-trace of tools/latency_probe.py; per call, the kernels' durations and the gaps between them
# (start of the first kernel .. end of the last).  -> gpurun_out/latency_trace.txt
R=$PWD; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/lt -o lt --output-format csv -- python $R/tools/latency_probe.py cfg3 > $R/gpurun_out/lt.log 2>&1
cd $R
python - <<'PY' > gpurun_out/latency_trace.txt
import csv, glob, collections
rows = []
for f in glob.glob("gpurun_out/lt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
rows.sort()
# group into calls: a gap of > 15 us between kernels starts a new call
calls, cur = [], []
for s, e, n in rows:
    if cur and s - cur[-1][1] > 15000: calls.append(cur); cur = []
    cur.append((s, e, n))
if cur: calls.append(cur)
sig = collections.Counter(tuple(n for _, _, n in c) for c in calls)
for names, cnt in sig.most_common(4):
    sel = [c for c in calls if tuple(n for _, _, n in c) == names][20:]
    if not sel: continue
    import statistics as st
    print("%d calls with %d launches:" % (cnt, len(names)))
    for i, n in enumerate(names):
        d = st.median((c[i][1] - c[i][0]) / 1e3 for c in sel)
        g = st.median((c[i][0] - c[i - 1][1]) / 1e3 for c in sel) if i else 0.0
        print("   gap %5.1f us | %-60s %6.1f us" % (g, n, d))
    print("   first start .. last end: median %.1f us" % st.median((c[-1][1] - c[0][0]) / 1e3 for c in sel))
PY
rm -rf gpurun_out/lt; cat gpurun_out/latency_trace.txt; tail -3 gpurun_out/lt.log

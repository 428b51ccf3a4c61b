"""Overlap analysis of a rocprofv3 --kernel-trace --memory-copy-trace run (csv): for the last `window_ms` of activity, the busy time of kernels, of
copies per direction, their union, and a coarse timeline.  usage: python tools/trace_overlap.py <dir> [window_ms]"""
import csv, glob, os, sys
d = sys.argv[1]; win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 45e6
def rows(pat):
    out = []
    for f in glob.glob(os.path.join(d, "**", pat), recursive=True):
        out += list(csv.DictReader(open(f)))
    return out
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in rows("*kernel_trace.csv")]
cs = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", "?"), r.get("Stream_Id", "?")) for r in rows("*memory_copy_trace.csv")]
end = max([k[1] for k in ks] + [c[1] for c in cs]); lo = end - win
ks = [k for k in ks if k[1] > lo]; cs = [c for c in cs if c[1] > lo]
def union(iv):
    iv = sorted(iv); tot = 0; cur_s, cur_e = None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None: tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    if cur_e is not None: tot += cur_e - cur_s
    return tot
kb = union([(k[0], k[1]) for k in ks]); 
print("window %.1f ms: kernels busy %.2f ms (%d launches, sum %.2f ms)" % (win / 1e6, kb / 1e6, len(ks), sum(k[1] - k[0] for k in ks) / 1e6))
for dname in sorted(set(c[2] for c in cs)):
    sel = [(c[0], c[1]) for c in cs if c[2] == dname]
    print("  copies %-22s busy %.2f ms (%d copies, sum %.2f ms)" % (dname, union(sel) / 1e6, len(sel), sum(e - s for s, e in sel) / 1e6))
allb = union([(k[0], k[1]) for k in ks] + [(c[0], c[1]) for c in cs])
print("  union of kernels and copies %.2f ms; overlap of copies with kernels %.2f ms" % (allb / 1e6, (kb + union([(c[0], c[1]) for c in cs]) - allb) / 1e6))
by = {}
for k in ks: by.setdefault((k[2], k[3], k[4]), []).append(k[1] - k[0])
for key, v in sorted(by.items(), key=lambda x: -sum(x[1]))[:12]:
    print("  %-42s queue %s stream %s: %d x avg %.3f ms" % (key[0], key[1], key[2], len(v), sum(v) / len(v) / 1e6))
ev = sorted([(k[0], k[1], "K:" + k[2][:18] + "/q" + str(k[3])) for k in ks if k[1] - k[0] > 200e3] + [(c[0], c[1], "C:" + c[2][:14]) for c in cs if c[1] - c[0] > 100e3])
print("timeline (events > 0.1-0.2 ms), ms relative to window start:")
for s, e, n in ev[-60:]:
    print("   %8.2f .. %8.2f  %s" % ((s - lo) / 1e6, (e - lo) / 1e6, n))

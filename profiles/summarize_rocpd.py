#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace: per-kernel totals like `--stats`, plus the
per-launch durations of the full-batch vmis_predict_kernel launches (grid >= 1024 workgroups) that bench.py times.
Usage: summarize_rocpd.py <results.db>  > profiles/<name>.txt"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
print("# rocprofv3 --kernel-trace --stats summary (from %s)" % sys.argv[1].split("/")[-1])
print("%-110s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
for name, calls, total, avg, pct in db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
    print("%-110s %8d %14.1f %12.2f %6.2f%%" % (name[:110], calls, total, avg, pct))
rows = db.execute("select grid_x, workgroup_x, lds_size, vgpr_count, sgpr_count, scratch_size, (end - start) / 1e6 from kernels "
                  "where name like '%vmis_predict_kernel%' and name like '%int, false%' and grid_x >= 1024 * workgroup_x order by start").fetchall()
if rows:
    big = max(r[6] for r in rows)
    small = [r for r in rows if r[6] < 0.1 * big]
    if small:
        print("\n# second-tier launches (what the small LDS geometry could not hold): %d, avg %.3f ms" % (len(small), sum(r[6] for r in small) / len(small)))
    rows = [r for r in rows if r[6] >= 0.1 * big]
    ms = [r[6] for r in rows]
    med = sorted(ms)[len(ms) // 2]
    cut = next((i for i, v in enumerate(ms) if v > 2.0 * med), len(ms))   # bench.py's stats pass: debug counters on, sketch pre-filter off
    if cut < len(ms):
        print("\n# (launch %d is bench.py's stats pass, not a timed step: %.3f ms; the %d launches after it are the host-buffer batch calls "
              "of the latency section: %s)" % (cut + 1, ms[cut], len(ms) - cut - 1, " ".join("%.3f" % v for v in ms[cut + 1:])))
    ms = ms[:cut]
    print("# timed and warm-up launches of vmis_predict_kernel<..., GLOBAL_TABLES=false, ...>: %d" % len(ms))
    print("grid_threads=%d workgroup=%d lds_bytes=%d vgpr=%d sgpr=%d scratch=%d" % rows[0][:6])
    print("duration_ms: " + " ".join("%.3f" % v for v in ms))
    print("avg_ms=%.3f min_ms=%.3f max_ms=%.3f" % (sum(ms) / len(ms), min(ms), max(ms)))

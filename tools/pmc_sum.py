"""Sum rocprofv3 --pmc csv counters per predict kernel: python tools/pmc_sum.py <dir> [queries per dispatch]"""
import csv, glob, sys, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        k = "sback" if "vmis_shard_back_kernel" in kn else "fast" if "vmis_fast_kernel" in kn else "general" if "vmis_predict_kernel" in kn else "prep" if "vmis_prep" in kn else None
        if k is None: continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in sorted(tot):
    for c in sorted(tot[k]): print("%-8s %-28s %16.0f  (%d dispatches)  per dispatch %14.0f" % (k, c, tot[k][c], n[k][c], tot[k][c] / max(1, n[k][c])))

"""Sum rocprofv3 --pmc csv counters for the predict kernel: python tools/pmc_sum.py <dir>"""
import csv, glob, sys, collections
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "vmis_predict_kernel" not in r["Kernel_Name"]: continue
        tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(tot): print("%-28s %16.0f  (%d dispatches)" % (k, tot[k], n[k]))

"""HBM traffic of the predict kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over `bench.py --steps 2 --warmup 1`
(tools/pmc_bench.sh fetch FETCH_SIZE; tools/pmc_bench.sh write WRITE_SIZE):  python tools/pmc_traffic.py <fetch_dir> <write_dir> <config> <batch> [launches = warm-up + steps, default 3] > profiles/rNN_traffic_<config>.json"""
import csv, glob, json, sys

def per_launch(d, counter, min_grid_wg=1024):
    vals = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "vmis_predict_kernel" in r["Kernel_Name"] and "int, false" in r["Kernel_Name"] and r["Counter_Name"] == counter \
                    and int(r["Grid_Size"]) >= min_grid_wg * int(r["Workgroup_Size"]):
                vals.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    vals.sort()
    top = max(v for _, v in vals)
    n_timed = int(sys.argv[5]) if len(sys.argv) > 5 else 3   # warm-up + timed steps of the profiled command; then come the stats pass and the host-buffer batch calls
    big = [v for _, v in vals if v >= 0.1 * top][:n_timed]
    small = [v for _, v in vals if v < 0.1 * top][:n_timed]    # second-tier launches (what the small LDS geometry could not hold), one per step
    return [b + (sum(small) / len(small) if small else 0.0) for b in big]

fetch, write = per_launch(sys.argv[1], "FETCH_SIZE"), per_launch(sys.argv[2], "WRITE_SIZE")
f, w = sum(fetch) / len(fetch), sum(write) / len(write)
print(json.dumps({"config": sys.argv[3], "batch_per_gpu": int(sys.argv[4]), "kernel": "vmis_predict_kernel<512,u32,false,0,true>",
                  "launches": {"FETCH_SIZE": len(fetch), "WRITE_SIZE": len(write)},
                  "FETCH_SIZE_kb_per_launch": f, "WRITE_SIZE_kb_per_launch": w, "fetch_correction": 2.0,
                  "traffic_bytes_per_launch": (2.0 * f + w) * 1024.0,
                  "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (kernel-trace only) over `bench.py --steps 2 --warmup 1` "
                            "(tools/pmc_bench.sh), full-batch launches only; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B; "
                            "re-verified with tools/fetch_calib.hip), WRITE_SIZE exact"}, indent=1))

"""Turns the raw rocprofv3 csv output of tools/archive/r02_profiles.sh into the tracked summaries under profiles/ (round 2).
python tools/rocprof_summarize.py kernel_trace <dir>            > profiles/r02_kernel_trace_cfg3.txt
python tools/rocprof_summarize.py traffic <fetch_dir> <write_dir> <config> <batch> > profiles/r02_traffic_cfg3.json
python tools/rocprof_summarize.py sq <dir_a> <dir_b> <queries per dispatch> <phase log> > profiles/r02_sq_counters_cfg3.json"""
import csv, glob, json, re, sys, collections


def short(n):
    for k in ("vmis_fast_kernel", "vmis_finish_big_kernel", "vmis_finish_kernel", "vmis_prep_kernel", "vmis_predict_kernel", "rows_to_packed_kernel", "rows_to_slots_kernel"):
        if k in n:
            if k == "vmis_predict_kernel":
                return "vmis_predict_kernel<global tables>" if "unsigned int, true" in n else "vmis_predict_kernel (general kernel)"
            return k
    return None


def kernel_trace(d):
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    per = collections.defaultdict(list)
    for r in rows:
        k = short(r["Kernel_Name"])
        if k:
            per[k].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, int(r["Grid_Size_X"]) if "Grid_Size_X" in r else 0, r))
    print("# rocprofv3 --kernel-trace --stats of `python bench.py --steps 10 --warmup 3 --no-sweep --no-cpu-baseline` (config 3, 2^20 queries per step)")
    print("%-44s %6s %12s %10s %10s %10s" % ("kernel", "calls", "total_ms", "avg_ms", "min_ms", "max_ms"))
    tot = sum(sum(x[1] for x in v) for v in per.values())
    for k, v in sorted(per.items(), key=lambda kv: -sum(x[1] for x in kv[1])):
        ms = [x[1] for x in v]
        print("%-44s %6d %12.3f %10.4f %10.4f %10.4f  %5.1f%%" % (k, len(ms), sum(ms), sum(ms) / len(ms), min(ms), max(ms), 100 * sum(ms) / tot))
    fk = sorted(per.get("vmis_fast_kernel", []))
    if fk:
        big = max(x[1] for x in fk)
        full = [x for x in fk if x[1] > 0.5 * big]
        r0 = full[0][3]
        print("\n# full-batch launches of vmis_fast_kernel (the 13 warm-up + timed steps; smaller launches: the parity gate / stats sample)")
        print("grid=%s workgroup=%s lds_bytes=%s vgpr=%s sgpr=%s scratch=%s" % (r0.get("Grid_Size_X", r0.get("Grid_Size")), r0.get("Workgroup_Size_X", r0.get("Workgroup_Size")),
              r0.get("LDS_Block_Size"), r0.get("VGPR_Count"), r0.get("SGPR_Count"), r0.get("Scratch_Size")))
        ms = [x[1] for x in full]
        print("duration_ms: " + " ".join("%.3f" % v for v in ms))
        print("avg_ms=%.3f min_ms=%.3f max_ms=%.3f" % (sum(ms) / len(ms), min(ms), max(ms)))


def counters(d, want_kernel):
    out = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if short(r["Kernel_Name"]) == want_kernel:
                out[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), int(r["Grid_Size"])))
    return out


def traffic(fd, wd, config, batch):
    res = {}
    for name, d, c in (("FETCH_SIZE", fd, "FETCH_SIZE"), ("WRITE_SIZE", wd, "WRITE_SIZE")):
        per_kernel = {}
        for k in ("vmis_fast_kernel", "vmis_finish_kernel", "vmis_prep_kernel", "vmis_predict_kernel (general kernel)", "vmis_finish_big_kernel"):
            v = counters(d, k).get(c, [])
            if not v:
                continue
            top = max(g for _, _, g in v)
            full = [x for _, x, g in sorted(v) if g >= 0.5 * top]
            # the general kernel's full-grid launches are the handed-over passes (its grid is sized for the batch, most workgroups exit at once)
            per_kernel[k] = sum(full[:3]) / max(1, len(full[:3]))
        res[name] = per_kernel
    f_fast, w_fast = res["FETCH_SIZE"].get("vmis_fast_kernel", 0.0), res["WRITE_SIZE"].get("vmis_fast_kernel", 0.0)
    f_all, w_all = sum(res["FETCH_SIZE"].values()), sum(res["WRITE_SIZE"].values())
    print(json.dumps({"config": config, "batch_per_gpu": int(batch), "kernel": "vmis_fast_kernel<3>",
                      "FETCH_SIZE_kb_per_launch": f_fast, "WRITE_SIZE_kb_per_launch": w_fast, "fetch_correction": 2.0,
                      "traffic_bytes_per_launch": (2.0 * f_fast + w_fast) * 1024.0,
                      "all_kernels_of_a_step": {"FETCH_SIZE_kb": res["FETCH_SIZE"], "WRITE_SIZE_kb": res["WRITE_SIZE"], "traffic_bytes_per_step": (2.0 * f_all + w_all) * 1024.0},
                      "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (kernel trace only) over `bench.py --steps 2 --warmup 1 --no-sweep "
                                "--no-cpu-baseline`, the three full-batch launches of each kernel averaged; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies "
                                "128-B requests at 64 B; re-verified in round 1 with tools/fetch_calib.hip), WRITE_SIZE as reported"}, indent=1))


def sq(da, db, nq, phase_log):
    nq = float(nq)
    c = {}
    for d in (da, db):
        for k, v in counters(d, "vmis_fast_kernel").items():
            top = max(g for _, _, g in v)
            vals = [x for _, x, g in v if g >= 0.5 * top]   # (the full-batch launches of the LEAN instantiation: the MID / BIG launches behind it carry the same kernel name with a small grid --
            c[k] = sum(vals) / len(vals)                       #  until round 5 they were averaged in, which halved every per-query figure of this summary)
    wave_cycles = c.get("SQ_WAVE_CYCLES", 0.0)
    out = {"kernel": "vmis_fast_kernel<3> (512 threads, 80 VGPRs, 48 KB LDS: 3 workgroups = 6 waves per SIMD)", "queries_per_dispatch": int(nq),
           "per_dispatch": c,
           "per_query": {k: c[k] / nq for k in c},
           "derived": {
               "valu_busy_per_simd": 6.0 * c.get("SQ_ACTIVE_INST_VALU", 0) / wave_cycles if wave_cycles else None,
               "salu_issue_share_of_wave_time": c.get("SQ_INST_CYCLES_SALU", 0) / wave_cycles if wave_cycles else None,
               "wave_time_parked_waitcnt_or_barrier": c.get("SQ_WAIT_ANY", 0) / wave_cycles if wave_cycles else None,
               "wave_time_waiting_for_issue": c.get("SQ_WAIT_INST_ANY", 0) / wave_cycles if wave_cycles else None,
               "wave_time_issuing": c.get("SQ_ACTIVE_INST_ANY", 0) / wave_cycles if wave_cycles else None,
               "lds_bank_conflict_share_of_lds_cycles": c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"] if c.get("SQ_LDS_IDX_ACTIVE") else None,
               # a CU retires a query every (a workgroup's wave time per query = SQ_WAVE_CYCLES x 4 / 8 waves) / 3 resident workgroups; its one LDS pipe is active SQ_LDS_IDX_ACTIVE of them
               "lds_pipe_busy": c["SQ_LDS_IDX_ACTIVE"] / (wave_cycles * 4.0 / 8.0 / 3.0) if c.get("SQ_LDS_IDX_ACTIVE") and wave_cycles else None,
               "instruction_cache_miss_rate": c.get("SQ_IFETCH_LEVEL", None),
               "note": "SQ_WAVE_CYCLES, SQ_ACTIVE_INST_*, SQ_WAIT_* count quad-cycles summed over waves; valu_busy = 6 waves per SIMD x the per-wave share"},
           "phase_cycles_per_query": {}, "method": "rocprofv3 --pmc in two passes (kernel trace only) over tools/count_run.py (three launches of the given batch on config 3); "
                                                   "phase cycles: srn_debug_phase_cycles (shader clock of thread 0 of every workgroup, summed, divided by the queries), tools/phase_profile.py"}
    for l in open(phase_log):
        m = re.match(r"\s+(\d+) (.*?)\s+([\d.]+)%\s+(\d+) cyc/query", l)
        if m and int(m.group(4)) > 5:
            out["phase_cycles_per_query"]["tick %s" % m.group(1)] = int(m.group(4))
        if l.startswith("fast kernel:") or l.startswith("path counts") or l.startswith("main "):
            out.setdefault("phase_profile_lines", []).append(l.strip())
    out["phase_legend"] = {"0": "record + barrier", "1": "stage lists", "3": "merge tree (fast kernel, since round 3)", "2": "m-cut (+ merge tree before round 3)", "4": "k-cut", "8": "row requests + clears", "9": "walk A", "10": "phase 4a (threshold, floors)",
                           "11": "walk B (list)", "12": "resolve the hit list", "13": "hand-off record"}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    {"kernel_trace": kernel_trace, "traffic": traffic, "sq": sq}[sys.argv[1]](*sys.argv[2:])

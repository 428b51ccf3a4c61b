#!/usr/bin/env python3
"""bench.py -- VMIS-kNN predict_next throughput on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path (libserenade_hip.so: srn_predict_batch_device) over one batch of
synthetic evolving sessions, with the index and the query / result buffers already resident in HBM.
At N=1 the workload is BASELINE.json configs[2] -- the 60 M-interaction / 1.76 M-item synthetic index the
metric's target is quoted on (k=1500, m=2500, idf_weighting=2).  With N>1 (launched by torch.distributed.run,
one rank per GPU) every rank holds the full index and serves its own slice of the query stream: queries are
independent, so the path shards by query with no data-path collective (weak scaling, DESIGN.md "Multi-GPU").

Prints ONE JSON line on rank 0:
  value      whole-job predict_next queries/s = N * K * batch / max-over-ranks wall time of the K timed steps
  roofline   dominant kernel (vmis_predict_kernel): algorithmic bytes per launch / its HIP-event duration
  cpu_baseline  the oracle's literal restatement of the reference CPU path, all host cores, bounded sample
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def algorithmic_bytes(stats):
    """B(q) = 4P + 4C + 8K + 4I + 8D + 16H + 8L, every datum counted once in the compact layout."""
    s = stats.astype(np.float64)
    return 4 * s[:, 0] + 4 * s[:, 1] + 8 * s[:, 2] + 4 * s[:, 3] + 8 * s[:, 4] + 16 * s[:, 5] + 8 * s[:, 6]


def usable_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return max(1, n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="cfg3", help="tiny | cfg2 | cfg3 | cfg4 (synth.CONFIGS)")
    ap.add_argument("--batch", type=int, default=131072, help="evolving sessions per step and per GPU")
    ap.add_argument("--pool", type=int, default=4, help="distinct query batches cycled through")
    ap.add_argument("--mode", default="replicas", choices=["replicas", "item-sharded"],
                    help="replicas: every GPU holds the index and serves its own queries (default, no data-path collective); "
                         "item-sharded: the north-star capacity mode, index split by item over the GPUs, 3 RCCL collectives per batch")
    ap.add_argument("--builder", default="gpu", choices=["gpu", "host"], help="index construction: rocPRIM sorts on the GPU, or the host builder (same bytes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the baseline sample")
    ap.add_argument("--traffic-file", default=None, help="profiles/*_traffic_<config>.json from the PMC passes (default: newest match)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print("bench.py: --gpus %d needs torch.distributed.run with %d ranks" % (args.gpus, args.gpus), file=sys.stderr)
            sys.exit(2)
        args.gpus = world

    import torch
    import torch.distributed as dist
    import serenade_amd as sa
    from serenade_amd import distributed as D
    from serenade_amd import synth

    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the predict path has no CPU fallback", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    D.init("nccl", dev)   # RCCL; control plane only (barrier + max-over-ranks), the data path has no collective

    inter, n_items, k, m, idfw = synth.CONFIGS[args.config]
    how_many, last_items = synth.HOW_MANY, synth.LAST_ITEMS
    t0 = time.time()
    off, items, ts = synth.training_sessions(inter, n_items)
    t_gen = time.time() - t0
    t0 = time.time()
    sharded_mode = args.mode == "item-sharded"
    if sharded_mode:
        from serenade_amd import sharded as SH
        index = SH.ShardedVMISIndex(off, items, ts, m, 34, idfw, rank, world, device=local_rank)
        comm = SH.DistComm() if world > 1 else SH.SoloComm()
        if args.batch == 131072:
            args.batch = 16384          # the exchange buffers are m * 4 B per query and shard
    else:
        index = sa.VMISIndex.from_sessions(off, items, ts, m, 34, idfw, device=local_rank, builder=args.builder)
    t_build = time.time() - t0
    info = index.info

    # ---- this rank's slice of the query stream: `pool` distinct batches of `batch` evolving sessions ----
    B = args.batch
    need = B * args.pool
    n_sess = max(1024, int(need / 3.2) + 4096)
    while True:
        # replicas: every rank draws its own slice of the query stream; item-sharded: all ranks see the same batch
        q_items, q_off = synth.queries(n_sess, n_items, seed=synth.SEED + 7919 * ((0 if sharded_mode else rank) + 1), max_items=last_items)
        if len(q_off) - 1 >= need:
            break
        n_sess = int(n_sess * 1.5)
    batches = []
    for b in range(args.pool):
        lo, hi = b * B, (b + 1) * B
        fo = q_off[lo:hi + 1].astype(np.int64)
        flat = q_items[fo[0]:fo[-1]]
        qo = (fo - fo[0]).astype(np.uint32)
        batches.append((torch.from_numpy(flat.view(np.int64).copy()).to(dev), torch.from_numpy(qo.view(np.int32).copy()).to(dev), flat, qo))
    out_ids = torch.zeros(B * how_many, dtype=torch.int64, device=dev)
    out_sc = torch.zeros(B * how_many, dtype=torch.float64, device=dev)
    out_cnt = torch.zeros(B, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream()

    def step(i):
        d_flat, d_off, _, _ = batches[i % args.pool]
        if sharded_mode:
            res = SH.predict_batch_sharded(index, comm, d_flat, d_off, B, last_items, k, m, how_many, False, stream.cuda_stream)
            out_cnt.copy_(res[2])
        else:
            sa.predict_batch_device(index, d_flat.data_ptr(), d_off.data_ptr(), B, last_items, k, m, how_many, False,
                                    out_ids.data_ptr(), out_sc.data_ptr(), out_cnt.data_ptr(), stream.cuda_stream)

    barrier = D.barrier

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record(stream)
        step(args.warmup + i)
        ev[i][1].record(stream)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = D.max_over_ranks(elapsed, dev)
    step_ms = np.array([a.elapsed_time(b) for a, b in ev])
    served = int((out_cnt.cpu().numpy().view(np.uint32) != 0xFFFFFFFF).sum())
    if sharded_mode:
        if rank == 0:
            print(json.dumps({"metric": "predict_next queries/sec", "value": args.steps * B / elapsed, "unit": "queries/s", "n_gpus": args.gpus,
                              "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
                              "scaling": "strong", "vs_baseline": None, "dtype": "u32 ids / i32 accumulators / f64 scores", "data": "synthetic",
                              "config": {"workload": "synth.CONFIGS[%s], index item-sharded over %d GPU(s), every rank sees the whole batch" % (args.config, world),
                                         "name": args.config, "batch": B, "parallelism": "item-sharded x%d: all-gather + all-reduce(min) + all-gather per batch" % world,
                                         "items_on_rank0": int(info["n_items"]), "index_bytes_hbm_rank0": int(info["device_bytes"])},
                              "roofline": None, "cpu_baseline": None, "queries_served_last_step": served,
                              "note": "capacity mode; the headline bench line is --mode replicas"}))
        if world > 1:
            dist.destroy_process_group()
        return
    k_main, k_retry = index.kernel_times(min(64, args.steps))

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel: algorithmic bytes per launch / measured launch duration -------
    # per-query counters come from the kernel's own stats output (validated against the oracle in tests/)
    nstat = min(B, 32768)
    _, _, flat0, qo0 = batches[0]
    dbg = sa.predict_batch_debug(index, (flat0[:qo0[nstat]], qo0[:nstat + 1]), k, m, how_many, False, neighbours=False)
    bq = algorithmic_bytes(dbg["stats"])
    bytes_per_launch = float(bq.mean()) * B
    kernel_ms = float(k_main.mean()) if len(k_main) else float("nan")
    achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
    retried = int((dbg["stats"][:, 7] == 1).sum())

    # ---- single-query latency through the host-pointer entry point (the reference's call shape) ---------
    lat = []
    for i in range(200):
        q = flat0[qo0[i]:qo0[i + 1]]
        t1 = time.perf_counter()
        sa.predict(index, q, k, m, how_many, False)
        lat.append((time.perf_counter() - t1) * 1e6)
    lat = np.array(lat[20:])
    # the same batch through the host-pointer batch entry point: uploads, launches, downloads (never the headline value)
    host_ms = []
    for i in range(4):
        t1 = time.perf_counter()
        sa.predict_batch(index, (flat0[:qo0[B]], qo0[:B + 1]), k, m, how_many, False)
        host_ms.append((time.perf_counter() - t1) * 1e3)
    host_ms = float(np.median(host_ms[1:]))

    # HBM traffic per launch comes from separate rocprofv3 --pmc passes over this same command (tools/pmc_bench.sh);
    # the committed summary is read back here so that the line carries it (null if no summary matches the workload)
    traffic, traffic_src = None, None
    try:
        import glob
        cand = [args.traffic_file] if args.traffic_file else sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic_%s.json" % args.config)))
        if cand:
            tj = json.load(open(cand[-1]))
            if tj.get("config") == args.config and tj.get("batch_per_gpu") == B:
                traffic, traffic_src = tj["traffic_bytes_per_launch"], os.path.relpath(cand[-1], ROOT)
    except Exception:
        pass

    result = {
        "metric": "predict_next queries/sec", "value": args.gpus * args.steps * B / elapsed, "unit": "queries/s",
        "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32 ids / i32 accumulators / f64 scores",
        "data": "synthetic",
        "config": {"workload": "BASELINE configs[2]: synthetic %d interactions / %d items, k=%d m=%d idf_weighting=%g "
                               "last_items=%d how_many=%d" % (inter, n_items, k, m, idfw, last_items, how_many)
                               if args.config == "cfg3" else "synth.CONFIGS[%s]" % args.config,
                   "name": args.config, "batch_per_gpu": B, "query_pool_batches": args.pool,
                   "sessions": int(info["n_sessions_kept"]), "items": int(info["n_items"]), "interactions": int(info["nnz_rows"]),
                   "posting_entries": int(info["nnz_postings"]), "index_bytes_hbm": int(info["device_bytes"]),
                   "parallelism": "query-sharded replicas x%d (no data-path collective)" % args.gpus,
                   "setup_s": {"generate": round(t_gen, 2), "index_build_upload": round(t_build, 2), "index_builder": "item-sharded host" if sharded_mode else args.builder}},
        "roofline": {"bound": "hbm", "kernel": "vmis_predict_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_unit": "bytes per launch (2*FETCH_SIZE + WRITE_SIZE)",
                     "traffic_source": traffic_src, "algorithmic_bytes_per_launch": bytes_per_launch,
                     "algorithmic_bytes_per_query": float(bq.mean()), "queries_per_launch": B,
                     "kernel_ms_avg": kernel_ms, "kernel_ms_min": float(k_main.min()) if len(k_main) else None,
                     "retry_pass_ms_avg": float(k_retry.mean()) if len(k_retry) else None,
                     "queries_via_global_table_pass_in_sample": retried, "stats_sample_queries": nstat},
        "latency": {"batch_ms_p50": float(np.percentile(step_ms, 50)), "batch_ms_p90": float(np.percentile(step_ms, 90)),
                    "single_query_us_p50": float(np.percentile(lat, 50)), "single_query_us_p90": float(np.percentile(lat, 90)),
                    "host_batch_ms": host_ms, "host_batch_queries_per_s": B / (host_ms * 1e-3),
                    "note": "single_query = srn_predict (host pointers, one evolving session per call, PCIe-inclusive); "
                            "host_batch = srn_predict_batch on host buffers (pageable numpy arrays: upload + launches + download)"},
        "queries_served_last_step": served,
    }

    if args.gpus == 1 and not args.no_cpu_baseline:
        # the oracle is used here ONLY as the timed CPU baseline (literal restatement of the reference loops)
        from oracle import oracle as O
        t1 = time.time()
        oix = O.OracleIndex(off, items, ts, m, 34, idfw, fast=True)
        t_obuild = time.time() - t1
        cores = usable_cores()
        probe_n = min(B, 64 * cores)
        r = oix.predict_batch("literal", flat0[:qo0[probe_n]], qo0[:probe_n + 1], k, m, how_many, False, threads=cores, want_results=False)
        rate = probe_n / max(r["elapsed"], 1e-6)
        n_cpu = int(min(B, max(probe_n, rate * args.cpu_seconds)))
        r = oix.predict_batch("literal", flat0[:qo0[n_cpu]], qo0[:n_cpu + 1], k, m, how_many, False, threads=cores,
                              want_results=False, want_latency=True)
        lat_cpu = r["lat_us"]
        result["cpu_baseline"] = {
            "value": n_cpu / r["elapsed"], "unit": "queries/s", "cores": cores, "kind": "port",
            "sample": "first %d queries of the same batch, %d threads over one shared read-only index (%.1f s); "
                      "oracle/vmis_oracle.cpp literal restatement of the reference's Rust loops, not the Rust binary"
                      % (n_cpu, cores, r["elapsed"]),
            "per_call_us_p50": float(np.percentile(lat_cpu, 50)), "per_call_us_p90": float(np.percentile(lat_cpu, 90)),
            "index_build_s": round(t_obuild, 2)}
    print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

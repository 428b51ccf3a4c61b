"""Where do host-pointer batches of 4 096 .. 65 536 queries spend their time beyond the resident call?  Up to 65 536 queries srn_predict_batch is ONE chunk:
queries user buffer -> pinned staging -> HBM, the launch sequence, results HBM -> pinned staging -> the caller's (pageable) buffers, in that order -- nothing of a chunk's
results exists before its finish kernels have run.  This tool times, on the same box and process: the resident call; the host-pointer call (with the library's own stage
trace, SRN_HOST_TRACE); and the raw pieces a one-chunk call cannot overlap with anything -- the upload of the queries, the download of the result rows from HBM to pinned
memory (torch, HIP events), the copy pinned -> pageable (one thread; the library uses its pool of copy threads).  floor = resident + upload + download + copy.
usage: python tools/host_batch_floor.py [cfg3] [sizes ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["SRN_HOST_TRACE"] = "1"
import numpy as np
import torch
import serenade_amd as sa
from serenade_amd import synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
sizes = [int(a) for a in sys.argv[2:]] or [4096, 16384, 65536]
inter, n_items, k, m, idfw = synth.CONFIGS[cfg]
off, items, ts = synth.training_sessions(inter, n_items)
ix = sa.VMISIndex.from_sessions(off, items, ts, m, 34, idfw, device=0, builder="gpu")
L, n = synth.LAST_ITEMS, synth.HOW_MANY
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream()


def ev_ms(fn, reps=20):
    out = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream); fn(); b.record(stream); torch.cuda.synchronize(); out.append(a.elapsed_time(b))
    return float(np.median(out[3:]))


for B in sizes:
    qi, qo = synth.queries(int(B / 3.0) + 4096, n_items, seed=synth.SEED + 7919, max_items=L)
    f, o = np.ascontiguousarray(qi[:qo[B]]), np.ascontiguousarray(qo[:B + 1])
    d_flat = torch.from_numpy(f.view(np.int64).copy()).to(dev); d_off = torch.from_numpy(o.astype(np.uint32).view(np.int32).copy()).to(dev)
    d_ids = torch.zeros(B * n, dtype=torch.int64, device=dev); d_sc = torch.zeros(B * n, dtype=torch.float64, device=dev); d_cnt = torch.zeros(B, dtype=torch.int32, device=dev)
    t_res = ev_ms(lambda: sa.predict_batch_device(ix, d_flat.data_ptr(), d_off.data_ptr(), B, L, k, m, n, False, d_ids.data_ptr(), d_sc.data_ptr(), d_cnt.data_ptr(), stream.cuda_stream))
    out = None; th = []
    sys.stderr.flush()
    for c in range(12):
        t0 = time.perf_counter(); out = sa.predict_batch(ix, (f, o), k, m, n, False, out=out); th.append((time.perf_counter() - t0) * 1e3)
    t_host = float(np.median(th[3:]))
    in_bytes, out_bytes = f.nbytes + o.astype(np.uint32).nbytes, B * n * 16 + B * 4
    p_in = torch.empty(in_bytes, dtype=torch.uint8).pin_memory(); dd_in = torch.empty(in_bytes, dtype=torch.uint8, device=dev)
    p_out = torch.empty(out_bytes, dtype=torch.uint8).pin_memory(); dd_out = torch.empty(out_bytes, dtype=torch.uint8, device=dev)
    t_up = ev_ms(lambda: dd_in.copy_(p_in, non_blocking=True))
    t_down = ev_ms(lambda: p_out.copy_(dd_out, non_blocking=True))
    pageable = np.empty(out_bytes, np.uint8); src = p_out.numpy(); tc = []
    for _ in range(12):
        t0 = time.perf_counter(); np.copyto(pageable, src); tc.append((time.perf_counter() - t0) * 1e3)
    t_copy = float(np.median(tc[3:]))
    floor = t_res + t_up + t_down + t_copy
    print("%s, %6d queries: resident %.3f ms | host pointers %.3f ms (%.2fx) | upload %.2f MB %.3f ms, download %.2f MB %.3f ms (%.1f GB/s), pinned -> pageable on one thread %.3f ms"
          " | resident + upload + download = %.3f ms (%.2fx), + the copy = %.3f ms (%.2fx)" % (
              cfg, B, t_res, t_host, t_host / t_res, in_bytes / 1e6, t_up, out_bytes / 1e6, t_down, out_bytes / t_down / 1e6, t_copy, t_res + t_up + t_down, (t_res + t_up + t_down) / t_res,
              floor, floor / t_res), flush=True)

"""Where a session's time goes on the persistent latency path (round 6): host-side per-call microseconds and the resident workgroup's own stamps.
python tools/resident_probe.py [cfg] [calls]   (needs a GPU)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import serenade_amd as sa
from serenade_amd import capi, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
inter, n_items, k, m, idfw = synth.CONFIGS[cfg]
off, items, ts = synth.training_sessions(inter, n_items)
ix = sa.VMISIndex.from_sessions(off, items, ts, m, 34, idfw, builder="gpu")
qi, qo = synth.queries(4000, n_items, seed=synth.SEED + 7919)
L = capi.lib()
ids, sc, n = np.zeros(21, np.uint64), np.zeros(21), C.c_size_t()
def run(tag):
    lat, st = [], []
    s4 = (C.c_uint32 * 4)()
    for q in range(calls):
        ev = np.ascontiguousarray(qi[qo[q]:qo[q + 1]])
        t0 = time.perf_counter()
        L.srn_predict(ix._h, ev.ctypes.data, len(ev), k, m, 21, 0, ids.ctypes.data, sc.ctypes.data, C.byref(n))
        lat.append((time.perf_counter() - t0) * 1e6)
        if tag == "resident" and L.srn_debug_serve_stamps(ix._h, s4) == 0:
            st.append([s4[0], s4[1], s4[2], s4[3]])
    lat = np.array(lat[200:])
    print("%-10s per call us: p50 %.1f p90 %.1f p99 %.1f" % (tag, np.percentile(lat, 50), np.percentile(lat, 90), np.percentile(lat, 99)))
    if st:
        a = np.array(st[200:], float) / 100.0   # 100 MHz ticks -> us
        print("           resident workgroup us (p50 / p90): waited for the doorbell %.1f / %.1f | doorbell -> prep record %.1f / %.1f | doorbell -> answer posted %.1f / %.1f"
              % tuple(np.percentile(a[:, j], p) for j in range(3) for p in (50, 90)))
        cyc = np.array(st[200:], float)[:, 3]
        print("           doorbell -> answer in shader cycles: p50 %.0f p90 %.0f  => shader clock %.2f GHz" % (np.percentile(cyc, 50), np.percentile(cyc, 90), np.median(cyc / (a[:, 2] * 1e3))))
run("launch")
ix.serve_start(k, m, 21, False, lanes=1, max_items_in_session=4, idle_ms=3000)
run("resident")
print("serve stats (answered, launch path, launches, resident):", ix.serve_stats())
ix.serve_stop(); ix.close()

# ---- the resident workgroups beside a throughput batch: what a parked server costs the batch, what the batch costs a session's latency ----
import torch
ix = sa.VMISIndex.from_sessions(off, items, ts, m, 34, idfw, builder="gpu")
B = 1 << 18
bq, bo = synth.queries(int(B / 3.2) + 4096, n_items)
bo = bo[:B + 1]; bq = bq[:bo[-1]]
dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
d_flat = torch.from_numpy(bq.view(np.int64).copy()).to(dev); d_off = torch.from_numpy(bo.view(np.int32).copy()).to(dev)
o_ids = torch.zeros(B * 21, dtype=torch.int64, device=dev); o_sc = torch.zeros(B * 21, dtype=torch.float64, device=dev); o_cnt = torch.zeros(B, dtype=torch.int32, device=dev)
sa.reserve(ix, B, 4, k, m, 21, False, st)
def batch_ms(reps=8):
    for _ in range(2):
        sa.predict_batch_device(ix, d_flat.data_ptr(), d_off.data_ptr(), B, 4, k, m, 21, False, o_ids.data_ptr(), o_sc.data_ptr(), o_cnt.data_ptr(), st)
    torch.cuda.current_stream().synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        sa.predict_batch_device(ix, d_flat.data_ptr(), d_off.data_ptr(), B, 4, k, m, 21, False, o_ids.data_ptr(), o_sc.data_ptr(), o_cnt.data_ptr(), st)
    torch.cuda.current_stream().synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
sa.predict(ix, qi[qo[0]:qo[1]], k, m, 21, False)
t_alone = batch_ms()
ix.serve_start(k, m, 21, False, lanes=4, max_items_in_session=4, idle_ms=3000)
t_parked = batch_ms()
print("2^18-query batch: %.3f ms alone, %.3f ms with 4 resident workgroups parked (idle) on the GPU" % (t_alone, t_parked))
lat = []
for _ in range(40):
    sa.predict_batch_device(ix, d_flat.data_ptr(), d_off.data_ptr(), B, 4, k, m, 21, False, o_ids.data_ptr(), o_sc.data_ptr(), o_cnt.data_ptr(), st)
q = 0
t_end = time.perf_counter() + 0.15      # (40 batches of ~6 ms keep the GPU busy for longer than this)
while time.perf_counter() < t_end:
    ev = np.ascontiguousarray(qi[qo[q]:qo[q + 1]]); q += 1
    t0 = time.perf_counter()
    L.srn_predict(ix._h, ev.ctypes.data, len(ev), k, m, 21, 0, ids.ctypes.data, sc.ctypes.data, C.byref(n))
    lat.append((time.perf_counter() - t0) * 1e6)
torch.cuda.current_stream().synchronize()
lat = np.array(lat)
print("srn_predict through a resident workgroup WHILE 2^18-query batches keep every CU busy: %d calls, p50 %.1f us p90 %.1f us p99 %.1f us; serve stats %s"
      % (len(lat), np.percentile(lat, 50), np.percentile(lat, 90), np.percentile(lat, 99), ix.serve_stats()))
ix.serve_stop(); ix.close()

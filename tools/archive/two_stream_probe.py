"""Do consecutive 2^20-query steps overlap usefully when they alternate between two streams (the tail of step i -- general kernel + finish kernels, low occupancy --
beside the fast kernel of step i + 1)?  usage: python tools/two_stream_probe.py [cfg3]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import serenade_amd as sa
from serenade_amd import synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
B = 1 << 20
inter, n_items, k, m, idfw = synth.CONFIGS[cfg]
off, items, ts = synth.training_sessions(inter, n_items)
ix = sa.VMISIndex.from_sessions(off, items, ts, m, 34, idfw, builder="gpu")
qi, qo = synth.queries(int(2 * B / 3.0) + 4096, n_items, seed=synth.SEED + 7919, max_items=synth.LAST_ITEMS)
dev = torch.device("cuda:0"); n = synth.HOW_MANY
bat = []
for b in range(2):
    fo = qo[b * B:(b + 1) * B + 1].astype(np.int64); f = qi[fo[0]:fo[-1]]; o = (fo - fo[0]).astype(np.uint32)
    bat.append((torch.from_numpy(f.view(np.int64).copy()).to(dev), torch.from_numpy(o.view(np.int32).copy()).to(dev)))
outs = [(torch.zeros(B * n, dtype=torch.int64, device=dev), torch.zeros(B * n, dtype=torch.float64, device=dev), torch.zeros(B, dtype=torch.int32, device=dev)) for _ in range(2)]
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
def run(nstreams, steps):
    for i in range(steps):
        s = streams[i % nstreams]; d_f, d_o = bat[i % 2]; o = outs[i % nstreams]
        sa.predict_batch_device(ix, d_f.data_ptr(), d_o.data_ptr(), B, synth.LAST_ITEMS, k, m, n, False, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), s.cuda_stream)
for ns in (1, 2, 1, 2):
    run(ns, 4); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(ns, 20); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%d stream(s): %.3f ms per step, %.2f M queries/s" % (ns, dt / 20 * 1e3, 20 * B / dt / 1e6))

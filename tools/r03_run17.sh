#!/bin/bash
R=$PWD; O=$R/gpurun_out/r03; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_config5.py -x -q -s 2>&1 | tail -4
timeout 1200 python - <<'PY' 2>&1 | tail -5
# config 5 throughput on one GPU (bench-style loop), with the forked retry pass
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import serenade_amd as sa
from serenade_amd import synth
inter, n_items, k, m, idfw = synth.CONFIGS["cfg5"]
off, items, ts = synth.training_sessions(inter, n_items)
full = sa.VMISIndex.from_sessions(off, items, ts, m, 34, idfw, device=0, builder="gpu")
B = 1 << 18
qi, qo = synth.queries(int(B / 3.0) + 4096, n_items, seed=synth.SEED + 7919, max_items=4)
qo = qo[:B + 1]; qi = qi[:qo[-1]]
dev = torch.device("cuda:0"); n = 21
d_flat = torch.from_numpy(qi.view(np.int64).copy()).to(dev); d_off = torch.from_numpy(qo.view(np.int32).copy()).to(dev)
o = (torch.zeros(B * n, dtype=torch.int64, device=dev), torch.zeros(B * n, dtype=torch.float64, device=dev), torch.zeros(B, dtype=torch.int32, device=dev))
st = torch.cuda.current_stream().cuda_stream
def run(): sa.predict_batch_device(full, d_flat.data_ptr(), d_off.data_ptr(), B, 4, k, m, n, False, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), st)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
tp, tf, tpr, tr = full.kernel_times_detail(8)
print("config 5, %d queries per call, resident: %.3f ms per call = %.2f M queries/s; fast kernel %.3f ms, all predict launches %.3f, prep %.3f, retry window %.3f; path counts %s" % (B, ms, B / ms / 1e3, tf.mean(), tpr.mean(), tp.mean(), tr.mean(), full.last_path_counts()))
PY

"""Fast-kernel timing of the library in use (SRN_LIB_PATH selects a variant): config 3, 2^20 resident queries, HIP-event duration of vmis_fast_kernel over 8 launches,
plus a checksum of the results (variants must agree bit for bit).  usage: python tools/fast_time.py [cfg3] [nq]"""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import serenade_amd as sa
from serenade_amd import synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
inter, n_items, k, m, idfw = synth.CONFIGS[cfg]
off, items, ts = synth.training_sessions(inter, n_items)
ix = sa.VMISIndex.from_sessions(off, items, ts, m, 34, idfw, builder="gpu")
qi, qo = synth.queries(int(B / 3.0) + 4096, n_items, seed=synth.SEED + 7919, max_items=synth.LAST_ITEMS)
qo = qo[:B + 1]; qi = qi[:qo[-1]]
dev = torch.device("cuda:0"); n = synth.HOW_MANY
d_flat = torch.from_numpy(qi.view(np.int64).copy()).to(dev); d_off = torch.from_numpy(qo.view(np.int32).copy()).to(dev)
o_ids = torch.zeros(B * n, dtype=torch.int64, device=dev); o_sc = torch.zeros(B * n, dtype=torch.float64, device=dev); o_cnt = torch.zeros(B, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
sa.reserve(ix, B, synth.LAST_ITEMS, k, m, n, False, st)
ix.kernel_timing(True)
for _ in range(11):
    sa.predict_batch_device(ix, d_flat.data_ptr(), d_off.data_ptr(), B, synth.LAST_ITEMS, k, m, n, False, o_ids.data_ptr(), o_sc.data_ptr(), o_cnt.data_ptr(), st)
torch.cuda.synchronize()
t_prep, t_fast, t_pred, t_retry = ix.kernel_times_detail(8)
nq_, general, glob = ix.last_path_counts()
h = hashlib.sha256(); h.update(o_ids.cpu().numpy().tobytes()); h.update(o_sc.cpu().numpy().tobytes()); h.update(o_cnt.cpu().numpy().tobytes())
print("%-40s fast kernel %.3f ms avg (min %.3f), all predict launches %.3f ms, prep %.3f ms; handed over %d; results %s" %
      (os.path.basename(os.environ.get("SRN_LIB_PATH", "default")), t_fast.mean(), t_fast.min(), t_pred.mean(), t_prep.mean(), general, h.hexdigest()[:16]))

"""Determinism soak: the same batch through srn_predict_batch_device N times must give the same bytes every time (a race in a rare path shows up
as a differing row).  usage: python tools/determinism_soak.py [cfg3] [batch] [repeats]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import serenade_amd as sa
from serenade_amd import synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
R = int(sys.argv[3]) if len(sys.argv) > 3 else 30
inter, n_items, k, m, idfw = synth.CONFIGS[cfg]
off, items, ts = synth.training_sessions(inter, n_items)
ix = sa.VMISIndex.from_sessions(off, items, ts, m, 34, idfw, device=0, builder="gpu")
qi, qo = synth.queries(int(B / 3.0) + 4096, n_items, seed=synth.SEED + 4242, max_items=synth.LAST_ITEMS)
qo = qo[:B + 1]; qi = qi[:qo[-1]]
dev = torch.device("cuda:0")
d_flat = torch.from_numpy(qi.view(np.int64).copy()).to(dev); d_off = torch.from_numpy(qo.view(np.int32).copy()).to(dev)
n = synth.HOW_MANY
st = torch.cuda.current_stream().cuda_stream
ref = None; bad = 0
for r in range(R):
    ids = torch.zeros(B * n, dtype=torch.int64, device=dev); sc = torch.zeros(B * n, dtype=torch.float64, device=dev); cnt = torch.zeros(B, dtype=torch.int32, device=dev)
    sa.predict_batch_device(ix, d_flat.data_ptr(), d_off.data_ptr(), B, synth.LAST_ITEMS, k, m, n, False, ids.data_ptr(), sc.data_ptr(), cnt.data_ptr(), st)
    torch.cuda.synchronize()
    if ref is None:
        ref = (ids, sc, cnt); print("path counts", ix.last_path_counts())
    else:
        d = int(((ids.view(B, n) != ref[0].view(B, n)).any(1) | (sc.view(B, n) != ref[1].view(B, n)).any(1) | (cnt != ref[2])).sum().item())
        if d: bad += 1; print("repeat", r, ":", d, "rows differ")
print("%s batch %d: %d repeats, %d differed" % (cfg, B, R, bad))
sys.exit(1 if bad else 0)

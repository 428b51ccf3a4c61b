"""Scratch experiment driver (GPU box): phase-cycle breakdown at a given config. Not part of the product."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import serenade_amd as sa
from serenade_amd import synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
inter, n_items, k, m, idfw = synth.CONFIGS[cfg]
off, items, ts = synth.training_sessions(inter, n_items)
ix = sa.VMISIndex.from_sessions(off, items, ts, m, 34, idfw, device=0)
ix.kernel_timing(True)
qi, qo = synth.queries(int(B / 3) + 2048, n_items, seed=synth.SEED + 7919)
qi, qo = qi[:qo[B]], qo[:B + 1]
print("cpus", os.cpu_count(), len(os.sched_getaffinity(0)), open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "no cgroup cpu.max")
import serenade_amd.capi as capi
FL = int(os.environ.get("SRN_FL", "0"), 0)
capi.FLAG_BUSINESS_LOGIC = FL
for rep in range(2):
    sa.predict_batch(ix, (qi, qo), k, m, 21, bool(FL))
ix.debug_phase_cycles(True)
t0 = time.time(); sa.predict_batch(ix, (qi, qo), k, m, 21, bool(FL)); dt = time.time() - t0
cyc = ix.debug_phase_cycles(False).astype(np.float64)
ms, msr, _ = ix.last_kernel_ms()
print("path counts (nq, general kernel, global pass):", ix.last_path_counts())
r = sa.predict_batch_debug(ix, (qi, qo), k, m, 21, bool(FL), neighbours=False)
names = ["0 prep+clear", "1 stage lists", "2 m-cut (fast kernel; general kernel: merges+m-cut)", "3 merge tree (fast kernel; general kernel: class count)", "4 k-cut scans+publish", "5", "6", "7", "8 clear hot+sketch", "9 walk A", "10 harvest hot", "11 live check+clear", "12 walk B (+surv)", "13 harvest exact/final", "14", "15"]
print("main %.2f ms retry %.2f ms  total cycles %.3g" % (ms, msr, cyc.sum()))
for n, c in zip(names, cyc):
    print("  %-18s %6.2f%%  %.0f cyc/query" % (n, 100 * c / cyc.sum(), c / B))
c15 = int(cyc[15])
print("queries whose walk B listed nothing: %d" % cyc[7])
print("handed over after a list overflowed: candidate list %d, floor survivors %d, hit list %d, exact table %d" % (c15 & 0xFFFF, (c15 >> 16) & 0xFFFF, (c15 >> 32) & 0xFFFF, (c15 >> 48) & 0xFFFF))
if cyc[14]:
    print("fast kernel: queries %d, walk-B hit elements/query %.1f, candidates/query %.1f, floor survivors/query %.1f" % (cyc[14], cyc[5] / cyc[14], cyc[6] / cyc[14], cyc[7] / cyc[14]))
st = r["stats"].astype(np.float64)
print("mean P,C,K,I,D,H,L", st[:, :7].mean(0).round(1), "retry frac", (st[:, 7] == 1).mean())
print("D pct", np.percentile(st[:, 4], [50, 90, 99, 100]), "I pct", np.percentile(st[:, 3], [50, 90, 99, 100]), "P pct", np.percentile(st[:,0],[50,90,99,100]))

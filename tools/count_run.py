"""Experiment driver (GPU box): three full predict launches on a config, nothing else -- for instruction counting under rocprofv3 --pmc."""
import os, sys
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
for rep in range(3):
    sa.predict_batch(ix, (qi, qo), k, m, 21, False)
print("main ms", ix.last_kernel_ms()[0])

"""Scratch measurement (GPU box): time to build ONE shard of an item-sharded index on one GPU (srn_index_build_shard_gpu = rocPRIM build of the
unsharded index + one host pass that cuts the shard + upload), and the G = 1 overhead of the three-stage pipeline against the fused path.
python tools/shard_build_time.py [config] [n_shards]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import serenade_amd as sa
from serenade_amd import sharded, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg5_8th"
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
inter, n_items, k, m, idfw = synth.CONFIGS[cfg]
t0 = time.time(); off, items, ts = synth.training_sessions(inter, n_items); print("generate %.1f s: %d sessions, %d interactions" % (time.time() - t0, len(ts), len(items)))
for g in (0, G - 1):
    t0 = time.time(); ix = sharded.ShardedVMISIndex(off, items, ts, m, 34, idfw, g, G, device=0, builder="gpu"); dt = time.time() - t0
    info = ix.info
    print("shard %d of %d built on the GPU in %.2f s: %d items, %d posting entries, %d row items, %.2f GB in HBM" % (g, G, dt, info["n_items"], info["nnz_postings"], info["nnz_rows"], info["device_bytes"] / 1e9))
    del ix
t0 = time.time(); full = sa.VMISIndex.from_sessions(off, items, ts, m, 34, idfw, device=-1, builder="host") if inter <= 60_000_010 else None
if full is not None:
    print("(host builder, unsharded, for comparison: %.1f s)" % (time.time() - t0))

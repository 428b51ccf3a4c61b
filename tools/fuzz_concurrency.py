"""Concurrency soak (GPU box): several host threads hammer ONE index with a random mix of entry points -- srn_predict (combining rounds), host batches of every size class
(latency path, one chunk, one chunk in pieces, several chunks), device-resident batches on a stream per thread, a shared shard group -- and every answer is compared with
the oracle's.  usage: python tools/fuzz_concurrency.py [seconds] [threads] [seed]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import serenade_amd as sa
from serenade_amd import sharded, synth
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
inter, n_items, k, m, idfw = synth.CONFIGS["tiny"]
off, items, ts = synth.training_sessions(inter, n_items)
gix = sa.VMISIndex.from_sessions(off, items, ts, m, 34, idfw)
oix = O.OracleIndex(off, items, ts, m, 34, idfw, fast=True)
NQ = 150000
MAX_ITEMS = int(os.environ.get("SOAK_MAX_ITEMS", "10"))
flat, qoff = synth.queries(NQ, n_items, max_items=MAX_ITEMS)   # (sessions of up to 10 items since round 4: the fast kernel's MID instantiation and the latency path's fast sequence are in the mix; 4 = the headline workload)
NQ = len(qoff) - 1
n = 21
# several parameter sets in flight at once: the workspaces are re-sized and the kernel path (fast kernel or not, position sets or not) changes from call to call
PARAMS = [(k, m, 21), (50, 100, 5), (700, 300, 24), (k, m, 100)]
REFS = [oix.predict_batch("canonical", flat, qoff, kk_, mm_, nn_, False, threads=16) for (kk_, mm_, nn_) in PARAMS]
ref = REFS[0]
dev = torch.device("cuda:0")
d_flat_all = torch.from_numpy(flat.view(np.int64).copy()).to(dev)
shards = [sharded.ShardedVMISIndex.from_full(gix, g, 3) for g in range(3)]
grp = sharded.ShardGroup.local(shards); grp_lock = threading.Lock()
OPS = os.environ.get("SOAK_OPS", "predict,predict,host,host,device,group").split(",")   # (SOAK_OPS=predict,host: the library's own paths only, no torch tensors on the data path)
fail = []; counts = {"predict": 0, "host": 0, "device": 0, "group": 0}; cl = threading.Lock()

def cmp(lo, hi, ids, sc, cnt, what, pi=0):
    ref = REFS[pi]; n = PARAMS[pi][2]
    r_cnt = ref["counts"][lo:hi]
    ok = np.array_equal(cnt, r_cnt)
    if ok:
        mask = np.arange(n)[None, :] < r_cnt[:, None].astype(np.int64)
        ok = np.array_equal(ids[mask], ref["ids"][lo:hi][mask]) and np.allclose(sc[mask], ref["scores"][lo:hi][mask], rtol=1e-12, atol=0)
    if not ok:
        fail.append("%s: queries [%d, %d)" % (what, lo, hi))
    return ok

def worker(tid):
    rng = np.random.default_rng(seed * 1000 + tid)
    st = torch.cuda.Stream(dev)
    t_end = time.time() + budget
    try:
        while time.time() < t_end and not fail:
            op = rng.choice(OPS)
            pi = int(rng.integers(0, len(PARAMS))); k, m, n = PARAMS[pi]; ref = REFS[pi]
            if op == "predict":
                q = int(rng.integers(0, NQ))
                recs = sa.predict(gix, flat[qoff[q]:qoff[q + 1]], k, m, n, False)
                c = int(ref["counts"][q])
                if [r.id for r in recs] != ref["ids"][q, :c].tolist() or not np.allclose([r.score for r in recs], ref["scores"][q, :c], rtol=1e-12, atol=0):
                    fail.append("srn_predict query %d" % q)
            else:
                size = int(rng.choice([2, 40, 256, 257, 3000, 5000, 20000, 70000, 140000]))
                lo = int(rng.integers(0, NQ - size)); hi = lo + size
                f, o = flat[qoff[lo]:qoff[hi]], (qoff[lo:hi + 1] - qoff[lo]).astype(np.uint32)
                if op == "host":
                    ids, sc, cnt = sa.predict_batch(gix, (f, o), k, m, n, False)
                    cmp(lo, hi, ids, sc, cnt, "host batch of %d, params %r" % (size, PARAMS[pi]), pi)
                else:
                    with torch.cuda.stream(st):
                        d_f = d_flat_all[int(qoff[lo]):int(qoff[hi])]; d_o = torch.from_numpy(o.view(np.int32).copy()).to(dev, non_blocking=False)
                        if op == "device":
                            r_ids = torch.zeros(size * n, dtype=torch.int64, device=dev); r_sc = torch.zeros(size * n, dtype=torch.float64, device=dev); r_cnt = torch.zeros(size, dtype=torch.int32, device=dev)
                            sa.predict_batch_device(gix, d_f.data_ptr(), d_o.data_ptr(), size, MAX_ITEMS, k, m, n, False, r_ids.data_ptr(), r_sc.data_ptr(), r_cnt.data_ptr(), st.cuda_stream)
                            st.synchronize()
                            cmp(lo, hi, r_ids.cpu().numpy().view(np.uint64).reshape(size, n), r_sc.cpu().numpy().reshape(size, n), r_cnt.cpu().numpy().view(np.uint32), "device batch of %d, params %r" % (size, PARAMS[pi]), pi)
                        else:
                            size = min(size, 20000); hi = lo + size
                            d_f = d_flat_all[int(qoff[lo]):int(qoff[hi])]; d_o = torch.from_numpy((qoff[lo:hi + 1] - qoff[lo]).astype(np.int32)).to(dev)
                            res = grp.predict_batch(d_f, d_o, size, MAX_ITEMS, k, m, n, False, stream=st.cuda_stream)   # (the group serialises its callers; consecutive batches may come on different streams)
                            st.synchronize()
                            cmp(lo, hi, res[0].cpu().numpy().view(np.uint64), res[1].cpu().numpy(), res[2].cpu().numpy().view(np.uint32), "shard group batch of %d, params %r" % (size, PARAMS[pi]), pi)
            with cl: counts[op] += 1
    except Exception as e:
        import traceback
        fail.append("thread %d: %r\n%s" % (tid, e, traceback.format_exc()))

ths = [threading.Thread(target=worker, args=(i,)) for i in range(T)]
for t in ths: t.start()
for t in ths: t.join()
if fail:
    print("CONCURRENCY SOAK FAILED:", fail[:5]); sys.exit(1)
print("concurrency soak ok: %d threads, %.0f s, ops %s" % (T, budget, counts))

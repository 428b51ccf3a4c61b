"""Multi-GPU plumbing (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

predict_next queries are independent of each other, so the path shards by QUERY: every rank holds the whole
index in its own HBM (even BASELINE config 5 is ~25 GB against 288 GB) and serves a contiguous slice of each
global batch.  There is no collective on the data path; the only communication is control-plane (barriers,
the max-over-ranks timing reduction, and an optional gather of the per-rank result slices to one rank).
The reference scales the same way -- replicas behind a load balancer with session affinity
(src/endpoints/recommend_resource.rs:17-19) -- here the replicas are the GPUs of one node.
"""
import os

import numpy as np


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend, device=None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (no-op for world 1)."""
    import torch.distributed as dist
    rank, _local, world = env_rank_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {"device_id": device} if (device is not None and backend == "nccl") else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def shard_range(n, rank, world):
    """Contiguous slice [lo, hi) of n queries owned by `rank`; the first n % world ranks get one extra."""
    base, extra = divmod(int(n), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_queries(items_flat, q_off, rank, world):
    """This rank's slice of a CSR query batch, re-based to start at offset 0."""
    nq = len(q_off) - 1
    lo, hi = shard_range(nq, rank, world)
    fo = np.asarray(q_off[lo:hi + 1], dtype=np.int64)
    return np.ascontiguousarray(items_flat[fo[0]:fo[-1]]), (fo - fo[0]).astype(np.uint32), lo, hi


def max_over_ranks(value, device="cpu"):
    """Max of a python float over all ranks (the bench's step time is the slowest rank's)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def gather_results(ids, scores, counts, nq_total, device="cpu"):
    """All-gather the per-rank result slices (numpy [nq_r, n] / [nq_r]) into full-batch arrays in query order.
    Slices are contiguous and ordered by rank (shard_range), so this is a concatenation."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return ids, scores, counts
    world, n = dist.get_world_size(), ids.shape[1]
    sizes = [shard_range(nq_total, r, world) for r in range(world)]
    cap = max(hi - lo for lo, hi in sizes)

    def ag(a, dtype, width):
        pad = np.zeros((cap, width), a.dtype)
        pad[:a.shape[0]] = a.reshape(a.shape[0], width)
        t = torch.from_numpy(pad.view(dtype)).to(device)
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        return np.concatenate([o.cpu().numpy().view(a.dtype)[:hi - lo] for o, (lo, hi) in zip(out, sizes)])

    return ag(ids, np.int64, n), ag(scores, np.float64, n), ag(counts.reshape(-1, 1), np.int32, 1).reshape(-1)

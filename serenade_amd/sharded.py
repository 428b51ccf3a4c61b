"""Item-sharded VMIS-kNN index across the GPUs of one node (BASELINE.json north star / config 5): the binding of srn_index_shard / srn_shard_group_*.

Shard g (one process, one GPU) holds the postings, row fragments and idf of the items with owner(id) == g; session recency ranks are global.  A batch of evolving
sessions, replicated on every rank, goes through ONE C call per rank (ShardGroup.predict_batch -> srn_shard_group_predict_batch); the pipelines -- neighbours (posting
lists replicated, candidate work divided over the ranks), lists, three-stage -- and their collectives (RCCL over xGMI, or application callbacks) live inside the
library (csrc/srn_group.hip).  An item's whole score lives on its owner and every integer that enters it is global, so the result is
bit-identical to the unsharded path.  Rounds 1-2 drove the stages from Python (torch.distributed collectives, tensor-library scans and merges); that code is gone.
"""
import ctypes as C

import numpy as np

from . import capi



class ShardedVMISIndex:
    """One shard.  Every rank passes the SAME training sessions (or the same unsharded index) and its own shard number.
    builder="gpu" (default when the shard is attached to a device): the unsharded index is built with rocPRIM sorts on this
    rank's GPU and the shard is cut out of it in one pass -- seconds at BASELINE config 4/5 scale; builder="host": the
    single-threaded host builder restricted to the shard's items (same bytes)."""

    def __init__(self, sess_off, items, max_ts, m_index, max_session_len, idf_weighting, shard, n_shards, device=0, builder=None):
        sess_off, items, max_ts = capi.as_u64(sess_off), capi.as_u64(items), capi.as_u32(max_ts)
        v = capi.SessionsView(sess_off.ctypes.data, items.ctypes.data, max_ts.ctypes.data, len(max_ts))
        h = C.c_void_p()
        if builder is None:
            builder = "gpu" if device >= 0 and len(max_ts) < 0xFFFFFFFF and len(items) < 0xFFFFFFFF else "host"
        fn = capi.lib().srn_index_build_shard_gpu if builder == "gpu" else capi.lib().srn_index_build_shard
        capi.check(fn(C.byref(v), int(m_index), int(max_session_len), float(idf_weighting), int(shard), int(n_shards), int(device), C.byref(h)))
        self._h, self.shard, self.n_shards, self.device = h, int(shard), int(n_shards), int(device)

    @classmethod
    def _wrap(cls, h, shard, n_shards, device):
        self = cls.__new__(cls)
        self._h, self.shard, self.n_shards, self.device = h, int(shard), int(n_shards), int(device)
        return self

    @classmethod
    def from_full(cls, full_index, shard, n_shards, device=0):
        """Cut shard `shard` out of an unsharded VMISIndex (srn_index_shard)."""
        h = C.c_void_p()
        capi.check(capi.lib().srn_index_shard(full_index._h, int(shard), int(n_shards), int(device), C.byref(h)))
        return cls._wrap(h, shard, n_shards, device)

    @classmethod
    def load(cls, path, shard, n_shards, device=0):
        """From a saved index (VMISIndex.save): an unsharded file is cut, a file that already holds this shard is taken as is."""
        h = C.c_void_p()
        capi.check(capi.lib().srn_index_load_shard(str(path).encode(), int(shard), int(n_shards), int(device), C.byref(h)))
        return cls._wrap(h, shard, n_shards, device)

    def save(self, path):
        capi.check(capi.lib().srn_index_save(self._h, str(path).encode()))

    def close(self):
        if getattr(self, "_h", None) and capi is not None and getattr(capi, "lib", None) is not None:   # (None at interpreter shutdown)
            capi.lib().srn_index_free(self._h)
            self._h = None

    __del__ = close

    @property
    def info(self):
        out = capi.IndexInfo()
        capi.check(capi.lib().srn_index_info(self._h, C.byref(out)))
        return {n: getattr(out, n) for n, _ in capi.IndexInfo._fields_}


def postings_view(full_index, device=0):
    """srn_index_postings_view: the replicated part of an item-sharded index (dictionary, idf / attributes, posting lists of the WHOLE index, no rows)."""
    from .vmisknn import VMISIndex
    h = C.c_void_p()
    capi.check(capi.lib().srn_index_postings_view(full_index._h, int(device), C.byref(h)))
    return VMISIndex(h)


class DistComm:
    """Collectives of the sharded pipeline over torch.distributed.  With backend "nccl" (RCCL) tensors stay on the GPU;
    with "gloo" (CPU tests) they are staged through host memory."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.world = dist.get_world_size(group)
        self.staged = dist.get_backend(group) != "nccl"

    def all_gather(self, t):
        import torch
        src = (t.cpu() if self.staged else t).contiguous()
        out = torch.empty(self.world * src.numel(), dtype=src.dtype, device=src.device)   # flat: accepted by every backend
        self.dist.all_gather_into_tensor(out, src.view(-1), group=self.group)
        out = out.view((self.world,) + tuple(src.shape))
        return out.to(t.device) if self.staged else out

    def all_reduce_min(self, t):
        if self.staged:
            h = t.cpu()
            self.dist.all_reduce(h, op=self.dist.ReduceOp.MIN, group=self.group)
            t.copy_(h)
        else:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, group=self.group)
        return t


    def all_reduce_max(self, t):
        if self.staged:
            h = t.cpu()
            self.dist.all_reduce(h, op=self.dist.ReduceOp.MAX, group=self.group)
            t.copy_(h)
        else:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return t


def _ptr(t):
    return C.c_void_p(t.data_ptr())


# ---- the shard group: the whole sharded batch in ONE C call, collectives inside the library (srn_group.hip) ----------------------------------
class ShardGroup:
    """srn_shard_group_*: this rank's end of an item-sharded index.  Three transports, one pipeline:
    ShardGroup.rccl(shard, rank, world)      RCCL called from inside the library (one process per GPU); the 256-byte id travels over torch.distributed
    ShardGroup.over(shard, rank, world, comm)  the collectives as callbacks into a DistComm (gloo in the tests: two processes sharing one GPU)
    ShardGroup.local(shards)                 all shards in this process on one device: collectives degenerate to kernels"""

    def __init__(self, h, keep=()):
        self._h, self._keep = h, keep

    @classmethod
    def rccl(cls, shard, rank, world, group=None):
        import torch
        import torch.distributed as dist
        buf = (C.c_uint8 * capi.SHARD_GROUP_ID_BYTES)()
        if rank == 0:
            capi.check(capi.lib().srn_shard_group_unique_id(buf, capi.SHARD_GROUP_ID_BYTES))
        if world > 1:   # the id reaches the other ranks over whatever control plane the host has: here the process group
            t = torch.tensor(list(bytes(buf)), dtype=torch.uint8)
            if dist.get_backend(group) == "nccl":
                t = t.cuda()
            dist.broadcast(t, src=0, group=group)
            buf = (C.c_uint8 * capi.SHARD_GROUP_ID_BYTES)(*t.cpu().tolist())
        h = C.c_void_p()
        capi.check(capi.lib().srn_shard_group_create(shard._h, buf, int(rank), int(world), C.byref(h)))
        return cls(h, (shard,))

    @classmethod
    def local(cls, shards):
        arr = (C.c_void_p * len(shards))(*[s._h for s in shards])
        h = C.c_void_p()
        capi.check(capi.lib().srn_shard_group_create_local(arr, len(shards), C.byref(h)))
        return cls(h, tuple(shards))

    @classmethod
    def over(cls, shard, rank, world, comm):
        """Collectives through `comm` (a DistComm): device buffers are wrapped as torch tensors by address; with a host-staged backend (gloo) each
        callback synchronises the stream, moves the bytes through host memory and back."""
        import torch

        dev = torch.device("cuda", shard.device)

        def as_tensor(ptr, nbytes):
            # a torch view of raw device memory: through the __cuda_array_interface__ protocol
            class _Raw:
                pass
            raw = _Raw()
            raw.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}
            return torch.as_tensor(raw, device=dev)

        def sync(stream):
            torch.cuda.synchronize(dev)   # (test transport: everything enqueued so far is done before the bytes leave)

        def arm(user, channel, d_buf, count, stream):
            try:
                sync(stream)
                t = as_tensor(d_buf, count * 4).view(torch.int32)
                comm.all_reduce_max(t)
                torch.cuda.synchronize(dev)
                return 0
            except Exception:   # pragma: no cover
                import traceback; traceback.print_exc()
                return capi.SRN_EHIP

        def ag(user, channel, d_buf, block_bytes, stream):
            try:
                sync(stream)
                t = as_tensor(d_buf, block_bytes * world).view(world, block_bytes)
                g = comm.all_gather(t[rank].clone())
                t.copy_(g.view(world, block_bytes))
                torch.cuda.synchronize(dev)
                return 0
            except Exception:   # pragma: no cover
                import traceback; traceback.print_exc()
                return capi.SRN_EHIP

        def agv(user, channel, d_buf, byte_off, byte_cnt, stream):
            try:
                sync(stream)
                offs = [int(byte_off[i]) for i in range(world)]; cnts = [int(byte_cnt[i]) for i in range(world)]
                width = max(1, max(cnts))
                mine = torch.zeros(width, dtype=torch.uint8, device=dev)
                if cnts[rank]:
                    mine[:cnts[rank]] = as_tensor(d_buf + offs[rank], cnts[rank])
                g = comm.all_gather(mine).view(world, width)
                for r in range(world):
                    if r != rank and cnts[r]:
                        as_tensor(d_buf + offs[r], cnts[r]).copy_(g[r, :cnts[r]])
                torch.cuda.synchronize(dev)
                return 0
            except Exception:   # pragma: no cover
                import traceback; traceback.print_exc()
                return capi.SRN_EHIP

        def armin(user, channel, d_buf, count, stream):
            try:
                sync(stream)
                t = as_tensor(d_buf, count * 4).view(torch.int32)
                comm.all_reduce_min(t)
                torch.cuda.synchronize(dev)
                return 0
            except Exception:   # pragma: no cover
                import traceback; traceback.print_exc()
                return capi.SRN_EHIP

        cbs = (capi.ALL_REDUCE_MAX_FN(arm), capi.ALL_GATHER_FN(ag), capi.ALL_GATHER_V_FN(agv), capi.ALL_REDUCE_MAX_FN(armin))
        sc = capi.ShardComm(None, *cbs)
        h = C.c_void_p()
        capi.check(capi.lib().srn_shard_group_create_with_comm(shard._h, int(rank), int(world), C.byref(sc), C.byref(h)))
        return cls(h, (shard, cbs, sc, comm))

    def predict_batch(self, d_items_flat, d_q_off, nq, max_len, k, m, how_many, enable_business_logic=False, stream=None, resident=False, out=None):
        """d_items_flat (int64 view of the u64 ids) / d_q_off (int32): torch tensors on the group's GPU holding the SAME batch on every rank.
        -> (ids int64 [nq, n], scores f64 [nq, n], counts int32 [nq]) torch tensors, identical on every rank; asynchronous on `stream`."""
        import torch
        dev = d_items_flat.device
        if stream is None:
            stream = torch.cuda.current_stream(dev).cuda_stream
        if out is None:
            out = (torch.empty((nq, how_many), dtype=torch.int64, device=dev), torch.empty((nq, how_many), dtype=torch.float64, device=dev),
                   torch.empty(nq, dtype=torch.int32, device=dev))
        flags = (capi.FLAG_BUSINESS_LOGIC if enable_business_logic else 0) | (capi.FLAG_INPUTS_RESIDENT if resident else 0)
        capi.check(capi.lib().srn_shard_group_predict_batch(self._h, _ptr(d_items_flat), _ptr(d_q_off), int(nq), int(max_len), int(k), int(m), int(how_many), flags,
                                                            _ptr(out[0]), _ptr(out[1]), _ptr(out[2]), C.c_void_p(stream)))
        return out

    def set_postings(self, postings):
        """srn_shard_group_set_postings: the NEIGHBOURS pipeline -- `postings` = the unsharded VMISIndex (or its rows-free postings_view) on this rank's device; None: off."""
        capi.check(capi.lib().srn_shard_group_set_postings(self._h, postings._h if postings is not None else None))
        self._postings = postings          # (must outlive the group)

    def set_overlap(self, on):
        """srn_shard_group_set_overlap: batch i + 1's exchange beside batch i's kernels and result gather (two communicators in flight), or everything in
        issue order on the caller's stream (the default since round 5)."""
        capi.check(capi.lib().srn_shard_group_set_overlap(self._h, 1 if on else 0))

    def wait(self, timeout_ms):
        """srn_shard_group_wait: a bounded wait for the group's most recent batch (SerenadeError with code SRN_ETIMEOUT -- and a broken group -- if a peer never shows up)."""
        capi.check(capi.lib().srn_shard_group_wait(self._h, int(timeout_ms)))

    @property
    def stats(self):
        st = capi.ShardGroupStats()
        capi.check(capi.lib().srn_shard_group_stats(self._h, C.byref(st)))
        return {n: getattr(st, n) for n, _ in capi.ShardGroupStats._fields_}

    def close(self):
        if getattr(self, "_h", None) and capi is not None and getattr(capi, "lib", None) is not None:
            capi.lib().srn_shard_group_free(self._h)
            self._h = None

    __del__ = close

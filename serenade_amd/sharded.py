"""Item-sharded VMIS-kNN index across the GPUs of one node (BASELINE.json north star / config 5).

Shard g (one process, one GPU) holds the postings, row fragments and idf of the items with owner(id) == g; session
recency ranks are global.  A batch of evolving sessions, replicated on every rank, takes three kernel stages around
three collectives (RCCL over xGMI when the backend is "nccl"):

    stage A  (kernel)   this shard's candidate sessions (rank, partial similarity numerator), locally cut to m
    all-gather          candidate lists                      <= m * 4..8 B per query and shard
    stage B  (kernel)   merge, global m-cut and k-cut -> the neighbour list (identical on every rank);
                        partial first-match positions over the evolving items this shard owns
    all-reduce(min)     first-match positions                (k + 1) * 4 B per query
    stage C  (kernel)   accumulate over this shard's row fragments, exact top-n among the items it owns
    all-gather          per-shard top-n                      n * 16 B per query and shard
    merge               top-n of the G * n candidates by (score desc, item id asc) -- a few torch ops on device

An item's whole score lives on its owner, and every integer that enters it (neighbour set, numerators, first-match
positions) is global, so the result is bit-identical to the unsharded path.  The exchange volumes make this the mode
for indices that do not fit one GPU, not the fast path: while the index fits, query-sharded replicas
(serenade_amd.distributed) need no collective at all.
"""
import ctypes as C

import numpy as np

from . import capi

MINPOS_NONE = 0x7FFFFFFF


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

    def slot_bytes(self, max_len):
        out = C.c_uint32()
        capi.check(capi.lib().srn_shard_slot_bytes(self._h, int(max_len), C.byref(out)))
        return out.value

    def slot_info(self, max_len):
        """(bytes per packed entry, low payload bits): rank = entry >> bits."""
        a, b = C.c_uint32(), C.c_uint32()
        capi.check(capi.lib().srn_shard_slot_info(self._h, int(max_len), C.byref(a), C.byref(b)))
        return a.value, b.value


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


class SoloComm:
    """world size 1 (single shard): the collectives degenerate."""
    world = 1

    def all_gather(self, t):
        return t.unsqueeze(0)

    def all_reduce_min(self, t):
        return t

    def all_reduce_max(self, t):
        return t


def _local_mth_rank(cand, cand_cnt, nq, m, slot_bytes, num_bits):
    """Per query: the rank of this shard's m-th most recent candidate, 0 if it holds fewer than m."""
    import torch
    c2 = cand.view(nq, m)
    u = (c2.to(torch.int64) & 0xFFFFFFFF) if slot_bytes == 4 else c2            # packed value, unsigned
    rank = u >> num_bits
    cnt = cand_cnt.to(torch.int64)
    valid = torch.arange(m, device=cand.device).view(1, m) < cnt.clamp(min=0).view(nq, 1)
    mth = torch.where(cnt == m, rank.masked_fill(~valid, torch.iinfo(torch.int64).max).min(dim=1).values, torch.zeros_like(cnt))
    return mth, rank, valid


def _keep_at_or_above(cand, cand_cnt, rank, valid, tau, nq, m):
    """Entries with rank >= tau moved to the front of each query's list (in their order) -> (packed [nq, m], counts [nq])."""
    import torch
    keep = valid & (rank >= tau.view(nq, 1))
    order = torch.argsort((~keep).to(torch.int8), dim=1, stable=True)
    packed = torch.gather(cand.view(nq, m), 1, order)
    new_cnt = keep.sum(dim=1)
    return packed, torch.where(cand_cnt < 0, cand_cnt, new_cnt.to(cand_cnt.dtype))   # (the overflow marker travels)


def compact_candidates(cand, cand_cnt, nq, m, slot_bytes, num_bits, reduce_max):
    """all-gather #1 is the bandwidth-relevant exchange (m entries per query and shard).  A tiny all-reduce(max)
    of the shards' local m-th ranks gives, per query, a rank below which no entry can be among the m most recent distinct
    sessions of the union (the shard that attains the maximum alone holds m sessions at or above it); only the entries at or
    above it are shipped, in slabs as wide as the fullest list (a second, scalar all-reduce(max)).
    cand [nq * m] packed (rank << num_bits | payload), cand_cnt [nq] (-1 = overflow on this shard) -> (slab [nq, w], cnt [nq], w)."""
    mth, rank, valid = _local_mth_rank(cand, cand_cnt, nq, m, slot_bytes, num_bits)
    tau = reduce_max(mth)
    packed, cnt = _keep_at_or_above(cand, cand_cnt, rank, valid, tau, nq, m)
    w = int(reduce_max(cnt.clamp(min=0).max().to(mth.dtype).view(1)).item())
    w = min(m, (max(1, w) + 63) // 64 * 64)
    return packed[:, :w].contiguous(), cnt, w


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _stage_a(ix, d_flat, d_off, nq, max_len, k, m, dtype, stream):
    import torch
    cand = torch.empty(nq * m, dtype=dtype, device=d_flat.device)
    cnt = torch.empty(nq, dtype=torch.int32, device=d_flat.device)
    capi.check(capi.lib().srn_shard_stage_a(ix._h, _ptr(d_flat), _ptr(d_off), nq, max_len, k, m, _ptr(cand), _ptr(cnt), C.c_void_p(stream)))
    return cand, cnt


def _stage_b(ix, d_flat, d_off, nq, max_len, k, m, gathered, gathered_cnt, dtype, stream, stride=None):
    import torch
    dev = d_flat.device
    nb = torch.zeros(nq * k, dtype=dtype, device=dev)
    nb_cnt = torch.empty(nq, dtype=torch.int32, device=dev)
    minpos = torch.full((nq * (k + 1),), MINPOS_NONE, dtype=torch.int32, device=dev)
    capi.check(capi.lib().srn_shard_stage_b_strided(ix._h, _ptr(d_flat), _ptr(d_off), nq, max_len, k, m, gathered.shape[0], int(stride or m),
                                                    _ptr(gathered), _ptr(gathered_cnt), _ptr(nb), _ptr(nb_cnt), _ptr(minpos), C.c_void_p(stream)))
    return nb, nb_cnt, minpos


def _stage_c(ix, d_flat, d_off, nq, max_len, k, m, how_many, business, nb, nb_cnt, minpos, stream):
    import torch
    dev = d_flat.device
    ids = torch.zeros(nq * how_many, dtype=torch.int64, device=dev)
    sc = torch.zeros(nq * how_many, dtype=torch.float64, device=dev)
    cnt = torch.zeros(nq, dtype=torch.int32, device=dev)
    capi.check(capi.lib().srn_shard_stage_c(ix._h, _ptr(d_flat), _ptr(d_off), nq, max_len, k, m, how_many,
                                            capi.FLAG_BUSINESS_LOGIC if business else 0, _ptr(nb), _ptr(nb_cnt), _ptr(minpos),
                                            _ptr(ids), _ptr(sc), _ptr(cnt), C.c_void_p(stream)))
    return ids, sc, cnt


def merge_topn(ids, scores, counts, how_many):
    """[G, nq, n] per-shard top-n -> [nq, n] global top-n by (score desc, item id asc); counts [G, nq] -> [nq].
    Item ids are u64 carried in int64: flipping the sign bit makes the signed order the unsigned one."""
    import torch
    G, nq, n = ids.shape
    if bool((counts == -1).any()):
        raise capi.SerenadeError(capi.SRN_ERANGE, "a query exceeded the kernel's table limits on some shard")
    valid = torch.arange(n, device=ids.device).view(1, 1, n) < counts.view(G, nq, 1)
    key = (ids ^ torch.iinfo(torch.int64).min).masked_fill(~valid, torch.iinfo(torch.int64).max)
    sc = scores.masked_fill(~valid, float("-inf"))
    key = key.permute(1, 0, 2).reshape(nq, G * n)
    sc = sc.permute(1, 0, 2).reshape(nq, G * n)
    o1 = torch.argsort(key, dim=1, stable=True)                          # secondary key: id ascending
    key, sc = torch.gather(key, 1, o1), torch.gather(sc, 1, o1)
    o2 = torch.argsort(sc, dim=1, descending=True, stable=True)          # primary key: score descending
    key, sc = torch.gather(key, 1, o2)[:, :how_many], torch.gather(sc, 1, o2)[:, :how_many]
    total = counts.to(torch.int64).sum(0).clamp(max=how_many).to(torch.int32)
    keep = torch.arange(how_many, device=ids.device).view(1, -1) < total.view(-1, 1)
    out_ids = (key ^ torch.iinfo(torch.int64).min).masked_fill(~keep, 0)
    out_sc = sc.masked_fill(~keep, 0.0)
    return out_ids.contiguous(), out_sc.contiguous(), total


# ---- lists mode: the shards exchange the posting LISTS of the batch, then every rank runs the unsharded kernels (srn_shard.hip) ----
def lists_supported(index, max_len, k, m, how_many, enable_business_logic=False):
    """True where the lists mode applies: position-set geometry (sessions of <= 8 items, m <= m_index, complete lists), business rules on or off."""
    out = C.c_int(0)
    capi.check(capi.lib().srn_shard_lists_supported(index._h, max_len, k, m, how_many, capi.FLAG_BUSINESS_LOGIC if enable_business_logic else 0, C.byref(out)))
    return bool(out.value)


def _lists_head(ix, d_flat, d_off, nq, max_len, m, stream):
    import torch
    dev = d_flat.device
    pos = torch.empty(nq * max_len * 2, dtype=torch.int64, device=dev)          # 16 bytes per (query, position)
    head = torch.empty((nq, 3), dtype=torch.int32, device=dev)                  # local (x_lo, r_max, attribute byte of the current item | -1)
    capi.check(capi.lib().srn_shard_lists_head(ix._h, _ptr(d_flat), _ptr(d_off), nq, max_len, m, _ptr(pos), _ptr(head), C.c_void_p(stream)))
    return pos, head


def _lists_count(ix, d_off, nq, max_len, pos, head, stream):
    import torch
    dev = d_off.device
    kept = torch.empty(nq * max_len, dtype=torch.int32, device=dev)
    tot = torch.empty(nq, dtype=torch.int32, device=dev)
    capi.check(capi.lib().srn_shard_lists_count(ix._h, _ptr(d_off), nq, max_len, _ptr(pos), _ptr(head), _ptr(kept), _ptr(tot), C.c_void_p(stream)))
    off = torch.cumsum(tot, 0, dtype=torch.int64)
    total = off[-1:].clone()
    off -= tot
    return kept, off, total


def _lists_copy(ix, nq, max_len, pos, kept, off, stride, stream):
    import torch
    flat = torch.empty(stride, dtype=torch.int32, device=kept.device)           # (the tail past this shard's total is never addressed)
    capi.check(capi.lib().srn_shard_lists_copy(ix._h, nq, max_len, _ptr(pos), _ptr(kept), _ptr(off), _ptr(flat), C.c_void_p(stream)))
    return flat


def _lists_predict(ix, d_flat, d_off, nq, max_len, k, m, how_many, kept_g, off_g, lists_g, head, pos, stream, business=False):
    import torch
    dev = d_flat.device
    rec = torch.empty(nq * int(capi.lib().srn_shard_lists_record_bytes(max_len)), dtype=torch.uint8, device=dev)
    ids = torch.zeros(nq * how_many, dtype=torch.int64, device=dev)
    sc = torch.zeros(nq * how_many, dtype=torch.float64, device=dev)
    cnt = torch.zeros(nq, dtype=torch.int32, device=dev)
    capi.check(capi.lib().srn_shard_lists_predict(ix._h, _ptr(d_flat), _ptr(d_off), nq, max_len, k, m, how_many, capi.FLAG_BUSINESS_LOGIC if business else 0, kept_g.shape[0], _ptr(kept_g), _ptr(off_g),
                                                  lists_g.shape[1], _ptr(lists_g), _ptr(head), _ptr(pos), _ptr(rec), _ptr(ids), _ptr(sc), _ptr(cnt), C.c_void_p(stream)))
    return ids, sc, cnt


def predict_batch_sharded_lists(index, comm, d_items_flat, d_q_off, nq, max_len, k, m, how_many, stream=None, enable_business_logic=False):
    """One batch through the LISTS pipeline on this rank (see srn_shard.hip): all-reduce(max) of the cuts, all-gather of the kept list
    prefixes, the unsharded kernels over this shard's row fragments, all-gather + merge of the per-shard top-n.  Same arguments
    and results as predict_batch_sharded; one host synchronisation per batch (the size of the exchange buffer)."""
    import torch
    if nq == 0:   # (the C side returns SRN_OK for an empty batch; `off[-1:]` of nothing has no .item())
        dev = d_items_flat.device
        return (torch.zeros((0, how_many), dtype=torch.int64, device=dev), torch.zeros((0, how_many), dtype=torch.float64, device=dev),
                torch.zeros(0, dtype=torch.int32, device=dev))
    if stream is None:
        stream = torch.cuda.current_stream(d_items_flat.device).cuda_stream
    pos, head = _lists_head(index, d_items_flat, d_q_off, nq, max_len, m, stream)
    comm.all_reduce_max(head)
    kept, off, total = _lists_count(index, d_q_off, nq, max_len, pos, head, stream)
    stride = (max(1, int(comm.all_reduce_max(total).item())) + 63) // 64 * 64
    flat = _lists_copy(index, nq, max_len, pos, kept, off, stride, stream)
    kept_g, off_g, lists_g = comm.all_gather(kept).contiguous(), comm.all_gather(off).contiguous(), comm.all_gather(flat).contiguous()
    ids, sc, cnt = _lists_predict(index, d_items_flat, d_q_off, nq, max_len, k, m, how_many, kept_g, off_g, lists_g, head, pos, stream, enable_business_logic)
    if comm.world == 1:
        return ids.view(nq, how_many), sc.view(nq, how_many), cnt
    return merge_topn(comm.all_gather(ids.view(nq, how_many)), comm.all_gather(sc.view(nq, how_many)), comm.all_gather(cnt), how_many)


def predict_batch_sharded_lists_local(shards, d_items_flat, d_q_off, nq, max_len, k, m, how_many, enable_business_logic=False):
    """The lists pipeline with all shards in ONE process on one GPU (tests): the collectives become tensor ops."""
    import torch
    stream = torch.cuda.current_stream(d_items_flat.device).cuda_stream
    h = [_lists_head(ix, d_items_flat, d_q_off, nq, max_len, m, stream) for ix in shards]
    head = torch.stack([x[1] for x in h]).max(dim=0).values.contiguous()
    c = [_lists_count(ix, d_q_off, nq, max_len, x[0], head, stream) for ix, x in zip(shards, h)]
    stride = (max(1, max(int(x[2].item()) for x in c)) + 63) // 64 * 64
    flats = [_lists_copy(ix, nq, max_len, x[0], y[0], y[1], stride, stream) for ix, x, y in zip(shards, h, c)]
    kept_g, off_g, lists_g = torch.stack([y[0] for y in c]).contiguous(), torch.stack([y[1] for y in c]).contiguous(), torch.stack(flats).contiguous()
    r = [_lists_predict(ix, d_items_flat, d_q_off, nq, max_len, k, m, how_many, kept_g, off_g, lists_g, head, x[0], stream, enable_business_logic) for ix, x in zip(shards, h)]
    return merge_topn(torch.stack([x[0].view(nq, how_many) for x in r]), torch.stack([x[1].view(nq, how_many) for x in r]),
                      torch.stack([x[2] for x in r]), how_many)


def predict_batch_sharded(index, comm, d_items_flat, d_q_off, nq, max_len, k, m, how_many, enable_business_logic=False, stream=None, mode="auto"):
    """One batch through the item-sharded index on this rank: the lists pipeline where it applies (mode "auto" / "lists"), the
    three-stage pipeline otherwise (mode "stages": sessions of > 8 items, truncated lists)."""
    if mode == "lists" or (mode == "auto" and lists_supported(index, max_len, k, m, how_many, enable_business_logic)):
        return predict_batch_sharded_lists(index, comm, d_items_flat, d_q_off, nq, max_len, k, m, how_many, stream, enable_business_logic)
    return predict_batch_sharded_stages(index, comm, d_items_flat, d_q_off, nq, max_len, k, m, how_many, enable_business_logic, stream)


def predict_batch_sharded_stages(index, comm, d_items_flat, d_q_off, nq, max_len, k, m, how_many, enable_business_logic=False, stream=None):
    """One batch through the THREE-STAGE sharded pipeline on this rank.  d_items_flat (int64 view of the u64 ids) and d_q_off
    (int32) are torch tensors on the shard's GPU and hold the SAME batch on every rank.  Returns torch tensors
    (ids int64 [nq, n] = u64 bit patterns, scores f64 [nq, n], counts int32 [nq]), identical on every rank."""
    import torch
    if stream is None:
        stream = torch.cuda.current_stream(d_items_flat.device).cuda_stream
    dtype = torch.int32 if index.slot_bytes(max_len) == 4 else torch.int64
    cand, cand_cnt = _stage_a(index, d_items_flat, d_q_off, nq, max_len, k, m, dtype, stream)
    sbytes, nbits = index.slot_info(max_len)
    stride = m
    if comm.world > 1:   # ship only what can still be among the m most recent sessions of the union
        cand, cand_cnt, stride = compact_candidates(cand, cand_cnt, nq, m, sbytes, nbits, comm.all_reduce_max)
    gathered, gathered_cnt = comm.all_gather(cand), comm.all_gather(cand_cnt)
    if bool((gathered_cnt == -1).any()):
        raise capi.SerenadeError(capi.SRN_ERANGE, "a query exceeded the session-table limits on some shard")
    nb, nb_cnt, minpos = _stage_b(index, d_items_flat, d_q_off, nq, max_len, k, m, gathered.contiguous(), gathered_cnt.contiguous(), dtype, stream, stride)
    comm.all_reduce_min(minpos)
    ids, sc, cnt = _stage_c(index, d_items_flat, d_q_off, nq, max_len, k, m, how_many, enable_business_logic, nb, nb_cnt, minpos, stream)
    g_ids = comm.all_gather(ids.view(nq, how_many))
    g_sc = comm.all_gather(sc.view(nq, how_many))
    g_cnt = comm.all_gather(cnt)
    return merge_topn(g_ids, g_sc, g_cnt, how_many)


def predict_batch_sharded_local(shards, d_items_flat, d_q_off, nq, max_len, k, m, how_many, enable_business_logic=False, compact=True):
    """All shards in ONE process on one GPU (tests, debugging): the collectives become tensor ops."""
    import torch
    stream = torch.cuda.current_stream(d_items_flat.device).cuda_stream
    dtype = torch.int32 if shards[0].slot_bytes(max_len) == 4 else torch.int64
    a = [_stage_a(ix, d_items_flat, d_q_off, nq, max_len, k, m, dtype, stream) for ix in shards]
    stride = m
    if compact and len(shards) > 1:   # compact_candidates with its two all-reduce(max) steps as tensor ops over the shards
        sbytes, nbits = shards[0].slot_info(max_len)
        loc = [_local_mth_rank(x[0], x[1], nq, m, sbytes, nbits) for x in a]
        tau = torch.stack([l[0] for l in loc]).max(dim=0).values
        kept = [_keep_at_or_above(x[0], x[1], l[1], l[2], tau, nq, m) for x, l in zip(a, loc)]
        stride = min(m, (max(1, max(int(kc.clamp(min=0).max().item()) for _, kc in kept)) + 63) // 64 * 64)
        a = [(kp[:, :stride].contiguous().view(-1), kc) for kp, kc in kept]
    gathered, gathered_cnt = torch.stack([x[0] for x in a]).contiguous(), torch.stack([x[1] for x in a]).contiguous()
    if bool((gathered_cnt == -1).any()):
        raise capi.SerenadeError(capi.SRN_ERANGE, "a query exceeded the session-table limits on some shard")
    b = [_stage_b(ix, d_items_flat, d_q_off, nq, max_len, k, m, gathered, gathered_cnt, dtype, stream, stride) for ix in shards]
    for x in b[1:]:   # stage B computes the same neighbour list on every shard
        assert torch.equal(x[0], b[0][0]) and torch.equal(x[1], b[0][1]), "shards disagree on the neighbour list"
    minpos = torch.stack([x[2] for x in b]).min(dim=0).values.contiguous()
    c = [_stage_c(ix, d_items_flat, d_q_off, nq, max_len, k, m, how_many, enable_business_logic, b[0][0], b[0][1], minpos, stream) for ix in shards]
    return merge_topn(torch.stack([x[0].view(nq, how_many) for x in c]), torch.stack([x[1].view(nq, how_many) for x in c]),
                      torch.stack([x[2] for x in c]), how_many)


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
        issue order on the caller's stream."""
        capi.check(capi.lib().srn_shard_group_set_overlap(self._h, 1 if on else 0))

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

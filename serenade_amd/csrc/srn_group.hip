// =====================================================================================
// The item-sharded index behind the C ABI: srn_shard_group_* (include/serenade_hip.h).
//
// North star (BASELINE.json): "the index shards by item-id hash across the 8 GPUs of one node with per-query partial top-k merged over RCCL
// all-gather on xGMI ... host side stays Rust".  Round 2 drove the shards from Python: the collectives went through torch.distributed and the
// scan / merge steps through the host's tensor library, so a Rust `serving` / `evaluator` host (src/endpoints/recommend_resource.rs:56,
// src/bin/evaluator.rs:58) bound to this library could not drive more than one GPU.  Here the whole LISTS pipeline of srn_shard.hip runs inside
// one call, RCCL called from C++ (dlopen: whichever librccl the process already has, else /opt/rocm's):
//
//   exchange stream (its own communicator, overlaps the previous batch's kernels on the caller's stream)
//     head kernel -> all-reduce(max) of (x_lo, r_max, attribute)                        12 B per query
//     count kernel -> all-gather of the kept counts -> offsets + per-shard totals        4 * max_len B per query and shard
//     ONE short host synchronisation: the totals size the next step                       (the GPU keeps running the previous batch meanwhile)
//     copy kernel -> variable-length exchange of the kept list prefixes                   grouped send / recv: every rank ships exactly what it holds
//   caller's stream (second communicator)
//     prep records against the gathered lists -> the unsharded launch sequence over this shard's row fragments
//     all-gather of the per-shard top-n -> merge kernel (score desc, item id asc)         (16 n + 4) B per query and shard
//
// Three transports behind one interface: RCCL (one process per GPU), host callbacks (an application's own transport; the tests drive two
// processes over gloo with it), and an in-process group (all shards of the group on this process's device: collectives degenerate to kernels --
// tests, and capacity experiments on one GPU).  Same kernels, same bytes in all three.
// =====================================================================================
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "srn_runtime.h"
#include "srn_hipsync.h"

namespace srn {

namespace {
template <typename F> int guarded(F f) {   // never let an exception cross the C boundary
    try { return f(); }
    catch (const std::bad_alloc&) { return fail(SRN_ENOMEM, "out of host memory"); }
    catch (const std::exception& e) { return fail(SRN_EINVAL, std::string("internal error: ") + e.what()); }
    catch (...) { return fail(SRN_EINVAL, "internal error"); }
}

// ---- RCCL through dlopen: no link-time dependency, and ONE RCCL per process (a host that already loaded librccl -- torch does -- shares it) ----
struct RcclApi {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;   // (optional: tears a communicator down without waiting for its peers)
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    std::string err;
};
RcclApi* rccl() {
    static RcclApi api; static std::once_flag once;
    std::call_once(once, [] {
        const char* env = getenv("SRN_RCCL_LIB");
        void* h = env ? dlopen(env, RTLD_NOW | RTLD_GLOBAL) : nullptr;
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);   // already in the process (torch's bundled copy, or the host's own)
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) { api.err = std::string("cannot load librccl: ") + (dlerror() ? dlerror() : "?"); return; }
        api.h = h;
        auto sym = [&](const char* n) -> void* { void* p = dlsym(h, n); if (!p && api.err.empty()) api.err = std::string("librccl lacks ") + n; return p; };
        api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId"); api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
        api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy"); api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
        api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce"); api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
        api.Send = (decltype(api.Send))sym("ncclSend"); api.Recv = (decltype(api.Recv))sym("ncclRecv");
        api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart"); api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
        api.CommAbort = (decltype(api.CommAbort))dlsym(h, "ncclCommAbort");
    });
    return &api;
}
#define NCCL_TRY(expr)                                                                                                              \
    do {                                                                                                                            \
        ncclResult_t r_ = (expr);                                                                                                   \
        if (r_ != ncclSuccess) return fail(SRN_EHIP, std::string(#expr) + ": " + (rccl()->GetErrorString ? rccl()->GetErrorString(r_) : "rccl error")); \
    } while (0)

struct Slot {   // everything one batch in flight needs (grow-only); two slots: batch i + 1's exchange overlaps batch i's kernels
    std::vector<char*> pos; std::vector<size_t> pos_bytes;   // per local shard
    char* head = nullptr; size_t head_bytes = 0;             // [nq][3] int32 (in-process group: + one scratch copy behind it)
    char* kept = nullptr; size_t kept_bytes = 0;             // [G][nq * max_len] u32
    char* tot = nullptr; size_t tot_bytes = 0;               // [nq] int32 (the count kernel's per-query totals; unused here)
    char* off = nullptr; size_t off_bytes = 0;               // [G][nq] int64
    char* small = nullptr; size_t small_bytes = 0;           // tot_dev [G] u64 | base_dev [G] u64
    unsigned long long* tot_host = nullptr;                  // pinned, device-mapped: [G] totals | [G] bases
    char* lists = nullptr; size_t lists_bytes = 0;           // the gathered list prefixes, segment g at base[g]
    char* records = nullptr; size_t records_bytes = 0;
    char* part = nullptr; size_t part_bytes = 0;             // [G] blocks: ids | scores | counts
    // three-stage pipeline: [G] candidate blocks + [G][nq] counts, the neighbour list, first-match positions (+ one scratch copy for an in-process group), overflow flags
    char* cand = nullptr; size_t cand_bytes = 0; char* cand_cnt = nullptr; size_t cand_cnt_bytes = 0;
    char* nb = nullptr; size_t nb_bytes = 0; char* nb_cnt = nullptr; size_t nb_cnt_bytes = 0; char* minpos = nullptr; size_t minpos_bytes = 0; char* flagq = nullptr; size_t flagq_bytes = 0;
    // neighbours pipeline: [G * per][k + 1] words (a query's row: K | K packed slots), and every local shard's prep records of the whole batch
    char* xpos = nullptr; size_t xpos_bytes = 0;   // the neighbours as posting positions (the streaming back end's exchange format)
    char* xchg = nullptr; size_t xchg_bytes = 0; std::vector<char*> nrec; std::vector<size_t> nrec_bytes;
    std::vector<char*> nord; std::vector<size_t> nord_bytes; std::vector<const unsigned long long*> nord_ptr;   // ... and the batch's serving order (keys, sorted keys, scratch)
    hipEvent_t e_done = nullptr;
};
}  // namespace

}  // namespace srn

using namespace srn;

struct srn_shard_group {
    enum Kind { RCCL, CALLBACKS, LOCAL } kind = LOCAL;
    int rank = 0, world = 1, device = 0;
    std::vector<const srn_index*> shards;   // LOCAL: all of them; otherwise this rank's one
    ncclComm_t comm[2] = {nullptr, nullptr};   // [0] exchange stream, [1] caller's stream: operations on one communicator serialise in issue order
    srn_shard_comm_t cb{};
    hipStream_t s_x = nullptr; hipEvent_t e_in = nullptr, e_x = nullptr;
    int* agree_dev = nullptr;   // three flags of set_postings' agreement
    char* pres_all = nullptr; size_t pres_all_bytes = 0; uint8_t* nb_pbytes = nullptr;   // the shards' presence bitmaps (all-gathered at set_postings) and the byte per session made of them: bit g = shard g holds an item of the session
    bool stream_ok = false;   // every shard of this rank holds its fragments in the posting order of g->postings (set_postings): batches of the streaming form's shape exchange positions
    bool overlap = false, no_direct = false;   // overlap: opt-in (srn_shard_group_set_overlap) -- two communicators with collectives in flight at once have never been soaked on more than one GPU
    // A batch that failed after its first collective was issued leaves the peers' collectives without their partner: the group is BROKEN on this rank from then on (every
    // further call fails at once with SRN_ESTATE, the RCCL communicators are aborted so that nothing of this rank keeps a peer waiting), and the peers find out through their
    // own transport's error / srn_shard_group_wait's timeout.  `issued`: a collective of the current batch has been handed to the transport.
    bool broken = false, issued = false; std::string broken_why;
    hipEvent_t e_last = nullptr;           // the end of the most recent batch on the caller's stream (srn_shard_group_wait)
    const srn_index* postings = nullptr;   // the replicated posting lists (srn_shard_group_set_postings): batches of the fast kernel's shape take the neighbours pipeline
    uint64_t postings_max_row_len = 0;     // ... and the longest row of the WHOLE index: what the choice of pipeline is derived from (the same on every rank)
    Slot slot[2];
    uint64_t calls = 0;
    uint64_t st_queries = 0, st_bytes_head = 0, st_bytes_kept = 0, st_bytes_lists = 0, st_bytes_results = 0, st_lists_max = 0;
    uint64_t st_nb_batches = 0, st_bytes_nb = 0;
    bool timing = false; hipEvent_t e_t[6] = {}; float last_ms[3] = {0, 0, 0};   // SRN_GROUP_TIMING (measurement aid): local shard 0's prep + front end | back end | merge, of the last neighbours batch
    uint64_t st_stage_batches = 0, st_bytes_stage_cand = 0, st_bytes_stage_minpos = 0;   // batches that took the three-stage pipeline
    std::mutex mu;   // one batch is issued at a time per group (the collectives must be issued in the same order on every rank anyway)
};

namespace {
uint32_t G_of(const srn_shard_group* g) { return g->kind == srn_shard_group::LOCAL ? (uint32_t)g->shards.size() : (uint32_t)g->world; }

// ---- the three collectives.  channel 0 = exchange stream, 1 = caller's stream ----
int all_reduce_max_i32(srn_shard_group* g, int channel, int* buf, size_t count, hipStream_t st) {
    if (g->kind != srn_shard_group::LOCAL) g->issued = true;   // (an in-process group has no peers to leave waiting)
    if (g->kind == srn_shard_group::RCCL) NCCL_TRY(rccl()->AllReduce(buf, buf, count, ncclInt32, ncclMax, g->comm[channel], st));
    else if (g->kind == srn_shard_group::CALLBACKS) { const int rc = g->cb.all_reduce_max_i32(g->cb.user, channel, buf, count, st); if (rc) return fail(rc, "the application's all-reduce callback failed"); }
    return SRN_OK;
}
int all_reduce_min_i32(srn_shard_group* g, int channel, int* buf, size_t count, hipStream_t st) {
    if (g->kind != srn_shard_group::LOCAL) g->issued = true;   // (an in-process group has no peers to leave waiting)
    if (g->kind == srn_shard_group::RCCL) NCCL_TRY(rccl()->AllReduce(buf, buf, count, ncclInt32, ncclMin, g->comm[channel], st));
    else if (g->kind == srn_shard_group::CALLBACKS) { const int rc = g->cb.all_reduce_min_i32(g->cb.user, channel, buf, count, st); if (rc) return fail(rc, "the application's all-reduce(min) callback failed"); }
    return SRN_OK;
}
int all_gather_blocks(srn_shard_group* g, int channel, char* buf, size_t block_bytes, hipStream_t st) {   // block `rank` in place
    if (g->kind != srn_shard_group::LOCAL) g->issued = true;   // (an in-process group has no peers to leave waiting)
    if (g->kind == srn_shard_group::RCCL) NCCL_TRY(rccl()->AllGather(buf + (size_t)g->rank * block_bytes, buf, block_bytes, ncclChar, g->comm[channel], st));
    else if (g->kind == srn_shard_group::CALLBACKS) { const int rc = g->cb.all_gather(g->cb.user, channel, buf, block_bytes, st); if (rc) return fail(rc, "the application's all-gather callback failed"); }
    return SRN_OK;
}
int all_gather_v(srn_shard_group* g, int channel, char* buf, const unsigned long long* byte_off, const unsigned long long* byte_cnt, hipStream_t st) {   // segment `rank` in place
    if (g->kind != srn_shard_group::LOCAL) g->issued = true;   // (an in-process group has no peers to leave waiting)
    if (g->kind == srn_shard_group::RCCL) {
        // every rank ships exactly the entries it holds: grouped point-to-point pairs (xGMI is point-to-point: 7 links per GPU, one per peer)
        NCCL_TRY(rccl()->GroupStart());
        for (int p = 0; p < g->world; ++p) {
            if (p == g->rank) continue;
            if (byte_cnt[g->rank]) NCCL_TRY(rccl()->Send(buf + byte_off[g->rank], byte_cnt[g->rank], ncclChar, p, g->comm[channel], st));
            if (byte_cnt[p]) NCCL_TRY(rccl()->Recv(buf + byte_off[p], byte_cnt[p], ncclChar, p, g->comm[channel], st));
        }
        NCCL_TRY(rccl()->GroupEnd());
    } else if (g->kind == srn_shard_group::CALLBACKS) {
        const int rc = g->cb.all_gather_v(g->cb.user, channel, buf, (const uint64_t*)byte_off, (const uint64_t*)byte_cnt, st);
        if (rc) return fail(rc, "the application's all-gather-v callback failed");
    }
    return SRN_OK;
}

void slot_free(Slot& s) {
    for (char* p : s.pos) if (p) hipFree(p);
    for (char* p : s.nrec) if (p) hipFree(p);
    for (char* p : s.nord) if (p) hipFree(p);
    if (s.xchg) hipFree(s.xchg);
    if (s.xpos) hipFree(s.xpos);
    for (char* p : {s.head, s.kept, s.tot, s.off, s.small, s.lists, s.records, s.part, s.cand, s.cand_cnt, s.nb, s.nb_cnt, s.minpos, s.flagq}) if (p) hipFree(p);
    if (s.tot_host) hipHostFree(s.tot_host);
    if (s.e_done) hipEventDestroy(s.e_done);
    s = Slot();
}

int group_init_common(srn_shard_group* g) {
    HIP_TRY(hipSetDevice(g->device));
    { int prio_lo = 0, prio_hi = 0; (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);   // the exchange stream: highest priority = a hardware queue of its own, and its small kernels and RCCL's
      HIP_TRY(hipStreamCreateWithPriority(&g->s_x, hipStreamNonBlocking, prio_hi)); }          // are dispatched ahead of the previous batch's persistent workgroups on the caller's stream
    HIP_TRY(hipEventCreateWithFlags(&g->e_in, hipEventDisableTiming)); HIP_TRY(hipEventCreateWithFlags(&g->e_x, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&g->e_last, hipEventDisableTiming));
    const uint32_t G = G_of(g);
    for (Slot& s : g->slot) {
        HIP_TRY(hipEventCreateWithFlags(&s.e_done, hipEventDisableTiming));
        HIP_TRY(hipHostMalloc((void**)&s.tot_host, (size_t)G * 16, hipHostMallocMapped));
        s.pos.assign(g->shards.size(), nullptr); s.pos_bytes.assign(g->shards.size(), 0);
        s.nrec.assign(g->shards.size(), nullptr); s.nrec_bytes.assign(g->shards.size(), 0);
        s.nord.assign(g->shards.size(), nullptr); s.nord_bytes.assign(g->shards.size(), 0); s.nord_ptr.assign(g->shards.size(), nullptr);
    }
    if (const char* e = getenv("SRN_GROUP_OVERLAP")) g->overlap = atoi(e) != 0;
    g->no_direct = getenv("SRN_GROUP_NO_DIRECT") != nullptr;
    if (getenv("SRN_GROUP_TIMING")) { g->timing = true; for (auto& e : g->e_t) HIP_TRY(hipEventCreate(&e)); }
    return SRN_OK;
}

int check_shard(const srn_index* ix, uint32_t want_shard, uint32_t n_shards) {
    if (!ix) return fail(SRN_EINVAL, "null shard");
    if (!ix->dev) return fail(SRN_ENODEV, "shard has no device attached; there is no CPU fallback behind this ABI");
    { int rc0 = check_has_rows(ix->flat, "srn_shard_group_create"); if (rc0) return rc0; }
    if (ix->flat.n_shards != n_shards || ix->flat.shard != want_shard) return fail(SRN_EINVAL, "the index is not shard " + std::to_string(want_shard) + " of " + std::to_string(n_shards));
    return SRN_OK;
}

// The three-stage pipeline (round 1's kernels, `STAGE` instantiations of vmis_predict_kernel) for what the lists pipeline does not serve -- sessions of > 8 items,
// m > m_index, incomplete posting lists: stage A (this shard's candidates, locally cut to m) -> all-gather -> stage B (global cuts: the neighbour list, identical on every
// rank; partial first-match positions over the evolving items this shard owns) -> all-reduce(min) -> stage C (accumulate over the shard's row fragments, exact top-n among
// the items it owns) -> all-gather -> merge.  Everything on the caller's stream, no host synchronisation; a query whose session table overflowed on some shard comes back
// with out_counts = 0xFFFFFFFF like on the unsharded path.  (serenade_amd/sharded.py drove these stages from Python until round 3, with a compaction of the candidate slabs
// that is not repeated here: the lists pipeline is the fast path, this is the complete one.)
int group_predict_stages(srn_shard_group* g, const LaunchParams& p, uint64_t* d_out_ids, double* d_out_scores, uint32_t* d_out_counts, hipStream_t user) {
    const uint32_t G = G_of(g), nq = p.nq, n = p.how_many;
    const bool local = g->kind == srn_shard_group::LOCAL;
    if (g->kind == srn_shard_group::CALLBACKS && !g->cb.all_reduce_min_i32) return fail(SRN_EINVAL, "this batch needs the three-stage pipeline, and the group's transport has no all_reduce_min_i32");
    Slot& s = g->slot[g->calls & 1u];
    uint32_t num_bits = 0;
    const int sb = device_slot_bytes(g->shards[0]->dev, g->shards[0]->flat, p.max_len, &num_bits);
    if (sb < 0) return SRN_ERANGE;
    const size_t cand_block = (size_t)nq * p.m * (size_t)sb /* exactly [nq][m] slots: stage B indexes the gathered buffer as [G][nq][m] */, block_bytes = ((size_t)nq * n * 16 + (size_t)nq * 4 + 255) / 256 * 256;
    const size_t mp_count = (size_t)nq * ((size_t)p.k + 1);
    {
        int rc = ensure(&s.cand, &s.cand_bytes, (size_t)G * cand_block);
        if (!rc) rc = ensure(&s.cand_cnt, &s.cand_cnt_bytes, (size_t)G * nq * 4);
        if (!rc) rc = ensure(&s.nb, &s.nb_bytes, (size_t)nq * p.k * (size_t)sb);
        if (!rc) rc = ensure(&s.nb_cnt, &s.nb_cnt_bytes, (size_t)nq * 4);
        if (!rc) rc = ensure(&s.minpos, &s.minpos_bytes, mp_count * 4 * 2);
        if (!rc) rc = ensure(&s.flagq, &s.flagq_bytes, (size_t)nq * 4);
        if (!rc) rc = ensure(&s.part, &s.part_bytes, (size_t)G * block_bytes);
        if (rc) return rc;
    }
    uint32_t* cnt_g = (uint32_t*)s.cand_cnt; int* minpos = (int*)s.minpos;
    if (g->calls >= 2) HIP_TRY(hipStreamWaitEvent(user, s.e_done, 0));   // (the batch that used this slot two calls ago -- on the exchange stream, or on ANOTHER caller's stream -- is done with its buffers)
    // stage A
    for (size_t i = 0; i < g->shards.size(); ++i) {
        const uint32_t gi = local ? (uint32_t)i : (uint32_t)g->rank;
        ShardIO sh{}; sh.cand = s.cand + (size_t)gi * cand_block; sh.cand_cnt = cnt_g + (size_t)gi * nq;
        int rc = device_shard_stage(g->shards[i]->dev, g->shards[i]->flat, 1, p, sh, user); if (rc) return rc;
    }
    { int rc = all_gather_blocks(g, 1, s.cand, cand_block, user); if (rc) return rc; }
    { int rc = all_gather_blocks(g, 1, s.cand_cnt, (size_t)nq * 4, user); if (rc) return rc; }
    HIP_TRY(launch_shard_scrub_counts(user, cnt_g, G, nq, (uint32_t*)s.flagq));
    // stage B (every rank computes the same neighbour list; its own partial first-match positions)
    for (size_t i = 0; i < g->shards.size(); ++i) {
        int* mp_i = i == 0 ? minpos : minpos + mp_count;
        HIP_TRY(hipMemsetAsync(s.nb, 0, (size_t)nq * p.k * (size_t)sb, user));
        HIP_TRY(launch_shard_fill_i32(user, mp_i, 0x7FFFFFFF, mp_count));
        ShardIO sh{}; sh.gathered = s.cand; sh.gathered_cnt = cnt_g; sh.n_shards = G; sh.gathered_stride = 0;
        sh.nb = s.nb; sh.nb_cnt = (uint32_t*)s.nb_cnt; sh.minpos = mp_i;
        int rc = device_shard_stage(g->shards[i]->dev, g->shards[i]->flat, 2, p, sh, user); if (rc) return rc;
        if (i > 0) HIP_TRY(launch_shard_min(user, minpos, mp_i, mp_count));
    }
    { int rc = all_reduce_min_i32(g, 1, minpos, mp_count, user); if (rc) return rc; }
    // stage C
    for (size_t i = 0; i < g->shards.size(); ++i) {
        const uint32_t gi = local ? (uint32_t)i : (uint32_t)g->rank;
        char* blk = s.part + (size_t)gi * block_bytes;
        LaunchParams pi = p;
        const bool direct = G == 1;
        pi.out_ids = direct ? d_out_ids : (uint64_t*)blk; pi.out_scores = direct ? d_out_scores : (double*)(blk + (size_t)nq * n * 8); pi.out_counts = direct ? d_out_counts : (uint32_t*)(blk + (size_t)nq * n * 16);
        HIP_TRY(hipMemsetAsync(pi.out_ids, 0, (size_t)nq * n * 8, user)); HIP_TRY(hipMemsetAsync(pi.out_scores, 0, (size_t)nq * n * 8, user)); HIP_TRY(hipMemsetAsync(pi.out_counts, 0, (size_t)nq * 4, user));
        ShardIO sh{}; sh.nb = s.nb; sh.nb_cnt = (uint32_t*)s.nb_cnt; sh.minpos = minpos;
        int rc = device_shard_stage(g->shards[i]->dev, g->shards[i]->flat, 3, pi, sh, user); if (rc) return rc;
    }
    if (G > 1) {
        int rc = all_gather_blocks(g, 1, s.part, block_bytes, user); if (rc) return rc;
        HIP_TRY(launch_shard_merge_topn(user, s.part, block_bytes, G, nq, n, d_out_ids, d_out_scores, d_out_counts));
    }
    HIP_TRY(launch_shard_mark(user, (const uint32_t*)s.flagq, nq, d_out_counts));
    HIP_TRY(hipEventRecord(s.e_done, user));
    ++g->calls;
    g->st_queries += nq; g->st_bytes_stage_cand += (uint64_t)(cand_block + (size_t)nq * 4) * (local ? G : 1); g->st_bytes_stage_minpos += (uint64_t)mp_count * 4;
    g->st_bytes_results += G > 1 ? (uint64_t)block_bytes * (local ? G : 1) : 0; ++g->st_stage_batches;
    return SRN_OK;
}

// The NEIGHBOURS pipeline (round 4): posting lists replicated (g->postings), rows sharded.  Rank r fronts the queries [r per, (r + 1) per): prep records against the
// replicated lists, the fast kernel's front end (lists -> merge tree -> cuts) -> neighbour lists in its block of the exchange buffer; ONE all-gather of fixed-size blocks
// (no host synchronisation anywhere in the batch); then every rank runs the fast kernel's back end (walks, thresholds) over ALL queries on its own row fragments, the
// general kernel behind it for the few queries no front end could take (marker in the exchange row: every rank then does that query's candidate work itself, against the
// replicated lists), the finish kernels, and the usual all-gather + merge of the per-shard top-n.  The candidate work of a batch is done ONCE per group, not once per rank.
int group_predict_neighbours(srn_shard_group* g, const LaunchParams& p_in, bool resident, uint64_t* d_out_ids, double* d_out_scores, uint32_t* d_out_counts, hipStream_t user) {
    const uint32_t G = G_of(g), nq = p_in.nq, n = p_in.how_many;
    const bool local = g->kind == srn_shard_group::LOCAL;
    LaunchParams p = p_in; p.out_ids = nullptr; p.out_scores = nullptr; p.out_counts = nullptr;
    Slot& s = g->slot[g->calls & 1u];
    const bool overlap = g->overlap && !local;
    hipStream_t sx = overlap ? g->s_x : user;
    const bool pbytes = g->nb_pbytes != nullptr && device_shard_nb_presence_wanted() && p.max_len <= 8;   // (rank-invariant: the knob, the batch shape, what set_postings built on every rank)
    const uint32_t rec_stride = device_prep_stride(p.max_len), xstride = p.k + 1u + (pbytes ? (p.k + 3u) / 4u : 0u), per = (nq + G - 1) / G;
    const size_t block_bytes = ((size_t)nq * n * 16 + (size_t)nq * 4 + 255) / 256 * 256, xblock = (size_t)per * xstride * 4;
    // Streaming form (round 5): the exchange carries, instead of the neighbour slots, WHERE the neighbours sit in the query's posting lists -- a third of the bytes --, and
    // every rank's back end streams its fragments in posting order.  Chosen from rank-invariant inputs (batch shape, knobs) + what set_postings settled for this rank.
    // streaming or gather form (rank-invariant inputs only: the knob, the group's kind, the overlap switch, what set_postings agreed on): see Knobs::sback_stream_mode
    // AUTO: the form with the lower modelled batch time when the exchange is on the critical path -- the streaming form saves ((k + 1) - stride) * 4 / G bytes per query and
    // link, and costs 8.5 ns more compute per query and rank (config 3, G = 8: 2.86 against 1.75 ms per 131 072 queries since the gather form's back end was rebuilt in round 6
    // -- 5.9 ns until then --, bench.py's exchange_model); SRN_XGMI_GBPS = the
    // link's rate per direction (default 76.8: a 153.6 GB/s link counted both ways).  Overlapped, the gather form's smaller compute decides.
    const int smode = knobs().sback_stream_mode;
    const uint32_t pstride_all = g->stream_ok && smode != 0 ? device_shard_nb_positions_stride(p) : 0u;
    bool want_stream = smode == 1;
    if (smode < 0 && pstride_all != 0u && g->kind != srn_shard_group::LOCAL && !g->overlap) {
        const double saved_ns = ((double)(p.k + 1u) - (double)pstride_all) * 4.0 / (double)G / knobs().xgmi_gbps;   // bytes / (GB/s) = ns
        want_stream = saved_ns > 8.5;
    }
    const uint32_t pstride = want_stream ? pstride_all : 0u;
    const bool positions = pstride != 0u;
    const size_t pblock = (size_t)per * pstride * 4;
    {
        int rc = ensure(&s.xchg, &s.xchg_bytes, (size_t)G * xblock);
        if (!rc && positions) rc = ensure(&s.xpos, &s.xpos_bytes, (size_t)G * pblock);
        for (size_t i = 0; i < g->shards.size() && !rc; ++i) rc = ensure(&s.nrec[i], &s.nrec_bytes[i], (size_t)nq * rec_stride);
        if (!rc) rc = ensure(&s.part, &s.part_bytes, (size_t)G * block_bytes);
        if (rc) return rc;
    }
    if (overlap) {
        if (!resident) { HIP_TRY(hipEventRecord(g->e_in, user)); HIP_TRY(hipStreamWaitEvent(sx, g->e_in, 0)); }
        if (g->calls >= 2) HIP_TRY(hipStreamWaitEvent(sx, s.e_done, 0));
    }
    if (g->calls >= 2) HIP_TRY(hipStreamWaitEvent(user, s.e_done, 0));
    DeviceState* post = g->postings->dev;
    for (size_t i = 0; i < g->shards.size(); ++i) {
        const uint32_t gi = local ? (uint32_t)i : (uint32_t)g->rank;
        if (g->timing && i == 0) HIP_TRY(hipEventRecord(g->e_t[0], sx));
        int rc = device_shard_nb_prep(g->shards[i]->dev, post, p, s.nrec[i], sx, &s.nord[i], &s.nord_bytes[i], &s.nord_ptr[i]); if (rc) return rc;
        const uint32_t q_lo = std::min<uint64_t>(nq, (uint64_t)gi * per), q_hi = std::min<uint64_t>(nq, (uint64_t)q_lo + per);
        rc = device_shard_nb_front(g->shards[i]->dev, g->shards[i]->flat, post, p, s.nrec[i], (uint32_t*)s.xchg, xstride, q_lo, q_hi, sx); if (rc) return rc;
        if (pbytes && !positions) { rc = device_shard_nb_presence(g->shards[i]->dev, g->shards[i]->flat, p, s.nrec[i], (uint32_t*)s.xchg, xstride, g->nb_pbytes, q_lo, q_hi, sx); if (rc) return rc; }
        if (positions) { rc = device_shard_nb_positions(g->shards[i]->dev, g->shards[i]->flat, post, p, s.nrec[i], (const uint32_t*)s.xchg, xstride, (uint32_t*)s.xpos, pstride, q_lo, q_hi, sx); if (rc) return rc; }
        if (g->timing && i == 0) HIP_TRY(hipEventRecord(g->e_t[1], sx));
    }
    { int rc = all_gather_blocks(g, 0, positions ? s.xpos : s.xchg, positions ? pblock : xblock, sx); if (rc) return rc; }
    if (overlap) { HIP_TRY(hipEventRecord(g->e_x, sx)); HIP_TRY(hipStreamWaitEvent(user, g->e_x, 0)); }
    for (size_t i = 0; i < g->shards.size(); ++i) {
        const uint32_t gi = local ? (uint32_t)i : (uint32_t)g->rank;
        char* blk = s.part + (size_t)gi * block_bytes;
        LaunchParams pi = p;
        const bool direct = G == 1;   // one shard: its top-n IS the result
        pi.out_ids = direct ? d_out_ids : (uint64_t*)blk; pi.out_scores = direct ? d_out_scores : (double*)(blk + (size_t)nq * n * 8); pi.out_counts = direct ? d_out_counts : (uint32_t*)(blk + (size_t)nq * n * 16);
        if (g->timing && i == 0) HIP_TRY(hipEventRecord(g->e_t[2], user));
        HIP_TRY(hipMemsetAsync(pi.out_ids, 0, (size_t)nq * n * 8, user)); HIP_TRY(hipMemsetAsync(pi.out_scores, 0, (size_t)nq * n * 8, user));
        int rc = device_shard_nb_back(g->shards[i]->dev, g->shards[i]->flat, post, pi, s.nrec[i], (uint32_t*)(positions ? s.xpos : s.xchg), positions ? pstride : xstride, user, s.nord_ptr[i], positions, pbytes && !positions); if (rc) return rc;
        if (g->timing && i == 0) HIP_TRY(hipEventRecord(g->e_t[3], user));
    }
    if (G > 1) {
        int rc = all_gather_blocks(g, 1, s.part, block_bytes, user); if (rc) return rc;
        if (g->timing) HIP_TRY(hipEventRecord(g->e_t[4], user));
        HIP_TRY(launch_shard_merge_topn(user, s.part, block_bytes, G, nq, n, d_out_ids, d_out_scores, d_out_counts));
        if (g->timing) HIP_TRY(hipEventRecord(g->e_t[5], user));
    }
    HIP_TRY(hipEventRecord(s.e_done, user));
    if (g->timing) {   // (measurement runs only: synchronises)
        HIP_TRY(hipEventSynchronize(s.e_done));
        HIP_TRY(hipEventElapsedTime(&g->last_ms[0], g->e_t[0], g->e_t[1])); HIP_TRY(hipEventElapsedTime(&g->last_ms[1], g->e_t[2], g->e_t[3]));
        g->last_ms[2] = 0.f; if (G > 1) HIP_TRY(hipEventElapsedTime(&g->last_ms[2], g->e_t[4], g->e_t[5]));
    }
    ++g->calls;
    g->st_queries += nq; ++g->st_nb_batches; g->st_bytes_nb += G > 1 ? (uint64_t)(positions ? pblock : xblock) * (local ? G : 1) : 0;
    g->st_bytes_results += G > 1 ? (uint64_t)block_bytes * (local ? G : 1) : 0;
    return SRN_OK;
}

int group_predict_locked(srn_shard_group* g, const uint64_t* d_items_flat, const uint32_t* d_q_off, size_t nq_, size_t max_len_hint, size_t k, size_t m, size_t how_many,
                         unsigned flags, uint64_t* d_out_ids, double* d_out_scores, uint32_t* d_out_counts, hipStream_t user);
// The group gives up: nothing of this rank may keep a peer waiting, and nothing further is issued.  RCCL: both communicators are aborted (ncclCommAbort ends their kernels
// on this GPU; a peer blocked in the matching collective sees the connection go and fails, or runs into its own srn_shard_group_wait timeout); callbacks: the
// application's transport is the application's to tear down -- it learns of the failure from the return code.
void group_break(srn_shard_group* g, const std::string& why) {
    if (g->broken) return;
    g->broken = true; g->broken_why = why;
    if (g->kind == srn_shard_group::RCCL)
        for (auto& c : g->comm) if (c) { if (rccl()->CommAbort) rccl()->CommAbort(c); else if (rccl()->CommDestroy) rccl()->CommDestroy(c); c = nullptr; }
}
int group_predict(srn_shard_group* g, const uint64_t* d_items_flat, const uint32_t* d_q_off, size_t nq_, size_t max_len_hint, size_t k, size_t m, size_t how_many,
                  unsigned flags, uint64_t* d_out_ids, double* d_out_scores, uint32_t* d_out_counts, hipStream_t user) {
    std::lock_guard<std::mutex> lk(g->mu);
    if (g->broken) return fail(SRN_ESTATE, "the shard group is unusable since an earlier batch failed (" + g->broken_why + "): free it and create a new one on every rank");
    g->issued = false;
    const int rc = group_predict_locked(g, d_items_flat, d_q_off, nq_, max_len_hint, k, m, how_many, flags, d_out_ids, d_out_scores, d_out_counts, user);
    if (rc == SRN_OK) { if (g->e_last) (void)hipEventRecord(g->e_last, user); return rc; }
    // A refusal that every rank makes alike before anything was issued (a parameter beyond a limit) leaves the group as it was; anything else -- a collective that failed,
    // a HIP error, an allocation that failed on THIS rank -- may have left the peers in a collective of this batch, or be about to.
    const bool peers = g->kind != srn_shard_group::LOCAL && g->world > 1;
    if (g->issued || (peers && rc != SRN_EINVAL && rc != SRN_ERANGE)) { const std::string why = last_error_string(); group_break(g, why); set_error(why); }
    return rc;
}
int group_predict_locked(srn_shard_group* g, const uint64_t* d_items_flat, const uint32_t* d_q_off, size_t nq_, size_t max_len_hint, size_t k, size_t m, size_t how_many,
                         unsigned flags, uint64_t* d_out_ids, double* d_out_scores, uint32_t* d_out_counts, hipStream_t user) {
    HIP_TRY(hipSetDevice(g->device));
    const uint32_t G = G_of(g), nq = (uint32_t)nq_, ML = (uint32_t)max_len_hint, n = (uint32_t)how_many;
    const bool local = g->kind == srn_shard_group::LOCAL;
    LaunchParams p{};
    const bool resident = (flags & SRN_FLAG_INPUTS_RESIDENT) != 0u;   // the inputs do not hang on the caller's stream: this batch's exchange may start while the previous batch still runs there
    flags &= ~(unsigned)SRN_FLAG_INPUTS_RESIDENT;
    p.nq = nq; p.k = (uint32_t)k; p.m = (uint32_t)m; p.how_many = n; p.flags = flags; p.max_len = ML; p.items_flat = d_items_flat; p.q_off = d_q_off;
    p.out_ids = d_out_ids; p.out_scores = d_out_scores; p.out_counts = d_out_counts;
    for (const srn_index* ix : g->shards)
        if (!device_shard_lists_supported(ix->dev, ix->flat, p)) return group_predict_stages(g, p, d_out_ids, d_out_scores, d_out_counts, user);   // (every rank holds the same index parameters: the same choice everywhere)
    if (g->postings && G > 1) {   // (the same index parameters and the same batch shape on every rank: the same choice everywhere.  A group of ONE shard gains nothing from
                                 //  dividing the candidate work: its lists pipeline reads the lists in place and runs the fused kernel -- 23.4 against 10 + 16.7 ms per 2^20 queries)
        // (rank-invariant inputs only: the batch shape, the index parameters, the whole index's longest row; that every shard has its packed rows was checked by set_postings.
        //  a choice made from rank-local state would pair one rank's all-reduce with its peers' all-gather on the same communicator)
        bool all = true;
        for (const srn_index* ix : g->shards) all = all && device_fast_eligible(ix->dev, ix->flat, p, std::max<uint64_t>(1, g->postings_max_row_len));
        if (all) return group_predict_neighbours(g, p, resident, d_out_ids, d_out_scores, d_out_counts, user);
    }
    p.out_ids = nullptr; p.out_scores = nullptr; p.out_counts = nullptr;
    Slot& s = g->slot[g->calls & 1u];
    const bool overlap = g->overlap && !local;
    hipStream_t sx = overlap ? g->s_x : user;
    // ---- buffers (grow-only; growing synchronises the device, like every hipFree) ----
    const uint32_t rec_stride = device_prep_stride(ML);
    const size_t block_bytes = ((size_t)nq * n * 16 + (size_t)nq * 4 + 255) / 256 * 256;
    {
        int rc = SRN_OK;
        for (size_t i = 0; i < g->shards.size() && !rc; ++i) rc = ensure(&s.pos[i], &s.pos_bytes[i], (size_t)nq * ML * 16);
        if (!rc) rc = ensure(&s.head, &s.head_bytes, (size_t)nq * 12 * 2);
        if (!rc) rc = ensure(&s.kept, &s.kept_bytes, (size_t)G * nq * ML * 4);
        if (!rc) rc = ensure(&s.tot, &s.tot_bytes, (size_t)nq * 4);
        if (!rc) rc = ensure(&s.off, &s.off_bytes, (size_t)G * nq * 8);
        if (!rc) rc = ensure(&s.small, &s.small_bytes, (size_t)G * 16 + (size_t)G * ((nq + 1023) / 1024) * 8);
        if (!rc) rc = ensure(&s.records, &s.records_bytes, (size_t)nq * rec_stride);
        if (!rc) rc = ensure(&s.part, &s.part_bytes, (size_t)G * block_bytes);
        if (rc) return rc;
    }
    int* head = (int*)s.head; uint32_t* kept_g = (uint32_t*)s.kept; long long* off_g = (long long*)s.off;
    unsigned long long* tot_dev = (unsigned long long*)s.small; unsigned long long* base_dev = tot_dev + G;
    unsigned long long* tot_host_dev = nullptr; HIP_TRY(hipHostGetDevicePointer((void**)&tot_host_dev, s.tot_host, 0));
    // ---- exchange stream: ordered behind the caller's inputs and behind the batch that used this slot two calls ago ----
    if (overlap) {
        if (!resident) { HIP_TRY(hipEventRecord(g->e_in, user)); HIP_TRY(hipStreamWaitEvent(sx, g->e_in, 0)); }   // (behind everything on the caller's stream: correct, but no overlap)
        if (g->calls >= 2) HIP_TRY(hipStreamWaitEvent(sx, s.e_done, 0));
    }
    if (g->calls >= 2) HIP_TRY(hipStreamWaitEvent(user, s.e_done, 0));   // (callers may alternate between streams: the slot's previous user may have run on another one)
    if (g->timing) HIP_TRY(hipEventRecord(g->e_t[0], sx));
    // head: every shard's view of the evolving positions, the local cuts -> global cuts
    for (size_t i = 0; i < g->shards.size(); ++i) {
        int* h_i = i == 0 ? head : head + (size_t)nq * 3;
        int rc = device_shard_lists_head(g->shards[i]->dev, p, s.pos[i], h_i, sx); if (rc) return rc;
        if (i > 0) HIP_TRY(launch_shard_max(sx, head, h_i, (size_t)nq * 3));
    }
    { int rc = all_reduce_max_i32(g, 0, head, (size_t)nq * 3, sx); if (rc) return rc; }
    // count: the entries at or above the cut, per owned list -> every rank knows every shard's counts
    for (size_t i = 0; i < g->shards.size(); ++i) {
        const uint32_t gi = local ? (uint32_t)i : (uint32_t)g->rank;
        int rc = device_shard_lists_count(g->shards[i]->dev, p, s.pos[i], head, kept_g + (size_t)gi * nq * ML, (int*)s.tot, sx); if (rc) return rc;
    }
    { int rc = all_gather_blocks(g, 0, s.kept, (size_t)nq * ML * 4, sx); if (rc) return rc; }
    HIP_TRY(launch_shard_offsets(sx, kept_g, nq, ML, G, off_g, tot_dev, tot_host_dev, base_dev + G));
    HIP_TRY(hipStreamSynchronize(sx));   // the batch's one host synchronisation (with the exchange stream: only this batch's small kernels are waited for)
    unsigned long long base[64], cnt_b[64], off_b[64], total = 0;
    if (G > 64) return fail(SRN_ERANGE, "more than 64 shards");
    for (uint32_t i = 0; i < G; ++i) { base[i] = total; total += (s.tot_host[i] + 63ull) / 64ull * 64ull; }   // (segments start on 256-byte boundaries)
    for (uint32_t i = 0; i < G; ++i) { s.tot_host[G + i] = base[i]; off_b[i] = base[i] * 4ull; cnt_b[i] = s.tot_host[i] * 4ull; }
    // A group of ONE shard ships nothing: its kept prefixes are read where they lie in the shard's posting array (no copy, no exchange buffer).  SRN_GROUP_NO_DIRECT
    // keeps the copy + exchange steps even then (tests: every RCCL call of a multi-rank run on the 1-rank communicator one GPU allows).
    const bool direct = G == 1 && !g->no_direct;
    uint32_t* lists_g = nullptr;
    if (!direct) {
        if (s.lists_bytes < total * 4 + 256) { int rc = ensure(&s.lists, &s.lists_bytes, (size_t)(total * 4 + total / 2 + 4096)); if (rc) return rc; }   // (head room: the totals of like batches differ by a per cent or two)
        HIP_TRY(hipMemcpyAsync(base_dev, s.tot_host + G, (size_t)G * 8, hipMemcpyHostToDevice, sx));
        lists_g = (uint32_t*)s.lists;
        for (size_t i = 0; i < g->shards.size(); ++i) {
            const uint32_t gi = local ? (uint32_t)i : (uint32_t)g->rank;
            int rc = device_shard_lists_copy(g->shards[i]->dev, p, s.pos[i], kept_g + (size_t)gi * nq * ML, off_g + (size_t)gi * nq, lists_g + base[gi], sx); if (rc) return rc;
        }
        { int rc = all_gather_v(g, 0, s.lists, off_b, cnt_b, sx); if (rc) return rc; }
    }
    if (g->timing) HIP_TRY(hipEventRecord(g->e_t[1], sx));
    if (overlap) { HIP_TRY(hipEventRecord(g->e_x, sx)); HIP_TRY(hipStreamWaitEvent(user, g->e_x, 0)); }
    // ---- caller's stream: the unsharded launch sequence over every local shard's row fragments ----
    for (size_t i = 0; i < g->shards.size(); ++i) {
        const uint32_t gi = local ? (uint32_t)i : (uint32_t)g->rank;
        char* blk = s.part + (size_t)gi * block_bytes;
        LaunchParams pi = p;
        const bool direct = G == 1;   // one shard: its top-n IS the result
        pi.out_ids = direct ? d_out_ids : (uint64_t*)blk; pi.out_scores = direct ? d_out_scores : (double*)(blk + (size_t)nq * n * 8); pi.out_counts = direct ? d_out_counts : (uint32_t*)(blk + (size_t)nq * n * 16);
        if (g->timing && i == 0) HIP_TRY(hipEventRecord(g->e_t[2], user));
        HIP_TRY(hipMemsetAsync(pi.out_ids, 0, (size_t)nq * n * 8, user)); HIP_TRY(hipMemsetAsync(pi.out_scores, 0, (size_t)nq * n * 8, user));
        int rc = device_shard_lists_predict(g->shards[i]->dev, g->shards[i]->flat, pi, G, kept_g, off_g, 0ull, lists_g, head, s.pos[i], s.records, user, base_dev, direct); if (rc) return rc;
        if (g->timing && i == 0) HIP_TRY(hipEventRecord(g->e_t[3], user));
    }
    if (G > 1) {
        int rc = all_gather_blocks(g, 1, s.part, block_bytes, user); if (rc) return rc;
        if (g->timing) HIP_TRY(hipEventRecord(g->e_t[4], user));
        HIP_TRY(launch_shard_merge_topn(user, s.part, block_bytes, G, nq, n, d_out_ids, d_out_scores, d_out_counts));
        if (g->timing) HIP_TRY(hipEventRecord(g->e_t[5], user));
    }
    HIP_TRY(hipEventRecord(s.e_done, user));
    if (g->timing) {   // (measurement runs only: synchronises.  [0] = head + count + copy of ALL local shards incl. the offsets kernels and the one host synchronisation)
        HIP_TRY(hipEventSynchronize(s.e_done));
        HIP_TRY(hipEventElapsedTime(&g->last_ms[0], g->e_t[0], g->e_t[1])); HIP_TRY(hipEventElapsedTime(&g->last_ms[1], g->e_t[2], g->e_t[3]));
        g->last_ms[2] = 0.f; if (G > 1) HIP_TRY(hipEventElapsedTime(&g->last_ms[2], g->e_t[4], g->e_t[5]));
    }
    ++g->calls;
    const uint32_t me = local ? 0u : (uint32_t)g->rank;
    g->st_queries += nq; g->st_bytes_head += (uint64_t)nq * 12; g->st_bytes_kept += (uint64_t)nq * ML * 4 * (local ? G : 1);
    { uint64_t mine = 0, mx = 0; for (uint32_t i = 0; i < G; ++i) { mx = std::max<uint64_t>(mx, s.tot_host[i]); if (local || i == me) mine += s.tot_host[i]; }
      if (!direct) { g->st_bytes_lists += mine * 4; g->st_lists_max += mx * 4; } }
    g->st_bytes_results += G > 1 ? (uint64_t)block_bytes * (local ? G : 1) : 0;
    return SRN_OK;
}
}  // namespace

extern "C" {

int srn_shard_group_unique_id(void* out, size_t bytes) {
    return guarded([&]() -> int {
        if (!out || bytes < SRN_SHARD_GROUP_ID_BYTES) return fail(SRN_EINVAL, "the id buffer must hold SRN_SHARD_GROUP_ID_BYTES");
        RcclApi* r = rccl();
        if (!r->h || !r->err.empty()) return fail(SRN_ENODEV, r->err.empty() ? "librccl is not available" : r->err);
        static_assert(SRN_SHARD_GROUP_ID_BYTES == 2 * sizeof(ncclUniqueId), "two communicators per group");
        ncclUniqueId id[2];
        NCCL_TRY(r->GetUniqueId(&id[0])); NCCL_TRY(r->GetUniqueId(&id[1]));
        memcpy(out, id, sizeof id);
        return SRN_OK; });
}

int srn_shard_group_create(const srn_index_t* shard, const void* unique_id, int rank, int world, srn_shard_group_t** out) {
    return guarded([&]() -> int {
        if (!out) return fail(SRN_EINVAL, "null argument");
        *out = nullptr;
        if (!unique_id || world < 1 || world > 64 || rank < 0 || rank >= world) return fail(SRN_EINVAL, "bad rank / world / id");
        int rc = check_shard(shard, (uint32_t)rank, (uint32_t)world); if (rc) return rc;
        RcclApi* r = rccl();
        if (!r->h || !r->err.empty()) return fail(SRN_ENODEV, r->err.empty() ? "librccl is not available" : r->err);
        srn_shard_group* g = new srn_shard_group();
        g->kind = srn_shard_group::RCCL; g->rank = rank; g->world = world; g->device = shard->device; g->shards = {shard};
        rc = group_init_common(g);
        if (!rc) {
            ncclUniqueId id[2]; memcpy(id, unique_id, sizeof id);
            auto init = [&]() -> int { NCCL_TRY(r->CommInitRank(&g->comm[0], world, id[0], rank)); NCCL_TRY(r->CommInitRank(&g->comm[1], world, id[1], rank)); return SRN_OK; };
            rc = init();
        }
        if (rc) { srn_shard_group_free(g); return rc; }
        *out = g; return SRN_OK; });
}

int srn_shard_group_create_with_comm(const srn_index_t* shard, int rank, int world, const srn_shard_comm_t* comm, srn_shard_group_t** out) {
    return guarded([&]() -> int {
        if (!out) return fail(SRN_EINVAL, "null argument");
        *out = nullptr;
        if (!comm || !comm->all_reduce_max_i32 || !comm->all_gather || !comm->all_gather_v || world < 1 || world > 64 || rank < 0 || rank >= world) return fail(SRN_EINVAL, "bad rank / world / callbacks");
        int rc = check_shard(shard, (uint32_t)rank, (uint32_t)world); if (rc) return rc;
        srn_shard_group* g = new srn_shard_group();
        g->kind = srn_shard_group::CALLBACKS; g->rank = rank; g->world = world; g->device = shard->device; g->shards = {shard}; g->cb = *comm;
        rc = group_init_common(g);
        if (rc) { srn_shard_group_free(g); return rc; }
        *out = g; return SRN_OK; });
}

int srn_shard_group_create_local(const srn_index_t* const* shards, int n_shards, srn_shard_group_t** out) {
    return guarded([&]() -> int {
        if (!out) return fail(SRN_EINVAL, "null argument");
        *out = nullptr;
        if (!shards || n_shards < 1 || n_shards > 64) return fail(SRN_EINVAL, "bad shard list");
        for (int i = 0; i < n_shards; ++i) {
            int rc = check_shard(shards[i], (uint32_t)i, (uint32_t)n_shards); if (rc) return rc;
            if (shards[i]->device != shards[0]->device) return fail(SRN_EINVAL, "an in-process group keeps all its shards on one device (one process per GPU otherwise: srn_shard_group_create)");
        }
        srn_shard_group* g = new srn_shard_group();
        g->kind = srn_shard_group::LOCAL; g->rank = 0; g->world = 1; g->device = shards[0]->device; g->shards.assign(shards, shards + n_shards);
        int rc = group_init_common(g);
        if (rc) { srn_shard_group_free(g); return rc; }
        *out = g; return SRN_OK; });
}

void srn_shard_group_free(srn_shard_group_t* g) {
    if (!g) return;
    hipSetDevice(g->device);
    if (!g->broken) hipDeviceSynchronize();   // (a broken group's communicators were aborted; what a dead peer left on the device is not waited for)
    for (auto& c : g->comm) if (c && rccl()->CommDestroy) rccl()->CommDestroy(c);
    for (Slot& s : g->slot) slot_free(s);
    if (g->s_x) hipStreamDestroy(g->s_x);
    if (g->e_in) hipEventDestroy(g->e_in); if (g->e_x) hipEventDestroy(g->e_x); if (g->e_last) hipEventDestroy(g->e_last);
    for (auto& e : g->e_t) if (e) hipEventDestroy(e);
    if (g->pres_all) hipFree(g->pres_all); if (g->nb_pbytes) hipFree(g->nb_pbytes); if (g->agree_dev) hipFree(g->agree_dev);
    delete g;
}

int srn_shard_group_predict_batch(srn_shard_group_t* g, const uint64_t* d_items_flat, const uint32_t* d_q_off, size_t nq, size_t max_len_hint, size_t k, size_t m,
                                  size_t how_many, unsigned flags, uint64_t* d_out_ids, double* d_out_scores, uint32_t* d_out_counts, void* stream) {
    return guarded([&]() -> int {
        if (!g) return fail(SRN_EINVAL, "null group");
        if (k == 0 || m == 0 || how_many == 0) return fail(SRN_EINVAL, "k, m and how_many must be > 0");
        if (how_many > SRN_MAX_HOW_MANY || k > SRN_MAX_K || m > 0x7FFFFFFFull) return fail(SRN_ERANGE, "k, m or how_many above the limits (srn_limits)");
        if (nq == 0) return SRN_OK;
        if (!d_items_flat || !d_q_off || !d_out_ids || !d_out_scores || !d_out_counts) return fail(SRN_EINVAL, "null buffer");
        if (nq > 0x7FFFFFFFull) return fail(SRN_ERANGE, "too many queries in one batch");
        if (max_len_hint == 0 || max_len_hint > SRN_MAX_SESSION_LEN) return fail(SRN_ERANGE, "max_len_hint out of range");
        return group_predict(g, d_items_flat, d_q_off, nq, max_len_hint, k, m, how_many, flags, d_out_ids, d_out_scores, d_out_counts, (hipStream_t)stream); });
}

int srn_debug_shard_group_times(const srn_shard_group_t* g, double* out3) {   // SRN_GROUP_TIMING=1: ms of local shard 0's prep + front end, back end (incl. general + finish kernels), merge
    if (!g || !out3 || !g->timing) return fail(SRN_EINVAL, "group timing is off (SRN_GROUP_TIMING=1 before the group is created)");
    for (int i = 0; i < 3; ++i) out3[i] = g->last_ms[i];
    return SRN_OK;
}
// Collective over the group (every rank calls it with the postings of the same index, or every rank with NULL): what is rank-LOCAL and can fail -- packed rows that did
// not fit at attach time, no room for the streaming form's second copy of the fragments, the presence bytes' buffers -- is attempted first, the outcomes are AGREED on
// (one all-reduce-min of three flags), and only then is anything committed: either every rank takes the neighbours pipeline in the same exchange format, or none does.
// (a rank that failed alone used to keep the lists pipeline while its peers issued the neighbours pipeline's collectives on the same communicator.)
int srn_shard_group_set_postings(srn_shard_group_t* g, const srn_index_t* postings) {
    if (!g) return fail(SRN_EINVAL, "null group");
    std::lock_guard<std::mutex> lk(g->mu);
    if (g->broken) return fail(SRN_ESTATE, "the shard group is unusable since an earlier batch failed (" + g->broken_why + ")");
    HIP_TRY(hipSetDevice(g->device));
    auto detach = [&]() {   // back to the lists pipeline: nothing of an earlier set_postings stays
        for (const srn_index* sh : g->shards) (void)device_sback_attach_postings(sh->dev, nullptr, 0);
        if (g->nb_pbytes) { (void)hipDeviceSynchronize(); (void)hipFree(g->nb_pbytes); g->nb_pbytes = nullptr; }
        g->postings = nullptr; g->postings_max_row_len = 0; g->stream_ok = false;
    };
    detach();
    if (!postings) return SRN_OK;
    int local_rc = SRN_OK; std::string why;
    auto bad = [&](int code, const std::string& msg) { if (local_rc == SRN_OK) { local_rc = code; why = msg; } };
    if (!postings->dev) bad(SRN_ENODEV, "the postings index has no device attached");
    else if (postings->flat.n_shards != 1) bad(SRN_EINVAL, "the replicated postings are those of the UNSHARDED index (the index itself, or srn_index_postings_view of it)");
    else if (postings->device != g->device) bad(SRN_EINVAL, "the postings must live on the group's device");
    else {
        for (const srn_index* sh : g->shards)
            if (sh->flat.n_kept != postings->flat.n_kept || sh->flat.total_pairs != postings->flat.total_pairs || sh->flat.m_index != postings->flat.m_index ||
                sh->flat.n_sessions_total != postings->flat.n_sessions_total)
                bad(SRN_EINVAL, "the postings index is not the index these shards were cut from (sessions / pairs / m_index differ)");
        if (!postings->flat.lists_complete) bad(SRN_EINVAL, "the neighbours pipeline needs complete posting lists");
        // a shard whose packed rows did not fit its GPU (device_attach lets that allocation fail) cannot take the neighbours pipeline
        for (const srn_index* sh : g->shards)
            if (!device_has_packed_rows(sh->dev)) bad(SRN_ENOMEM, "this rank's shard has no packed rows (no room at attach time): the neighbours pipeline cannot run on this group");
        if (getenv("SRN_DEBUG_FAIL_SET_POSTINGS")) bad(SRN_ENOMEM, "simulated rank-local failure (SRN_DEBUG_FAIL_SET_POSTINGS)");   // (tests: one rank of a group fails alone)
    }
    // this rank's shards keep their fragments a second time, in the posting order of these lists, where there is room (the streaming form of the back end: optional, both
    // forms give the same rows -- but the exchange FORMAT follows from it, so it is used only if EVERY rank has it)
    bool stream_here = false;
    const bool copy_wanted = knobs().sback_stream_mode == 1 || (knobs().sback_stream_mode < 0 && g->kind != srn_shard_group::LOCAL);   // (an in-process group's AUTO is the gather form: no second copy of the fragments)
    if (local_rc == SRN_OK) {
        bool all = !g->shards.empty();
        for (const srn_index* sh : g->shards) {
            const int rc = device_sback_attach_postings(sh->dev, postings->dev, copy_wanted ? postings->flat.nnz_post : 0);
            if (rc) { bad(rc, last_error_string()); break; }
            all = all && device_sback_streams(sh->dev);
        }
        stream_here = local_rc == SRN_OK && all;
    }
    // The neighbours' presence bytes (round 5, opt-in knob): every shard's presence bitmap all-gathered once, then a byte per session.  Up to 8 shards; every local shard
    // must have its bitmap (the wave-per-query back end's rows: from SRN_SBACK_MIN_SHARDS shards on).  Buffers first, the collective only if every rank has them.
    bool pbytes_here = false; size_t block_words = 0; uint8_t* pb = nullptr;
    const uint64_t n_sess = postings->flat.n_kept + 1;
    if (local_rc == SRN_OK && G_of(g) >= 2 && G_of(g) <= 8 && device_shard_nb_presence_wanted()) {
        bool have = !g->shards.empty(); size_t words = 0;
        for (const srn_index* sh : g->shards) { size_t w = 0; have = have && device_sback_present(sh->dev, &w) != nullptr; words = std::max(words, w); }
        if (have) {
            block_words = (std::max<size_t>(words, (size_t)((n_sess + 31) / 32)) + 63) / 64 * 64;
            pbytes_here = ensure(&g->pres_all, &g->pres_all_bytes, (size_t)G_of(g) * block_words * 4) == SRN_OK && hipMalloc((void**)&pb, (size_t)n_sess + 64) == hipSuccess;
            if (!pbytes_here && pb) { (void)hipFree(pb); pb = nullptr; }
        }
    }
    // ---- the agreement: min over the ranks of (this rank can take the postings | ... stream | ... has the presence buffers) ----
    int flags[3] = {local_rc == SRN_OK ? 1 : 0, stream_here ? 1 : 0, pbytes_here ? 1 : 0};
    if (g->kind != srn_shard_group::LOCAL) {
        if (!g->agree_dev) HIP_TRY(hipMalloc((void**)&g->agree_dev, 64));
        HIP_TRY(hipMemcpyAsync(g->agree_dev, flags, sizeof flags, hipMemcpyHostToDevice, g->s_x));
        { const int rc = all_reduce_min_i32(g, 0, g->agree_dev, 3, g->s_x); g->issued = false; if (rc) { if (pb) (void)hipFree(pb); detach(); return rc; } }
        HIP_TRY(hipMemcpyAsync(flags, g->agree_dev, sizeof flags, hipMemcpyDeviceToHost, g->s_x));
        HIP_TRY(hipStreamSynchronize(g->s_x));
    }
    if (!flags[0]) {
        if (pb) (void)hipFree(pb);
        detach();
        return local_rc != SRN_OK ? fail(local_rc, why) : fail(SRN_ESTATE, "a peer rank could not take the replicated postings (its error says why): no rank of the group took them -- the lists pipeline stays");
    }
    if (!flags[1] && stream_here)   // a peer has no room for the streaming form: every rank runs the gather form
        for (const srn_index* sh : g->shards) (void)device_sback_attach_postings(sh->dev, nullptr, 0);
    if (flags[2]) {
        HIP_TRY(hipMemsetAsync(g->pres_all, 0, (size_t)G_of(g) * block_words * 4, nullptr));
        for (size_t i = 0; i < g->shards.size(); ++i) {
            const uint32_t gi = g->kind == srn_shard_group::LOCAL ? (uint32_t)i : (uint32_t)g->rank;
            size_t w = 0; const uint32_t* pr = device_sback_present(g->shards[i]->dev, &w);
            HIP_TRY(hipMemcpyAsync(g->pres_all + (size_t)gi * block_words * 4, pr, w * 4, hipMemcpyDeviceToDevice, nullptr));
        }
        { const int rc = all_gather_blocks(g, 0, g->pres_all, block_words * 4, nullptr); g->issued = false; if (rc) { (void)hipFree(pb); detach(); return rc; } }
        HIP_TRY(launch_presence_bytes(nullptr, (const uint32_t*)g->pres_all, block_words, G_of(g), n_sess, pb));
        HIP_TRY(hipDeviceSynchronize());
        g->nb_pbytes = pb;
    } else if (pb) (void)hipFree(pb);
    // committed last, and the same on every rank
    g->stream_ok = flags[1] != 0; g->postings_max_row_len = postings->flat.max_row_len; g->postings = postings;
    return SRN_OK;
}

int srn_shard_group_set_overlap(srn_shard_group_t* g, int on) {
    if (!g) return fail(SRN_EINVAL, "null group");
    std::lock_guard<std::mutex> lk(g->mu);   // (between batches: a batch in flight keeps the form it was issued in; the slots' events order the next one behind it either way)
    g->overlap = on != 0;
    return SRN_OK;
}

// The host's bounded wait for the group's most recent batch: 0 when its results are complete, SRN_ETIMEOUT after timeout_ms -- a peer that died leaves this rank's
// collective waiting on the GPU for ever, and a plain stream synchronise with it.  On a timeout the group is broken (communicators aborted): free it on every rank.
int srn_shard_group_wait(srn_shard_group_t* g, uint64_t timeout_ms) {
    if (!g) return fail(SRN_EINVAL, "null group");
    return guarded([&]() -> int {
        std::lock_guard<std::mutex> lk(g->mu);
        if (g->broken) return fail(SRN_ESTATE, "the shard group is unusable since an earlier batch failed (" + g->broken_why + ")");
        if (!g->calls || !g->e_last) return SRN_OK;
        HIP_TRY(hipSetDevice(g->device));
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t spin = 0;; ++spin) {
            const hipError_t e = hipEventQuery(g->e_last);
            if (e == hipSuccess) return SRN_OK;
            if (e != hipErrorNotReady) { group_break(g, std::string("hipEventQuery: ") + hipGetErrorString(e)); return fail(SRN_EHIP, std::string("hipEventQuery: ") + hipGetErrorString(e)); }
            if ((uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() >= timeout_ms) {
                group_break(g, "a batch did not finish within " + std::to_string(timeout_ms) + " ms (a peer gone?)");
                return fail(SRN_ETIMEOUT, "the group's last batch did not finish within " + std::to_string(timeout_ms) + " ms: the group is now unusable on this rank (free it on every rank)");
            }
            if (spin > 64) std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
    });
}

int srn_shard_group_stats(const srn_shard_group_t* g, srn_shard_group_stats_t* out) {
    if (!g || !out) return fail(SRN_EINVAL, "null argument");
    *out = srn_shard_group_stats_t{(uint64_t)G_of(g), g->calls, g->st_queries, g->st_bytes_head, g->st_bytes_kept, g->st_bytes_lists, g->st_bytes_results, g->st_lists_max,
                                   g->kind == srn_shard_group::RCCL ? 1u : g->kind == srn_shard_group::CALLBACKS ? 2u : 0u, g->overlap && g->kind != srn_shard_group::LOCAL ? 1u : 0u,
                                   g->st_stage_batches, g->st_bytes_stage_cand, g->st_bytes_stage_minpos, g->st_nb_batches, g->st_bytes_nb};
    return SRN_OK;
}

}  // extern "C"

// =====================================================================================
// Device runtime of libserenade_hip.so: index upload (row slots laid out on the device), per-call workspaces (own stream,
// staging buffers, event ring), LDS geometry of a launch, the predict launches (prep kernel, predict kernel, global-table
// retry pass), the item-sharded pipeline's stages, timing and debug counters.  The kernels live in srn_kernels.hip.
// =====================================================================================
#include <hip/hip_runtime.h>

#include <chrono>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "srn_runtime.h"
#include "srn_hipsync.h"

namespace srn {

static Knobs g_knobs; static std::once_flag g_knobs_once; static std::mutex g_knobs_mu;
static void knobs_read() {
    Knobs k;
    if (const char* e = getenv("SRN_FAST_HOW_MANY_MAX")) k.fast_how_many_max = std::max(1, std::min(64, atoi(e)));
    k.no_masks = getenv("SRN_NO_MASKS") != nullptr; k.no_viol = getenv("SRN_NO_VIOL") != nullptr; k.no_merge = getenv("SRN_NO_MERGE") != nullptr; k.dense = getenv("SRN_DENSE") != nullptr;
    k.no_fast = getenv("SRN_NO_FAST") != nullptr; k.no_mid = getenv("SRN_NO_MID") != nullptr; k.no_big = getenv("SRN_NO_BIG") != nullptr; k.no_long = getenv("SRN_NO_LONG") != nullptr; k.debug = getenv("SRN_DEBUG") != nullptr;
    if (const char* e = getenv("SRN_ROW_SLOTS")) k.row_slots16 = atoi(e) == 16 ? 1 : atoi(e) == 64 ? 0 : -1;
    if (const char* e = getenv("SRN_HOT_SLOTS")) k.hot_slots = std::max(0, atoi(e));
    if (const char* e = getenv("SRN_SKETCH_SLOTS")) k.sketch_slots = std::max(0, atoi(e));
    if (const char* e = getenv("SRN_LDS_BUDGET_KB")) k.lds_budget_kb = std::max(0, atoi(e));
    if (const char* e = getenv("SRN_GRID_MULT")) { k.grid_mult = std::max(1, atoi(e)); k.grid_mult_set = true; }
    if (const char* e = getenv("SRN_HOST_CHUNKS")) k.host_chunks = std::max(0, atoi(e));
    if (const char* e = getenv("SRN_HOST_FIRST_PCT")) k.host_first_pct = std::min(99, std::max(0, atoi(e)));
    if (const char* e = getenv("SRN_COPY_SLICES")) k.copy_slices = std::max(0, atoi(e));
    k.host_nocopy = getenv("SRN_HOST_NOCOPY") != nullptr; k.host_trace = getenv("SRN_HOST_TRACE") != nullptr; k.timing = getenv("SRN_TIMING") != nullptr && atoi(getenv("SRN_TIMING")) != 0;
    if (const char* e = getenv("SRN_D2H_BLOCKS")) k.d2h_blocks = std::max(0, atoi(e));
    if (const char* e = getenv("SRN_TINY_MAX")) k.tiny_max = std::max(0, atoi(e));
    k.no_tiny_fused = getenv("SRN_TINY_FUSED") != nullptr && atoi(getenv("SRN_TINY_FUSED")) == 0;
    if (const char* e = getenv("SRN_TINY_FUSED_MAX")) k.tiny_fused_max = std::min(256, std::max(1, atoi(e)));
    k.no_tiny_spin = getenv("SRN_TINY_SPIN") != nullptr && atoi(getenv("SRN_TINY_SPIN")) == 0;
    if (const char* e = getenv("SRN_TINY_PHASES")) k.tiny_phases = std::min(2, std::max(0, atoi(e)));
    if (const char* e = getenv("SRN_TINY_FAST")) k.tiny_fast = std::min(3, std::max(0, atoi(e)));
    if (const char* e = getenv("SRN_PREDICT_LANES")) k.lanes = std::max(0, atoi(e));
    if (const char* e = getenv("SRN_ORDER_MIN")) k.order_min = std::max(0, atoi(e));
    k.no_sback_second = getenv("SRN_NO_SBACK_SECOND") != nullptr;
    k.no_sback_pbytes = !(getenv("SRN_SBACK_PBYTES") != nullptr && atoi(getenv("SRN_SBACK_PBYTES")) != 0);
    k.no_sback_finish = !(getenv("SRN_SBACK_FINISH") != nullptr && atoi(getenv("SRN_SBACK_FINISH")) != 0);
    k.no_sback = getenv("SRN_NO_SBACK") != nullptr; k.sback_bitmap = getenv("SRN_SBACK_BITMAP") != nullptr && atoi(getenv("SRN_SBACK_BITMAP")) != 0;
    k.sback_stream_mode = getenv("SRN_SBACK_STREAM") == nullptr ? -1 : atoi(getenv("SRN_SBACK_STREAM")) != 0 ? 1 : 0;
    k.no_sback_stream = k.sback_stream_mode == 0;
    if (const char* e = getenv("SRN_XGMI_GBPS")) { const double v = atof(e); if (v > 0.0) k.xgmi_gbps = v; }
    if (const char* e = getenv("SRN_SBACK_MIN_SHARDS")) k.sback_min_shards = std::max(2, atoi(e));
    if (const char* e = getenv("SRN_FAST_RUNS")) k.fast_runs = atoi(e) == 3 ? 3 : 0;   // tests: the fast kernel's 29-bit-rank form (3 lists per query) on a small index
    std::lock_guard<std::mutex> lk(g_knobs_mu); g_knobs = k;
}
Knobs knobs() { std::call_once(g_knobs_once, knobs_read); std::lock_guard<std::mutex> lk(g_knobs_mu); return g_knobs; }
int knob_predict_lanes() { return knobs().lanes; }
size_t knob_tiny_max() { return (size_t)std::max(1, knobs().tiny_max); }
void reload_knobs() { std::call_once(g_knobs_once, knobs_read); knobs_read(); }

namespace {
template <typename T> const T* upload(DeviceState* d, const std::vector<T>& v, bool& ok, size_t pad = 0, size_t = 0) {   // (hipMalloc is 256-byte aligned)
    void* p = nullptr; const size_t n = std::max<size_t>(v.size() * sizeof(T), 16) + pad;
    if (!ok) return nullptr;
    if (hipMalloc(&p, n) != hipSuccess) { ok = false; set_error("hipMalloc failed for index array"); return nullptr; }
    d->allocs.push_back(p); d->bytes += n;
    if (!v.empty() && hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) { ok = false; set_error("hipMemcpy H2D failed"); }
    return (const T*)p;
}
}  // namespace

DeviceState* device_attach(const FlatIndex& ix, int device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device >= ndev) { set_error("no such HIP device"); return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { set_error("hipSetDevice failed"); return nullptr; }
    DeviceState* d = new DeviceState(); d->device = device; d->timing.store(knobs().timing);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) { d->n_cu = prop.multiProcessorCount; d->lds_per_block_max = (int)prop.sharedMemPerBlock; }
    bool ok = true;
    d->di.id_table = upload(d, ix.id_table, ok); d->di.id_mask = ix.id_mask;
    { std::vector<ItemMeta> meta(ix.n_items); std::vector<uint64_t> id_sorted(ix.n_items);
      for (size_t i = 0; i < ix.n_items; ++i) { meta[i] = ItemMeta{ix.idf[i], ix.id_rank[i], ix.attr[i]}; id_sorted[ix.id_rank[i]] = ix.item_id[i]; }
      d->d_meta = (ItemMeta*)upload(d, meta, ok); d->di.meta = d->d_meta; d->di.id_sorted = upload(d, id_sorted, ok);
      std::vector<ItemMeta> ms(512, ItemMeta{0.0, 0u, 0u});   // the fast kernel's threshold sample: one coalesced read per wave
      for (uint32_t w8 = 0; w8 < 8; ++w8) for (uint32_t l = 0; l < 64; ++l) if (8 * l + w8 < ix.n_items) ms[64 * w8 + l] = meta[8 * l + w8];
      d->fast.meta_sample = upload(d, ms, ok); }
    d->di.post_off = upload(d, ix.post_off, ok); d->di.post_rank = upload(d, ix.post_rank, ok);
    d->di.viol = !ix.lists_complete && ix.viol.size() == ix.n_items && ix.n_shards == 1 ? upload(d, ix.viol, ok) : nullptr;   // (a pre-built index whose lists are not all complete: the prep kernel's per-query test)
    if (ok && !ix.postings_only) {   // rows -> 64-byte slots (+ overflow area) on the device, see DeviceIndex and rows_to_slots_kernel
        // item shards: 16-byte FRAGMENT slots (a shard holds ~1/n_shards of a row's items), DeviceIndex::row_frag
        const int rs16 = knobs().row_slots16;
        const bool frag = rs16 >= 0 ? rs16 == 1 : ix.n_shards > 1;   // (SRN_ROW_SLOTS=16|64 forces a form: experiments)
        const uint64_t inl = frag ? 2 : 14;            // items inline in a slot that also carries an overflow offset
        const size_t slot_bytes = frag ? 16 : 64;
        const size_t n = ix.n_kept, nblocks = (n + 1 + 1023) / 1024;
        std::vector<uint32_t> block_base(nblocks);
        uint64_t ext_total = 16;   // ext[0..15] = EMPTY32: what short rows read
        for (size_t b0 = 0; b0 < nblocks; ++b0) {
            block_base[b0] = (uint32_t)ext_total;
            const size_t hi = std::min(n, (b0 + 1) * 1024);
            for (size_t r = b0 * 1024; r < hi; ++r) { const uint64_t len = ix.row_off[r + 1] - ix.row_off[r]; if (len > inl + 1) ext_total += len - inl; }
            if (ext_total >= 0xFFFFFFF0ull) { set_error("row overflow area exceeds 2^32 items"); device_release(d); return nullptr; }
        }
        void *d_off = nullptr, *d_items = nullptr, *d_base = nullptr, *d_slots = nullptr, *d_ext = nullptr;
        const size_t ext_words = ext_total + 16;   // (4-item loads may run past the last row)
        bool good = hipMalloc(&d_off, (n + 1) * 8) == hipSuccess && hipMalloc(&d_items, std::max<size_t>(ix.row_items.size() * 4, 16)) == hipSuccess &&
                    hipMalloc(&d_base, nblocks * 4) == hipSuccess && hipMalloc(&d_slots, (n + 1) * slot_bytes) == hipSuccess && hipMalloc(&d_ext, ext_words * 4) == hipSuccess;
        good = good && hipMemcpy(d_off, ix.row_off.data(), (n + 1) * 8, hipMemcpyHostToDevice) == hipSuccess &&
               (ix.row_items.empty() || hipMemcpy(d_items, ix.row_items.data(), ix.row_items.size() * 4, hipMemcpyHostToDevice) == hipSuccess) &&
               hipMemcpy(d_base, block_base.data(), nblocks * 4, hipMemcpyHostToDevice) == hipSuccess &&
               hipMemset(d_ext, 0xFF, ext_words * 4) == hipSuccess;
        good = good && (frag ? launch_rows_to_frags : launch_rows_to_slots)(nullptr, (const uint64_t*)d_off, (const uint32_t*)d_items, (uint64_t)n, (const uint32_t*)d_base,
                                                                             (uint32_t*)d_slots, (uint32_t*)d_ext) == hipSuccess &&
               hipDeviceSynchronize() == hipSuccess;
        if (good) {   // the fast kernel's rows: 64-byte slots of 16-bit LDS offsets (+ overflow blocks of 8 items); item shards: 16-byte fragment slots, srn_fast.hip
            std::vector<uint32_t> bb(nblocks); uint64_t blocks = 1;   // (block 0: what a stray read finds)
            for (size_t b0 = 0; b0 < nblocks; ++b0) {
                bb[b0] = (uint32_t)blocks;
                const size_t hi = std::min(n, (b0 + 1) * 1024);
                for (size_t r = b0 * 1024; r < hi; ++r) { const uint64_t len = ix.row_off[r + 1] - ix.row_off[r];
                    if (frag) { if (len > 6) blocks += (len - 4 + 7) / 8; } else if (len > 30) blocks += (len - 28 + 7) / 8; }
            }
            void *d_pk = nullptr, *d_e16 = nullptr;
            bool g2 = blocks < 0xFFFFFFF0ull && hipMalloc(&d_pk, (n + 1) * slot_bytes) == hipSuccess && hipMalloc(&d_e16, (blocks + 2) * 16) == hipSuccess &&
                      hipMemcpy(d_base, bb.data(), nblocks * 4, hipMemcpyHostToDevice) == hipSuccess && hipMemset(d_e16, 0, (blocks + 2) * 16) == hipSuccess &&
                      launch_rows_to_packed(nullptr, (const uint64_t*)d_off, (const uint32_t*)d_items, (uint64_t)n, (const uint32_t*)d_base, (uint32_t*)d_pk, (uint32_t*)d_e16, frag) == hipSuccess &&
                      hipDeviceSynchronize() == hipSuccess;
            // The packed rows are OPTIONAL (row_packed == nullptr: every query takes the general kernel): they cost another (n + 1) * slot_bytes of HBM
            // next to row_slots, and an index that fits without them must still attach when they do not.
            if (g2) { d->allocs.push_back(d_pk); d->bytes += (n + 1) * slot_bytes; d->allocs.push_back(d_e16); d->bytes += (blocks + 2) * 16;
                      d->fast.row_packed = (const RowQuad*)d_pk; d->fast.row_ext16 = (const uint32_t*)d_e16; }
            else {
                (void)hipGetLastError();   // (clear the sticky allocation error)
                if (d_pk) hipFree(d_pk); if (d_e16) hipFree(d_e16);
                d->fast.row_packed = nullptr; d->fast.row_ext16 = nullptr;
                if (knobs().debug) fprintf(stderr, "[srn] no room for the fast kernel's packed rows (%zu bytes): general kernel only\n", (size_t)((n + 1) * slot_bytes + (blocks + 2) * 16));
            }
        }
        if (good && frag && ix.n_shards >= (uint32_t)knobs().sback_min_shards) {   // the wave-per-query back end's rows (srn_sback.hip): 8-byte slots + overflow blocks + presence bitmap; optional like the packed rows
            std::vector<uint32_t> bb(nblocks); uint64_t blocks = 1, longest = 0;
            for (size_t b0 = 0; b0 < nblocks; ++b0) {
                bb[b0] = (uint32_t)blocks;
                const size_t hi = std::min(n, (b0 + 1) * 1024);
                for (size_t r = b0 * 1024; r < hi; ++r) { const uint64_t len = ix.row_off[r + 1] - ix.row_off[r]; longest = std::max(longest, len); if (len > 4) blocks += (len + 7) / 8; }
            }
            const size_t pwords = (n + 1 + 31) / 32 + 32;
            void *d_f8 = nullptr, *d_e8 = nullptr, *d_pr = nullptr, *d_sm = nullptr;
            std::vector<ItemMeta> sm(256, ItemMeta{0.0, 0u, 0u});
            for (size_t i = 0; i < std::min<size_t>(256, ix.n_items); ++i) sm[i] = ItemMeta{ix.idf[i], ix.id_rank[i], ix.attr[i]};
            const bool g3 = blocks < 0xFFFFFFF0ull && longest < (1ull << 11) &&   // (what a long fragment's slot can say: srn_sback.hip, SB_LONG)
                            hipMalloc(&d_f8, (n + 1) * 8) == hipSuccess && hipMalloc(&d_e8, (blocks + 2) * 16) == hipSuccess && hipMalloc(&d_pr, pwords * 4) == hipSuccess &&
                            hipMalloc(&d_sm, 256 * sizeof(ItemMeta)) == hipSuccess && hipMemcpy(d_sm, sm.data(), 256 * sizeof(ItemMeta), hipMemcpyHostToDevice) == hipSuccess &&
                            hipMemcpy(d_base, bb.data(), nblocks * 4, hipMemcpyHostToDevice) == hipSuccess && hipMemset(d_e8, 0, (blocks + 2) * 16) == hipSuccess && hipMemset(d_pr, 0, pwords * 4) == hipSuccess &&
                            launch_rows_to_frag8(nullptr, (const uint64_t*)d_off, (const uint32_t*)d_items, (uint64_t)n, (const uint32_t*)d_base, (uint2*)d_f8, (uint4*)d_e8, (uint32_t*)d_pr) == hipSuccess &&
                            hipDeviceSynchronize() == hipSuccess;
            if (g3) { for (void* q : {d_f8, d_e8, d_pr, d_sm}) d->allocs.push_back(q);
                      d->bytes += (n + 1) * 8 + (blocks + 2) * 16 + pwords * 4 + 256 * sizeof(ItemMeta);
                      d->sback.frag8 = (const uint2*)d_f8; d->sback.ext8 = (const uint4*)d_e8; d->sback.present = (const uint32_t*)d_pr; d->sback.sample = (const ItemMeta*)d_sm; d->sback_present_words = pwords; }
            else { (void)hipGetLastError(); for (void* q : {d_f8, d_e8, d_pr, d_sm}) if (q) hipFree(q); d->sback = SBackParams{}; }
        }
        if (d_off) hipFree(d_off); if (d_items) hipFree(d_items); if (d_base) hipFree(d_base);
        if (d_slots) { d->allocs.push_back(d_slots); d->bytes += (n + 1) * slot_bytes; }
        d->di.row_frag = frag ? 1u : 0u;
        if (d_ext) { d->allocs.push_back(d_ext); d->bytes += ext_words * 4; }
        if (!good) { ok = false; set_error("row slot layout on the device failed"); }
        d->di.row_slots = (const RowQuad*)d_slots; d->di.row_ext = (const uint32_t*)d_ext;
    }
    d->di.n_items = (uint32_t)ix.n_items; d->di.n_kept = (uint32_t)ix.n_kept;
    double hi = 1.0, lo = 1.0; bool any = false;   // bounds of idf_eff = (idf > 0 ? idf : 1) for the top-n pre-filter
    for (double v : ix.idf) { const double e = v > 0.0 ? v : 1.0; if (!any) { hi = lo = e; any = true; } else { hi = std::max(hi, e); lo = std::min(lo, e); } }
    d->di.idf_hi = hi; d->di.idf_lo = lo;
    device_refresh_fast_bounds(d, ix);
    if (knobs().debug) fprintf(stderr, "[srn] idf_hi=%g idf_lo=%g n_items=%zu\n", hi, lo, (size_t)ix.n_items);
    if (!ok) { device_release(d); return nullptr; }
    return d;
}

// idf bounds of the fast kernel's integer floors: the largest idf_eff of each chunk of 512 direct-mapped items, and of all items
void device_refresh_fast_bounds(DeviceState* d, const FlatIndex& ix) {
    double hi_all = 1.0; bool any = false;
    for (double v : ix.idf) { const double e = v > 0.0 ? v : 1.0; hi_all = any ? std::max(hi_all, e) : e; any = true; }
    for (uint32_t c = 0; c < 8; ++c) {
        double hi = 0.0;
        for (uint64_t i = (uint64_t)c * 512; i < std::min<uint64_t>(ix.n_items, (uint64_t)c * 512 + 512); ++i) hi = std::max(hi, ix.idf[i] > 0.0 ? ix.idf[i] : 1.0);
        d->fast.inv_idf_hot[c] = hi > 0.0 ? 1.0 / hi : 1.0;
    }
    d->fast.inv_idf_hi = 1.0 / hi_all;
    for (uint32_t c = 0; c < SB_H / 256; ++c) {   // the wave-per-query back end's floors: chunks of 256
        double hi = 0.0;
        for (uint64_t i = (uint64_t)c * 256; i < std::min<uint64_t>(ix.n_items, (uint64_t)c * 256 + 256); ++i) hi = std::max(hi, ix.idf[i] > 0.0 ? ix.idf[i] : 1.0);
        d->sback.inv_idf_chunk[c] = hi > 0.0 ? 1.0 / hi : 1.0;
    }
    d->sback.inv_idf_all = 1.0 / hi_all;
}

static void ws_free(Workspace* w) {
    if (!w) return;
    if (w->retry_list) hipFree(w->retry_list);
    if (w->retry_cnt) hipFree(w->retry_cnt);
    if (w->gscratch) hipFree(w->gscratch);
    if (w->spill) hipFree(w->spill);
    if (w->prep) hipFree(w->prep);
    if (w->order) hipFree(w->order);
    if (w->order2) hipFree(w->order2);
    if (w->sb_scr) hipFree(w->sb_scr);
    if (w->retry_list2) hipFree(w->retry_list2);
    if (w->retry_cnt2) hipFree(w->retry_cnt2);
    if (w->slow_list) hipFree(w->slow_list);
    if (w->slow_cnt) hipFree(w->slow_cnt);
    if (w->fin) hipFree(w->fin);
    if (w->big) hipFree(w->big);
    if (w->pin) hipHostFree(w->pin);
    if (w->stage) hipFree(w->stage);
    if (w->h_retry) hipHostFree(w->h_retry);
    if (w->ev_block) hipEventDestroy(w->ev_block);
    if (w->prep2) hipFree(w->prep2);
    for (int i = 0; i < 2; ++i) { if (w->ev_prep[i]) hipEventDestroy(w->ev_prep[i]); if (w->ev_done[i]) hipEventDestroy(w->ev_done[i]); }
    if (w->ev_fork) hipEventDestroy(w->ev_fork); if (w->ev_join) hipEventDestroy(w->ev_join);
    if (w->side) hipStreamDestroy(w->side);
    for (auto& t : w->ev) for (auto& e : t) if (e) hipEventDestroy(e);
    if (w->stream) hipStreamDestroy(w->stream);
    delete w;
}

void serve_free_retired(DeviceState* d);
void device_release(DeviceState* d) {
    if (!d) return;
    (void)device_serve_stop(d); serve_free_retired(d);
    hipSetDevice(d->device);
    hostpipes_free(d);
    for (Workspace* w : d->all_ws) ws_free(w);
    for (void* p : d->allocs) hipFree(p);
    if (d->sb_frag_post) hipFree(d->sb_frag_post);
    delete d;
}

int device_update_attr(DeviceState* d, const FlatIndex& ix) {
    HIP_TRY(hipSetDevice(d->device));
    std::vector<ItemMeta> meta(ix.n_items);
    for (size_t i = 0; i < ix.n_items; ++i) meta[i] = ItemMeta{ix.idf[i], ix.id_rank[i], ix.attr[i]};
    HIP_TRY(hipMemcpy(d->d_meta, meta.data(), meta.size() * sizeof(ItemMeta), hipMemcpyHostToDevice));
    if (d->fast.meta_sample) {   // the fast kernel's copy of the 512 most popular items' records (same order as at attach)
        std::vector<ItemMeta> ms(512, ItemMeta{0.0, 0u, 0u});
        for (uint32_t w8 = 0; w8 < 8; ++w8) for (uint32_t l = 0; l < 64; ++l) if (8 * l + w8 < ix.n_items) ms[64 * w8 + l] = meta[8 * l + w8];
        HIP_TRY(hipMemcpy((void*)d->fast.meta_sample, ms.data(), ms.size() * sizeof(ItemMeta), hipMemcpyHostToDevice));
    }
    if (d->sback.sample) {
        std::vector<ItemMeta> sm(256, ItemMeta{0.0, 0u, 0u});
        for (size_t i = 0; i < std::min<size_t>(256, ix.n_items); ++i) sm[i] = meta[i];
        HIP_TRY(hipMemcpy((void*)d->sback.sample, sm.data(), sm.size() * sizeof(ItemMeta), hipMemcpyHostToDevice));
    }
    return SRN_OK;
}
uint64_t device_bytes(const DeviceState* d) { return d ? d->bytes : 0; }
bool device_has_packed_rows(const DeviceState* d) { return d && d->fast.row_packed != nullptr; }
bool device_sback_streams(const DeviceState* d) { return d && d->sb_frag_post != nullptr; }
bool device_sback_wanted(const DeviceState* d) { return d && d->sback.frag8 != nullptr && !knobs().no_sback_stream && !knobs().no_sback; }
int device_sback_attach_postings(DeviceState* d, DeviceState* post, uint64_t n_postings) {
    if (!d) return SRN_OK;
    HIP_TRY(hipSetDevice(d->device));
    if (d->sb_frag_post && (!post || d->sb_post_for != post->di.post_rank)) {   // (another index's lists, or none: the old copy goes -- after whatever still reads it)
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipFree(d->sb_frag_post)); d->bytes -= d->sb_frag_post_bytes; d->sb_frag_post = nullptr; d->sb_post_for = nullptr; d->sb_frag_post_bytes = 0;
    }
    if (!post || !d->sback.frag8 || d->sb_frag_post || n_postings == 0 || knobs().no_sback_stream || knobs().no_sback) return SRN_OK;
    const uint64_t need = n_postings * 8;
    size_t free_b = 0, total_b = 0; HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    if (free_b < need + std::max<uint64_t>(total_b / 16, 4ull << 30)) {   // (optional: a sixteenth of the device stays free for the batches' workspaces)
        if (knobs().debug) fprintf(stderr, "[srn] no room for the fragments in posting order (%llu bytes, %zu free): the back end gathers\n", (unsigned long long)need, free_b);
        return SRN_OK;
    }
    void* fp = nullptr;
    if (hipMalloc(&fp, need) != hipSuccess) { (void)hipGetLastError(); return SRN_OK; }
    if (launch_frag_post(nullptr, post->di.post_rank, d->sback.frag8, (uint2*)fp, n_postings) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        const hipError_t e = hipGetLastError(); hipFree(fp); return fail(SRN_EHIP, std::string("fragments in posting order: ") + hipGetErrorString(e));
    }
    d->sb_frag_post = fp; d->sb_post_for = post->di.post_rank; d->sb_frag_post_bytes = need; d->bytes += need;
    return SRN_OK;
}
uint64_t device_sback_launches(const DeviceState* d) { return d ? d->sback_launches.load() : 0; }
int device_count() { int n = 0; return hipGetDeviceCount(&n) == hipSuccess ? n : 0; }

// Host-pointer calls borrow a workspace for the duration of the (synchronous) call.  Device-pointer
// calls return while their work is still in flight, so their scratch stays bound to the user's stream
// (stream order then serialises its reuse).
// A workspace from the pool (host-pointer calls: exclusively the caller's until released) or THE workspace bound to `user_stream` (device-pointer calls).  The latter is
// locked for the duration of the call: its buffers are reused in stream order, which holds only while every call's launches are contiguous in the stream -- two host
// threads on one stream (found by tools/fuzz_concurrency.py with 64 threads on torch's pool of 32 streams: wrong rows, then a GPU memory fault) now take turns.
Workspace* ws_acquire(DeviceState* d, bool bind_to_stream, void* user_stream) {
    std::unique_lock<std::mutex> lk(d->mu);
    if (bind_to_stream) for (auto& sw : d->stream_ws) if (sw.first == user_stream) { Workspace* w = sw.second; lk.unlock(); w->call_mu.lock(); return w; }
    if (!bind_to_stream && !d->free_ws.empty()) { Workspace* w = d->free_ws.back(); d->free_ws.pop_back(); return w; }
    Workspace* w = new Workspace();
    if (hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking) != hipSuccess) { delete w; return nullptr; }
    for (auto& t : w->ev) for (auto& e : t) if (hipEventCreate(&e) != hipSuccess) { ws_free(w); return nullptr; }
    if (hipMalloc((void**)&w->retry_cnt, 16) != hipSuccess || hipMalloc((void**)&w->retry_cnt2, 16) != hipSuccess || hipMalloc((void**)&w->slow_cnt, 32) != hipSuccess || hipHostMalloc((void**)&w->h_retry, 32, hipHostMallocCoherent) != hipSuccess)   /* (coherent whatever HIP_HOST_COHERENT says: the latency path's caller reads these words while the kernel that writes them is still running) */ { ws_free(w); return nullptr; }
    memset(w->h_retry, 0, 32);
    if (hipHostGetDevicePointer((void**)&w->h_retry_dev, w->h_retry, 0) != hipSuccess) { ws_free(w); return nullptr; }
    d->all_ws.push_back(w);
    if (bind_to_stream) { d->stream_ws.emplace_back(user_stream, w); w->call_mu.lock(); }   // (nobody else can have found it yet: d->mu is held)
    return w;
}
void ws_release(DeviceState* d, Workspace* w, bool bound) {
    { std::lock_guard<std::mutex> lk(d->mu);
      if (!bound) d->free_ws.push_back(w);
      d->last_ws = w; }
    if (bound) w->call_mu.unlock();
}

int ensure(char** p, size_t* have, size_t need) {
    if (*have >= need) return SRN_OK;
    if (*p) HIP_TRY(hipFree(*p));
    *p = nullptr; *have = 0;
    need = need + need / 4 + 256;
    HIP_TRY(hipMalloc((void**)p, need));
    *have = need; return SRN_OK;
}

static inline uint32_t round_up(uint32_t v, uint32_t a) { return (v + a - 1) / a * a; }
static inline int bits_host(uint64_t v) { int b = 0; while (v) { ++b; v >>= 1; } return b; }


static inline uint32_t floor_pow2(uint64_t v) { uint32_t p2 = 1; while ((uint64_t)p2 * 2 <= v) p2 <<= 1; return p2; }
static inline uint64_t ceil_pow2(uint64_t v) { uint64_t p2 = 1; while (p2 < v) p2 <<= 1; return p2; }
static inline bool is_prime(uint32_t n) { if (n < 2) return false; for (uint32_t d = 2; (uint64_t)d * d <= n; ++d) if (n % d == 0) return false; return true; }
static inline uint32_t prime_at_most(uint32_t n) { while (n > 2 && !is_prime(n)) --n; return std::max<uint32_t>(n, 2); }
static inline uint32_t prime_at_least(uint64_t n) { uint32_t v = (uint32_t)std::min<uint64_t>(n, 0x3FFFFFFFull); while (!is_prime(v)) ++v; return v; }

// a sketch word sums max(w, 0) * num over ALL elements that share it (<= k rows of <= max_row_len items): can that pass 2^32?
static inline bool sketch_wraps(uint64_t k, uint64_t max_row_len, uint64_t Lmax) { return k * std::max<uint64_t>(1, max_row_len) * 9ull * (Lmax * (Lmax + 1) / 2) >= (1ull << 32); }
struct Geometry {
    KernelCfg c{}; bool slot64 = false, masks = false; uint32_t slot_bytes = 4; size_t lds = 0;
    uint64_t need_sess = 0, need_item = 0; bool sess_may_overflow = false, item_may_overflow = false, sketch_may_wrap = false;
};
// LDS layout + table sizes for one launch (all blocks alike).  min_region_b: extra room the caller needs in region B.
static int make_geometry(const DeviceState* d, const FlatIndex& ix, const LaunchParams& p, uint32_t min_region_b, Geometry& g, uint32_t budget_bytes = 0) {
    const Knobs kn = knobs();
    const uint64_t Lmax = p.max_len;
    // position-set slots (no first-match pass over the rows) need <= 8 evolving items and lists complete above x_lo
    g.masks = Lmax <= 8 && p.m <= ix.m_index && ix.lists_complete && !kn.no_masks;
    const int num_bits = g.masks ? (int)Lmax : std::max(1, bits_host(Lmax * (Lmax + 1) / 2));
    const int rank_bits = std::max(1, bits_host(ix.n_kept ? ix.n_kept - 1 : 0));
    g.slot64 = rank_bits + num_bits > 32 || ix.n_kept >= 0xFFFFFFF0ull;
    g.slot_bytes = g.slot64 ? 8 : 4;
    const uint32_t slot_bytes = g.slot_bytes;
    // what the query set could need at most
    const uint64_t m_eff = std::min<uint64_t>(p.m, ix.m_index);
    g.need_sess = std::max<uint64_t>(1, std::min<uint64_t>(Lmax * m_eff, ix.n_kept));
    g.need_item = std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)p.k * std::max<uint64_t>(1, ix.max_row_len), ix.n_items));
    KernelCfg& c = g.c;
    c.num_bits = (uint32_t)num_bits;
    c.q_cap = round_up((uint32_t)Lmax + 1, 4);
    c.off_q = MISC_WORDS * 4 + 1024;
    c.off_wave = round_up(c.off_q + c.q_cap * 24 + (c.q_cap + 4) * 4, 16);
    c.off_b = c.off_wave;   // (no per-wave scratch any more)
    const uint32_t region_b = round_up(std::max<uint32_t>(std::max<uint32_t>(p.k * slot_bytes, CAND_CAP * 12), min_region_b), 16);
    c.off_a = c.off_b + region_b;
    const uint32_t lds_max = (uint32_t)std::min(d->lds_per_block_max, 160 * 1024);
    uint32_t budget = budget_bytes ? budget_bytes : 80 * 1024;   // default: two 512-thread blocks per CU
    if (kn.lds_budget_kb) budget = (uint32_t)kn.lds_budget_kb * 1024;   // experiment knob (occupancy studies)
    if (c.off_a + 32 * 1024 > budget) budget = lds_max;   // long sessions / large k: one block per CU
    if (c.off_a + 8 * 1024 > budget) return fail(SRN_ERANGE, "k / session length too large for the LDS layout");
    const uint32_t a_max = budget - c.off_a;
    c.sess_slots = std::min<uint32_t>(floor_pow2(a_max / slot_bytes), (uint32_t)std::min<uint64_t>(1u << 30, ceil_pow2(g.need_sess * 2)));
    c.sess_slots = std::max<uint32_t>(c.sess_slots, 256);
    // item side of region A: direct-mapped accumulators for the most popular idx, the rest a hash of 4-slot buckets.
    // A direct-mapped word packs (touch count, signed weight sum); if those do not fit 32 bits the hot part is disabled.
    // |10 * linear_score| is 9 at the first position but reaches 89 at position 99 (negative weights beyond position 10, Q3): without
    // position sets a session of >= 20 items can see weights up to min(Lmax, 99) - 10
    const uint64_t w_abs = g.masks ? 9 : std::max<uint64_t>(9, std::min<uint64_t>(Lmax, 99) > 10 ? std::min<uint64_t>(Lmax, 99) - 10 : 0);
    const uint64_t w_max = (uint64_t)p.k * w_abs * (Lmax * (Lmax + 1) / 2) + 1;
    // The exact accumulators (item table, direct-mapped words) are 32-bit integers: |acc(item)| <= k * max over first-match positions q of
    // |10 - q| * num_max(q), where a neighbour whose first match is at position q holds no more recent evolving item: num_max(q) = sum of (L - pos) over
    // pos >= q - 1 = (L - q + 1)(L - q + 2) / 2.  Beyond 2^31 the sums would wrap silently: refuse the call instead (k = 8192 with sessions of ~90+ items).
    {
        uint64_t wn_max = 0;
        for (uint64_t q = 1; q <= std::min<uint64_t>(Lmax, 99); ++q) {
            const uint64_t wq = q <= 10 ? 10 - q : q - 10, nq_ = (Lmax - q + 1) * (Lmax - q + 2) / 2;
            wn_max = std::max(wn_max, wq * nq_);
        }
        if ((uint64_t)p.k * wn_max >= (1ull << 31))
            return fail(SRN_ERANGE, "k * |10 * linear_score| * similarity numerator can exceed the 32-bit item accumulators at this k and session length: lower k or max_items_in_session");
        // a sketch word sums max(w, 0) * num over ALL elements that share it (<= k rows of <= max_row_len items): if even that could wrap, no sketch
        g.sketch_may_wrap = sketch_wraps(p.k, ix.max_row_len, Lmax);
    }
    const int sbits = std::max(2, bits_host(w_max) + 1), cbits = bits_host(p.k);
    // exact words for the 2048 most popular items, 4096 where a query walks many rows (measured with the end-of-round kernel, 2048 -> 4096:
    // config 3 / 4 (k = 1500) +2.6 % / +2.3 %, config 2 (k = 500) -2.6 %: clearing and harvesting the extra words costs more than they save there)
    uint32_t hot = sbits + cbits <= 32 ? (p.k >= 1024 ? 4096u : 2048u) : 0u;
    if (kn.hot_slots >= 0) hot = sbits + cbits <= 32 ? (uint32_t)kn.hot_slots : 0u;   // test knob
    hot = std::min<uint32_t>(std::min<uint32_t>(hot, a_max / 8), round_up((uint32_t)std::min<uint64_t>(ix.n_items, 1u << 20), 4)) / 4 * 4;
    c.hot_slots = hot; c.sum_bits = (uint32_t)sbits;
    // sketch: upper-bound words for all other items (needs the direct-mapped part: its top n give the threshold)
    uint32_t sk = hot && !g.sketch_may_wrap ? 8192u : 0u;
    if (kn.sketch_slots >= 0) sk = hot && !g.sketch_may_wrap && kn.sketch_slots > 0 ? floor_pow2((uint64_t)kn.sketch_slots) : 0u;   // test knob
    while (sk && (uint64_t)sk * 4 * 8 > (uint64_t)(a_max - hot * 4) * 5) sk >>= 1;   // leave >= 3/8 of the room to the exact table
    c.sketch_slots = sk; c.sketch_shift = sk ? 32u - (uint32_t)(bits_host(sk) - 1) : 31u;
    const uint64_t want_buckets = std::max<uint64_t>(61, g.need_item / 2 + 8);          // load <= 0.5 at the worst case
    c.item_buckets = prime_at_most((uint32_t)std::min<uint64_t>((a_max - hot * 4 - sk * 4) / 32, want_buckets));
    c.item_slots = c.item_buckets * 4;
    const uint32_t region_a = std::max<uint32_t>(hot * 4 + sk * 4 + c.item_slots * 8, c.sess_slots * slot_bytes);
    c.region_a_bytes = region_a;
    c.no_merge = kn.no_merge ? 1u : 0u;
    g.lds = (size_t)c.off_a + region_a;
    // tables at least twice the worst case cannot exhaust the probe budget: no retry machinery needed
    g.sess_may_overflow = (uint64_t)c.sess_slots < g.need_sess * 2; g.item_may_overflow = (uint64_t)c.item_slots < g.need_item * 2;
    return SRN_OK;
}

// Does the fast kernel (srn_fast.hip: lean instantiation, then MID over its hand-overs, then the general kernel over what is left) serve this launch?
// What the fast kernel needs of the LAUNCH is the position-set condition (DESIGN.md "Why MASKS is exact": m <= m_index, lists complete) and sketch words that cannot wrap
// at ITS session lengths (<= 10 items: the admission is per query, on the device); the batch's longest session only decides which kernels serve the hand-overs -- up to
// round 3 one session of nine items sent the whole batch to the general kernel.
struct FastPlan { bool fast = false, mid_tier = false, long_tier = false; uint32_t nb_fast = 0; };
static uint32_t fast_nb(const FlatIndex& ix, const Knobs& kn) {   // bits of a fast-kernel slot's list set (0: the index has too many sessions for any)
    const int rank_bits_f = std::max(1, bits_host(ix.n_kept ? ix.n_kept - 1 : 0));
    return kn.fast_runs == 3 && rank_bits_f <= 29 ? 3u : rank_bits_f <= 28 ? 4u : rank_bits_f <= 29 ? 3u : 0u;
}
static FastPlan fast_plan(const DeviceState* d, const FlatIndex& ix, const LaunchParams& p, const Geometry& geo, const Knobs& kn, bool has_ext) {
    FastPlan f;
    // the fast kernel packs (rank, set of <= 4 lists) into 32 bits whatever the general kernel's slots look like: up to 2^28 sessions with 4 lists per query,
    // up to 2^29 with 3 (queries with more go to the MID instantiation / the general kernel)
    f.nb_fast = fast_nb(ix, kn);
    // (round 6) an index whose lists are not ALL complete (a pre-built Avro index: srn_avro.cpp) keeps the fast kernels: the prep kernel marks the queries that an
    // incomplete list can affect (PrepHead::unsafe) and only those go to the general kernel, which then runs its row pass (geo.masks is false for such an index)
    const bool sets_exact = p.m <= ix.m_index && (ix.lists_complete || (d->di.viol != nullptr && !kn.no_viol)) && !kn.no_masks;
    // (a sketch word sums the POSITIVE parts: <= k rows of <= max_row_len items, weight <= 9 * numerator; numerators <= 55 at 10 items, <= 210 at 20)
    const bool fast_sketch_ok = (uint64_t)p.k * std::max<uint64_t>(1, ix.max_row_len) * 9ull * (p.max_len > 10 ? 210ull : 55ull) < (1ull << 32);
    f.mid_tier = !kn.no_mid && !has_ext && d->di.row_frag == 0u && p.max_len >= 5;   // (a query of <= 4 items has <= 4 lists and numerators <= 10: nothing for the MID instantiation)
    f.long_tier = f.mid_tier && !kn.no_long && !kn.no_big && p.max_len > 10;          // (round 5: sessions of 11..20 items -- LONG, a form of MID's BIG layout)
    f.fast = d->fast.row_packed != nullptr && sets_exact && (p.max_len <= 8 ? (geo.masks || !ix.lists_complete) && !geo.sketch_may_wrap : f.mid_tier && fast_sketch_ok) && f.nb_fast != 0 && kn.geometry_default() &&
             !kn.no_fast && p.k <= F_K_MAX && p.m <= F_M_MAX && p.how_many <= (uint32_t)kn.fast_how_many_max && (p.flags & ~(unsigned)SRN_FLAG_BUSINESS_LOGIC) == 0 && p.stats == nullptr && p.nb_rank == nullptr;
    return f;
}

// The serving order of a batch (FastParams::order): room for the keys the prep kernel writes, their sorted copy and the sort's scratch in ONE grow-only buffer; then the sort
// itself, enqueued on the stream the prep kernel ran on.
static int order_room(char** buf, size_t* have, uint32_t nq, unsigned long long** keys_in, unsigned long long** keys_out, void** temp, size_t* temp_bytes) {
    size_t tb = 0;
    if (sort_order_keys(nullptr, nullptr, nullptr, nq, nullptr, &tb) != hipSuccess) return fail(SRN_EHIP, "rocprim::radix_sort_keys (size query) failed");
    const size_t kb = ((size_t)nq * 8 + 255) / 256 * 256;
    int rc = ensure(buf, have, 2 * kb + tb + 256); if (rc) return rc;
    *keys_in = (unsigned long long*)*buf; *keys_out = (unsigned long long*)(*buf + kb); *temp = *buf + 2 * kb; *temp_bytes = tb;
    return SRN_OK;
}

// Latency path: a handful of evolving sessions on host pointers (srn_predict: the reference's call shape, one session per call).
// No copies, no memsets, no events: the queries are written into pinned, device-mapped memory that the kernels read directly, the
// results come back the same way; two launches (prep kernel + general kernel, one workgroup per query) and one stream synchronise.
// Returns 1 if a query needs the global-table pass (the caller then takes the normal path).
static int device_predict_tiny(DeviceState* d, const FlatIndex& ix, Workspace* w, const Geometry& geo, LaunchParams p, const uint64_t* h_items, const uint32_t* h_qoff,
                               uint64_t* h_ids, double* h_scores, uint32_t* h_counts, bool blocking_wait) {
    const size_t nitems = h_qoff[p.nq], n_out = (size_t)p.nq * p.how_many;
    size_t off = 0; auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 63) / 64 * 64; return o; };
    const size_t o_items = take(nitems * 8), o_qoff = take(((size_t)p.nq + 1) * 4), o_ids = take(n_out * 8), o_sc = take(n_out * 8), o_cnt = take((size_t)p.nq * 4),
                 o_rc = take(4), o_rl = take((size_t)p.nq * 4 + 4);
    if (w->pin_bytes < off) {
        if (w->pin) HIP_TRY(hipHostFree(w->pin));
        w->pin = nullptr; w->pin_bytes = 0;
        const size_t want = std::max<size_t>(off * 2, 256 * 1024);   // (room for a full round of the combiner at once: growing a pinned buffer stalls every lane)
        HIP_TRY(hipHostMalloc((void**)&w->pin, want, hipHostMallocMapped | hipHostMallocCoherent));   // (coherent: the fused launch's caller reads its rows as soon as the kernel's last pinned word says so)
        w->pin_bytes = want;
    }
    char* dp = nullptr; HIP_TRY(hipHostGetDevicePointer((void**)&dp, w->pin, 0));
    memcpy(w->pin + o_items, h_items, nitems * 8); memcpy(w->pin + o_qoff, h_qoff, ((size_t)p.nq + 1) * 4);
    *(volatile uint32_t*)(w->pin + o_rc) = 0;
    p.items_flat = (const uint64_t*)(dp + o_items); p.q_off = (const uint32_t*)(dp + o_qoff);
    p.out_ids = (uint64_t*)(dp + o_ids); p.out_scores = (double*)(dp + o_sc); p.out_counts = (uint32_t*)(dp + o_cnt);
    p.stats = nullptr; p.nb_rank = p.nb_num = p.nb_cnt = nullptr; p.phase_cycles = nullptr;
    hipStream_t st = w->stream;
    const size_t cap_q = std::max<size_t>(p.nq, (size_t)std::max(1, knobs().tiny_max));   // (sized once for the largest round: hipFree / hipMalloc synchronise the device)
    { int rc = ensure(&w->spill, &w->spill_bytes, cap_q * p.k * geo.slot_bytes); if (rc) return rc; }
    const uint32_t prep_stride = (uint32_t)(sizeof(PrepHead) + (size_t)p.max_len * sizeof(PrepItem));
    { int rc = ensure(&w->prep, &w->prep_bytes, cap_q * prep_stride); if (rc) return rc; }
    // The fast kernel's launch sequence on the same zero-copy buffers where it pays (round 4): a batch with a session of > 8 items puts ALL its queries on the general kernel's
    // non-position-set build -- 143 us per call against 52 us on config 3, and one such session in a combined round of srn_predict calls slows the whole round -- while the
    // lean / MID instantiations serve sessions of <= 10 items at the short sessions' cost.  Four more launches (~2 us each on an idle stream), no more synchronisation.
    const Knobs kn = knobs();
    const FastPlan plan = fast_plan(d, ix, p, geo, kn, false);
    const bool tiny_fast = plan.fast && (kn.tiny_fast == 3 || (kn.tiny_fast >= 1 && p.max_len > 8) || (kn.tiny_fast == 2 && p.nq <= (kn.no_tiny_fused ? 32u : (uint32_t)kn.tiny_fused_max)));   // (the crossover measured with tools/tiny_crossover.py: profiles/r04_serving_tiny_fast.txt)
    const uint64_t big_entries = (uint64_t)cap_q * 16 + 4096;
    if (tiny_fast) {   // (sized once, for the largest round)
        if (w->slow_cap < cap_q) { if (w->slow_list) HIP_TRY(hipFree(w->slow_list)); w->slow_list = nullptr; w->slow_cap = 0;
            HIP_TRY(hipMalloc((void**)&w->slow_list, (cap_q * 4 + 64) * 4)); w->slow_cap = cap_q; }
        { int rc = ensure(&w->fin, &w->fin_bytes, cap_q * F_FIN_BYTES + 1024); if (rc) return rc; }
        { int rc = ensure(&w->big, &w->big_bytes, big_entries * 16 + cap_q * 4 + 64); if (rc) return rc; }
    }
    // (round 5) ONE evolving session per call -- srn_predict, the reference's call shape: one launch.  The fast kernel's TINY instantiation writes the prep record itself,
    // serves the query and finishes its row from registers; the counters it publishes say whether anything is left for the kernels behind it (SRN_TINY_FUSED=0: five launches).
    const bool fused = tiny_fast && p.nq <= (uint32_t)kn.tiny_fused_max && !kn.no_tiny_fused;
    if (!fused) { w->cnt_dirty = true; HIP_TRY(launch_prep(st, d->di, p.items_flat, p.q_off, p.nq, p.m, p.max_len, w->prep, prep_stride, tiny_fast ? w->slow_cnt : nullptr)); }
    p.prep = w->prep; p.prep_stride = prep_stride;
    if (tiny_fast) {
        FastParams fp = d->fast; fp.slow_list = w->slow_list; fp.slow_cnt = w->slow_cnt; fp.nb = plan.nb_fast; fp.max_runs = plan.nb_fast; fp.fin = w->fin;
        fp.big_arena = w->big; fp.big_list = (uint32_t*)(w->big + big_entries * 16); fp.big_ticket = (unsigned long long*)(w->slow_cnt + 2); fp.big_cap_entries = (uint32_t)big_entries;
        fp.xchg = nullptr; fp.xchg_stride = 0; fp.q_base = 0; fp.order = nullptr;
        fp.mid_list = plan.mid_tier ? w->slow_list + (w->slow_cap + 16) : nullptr; fp.mid_cnt = plan.mid_tier ? w->slow_cnt + 1 : nullptr;
        fp.bigq_list = nullptr; fp.bigq_cnt = nullptr; fp.long_list = nullptr; fp.long_cnt = nullptr;   // (no BIG / LONG tier on the latency path)
        uint32_t seq = 0;
        if (fused) {
            uint32_t L1 = 0; for (uint32_t q = 0; q < p.nq; ++q) L1 = std::max(L1, h_qoff[q + 1] - h_qoff[q]);   // the call's longest session
            if (w->cnt_dirty) HIP_TRY(hipMemsetAsync(w->slow_cnt, 0, 32, st));   // (only after a call that left the counters non-zero: the fused launch leaves them at 0 unless it hands work on)
            w->cnt_dirty = true;   // (until this call is known to have ended cleanly with nothing handed on: an error return in between must not leave a stale ticket behind a "clean" flag -- ADVICE r5)
            fp.tiny_len = p.nq == 1 && L1 >= 1u && L1 <= 8u ? L1 : 0u;   // (a single session's items in the kernel arguments where they fit)
            for (uint32_t i = 0; i < fp.tiny_len; ++i) fp.tiny_items[i] = h_items[h_qoff[0] + i];
            seq = ++w->tiny_seq; if (seq == 0u) seq = ++w->tiny_seq;
            fp.host_seq = seq; fp.host_words = w->h_retry_dev;
            // (a session of > 4 items goes to the MID form at once -- the lean form would only list it: five lists, numerators beyond 15)
            HIP_TRY(launch_fast(dim3(p.nq), st, d->di, p, fp, kn.debug, 0, plan.mid_tier && L1 > 4u, false, false, true));
            fp.host_words = nullptr; fp.tiny_len = 0;
        }
        else
        HIP_TRY(launch_fast(dim3(p.nq), st, d->di, p, fp, kn.debug, 0));
        // (round 5, experiment: SRN_TINY_PHASES) A single query's call is five launches -- prep, fast kernel, general kernel, finish, finish-big -- and two of them find nothing to
        // do in 99 calls of 100.  With the path counters published by the finish kernel the host can launch what is behind it only for a call that listed work for it, and wait a
        // second time then: 46 us against 49 at p50 for one query per call, but 71 against 63 at p99, and calls of 4 / 16 queries lose at p90.  The default stays one phase.
        const int phases = fused ? 3 : kn.tiny_phases;   // (3: the fused launch is the whole first phase)   // 0: one phase (round 4); 1: finish-big stays in the first phase (a second wait only for handed-over / MID queries); 2: the first phase ends with the finish kernel
        const bool two_phase = phases != 0;
        auto rest = [&]() -> int {
            if (plan.mid_tier) HIP_TRY(launch_fast(dim3(p.nq), st, d->di, p, fp, kn.debug, 0, true));
            HIP_TRY(launch_predict(geo.masks, geo.slot64, false, 0, dim3(p.nq), geo.lds, st, d->di, p, geo.c, w->slow_list, w->slow_cnt, (uint32_t*)(dp + o_rl), (uint32_t*)(dp + o_rc), nullptr, 0,
                                   w->spill, ShardIO{}));
            HIP_TRY(launch_finish(st, d->di, fp, p.out_ids, p.out_scores, p.out_counts, p.nq, p.how_many));
            // (it also publishes the call's path counters -- handed over, listed for MID -- in the workspace's pinned words: srn_last_path_counts / srn_debug_last_mid_count after a
            //  latency-path call used to say "all through the general kernel")
            HIP_TRY(launch_finish_big(st, d->di, fp, p.out_ids, p.out_scores, p.out_counts, p.how_many, (uint32_t)std::min<uint64_t>(p.nq, (uint64_t)d->n_cu * 16), nullptr, w->slow_cnt, w->h_retry_dev));
            return SRN_OK;
        };
        if (phases == 2) HIP_TRY(launch_finish(st, d->di, fp, p.out_ids, p.out_scores, p.out_counts, p.nq, p.how_many, w->slow_cnt, w->h_retry_dev));
        else if (phases == 1) {
            HIP_TRY(launch_finish(st, d->di, fp, p.out_ids, p.out_scores, p.out_counts, p.nq, p.how_many));
            HIP_TRY(launch_finish_big(st, d->di, fp, p.out_ids, p.out_scores, p.out_counts, p.how_many, (uint32_t)std::min<uint64_t>(p.nq, (uint64_t)d->n_cu * 16), nullptr, w->slow_cnt, w->h_retry_dev));
        }
        else if (phases == 0) { int rc = rest(); if (rc) return rc; }
        auto wait = [&]() -> int {
            if (blocking_wait) {   // a round several callers share: sleep on an interrupt instead of spinning on the signal (the host's cores belong to the callers)
                if (!w->ev_block) HIP_TRY(hipEventCreateWithFlags(&w->ev_block, hipEventBlockingSync | hipEventDisableTiming));
                HIP_TRY(hipEventRecord(w->ev_block, st));
                HIP_TRY(hipEventSynchronize(w->ev_block));
            } else HIP_TRY(hipStreamSynchronize(st));
            return SRN_OK;
        };
        const volatile uint32_t* hw = w->h_retry;
        bool seen = false;
        if (fused && !blocking_wait && !kn.no_tiny_spin) {
            // the kernel's last store is the call's number in a pinned word, behind a system-scope fence: the row is readable as soon as the host sees it -- before the
            // queue's completion signal (end-of-kernel cache write-back, signal, the runtime's poll of it) has made its way.  A bounded spin: a fault shows in the stream's wait
            const auto t0 = std::chrono::steady_clock::now();
            for (uint32_t spin = 0; !seen; ++spin) {
                seen = hw[5] == seq;
                if (!seen) {
                    __builtin_ia32_pause();   // (the sibling hyper-thread may be another caller spinning on ITS workspace)
                    if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) break;
                }
            }
            std::atomic_thread_fence(std::memory_order_acquire);
        }
        if (!seen) { int rc = wait(); if (rc) return rc; }
        // the counters hw[1..4] are this call's only if the kernel published its number behind them; a fused launch that ended without doing so (it cannot, short of a
        // fault the wait above would have reported) gets the full second phase rather than a guess from stale words
        const bool published = !fused || hw[5] == seq;
        if (fused && !published) { int rc = rest(); if (rc) return rc; rc = wait(); if (rc) return rc; }
        else
        if (kn.debug && fused) {   // (SRN_DEBUG: how often a single-session call needs the kernels behind the fused launch, and why)
            static std::atomic<uint64_t> calls{0}, second{0}, why[3] = {{0}, {0}, {0}};
            const uint64_t c = calls.fetch_add(1) + 1;
            if ((hw[1] | hw[2] | hw[4]) != 0u) { second.fetch_add(1); if (hw[1]) why[0].fetch_add(1); if (hw[2]) why[1].fetch_add(1); if (hw[4]) why[2].fetch_add(1); }
            if (c % 1000 == 0) fprintf(stderr, "[srn] fused single-session calls: %llu, with a second phase %llu (handed to the general kernel %llu, listed for MID %llu, > 63 entries %llu)\n",
                                       (unsigned long long)c, (unsigned long long)second.load(), (unsigned long long)why[0].load(), (unsigned long long)why[1].load(), (unsigned long long)why[2].load());
        }
        if (fused && published && (hw[1] | hw[2] | hw[4]) == 0u) w->cnt_dirty = false;   // (a clean end with nothing handed on: the counters and the workgroup ticket are back at 0)
        if (published && two_phase && (hw[1] | hw[2] | (phases >= 2 ? hw[4] : 0u)) != 0u) {   // (handed to the general kernel | listed for MID | queries with > 63 entries)
            int rc = SRN_OK;
            w->cnt_dirty = true;
            if (fused && hw[2] == 0u) {   // (nothing listed for MID: the general kernel over what was handed to it -- it writes its rows itself -- and finish-big for a row the fused launch could not finish, each only if needed)
                const uint32_t n_slow = hw[1], n_big = hw[4];
                if (n_slow) HIP_TRY(launch_predict(geo.masks, geo.slot64, false, 0, dim3(std::min<uint32_t>(p.nq, n_slow)), geo.lds, st, d->di, p, geo.c, w->slow_list, w->slow_cnt, (uint32_t*)(dp + o_rl), (uint32_t*)(dp + o_rc), nullptr, 0,
                                                   w->spill, ShardIO{}));
                if (n_big) HIP_TRY(launch_finish_big(st, d->di, fp, p.out_ids, p.out_scores, p.out_counts, p.how_many, (uint32_t)std::min<uint64_t>(p.nq, (uint64_t)d->n_cu * 16), nullptr, w->slow_cnt, w->h_retry_dev));
            }
            else rc = rest();
            if (rc) return rc;
            rc = wait(); if (rc) return rc;
        }
    } else {
    HIP_TRY(launch_predict(geo.masks, geo.slot64, false, 0, dim3(p.nq), geo.lds, st, d->di, p, geo.c, nullptr, nullptr, (uint32_t*)(dp + o_rl), (uint32_t*)(dp + o_rc), nullptr, 0,
                           w->spill, ShardIO{}));
    if (blocking_wait) {   // a round several callers share: sleep on an interrupt instead of spinning on the signal (the host's cores belong to the callers)
        if (!w->ev_block) HIP_TRY(hipEventCreateWithFlags(&w->ev_block, hipEventBlockingSync | hipEventDisableTiming));
        HIP_TRY(hipEventRecord(w->ev_block, st));
        HIP_TRY(hipEventSynchronize(w->ev_block));
    } else HIP_TRY(hipStreamSynchronize(st));
    }
    if (*(volatile uint32_t*)(w->pin + o_rc) != 0) return 1;   // (rare: tables too small for some query)
    // what the timing / path-count APIs report after this call: its query count, how many of them the fast sequence handed to the general kernel (all of them where the
    // two-launch form ran), not timed (the stream is idle here: nothing of an earlier call is still writing the pinned words)
    ++w->untimed_calls; w->last_nq = p.nq; w->last_retry = 0; w->last_fast = tiny_fast; w->last_mid = tiny_fast && plan.mid_tier; w->last_untimed = true;   // (`calls` indexes the event ring: timed calls only)
    const uint64_t* r_ids = (const uint64_t*)(w->pin + o_ids); const double* r_sc = (const double*)(w->pin + o_sc); const uint32_t* r_cnt = (const uint32_t*)(w->pin + o_cnt);
    for (uint32_t q = 0; q < p.nq; ++q) {
        const uint32_t n = r_cnt[q] == 0xFFFFFFFFu ? 0u : std::min<uint32_t>(r_cnt[q], p.how_many);
        h_counts[q] = r_cnt[q];
        memcpy(h_ids + (size_t)q * p.how_many, r_ids + (size_t)q * p.how_many, (size_t)n * 8);
        memcpy(h_scores + (size_t)q * p.how_many, r_sc + (size_t)q * p.how_many, (size_t)n * 8);
        std::fill(h_ids + (size_t)q * p.how_many + n, h_ids + (size_t)(q + 1) * p.how_many, 0ull);      // the unused tail of each row reads as 0
        std::fill(h_scores + (size_t)q * p.how_many + n, h_scores + (size_t)(q + 1) * p.how_many, 0.0);
    }
    return SRN_OK;
}


// =====================================================================================================================================================================
// The PERSISTENT latency path (round 6).  srn_predict on the one-launch form costs 28 us from a C++ host of which ~13 us are the GPU's work: the rest
// is queue submission and completion signalling.  Here a workgroup of vmis_fast_kernel<TINY> stays RESIDENT (one third of one CU) and polls a pinned control block
// (ServeCtl, srn_kernels.h): the host writes the session and then its number, the workgroup serves it -- same code, same rows -- and answers with the number; the host
// spins on its own memory.  Doorbell round trip measured on this platform: 1.6 us (tools/ring_probe.hip).  A session the fused form would hand on (general kernel, MID,
// > 63 entries: 0.3-3 % of config 3's) comes back with status 1 and takes the launch path, as does any call whose k / m / how_many / flags are not the resident ones.
// hipFree / hipDeviceSynchronize would wait for a resident workgroup for ever: every such call of the library goes through srn_hipsync.h, which makes them leave first.
// =====================================================================================================================================================================
struct ServeForm {   // one resident launch: N workgroups of the lean form (sessions of <= 4 items) or of the form for 5..10 items
    DeviceState* d = nullptr; bool mid = false; std::atomic<bool> dead{false}; uint32_t n = 0;
    hipStream_t st = nullptr;
    ServeCtl* ctl = nullptr; ServeCtl* ctl_dev = nullptr;   // [n], pinned
    char* rows = nullptr; char* rows_dev = nullptr; size_t row_bytes = 0;   // [n] x (ids | scores), pinned
    char *prep = nullptr, *fin = nullptr, *big = nullptr; uint32_t *slow_list = nullptr, *slow_cnt = nullptr, *counts = nullptr;
    LaunchParams p{}; FastParams fp{};
    std::unique_ptr<std::atomic<int>[]> busy; std::vector<uint32_t> seq;   // per workgroup: taken by a caller; the last number posted
    std::mutex mu; bool running = false;   // (mu: starting / parking the launch)
};
struct ServeState {
    std::vector<ServeForm*> forms; uint32_t k = 0, m = 0, how_many = 0, flags = 0, max_items = 0; unsigned long long idle_ticks = 0;
    std::atomic<uint64_t> served{0}, not_served{0}, launches{0};
};
static std::mutex g_serve_mu; static std::vector<ServeForm*> g_serve_forms; static std::atomic<int> g_serve_running{0};

static void lane_lock(ServeForm* f, uint32_t i) { int e = 0; while (!f->busy[i].compare_exchange_weak(e, 1, std::memory_order_acquire)) { e = 0; __builtin_ia32_pause(); } }
static void lane_unlock(ServeForm* f, uint32_t i) { f->busy[i].store(0, std::memory_order_release); }
// (f->mu held, no lane held by this thread) the resident workgroups leave -- after the sessions they are serving
static void form_park(ServeForm* f) {
    if (!f->running) return;
    for (uint32_t i = 0; i < f->n; ++i) lane_lock(f, i);   // (nobody is posting, nothing is in flight)
    int cur = 0; (void)hipGetDevice(&cur);
    if (cur != f->d->device) (void)hipSetDevice(f->d->device);
    for (uint32_t i = 0; i < f->n; ++i) __atomic_store_n(&f->ctl[i].stop, 1u, __ATOMIC_RELEASE);
    (void)hipStreamSynchronize(f->st);
    if (cur != f->d->device) (void)hipSetDevice(cur);
    f->running = false; g_serve_running.fetch_sub(1);
    for (uint32_t i = 0; i < f->n; ++i) lane_unlock(f, i);
}
static std::atomic<int> g_serve_hold{0};   // > 0: some thread is inside a call that waits for the whole device (hipFree ...): no resident launch may start
void serve_hold_begin() { g_serve_hold.fetch_add(1, std::memory_order_acq_rel); serve_quiesce_all(); }
void serve_hold_end() { g_serve_hold.fetch_sub(1, std::memory_order_acq_rel); }
void serve_quiesce_all() {
    if (g_serve_running.load(std::memory_order_acquire) == 0) return;
    std::lock_guard<std::mutex> lk(g_serve_mu);
    for (ServeForm* f : g_serve_forms) { std::lock_guard<std::mutex> fl(f->mu); form_park(f); }
}
// (f->mu held, no lane held by this thread) start the resident launch
static int form_launch(ServeState* s, ServeForm* f) {
    // (a launch that starts now would be parked at once, or worse, waited for by the free in progress: the caller takes the launch path this time.  A hold that begins
    //  right after this look parks what is started here: its quiesce takes f->mu, which the caller of this function holds)
    if (g_serve_hold.load(std::memory_order_acquire) > 0) return SRN_ESTATE;
    if (f->running) {   // (workgroups that left by their idle timeout: the launch is over once all have)
        bool all_gone = true; for (uint32_t i = 0; i < f->n; ++i) all_gone = all_gone && __atomic_load_n(&f->ctl[i].alive, __ATOMIC_ACQUIRE) == 0u;
        if (!all_gone) { form_park(f); } else { HIP_TRY(hipSetDevice(f->d->device)); HIP_TRY(hipStreamSynchronize(f->st)); f->running = false; g_serve_running.fetch_sub(1); }
    }
    for (uint32_t i = 0; i < f->n; ++i) lane_lock(f, i);
    auto unlock_all = [&]() { for (uint32_t i = 0; i < f->n; ++i) lane_unlock(f, i); };
    hipError_t e = hipSetDevice(f->d->device);
    for (uint32_t i = 0; i < f->n && e == hipSuccess; ++i) { ServeCtl& c = f->ctl[i]; c.stop = 0; c.alive = 1; c.idle_ticks = s->idle_ticks; c.seq = f->seq[i]; c.done_seq = f->seq[i]; }   // (a post an earlier launch never answered is not served late)
    std::atomic_thread_fence(std::memory_order_seq_cst);
    if (e == hipSuccess) e = hipMemsetAsync(f->slow_cnt, 0, 32, f->st);
    if (e == hipSuccess) e = launch_fast(dim3(f->n), f->st, f->d->di, f->p, f->fp, false, 0, f->mid, false, false, true);
    if (e == hipSuccess) { f->running = true; g_serve_running.fetch_add(1); s->launches.fetch_add(1); }
    unlock_all();
    return e == hipSuccess ? SRN_OK : fail(SRN_EHIP, std::string("the persistent latency path: ") + hipGetErrorString(e));
}
static void form_free(ServeForm* f) {
    { std::lock_guard<std::mutex> lk(g_serve_mu); g_serve_forms.erase(std::remove(g_serve_forms.begin(), g_serve_forms.end(), f), g_serve_forms.end()); }
    { std::lock_guard<std::mutex> fl(f->mu); form_park(f); }
    for (void* q : {(void*)f->prep, (void*)f->fin, (void*)f->big, (void*)f->slow_list, (void*)f->slow_cnt, (void*)f->counts}) if (q) (void)hipFree(q);
    if (f->ctl) (void)hipHostFree(f->ctl); if (f->rows) (void)hipHostFree(f->rows);
    if (f->st) (void)hipStreamDestroy(f->st);
    delete f;
}
// srn_index_serve_stop may run while other threads are inside srn_predict: a caller that read the state's pointer a moment ago must still find its memory.  Stop therefore
// RETIRES the state -- its forms are marked dead (no new session is posted, no launch is started again) and parked (every session in flight is answered first) -- and
// only device_release (srn_index_free: nobody may be using the index any more) frees it.
void serve_free_retired(DeviceState* d) {
    std::vector<ServeState*> old;
    { std::lock_guard<std::mutex> lk(d->mu); old.swap(d->serve_retired); }
    for (ServeState* s : old) { for (ServeForm* f : s->forms) form_free(f); delete s; }
}
int device_serve_stop(DeviceState* d) {
    ServeState* s = d->serve.exchange(nullptr);
    if (!s) return SRN_OK;
    HIP_TRY(hipSetDevice(d->device));
    for (ServeForm* f : s->forms) { f->dead.store(true); std::lock_guard<std::mutex> fl(f->mu); form_park(f); }
    { std::lock_guard<std::mutex> lk(d->mu); d->serve_retired.push_back(s); }
    return SRN_OK;
}
int device_serve_start(DeviceState* d, const FlatIndex& ix, uint32_t k, uint32_t m, uint32_t how_many, uint32_t flags, uint32_t lanes, uint32_t max_items, uint32_t idle_ms) {
    int rc = device_serve_stop(d); if (rc) return rc;
    if (lanes == 0) return SRN_OK;
    if (lanes > 256) return fail(SRN_ERANGE, "at most 256 resident workgroups per form");
    HIP_TRY(hipSetDevice(d->device));
    const Knobs kn = knobs();
    ServeState* s = new ServeState(); s->k = k; s->m = m; s->how_many = how_many; s->flags = flags; s->max_items = std::min<uint32_t>(std::max<uint32_t>(max_items, 1), F_MID_LMAX);
    s->idle_ticks = (unsigned long long)std::max<uint32_t>(idle_ms, 1) * 100000ull;   // (wall_clock64: 100 MHz)
    auto undo = [&](int code, const std::string& why) { for (ServeForm* f : s->forms) form_free(f); delete s; return fail(code, why); };   // (never published: nobody else holds it)
    // A resident launch never ends, and HIP multiplexes its streams onto a few hardware queues per priority level: on a queue shared with another stream everything
    // behind the resident kernel would wait for ever.  The resident streams take the LOWEST priority level, which nothing else in this library uses.
    int prio_lo = 0, prio_hi = 0; (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    for (int form = 0; form < (s->max_items > 4 ? 2 : 1); ++form) {
        ServeForm* f = new ServeForm(); f->d = d; f->mid = form == 1; f->n = lanes; s->forms.push_back(f);
        f->busy.reset(new std::atomic<int>[lanes]); for (uint32_t i = 0; i < lanes; ++i) f->busy[i].store(0); f->seq.assign(lanes, 0u);
        { std::lock_guard<std::mutex> lk(g_serve_mu); g_serve_forms.push_back(f); }
        LaunchParams& p = f->p;
        p.nq = lanes; p.k = k; p.m = m; p.how_many = how_many; p.flags = flags; p.max_len = f->mid ? F_MID_LMAX : 8u;
        Geometry geo; if (make_geometry(d, ix, p, 0, geo) != SRN_OK) return undo(SRN_EINVAL, "the persistent latency path: no LDS geometry for these parameters");
        const FastPlan plan = fast_plan(d, ix, p, geo, kn, false);
        if (!plan.fast || (f->mid && !plan.mid_tier)) return undo(SRN_EINVAL, "the persistent latency path serves what the fast kernels serve (k <= 1536, m <= 2560 and <= m_index, how_many <= 64, complete lists)");
        f->row_bytes = ((size_t)how_many * 16 + 63) / 64 * 64;
        if (hipStreamCreateWithPriority(&f->st, hipStreamNonBlocking, prio_lo) != hipSuccess ||
            hipHostMalloc((void**)&f->ctl, sizeof(ServeCtl) * lanes, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
            hipHostMalloc((void**)&f->rows, f->row_bytes * lanes, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
            hipHostGetDevicePointer((void**)&f->ctl_dev, f->ctl, 0) != hipSuccess || hipHostGetDevicePointer((void**)&f->rows_dev, f->rows, 0) != hipSuccess)
            return undo(SRN_ENOMEM, "the persistent latency path: pinned control blocks");
        memset(f->ctl, 0, sizeof(ServeCtl) * lanes); memset(f->rows, 0, f->row_bytes * lanes);
        const uint32_t prep_stride = (uint32_t)(sizeof(PrepHead) + (size_t)p.max_len * sizeof(PrepItem));
        const uint64_t big_entries = (uint64_t)lanes * 1024 + 4096;
        if (hipMalloc((void**)&f->prep, (size_t)prep_stride * lanes + 256) != hipSuccess || hipMalloc((void**)&f->fin, (size_t)F_FIN_BYTES * lanes + 1024) != hipSuccess ||
            hipMalloc((void**)&f->big, big_entries * 16 + (size_t)lanes * 4 + 64) != hipSuccess || hipMalloc((void**)&f->slow_list, ((size_t)lanes * 4 + 64) * 4) != hipSuccess ||
            hipMalloc((void**)&f->slow_cnt, 32) != hipSuccess || hipMalloc((void**)&f->counts, (size_t)lanes * 4 + 64) != hipSuccess)
            return undo(SRN_ENOMEM, "the persistent latency path: device scratch");
        p.items_flat = nullptr; p.q_off = nullptr;
        // a workgroup's row: ids at rows + q * how_many * 8 ... the kernels address rows as out_ids[q * how_many + rank]: ids of all workgroups, then scores of all
        p.out_ids = (uint64_t*)f->rows_dev; p.out_scores = (double*)(f->rows_dev + (size_t)lanes * how_many * 8); p.out_counts = f->counts;
        if ((size_t)lanes * how_many * 16 > f->row_bytes * lanes) return undo(SRN_EINVAL, "row buffer");
        p.stats = nullptr; p.nb_rank = p.nb_num = p.nb_cnt = nullptr; p.phase_cycles = nullptr; p.prep = f->prep; p.prep_stride = prep_stride;
        FastParams& fp = f->fp; fp = d->fast;
        fp.slow_list = f->slow_list; fp.slow_cnt = f->slow_cnt; fp.nb = plan.nb_fast; fp.max_runs = plan.nb_fast; fp.fin = f->fin;
        fp.big_arena = f->big; fp.big_list = (uint32_t*)(f->big + big_entries * 16); fp.big_ticket = (unsigned long long*)(f->slow_cnt + 2); fp.big_cap_entries = (uint32_t)big_entries;
        fp.xchg = nullptr; fp.xchg_stride = 0; fp.q_base = 0; fp.order = nullptr; fp.mid_list = nullptr; fp.mid_cnt = nullptr; fp.bigq_list = nullptr; fp.bigq_cnt = nullptr; fp.long_list = nullptr; fp.long_cnt = nullptr;
        fp.tiny_len = 0; fp.host_seq = 0; fp.host_words = nullptr; fp.serve = f->ctl_dev;
    }
    for (ServeForm* f : s->forms) { std::lock_guard<std::mutex> fl(f->mu); rc = form_launch(s, f); if (rc && rc != SRN_ESTATE) return undo(rc, last_error_string()); }   // (SRN_ESTATE: another thread is freeing device memory right now -- the first call that wants the form starts it)
    d->serve.store(s);
    return SRN_OK;
}
int device_serve_stats(DeviceState* d, uint64_t* served, uint64_t* not_served, uint64_t* launches, uint32_t* lanes) {
    ServeState* s = d->serve.load();
    if (served) *served = s ? s->served.load() : 0; if (not_served) *not_served = s ? s->not_served.load() : 0; if (launches) *launches = s ? s->launches.load() : 0;
    uint32_t n = 0; if (s) for (ServeForm* f : s->forms) n += f->n;
    if (lanes) *lanes = n;
    return SRN_OK;
}
int device_serve_last_stamps(DeviceState* d, uint32_t* out4) {   // (measurement aid) 100 MHz ticks of the lean form's first workgroup's last session: waited | record written | answered
    ServeState* s = d->serve.load();
    if (!s || s->forms.empty()) return fail(SRN_ESTATE, "nothing resident");
    for (int i = 0; i < 4; ++i) out4[i] = s->forms[0]->ctl[0].stamp[i];
    return SRN_OK;
}
int device_serve_predict(DeviceState* d, const uint64_t* items, uint32_t len, uint32_t k, uint32_t m, uint32_t how_many, uint32_t flags, uint64_t* out_ids, double* out_scores, size_t* out_n) {
    ServeState* s = d->serve.load(std::memory_order_acquire);
    if (!s || k != s->k || m != s->m || how_many != s->how_many || flags != s->flags || len == 0 || len > s->max_items) return 1;
    ServeForm* f = s->forms[len > 4 ? 1 : 0];
    if (f->dead) return 1;
    for (int attempt = 0; attempt < 2; ++attempt) {
        uint32_t li = f->n;
        for (uint32_t i = 0; i < f->n; ++i) { int e = 0; if (f->busy[i].compare_exchange_strong(e, 1, std::memory_order_acquire)) { li = i; break; } }
        if (li == f->n) { s->not_served.fetch_add(1); return 1; }   // (every resident workgroup of that form is taken: the launch path)
        ServeCtl* c = &f->ctl[li];
        if (!f->running || __atomic_load_n(&c->alive, __ATOMIC_ACQUIRE) == 0u) {   // parked (something freed device memory) or gone (idle): start the launch again, then look for a lane again
            lane_unlock(f, li);
            std::lock_guard<std::mutex> fl(f->mu);
            if (f->dead.load()) { s->not_served.fetch_add(1); return 1; }   // (retired by srn_index_serve_stop in the meantime: never started again)
            bool gone = !f->running; for (uint32_t i = 0; i < f->n && !gone; ++i) gone = __atomic_load_n(&f->ctl[i].alive, __ATOMIC_ACQUIRE) == 0u;
            if (gone) { const int lrc = form_launch(s, f); if (lrc != SRN_OK) { if (lrc != SRN_ESTATE) f->dead = true; s->not_served.fetch_add(1); return 1; } }
            continue;
        }
        uint32_t chk = 0; for (uint32_t i = 0; i < std::min<uint32_t>(len, 5u); ++i) chk ^= (uint32_t)items[i] ^ (uint32_t)(items[i] >> 32);
        for (uint32_t i = 5; i < len; ++i) c->more[i - 5] = items[i];
        for (uint32_t i = 0; i < std::min<uint32_t>(len, 5u); ++i) c->items[i] = items[i];
        uint32_t seq = ++f->seq[li]; if (seq == 0u) seq = ++f->seq[li];
        c->len = len; c->check = chk ^ seq ^ len;
        __atomic_store_n(&c->seq, seq, __ATOMIC_RELEASE);
        const auto t0 = std::chrono::steady_clock::now();
        bool seen = false;
        for (uint32_t spin = 0;; ++spin) {
            if (__atomic_load_n(&c->done_seq, __ATOMIC_ACQUIRE) == seq) { seen = true; break; }
            __builtin_ia32_pause();
            if ((spin & 1023u) == 1023u) {
                if (__atomic_load_n(&c->alive, __ATOMIC_ACQUIRE) == 0u) break;   // (it left between our look and our post: its idle timeout)
                if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(500)) { f->dead = true; break; }   // (a fault: the form is not used again)
            }
        }
        int ret = 1;
        if (seen && __atomic_load_n(&c->status, __ATOMIC_RELAXED) == 0u) {
            const uint32_t n = std::min<uint32_t>(__atomic_load_n(&c->count, __ATOMIC_RELAXED), how_many);
            memcpy(out_ids, f->rows + ((size_t)li * how_many) * 8, (size_t)n * 8);
            memcpy(out_scores, f->rows + ((size_t)f->n * how_many + (size_t)li * how_many) * 8, (size_t)n * 8);
            *out_n = n; ret = 0;
        }
        lane_unlock(f, li);
        (ret == 0 ? s->served : s->not_served).fetch_add(1);
        return ret;
    }
    s->not_served.fetch_add(1);
    return 1;
}

// the workspace's side stream (highest priority: a hardware queue of its own, dispatched ahead of the running call's persistent workgroups) and its events
static int ensure_side(Workspace* w) {
    if (w->side) return SRN_OK;
    int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    HIP_TRY(hipStreamCreateWithPriority(&w->side, hipStreamNonBlocking, hi));
    for (int i = 0; i < 2; ++i) { HIP_TRY(hipEventCreateWithFlags(&w->ev_prep[i], hipEventDisableTiming)); HIP_TRY(hipEventCreateWithFlags(&w->ev_done[i], hipEventDisableTiming)); }
    HIP_TRY(hipEventCreateWithFlags(&w->ev_fork, hipEventDisableTiming)); HIP_TRY(hipEventCreateWithFlags(&w->ev_join, hipEventDisableTiming));
    return SRN_OK;
}

int device_predict(DeviceState* d, const FlatIndex& ix, const LaunchParams& p_in, bool on_device, void* user_stream,
                   const uint64_t* h_items, const uint32_t* h_qoff, uint64_t* h_ids, double* h_scores, uint32_t* h_counts,
                   uint32_t* h_stats, uint32_t* h_nb_rank, uint32_t* h_nb_num, uint32_t* h_nb_cnt, const ExtLists* ext, bool reserve_only, bool blocking_wait) {
    HIP_TRY(hipSetDevice(d->device));
    LaunchParams p = p_in;
    if (p.nq == 0) return SRN_OK;
    // SRN_FLAG_INPUTS_RESIDENT (device-pointer calls): the query buffers are complete in device memory at call time, so the prep kernel of THIS call may run on a side
    // stream while the previous call's kernels still occupy the caller's stream (see below); the kernels never see the flag
    const bool resident = on_device && !ext && (p.flags & SRN_FLAG_INPUTS_RESIDENT) != 0u;
    p.flags &= ~(unsigned)SRN_FLAG_INPUTS_RESIDENT;
    // item-sharded index, lists mode (device_shard_lists_*): the posting lists of ALL shards for this batch arrive in one gathered
    // buffer with the prep records already written against it; everything below runs unchanged on "an index whose postings live there"
    DeviceIndex di = d->di;
    if (ext) di.post_rank = ext->post_rank;
    p.phase_cycles = d->phase_on ? d->d_phase : nullptr;
    Workspace* w = ws_acquire(d, on_device, user_stream);
    if (!w) return fail(SRN_EHIP, "cannot create HIP stream / events");
    struct Rel { DeviceState* d; Workspace* w; bool b; ~Rel() { if (w) ws_release(d, w, b); } } rel{d, w, on_device};
    hipStream_t st = on_device ? (hipStream_t)user_stream : w->stream;

    // ---- geometry ----------------------------------------------------------------------
    Geometry geo; { int rc = make_geometry(d, ix, p, 0, geo); if (rc) return rc; }
    const KernelCfg& c = geo.c; const bool slot64 = geo.slot64; const uint32_t slot_bytes = geo.slot_bytes; const size_t lds = geo.lds;
    const uint64_t need_sess = geo.need_sess, need_item = geo.need_item;
    const bool may_overflow = geo.sess_may_overflow || geo.item_may_overflow;

    if (!on_device && !reserve_only && !h_stats && !h_nb_rank) {
        const Knobs kn0 = knobs();
        if (p.nq <= (uint32_t)kn0.tiny_max && !d->phase_on && !kn0.dense) {
            const int rc = device_predict_tiny(d, ix, w, geo, p, h_items, h_qoff, h_ids, h_scores, h_counts, blocking_wait);
            if (rc != 1) return rc;   // (1: some query needs the global-table pass -- the paths below have it)
        }
        // everything larger: chunks through pinned staging, uploads / kernels / downloads overlapped (srn_hostpipe.hip); each chunk comes back here as a
        // device-pointer call on one of the pipeline's kernel streams.  (This call's own workspace goes back to the pool first.)
        ws_release(d, w, false); rel.w = nullptr;
        return device_predict_host_pipelined(d, ix, p, h_items, h_qoff, h_ids, h_scores, h_counts);
    }
    // ---- buffers -----------------------------------------------------------------------
    const size_t n_out = (size_t)p.nq * p.how_many;
    size_t nitems = 0;
    if (!on_device) {
        nitems = h_qoff[p.nq];
        size_t off = 0; auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
        const size_t o_items = take(nitems * 8), o_qoff = take(((size_t)p.nq + 1) * 4), o_ids = take(n_out * 8), o_sc = take(n_out * 8),
                     o_cnt = take((size_t)p.nq * 4), o_st = take(h_stats ? (size_t)p.nq * 32 : 0),
                     o_nr = take(h_nb_rank ? (size_t)p.nq * p.k * 4 : 0), o_nn = take(h_nb_rank ? (size_t)p.nq * p.k * 4 : 0),
                     o_nc = take(h_nb_rank ? (size_t)p.nq * 4 : 0);
        int rc = ensure(&w->stage, &w->stage_bytes, off); if (rc) return rc;
        char* s = w->stage;
        p.items_flat = (const uint64_t*)(s + o_items); p.q_off = (const uint32_t*)(s + o_qoff);
        p.out_ids = (uint64_t*)(s + o_ids); p.out_scores = (double*)(s + o_sc); p.out_counts = (uint32_t*)(s + o_cnt);
        p.stats = h_stats ? (uint32_t*)(s + o_st) : nullptr;
        p.nb_rank = h_nb_rank ? (uint32_t*)(s + o_nr) : nullptr; p.nb_num = h_nb_rank ? (uint32_t*)(s + o_nn) : nullptr;
        p.nb_cnt = h_nb_rank ? (uint32_t*)(s + o_nc) : nullptr;
        HIP_TRY(hipMemsetAsync(p.out_ids, 0, n_out * 8, st));      // unused tail of each row reads as 0
        HIP_TRY(hipMemsetAsync(p.out_scores, 0, n_out * 8, st));
        HIP_TRY(hipMemcpyAsync((void*)p.items_flat, h_items, nitems * 8, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync((void*)p.q_off, h_qoff, ((size_t)p.nq + 1) * 4, hipMemcpyHostToDevice, st));
    }
    // Opt-in (SRN_DENSE=1): THREE workgroups per CU -- the 80-VGPR build of the kernel with a 52 KB LDS geometry -- and what that
    // geometry cannot hold goes through the normal one (second tier) before the global-table pass.  Measured on config 3 / 4:
    // +8.6 % / +4.7 % queries/s (-1 % / -2 % with the end-of-round kernel), but the 80-VGPR build's register spills double the memory traffic (14.5 -> 31.8 GB per launch), so it
    // stays off until the build fits without spilling.
    const Knobs kn = knobs();
    Geometry g3; bool dense = !slot64 && kn.dense && !kn.lds_budget_kb;
    if (dense && (make_geometry(d, ix, p, 0, g3, 52 * 1024) != SRN_OK || g3.slot64 || g3.masks != geo.masks)) dense = false;
    uint64_t g_stride = 0; int retry_blocks = 0;
    KernelCfg cg = c;
    if (may_overflow || dense) {
        if (w->retry_cap < p.nq) { if (w->retry_list) HIP_TRY(hipFree(w->retry_list)); w->retry_list = nullptr; w->retry_cap = 0;
            HIP_TRY(hipMalloc((void**)&w->retry_list, (size_t)p.nq * 4 + 64)); w->retry_cap = p.nq; }
        cg.sess_slots = (uint32_t)std::min<uint64_t>(1u << 30, std::max<uint64_t>(256, ceil_pow2(need_sess * 2)));
        cg.item_buckets = prime_at_least(need_item / 2 + 64); cg.item_slots = cg.item_buckets * 4;
        g_stride = std::max<uint64_t>((uint64_t)cg.sess_slots * slot_bytes, (uint64_t)cg.hot_slots * 4 + (uint64_t)cg.sketch_slots * 4 + (uint64_t)cg.item_slots * 8);
        g_stride = (g_stride + 255) / 256 * 256;
        retry_blocks = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)d->n_cu, (2ull << 30) / g_stride));
        int rc = ensure(&w->gscratch, &w->gscratch_bytes, g_stride * retry_blocks); if (rc) return rc;
        if (dense) {
            if (w->retry_cap2 < p.nq) { if (w->retry_list2) HIP_TRY(hipFree(w->retry_list2)); w->retry_list2 = nullptr; w->retry_cap2 = 0;
                HIP_TRY(hipMalloc((void**)&w->retry_list2, (size_t)p.nq * 4 + 64)); w->retry_cap2 = p.nq; }
        }
    }

    // ---- launches ----------------------------------------------------------------------
    const uint32_t blocks_per_cu = std::max<uint32_t>(1, (160 * 1024) / (uint32_t)lds);
    // 16 workgroups per resident slot: the hardware hands out workgroups as slots free up, so smaller chunks of queries shorten the
    // tail of a launch (measured on config 3: x1 10.62 ms, x4 10.50, x16 10.35, x32 10.32)
    const uint32_t grid_mult = (uint32_t)kn.grid_mult;
    const uint32_t grid = (uint32_t)std::min<uint64_t>(p.nq, (uint64_t)d->n_cu * std::min<uint32_t>(blocks_per_cu, 2048 / kBlock) * grid_mult);
    // per-block global copy of the neighbour list (walk B reads it after phase 4a has reused the LDS)
    const uint32_t grid3 = dense ? (uint32_t)std::min<uint64_t>(p.nq, (uint64_t)d->n_cu * std::min<uint32_t>(std::max<uint32_t>(1, (160 * 1024) / (uint32_t)g3.lds), 2048 / kBlock) * grid_mult) : 0u;
    { int rc = ensure(&w->spill, &w->spill_bytes, (size_t)std::max<uint32_t>(std::max(grid, grid3), (uint32_t)retry_blocks) * p.k * slot_bytes); if (rc) return rc; }
    char* spill = w->spill;
    const uint32_t prep_stride = (uint32_t)(sizeof(PrepHead) + (size_t)p.max_len * sizeof(PrepItem));
    if (!ext) { int rc = ensure(&w->prep, &w->prep_bytes, (size_t)p.nq * prep_stride); if (rc) return rc; }
    if (resident) {   // a second set of prep records: call i + 1's prep kernel runs while call i's kernels still read theirs
        int rc = ensure(&w->prep2, &w->prep2_bytes, (size_t)p.nq * prep_stride); if (rc) return rc;
        rc = ensure_side(w); if (rc) return rc;
    }
    const FastPlan plan = fast_plan(d, ix, p, geo, kn, ext != nullptr);
    const uint32_t nb_fast = plan.nb_fast; const bool mid_tier = plan.mid_tier, fast = plan.fast;
    bool sback_second = false;   // the item shard's wave-per-query back end ran with the fast kernel's back-end form behind it (its list is the MID tier's: srn_debug_last_mid_count reports it)
    if (fast) {   // (all allocations of a call happen before its first launch)
        if (w->slow_cap < p.nq) { if (w->slow_list) HIP_TRY(hipFree(w->slow_list)); w->slow_list = nullptr; w->slow_cap = 0;
            HIP_TRY(hipMalloc((void**)&w->slow_list, ((size_t)p.nq * 4 + 64) * 4)); w->slow_cap = p.nq; }   // (second to fourth part: the MID instantiation's list, its BIG form's, the LONG form's)
        { int rc = ensure(&w->fin, &w->fin_bytes, (size_t)p.nq * F_FIN_BYTES + 1024); if (rc) return rc; }   // one record per query for vmis_finish_kernel
        const uint64_t big_entries = std::min<uint64_t>(0x7FFFFFF0ull, (uint64_t)p.nq * 16 + 4096);
        { int rc = ensure(&w->big, &w->big_bytes, big_entries * 16 + (size_t)p.nq * 4 + 64); if (rc) return rc; }
    }
    // the serving order: the batch sorted by each query's most popular item (keys from the prep kernel, one radix sort behind it)
    const bool ordered = fast && !ext && kn.order_min > 0 && p.nq >= (uint32_t)kn.order_min;
    unsigned long long *okeys_in = nullptr, *okeys_out = nullptr; void* otemp = nullptr; size_t otemp_bytes = 0;
    // (one order per set of prep records: a resident call's prep kernel and sort run on the side stream while the previous call's kernels still read THEIR order)
    const bool order_set2 = resident && (w->resident_calls & 1u) != 0u;
    if (ordered) { int rc = order_room(order_set2 ? &w->order2 : &w->order, order_set2 ? &w->order2_bytes : &w->order_bytes, p.nq, &okeys_in, &okeys_out, &otemp, &otemp_bytes); if (rc) return rc; }
    if (reserve_only) return SRN_OK;   // (srn_index_reserve: the workspace is sized, nothing was enqueued)
    if (ext && ext->mode == 1) {
        // The shard group's neighbours pipeline, FRONT: find_neighbors alone for the queries [q_lo, nq) of the batch this rank fronts -- the fast kernel's front end against
        // the replicated posting lists, neighbour lists into the exchange buffer.  Nothing else of the launch sequence runs (the back end of every rank follows the all-gather).
        if (!fast) return fail(SRN_EINVAL, "the neighbours pipeline needs the fast kernel's query shape (device_fast_eligible)");
        if (ext->prep_stride != prep_stride) return fail(SRN_EINVAL, "prep record stride mismatch");
        if (ext->q_lo >= p.nq) return SRN_OK;
        p.prep = ext->prep; p.prep_stride = prep_stride;
        FastParams fp = d->fast; fp.nb = nb_fast; fp.max_runs = nb_fast; fp.slow_list = nullptr; fp.slow_cnt = nullptr; fp.fin = nullptr; fp.mid_list = nullptr; fp.mid_cnt = nullptr; fp.bigq_list = nullptr; fp.bigq_cnt = nullptr; fp.long_list = nullptr; fp.long_cnt = nullptr;
        fp.xchg = ext->xchg; fp.xchg_stride = ext->xchg_stride; fp.q_base = ext->q_lo; fp.order = nullptr;
        const uint64_t cnt = p.nq - ext->q_lo, res_wg = (uint64_t)d->n_cu * F_WG_PER_CU;
        const uint32_t grid_front = (uint32_t)std::min<uint64_t>(cnt, std::min<uint64_t>(res_wg * 64, std::max<uint64_t>(res_wg * 16, cnt / 12)));
        HIP_TRY(launch_fast(dim3(grid_front), st, di, p, fp, kn.debug, 1));
        return SRN_OK;
    }
    if (ext && ext->mode == 2 && !fast) return fail(SRN_EINVAL, "the neighbours pipeline needs the fast kernel's query shape (device_fast_eligible)");
    // The prep kernel clears the launch sequence's counters where it runs on this stream ahead of everything that uses them (two fill kernels otherwise: 6 us each)
    const bool prep_clears = !ext && !resident;
    if ((may_overflow || dense) && !prep_clears) HIP_TRY(hipMemsetAsync(w->retry_cnt, 0, 4, st));
    if (dense) HIP_TRY(hipMemsetAsync(w->retry_cnt2, 0, 4, st));
    hipEvent_t* ev = w->ev[w->calls % Workspace::RING];
    // Per-kernel events only when someone asked for kernel times: an event between two kernels idles the stream for ~6 us (three of them: 18 us of a 4 096-query batch's 250)
    const bool timed = d->timing.load(std::memory_order_relaxed);
    w->ring_timed[w->calls % Workspace::RING] = timed;
    if (timed) HIP_TRY(hipEventRecord(ev[0], st));
    if (ext) { if (ext->prep_stride != prep_stride) return fail(SRN_EINVAL, "prep record stride mismatch"); p.prep = ext->prep; p.prep_stride = prep_stride; }
    else if (resident) {
        // prep of this call on the side stream: behind the call that used this set of records two calls ago, beside the previous call's kernels
        const int par = (int)(w->resident_calls & 1u);
        char* rec = par ? w->prep2 : w->prep;
        // (ev_done[set] = the end of the LAST call that read record set `set`, resident or not: a non-resident device-pointer call on this stream reads w->prep too)
        if (!w->rec_used[par] && w->calls > 0) { HIP_TRY(hipEventRecord(w->ev_done[par], st)); w->rec_used[par] = true; }   // (earlier calls of this workspace that left no event: everything enqueued so far)
        if (w->rec_used[par]) HIP_TRY(hipStreamWaitEvent(w->side, w->ev_done[par], 0));
        HIP_TRY(launch_prep(w->side, di, p.items_flat, p.q_off, p.nq, p.m, p.max_len, rec, prep_stride, nullptr, nullptr, nullptr, 0, okeys_in));
        if (ordered) HIP_TRY(sort_order_keys(w->side, okeys_in, okeys_out, p.nq, otemp, &otemp_bytes));
        HIP_TRY(hipEventRecord(w->ev_prep[par], w->side));
        HIP_TRY(hipStreamWaitEvent(st, w->ev_prep[par], 0));
        p.prep = rec; p.prep_stride = prep_stride;
    }
    else { HIP_TRY(launch_prep(st, di, p.items_flat, p.q_off, p.nq, p.m, p.max_len, w->prep, prep_stride, fast ? w->slow_cnt : nullptr, (may_overflow || dense) ? w->retry_cnt : nullptr, nullptr, 0, okeys_in));
           if (ordered) HIP_TRY(sort_order_keys(st, okeys_in, okeys_out, p.nq, otemp, &otemp_bytes));
           p.prep = w->prep; p.prep_stride = prep_stride; }
    if (timed) HIP_TRY(hipEventRecord(ev[3], st));
    const uint32_t* final_list = w->retry_list; uint32_t* final_cnt = w->retry_cnt;
    // The global-table retry pass serves a handful of queries with one workgroup each, ~0.3 ms behind everything else on config 5.  Where earlier
    // calls on this workspace did retry queries (the pinned counter of the last finished call says so: a hint, read without synchronising) it is forked onto the side
    // stream right behind the general kernel and runs beside the finish kernels; where nothing is ever retried (configs 2-4) the two extra events would cost more
    // than the empty pass.
    // (Tried in round 3 and dropped: the general kernel over the handed-over queries + the global-table pass on the side stream, beside the two finish kernels -- 38 + 4 us
    //  against 17 + 10 us at 4 096 queries.  4 096 / 65 536 / 2^20 queries: 0.220 / 1.759 / 25.28 ms forked, 0.215 / 1.759 / 25.25 ms in line.)
    const bool fork_retry = fast && may_overflow && !dense && on_device && *(volatile uint32_t*)w->h_retry != 0u && w->h_retry_valid;
    if (fork_retry) { int rc = ensure_side(w); if (rc) return rc; }
    // The fast kernel (srn_fast.hip) serves the common query shape; what it cannot take -- decided per query, on the device -- is
    // queued on slow_list and served by the general kernel right behind it.
    if (fast) {
        w->cnt_dirty = true;   // (the latency path's fused launch wants the counters at 0 and clears them itself after a sequence like this one)
        if (!prep_clears) HIP_TRY(hipMemsetAsync(w->slow_cnt, 0, 32, st));   // (slow_cnt[0] = handed-over queries, [1] = the MID instantiation's list, [2..3] = the 64-bit ticket of vmis_finish_big_kernel's list, [4] = MID-BIG's list)
        // The fast kernel's workgroups walk their queries in a pipeline (the next record is fetched during the current query), so they want ~12 queries each;
        // beyond that, more and smaller workgroups shorten the tail of the launch.  Measured on config 3 (ms per launch at 8 / 16 / 32 / 64 resident sets): 2^20 queries
        // 26.29 / 26.06 / 25.85 / 25.81; 2^18: - / 6.62 / 6.57 / 6.63; 2^16: - / 1.70 / 1.72 / 1.78.
        const uint64_t resident = (uint64_t)d->n_cu * F_WG_PER_CU;
        const uint32_t grid_f = kn.grid_mult_set ? (uint32_t)std::min<uint64_t>(p.nq, resident * grid_mult)
                                                 : (uint32_t)std::min<uint64_t>(p.nq, std::min<uint64_t>(resident * 64, std::max<uint64_t>(resident * 16, p.nq / 12)));
        const uint64_t big_entries = std::min<uint64_t>(0x7FFFFFF0ull, (uint64_t)p.nq * 16 + 4096);
        FastParams fp = d->fast; fp.slow_list = w->slow_list; fp.slow_cnt = w->slow_cnt; fp.nb = nb_fast; fp.max_runs = nb_fast; fp.fin = w->fin;
        fp.big_arena = w->big; fp.big_list = (uint32_t*)(w->big + big_entries * 16); fp.big_ticket = (unsigned long long*)(w->slow_cnt + 2); fp.big_cap_entries = (uint32_t)big_entries;
        fp.xchg = nullptr; fp.xchg_stride = 0; fp.q_base = 0;
        fp.mid_list = mid_tier ? w->slow_list + (w->slow_cap + 16) : nullptr; fp.mid_cnt = mid_tier ? w->slow_cnt + 1 : nullptr;
        // (round 5: the BIG form also without a MID tier -- it takes the lean shape's oversized queries of the headline batches; not over an item shard's fragments)
        fp.bigq_list = !kn.no_big && !ext && d->di.row_frag == 0u ? w->slow_list + 2 * (w->slow_cap + 16) : nullptr; fp.bigq_cnt = fp.bigq_list ? w->slow_cnt + 4 : nullptr;
        fp.long_list = plan.long_tier ? w->slow_list + 3 * (w->slow_cap + 16) : nullptr; fp.long_cnt = fp.long_list ? w->slow_cnt + 5 : nullptr;
        const bool back = ext && ext->mode == 2;   // neighbour lists from the exchange buffer (any rank's front end), this shard's rows
        if (back) { fp.xchg = ext->xchg; fp.xchg_stride = ext->xchg_stride; }
        fp.order = ordered ? okeys_out : back ? ext->order : nullptr;
        const uint32_t grid_fo = fp.order ? std::max<uint32_t>(8u, grid_f / 8u * 8u) : grid_f;   // (an ordered launch walks an eighth of the order per XCD: the grid is a multiple of 8)
        if (back && d->sback.frag8 && !kn.no_sback && p.max_len <= 8) {
            // the item shard's own back end (srn_sback.hip): one wave per query, 12 per CU; a persistent grid of a few waves per resident slot
            const bool small = (uint64_t)d->di.n_kept + 1u < (1ull << 29);   // (the kernel's presence forms address the fragments through a 32-bit buffer descriptor)
            SBackParams sbp = d->sback; if (!kn.sback_bitmap || !small) sbp.present = nullptr;
            sbp.finish_here = kn.no_sback_finish ? 0u : 1u;
            sbp.pbyte_shift = ext->pbytes && small ? (uint32_t)ix.shard : 8u;   // (not small: the bytes are in the records, this shard fetches every fragment all the same)
            // the streaming form where the shard holds its fragments in the posting order of the very lists the records were written against (9 waves per CU: 16.8 KB each)
            const bool stream = ext->positions;
            if (stream && !(d->sb_frag_post && d->sb_post_for == ext->post_rank)) return fail(SRN_ESTATE, "the batch's neighbours came as posting positions, but this shard does not hold its fragments in posting order");
            const uint64_t slots = (uint64_t)d->n_cu * (stream ? 9 : 12);
            uint32_t grid_b = (uint32_t)std::min<uint64_t>(p.nq, slots * (kn.grid_mult_set ? grid_mult : stream ? 4 : 16));   // (gather form, round 6: 1 / 2 / 4 / 8 / 16 / 32 waves per slot = 1.295 / 1.247 / 1.191 / 1.148 / 1.115 / 1.114 ms -- the tail of a persistent grid costs more than the first query of a wave, which has no prefetched record)
            if (fp.order) grid_b = std::max<uint32_t>(8u, grid_b / 8u * 8u);
            if (stream) {
                int rc = ensure(&w->sb_scr, &w->sb_scr_bytes, (size_t)grid_b * shard_back_scratch_words() * 4); if (rc) return rc;
                sbp.frag_post = (const uint2*)d->sb_frag_post; sbp.post_rank = ext->post_rank; sbp.scr = (uint32_t*)w->sb_scr;
            }
            const bool second = !kn.no_sback_second && !stream; sback_second = second;   // (the streaming form's records are not neighbour slots)  // what outgrows the wave's room: the round-4 back end (eight waves per query) over a list, before the general kernel
            if (second) { fp.mid_list = w->slow_list + (w->slow_cap + 16); fp.mid_cnt = w->slow_cnt + 1; }
            HIP_TRY(launch_shard_back(dim3(grid_b), st, di, p, fp, sbp, kn.debug));
            d->sback_launches.fetch_add(1, std::memory_order_relaxed);
            if (second) { FastParams fl = fp; fl.order = nullptr; HIP_TRY(launch_fast(dim3((uint32_t)std::min<uint64_t>(p.nq, resident * 4)), st, di, p, fl, kn.debug, 2)); fp.mid_list = nullptr; fp.mid_cnt = nullptr; }
        } else
        HIP_TRY(launch_fast(dim3(grid_fo), st, di, p, fp, kn.debug, back ? 2 : 0));
        if (timed) HIP_TRY(hipEventRecord(ev[4], st));
        // The MID instantiation over what the lean one listed for it (sessions of <= 10 items, <= 8 lists); what it cannot take either joins slow_list.  The list's length
        // is known on the device only: a fixed grid, workgroups beyond the list leave at once.
        if (mid_tier) HIP_TRY(launch_fast(dim3((uint32_t)std::min<uint64_t>(p.nq, resident * 8)), st, di, p, fp, kn.debug, 0, true));
        // ... and MID's BIG form (80 KB of LDS, two workgroups per CU) over what MID passed on only for want of merge-buffer room
        if (fp.bigq_list) HIP_TRY(launch_fast(dim3((uint32_t)std::min<uint64_t>(p.nq, (uint64_t)d->n_cu * 2 * 4)), st, di, p, fp, kn.debug, 0, true, true));
        // ... and the LONG form over the sessions of 11..20 items the lean instantiation listed for it
        if (fp.long_list) HIP_TRY(launch_fast(dim3((uint32_t)std::min<uint64_t>(p.nq, (uint64_t)d->n_cu * 2 * 4)), st, di, p, fp, kn.debug, 0, true, true, true));
        HIP_TRY(launch_predict(geo.masks, slot64, false, 0, dim3(grid), lds, st, di, p, c, w->slow_list, w->slow_cnt, w->retry_list, w->retry_cnt, nullptr, 0, spill, ShardIO{}));
        if (fork_retry) {   // the global-table pass beside the finish kernels (they touch disjoint rows: a finish kernel only completes rows flagged by the fast kernel)
            HIP_TRY(hipEventRecord(w->ev_fork, st)); HIP_TRY(hipStreamWaitEvent(w->side, w->ev_fork, 0));
            HIP_TRY(launch_predict(geo.masks, slot64, true, 0, dim3(retry_blocks), c.off_a, w->side, di, p, cg, final_list, final_cnt, nullptr, nullptr, w->gscratch, g_stride, spill, ShardIO{}));
            HIP_TRY(hipEventRecord(w->ev_join, w->side));
        }
        HIP_TRY(launch_finish(st, di, fp, p.out_ids, p.out_scores, p.out_counts, p.nq, p.how_many));
        // (it also writes the call's two counters -- global-table queries, handed-over queries; both final once the general kernel is done -- into the pinned words)
        HIP_TRY(launch_finish_big(st, di, fp, p.out_ids, p.out_scores, p.out_counts, p.how_many, (uint32_t)std::min<uint64_t>(p.nq, (uint64_t)d->n_cu * 16),
                                  (may_overflow || dense) ? final_cnt : nullptr, w->slow_cnt, w->h_retry_dev));
    } else
    if (dense) {
        HIP_TRY(launch_predict(geo.masks, false, false, 0, dim3(grid3), g3.lds, st, di, p, g3.c, nullptr, nullptr, w->retry_list, w->retry_cnt, nullptr, 0, spill, ShardIO{}, 3));
        HIP_TRY(launch_predict(geo.masks, false, false, 0, dim3(grid), lds, st, di, p, c, w->retry_list, w->retry_cnt, w->retry_list2, w->retry_cnt2, nullptr, 0, spill, ShardIO{}));
        final_list = w->retry_list2; final_cnt = w->retry_cnt2;
    } else
        HIP_TRY(launch_predict(geo.masks, slot64, false, 0, dim3(grid), lds, st, di, p, c, nullptr, nullptr, w->retry_list, w->retry_cnt, nullptr, 0, spill, ShardIO{}));
    if (!fast && timed) HIP_TRY(hipEventRecord(ev[4], st));
    if (timed) HIP_TRY(hipEventRecord(ev[1], st));
    if (fork_retry) HIP_TRY(hipStreamWaitEvent(st, w->ev_join, 0));
    else if (may_overflow || dense) {
        const size_t lds_g = c.off_a;
        HIP_TRY(launch_predict(geo.masks, slot64, true, 0, dim3(retry_blocks), lds_g, st, di, p, cg, final_list, final_cnt, nullptr, nullptr,
                               w->gscratch, g_stride, spill, ShardIO{}));
        if (!fast) HIP_TRY(hipMemcpyAsync(w->h_retry, final_cnt, 4, hipMemcpyDeviceToHost, st));   // (fast: vmis_finish_big_kernel wrote both words.  Not fast: last_fast = false says
                                                                                                  //  "all of last_nq" -- no host write into a pinned word that an earlier call may still be writing)
    }
    HIP_TRY(hipEventRecord(ev[2], st));
    if (resident) { const int par = (int)(w->resident_calls & 1u); HIP_TRY(hipEventRecord(w->ev_done[par], st)); w->rec_used[par] = true; ++w->resident_calls; }
    else if (!ext && w->side) { HIP_TRY(hipEventRecord(w->ev_done[0], st)); w->rec_used[0] = true; }   // (a workspace that has served resident calls: the next one's side-stream prep must not overwrite w->prep under this call's kernels)
    ++w->calls; w->last_retry = (may_overflow || dense) ? 1 : 0; w->last_nq = p.nq; w->last_fast = fast; w->last_mid = fast && (mid_tier || sback_second); w->last_untimed = false;
    if (may_overflow || dense) w->h_retry_valid = true;   // (from now on the pinned counter holds a finished call's count -- or is being overwritten by a newer one)

    if (!on_device) {
        HIP_TRY(hipMemcpyAsync(h_ids, p.out_ids, n_out * 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(h_scores, p.out_scores, n_out * 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(h_counts, p.out_counts, (size_t)p.nq * 4, hipMemcpyDeviceToHost, st));
        if (h_stats) HIP_TRY(hipMemcpyAsync(h_stats, p.stats, (size_t)p.nq * 32, hipMemcpyDeviceToHost, st));
        if (h_nb_rank) {
            HIP_TRY(hipMemcpyAsync(h_nb_rank, p.nb_rank, (size_t)p.nq * p.k * 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(h_nb_num, p.nb_num, (size_t)p.nq * p.k * 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(h_nb_cnt, p.nb_cnt, (size_t)p.nq * 4, hipMemcpyDeviceToHost, st));
        }
        HIP_TRY(hipStreamSynchronize(st));
    }
    return SRN_OK;
}

// One stage of the item-sharded pipeline (device buffers, asynchronous on `stream`); see ShardIO.  Like the unsharded launch sequence, a stage has a second pass
// with its tables in global memory for the queries whose session or item table does not fit LDS (numerator slots only: with position sets the shard group runs the
// lists pipeline, and the stand-alone srn_shard_stage_* calls mark such a query 0xFFFFFFFF as before).
int device_shard_stage(DeviceState* d, const FlatIndex& ix, int stage, const LaunchParams& p_in, const ShardIO& sh, void* stream) {
    HIP_TRY(hipSetDevice(d->device));
    LaunchParams p = p_in;
    if (p.nq == 0) return SRN_OK;
    p.phase_cycles = nullptr;
    Geometry geo;
    const uint32_t sort_room = stage == 2 ? (uint32_t)ceil_pow2(std::max<uint32_t>(p.k, 2)) * 8 : 0;   // stage B sorts the neighbour list in region B
    int rc = make_geometry(d, ix, p, sort_room, geo); if (rc) return rc;
    const uint32_t blocks_per_cu = std::max<uint32_t>(1, (160 * 1024) / (uint32_t)geo.lds);
    const uint32_t grid = (uint32_t)std::min<uint64_t>(p.nq, (uint64_t)d->n_cu * std::min<uint32_t>(blocks_per_cu, 2048 / kBlock) * 4);
    hipStream_t st = (hipStream_t)stream;
    hipError_t e;
    if (stage < 1 || stage > 3) return fail(SRN_EINVAL, "bad stage");
    const bool retry = !geo.masks && (geo.sess_may_overflow || geo.item_may_overflow);
    Workspace* w = nullptr;
    struct Rel { DeviceState* d; Workspace* w; ~Rel() { if (w) ws_release(d, w, true); } } rel{d, nullptr};
    KernelCfg cg = geo.c; uint64_t g_stride = 0; int retry_blocks = 0;
    if (retry) {
        w = ws_acquire(d, true, stream); rel.w = w;
        if (!w) return fail(SRN_EHIP, "cannot create HIP stream / events");
        if (w->retry_cap < p.nq) { if (w->retry_list) HIP_TRY(hipFree(w->retry_list)); w->retry_list = nullptr; w->retry_cap = 0;
            HIP_TRY(hipMalloc((void**)&w->retry_list, (size_t)p.nq * 4 + 64)); w->retry_cap = p.nq; }
        cg.sess_slots = (uint32_t)std::min<uint64_t>(1u << 30, std::max<uint64_t>(256, ceil_pow2(geo.need_sess * 2)));
        cg.item_buckets = prime_at_least(geo.need_item / 2 + 64); cg.item_slots = cg.item_buckets * 4;
        g_stride = std::max<uint64_t>((uint64_t)cg.sess_slots * geo.slot_bytes, (uint64_t)cg.hot_slots * 4 + (uint64_t)cg.sketch_slots * 4 + (uint64_t)cg.item_slots * 8);
        g_stride = (g_stride + 255) / 256 * 256;
        retry_blocks = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)d->n_cu, (2ull << 30) / g_stride));
        rc = ensure(&w->gscratch, &w->gscratch_bytes, g_stride * retry_blocks); if (rc) return rc;
        HIP_TRY(hipMemsetAsync(w->retry_cnt, 0, 4, st));
    }
    e = launch_predict(geo.masks, geo.slot64, false, stage, dim3(grid), geo.lds, st, d->di, p, geo.c, nullptr, nullptr, w ? w->retry_list : nullptr, w ? w->retry_cnt : nullptr, nullptr, 0, nullptr, sh);
    if (e != hipSuccess) return fail(SRN_EHIP, std::string("shard stage launch: ") + hipGetErrorString(e));
    if (retry) {
        e = launch_predict(false, geo.slot64, true, stage, dim3(retry_blocks), geo.c.off_a, st, d->di, p, cg, w->retry_list, w->retry_cnt, nullptr, nullptr, w->gscratch, g_stride, nullptr, sh);
        if (e != hipSuccess) return fail(SRN_EHIP, std::string("shard stage launch (global tables): ") + hipGetErrorString(e));
    }
    return SRN_OK;
}
// ---- item-sharded index, lists mode (srn_shard.hip): the steps either side of the exchanges; device buffers, asynchronous on `stream` ----
bool device_shard_lists_supported(DeviceState* d, const FlatIndex& ix, const LaunchParams& p) {
    Geometry g;
    return make_geometry(d, ix, p, 0, g) == SRN_OK && g.masks && (p.flags & ~(unsigned)SRN_FLAG_BUSINESS_LOGIC) == 0 && p.stats == nullptr && p.nb_rank == nullptr;
}
// the batch shape the fast kernel takes (the same test device_predict makes): what the shard group's neighbours pipeline needs on top of lists mode
// (the shard group's choice of pipeline must be the SAME on every rank: `max_row_len_all` = the longest row of the whole index -- a shard's own longest fragment differs from
//  rank to rank and is never longer -- and the packed rows, which device_attach may fail to allocate on ONE rank, are checked where the postings are set, not here)
bool device_fast_eligible(DeviceState* d, const FlatIndex& ix, const LaunchParams& p, uint64_t max_row_len_all) {
    Geometry geo;
    if (make_geometry(d, ix, p, 0, geo) != SRN_OK) return false;
    const Knobs kn = knobs();
    const int rank_bits_f = std::max(1, bits_host(ix.n_kept ? ix.n_kept - 1 : 0));
    const bool wraps = max_row_len_all ? sketch_wraps(p.k, std::max<uint64_t>(max_row_len_all, ix.max_row_len), p.max_len) : geo.sketch_may_wrap;
    return (max_row_len_all != 0 || d->fast.row_packed != nullptr) && geo.masks && !wraps && rank_bits_f <= 29 && kn.geometry_default() && !kn.no_fast && p.k <= F_K_MAX && p.m <= F_M_MAX &&
           p.how_many <= 24 && (p.flags & ~(unsigned)SRN_FLAG_BUSINESS_LOGIC) == 0 && p.stats == nullptr && p.nb_rank == nullptr && p.max_len <= 8;
}
uint32_t device_prep_stride(uint32_t max_len) { return (uint32_t)(sizeof(PrepHead) + (size_t)max_len * sizeof(PrepItem)); }
int device_shard_lists_head(DeviceState* d, const LaunchParams& p, void* pos, int* head, void* stream) {
    HIP_TRY(hipSetDevice(d->device));
    if (p.nq == 0) return SRN_OK;
    HIP_TRY(launch_shard_lists_head((hipStream_t)stream, d->di, p.items_flat, p.q_off, p.nq, p.m, p.max_len, (ShardPos*)pos, head));
    return SRN_OK;
}
int device_shard_lists_count(DeviceState* d, const LaunchParams& p, const void* pos, const int* head, uint32_t* kept, int* tot, void* stream) {
    HIP_TRY(hipSetDevice(d->device));
    if (p.nq == 0) return SRN_OK;
    HIP_TRY(launch_shard_lists_count((hipStream_t)stream, d->di, p.q_off, p.nq, p.max_len, (const ShardPos*)pos, head, kept, tot));
    return SRN_OK;
}
int device_shard_lists_copy(DeviceState* d, const LaunchParams& p, const void* pos, const uint32_t* kept, const long long* off, uint32_t* out, void* stream) {
    HIP_TRY(hipSetDevice(d->device));
    if (p.nq == 0) return SRN_OK;
    HIP_TRY(launch_shard_lists_copy((hipStream_t)stream, d->di, p.nq, p.max_len, (const ShardPos*)pos, kept, off, out));
    return SRN_OK;
}
int device_shard_lists_predict(DeviceState* d, const FlatIndex& ix, const LaunchParams& p, uint32_t n_shards, const uint32_t* kept_g, const long long* off_g,
                               unsigned long long shard_stride, const uint32_t* lists_g, const int* head, const void* pos_local, char* records, void* stream,
                               const unsigned long long* shard_base, bool direct) {
    HIP_TRY(hipSetDevice(d->device));
    if (p.nq == 0) return SRN_OK;
    if (n_shards != ix.n_shards) return fail(SRN_EINVAL, "n_shards differs from the number of shards this index was cut into");
    if (!device_shard_lists_supported(d, ix, p)) return fail(SRN_EINVAL, "lists mode needs position-set slots (sessions of <= 8 items, m <= m_index, complete lists): use the three-stage pipeline");
    const uint32_t stride = device_prep_stride(p.max_len);
    HIP_TRY(launch_shard_prep((hipStream_t)stream, p.items_flat, p.q_off, p.nq, p.max_len, n_shards, kept_g, off_g, shard_stride, head, (const ShardPos*)pos_local, records, stride, shard_base, direct));
    ExtLists ext{records, stride, direct ? d->di.post_rank : lists_g};
    return device_predict(d, ix, p, true, stream, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &ext);
}

// ---- item-sharded index, neighbours pipeline (round 4): the posting lists are replicated (`post` = the device state of the whole index or of its postings-only view),
// the rows stay sharded.  The record of a query is the same on every rank except for the dense idx of its items (this shard's numbering).
int device_shard_nb_prep(DeviceState* d, DeviceState* post, const LaunchParams& p, char* records, void* stream, char** order_buf, size_t* order_bytes, const unsigned long long** order_out) {
    HIP_TRY(hipSetDevice(d->device));
    if (order_out) *order_out = nullptr;
    if (p.nq == 0) return SRN_OK;
    const Knobs kn = knobs();
    const bool ordered = order_buf && kn.order_min > 0 && p.nq >= (uint32_t)kn.order_min;
    unsigned long long *kin = nullptr, *kout = nullptr; void* temp = nullptr; size_t tb = 0;
    if (ordered) { int rc = order_room(order_buf, order_bytes, p.nq, &kin, &kout, &temp, &tb); if (rc) return rc; }
    HIP_TRY(launch_prep((hipStream_t)stream, post->di, p.items_flat, p.q_off, p.nq, p.m, p.max_len, records, device_prep_stride(p.max_len), nullptr, nullptr, d->di.id_table, d->di.id_mask, kin));
    if (ordered) { HIP_TRY(sort_order_keys((hipStream_t)stream, kin, kout, p.nq, temp, &tb)); *order_out = kout; }
    return SRN_OK;
}
int device_shard_nb_front(DeviceState* d, const FlatIndex& ix, DeviceState* post, const LaunchParams& p_in, const char* records, uint32_t* xchg, uint32_t xchg_stride, uint32_t q_lo, uint32_t q_hi, void* stream) {
    if (q_lo >= q_hi) return SRN_OK;
    LaunchParams p = p_in; p.nq = q_hi;
    ExtLists ext{records, device_prep_stride(p.max_len), post->di.post_rank, 1, xchg, xchg_stride, q_lo};
    return device_predict(d, ix, p, true, stream, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &ext);
}
// the fronting rank's neighbour lists [q_lo, q_hi) of `xin` -> position records in `xout` (the streaming form's exchange format, srn_sback.hip)
int device_shard_nb_positions(DeviceState* d, const FlatIndex& ix, DeviceState* post, const LaunchParams& p, const char* records, const uint32_t* xin, uint32_t in_stride, uint32_t* xout, uint32_t out_stride,
                              uint32_t q_lo, uint32_t q_hi, void* stream) {
    if (q_lo >= q_hi) return SRN_OK;
    HIP_TRY(hipSetDevice(d->device));
    const uint32_t grid = (uint32_t)std::min<uint64_t>(q_hi - q_lo, (uint64_t)d->n_cu * 16 * 4);
    HIP_TRY(launch_shard_nb_positions(dim3(grid), (hipStream_t)stream, records, device_prep_stride(p.max_len), p.max_len, xin, in_stride, xout, out_stride, post->di.post_rank, q_lo, q_hi, p.m,
                                      fast_nb(ix, knobs()) == 3u));
    return SRN_OK;
}
// words per query of that format for this batch shape, 0 if the batch (or this build's knobs) has no streaming form: RANK-INVARIANT inputs only
uint32_t device_shard_nb_positions_stride(const LaunchParams& p) {
    const Knobs kn = knobs();
    if (kn.no_sback_stream || kn.no_sback || p.max_len > 8) return 0u;
    return shard_nb_positions_stride(p.k, p.m);
}
const uint32_t* device_sback_present(const DeviceState* d, size_t* words) {
    if (!d || !d->sback.present) return nullptr;
    if (words) *words = d->sback_present_words;
    return d->sback.present;
}
bool device_shard_nb_presence_wanted() { const Knobs kn = knobs(); return !kn.no_sback_pbytes && !kn.no_sback; }
int device_shard_nb_presence(DeviceState* d, const FlatIndex& ix, const LaunchParams& p, const char* records, uint32_t* xchg, uint32_t stride, const uint8_t* pbytes, uint32_t q_lo, uint32_t q_hi, void* stream) {
    if (q_lo >= q_hi) return SRN_OK;
    HIP_TRY(hipSetDevice(d->device));
    const uint32_t grid = (uint32_t)std::min<uint64_t>(q_hi - q_lo, (uint64_t)d->n_cu * 32 * 4);
    HIP_TRY(launch_shard_nb_presence(dim3(grid), (hipStream_t)stream, records, device_prep_stride(p.max_len), p.max_len, xchg, stride, p.k, pbytes, (uint32_t)ix.n_kept, q_lo, q_hi, fast_nb(ix, knobs()) == 3u));
    return SRN_OK;
}
int device_shard_nb_back(DeviceState* d, const FlatIndex& ix, DeviceState* post, const LaunchParams& p, const char* records, uint32_t* xchg, uint32_t xchg_stride, void* stream, const unsigned long long* order, bool positions, bool pbytes) {
    if (p.nq == 0) return SRN_OK;
    ExtLists ext{records, device_prep_stride(p.max_len), post->di.post_rank, 2, xchg, xchg_stride, 0u, order, positions, pbytes};
    return device_predict(d, ix, p, true, stream, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &ext);
}

int device_slot_bytes(DeviceState* d, const FlatIndex& ix, uint32_t max_len, uint32_t* num_bits) {   // element size of the packed candidate / neighbour buffers
    LaunchParams p{}; p.max_len = max_len; p.k = 1; p.m = 1; Geometry g;
    if (make_geometry(d, ix, p, 0, g) != SRN_OK) return -1;
    if (num_bits) *num_bits = g.c.num_bits;
    return (int)g.slot_bytes;
}

int device_last_kernel_ms(DeviceState* d, double* ms_main, double* ms_retry, uint32_t* retried) {
    HIP_TRY(hipSetDevice(d->device));
    Workspace* w;
    { std::lock_guard<std::mutex> lk(d->mu); w = d->last_ws; }
    if (!w || !(w->calls + w->untimed_calls)) return fail(SRN_EINVAL, "no timed predict call yet");
    if (w->last_untimed) return fail(SRN_EINVAL, "the last call took the latency path (<= 256 sessions on host pointers): it records no events");
    if (!w->ring_timed[(w->calls - 1) % Workspace::RING]) return fail(SRN_EINVAL, "kernel timing is off: srn_kernel_timing(idx, 1) before the calls to be timed");
    hipEvent_t* ev = w->ev[(w->calls - 1) % Workspace::RING];
    HIP_TRY(hipEventSynchronize(ev[2]));
    float a = 0, b = 0;
    HIP_TRY(hipEventElapsedTime(&a, ev[3], ev[1]));   // the predict kernel alone (the prep kernel runs between ev[0] and ev[3])
    HIP_TRY(hipEventElapsedTime(&b, ev[1], ev[2]));
    if (ms_main) *ms_main = a;
    if (ms_retry) *ms_retry = b;
    if (retried) *retried = w->last_retry ? *w->h_retry : 0;
    return SRN_OK;
}

// how the last call's queries were served: by the fast kernel / handed to the general kernel / through the global-table pass
// Sizes the workspace bound to `stream` for device-pointer calls of up to nq queries with these parameters, so that
// srn_predict_batch_device allocates nothing (hipMalloc / hipFree synchronise the device) once traffic starts.
int device_reserve(DeviceState* d, const FlatIndex& ix, const LaunchParams& p, void* stream) {
    return device_predict(d, ix, p, true, stream, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, /*reserve_only=*/true);
}

int device_last_path_counts(DeviceState* d, uint32_t* nq, uint32_t* general, uint32_t* global_pass) {
    HIP_TRY(hipSetDevice(d->device));
    Workspace* w;
    { std::lock_guard<std::mutex> lk(d->mu); w = d->last_ws; }
    if (!w || !(w->calls + w->untimed_calls)) return fail(SRN_EINVAL, "no predict call yet");
    if (!w->last_untimed) HIP_TRY(hipEventSynchronize(w->ev[(w->calls - 1) % Workspace::RING][2]));   // (the latency path returns with its stream idle)
    if (nq) *nq = w->last_nq;
    if (general) *general = w->last_fast ? w->h_retry[1] : w->last_nq;
    if (global_pass) *global_pass = w->last_retry ? w->h_retry[0] : 0;
    return SRN_OK;
}
int device_last_mid_count(DeviceState* d, uint32_t* listed, uint32_t* big_listed) {   // queries the last call's lean fast kernel listed for the MID instantiation (0: no such tier in that call)
    uint32_t nq = 0;
    const int rc = device_last_path_counts(d, &nq, nullptr, nullptr);   // (waits for the call's end)
    if (rc != SRN_OK) return rc;
    Workspace* w;
    { std::lock_guard<std::mutex> lk(d->mu); w = d->last_ws; }
    if (listed) *listed = w->last_fast && w->last_mid ? w->h_retry[2] : 0u;
    if (big_listed) *big_listed = w->last_fast ? w->h_retry[3] : 0u;   // (the BIG form also takes the lean shape's oversized queries: no MID tier needed)
    return SRN_OK;
}

int device_kernel_timing(DeviceState* d, int enable) { d->timing.store(enable != 0, std::memory_order_relaxed); return SRN_OK; }

int device_phase_cycles(DeviceState* d, int enable, unsigned long long* out16) {
    HIP_TRY(hipSetDevice(d->device));
    HIP_TRY(hipDeviceSynchronize());
    if (!d->d_phase) { HIP_TRY(hipMalloc((void**)&d->d_phase, 16 * 8)); d->allocs.push_back(d->d_phase); HIP_TRY(hipMemset(d->d_phase, 0, 16 * 8)); }
    if (out16) HIP_TRY(hipMemcpy(out16, d->d_phase, 16 * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(d->d_phase, 0, 16 * 8));
    d->phase_on = enable != 0;
    return SRN_OK;
}

// durations of the most recent min(max_n, calls, RING) predict launches of the last-used workspace, oldest first
int device_kernel_times(DeviceState* d, uint32_t max_n, double* ms_main, double* ms_retry, uint32_t* out_n, double* ms_prep, double* ms_fast) {
    HIP_TRY(hipSetDevice(d->device));
    Workspace* w;
    { std::lock_guard<std::mutex> lk(d->mu); w = d->last_ws; }
    *out_n = 0;
    if (!w || !w->calls) return SRN_OK;
    uint64_t n = std::min<uint64_t>(std::min<uint64_t>(max_n, w->calls), Workspace::RING);
    { uint64_t timed = 0; while (timed < n && w->ring_timed[(w->calls - 1 - timed) % Workspace::RING]) ++timed; n = timed; }   // the trailing run of timed calls (srn_kernel_timing)
    if (!n) return SRN_OK;
    HIP_TRY(hipEventSynchronize(w->ev[(w->calls - 1) % Workspace::RING][2]));
    for (uint64_t i = 0; i < n; ++i) {
        hipEvent_t* ev = w->ev[(w->calls - n + i) % Workspace::RING];
        float a = 0, b = 0;
        HIP_TRY(hipEventElapsedTime(&a, ev[3], ev[1]));
        HIP_TRY(hipEventElapsedTime(&b, ev[1], ev[2]));
        if (ms_main) ms_main[i] = a;
        if (ms_retry) ms_retry[i] = b;
        if (ms_prep) { float c2 = 0; HIP_TRY(hipEventElapsedTime(&c2, ev[0], ev[3])); ms_prep[i] = c2; }
        if (ms_fast) { float c2 = 0; HIP_TRY(hipEventElapsedTime(&c2, ev[3], ev[4])); ms_fast[i] = c2; }
    }
    *out_n = (uint32_t)n;
    return SRN_OK;
}

}  // namespace srn

// Index construction on the GPU (SURVEY.md 8f "next" #1): the same flat index as build_flat_index() in
// srn_index.cpp -- i.e. prepare_hashmap (src/vmisknn/vmis_index.rs:422-528) in the layout of DESIGN.md section 3 --
// built with rocPRIM radix sorts / scans instead of host loops.  The result is bit-identical to the host builder
// (tests/test_gpu_build.py); idf is evaluated on the HOST from the device-computed counts so that std::log is the
// same libm call as in the host builder and in the oracle.
//
//   1. kept sessions (0 < len <= max_session_len) sorted by (timestamp, session index)      -> recency ranks
//   2. (session,item) pairs expanded in rank order; sorted by item id; run-length encoded     -> dictionary + counts
//   3. dense idx = popularity order (count desc, id asc); id_rank = id order                 -> row_items
//   4. pairs sorted by (idx asc, rank desc); position within the item's run < m_index        -> posting lists
#include <cstring>   // rocprim's texture iterator calls the host memset
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cmath>
#include <string>
#include <vector>

#include "srn_internal.h"
#include "srn_hipsync.h"

namespace srn {

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return fail(SRN_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

namespace {

struct DevBuf {   // RAII device allocation
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, std::max<size_t>(bytes, 16)); }
    template <typename T> T* as() { return (T*)p; }
};

__global__ void k_kept_keys(const uint64_t* sess_off, const uint32_t* ts, uint64_t n, uint64_t max_len, uint64_t* keys, uint32_t* flags) {
    for (uint64_t s = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; s < n; s += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t len = sess_off[s + 1] - sess_off[s];
        const bool keep = len > 0 && len <= max_len;
        flags[s] = keep ? 1u : 0u;
        keys[s] = keep ? (((uint64_t)ts[s] << 32) | s) : ~0ull;   // dropped sessions sort to the end
    }
}
__global__ void k_row_len(const uint64_t* sorted_keys, const uint64_t* sess_off, uint64_t n_kept, uint32_t* rank_to_session, uint64_t* len) {
    for (uint64_t r = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; r < n_kept; r += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t s = (uint32_t)(sorted_keys[r] & 0xFFFFFFFFull);
        rank_to_session[r] = s; len[r] = sess_off[s + 1] - sess_off[s];
    }
}
// pair p of the rank-ordered concatenation of rows: rank by binary search in row_off, then the item id
__global__ void k_expand(const uint64_t* row_off, uint64_t n_kept, uint64_t nnz, const uint32_t* rank_to_session, const uint64_t* sess_off,
                         const uint64_t* items, uint64_t* pair_id, uint32_t* pair_rank, uint32_t* pair_pos) {
    for (uint64_t p = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; p < nnz; p += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t lo = 0, hi = n_kept;   // last r with row_off[r] <= p
        while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (row_off[mid] <= p) lo = mid; else hi = mid; }
        const uint32_t s = rank_to_session[lo];
        pair_id[p] = items[sess_off[s] + (p - row_off[lo])];
        pair_rank[p] = (uint32_t)lo; pair_pos[p] = (uint32_t)p;
    }
}
__global__ void k_check_rows(const uint64_t* row_off, uint64_t n_kept, uint64_t nnz, const uint64_t* pair_id, const uint32_t* pair_rank, uint32_t* bad) {
    for (uint64_t p = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; p + 1 < nnz; p += (uint64_t)gridDim.x * blockDim.x)
        if (pair_rank[p] == pair_rank[p + 1] && pair_id[p] >= pair_id[p + 1]) *bad = 1;   // rows must be strictly ascending
}
__global__ void k_popkey(const uint32_t* counts, uint64_t n, uint32_t* key, uint32_t* val) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) { key[i] = ~counts[i]; val[i] = (uint32_t)i; }
}
// u = unique index in id order; newidx[u] = popularity position
__global__ void k_invert(const uint32_t* order, uint64_t n, uint32_t* newidx) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) newidx[order[i]] = (uint32_t)i;
}
__global__ void k_gather_meta(const uint32_t* order, const uint64_t* uniq, const uint32_t* counts, uint64_t n, uint64_t* item_id, uint32_t* id_rank, uint32_t* cnt_by_idx) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t u = order[i]; item_id[i] = uniq[u]; id_rank[i] = u; cnt_by_idx[i] = counts[u]; }
}
// sorted-by-id pair j belongs to unique run u(j) = (#run heads at or before j) - 1
__global__ void k_heads(const uint64_t* sorted_id, uint64_t nnz, uint32_t* head) {
    for (uint64_t j = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; j < nnz; j += (uint64_t)gridDim.x * blockDim.x) head[j] = (j == 0 || sorted_id[j] != sorted_id[j - 1]) ? 1u : 0u;
}
__global__ void k_assign(const uint32_t* run_incl, const uint32_t* newidx, const uint32_t* sorted_pos, const uint32_t* pair_rank, uint64_t nnz,
                         uint32_t* row_items, uint64_t* post_key) {
    for (uint64_t j = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; j < nnz; j += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t idx = newidx[run_incl[j] - 1], p = sorted_pos[j];
        row_items[p] = idx;
        post_key[j] = ((uint64_t)idx << 32) | (uint64_t)(0xFFFFFFFFu - pair_rank[p]);   // idx asc, rank desc
    }
}
__global__ void k_post_len(const uint32_t* cnt_by_idx, uint64_t n, uint64_t m_index, uint64_t* plen, uint64_t* full) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) { plen[i] = min((uint64_t)cnt_by_idx[i], m_index); full[i] = cnt_by_idx[i]; }
}
__global__ void k_postings(const uint64_t* sorted_key, uint64_t nnz, const uint64_t* seg_start, const uint64_t* post_off, uint64_t m_index, uint32_t* post_rank) {
    for (uint64_t j = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; j < nnz; j += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t idx = (uint32_t)(sorted_key[j] >> 32);
        const uint64_t pos = j - seg_start[idx];
        if (pos < m_index) post_rank[post_off[idx] + pos] = 0xFFFFFFFFu - (uint32_t)(sorted_key[j] & 0xFFFFFFFFull);
    }
}

}  // namespace
// The serving order of a batch (srn_kernels.h FastParams::order): 64-bit keys (dense idx of the query's most popular item << 32 | query) sorted on the 16 key bits;
// asynchronous on `st`, no allocation (the caller owns `temp`: first call with temp == nullptr for its size).
hipError_t sort_order_keys(hipStream_t st, const unsigned long long* in, unsigned long long* out, size_t n, void* temp, size_t* temp_bytes) {
#ifndef SRN_ORDER_KEY2
#define SRN_ORDER_KEY2 0
#endif
    return rocprim::radix_sort_keys(temp, *temp_bytes, in, out, n, 32, 32 + (SRN_ORDER_KEY2 ? 24 : 16), st);
}
namespace {
template <typename K, typename V>
hipError_t sort_pairs(const K* kin, K* kout, const V* vin, V* vout, size_t n, int end_bit) {
    size_t tb = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, tb, kin, kout, vin, vout, n, 0, end_bit);
    if (e != hipSuccess) return e;
    DevBuf tmp; e = tmp.alloc(tb); if (e != hipSuccess) return e;
    e = rocprim::radix_sort_pairs(tmp.p, tb, kin, kout, vin, vout, n, 0, end_bit);
    if (e != hipSuccess) return e;
    return hipDeviceSynchronize();
}
template <typename K>
hipError_t sort_keys(const K* kin, K* kout, size_t n, int end_bit) {
    size_t tb = 0;
    hipError_t e = rocprim::radix_sort_keys(nullptr, tb, kin, kout, n, 0, end_bit);
    if (e != hipSuccess) return e;
    DevBuf tmp; e = tmp.alloc(tb); if (e != hipSuccess) return e;
    e = rocprim::radix_sort_keys(tmp.p, tb, kin, kout, n, 0, end_bit);
    if (e != hipSuccess) return e;
    return hipDeviceSynchronize();
}
template <typename T>
hipError_t scan_excl(const T* in, T* out, size_t n) {   // exclusive prefix sum
    size_t tb = 0;
    hipError_t e = rocprim::exclusive_scan(nullptr, tb, in, out, T(0), n, rocprim::plus<T>());
    if (e != hipSuccess) return e;
    DevBuf tmp; e = tmp.alloc(tb); if (e != hipSuccess) return e;
    e = rocprim::exclusive_scan(tmp.p, tb, in, out, T(0), n, rocprim::plus<T>());
    if (e != hipSuccess) return e;
    return hipDeviceSynchronize();
}
template <typename T>
hipError_t scan_incl(const T* in, T* out, size_t n) {
    size_t tb = 0;
    hipError_t e = rocprim::inclusive_scan(nullptr, tb, in, out, n, rocprim::plus<T>());
    if (e != hipSuccess) return e;
    DevBuf tmp; e = tmp.alloc(tb); if (e != hipSuccess) return e;
    e = rocprim::inclusive_scan(tmp.p, tb, in, out, n, rocprim::plus<T>());
    if (e != hipSuccess) return e;
    return hipDeviceSynchronize();
}
template <typename T> int download(std::vector<T>& v, const void* d, size_t n) {
    v.resize(n);
    if (n) HIP_TRY(hipMemcpy(v.data(), d, n * sizeof(T), hipMemcpyDeviceToHost));
    return SRN_OK;
}
inline int bits_of(uint64_t v) { int b = 0; while (v) { ++b; v >>= 1; } return std::max(b, 1); }
constexpr int TPB = 256;
inline dim3 grid_for(uint64_t n) { return dim3((unsigned)std::min<uint64_t>((n + TPB - 1) / TPB, 1u << 16)); }

}  // namespace

int build_flat_index_gpu(const srn_sessions_view_t& v, size_t m_index, size_t max_session_len, double idf_weighting, int device, FlatIndex& ix) {
    if (!v.sess_off || !v.max_ts || (v.sess_off[v.n_sessions] && !v.items)) return fail(SRN_EINVAL, "null session arrays");
    if (m_index == 0) return fail(SRN_EINVAL, "m_index must be > 0");
    const uint64_t n = v.n_sessions, nnz_in = v.sess_off[n];
    if (n >= 0xFFFFFFF0ull || nnz_in >= 0xFFFFFFF0ull) return fail(SRN_ERANGE, "GPU builder handles < 2^32 sessions and interactions; use the host builder");
    HIP_TRY(hipSetDevice(device));
    ix = FlatIndex();
    ix.n_sessions_total = n; ix.m_index = m_index; ix.max_session_len = max_session_len; ix.idf_weighting = idf_weighting;

    DevBuf d_off, d_items, d_ts;
    HIP_TRY(d_off.alloc((n + 1) * 8)); HIP_TRY(d_items.alloc(nnz_in * 8)); HIP_TRY(d_ts.alloc(n * 4));
    HIP_TRY(hipMemcpy(d_off.p, v.sess_off, (n + 1) * 8, hipMemcpyHostToDevice));
    if (nnz_in) HIP_TRY(hipMemcpy(d_items.p, v.items, nnz_in * 8, hipMemcpyHostToDevice));
    if (n) HIP_TRY(hipMemcpy(d_ts.p, v.max_ts, n * 4, hipMemcpyHostToDevice));

    // 1. recency ranks of the kept sessions
    DevBuf d_key, d_key2, d_flag, d_flag_scan;
    HIP_TRY(d_key.alloc(n * 8)); HIP_TRY(d_key2.alloc(n * 8)); HIP_TRY(d_flag.alloc(n * 4)); HIP_TRY(d_flag_scan.alloc(n * 4));
    if (n) {
        hipLaunchKernelGGL(k_kept_keys, grid_for(n), dim3(TPB), 0, 0, d_off.as<uint64_t>(), d_ts.as<uint32_t>(), n, (uint64_t)max_session_len, d_key.as<uint64_t>(), d_flag.as<uint32_t>());
        HIP_TRY(scan_incl(d_flag.as<uint32_t>(), d_flag_scan.as<uint32_t>(), n));
        uint32_t kept = 0; HIP_TRY(hipMemcpy(&kept, d_flag_scan.as<uint32_t>() + (n - 1), 4, hipMemcpyDeviceToHost));
        ix.n_kept = kept;
        HIP_TRY(sort_keys(d_key.as<uint64_t>(), d_key2.as<uint64_t>(), n, 64));
    }
    const uint64_t nk = ix.n_kept;
    DevBuf d_r2s, d_len, d_rowoff;
    HIP_TRY(d_r2s.alloc(nk * 4)); HIP_TRY(d_len.alloc((nk + 1) * 8)); HIP_TRY(d_rowoff.alloc((nk + 1) * 8));
    HIP_TRY(hipMemset(d_len.p, 0, (nk + 1) * 8));
    if (nk) hipLaunchKernelGGL(k_row_len, grid_for(nk), dim3(TPB), 0, 0, d_key2.as<uint64_t>(), d_off.as<uint64_t>(), nk, d_r2s.as<uint32_t>(), d_len.as<uint64_t>());
    HIP_TRY(scan_excl(d_len.as<uint64_t>(), d_rowoff.as<uint64_t>(), nk + 1));
    uint64_t nnz = 0; HIP_TRY(hipMemcpy(&nnz, d_rowoff.as<uint64_t>() + nk, 8, hipMemcpyDeviceToHost));
    ix.nnz_rows = ix.total_pairs = nnz;
    { std::vector<uint64_t> lens; int rc = download(lens, d_len.p, nk); if (rc) return rc; for (uint64_t l : lens) ix.max_row_len = std::max(ix.max_row_len, l); }

    // 2. pairs in rank order, dictionary by sort + run-length encode
    DevBuf d_pid, d_prank, d_ppos, d_sid, d_spos, d_bad;
    HIP_TRY(d_pid.alloc(nnz * 8)); HIP_TRY(d_prank.alloc(nnz * 4)); HIP_TRY(d_ppos.alloc(nnz * 4)); HIP_TRY(d_sid.alloc(nnz * 8)); HIP_TRY(d_spos.alloc(nnz * 4));
    HIP_TRY(d_bad.alloc(4)); HIP_TRY(hipMemset(d_bad.p, 0, 4));
    uint64_t n_items = 0;
    DevBuf d_uniq, d_cnt, d_nruns;
    HIP_TRY(d_uniq.alloc(nnz * 8)); HIP_TRY(d_cnt.alloc(nnz * 4)); HIP_TRY(d_nruns.alloc(8));
    if (nnz) {
        hipLaunchKernelGGL(k_expand, grid_for(nnz), dim3(TPB), 0, 0, d_rowoff.as<uint64_t>(), nk, nnz, d_r2s.as<uint32_t>(), d_off.as<uint64_t>(), d_items.as<uint64_t>(),
                           d_pid.as<uint64_t>(), d_prank.as<uint32_t>(), d_ppos.as<uint32_t>());
        hipLaunchKernelGGL(k_check_rows, grid_for(nnz), dim3(TPB), 0, 0, d_rowoff.as<uint64_t>(), nk, nnz, d_pid.as<uint64_t>(), d_prank.as<uint32_t>(), d_bad.as<uint32_t>());
        uint32_t bad = 0; HIP_TRY(hipMemcpy(&bad, d_bad.p, 4, hipMemcpyDeviceToHost));
        if (bad) return fail(SRN_EINVAL, "session rows must be strictly ascending item ids");
        HIP_TRY(sort_pairs(d_pid.as<uint64_t>(), d_sid.as<uint64_t>(), d_ppos.as<uint32_t>(), d_spos.as<uint32_t>(), nnz, 64));
        size_t tb = 0;
        HIP_TRY(rocprim::run_length_encode(nullptr, tb, d_sid.as<uint64_t>(), nnz, d_uniq.as<uint64_t>(), d_cnt.as<uint32_t>(), d_nruns.as<uint64_t>()));
        DevBuf tmp; HIP_TRY(tmp.alloc(tb));
        HIP_TRY(rocprim::run_length_encode(tmp.p, tb, d_sid.as<uint64_t>(), nnz, d_uniq.as<uint64_t>(), d_cnt.as<uint32_t>(), d_nruns.as<uint64_t>()));
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(&n_items, d_nruns.p, 8, hipMemcpyDeviceToHost));
    }
    ix.n_items = n_items;
    if (n_items >= 0xFFFFFFF0ull) return fail(SRN_ERANGE, "too many items");

    // 3. popularity order (stable sort of the id-ordered uniques by descending count), id_rank, row_items
    DevBuf d_pk, d_pk2, d_pv, d_order, d_newidx, d_itemid, d_idrank, d_cntidx, d_head, d_run, d_rowitems, d_postkey, d_postkey2;
    HIP_TRY(d_pk.alloc(n_items * 4)); HIP_TRY(d_pk2.alloc(n_items * 4)); HIP_TRY(d_pv.alloc(n_items * 4)); HIP_TRY(d_order.alloc(n_items * 4));
    HIP_TRY(d_newidx.alloc(n_items * 4)); HIP_TRY(d_itemid.alloc(n_items * 8)); HIP_TRY(d_idrank.alloc(n_items * 4)); HIP_TRY(d_cntidx.alloc(n_items * 4));
    HIP_TRY(d_head.alloc(nnz * 4)); HIP_TRY(d_run.alloc(nnz * 4)); HIP_TRY(d_rowitems.alloc(nnz * 4)); HIP_TRY(d_postkey.alloc(nnz * 8)); HIP_TRY(d_postkey2.alloc(nnz * 8));
    if (n_items) {
        hipLaunchKernelGGL(k_popkey, grid_for(n_items), dim3(TPB), 0, 0, d_cnt.as<uint32_t>(), n_items, d_pk.as<uint32_t>(), d_pv.as<uint32_t>());
        HIP_TRY(sort_pairs(d_pk.as<uint32_t>(), d_pk2.as<uint32_t>(), d_pv.as<uint32_t>(), d_order.as<uint32_t>(), n_items, 32));   // radix sort is stable: ties keep id order
        hipLaunchKernelGGL(k_invert, grid_for(n_items), dim3(TPB), 0, 0, d_order.as<uint32_t>(), n_items, d_newidx.as<uint32_t>());
        hipLaunchKernelGGL(k_gather_meta, grid_for(n_items), dim3(TPB), 0, 0, d_order.as<uint32_t>(), d_uniq.as<uint64_t>(), d_cnt.as<uint32_t>(), n_items,
                           d_itemid.as<uint64_t>(), d_idrank.as<uint32_t>(), d_cntidx.as<uint32_t>());
        hipLaunchKernelGGL(k_heads, grid_for(nnz), dim3(TPB), 0, 0, d_sid.as<uint64_t>(), nnz, d_head.as<uint32_t>());
        HIP_TRY(scan_incl(d_head.as<uint32_t>(), d_run.as<uint32_t>(), nnz));
        hipLaunchKernelGGL(k_assign, grid_for(nnz), dim3(TPB), 0, 0, d_run.as<uint32_t>(), d_newidx.as<uint32_t>(), d_spos.as<uint32_t>(), d_prank.as<uint32_t>(), nnz,
                           d_rowitems.as<uint32_t>(), d_postkey.as<uint64_t>());
        HIP_TRY(hipDeviceSynchronize());
        // 4. postings
        HIP_TRY(sort_keys(d_postkey.as<uint64_t>(), d_postkey2.as<uint64_t>(), nnz, 32 + bits_of(n_items)));
    }
    DevBuf d_plen, d_full, d_postoff, d_segstart, d_postrank;
    HIP_TRY(d_plen.alloc((n_items + 1) * 8)); HIP_TRY(d_full.alloc((n_items + 1) * 8)); HIP_TRY(d_postoff.alloc((n_items + 1) * 8)); HIP_TRY(d_segstart.alloc((n_items + 1) * 8));
    HIP_TRY(hipMemset(d_plen.p, 0, (n_items + 1) * 8)); HIP_TRY(hipMemset(d_full.p, 0, (n_items + 1) * 8));
    if (n_items) hipLaunchKernelGGL(k_post_len, grid_for(n_items), dim3(TPB), 0, 0, d_cntidx.as<uint32_t>(), n_items, (uint64_t)m_index, d_plen.as<uint64_t>(), d_full.as<uint64_t>());
    HIP_TRY(scan_excl(d_plen.as<uint64_t>(), d_postoff.as<uint64_t>(), n_items + 1));
    HIP_TRY(scan_excl(d_full.as<uint64_t>(), d_segstart.as<uint64_t>(), n_items + 1));
    uint64_t nnz_post = 0; HIP_TRY(hipMemcpy(&nnz_post, d_postoff.as<uint64_t>() + n_items, 8, hipMemcpyDeviceToHost));
    ix.nnz_post = nnz_post;
    HIP_TRY(d_postrank.alloc(nnz_post * 4));
    if (nnz) hipLaunchKernelGGL(k_postings, grid_for(nnz), dim3(TPB), 0, 0, d_postkey2.as<uint64_t>(), nnz, d_segstart.as<uint64_t>(), d_postoff.as<uint64_t>(), (uint64_t)m_index, d_postrank.as<uint32_t>());
    HIP_TRY(hipDeviceSynchronize());

    // download, then the host-side pieces: idf (same libm log as the host builder), default attributes, id table
    int rc;
    if ((rc = download(ix.item_id, d_itemid.p, n_items)) || (rc = download(ix.id_rank, d_idrank.p, n_items)) || (rc = download(ix.post_off, d_postoff.p, n_items + 1)) ||
        (rc = download(ix.post_rank, d_postrank.p, nnz_post)) || (rc = download(ix.row_off, d_rowoff.p, nk + 1)) || (rc = download(ix.row_items, d_rowitems.p, nnz)) ||
        (rc = download(ix.rank_to_session, d_r2s.p, nk))) return rc;
    std::vector<uint32_t> count; if ((rc = download(count, d_cntidx.p, n_items))) return rc;
    ix.idf.resize(n_items); ix.attr.assign(n_items, (uint8_t)SRN_ATTR_FOR_SALE);
    for (uint64_t i = 0; i < n_items; ++i) ix.idf[i] = std::log((double)ix.total_pairs / (double)count[i]) * idf_weighting;   // vmis_index.rs:509-512
    size_t tcap = 16; while (tcap < n_items * 2) tcap <<= 1;
    ix.id_table.assign(tcap, IdSlot{0, kNone, 0}); ix.id_mask = (uint32_t)(tcap - 1);
    for (uint32_t i = 0; i < n_items; ++i) {
        uint32_t h = (uint32_t)mix64(ix.item_id[i]) & ix.id_mask;
        while (ix.id_table[h].idx != kNone) h = (h + 1) & ix.id_mask;
        ix.id_table[h] = IdSlot{ix.item_id[i], i, 0};
    }
    return SRN_OK;
}

}  // namespace srn

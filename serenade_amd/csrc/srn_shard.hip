// Item-sharded index, LISTS mode: the kernels either side of the exchange.
//
// A query's candidate sessions come from the posting lists of its evolving items, and those lists are small (<= m entries each, and
// only the entries at or above the global cut x_lo matter).  So instead of exchanging per-shard CANDIDATES and finishing the cuts in
// a second stage, every shard ships the kept prefixes of the lists it owns; after one all-gather every rank holds ALL lists of the
// batch and runs the unsharded kernels (vmis_fast_kernel / vmis_predict_kernel and their fall-back chain) unchanged on "an index
// whose postings live in the gathered buffer": same candidates, same neighbours, same position sets on every rank.  Only the
// rows are local -- each rank scores the items it owns from its row fragments, its top-n is exact for those items
// (an item's whole score lives on its owner), and a final all-gather + merge by (score desc, id asc) gives the result.
//
//   shard_lists_head_kernel    one thread per query: id -> dense idx of the items this shard owns, list bounds, local x_lo / r_max, the current item's
//                              attribute byte if this shard owns it (-1 otherwise)
//        all-reduce(max) of (x_lo, r_max, attribute)
//   shard_lists_count_kernel   one thread per query: entries >= x_lo of every owned list (a prefix), the query's total
//        exclusive scan of the totals (host side: torch.cumsum)
//   shard_lists_copy_kernel    one wave per query: the kept prefixes -> the shard's flat buffer
//        all-gather of (kept counts, offsets, flat buffers)
//   shard_prep_kernel          one thread per query: the prep record (PrepHead + PrepItems) against the gathered buffer
//
// Valid where the position sets give the first-match position (MASKS geometry: sessions of <= 8 items, complete lists), with or without
// business rules (the current item's attribute byte travels with the first all-reduce); everything else takes the three-stage pipeline.
#include <hip/hip_runtime.h>

#include "srn_device.h"
#include "srn_kernels.h"

namespace srn {

__global__ __launch_bounds__(256) void shard_lists_head_kernel(DeviceIndex ix, const uint64_t* __restrict__ items_flat, const uint32_t* __restrict__ q_off,
                                                               uint32_t nq, uint32_t m, uint32_t max_len, ShardPos* __restrict__ pos_out, int* __restrict__ head) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const uint32_t qb = q_off[q], L = q_off[q + 1] - qb;
    uint32_t xlo = 0, rmax = 0; int attr = -1;
    ShardPos* const out = pos_out + (size_t)q * max_len;
    if (L != 0 && L <= max_len) {
        for (uint32_t pos = 0; pos < L; ++pos) {
            const uint64_t raw = items_flat[qb + (L - 1 - pos)];   // pos 0 = most recent item
            bool first = true;
            for (uint32_t j = 0; j < pos; ++j) first = first && (items_flat[qb + (L - 1 - j)] != raw);   // Q2: most recent occurrence only
            uint32_t idx = kNone;
            { uint32_t hh = (uint32_t)dev_mix64(raw) & ix.id_mask;
              for (;;) { const IdSlot s = ix.id_table[hh]; if (s.idx == kNone) break; if (s.key == raw) { idx = s.idx; break; } hh = (hh + 1) & ix.id_mask; } }
            uint32_t len = 0; unsigned long long base = 0;
            if (first && idx != kNone) {
                const unsigned long long o0 = ix.post_off[idx], o1 = ix.post_off[idx + 1];
                len = (uint32_t)min((unsigned long long)m, o1 - o0); base = o0;
                if (len) { rmax = max(rmax, ix.post_rank[o0]); if (len >= m) xlo = max(xlo, ix.post_rank[o0 + m - 1]); }
            }
            if (pos == 0 && idx != kNone) attr = (int)ix.meta[idx].attr;   // (an item has one owner: every other shard says -1)
            out[pos] = ShardPos{idx, len, base};
        }
    }
    head[3 * (size_t)q] = (int)xlo; head[3 * (size_t)q + 1] = (int)rmax; head[3 * (size_t)q + 2] = attr;   // (ranks are < 2^31: they index sessions)
}

__global__ __launch_bounds__(256) void shard_lists_count_kernel(DeviceIndex ix, const uint32_t* __restrict__ q_off, uint32_t nq, uint32_t max_len,
                                                                const ShardPos* __restrict__ pos_in, const int* __restrict__ head,
                                                                uint32_t* __restrict__ kept, int* __restrict__ tot) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const uint32_t L = q_off[q + 1] - q_off[q], xlo = (uint32_t)head[3 * (size_t)q];
    uint32_t total = 0;
    for (uint32_t pos = 0; pos < max_len; ++pos) {
        uint32_t lo = 0;
        if (L != 0 && L <= max_len && pos < L) {
            const ShardPos sp = pos_in[(size_t)q * max_len + pos];
            if (sp.len) {   // first index whose entry is < x_lo (the lists are sorted by rank, descending)
                const uint32_t* __restrict__ lst = ix.post_rank + sp.base;
                uint32_t hi = sp.len;
                while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (lst[mid] >= xlo) lo = mid + 1; else hi = mid; }
            }
        }
        kept[(size_t)q * max_len + pos] = lo; total += lo;
    }
    tot[q] = (int)total;
}

__global__ __launch_bounds__(256) void shard_lists_copy_kernel(DeviceIndex ix, uint32_t nq, uint32_t max_len, const ShardPos* __restrict__ pos_in,
                                                               const uint32_t* __restrict__ kept, const long long* __restrict__ off, uint32_t* __restrict__ out) {
    const uint32_t q = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (q >= nq) return;
    uint32_t* dst = out + off[q];
    for (uint32_t pos = 0; pos < max_len; ++pos) {
        const uint32_t n = kept[(size_t)q * max_len + pos];   // (wave-uniform)
        if (n == 0) continue;
        const uint32_t* __restrict__ src = ix.post_rank + pos_in[(size_t)q * max_len + pos].base;
        for (uint32_t e = lane; e < n; e += 64u) dst[e] = src[e];
        dst += n;
    }
}

__global__ __launch_bounds__(256) void shard_prep_kernel(const uint64_t* __restrict__ items_flat, const uint32_t* __restrict__ q_off, uint32_t nq, uint32_t max_len,
                                                         uint32_t n_shards, const uint32_t* __restrict__ kept_g, const long long* __restrict__ off_g,
                                                         unsigned long long shard_stride, const int* __restrict__ head, const ShardPos* __restrict__ pos_local,
                                                         char* __restrict__ out, uint32_t stride) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const uint32_t qb = q_off[q], L = q_off[q + 1] - qb;
    PrepHead h{0u, 0u, 0u, 0u, 0u, 0u, L, 0u, {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}, SRN_ATTR_NONE, 0u};
    PrepItem* items = (PrepItem*)(out + (size_t)q * stride + sizeof(PrepHead));
    if (L != 0 && L <= max_len) {
        h.xlo = (uint32_t)head[3 * (size_t)q]; h.rmax = (uint32_t)head[3 * (size_t)q + 1];
        if (head[3 * (size_t)q + 2] >= 0) h.cur_attr = (uint32_t)head[3 * (size_t)q + 2];
        for (uint32_t pos = 0; pos < L; ++pos) {
            const uint64_t raw = items_flat[qb + (L - 1 - pos)];
            bool first = true;
            for (uint32_t j = 0; j < pos; ++j) first = first && (items_flat[qb + (L - 1 - j)] != raw);
            if (first) ++h.U;   // Q1: distinct raw ids, known or not
            uint32_t kp = 0; unsigned long long base = 0;
            for (uint32_t g = 0; g < n_shards; ++g) {   // an item has one owner: at most one shard shipped this list
                const uint32_t* kq = kept_g + ((size_t)g * nq + q) * max_len;
                if (kq[pos] == 0) continue;
                unsigned long long before = 0;
                for (uint32_t j = 0; j < pos; ++j) before += kq[j];
                kp = kq[pos]; base = (unsigned long long)g * shard_stride + (unsigned long long)off_g[(size_t)g * nq + q] + before;
                break;
            }
            if (kp) { h.sumw += L - pos; if (h.nruns < 8) h.run_start[h.nruns] = h.P; ++h.nruns; }
            items[pos] = PrepItem{pos_local[(size_t)q * max_len + pos].idx, kp, h.P, kp, base};
            h.P += kp;
        }
        h.n_staged = h.P;
    }
    *(PrepHead*)(out + (size_t)q * stride) = h;
}

hipError_t launch_shard_lists_head(hipStream_t st, const DeviceIndex& di, const uint64_t* items_flat, const uint32_t* q_off, uint32_t nq, uint32_t m, uint32_t max_len,
                                   ShardPos* pos_out, int* head) {
    hipLaunchKernelGGL(shard_lists_head_kernel, dim3((nq + 255) / 256), dim3(256), 0, st, di, items_flat, q_off, nq, m, max_len, pos_out, head);
    return hipGetLastError();
}
hipError_t launch_shard_lists_count(hipStream_t st, const DeviceIndex& di, const uint32_t* q_off, uint32_t nq, uint32_t max_len, const ShardPos* pos_in, const int* head,
                                    uint32_t* kept, int* tot) {
    hipLaunchKernelGGL(shard_lists_count_kernel, dim3((nq + 255) / 256), dim3(256), 0, st, di, q_off, nq, max_len, pos_in, head, kept, tot);
    return hipGetLastError();
}
hipError_t launch_shard_lists_copy(hipStream_t st, const DeviceIndex& di, uint32_t nq, uint32_t max_len, const ShardPos* pos_in, const uint32_t* kept, const long long* off,
                                   uint32_t* out) {
    hipLaunchKernelGGL(shard_lists_copy_kernel, dim3((nq + 3) / 4), dim3(256), 0, st, di, nq, max_len, pos_in, kept, off, out);
    return hipGetLastError();
}
hipError_t launch_shard_prep(hipStream_t st, const uint64_t* items_flat, const uint32_t* q_off, uint32_t nq, uint32_t max_len, uint32_t n_shards, const uint32_t* kept_g,
                             const long long* off_g, unsigned long long shard_stride, const int* head, const ShardPos* pos_local, char* out, uint32_t stride) {
    hipLaunchKernelGGL(shard_prep_kernel, dim3((nq + 255) / 256), dim3(256), 0, st, items_flat, q_off, nq, max_len, n_shards, kept_g, off_g, shard_stride, head, pos_local,
                       out, stride);
    return hipGetLastError();
}

}  // namespace srn

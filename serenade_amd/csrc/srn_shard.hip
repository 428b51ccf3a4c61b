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

#include <cstdlib>

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
                                                         unsigned long long shard_stride, const unsigned long long* __restrict__ shard_base, const int* __restrict__ head,
                                                         const ShardPos* __restrict__ pos_local, char* __restrict__ out, uint32_t stride, bool direct) {
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
                kp = kq[pos];
                if (direct) base = pos_local[(size_t)q * max_len + pos].base;   // (a group of ONE shard: nothing travels, the kept prefix is read where it lies in the shard's own posting array)
                else base = (shard_base ? shard_base[g] : (unsigned long long)g * shard_stride) + (unsigned long long)off_g[(size_t)g * nq + q] + before;   // (shard_base: segments of different lengths, back to back)
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
                             const long long* off_g, unsigned long long shard_stride, const int* head, const ShardPos* pos_local, char* out, uint32_t stride,
                             const unsigned long long* shard_base, bool direct) {
    hipLaunchKernelGGL(shard_prep_kernel, dim3((nq + 255) / 256), dim3(256), 0, st, items_flat, q_off, nq, max_len, n_shards, kept_g, off_g, shard_stride, shard_base, head, pos_local,
                       out, stride, direct);
    return hipGetLastError();
}

// ---- the shard group's own steps (srn_group.hip): what round 2 left to the host's tensor library ------------------------------------------

// elementwise maximum of G int32 arrays (the all-reduce(max) of the heads when all shards live in one process)
__global__ __launch_bounds__(256) void shard_max_kernel(int* __restrict__ dst, const int* __restrict__ src, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = max(dst[i], src[i]);
}
hipError_t launch_shard_max(hipStream_t st, int* dst, const int* src, size_t n) {
    if (n) hipLaunchKernelGGL(shard_max_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dst, src, n);
    return hipGetLastError();
}

// Where every query's kept prefixes start inside its shard's segment, for ALL shards at once, from the all-gathered kept counts: off_g[g][q] = number of
// entries of shard g before query q, tot[g] = the shard's total -- in device memory, and in a pinned word the host reads after the one short synchronisation
// of a batch (the totals size the variable-length exchange).  Three small launches (a query per thread, coalesced): chunk sums, a scan of the chunk sums
// per shard, offsets.  (A first version scanned a whole shard with ONE workgroup: 0.3 ms per 2^18 queries on a chip with 256 CUs.)
constexpr uint32_t OFFS_CHUNK = 1024;   // queries per workgroup
__device__ __forceinline__ unsigned long long row_sum(const uint32_t* __restrict__ kq, uint32_t max_len) {
    unsigned long long v = 0; for (uint32_t j = 0; j < max_len; ++j) v += kq[j]; return v;
}
__global__ __launch_bounds__(1024) void shard_offsets_sum_kernel(const uint32_t* __restrict__ kept_g, uint32_t nq, uint32_t max_len, uint32_t nchunks,
                                                                 unsigned long long* __restrict__ chunk_tot) {
    __shared__ unsigned long long wsum[16];
    const uint32_t g = blockIdx.y, c = blockIdx.x, q = c * OFFS_CHUNK + threadIdx.x, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    unsigned long long v = q < nq ? row_sum(kept_g + ((size_t)g * nq + q) * max_len, max_len) : 0ull;
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    if (lane == 0u) wsum[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) { unsigned long long t = 0; for (int w = 0; w < 16; ++w) t += wsum[w]; chunk_tot[(size_t)g * nchunks + c] = t; }
}
__global__ __launch_bounds__(1024) void shard_offsets_scan_kernel(unsigned long long* __restrict__ chunk_tot, uint32_t nchunks, unsigned long long* __restrict__ tot_dev,
                                                                  unsigned long long* __restrict__ tot_host) {   // one workgroup per shard: exclusive scan of its chunk sums, in place
    __shared__ unsigned long long part[1024];
    const uint32_t g = blockIdx.x, t = threadIdx.x;
    unsigned long long* ct = chunk_tot + (size_t)g * nchunks;
    const uint32_t per = (nchunks + 1023u) / 1024u, c0 = min(t * per, nchunks), c1 = min(c0 + per, nchunks);
    unsigned long long sum = 0; for (uint32_t c = c0; c < c1; ++c) sum += ct[c];
    part[t] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) { const unsigned long long v = t >= d ? part[t - d] : 0ull; __syncthreads(); part[t] += v; __syncthreads(); }
    unsigned long long run = part[t] - sum;
    for (uint32_t c = c0; c < c1; ++c) { const unsigned long long v = ct[c]; ct[c] = run; run += v; }
    if (t == 1023u) { tot_dev[g] = part[1023]; if (tot_host) tot_host[g] = part[1023]; }
}
__global__ __launch_bounds__(1024) void shard_offsets_write_kernel(const uint32_t* __restrict__ kept_g, uint32_t nq, uint32_t max_len, uint32_t nchunks,
                                                                   const unsigned long long* __restrict__ chunk_base, long long* __restrict__ off_g) {
    __shared__ unsigned long long wsum[16];
    const uint32_t g = blockIdx.y, c = blockIdx.x, q = c * OFFS_CHUNK + threadIdx.x, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long v = q < nq ? row_sum(kept_g + ((size_t)g * nq + q) * max_len, max_len) : 0ull;
    unsigned long long inc = v;   // inclusive scan inside the wave, then over the 16 waves
    for (int d = 1; d < 64; d <<= 1) { const unsigned long long o = __shfl_up(inc, d); if ((int)lane >= d) inc += o; }
    if (lane == 63u) wsum[wave] = inc;
    __syncthreads();
    unsigned long long before = chunk_base[(size_t)g * nchunks + c];
    for (uint32_t w = 0; w < wave; ++w) before += wsum[w];
    if (q < nq) off_g[(size_t)g * nq + q] = (long long)(before + inc - v);
}
hipError_t launch_shard_offsets(hipStream_t st, const uint32_t* kept_g, uint32_t nq, uint32_t max_len, uint32_t n_shards, long long* off_g, unsigned long long* tot_dev,
                                unsigned long long* tot_host, unsigned long long* chunk_scratch) {   // chunk_scratch: n_shards * ceil(nq / 1024) words
    const uint32_t nchunks = (nq + OFFS_CHUNK - 1) / OFFS_CHUNK;
    hipLaunchKernelGGL(shard_offsets_sum_kernel, dim3(nchunks, n_shards), dim3(1024), 0, st, kept_g, nq, max_len, nchunks, chunk_scratch);
    hipLaunchKernelGGL(shard_offsets_scan_kernel, dim3(n_shards), dim3(1024), 0, st, chunk_scratch, nchunks, tot_dev, tot_host);
    hipLaunchKernelGGL(shard_offsets_write_kernel, dim3(nchunks, n_shards), dim3(1024), 0, st, kept_g, nq, max_len, nchunks, (const unsigned long long*)chunk_scratch, off_g);
    return hipGetLastError();
}

// ---- helpers of the three-stage pipeline inside the shard group (sessions the lists pipeline does not serve) ----
__global__ __launch_bounds__(256) void shard_min_kernel(int* __restrict__ dst, const int* __restrict__ src, size_t n) {   // element-wise minimum (the all-reduce(min) of an in-process group)
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = min(dst[i], src[i]);
}
hipError_t launch_shard_min(hipStream_t st, int* dst, const int* src, size_t n) {
    if (n) hipLaunchKernelGGL(shard_min_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dst, src, n);
    return hipGetLastError();
}
// the gathered candidate counts: a shard whose session table overflowed says 0xFFFFFFFF for the query -- remember it (flag) and feed stage B a 0 instead
__global__ __launch_bounds__(256) void shard_scrub_counts_kernel(uint32_t* __restrict__ cnt_g, uint32_t n_shards, uint32_t nq, uint32_t* __restrict__ flag) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    uint32_t bad = 0;
    for (uint32_t g = 0; g < n_shards; ++g) if (cnt_g[(size_t)g * nq + q] == 0xFFFFFFFFu) { bad = 1; cnt_g[(size_t)g * nq + q] = 0u; }
    flag[q] = bad;
}
hipError_t launch_shard_scrub_counts(hipStream_t st, uint32_t* cnt_g, uint32_t n_shards, uint32_t nq, uint32_t* flag) {
    hipLaunchKernelGGL(shard_scrub_counts_kernel, dim3((nq + 255) / 256), dim3(256), 0, st, cnt_g, n_shards, nq, flag);
    return hipGetLastError();
}
__global__ __launch_bounds__(256) void shard_mark_kernel(const uint32_t* __restrict__ flag, uint32_t nq, uint32_t* __restrict__ out_counts) {   // flagged queries: the marker the unsharded path uses
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < nq && flag[q]) out_counts[q] = 0xFFFFFFFFu;
}
hipError_t launch_shard_mark(hipStream_t st, const uint32_t* flag, uint32_t nq, uint32_t* out_counts) {
    hipLaunchKernelGGL(shard_mark_kernel, dim3((nq + 255) / 256), dim3(256), 0, st, flag, nq, out_counts);
    return hipGetLastError();
}
__global__ __launch_bounds__(256) void shard_fill_i32_kernel(int* __restrict__ dst, int v, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = v;
}
hipError_t launch_shard_fill_i32(hipStream_t st, int* dst, int v, size_t n) {
    if (n) hipLaunchKernelGGL(shard_fill_i32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dst, v, n);
    return hipGetLastError();
}

// The last step of a sharded batch: the G per-shard top-n lists of a query -> its global top-n by (score desc, item id asc).  An item's whole score lives on its
// owner, so no item appears twice; every list is already in that order.  LPQ lanes per query (the power of two >= G), lane = list: a G-way merge by selection --
// every lane holds the head of its list (and the entry behind it, requested when the head moved up: the loads stay off the critical path), the group finds the best head
// with log2(LPQ) exchange steps, the winner writes it out and moves on; n rounds at most.  64 / LPQ queries per wave, ~70 instructions per query at G = 8.
// (Rounds 2-3: one wave per query, every one of the G n entries found its rank with a binary search in each other list -- 1.04 ms per 131 072 queries at G = 8, as much as
// half the back end; this form: see profiles/r04_shard_nb_rank_time.txt.)  part = G blocks of block_bytes: ids [nq * n] u64 | scores [nq * n] f64 | counts [nq] u32.
template <int LPQ>
__global__ __launch_bounds__(256) void shard_merge_topn_kernel(const char* __restrict__ part, size_t block_bytes, uint32_t G, uint32_t nq, uint32_t n,
                                                               uint64_t* __restrict__ out_ids, double* __restrict__ out_scores, uint32_t* __restrict__ out_counts) {
    constexpr uint32_t QPW = 64u / LPQ;
    const uint32_t lane = threadIdx.x & 63u, sub = lane % LPQ;
    const uint32_t q = (blockIdx.x * 4u + (threadIdx.x >> 6)) * QPW + lane / LPQ;
    const bool live = q < nq && sub < G;
    const uint32_t qc = min(q, nq - 1u), gc = min(sub, G - 1u);
    const uint64_t* ids = reinterpret_cast<const uint64_t*>(part + (size_t)gc * block_bytes) + (size_t)qc * n;
    const double* scs = reinterpret_cast<const double*>(part + (size_t)gc * block_bytes + (size_t)nq * n * 8) + (size_t)qc * n;
    const uint32_t craw = reinterpret_cast<const uint32_t*>(part + (size_t)gc * block_bytes + (size_t)nq * n * 16)[qc];
    uint32_t bad = live && craw == 0xFFFFFFFFu ? 1u : 0u;
    const uint32_t cnt = live && craw != 0xFFFFFFFFu ? min(craw, n) : 0u;
    uint32_t total = cnt;
#pragma unroll
    for (int d = 1; d < LPQ; d <<= 1) { total += __shfl_xor(total, d, LPQ); bad |= __shfl_xor(bad, d, LPQ); }
    const uint32_t keep = bad ? 0u : min(total, n);
    // head and the entry behind it (an exhausted list: -inf with the largest id, never the winner while a live entry is left)
    const double NEG = -__builtin_huge_val();
    double s0 = cnt > 0u ? scs[0] : NEG, s1 = cnt > 1u ? scs[1] : NEG;
    unsigned long long i0 = cnt > 0u ? ids[0] : ~0ull, i1 = cnt > 1u ? ids[1] : ~0ull;
    uint32_t pos = 0;
    uint32_t rounds = keep;
#pragma unroll
    for (int d = LPQ; d < 64; d <<= 1) rounds = max(rounds, (uint32_t)__shfl_xor((int)rounds, d));   // (the wave's longest query)
    for (uint32_t r = 0; r < rounds; ++r) {
        double bs = s0; unsigned long long bi = i0; uint32_t bl = sub;
#pragma unroll
        for (int d = 1; d < LPQ; d <<= 1) {
            const double os = __shfl_xor(bs, d, LPQ); const unsigned long long oi = __shfl_xor(bi, d, LPQ); const uint32_t ol = (uint32_t)__shfl_xor((int)bl, d, LPQ);
            const bool other = os > bs || (os == bs && oi < bi);
            bs = other ? os : bs; bi = other ? oi : bi; bl = other ? ol : bl;
        }
        if (r < keep && bl == sub) {   // (one lane of the group)
            out_ids[(size_t)q * n + r] = i0; out_scores[(size_t)q * n + r] = s0;
            ++pos; s0 = s1; i0 = i1;
            const bool more = pos + 1u < cnt;
            s1 = more ? scs[pos + 1u] : NEG; i1 = more ? ids[pos + 1u] : ~0ull;
        }
    }
    if (q < nq) {
        if (!bad) for (uint32_t r = keep + sub; r < n; r += LPQ) { out_ids[(size_t)q * n + r] = 0ull; out_scores[(size_t)q * n + r] = 0.0; }   // the unused tail of a row reads as 0
        if (sub == 0u) out_counts[q] = bad ? 0xFFFFFFFFu : keep;   // (a query some shard could not serve: the caller sees the marker, as in the unsharded path)
    }
}
// lane <- the value of its partner in step `d` of a butterfly over groups of LPQ lanes.  Up to 16 lanes per group the partners come over the VALU's DPP path (d = 1, 2:
// quad permutations; 4: the mirror of a half row, 8: the mirror of a row -- i <-> 7 - i / 15 - i instead of i ^ d, which a selection does not mind: after the quad steps
// all four lanes of a quad agree); __shfl_xor is ds_bpermute_b32, an LDS round trip per 32 bits: 15 of them per round, 21 rounds -- they, not memory, were the kernel's time.
template <int LPQ> __device__ __forceinline__ uint32_t merge_partner(uint32_t v, int d) {
    if constexpr (LPQ <= 16) {
        switch (d) {
            case 1: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);    // quad_perm [1, 0, 3, 2]
            case 2: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);    // quad_perm [2, 3, 0, 1]
            case 4: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true);   // row_half_mirror
            default: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true);  // row_mirror
        }
    } else return (uint32_t)__shfl_xor((int)v, d, LPQ);
}
template <int LPQ> __device__ __forceinline__ unsigned long long merge_partner64(unsigned long long v, int d) {
    return ((unsigned long long)merge_partner<LPQ>((uint32_t)(v >> 32), d) << 32) | merge_partner<LPQ>((uint32_t)v, d);
}
// Round 5: the same selection with every lane's WHOLE list staged in LDS first (n entries of 16 bytes per lane, one wave per workgroup).  In the form above a lane that wins
// twice in a row waits a memory round trip for its next head, and 64 / LPQ queries share the wave: ~21 dependent trips per wave, 0.20 ms per 131 072 queries at G = 8 for
// 357 MB of input.  Here all loads of a wave are in flight together and a round is an LDS read.
template <int LPQ>
__global__ __launch_bounds__(64) void shard_merge_topn_lds_kernel(const char* __restrict__ part, size_t block_bytes, uint32_t G, uint32_t nq, uint32_t n,
                                                                  uint64_t* __restrict__ out_ids, double* __restrict__ out_scores, uint32_t* __restrict__ out_counts) {
    extern __shared__ __attribute__((aligned(16))) char merge_sm[];
    constexpr uint32_t QPW = 64u / LPQ;
    const uint32_t lane = threadIdx.x, sub = lane % LPQ;
    const uint32_t q = blockIdx.x * QPW + lane / LPQ;
    const bool live = q < nq && sub < G;
    const uint32_t qc = min(q, nq - 1u), gc = min(sub, G - 1u);
    const uint32_t craw = reinterpret_cast<const uint32_t*>(part + (size_t)gc * block_bytes + (size_t)nq * n * 16)[qc];
    // staging: the wave's 64 / LPQ queries are consecutive, so a shard's ids (and scores) for them are ONE contiguous run of QPW n 8-byte words: read by consecutive lanes,
    // 2 G runs per wave.  (A first build had every lane read its own list, 8 bytes at a time: 88 lines touched per load instruction, 21.5 KB of strided footprint per wave
    // against a 32 KB L1 shared by seven waves -- the lines came from the L2 again and again: 0.19 ms for 357 MB.)
    unsigned long long* const lw = reinterpret_cast<unsigned long long*>(merge_sm);
    const uint32_t q0 = blockIdx.x * QPW, words = QPW * n, staged = 2u * G * words;
    const size_t avail = (size_t)(nq - min(q0, nq)) * n;   // words of a run that exist (the batch's last wave)
    for (uint32_t e0 = 0; e0 < staged; e0 += 8u * 64u) {
        unsigned long long v[8];
#pragma unroll
        for (uint32_t t = 0; t < 8u; ++t) {
            const uint32_t e = e0 + t * 64u + lane, r = e / words, w = e - r * words;
            v[t] = 0ull;
            if (e < staged && w < avail) v[t] = reinterpret_cast<const unsigned long long*>(part + (size_t)(r >> 1) * block_bytes + (size_t)(r & 1u) * nq * n * 8)[(size_t)q0 * n + w];
        }
#pragma unroll
        for (uint32_t t = 0; t < 8u; ++t) { const uint32_t e = e0 + t * 64u + lane; if (e < staged) lw[e] = v[t]; }
    }
    __syncthreads();   // (one wave: orders the LDS writes before the other lanes' reads)
    const unsigned long long* const li = lw + (size_t)(2u * gc) * words + (size_t)(lane / LPQ) * n;
    const double* const ls = reinterpret_cast<const double*>(lw + (size_t)(2u * gc + 1u) * words + (size_t)(lane / LPQ) * n);
    const double NEG = -__builtin_huge_val();
    uint32_t bad = live && craw == 0xFFFFFFFFu ? 1u : 0u;
    const uint32_t cnt = live && craw != 0xFFFFFFFFu ? min(craw, n) : 0u;
    uint32_t total = cnt;
#pragma unroll
    for (int d = 1; d < LPQ; d <<= 1) { total += merge_partner<LPQ>(total, d); bad |= merge_partner<LPQ>(bad, d); }
    const uint32_t keep = bad ? 0u : min(total, n);
    double s0 = cnt > 0u ? ls[0] : NEG; unsigned long long i0 = cnt > 0u ? li[0] : ~0ull;
    uint32_t pos = 0, rounds = keep;
#pragma unroll
    for (int d = LPQ; d < 64; d <<= 1) rounds = max(rounds, (uint32_t)__shfl_xor((int)rounds, d));   // (the wave's longest query)
    for (uint32_t r = 0; r < rounds; ++r) {
        double bs = s0; unsigned long long bi = i0; uint32_t bl = sub;
#pragma unroll
        for (int d = 1; d < LPQ; d <<= 1) {
            const double os = __longlong_as_double((long long)merge_partner64<LPQ>((unsigned long long)__double_as_longlong(bs), d)); const unsigned long long oi = merge_partner64<LPQ>(bi, d);
            const uint32_t ol = merge_partner<LPQ>(bl, d);
            const bool other = os > bs || (os == bs && oi < bi);
            bs = other ? os : bs; bi = other ? oi : bi; bl = other ? ol : bl;
        }
        if (r < keep && bl == sub) {   // (one lane of the group)
            out_ids[(size_t)q * n + r] = i0; out_scores[(size_t)q * n + r] = s0;
            ++pos;
            const bool more = pos < cnt;
            s0 = more ? ls[min(pos, n - 1u)] : NEG; i0 = more ? li[min(pos, n - 1u)] : ~0ull;
        }
    }
    // (tried: the result rows staged in LDS and copied out in one run -- 0.180 ms against 0.153: one wave less per CU costs more than the scattered 8-byte stores)
    if (q < nq) {
        if (!bad) for (uint32_t r = keep + sub; r < n; r += LPQ) { out_ids[(size_t)q * n + r] = 0ull; out_scores[(size_t)q * n + r] = 0.0; }   // the unused tail of a row reads as 0
        if (sub == 0u) out_counts[q] = bad ? 0xFFFFFFFFu : keep;   // (a query some shard could not serve: the caller sees the marker, as in the unsharded path)
    }
}
hipError_t launch_shard_merge_topn(hipStream_t st, const char* part, size_t block_bytes, uint32_t n_shards, uint32_t nq, uint32_t how_many, uint64_t* out_ids, double* out_scores,
                                   uint32_t* out_counts) {
    if (n_shards == 0 || n_shards > 64) return hipErrorInvalidValue;
    const uint32_t lpq = n_shards <= 2 ? 2u : n_shards <= 4 ? 4u : n_shards <= 8 ? 8u : n_shards <= 16 ? 16u : n_shards <= 32 ? 32u : 64u;
    static const bool old_form = getenv("SRN_MERGE_OLD") != nullptr;   // (experiments: the round-4 form)
    if (!old_form && how_many >= 1u && how_many <= 40u) {   // (<= 40 KB of LDS per wave: three waves per CU and more)
        const dim3 grid((nq + 64u / lpq - 1) / (64u / lpq)), block(64);
        const size_t lds = (size_t)2u * n_shards * (64u / lpq) * how_many * 8;   // the 2 G input runs
        switch (lpq) {
            case 2: hipLaunchKernelGGL(shard_merge_topn_lds_kernel<2>, grid, block, lds, st, part, block_bytes, n_shards, nq, how_many, out_ids, out_scores, out_counts); break;
            case 4: hipLaunchKernelGGL(shard_merge_topn_lds_kernel<4>, grid, block, lds, st, part, block_bytes, n_shards, nq, how_many, out_ids, out_scores, out_counts); break;
            case 8: hipLaunchKernelGGL(shard_merge_topn_lds_kernel<8>, grid, block, lds, st, part, block_bytes, n_shards, nq, how_many, out_ids, out_scores, out_counts); break;
            case 16: hipLaunchKernelGGL(shard_merge_topn_lds_kernel<16>, grid, block, lds, st, part, block_bytes, n_shards, nq, how_many, out_ids, out_scores, out_counts); break;
            case 32: hipLaunchKernelGGL(shard_merge_topn_lds_kernel<32>, grid, block, lds, st, part, block_bytes, n_shards, nq, how_many, out_ids, out_scores, out_counts); break;
            default: hipLaunchKernelGGL(shard_merge_topn_lds_kernel<64>, grid, block, lds, st, part, block_bytes, n_shards, nq, how_many, out_ids, out_scores, out_counts); break;
        }
        return hipGetLastError();
    }
    const uint32_t per_block = 4u * (64u / lpq);
    const dim3 grid((nq + per_block - 1) / per_block), block(256);
    switch (lpq) {
        case 2: hipLaunchKernelGGL(shard_merge_topn_kernel<2>, grid, block, 0, st, part, block_bytes, n_shards, nq, how_many, out_ids, out_scores, out_counts); break;
        case 4: hipLaunchKernelGGL(shard_merge_topn_kernel<4>, grid, block, 0, st, part, block_bytes, n_shards, nq, how_many, out_ids, out_scores, out_counts); break;
        case 8: hipLaunchKernelGGL(shard_merge_topn_kernel<8>, grid, block, 0, st, part, block_bytes, n_shards, nq, how_many, out_ids, out_scores, out_counts); break;
        case 16: hipLaunchKernelGGL(shard_merge_topn_kernel<16>, grid, block, 0, st, part, block_bytes, n_shards, nq, how_many, out_ids, out_scores, out_counts); break;
        case 32: hipLaunchKernelGGL(shard_merge_topn_kernel<32>, grid, block, 0, st, part, block_bytes, n_shards, nq, how_many, out_ids, out_scores, out_counts); break;
        default: hipLaunchKernelGGL(shard_merge_topn_kernel<64>, grid, block, 0, st, part, block_bytes, n_shards, nq, how_many, out_ids, out_scores, out_counts); break;
    }
    return hipGetLastError();
}

}  // namespace srn

// =====================================================================================
// VMIS-kNN predict_next on gfx950 (MI355X).  One workgroup per evolving session; the whole of
// find_neighbors (src/vmisknn/vmis_index.rs:325-415) and predict (src/vmisknn/mod.rs:118-215)
// runs inside one kernel with all per-query state in LDS:
//
//   phase 0  translate the evolving items (u64 -> dense idx), de-duplicate, position weights
//   phase 1  walk the <= U posting lists (coalesced, rank-descending), insert-or-add
//            (rank, weight) into an LDS open-addressing table  -> integer similarity numerators
//   phase 2  m-cut: radix-select the m-th largest recency rank;  k-cut: radix-select the k-th
//            largest (numerator, rank) composite;  compact the neighbours
//   phase 3  walk the neighbours' rows (sub-wave groups of lanes per row), first-match position
//            against the FULL row, insert-or-add w10*num into the LDS item table
//   phase 4  score = idf_eff * acc / (10 U) in f64, business rules, filtered top-n with an
//            in-LDS bitonic sort on (score desc, item idx asc)
//
// Everything up to the final f64 multiply/divide is integer arithmetic, so (neighbours,
// numerators, accumulators) are bit-identical to the canonical CPU oracle by construction.
// No MFMA: this is sparse gather/scatter, bounded by HBM/L2 traffic and LDS atomics.
//
// Queries whose candidate or item sets do not fit the LDS tables are queued on a device-side
// retry list and served by the same kernel instantiated with its tables in a global scratch
// arena (GLOBAL_TABLES = true) -- still on the GPU, never on the CPU.
// =====================================================================================
#include <hip/hip_runtime.h>

#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

#include "srn_internal.h"

namespace srn {

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return fail(SRN_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

static constexpr uint32_t EMPTY32 = 0xFFFFFFFFu;
static constexpr int CAND_CAP = 1024;       // candidate buffer of the final top-n (entries)
static constexpr int MISC_WORDS = 64;       // scalar words at the head of LDS
static constexpr int LMAX = SRN_MAX_SESSION_LEN + 1;

// launch-time geometry, identical for every block of a launch
struct KernelCfg {
    uint32_t sess_slots, sess_cap;   // session table: slots, max distinct before overflow
    uint32_t item_slots, item_cap;   // item table
    uint32_t num_bits;               // low bits of a session slot that hold the numerator
    uint32_t region_a_bytes, region_b_bytes;
    uint32_t rows_lanes;             // lanes per neighbour row in phase 3 (power of two <= 64)
};

// LDS scalar slots
enum { S_CNT = 0, S_OVF, S_XLO, S_RMAX, S_U, S_P, S_SUMW, S_SELD, S_SELR, S_NB, S_ICNT, S_CCNT, S_I, S_HAVE_T, S_TIDX,
       S_ERR, S_TKEY_LO, S_TKEY_HI, S_QCUR };

template <typename T> struct SlotTraits;
template <> struct SlotTraits<uint32_t> { static constexpr uint32_t EMPTY = 0xFFFFFFFFu; };
template <> struct SlotTraits<unsigned long long> { static constexpr unsigned long long EMPTY = ~0ull; };

__device__ __forceinline__ uint64_t dev_mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33; return x;
}
__device__ __forceinline__ uint32_t hash_slot(uint32_t key, uint32_t slots) { return __umulhi(key * 0x9E3779B1u, slots); }
__device__ __forceinline__ int bits_for(uint32_t v) { return v ? 32 - __clz((int)v) : 0; }

__device__ __forceinline__ uint64_t score_key(double s) {   // order-preserving f64 -> u64
    uint64_t b = (uint64_t)__double_as_longlong(s);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_score(uint64_t k) {
    uint64_t b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    return __longlong_as_double((long long)b);
}
// passes_business_rules, src/vmisknn/mod.rs:162-182; attribute byte SRN_ATTR_NONE = None
__device__ __forceinline__ bool business_ok(uint32_t cur, uint32_t reco) {
    if (reco == SRN_ATTR_NONE) return false;
    if (reco & SRN_ATTR_FOR_SALE) {
        if (reco & SRN_ATTR_ADULT) return cur != SRN_ATTR_NONE && (cur & SRN_ATTR_ADULT);
        return true;
    }
    return false;
}

// -------------------------------------------------------------------------------------
// r-th largest key among the valid entries of `n` slots (keys distinct, r >= 1, r <= #valid):
// MSD radix select, 8 bits per pass, histogram in LDS.  keyfn(i, &key) -> valid.
// -------------------------------------------------------------------------------------
template <int BLOCK, typename KeyT, typename F>
__device__ KeyT block_select_desc(F keyfn, uint32_t n, int nbits, uint32_t r, uint32_t* hist, volatile uint32_t* misc) {
    const int tid = threadIdx.x;
    KeyT prefix = 0;
    uint32_t remain = r;
    for (int shift = ((nbits + 7) / 8) * 8 - 8; shift >= 0; shift -= 8) {
        for (int i = tid; i < 256; i += BLOCK) hist[i] = 0;
        __syncthreads();
        const KeyT hi_mask = (shift + 8 >= (int)(8 * sizeof(KeyT))) ? (KeyT)0 : (KeyT)((~(KeyT)0) << (shift + 8));
        for (uint32_t i = tid; i < n; i += BLOCK) {
            KeyT key;
            if (keyfn(i, key) && (key & hi_mask) == prefix) atomicAdd(&hist[(uint32_t)(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid < 64) {
            const uint32_t c0 = hist[4 * tid], c1 = hist[4 * tid + 1], c2 = hist[4 * tid + 2], c3 = hist[4 * tid + 3];
            const uint32_t s = c0 + c1 + c2 + c3;
            uint32_t inc = s;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_down(inc, d, 64); if (tid + d < 64) inc += t; }
            const uint32_t a3 = inc - s, a2 = a3 + c3, a1 = a2 + c2, a0 = a1 + c1;
            if (a3 < remain && remain <= a3 + c3) { misc[S_SELD] = 4 * tid + 3; misc[S_SELR] = remain - a3; }
            if (a2 < remain && remain <= a2 + c2) { misc[S_SELD] = 4 * tid + 2; misc[S_SELR] = remain - a2; }
            if (a1 < remain && remain <= a1 + c1) { misc[S_SELD] = 4 * tid + 1; misc[S_SELR] = remain - a1; }
            if (a0 < remain && remain <= a0 + c0) { misc[S_SELD] = 4 * tid + 0; misc[S_SELR] = remain - a0; }
        }
        __syncthreads();
        prefix |= (KeyT)misc[S_SELD] << shift;
        remain = misc[S_SELR];
    }
    return prefix;
}

// Barrier between phases.  With the tables in global memory (retry pass) the CU's vector L1 may hold
// lines older than the L2 atomics of the previous phase: write back + invalidate, then re-converge.
template <bool GLOBAL_TABLES> __device__ __forceinline__ void phase_sync() {
    __syncthreads();
    if (GLOBAL_TABLES) { __threadfence(); __syncthreads(); }
}

// in-LDS bitonic sort of n (power of two) candidates, best first: (key desc, idx asc)
template <int BLOCK>
__device__ void block_sort_candidates(uint64_t* skey, uint32_t* sidx, uint32_t n) {
    const uint32_t tid = threadIdx.x;
    for (uint32_t kk = 2; kk <= n; kk <<= 1) {
        for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
            for (uint32_t t = tid; t < (n >> 1); t += BLOCK) {
                const uint32_t lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
                const uint64_t ka = skey[lo], kb = skey[hi];
                const uint32_t ia = sidx[lo], ib = sidx[hi];
                const bool a_better = ka > kb || (ka == kb && ia < ib);
                const bool descending = (lo & kk) == 0;
                if (a_better != descending) { skey[lo] = kb; skey[hi] = ka; sidx[lo] = ib; sidx[hi] = ia; }
            }
            __syncthreads();
        }
    }
}

template <int BLOCK, typename SlotT, typename OffT, bool GLOBAL_TABLES>
__global__ __launch_bounds__(BLOCK) void vmis_predict_kernel(DeviceIndex ix, LaunchParams p, KernelCfg c,
                                                             const uint32_t* __restrict__ qlist, const uint32_t* __restrict__ qlist_n,
                                                             uint32_t* retry_list, uint32_t* retry_cnt,
                                                             char* gscratch, unsigned long long gscratch_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    constexpr SlotT SEMPTY = SlotTraits<SlotT>::EMPTY;

    // ---- LDS carve-up (every offset a multiple of 16) ------------------------------------
    volatile uint32_t* misc = (volatile uint32_t*)smem;                       // MISC_WORDS
    uint32_t* hist = (uint32_t*)(smem + MISC_WORDS * 4);                      // 256
    uint64_t* q_raw = (uint64_t*)(smem + MISC_WORDS * 4 + 1024);              // LMAX
    unsigned long long* l_base = (unsigned long long*)(q_raw + LMAX);         // LMAX
    uint32_t* q_idx = (uint32_t*)(l_base + LMAX);                             // LMAX
    uint32_t* l_len = q_idx + LMAX;                                           // LMAX
    char* region_b = (char*)(l_len + LMAX);
    char* region_a = GLOBAL_TABLES ? (gscratch + (size_t)blockIdx.x * gscratch_stride) : (region_b + c.region_b_bytes);

    SlotT* stab = (SlotT*)region_a;                                           // phase 1-2
    uint32_t* ikeys = (uint32_t*)region_a;                                    // phase 3-4
    int* iacc = (int*)(region_a + (size_t)c.item_slots * 4);
    SlotT* nbl = (SlotT*)region_b;                                            // neighbours (phase 2-3)
    uint64_t* ckey = (uint64_t*)region_b;                                     // candidates (phase 4)
    uint32_t* cidx = (uint32_t*)(region_b + CAND_CAP * 8);

    const OffT* __restrict__ row_off = (const OffT*)ix.row_off;
    const uint32_t NB = c.num_bits;
    const SlotT num_mask = ((SlotT)1 << NB) - 1;
    const uint32_t nq_eff = qlist ? *qlist_n : p.nq;

    for (uint32_t qi = blockIdx.x; qi < nq_eff; qi += gridDim.x) {
        const uint32_t q = qlist ? qlist[qi] : qi;
        const uint32_t qb = p.q_off[q];
        const uint32_t L = p.q_off[q + 1] - qb;
        if (L == 0 || L > p.max_len) {   // block-uniform
            if (tid == 0) { p.out_counts[q] = 0xFFFFFFFFu;
                if (p.stats) { for (int i = 0; i < 8; ++i) p.stats[(size_t)q * 8 + i] = 0; p.stats[(size_t)q * 8 + 7] = 2; }
                if (p.nb_cnt) p.nb_cnt[q] = 0; }
            continue;
        }
        // ---- phase 0: reset, translate items ---------------------------------------------
        __syncthreads();   // previous query's LDS reads are done
        if (tid < MISC_WORDS) misc[tid] = 0;
        for (uint32_t i = tid; i < c.sess_slots; i += BLOCK) stab[i] = SEMPTY;
        if (tid < L) q_raw[tid] = p.items_flat[qb + (L - 1 - tid)];   // pos 0 = most recent item
        phase_sync<GLOBAL_TABLES>();
        if (tid < L) {
            const uint32_t pos = tid; const uint64_t raw = q_raw[pos];
            bool first = true;
            for (uint32_t j = 0; j < pos; ++j) first = first && (q_raw[j] != raw);   // Q2: most recent occurrence only
            uint32_t idx = kNone;
            { uint32_t h = (uint32_t)dev_mix64(raw) & ix.id_mask;
              for (;;) { const IdSlot s = ix.id_table[h]; if (s.idx == kNone) break; if (s.key == raw) { idx = s.idx; break; } h = (h + 1) & ix.id_mask; } }
            q_idx[pos] = idx;
            uint32_t len = 0;
            if (first) atomicAdd((uint32_t*)&misc[S_U], 1u);   // Q1: distinct raw ids, known or not
            if (first && idx != kNone) {
                const unsigned long long o0 = ix.post_off[idx], o1 = ix.post_off[idx + 1];
                len = (uint32_t)min((unsigned long long)p.m, o1 - o0);
                l_base[pos] = o0;
                if (len) {
                    atomicMax((uint32_t*)&misc[S_RMAX], ix.post_rank[o0]);
                    if (len >= p.m) atomicMax((uint32_t*)&misc[S_XLO], ix.post_rank[o0 + p.m - 1]);
                    atomicAdd((uint32_t*)&misc[S_P], len);
                    atomicAdd((uint32_t*)&misc[S_SUMW], L - pos);
                }
            }
            l_len[pos] = len;
        }
        __syncthreads();
        const uint32_t x_lo = misc[S_XLO], r_max = misc[S_RMAX], U = misc[S_U];
        const uint32_t cur_idx = q_idx[0];

        // ---- phase 1: posting lists -> session table -------------------------------------
        // Entries below x_lo (the m-th entry of a full list) can never be among the m most
        // recent distinct sessions; lists are rank-descending, so a lane stops at the first one.
        for (uint32_t pos = 0; pos < L; ++pos) {
            const uint32_t len = l_len[pos];
            if (!len) continue;
            const uint32_t* __restrict__ list = ix.post_rank + l_base[pos];
            const SlotT w = (SlotT)(L - pos);
            for (uint32_t i = tid; i < len; i += BLOCK) {
                const uint32_t r = list[i];
                if (r < x_lo) break;
                if (misc[S_CNT] > c.sess_cap) { misc[S_OVF] = 1; break; }
                uint32_t h = hash_slot(r, c.sess_slots);
                bool placed = false;
                for (uint32_t probe = 0; probe < c.sess_slots; ++probe) {
                    SlotT cur = __atomic_load_n(&stab[h], __ATOMIC_RELAXED);
                    if (cur == SEMPTY) {
                        const SlotT old = atomicCAS(&stab[h], SEMPTY, ((SlotT)r << NB) | w);
                        if (old == SEMPTY) { atomicAdd((uint32_t*)&misc[S_CNT], 1u); placed = true; break; }
                        cur = old;
                    }
                    if ((uint32_t)(cur >> NB) == r) { atomicAdd(&stab[h], w); placed = true; break; }
                    h = (h + 1 == c.sess_slots) ? 0 : h + 1;
                }
                if (!placed) { misc[S_OVF] = 1; break; }
            }
        }
        phase_sync<GLOBAL_TABLES>();
        if (misc[S_OVF]) {   // block-uniform: hand the query to the global-table pass
            if (GLOBAL_TABLES) { if (tid == 0) { p.out_counts[q] = 0xFFFFFFFFu; if (p.stats) p.stats[(size_t)q * 8 + 7] = 3; } }
            else if (tid == 0) retry_list[atomicAdd(retry_cnt, 1u)] = q;
            continue;
        }
        const uint32_t Call = misc[S_CNT];
        const uint32_t Cm = min(Call, p.m);

        // ---- phase 2: m-cut, k-cut, compaction -------------------------------------------
        uint32_t tau = x_lo;
        if (Call > p.m) {
            tau = x_lo + block_select_desc<BLOCK, uint32_t>(
                [&](uint32_t i, uint32_t& key) { const SlotT s = stab[i]; key = (uint32_t)(s >> NB) - x_lo; return s != SEMPTY; },
                c.sess_slots, bits_for(r_max - x_lo), p.m, hist, misc);
        }
        const int rbits = bits_for(r_max - tau);
        unsigned long long kappa = 0;   // composite threshold: (num << rbits) | (rank - tau)
        if (Cm > p.k) {
            const int nbits = rbits + bits_for(misc[S_SUMW]);
            auto comp = [&](uint32_t i, auto& key) {
                const SlotT s = stab[i]; const uint32_t r = (uint32_t)(s >> NB);
                key = ((decltype(key + 0))(s & num_mask) << rbits) | (r - tau);
                return s != SEMPTY && r >= tau; };
            if (nbits <= 32) kappa = block_select_desc<BLOCK, uint32_t>(comp, c.sess_slots, nbits, p.k, hist, misc);
            else kappa = block_select_desc<BLOCK, unsigned long long>(comp, c.sess_slots, nbits, p.k, hist, misc);
        }
        for (uint32_t i = tid; i < c.sess_slots; i += BLOCK) {
            const SlotT s = stab[i];
            const uint32_t r = (uint32_t)(s >> NB);
            if (s != SEMPTY && r >= tau && ((((unsigned long long)(s & num_mask)) << rbits) | (r - tau)) >= kappa) {
                const uint32_t at = atomicAdd((uint32_t*)&misc[S_NB], 1u);
                nbl[at] = s;
                if (p.nb_rank) { p.nb_rank[(size_t)q * p.k + at] = r; p.nb_num[(size_t)q * p.k + at] = (uint32_t)(s & num_mask); }
            }
        }
        __syncthreads();
        const uint32_t K = misc[S_NB];
        for (uint32_t i = tid; i < c.item_slots; i += BLOCK) { ikeys[i] = EMPTY32; iacc[i] = 0; }
        phase_sync<GLOBAL_TABLES>();

        // ---- phase 3: neighbour rows -> item table ---------------------------------------
        {
            const uint32_t G = c.rows_lanes, g = tid & (G - 1), per_iter = BLOCK / G;
            for (uint32_t base = 0; base < K; base += per_iter) {
                const uint32_t j = base + tid / G;
                const bool live = j < K;
                uint32_t r = 0, num = 0; unsigned long long o0 = 0, o1 = 0;
                if (live) { const SlotT s = nbl[j]; r = (uint32_t)(s >> NB); num = (uint32_t)(s & num_mask);
                            o0 = row_off[r]; o1 = row_off[r + 1]; }
                uint32_t minpos = 0xFFFFu;
                for (unsigned long long t = o0 + g; t < o1; t += G) {
                    const uint32_t it = ix.row_items[t];
                    for (uint32_t pp = 0; pp < L; ++pp) if (q_idx[pp] == it) { minpos = min(minpos, pp); break; }
                }
                for (uint32_t d = 1; d < G; d <<= 1) minpos = min(minpos, (uint32_t)__shfl_xor((int)minpos, (int)d, 64));
                if (live) {
                    if (minpos == 0xFFFFu) misc[S_ERR] = 1;   // inconsistent index (reference: unwrap panic, mod.rs:138)
                    const int p1 = (int)minpos + 1;
                    const int w = (p1 < 100 ? 10 - p1 : 0) * (int)num;   // 10 * linear_score(pos) * numerator, exact (Q3)
                    if (p.stats && g == 0) atomicAdd((uint32_t*)&misc[S_I], (uint32_t)(o1 - o0));
                    for (unsigned long long t = o0 + g; t < o1; t += G) {
                        const uint32_t it = ix.row_items[t];
                        if (misc[S_ICNT] > c.item_cap) { misc[S_OVF] = 1; break; }
                        uint32_t h = hash_slot(it, c.item_slots);
                        bool placed = false;
                        for (uint32_t probe = 0; probe < c.item_slots; ++probe) {
                            uint32_t cur = __atomic_load_n(&ikeys[h], __ATOMIC_RELAXED);
                            if (cur == EMPTY32) {
                                const uint32_t old = atomicCAS(&ikeys[h], EMPTY32, it);
                                if (old == EMPTY32) { atomicAdd((uint32_t*)&misc[S_ICNT], 1u); cur = it; } else cur = old;
                            }
                            if (cur == it) { atomicAdd(&iacc[h], w); placed = true; break; }
                            h = (h + 1 == c.item_slots) ? 0 : h + 1;
                        }
                        if (!placed) { misc[S_OVF] = 1; break; }
                    }
                }
            }
        }
        phase_sync<GLOBAL_TABLES>();
        if (misc[S_OVF]) {
            if (GLOBAL_TABLES) { if (tid == 0) { p.out_counts[q] = 0xFFFFFFFFu; if (p.stats) p.stats[(size_t)q * 8 + 7] = 3; } }
            else if (tid == 0) retry_list[atomicAdd(retry_cnt, 1u)] = q;
            continue;
        }

        // ---- phase 4: scores, business rules, top-n --------------------------------------
        const double denom = (double)(10u * U);
        const bool business = (p.flags & SRN_FLAG_BUSINESS_LOGIC) != 0;
        const uint32_t cur_attr = (business && cur_idx != kNone) ? ix.attr[cur_idx] : SRN_ATTR_NONE;
        const uint32_t n_out = p.how_many;
        for (uint32_t base = 0; base < c.item_slots; base += BLOCK) {
            const uint32_t i = base + tid;
            if (i < c.item_slots) {
                const uint32_t it = ikeys[i];
                if (it != EMPTY32 && it != cur_idx && (!business || business_ok(cur_attr, ix.attr[it]))) {   // Q6 + rules
                    const double idf = ix.idf[it];
                    const double sc = (idf > 0.0 ? idf : 1.0) * (double)iacc[i] / denom;
                    const uint64_t sk = score_key(sc);
                    bool take = true;
                    if (misc[S_HAVE_T]) { const uint64_t tk = ((uint64_t)misc[S_TKEY_HI] << 32) | misc[S_TKEY_LO];
                                          take = sk > tk || (sk == tk && it < misc[S_TIDX]); }
                    if (take) { const uint32_t at = atomicAdd((uint32_t*)&misc[S_CCNT], 1u); ckey[at] = sk; cidx[at] = it; }
                }
            }
            __syncthreads();
            const uint32_t cnt = misc[S_CCNT];
            const bool last = base + BLOCK >= c.item_slots;
            if (cnt > CAND_CAP - BLOCK || (last && cnt > 1)) {   // block-uniform
                uint32_t n2 = 2; while (n2 < cnt) n2 <<= 1;
                for (uint32_t t = cnt + tid; t < n2; t += BLOCK) { ckey[t] = 0; cidx[t] = EMPTY32; }
                __syncthreads();
                block_sort_candidates<BLOCK>(ckey, cidx, n2);
                if (tid == 0 && cnt >= n_out) {
                    misc[S_CCNT] = n_out; misc[S_HAVE_T] = 1; misc[S_TIDX] = cidx[n_out - 1];
                    misc[S_TKEY_LO] = (uint32_t)ckey[n_out - 1]; misc[S_TKEY_HI] = (uint32_t)(ckey[n_out - 1] >> 32);
                }
                __syncthreads();
            }
        }
        const uint32_t H = min(misc[S_CCNT], n_out);
        if (tid < H) {
            p.out_ids[(size_t)q * n_out + tid] = ix.item_id[cidx[tid]];
            p.out_scores[(size_t)q * n_out + tid] = key_score(ckey[tid]);
        }
        if (tid == 0) {
            p.out_counts[q] = H;
            if (p.nb_cnt) p.nb_cnt[q] = K;
            if (p.stats) { uint32_t* st = p.stats + (size_t)q * 8;
                st[0] = misc[S_P]; st[1] = Cm; st[2] = K; st[3] = misc[S_I]; st[4] = misc[S_ICNT]; st[5] = H; st[6] = L;
                st[7] = misc[S_ERR] ? 4u : (GLOBAL_TABLES ? 1u : 0u); }
        }
    }
}

// =====================================================================================
// host side: device state, workspaces, launch
// =====================================================================================
struct Workspace {
    hipStream_t stream = nullptr;   // own stream for host-pointer calls
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    bool timed = false; uint32_t last_retry = 0;
    // device scratch
    uint32_t* retry_list = nullptr; size_t retry_cap = 0; uint32_t* retry_cnt = nullptr;
    char* gscratch = nullptr; size_t gscratch_bytes = 0;
    // staging for host-pointer calls
    char* stage = nullptr; size_t stage_bytes = 0;
    uint32_t* h_retry = nullptr;   // pinned
};

struct DeviceState {
    int device = 0;
    std::vector<void*> allocs; uint64_t bytes = 0;
    DeviceIndex di{};
    uint8_t* d_attr = nullptr;
    bool off64 = false;
    int n_cu = 256;
    int lds_per_block_max = 65536;
    std::mutex mu; std::vector<Workspace*> free_ws; std::vector<Workspace*> all_ws;
    std::vector<std::pair<void*, Workspace*>> stream_ws;   // device-pointer calls: one workspace per user stream
    Workspace* last_ws = nullptr;   // for srn_last_kernel_ms (single-threaded measurement use)
};

namespace {
template <typename T> const T* upload(DeviceState* d, const std::vector<T>& v, bool& ok) {
    void* p = nullptr; const size_t n = std::max<size_t>(v.size() * sizeof(T), 16);
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
    DeviceState* d = new DeviceState(); d->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) { d->n_cu = prop.multiProcessorCount; d->lds_per_block_max = (int)prop.sharedMemPerBlock; }
    bool ok = true;
    d->di.id_table = upload(d, ix.id_table, ok); d->di.id_mask = ix.id_mask;
    d->di.item_id = upload(d, ix.item_id, ok); d->di.idf = upload(d, ix.idf, ok);
    d->d_attr = (uint8_t*)upload(d, ix.attr, ok); d->di.attr = d->d_attr;
    d->di.post_off = upload(d, ix.post_off, ok); d->di.post_rank = upload(d, ix.post_rank, ok);
    d->off64 = ix.nnz_rows >= 0xFFFFFFFFull;
    if (d->off64) d->di.row_off = upload(d, ix.row_off, ok);
    else { std::vector<uint32_t> o32(ix.row_off.begin(), ix.row_off.end()); d->di.row_off = upload(d, o32, ok); }
    d->di.row_items = upload(d, ix.row_items, ok);
    d->di.n_items = (uint32_t)ix.n_items; d->di.n_kept = (uint32_t)ix.n_kept;
    if (!ok) { device_release(d); return nullptr; }
    return d;
}

static void ws_free(Workspace* w) {
    if (!w) return;
    if (w->retry_list) hipFree(w->retry_list);
    if (w->retry_cnt) hipFree(w->retry_cnt);
    if (w->gscratch) hipFree(w->gscratch);
    if (w->stage) hipFree(w->stage);
    if (w->h_retry) hipHostFree(w->h_retry);
    for (auto& e : w->ev) if (e) hipEventDestroy(e);
    if (w->stream) hipStreamDestroy(w->stream);
    delete w;
}

void device_release(DeviceState* d) {
    if (!d) return;
    hipSetDevice(d->device);
    for (Workspace* w : d->all_ws) ws_free(w);
    for (void* p : d->allocs) hipFree(p);
    delete d;
}

int device_update_attr(DeviceState* d, const FlatIndex& ix) {
    HIP_TRY(hipSetDevice(d->device));
    HIP_TRY(hipMemcpy(d->d_attr, ix.attr.data(), ix.attr.size(), hipMemcpyHostToDevice));
    return SRN_OK;
}
uint64_t device_bytes(const DeviceState* d) { return d ? d->bytes : 0; }
int device_count() { int n = 0; return hipGetDeviceCount(&n) == hipSuccess ? n : 0; }

// Host-pointer calls borrow a workspace for the duration of the (synchronous) call.  Device-pointer
// calls return while their work is still in flight, so their scratch stays bound to the user's stream
// (stream order then serialises its reuse).
static Workspace* ws_acquire(DeviceState* d, bool bind_to_stream, void* user_stream) {
    std::lock_guard<std::mutex> lk(d->mu);
    if (bind_to_stream) for (auto& sw : d->stream_ws) if (sw.first == user_stream) return sw.second;
    if (!bind_to_stream && !d->free_ws.empty()) { Workspace* w = d->free_ws.back(); d->free_ws.pop_back(); return w; }
    Workspace* w = new Workspace();
    if (hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking) != hipSuccess) { delete w; return nullptr; }
    for (auto& e : w->ev) if (hipEventCreate(&e) != hipSuccess) { ws_free(w); return nullptr; }
    if (hipMalloc((void**)&w->retry_cnt, 16) != hipSuccess || hipHostMalloc((void**)&w->h_retry, 16) != hipSuccess) { ws_free(w); return nullptr; }
    d->all_ws.push_back(w);
    if (bind_to_stream) d->stream_ws.emplace_back(user_stream, w);
    return w;
}
static void ws_release(DeviceState* d, Workspace* w, bool bound) {
    std::lock_guard<std::mutex> lk(d->mu);
    if (!bound) d->free_ws.push_back(w);
    d->last_ws = w;
}

static int ensure(char** p, size_t* have, size_t need) {
    if (*have >= need) return SRN_OK;
    if (*p) HIP_TRY(hipFree(*p));
    *p = nullptr; *have = 0;
    need = need + need / 4 + 256;
    HIP_TRY(hipMalloc((void**)p, need));
    *have = need; return SRN_OK;
}

static inline uint32_t round_up(uint32_t v, uint32_t a) { return (v + a - 1) / a * a; }
static inline int bits_host(uint64_t v) { int b = 0; while (v) { ++b; v >>= 1; } return b; }

template <int BLOCK, bool GLOBAL_TABLES>
static hipError_t launch_variant(bool slot64, bool off64, dim3 grid, size_t lds, hipStream_t st, const DeviceIndex& di,
                                 const LaunchParams& p, const KernelCfg& c, const uint32_t* qlist, const uint32_t* qn,
                                 uint32_t* retry_list, uint32_t* retry_cnt, char* gs, unsigned long long gstride) {
#define SRN_LAUNCH(SLOT, OFF)                                                                                             \
    do {                                                                                                                  \
        auto kern = vmis_predict_kernel<BLOCK, SLOT, OFF, GLOBAL_TABLES>;                                                 \
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);     \
        if (e != hipSuccess) return e;                                                                                    \
        hipLaunchKernelGGL(kern, grid, dim3(BLOCK), lds, st, di, p, c, qlist, qn, retry_list, retry_cnt, gs, gstride);    \
        return hipGetLastError();                                                                                         \
    } while (0)
    if (!slot64 && !off64) SRN_LAUNCH(uint32_t, uint32_t);
    if (!slot64 && off64) SRN_LAUNCH(uint32_t, unsigned long long);
    if (slot64 && !off64) SRN_LAUNCH(unsigned long long, uint32_t);
    SRN_LAUNCH(unsigned long long, unsigned long long);
#undef SRN_LAUNCH
}

static constexpr int kBlock = 512;
static constexpr uint32_t kFixedLds = MISC_WORDS * 4 + 1024 + LMAX * (8 + 8 + 4 + 4);

int device_predict(DeviceState* d, const FlatIndex& ix, const LaunchParams& p_in, bool on_device, void* user_stream,
                   const uint64_t* h_items, const uint32_t* h_qoff, uint64_t* h_ids, double* h_scores, uint32_t* h_counts,
                   uint32_t* h_stats, uint32_t* h_nb_rank, uint32_t* h_nb_num, uint32_t* h_nb_cnt) {
    HIP_TRY(hipSetDevice(d->device));
    LaunchParams p = p_in;
    if (p.nq == 0) return SRN_OK;
    Workspace* w = ws_acquire(d, on_device, user_stream);
    if (!w) return fail(SRN_EHIP, "cannot create HIP stream / events");
    struct Rel { DeviceState* d; Workspace* w; bool b; ~Rel() { ws_release(d, w, b); } } rel{d, w, on_device};
    hipStream_t st = on_device ? (hipStream_t)user_stream : w->stream;

    // ---- geometry ----------------------------------------------------------------------
    const uint64_t Lmax = p.max_len;
    const int num_bits = std::max(1, bits_host(Lmax * (Lmax + 1) / 2));
    const int rank_bits = std::max(1, bits_host(ix.n_kept ? ix.n_kept - 1 : 0));
    const bool slot64 = rank_bits + num_bits > 32 || ix.n_kept >= 0xFFFFFFF0ull;
    const uint32_t slot_bytes = slot64 ? 8 : 4;
    KernelCfg c{};
    c.num_bits = (uint32_t)num_bits;
    c.region_b_bytes = round_up(std::max<uint32_t>(p.k * slot_bytes, CAND_CAP * 12), 16);
    const uint32_t lds_budget = 80 * 1024 - 256;   // two 512-thread blocks per CU
    if (kFixedLds + c.region_b_bytes + 16 * 1024 > (uint32_t)std::min(d->lds_per_block_max, 160 * 1024))
        return fail(SRN_ERANGE, "k too large for the LDS neighbour list");
    const uint32_t total_budget = std::max(lds_budget, kFixedLds + c.region_b_bytes + 16 * 1024);
    c.region_a_bytes = (total_budget - kFixedLds - c.region_b_bytes) / 16 * 16;
    // what the query set could need at most
    const uint64_t m_eff = std::min<uint64_t>(p.m, ix.m_index);
    const uint64_t need_sess = std::min<uint64_t>(Lmax * m_eff, ix.n_kept);
    const uint64_t need_item = std::min<uint64_t>((uint64_t)p.k * ix.max_row_len, ix.n_items);
    c.sess_slots = (uint32_t)std::min<uint64_t>(c.region_a_bytes / slot_bytes, need_sess * 3 / 2 + 64);
    c.item_slots = (uint32_t)std::min<uint64_t>(c.region_a_bytes / 8, need_item * 3 / 2 + 64);
    c.sess_cap = (uint32_t)std::min<uint64_t>(need_sess, (uint64_t)c.sess_slots * 85 / 100);
    c.item_cap = (uint32_t)std::min<uint64_t>(need_item, (uint64_t)c.item_slots * 85 / 100);
    c.rows_lanes = 4;
    const size_t lds = kFixedLds + c.region_b_bytes + c.region_a_bytes;
    const bool may_overflow = c.sess_cap < need_sess || c.item_cap < need_item;

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
    uint64_t g_stride = 0; int retry_blocks = 0;
    KernelCfg cg = c;
    if (may_overflow) {
        if (w->retry_cap < p.nq) { if (w->retry_list) HIP_TRY(hipFree(w->retry_list)); w->retry_list = nullptr; w->retry_cap = 0;
            HIP_TRY(hipMalloc((void**)&w->retry_list, (size_t)p.nq * 4 + 64)); w->retry_cap = p.nq; }
        cg.sess_slots = (uint32_t)std::min<uint64_t>(0xFFFFFF00ull, need_sess * 2 + 64); cg.sess_cap = (uint32_t)need_sess + 1;
        cg.item_slots = (uint32_t)std::min<uint64_t>(0xFFFFFF00ull, need_item * 2 + 64); cg.item_cap = (uint32_t)need_item + 1;
        g_stride = std::max<uint64_t>((uint64_t)cg.sess_slots * slot_bytes, (uint64_t)cg.item_slots * 8);
        g_stride = (g_stride + 255) / 256 * 256;
        retry_blocks = (int)std::max<uint64_t>(1, std::min<uint64_t>(2 * (uint64_t)d->n_cu, (4ull << 30) / g_stride));
        int rc = ensure(&w->gscratch, &w->gscratch_bytes, g_stride * retry_blocks); if (rc) return rc;
        cg.region_a_bytes = 0;
        HIP_TRY(hipMemsetAsync(w->retry_cnt, 0, 4, st));
    }

    // ---- launches ----------------------------------------------------------------------
    const uint32_t blocks_per_cu = std::max<uint32_t>(1, (160 * 1024) / (uint32_t)lds);
    const uint32_t grid = (uint32_t)std::min<uint64_t>(p.nq, (uint64_t)d->n_cu * blocks_per_cu * 4);
    HIP_TRY(hipEventRecord(w->ev[0], st));
    HIP_TRY((launch_variant<kBlock, false>(slot64, d->off64, dim3(grid), lds, st, d->di, p, c, nullptr, nullptr,
                                            w->retry_list, w->retry_cnt, nullptr, 0)));
    HIP_TRY(hipEventRecord(w->ev[1], st));
    if (may_overflow) {
        const size_t lds_g = kFixedLds + cg.region_b_bytes;
        HIP_TRY((launch_variant<kBlock, true>(slot64, d->off64, dim3(retry_blocks), lds_g, st, d->di, p, cg, w->retry_list, w->retry_cnt,
                                               nullptr, nullptr, w->gscratch, g_stride)));
        HIP_TRY(hipMemcpyAsync(w->h_retry, w->retry_cnt, 4, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(hipEventRecord(w->ev[2], st));
    w->timed = true; w->last_retry = may_overflow ? 1 : 0;

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

int device_last_kernel_ms(DeviceState* d, double* ms_main, double* ms_retry, uint32_t* retried) {
    HIP_TRY(hipSetDevice(d->device));
    Workspace* w;
    { std::lock_guard<std::mutex> lk(d->mu); w = d->last_ws; }
    if (!w || !w->timed) return fail(SRN_EINVAL, "no timed predict call yet");
    HIP_TRY(hipEventSynchronize(w->ev[2]));
    float a = 0, b = 0;
    HIP_TRY(hipEventElapsedTime(&a, w->ev[0], w->ev[1]));
    HIP_TRY(hipEventElapsedTime(&b, w->ev[1], w->ev[2]));
    if (ms_main) *ms_main = a;
    if (ms_retry) *ms_retry = b;
    if (retried) *retried = w->last_retry ? *w->h_retry : 0;
    return SRN_OK;
}

}  // namespace srn

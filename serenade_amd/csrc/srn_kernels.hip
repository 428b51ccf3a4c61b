// =====================================================================================
// VMIS-kNN predict_next on gfx950 (MI355X): the kernels.  find_neighbors (src/vmisknn/vmis_index.rs:325-415) and
// predict (src/vmisknn/mod.rs:118-215) for a batch of evolving sessions, one workgroup (8 waves) per session with all
// per-query state in LDS (DESIGN.md section 4 has the full story and the measurements):
//
//   vmis_prep_kernel      one THREAD per session: public id -> dense idx, posting-list bounds, first / m-th rank
//   vmis_predict_kernel
//     phase 0   the session's prep record -> per-list arrays
//     phase 1-2 candidate sessions.  Merge mode (default): the posting lists are sorted by recency, so they are staged
//               in LDS and merged (merge path); the first m distinct sessions are the candidates, the k-cut counts the
//               few numerator classes.  Hash mode (what does not fit): packed session hash table, one scan, histograms
//     phase 3a  walk A: one neighbour row (64-byte slot) per lane; popular items -> exact direct-mapped accumulators,
//               everything else -> a sketch of upper bounds.  One fire-and-forget LDS add per row item
//     phase 4a  exact top-n of the direct-mapped items -> threshold score
//     phase 3b  walk B: only items whose sketch word can still beat the threshold reach the exact hash table
//     phase 4b  f64 score = idf_eff * acc / (10 U), business rules, top-n by (score desc, public id asc)
//   rows_to_slots_kernel  CSR rows -> 64-byte row slots when an index is attached to a device
//
// Everything up to the final f64 multiply/divide is integer arithmetic, so (neighbours, numerators, accumulators) are
// bit-identical to the canonical CPU oracle by construction.  No MFMA: this is sparse gather/scatter.
//
// Queries whose candidate or item sets do not fit the LDS tables are queued on a device-side retry list and served by
// the same kernel instantiated with its tables in a global scratch arena (GLOBAL_TABLES = true) -- still on the GPU,
// never on the CPU.  STAGE 1..3 are the cuts of the item-sharded pipeline (stages A..C).
// =====================================================================================
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>

#include "srn_kernels.h"
#include "srn_device.h"
#include "srn_prep.h"

namespace srn {

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return fail(SRN_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)


static constexpr int MAX_ITEM_PASSES = 64;
// item-space partition passes (MAX_ITEM_PASSES) before giving up on the LDS table; the row cache size is per instantiation


// LDS scalar slots
enum { S_CNT = 0, S_OVF, S_XLO, S_RMAX, S_U, S_P, S_SUMW, S_SELD, S_SELR, S_NB, S_ICNT, S_CCNT, S_I, S_HAVE_T, S_TIDX,
       S_ERR, S_TKEY_LO, S_TKEY_HI, S_COVF, S_SORTED, S_LIVE, S_L1, S_L2, S_NK };

template <typename T> struct SlotTraits;
template <> struct SlotTraits<uint32_t> { static constexpr uint32_t EMPTY = 0xFFFFFFFFu; };
template <> struct SlotTraits<unsigned long long> { static constexpr unsigned long long EMPTY = ~0ull; };

// double hashing over a power-of-two table: start slot from the high product bits, odd stride
__device__ __forceinline__ uint32_t hash_start(uint32_t key, uint32_t mask) { return ((key * 0x9E3779B1u) >> 7) & mask; }
__device__ __forceinline__ uint32_t hash_step(uint32_t key, uint32_t mask) { return (((key * 0x85EBCA6Bu) >> 9) | 1u) & mask; }
__device__ __forceinline__ uint32_t hash_part(uint32_t key, uint32_t parts) { return __umulhi(key * 0xC2B2AE35u, parts); }
__device__ __forceinline__ int bits_for(uint32_t v) { return v ? 32 - __clz((int)v) : 0; }

__device__ __forceinline__ uint64_t score_key(double s) {   // order-preserving f64 -> u64
    uint64_t b = (uint64_t)__double_as_longlong(s);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_score(uint64_t k) {
    uint64_t b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    return __longlong_as_double((long long)b);
}

// -------------------------------------------------------------------------------------
// r-th largest key among the valid entries of `n` slots (keys distinct and < 2^nbits, 1 <= r <= #valid):
// MSD radix select, 11 bits per pass starting at the top SIGNIFICANT bit (so the first histogram is not
// degenerate), 2048-bin histogram in LDS.  `hist` must be all zero on entry and is all zero on return
// (the digit search clears the bins it reads), so a pass costs two barriers.  keyfn(i, key&) -> valid.
// -------------------------------------------------------------------------------------
// bin b lives at word b + (b >> 5): the digit search gives every lane 32 consecutive bins, and without the
// one-word skew all 64 lanes would read the same LDS bank (64-way conflict on each of the 32 reads)
__device__ __forceinline__ uint32_t sel_word(uint32_t bin) { return bin + (bin >> 5); }
// The bin (bins counted from the top) that holds the `remain`-th largest entry of a 2048-bin histogram: wave 0 searches
// (lane owns bins [32 lane, 32 lane + 32), suffix sums from the top bin down), writes misc[S_SELD] = bin and
// misc[S_SELR] = rank of the wanted entry inside that bin, and clears the histogram.  The histogram must be complete
// (barrier) on entry; ends with a barrier.
template <int BLOCK>
__device__ void hist_search(uint32_t* hist, uint32_t remain, uint32_t* misc) {
    const int tid = threadIdx.x;
    if (tid < 64) {
        constexpr int PER = SEL_BINS / 64;   // 32 bins per lane, kept as 4 group sums of 8 (not 32 registers)
        uint32_t g[4];
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) { uint32_t t = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) t += hist[sel_word(PER * tid + gi * 8 + j)];
            g[gi] = t; }
        const uint32_t s = g[0] + g[1] + g[2] + g[3];
        const uint32_t pre = wave_incl_scan(s);
        uint32_t above = (uint32_t)__builtin_amdgcn_readlane((int)pre, 63) - pre;   // entries in bins owned by higher lanes
        if (above < remain && remain <= above + s) {   // the target bin is one of mine: find the group of 8, then the bin
            int gsel = -1; uint32_t ab = above;
#pragma unroll
            for (int gi = 3; gi >= 0; --gi) {
                if (gsel < 0 && ab < remain && remain <= ab + g[gi]) { gsel = gi; above = ab; }
                ab += g[gi];
            }
            uint32_t cj[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) cj[j] = hist[sel_word(PER * tid + gsel * 8 + j)];
#pragma unroll
            for (int j = 7; j >= 0; --j) {
                if (above < remain && remain <= above + cj[j]) { misc[S_SELD] = PER * tid + gsel * 8 + j; misc[S_SELR] = remain - above; }
                above += cj[j];
            }
        }
#pragma unroll 8
        for (int j = 0; j < PER; ++j) hist[sel_word(PER * tid + j)] = 0;
    }
    __syncthreads();
}
template <int BLOCK, typename KeyT, typename F>
__device__ KeyT block_select_desc(F keyfn, uint32_t n, int nbits, uint32_t r, uint32_t* hist, uint32_t* misc) {
    const int tid = threadIdx.x;
    KeyT prefix = 0;   // value of the bits above `rem`
    uint32_t remain = r;
    int rem = nbits;
    while (rem > 0) {
        const int w = rem < SEL_BITS ? rem : SEL_BITS, shift = rem - w;
        for (uint32_t i0 = tid; i0 < n; i0 += 8 * BLOCK) {   // a wave is LDS-latency bound: 8 independent reads per wait
            KeyT key[8]; bool ok[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const uint32_t i = i0 + u * BLOCK; key[u] = 0; ok[u] = i < n && keyfn(i, key[u]); }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const KeyT above = rem >= (int)(8 * sizeof(KeyT)) ? (KeyT)0 : (KeyT)(key[u] >> rem);
                if (ok[u] && above == prefix) atomicAdd(&hist[sel_word((uint32_t)(key[u] >> shift) & ((1u << w) - 1u))], 1u);
            }
        }
        __syncthreads();
        hist_search<BLOCK>(hist, remain, misc);
        prefix = (KeyT)((prefix << w) | (KeyT)misc[S_SELD]);
        remain = misc[S_SELR];
        rem = shift;
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

// ---- top-64 of the candidate buffer without a full block sort (how_many <= 64, the common case) ---------------
// Every wave bitonic-sorts 64 candidates at a time in registers (cross-lane shuffles, no barriers) and folds them
// into its running best-64; the 8 per-wave runs are then merged pairwise through LDS (3 barriers).  Order:
// (key desc, idx asc); padding = (0, EMPTY32), which is worse than any real candidate.
__device__ __forceinline__ bool cand_better(uint64_t ka, uint32_t ia, uint64_t kb, uint32_t ib) { return ka > kb || (ka == kb && ia < ib); }
__device__ __forceinline__ void wave_merge_clean(uint64_t& k, uint32_t& i, int lane, int from_stride) {   // bitonic -> descending
#pragma unroll
    for (int stride = 32; stride > 0; stride >>= 1) {
        if (stride > from_stride) continue;
        const uint64_t pk = ((uint64_t)__shfl_xor((uint32_t)(k >> 32), stride, 64) << 32) | __shfl_xor((uint32_t)k, stride, 64);
        const uint32_t pi = __shfl_xor(i, stride, 64);
        const bool lower = (lane & stride) == 0;
        if (cand_better(k, i, pk, pi) != lower) { k = pk; i = pi; }
    }
}
__device__ __forceinline__ void wave_sort_desc(uint64_t& k, uint32_t& i, int lane) {
#pragma unroll
    for (int size = 2; size <= 64; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            const uint64_t pk = ((uint64_t)__shfl_xor((uint32_t)(k >> 32), stride, 64) << 32) | __shfl_xor((uint32_t)k, stride, 64);
            const uint32_t pi = __shfl_xor(i, stride, 64);
            const bool keep_better = ((lane & stride) == 0) == ((lane & size) == 0);   // descending sub-block: lower lane keeps the better
            if (cand_better(k, i, pk, pi) != keep_better) { k = pk; i = pi; }
        }
    }
}
// fold a descending run (sk, si) into the descending running best (rk, ri): keep the best 64 of the 128
__device__ __forceinline__ void wave_fold(uint64_t& rk, uint32_t& ri, uint64_t sk, uint32_t si, int lane) {
    const uint64_t pk = ((uint64_t)__shfl((uint32_t)(sk >> 32), 63 - lane, 64) << 32) | __shfl((uint32_t)sk, 63 - lane, 64);
    const uint32_t pi = __shfl(si, 63 - lane, 64);
    if (!cand_better(rk, ri, pk, pi)) { rk = pk; ri = pi; }
    wave_merge_clean(rk, ri, lane, 32);
}
template <int BLOCK>
__device__ void block_top64(uint64_t* ckey, uint32_t* cidx, uint32_t cnt) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NW = BLOCK / 64;
    uint64_t rk = 0; uint32_t ri = EMPTY32;
    bool first = true;
    for (uint32_t c0 = wave * 64; c0 < cnt; c0 += NW * 64) {   // wave-uniform
        const uint32_t e = c0 + lane;
        uint64_t k = e < cnt ? ckey[e] : 0; uint32_t i = e < cnt ? cidx[e] : EMPTY32;
        wave_sort_desc(k, i, lane);
        if (first) { rk = k; ri = i; first = false; } else wave_fold(rk, ri, k, i, lane);
    }
    __syncthreads();   // every chunk has been read
    ckey[wave * 64 + lane] = rk; cidx[wave * 64 + lane] = ri;
    __syncthreads();
#pragma unroll
    for (int n = NW; n > 1; n = (n + 1) / 2) {   // (any number of waves: fold the upper half onto the lower)
        const int up = (n + 1) / 2;
        if (wave + up < n) {
            wave_fold(rk, ri, ckey[(wave + up) * 64 + lane], cidx[(wave + up) * 64 + lane], lane);
            ckey[wave * 64 + lane] = rk; cidx[wave * 64 + lane] = ri;
        }
        __syncthreads();
    }
}

// Both LDS tables are bucketized: 4 consecutive slots (16 B) form a bucket that one ds_read_b128 fetches, and
// double hashing walks buckets.  A wave waits for its unluckiest lane, and the longest of 64 probe sequences at
// load 0.8 is ~16 single slots but only ~4-5 four-slot buckets.
//
// insert-or-add into the packed session table: slot = (rank << NB) | numerator.  Returns 1 if the rank was new,
// 0 if it existed, -1 if the probe budget ran out (table too full).
// MASKS: the low NB bits are a bit set of evolving-session positions (OR-combined) instead of a sum of weights.
template <typename SlotT, bool MASKS>
__device__ __forceinline__ int sess_insert(SlotT* stab, uint32_t bmask, uint32_t NB, uint32_t r, uint32_t w) {
    constexpr SlotT SEMPTY = SlotTraits<SlotT>::EMPTY;
    uint32_t b = hash_start(r, bmask);
    const uint32_t step = hash_step(r, bmask);
    for (int probe = 0; probe < MAX_PROBES;) {
        SlotT c[4];
        if constexpr (sizeof(SlotT) == 4) { const uint4 v = *reinterpret_cast<const uint4*>(&stab[4 * b]); c[0] = v.x; c[1] = v.y; c[2] = v.z; c[3] = v.w; }
        else { const ulonglong2 v0 = *reinterpret_cast<const ulonglong2*>(&stab[4 * b]), v1 = *reinterpret_cast<const ulonglong2*>(&stab[4 * b + 2]);
               c[0] = v0.x; c[1] = v0.y; c[2] = v1.x; c[3] = v1.y; }
        int hit = -1, empty = -1;
#pragma unroll
        for (int i = 3; i >= 0; --i) { if (c[i] == SEMPTY) empty = i; else if ((uint32_t)(c[i] >> NB) == r) hit = i; }
        if (hit >= 0) { if (MASKS) atomicOr(&stab[4 * b + hit], (SlotT)w); else atomicAdd(&stab[4 * b + hit], (SlotT)w); return 0; }
        if (empty >= 0) {
            const SlotT old = atomicCAS(&stab[4 * b + empty], SEMPTY, ((SlotT)r << NB) | (SlotT)w);
            if (old == SEMPTY) return 1;
            if ((uint32_t)(old >> NB) == r) { if (MASKS) atomicOr(&stab[4 * b + empty], (SlotT)w); else atomicAdd(&stab[4 * b + empty], (SlotT)w); return 0; }
            continue;   // another rank took that slot meanwhile: look at this bucket again
        }
        b = (b + step) & bmask; ++probe;
    }
    return -1;
}
// Item-sharded pipeline (DESIGN.md "Multi-GPU"): the fused kernel cut at its two exchange points.
//   STAGE 1 (A)  phases 0-1 + local m-cut          -> this shard's candidate (rank, partial numerator) list
//   -- all-gather of the candidate lists --
//   STAGE 2 (B)  merge the G lists, global m-cut / k-cut, neighbours sorted by rank (same order on every shard),
//                partial first-match position of each neighbour over the evolving items THIS shard owns
//   -- all-reduce(min) of the first-match positions (+ the current item's attribute byte) --
//   STAGE 3 (C)  phases 3-4 over this shard's row fragments -> exact top-n of the items it owns
//   -- all-gather of the per-shard top-n, merged on every rank --
static constexpr int MINPOS_NONE = 0x7FFFFFFF;
struct __attribute__((packed, aligned(4))) RowVec { uint32_t x, y, z, w; };   // 4 row items; rows are only 4-byte aligned

// in-LDS bitonic sort, descending, of n (power of two) packed session slots; pads must be 0
template <int BLOCK, typename SlotT>
__device__ void block_sort_slots(SlotT* a, uint32_t n) {
    const uint32_t tid = threadIdx.x;
    for (uint32_t kk = 2; kk <= n; kk <<= 1) {
        for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
            for (uint32_t t = tid; t < (n >> 1); t += BLOCK) {
                const uint32_t lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
                const SlotT x = a[lo], y = a[hi];
                const bool descending = (lo & kk) == 0;
                if ((x > y) != descending) { a[lo] = y; a[hi] = x; }
            }
            __syncthreads();
        }
    }
}

// optional per-phase cycle accounting (debug; p.phase_cycles == nullptr in normal operation)
// -------------------------------------------------------------------------------------
// Row layout kernel (attach time): CSR rows -> 64-byte slots + overflow area (see DeviceIndex).  One thread per row, 1024
// consecutive rows per block; the overflow offsets are a block-local exclusive scan on top of the block's base (computed
// on the host from the row lengths), so the layout is the sequential one.  Slot n is the empty row.
// -------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void rows_to_slots_kernel(const uint64_t* __restrict__ row_off, const uint32_t* __restrict__ row_items, uint64_t n,
                                                             const uint32_t* __restrict__ block_base, uint32_t* __restrict__ slots, uint32_t* __restrict__ ext) {
    __shared__ uint32_t wave_tot[16];
    const uint64_t r = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t o = 0, len = 0;
    if (r < n) { o = row_off[r]; len = row_off[r + 1] - o; }
    const uint32_t e = len > 15 ? (uint32_t)(len - 14) : 0u;
    uint32_t inc = e;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    uint32_t base = block_base[blockIdx.x];
    for (int w = 0; w < wave; ++w) base += wave_tot[w];
    const uint32_t eoff = base + inc - e;
    if (r > n) return;
    uint32_t sl[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) sl[i] = EMPTY32;
    sl[0] = (uint32_t)len;   // (0 for the empty row r == n)
    if (len <= 15) { for (uint32_t i = 0; i < (uint32_t)len; ++i) sl[1 + i] = row_items[o + i]; }
    else { sl[1] = eoff;
#pragma unroll
           for (int i = 0; i < 14; ++i) sl[2 + i] = row_items[o + i];
           for (uint64_t i = 14; i < len; ++i) ext[eoff + (i - 14)] = row_items[o + i]; }
    uint4* dst = reinterpret_cast<uint4*>(slots + r * 16);
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) dst[qd] = make_uint4(sl[4 * qd], sl[4 * qd + 1], sl[4 * qd + 2], sl[4 * qd + 3]);
}

// the same for an item shard's row FRAGMENTS: 16-byte slots {len, i0, i1, i2}; > 3 items: {len, offset of items 2.. in ext, i0, i1}
__global__ __launch_bounds__(1024) void rows_to_frags_kernel(const uint64_t* __restrict__ row_off, const uint32_t* __restrict__ row_items, uint64_t n,
                                                             const uint32_t* __restrict__ block_base, uint32_t* __restrict__ slots, uint32_t* __restrict__ ext) {
    __shared__ uint32_t wave_tot[16];
    const uint64_t r = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t o = 0, len = 0;
    if (r < n) { o = row_off[r]; len = row_off[r + 1] - o; }
    const uint32_t e = len > 3 ? (uint32_t)(len - 2) : 0u;
    const uint32_t inc = wave_incl_scan(e);
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    uint32_t base = block_base[blockIdx.x];
    for (int w = 0; w < wave; ++w) base += wave_tot[w];
    const uint32_t eoff = base + inc - e;
    if (r > n) return;
    uint32_t sl[4] = {(uint32_t)len, EMPTY32, EMPTY32, EMPTY32};   // (0 items for the empty row r == n)
    if (len <= 3) { for (uint32_t i = 0; i < (uint32_t)len; ++i) sl[1 + i] = row_items[o + i]; }
    else { sl[1] = eoff; sl[2] = row_items[o]; sl[3] = row_items[o + 1];
           for (uint64_t i = 2; i < len; ++i) ext[eoff + (i - 2)] = row_items[o + i]; }
    reinterpret_cast<uint4*>(slots)[r] = make_uint4(sl[0], sl[1], sl[2], sl[3]);
}

// -------------------------------------------------------------------------------------
// Prep kernel: the dependent look-ups of phase 0 (public id -> dense idx -> posting list bounds -> first / m-th rank) done ahead of the main kernel, where
// a whole workgroup would wait on that chain; the main kernel reads one record.  Record = PrepHead + max_len * PrepItem, positions counted from the most
// recent item.  EIGHT LANES PER SESSION, one evolving position each (sessions of > 8 items: in rounds of 8): the chain is ~5 dependent loads long and a
// thread per session walked it once per item -- 32 us for a batch of 4 096 sessions, where the chain itself is the cost; per-session totals are 8-lane
// shuffles.  Rounds 1-2 ran one thread per session; the records are the same byte for byte.
// zero_a / zero_b (either may be null): the 4-word / 1-word device counters of the launch sequence that follows on the same stream (the hand-over count and
// finish_big's ticket; the retry count), cleared here instead of by two fill kernels.
// -------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vmis_prep_kernel(DeviceIndex ix, const uint64_t* __restrict__ items_flat, const uint32_t* __restrict__ q_off,
                                                        uint32_t nq, uint32_t m, uint32_t max_len, char* out, uint32_t stride, uint32_t* zero_a, uint32_t* zero_b,
                                                        const IdSlot* __restrict__ loc_table, uint32_t loc_mask, unsigned long long* __restrict__ okeys) {
    // okeys (round 5, or null): one 64-bit sort key per query -- (dense idx of its MOST POPULAR known item, clamped to 16 bits) << 32 | q.  Queries that share their most
    // popular item share most of their posting-list entries and neighbour rows; the launch sequence sorts the batch by this key and serves it in that order, an eighth of
    // the order per XCD, so that the second of two such queries finds the first one's lines in its XCD's L2 (the dense idx is the popularity rank).
    // loc_table (the shard group's neighbours pipeline, srn_group.hip): `ix` is the REPLICATED dictionary + posting lists of the whole index, loc_table the id table of
    // the item shard whose rows the record's consumer walks -- the lists are looked up in the first, the dense idx written to the record (what the kernels compare
    // row items with: the current item) in the second.
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < 8 && zero_a) zero_a[t] = 0u;
    if (t == 8 && zero_b) *zero_b = 0u;
    const uint32_t q = t / PREP_LANES, sub = t % PREP_LANES;
    if (q >= nq) return;   // (nq is a multiple of nothing in particular: whole 8-lane groups leave together, the shuffles below stay within a group)
    prep_group(ix, items_flat, q_off, q, sub, m, max_len, out + (size_t)q * stride, loc_table, loc_mask, okeys);
}

// Wave priorities by phase (s_setprio), as in vmis_fast_kernel (srn_fast.hip, where the measurements are): four decimal digits = front end (record .. cuts) | clears |
// walk A | harvest (phase 4a .. final top-n); 0 = none.  Config 3, the general kernel alone over 2^18 queries (SRN_NO_FAST=1 tools/fast_time.py): 15.65 ms without, 15.46 with 2013
// (2003 / 2113 / 1002: 15.48-15.54) -- two workgroups per CU contend less than the fast kernel's three.
#ifndef SRN_GEN_PRIO_LEVELS
#define SRN_GEN_PRIO_LEVELS 2013
#endif
#define GEN_PRIO(ph) do { if (SRN_GEN_PRIO_LEVELS) __builtin_amdgcn_s_setprio((short)(((SRN_GEN_PRIO_LEVELS) / (ph)) % 10)); } while (0)
#ifndef SRN_STOP_AT
#define SRN_STOP_AT (-1)   // experiments only (tools/phase_insts.sh): leave the query after phase tick N (0..4, 8..10) to count instructions per phase
#endif
#define SRN_TICK(ph)                                                                                         \
    do { if (ticking && tid == 0) { const long long t_ = clock64(); atomicAdd(&p.phase_cycles[ph], (unsigned long long)(t_ - t_prev)); t_prev = t_; } } while (0); \
    if (SRN_STOP_AT == (ph)) continue

// WG_PER_CU = workgroups the build is meant to co-reside with: 2 (4 waves per SIMD, <= 128 VGPRs) or 3 (6 waves, <= 80 VGPRs: more
// latency hiding for the price of spills; used with the small LDS geometry when the queries are large, see device_predict)
template <int BLOCK, typename SlotT, bool GLOBAL_TABLES, int STAGE = 0, bool MASKS = false, int WG_PER_CU = 2>
__global__ __launch_bounds__(BLOCK, (WG_PER_CU * BLOCK) / 256) void vmis_predict_kernel(DeviceIndex ix_arg, LaunchParams p_arg, KernelCfg c_arg, LaunchAux aux_arg, ShardIO sh_arg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // The parameter blocks (the arguments, packed in order with their natural alignment) are read where they
    // are needed, straight from the kernel-argument segment -- constant memory, scalar loads.  Used as by-value arguments the
    // compiler keeps all ~60 words live in SGPRs for the whole kernel and spills them to VGPR lanes (-39 % spill moves this way).
    typedef const __attribute__((address_space(4))) char* KArg;
    const KArg ka = (KArg)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr size_t OFF_P = (sizeof(DeviceIndex) + alignof(LaunchParams) - 1) / alignof(LaunchParams) * alignof(LaunchParams);
    constexpr size_t OFF_C = (OFF_P + sizeof(LaunchParams) + alignof(KernelCfg) - 1) / alignof(KernelCfg) * alignof(KernelCfg);
    const __attribute__((address_space(4))) DeviceIndex& ix = *(const __attribute__((address_space(4))) DeviceIndex*)ka;
    const __attribute__((address_space(4))) LaunchParams& p = *(const __attribute__((address_space(4))) LaunchParams*)(ka + OFF_P);
    const __attribute__((address_space(4))) KernelCfg& c = *(const __attribute__((address_space(4))) KernelCfg*)(ka + OFF_C);
    constexpr size_t OFF_X = (OFF_C + sizeof(KernelCfg) + alignof(LaunchAux) - 1) / alignof(LaunchAux) * alignof(LaunchAux);
    constexpr size_t OFF_S = (OFF_X + sizeof(LaunchAux) + alignof(ShardIO) - 1) / alignof(ShardIO) * alignof(ShardIO);
    const __attribute__((address_space(4))) LaunchAux& aux = *(const __attribute__((address_space(4))) LaunchAux*)(ka + OFF_X);
    const __attribute__((address_space(4))) ShardIO& sh = *(const __attribute__((address_space(4))) ShardIO*)(ka + OFF_S);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NWAVES = BLOCK / 64;
    constexpr SlotT SEMPTY = SlotTraits<SlotT>::EMPTY;

    // ---- LDS carve-up ------------------------------------------------------------------
    uint32_t* misc = (uint32_t*)smem;                       // MISC_WORDS
    uint64_t* q_raw = (uint64_t*)(smem + c.off_q);                            // q_cap   raw ids, pos 0 = most recent
    unsigned long long* l_base = (unsigned long long*)(q_raw + c.q_cap);      // q_cap   posting list start
    uint32_t* q_idx = (uint32_t*)(l_base + c.q_cap);                          // q_cap   dense idx or kNone
    uint32_t* l_len = q_idx + c.q_cap;                                        // q_cap   truncated list length (0 = inactive)
    uint32_t* l_pre = l_len + c.q_cap;                                        // q_cap+4 exclusive prefix of l_len
    char* region_b = smem + c.off_b;
    uint32_t* hist = (uint32_t*)region_b;                                     // 2048-bin select histogram (phase 2 only)
    char* region_a = GLOBAL_TABLES ? (aux.gscratch + (size_t)blockIdx.x * aux.gscratch_stride) : (smem + c.off_a);

    SlotT* stab = (SlotT*)region_a;                                           // phase 1-2
    uint32_t* hot = (uint32_t*)region_a;                                      // phase 3-4: direct-mapped accumulators (idx < hot_slots)
    uint32_t* sketch = hot + c.hot_slots;                                     //            upper bounds of all other items' accumulators, by hash
    uint32_t* ikeys = sketch + c.sketch_slots;                                //            exact hash table keys (buckets of 4), then accumulators
    int* iacc = (int*)(ikeys + c.item_slots);
    SlotT* nbl = (SlotT*)region_b;                                            // neighbours (phase 2-3)
    SlotT* nb_spill = aux.nb_spill ? (SlotT*)aux.nb_spill + (size_t)blockIdx.x * p.k : nullptr;   // global copy: phase 4a reuses the LDS
    uint64_t* ckey = (uint64_t*)region_b;                                     // candidates (phase 4)
    uint32_t* cidx = (uint32_t*)(region_b + CAND_CAP * 8);

    const uint32_t NB = c.num_bits;
    const SlotT num_mask = ((SlotT)1 << NB) - 1;
    // MASKS (sessions of <= 8 items, m <= m_index): a session slot carries the SET of evolving positions whose posting
    // list holds the session.  For a candidate (rank >= x_lo) that set is exactly "which evolving items the row
    // contains" (lists are complete above their m-th entry), so both the similarity numerator (table lookup) and the
    // first-match position (lowest set bit) come from it and phase 3 needs no first-match pass over the rows.
    uint8_t* wlut = (uint8_t*)(smem + MISC_WORDS * 4);                        // 256 B: numerator of each position set
    auto num_of = [&](SlotT sl) -> uint32_t { return MASKS ? (uint32_t)wlut[(uint32_t)(sl & num_mask)] : (uint32_t)(sl & num_mask); };
    const uint32_t inb = c.item_buckets, H = c.hot_slots, SB = c.sum_bits, SK = c.sketch_slots, SKM = c.sketch_slots - 1u;
    const uint32_t nq_eff = aux.qlist ? *aux.qlist_n : p.nq;

    // Read once what the inner loops keep asking for: with machine LICM off (build flag) a kernel-argument field used inside a
    // loop is re-read there -- an s_load plus a wait that also drains the LDS queue -- every iteration.
    const uint32_t n_kept = ix.n_kept;
    const bool dump_nb = p.nb_rank != nullptr, ticking = p.phase_cycles != nullptr;
    const double idf_hi = ix.idf_hi;
    for (uint32_t qi = blockIdx.x; qi < nq_eff; qi += gridDim.x) {
        const uint32_t q = aux.qlist ? aux.qlist[qi] : qi;
        // with a prep record every global read of this phase is issued here, in one round: the head (uniform), the first
        // 8 item entries and the run starts (lanes 0..7) -- the session length comes with the head, not from q_off
        const bool use_prep = STAGE == 0 && p.prep != nullptr;
        const char* const rec = use_prep ? p.prep + (size_t)q * p.prep_stride : nullptr;
        PrepHead hd; PrepItem x0{kNone, 0u, 0u, 0u, 0ull}; uint32_t rs0 = 0;
        uint32_t qb = 0, L;
        if (use_prep) {
            hd = *(const PrepHead*)rec;   // (uniform address)
            if (lane < 8) { if ((uint32_t)lane < p.max_len) x0 = ((const PrepItem*)(rec + sizeof(PrepHead)))[lane]; rs0 = ((const PrepHead*)rec)->run_start[lane]; }   // (every wave: all need entry 0)
            L = hd.L;
        } else { qb = p.q_off[q]; L = p.q_off[q + 1] - qb; }
        if (L == 0 || L > p.max_len) {   // block-uniform
            if (tid == 0) { p.out_counts[q] = 0xFFFFFFFFu;
                if (p.stats) { for (int i = 0; i < 8; ++i) p.stats[(size_t)q * 8 + i] = 0; p.stats[(size_t)q * 8 + 7] = 2; }
                if (p.nb_cnt) p.nb_cnt[q] = 0; }
            continue;
        }
        // ---- phase 0: reset, translate items ---------------------------------------------
        __syncthreads();   // previous query's LDS reads are done
        long long t_prev = ticking ? clock64() : 0;
        GEN_PRIO(1000);
        if (tid < MISC_WORDS) misc[tid] = 0;
        for (uint32_t i = tid; i < SEL_WORDS; i += BLOCK) hist[i] = 0;
        if (MASKS && tid < 256) { uint32_t acc = 0; for (uint32_t b = 0; b < L; ++b) if ((tid >> b) & 1) acc += L - b; wlut[tid] = (uint8_t)acc; }
        uint32_t x_lo, r_max, U, P, cur_idx, sumw = 0, nruns = 0;
        if (use_prep) {   // the look-ups were done by the prep kernel: one record per query, no barrier needed here
            const PrepItem* pit = (const PrepItem*)(rec + sizeof(PrepHead));
            x_lo = hd.xlo; r_max = hd.rmax; U = hd.U; P = hd.P; sumw = hd.sumw;
            nruns = hd.nruns;
            cur_idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)x0.idx);
            for (uint32_t pos = tid; pos < L; pos += BLOCK) { const PrepItem x = pos < 8 ? x0 : pit[pos]; q_idx[pos] = x.idx; l_len[pos] = x.len; l_pre[pos] = x.pre; l_base[pos] = x.base; }
            if (tid == 0) { l_pre[L] = hd.P; misc[S_U] = hd.U; misc[S_RMAX] = hd.rmax; misc[S_XLO] = hd.xlo; misc[S_SUMW] = hd.sumw; misc[S_P] = hd.P; }   // (same wave zeroed them)
        } else {
        for (uint32_t i = tid; i < L; i += BLOCK) q_raw[i] = p.items_flat[qb + (L - 1 - i)];   // pos 0 = most recent item
        __syncthreads();
        for (uint32_t pos = tid; pos < L; pos += BLOCK) {
            const uint64_t raw = q_raw[pos];
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
                    atomicAdd((uint32_t*)&misc[S_SUMW], L - pos);
                }
            }
            l_len[pos] = len;
        }
        __syncthreads();
        if (tid == 0) { uint32_t acc = 0; for (uint32_t i = 0; i < L; ++i) { l_pre[i] = acc; acc += l_len[i]; } l_pre[L] = acc; misc[S_P] = acc;
            if (STAGE == 2) {   // the candidates come from the G gathered lists instead of the posting lists
                uint32_t tot = 0; for (uint32_t g = 0; g < sh.n_shards; ++g) tot += sh.gathered_cnt[(size_t)g * p.nq + q];
                misc[S_P] = tot; misc[S_XLO] = 0xFFFFFFFFu; misc[S_RMAX] = 0; } }
        phase_sync<GLOBAL_TABLES>();
        x_lo = misc[S_XLO]; r_max = misc[S_RMAX]; U = misc[S_U]; P = misc[S_P]; cur_idx = q_idx[0];
        }
        // Merge mode (the fused kernel's normal case): the posting lists are sorted, so the candidate sessions come out of a
        // merge tree over the lists staged in LDS -- no session hash table, no selects (see phases 1-2 below).  It needs two
        // buffers of P words in regions B + A (contiguous), u32 slots, <= 8 lists and <= 63 numerator classes.
        bool merge_mode = false;
        uint32_t* const wb = (uint32_t*)region_b;
        const uint32_t wwords = ((c.off_a - c.off_b) + c.region_a_bytes) / 4;
        if constexpr (STAGE == 0 && !GLOBAL_TABLES && sizeof(SlotT) == 4)
            merge_mode = p.prep != nullptr && nruns <= 8 && sumw <= 63 && (size_t)2 * P + 64 <= wwords && !c.no_merge;
        // session table sized to this query: at most P entries are inserted, keep the load <= 2/3
        uint32_t sslots = 256; while (sslots < c.sess_slots && sslots * 2 < P * 3) sslots <<= 1;
        const uint32_t smask = sslots / 4 - 1;   // bucket mask (4 slots per bucket)
        if (!merge_mode) for (uint32_t i = tid; i < sslots; i += BLOCK) stab[i] = SEMPTY;
        else if (tid < 64) {   // run table 0 (start, valid length of every active list) and the numerator class counters
            wb[wwords - 64 + tid] = 0;
            if (tid < 8) { misc[32 + tid] = rs0; misc[40 + tid] = 0; }
        }
        phase_sync<GLOBAL_TABLES>();
        SRN_TICK(0);

        uint32_t K = 0, Cm = 0;
        if (merge_mode) {
        if constexpr (STAGE == 0 && !GLOBAL_TABLES && sizeof(SlotT) == 4) {
            // ---- phases 1-2, merge mode ------------------------------------------------------
            // stage: every list entry >= x_lo as a packed slot (rank << NB | payload) at its list's offset -- each list is
            //        sorted by rank, so the kept entries are a prefix and the staged list is a sorted run;
            // merge: ceil(log2(#lists)) levels of pairwise merges (merge path: a binary search on the thread's diagonal,
            //        then g = ceil(n / 512) sequential steps), ping-pong between two buffers of P words;
            // m-cut: in the merged run a session's copies (one per list that holds it) are adjacent: flag the first of each
            //        group, block prefix sum = index among the distinct sessions, the first m are the candidates D, payloads
            //        of a group combined (OR of position bits / sum of weights);
            // k-cut: D is ordered by recency, so inside one numerator class the order is already right: count the classes,
            //        find the class n* that holds the k-th best and how many r* of it are wanted, prefix-count that class.
            uint32_t* bufx = wb; uint32_t* bufy = wb + P;
            const uint32_t nlv = nruns <= 1 ? 0u : (uint32_t)bits_for(nruns - 1);
            uint32_t* in = (nlv & 1u) ? bufy : bufx; uint32_t* out = (nlv & 1u) ? bufx : bufy;   // the final run lands in bufx
            {   // stage, list by list (uniform base and weight: no per-element list look-up), 5 loads in flight per lane
                uint32_t run = 0;
                for (uint32_t pos = 0; pos < L; ++pos) {
                    const uint32_t len = l_len[pos];   // (uniform)
                    if (len == 0) continue;
                    const uint32_t* __restrict__ src = ix.post_rank + l_base[pos];
                    uint32_t* dst = in + l_pre[pos];
                    const uint32_t w = MASKS ? (1u << pos) : L - pos;
                    uint32_t kept = 0;
                    constexpr int SU = 5;   // 5 * 512 entries per trip: a full list of m = 2500 is one trip, one HBM latency
                    for (uint32_t e0 = tid; e0 < len; e0 += SU * BLOCK) {
                        uint32_t r4[SU];
#pragma unroll
                        for (int u = 0; u < SU; ++u) { const uint32_t e = e0 + u * BLOCK; r4[u] = e < len ? src[e] : 0u; }
#pragma unroll
                        for (int u = 0; u < SU; ++u) {
                            const uint32_t e = e0 + u * BLOCK;
                            const bool keep = e < len && r4[u] >= x_lo;   // (the list is sorted: the kept entries are a prefix)
                            if (keep) dst[e] = (r4[u] << NB) | w;
                            kept += (uint32_t)__popcll(__ballot(keep));
                        }
                    }
                    if (lane == 0 && kept) atomicAdd((uint32_t*)&misc[40 + run], kept);
                    ++run;
                }
            }
            __syncthreads();
            SRN_TICK(1);
            // merge tree: level l merges runs pairwise (<= 4 pairs at level 0).  Merge path: thread t produces outputs
            // [t g, (t + 1) g) of every pair -- a binary search on its diagonal (all pairs of the level searched together: their
            // LDS round trips overlap), then g sequential steps.  Run tables (start x8 | length x8) ping-pong in misc[32..63].
            const uint32_t nl = nruns <= 1 ? 0u : (uint32_t)bits_for(nruns - 1);
            for (uint32_t lev = 0; lev < nl; ++lev) {
                const uint32_t tb = 32 + 16 * (lev & 1u), tn = 32 + 16 * ((lev + 1) & 1u);
                const uint32_t nr = (nruns + (1u << lev) - 1) >> lev, np = (nr + 1) >> 1;
                uint32_t sa[4], la[4], sb[4], lb[4], d0[4], d1[4], lo[4], hi[4];
#pragma unroll
                for (int pr = 0; pr < 4; ++pr) {
                    sa[pr] = la[pr] = sb[pr] = lb[pr] = d0[pr] = d1[pr] = lo[pr] = hi[pr] = 0;
                    if ((uint32_t)pr < np) {
                        sa[pr] = misc[tb + 2 * pr]; la[pr] = misc[tb + 8 + 2 * pr];
                        if (2u * pr + 1 < nr) { sb[pr] = misc[tb + 2 * pr + 1]; lb[pr] = misc[tb + 8 + 2 * pr + 1]; }
                        const uint32_t total = la[pr] + lb[pr], g = (total + BLOCK - 1) / BLOCK;
                        d0[pr] = min((uint32_t)tid * g, total); d1[pr] = min(d0[pr] + g, total);
                        lo[pr] = d0[pr] > lb[pr] ? d0[pr] - lb[pr] : 0u; hi[pr] = min(d0[pr], la[pr]);
                        if (tid == 0) { misc[tn + pr] = sa[pr]; misc[tn + 8 + pr] = total; }
                    }
                }
                for (int step = 0; step < 13; ++step) {   // (lists hold <= m <= 2^13 entries... any length: loop until every range is empty)
                    bool any = false;
#pragma unroll
                    for (int pr = 0; pr < 4; ++pr) {
                        if ((uint32_t)pr >= np) break;   // (uniform: no per-lane test for the pairs that do not exist)
                        if (lo[pr] < hi[pr]) {
                            const uint32_t mid = (lo[pr] + hi[pr]) >> 1;
                            if (in[sa[pr] + mid] > in[sb[pr] + d0[pr] - mid - 1]) lo[pr] = mid + 1; else hi[pr] = mid;
                            any = any || lo[pr] < hi[pr];
                        }
                    }
                    if (__ballot(any) == 0ull) break;
                    if (step == 12) step = 0;   // (ranges wider than 2^13: keep going)
                }
#pragma unroll
                for (int pr = 0; pr < 4; ++pr) {
                    if ((uint32_t)pr < np && d0[pr] < d1[pr]) {
                        const uint32_t a = la[pr], b = lb[pr];
                        uint32_t* Cc = out + sa[pr];
                        uint32_t i = lo[pr], j = d0[pr] - lo[pr];
                        uint32_t va = i < a ? in[sa[pr] + i] : 0u, vb = j < b ? in[sb[pr] + j] : 0u;   // (a staged slot is never 0: its payload is >= 1)
                        for (uint32_t o = d0[pr]; o < d1[pr]; ++o) {   // branch-free step: one LDS read, one LDS write
                            const bool ta = va > vb;
                            Cc[o] = ta ? va : vb;
                            i += ta; j += !ta;
                            const bool more = ta ? i < a : j < b;
                            const uint32_t nxt = in[ta ? sa[pr] + i : sb[pr] + j];   // (past a run's end this reads a neighbour's entry or scratch: discarded)
                            const uint32_t nv = more ? nxt : 0u;
                            va = ta ? nv : va; vb = ta ? vb : nv;
                        }
                    }
                }
                __syncthreads();
                uint32_t* t = in; in = out; out = t;
            }
            const uint32_t n = nruns ? misc[32 + 16 * (nl & 1u) + 8] : 0u;
            // `in` == bufx holds the merged run F[0, n)
            const uint32_t* F = bufx; uint32_t* D = bufy;
            uint32_t Call;
            {   // m-cut.  Every copy of a session ORs (adds) itself into D[index of its group]: no look-ahead over the group
                const uint32_t g = (n + BLOCK - 1) / BLOCK, o0 = min((uint32_t)tid * g, n), o1 = min(o0 + g, n);
                uint32_t firsts = 0;
                uint32_t prev = o0 > 0 && o0 < o1 ? F[o0 - 1] >> NB : 0xFFFFFFFFu;
                const uint32_t prev0 = prev;
                for (uint32_t o = o0; o < o1; ++o) { const uint32_t r = F[o] >> NB; firsts += o == 0 || r != prev; prev = r; }
                for (uint32_t i = tid; i < min(n, p.m); i += BLOCK) D[i] = 0;   // (the last merge level has read bufy: barrier above)
                uint32_t idx = block_excl_scan<BLOCK>(firsts, misc + 32, Call);   // (the run tables in misc[32..63] are dead: scan scratch; barrier inside)
                prev = prev0;
                for (uint32_t o = o0; o < o1; ++o) {
                    const uint32_t v = F[o], r = v >> NB;
                    const bool first = o == 0 || r != prev; prev = r;
                    idx += first;      // idx - 1 = index of v's group among the distinct sessions
                    if (idx - 1 < p.m) { if (MASKS) atomicOr(&D[idx - 1], v); else atomicAdd(&D[idx - 1], (v & (uint32_t)num_mask) + (first ? r << NB : 0u)); }
                }
            }
            Cm = min(Call, p.m);
            __syncthreads();
            SRN_TICK(2);
            auto publish = [&](uint32_t at, uint32_t v) {
                nbl[at] = v; nb_spill[at] = v;
                if (dump_nb) { p.nb_rank[(size_t)q * p.k + at] = v >> NB; p.nb_num[(size_t)q * p.k + at] = num_of(v); } };
            if (Cm <= p.k) {   // every candidate is a neighbour (F in bufx is dead: the neighbour list overwrites it)
                for (uint32_t e = tid; e < Cm; e += BLOCK) publish(e, D[e]);
                if (tid == 0) misc[S_NB] = Cm;
            } else {   // k-cut
                uint32_t* cls = wb + (wwords - 64);
                if (sumw <= 15 && Cm <= 15 * BLOCK) {
                    // <= 16 classes, <= 15 candidates per thread: count in packed fields.  A thread adds 1 << 4 class into a 64-bit
                    // word (16 fields of 4 bits), the word is spread over 4 x 64 bits with 16-bit fields (class c -> word c & 3, field
                    // c >> 2; a wave's sum is <= 64 * 15 per field), 8 DPP wave sums, lane c picks its class.  ~100 VALU
                    // instructions per wave where a compare + ballot per class and batch costs 4 * (sumw + 1) per batch.
                    unsigned long long acc = 0;
                    for (uint32_t e = tid; e < Cm; e += BLOCK) acc += 1ull << (4u * num_of(D[e]));
                    uint32_t tot[8];
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const unsigned long long t = (acc >> (4 * jj)) & 0x000F000F000F000Full;
                        tot[2 * jj] = wave_sum((uint32_t)t); tot[2 * jj + 1] = wave_sum((uint32_t)(t >> 32));
                    }
                    const uint32_t dw = ((uint32_t)lane & 3u) * 2u + ((uint32_t)lane >> 3);   // class = lane: word lane & 3, field lane >> 2
                    uint32_t pick = tot[0];
#pragma unroll
                    for (int x = 1; x < 8; ++x) pick = dw == (uint32_t)x ? tot[x] : pick;
                    const uint32_t mycnt = lane < 16 ? (pick >> (16u * (((uint32_t)lane >> 2) & 1u))) & 0xFFFFu : 0u;
                    if (mycnt) atomicAdd(&cls[lane], mycnt);
                } else {   // lane v counts class v (ballots), one scattered atomic per lane at the end: no same-address pile-up
                    uint32_t mycnt = 0;
                    for (uint32_t e0 = 0; e0 < Cm; e0 += BLOCK) {
                        const uint32_t e = e0 + tid; const uint32_t nm = e < Cm ? num_of(D[e]) : 0xFFFFFFFFu;
                        for (uint32_t v = 0; v <= sumw; ++v) { const uint32_t cv = (uint32_t)__popcll(__ballot(nm == v)); if ((uint32_t)lane == v) mycnt += cv; }
                    }
                    if (mycnt) atomicAdd(&cls[lane], mycnt);
                }
                __syncthreads();
                uint32_t nstar, rstar;
                {   // lane v holds class v: suffix sums from the best class down, the boundary class is the highest one whose suffix reaches k
                    const uint32_t cv = cls[lane]; const uint32_t pre = wave_incl_scan(cv);
                    const uint32_t suf = (uint32_t)__builtin_amdgcn_readlane((int)pre, 63) - pre + cv;
                    const unsigned long long reach = __ballot(suf >= p.k);
                    nstar = 63u - (uint32_t)__clzll((long long)reach);
                    rstar = p.k - ((uint32_t)__builtin_amdgcn_readlane((int)suf, (int)nstar) - (uint32_t)__builtin_amdgcn_readlane((int)cv, (int)nstar));
                }
                SRN_TICK(3);
                const uint32_t g = (Cm + BLOCK - 1) / BLOCK, o0 = min((uint32_t)tid * g, Cm), o1 = min(o0 + g, Cm);
                if (g <= 6) {   // (m <= 3072) the thread's candidates and their classes stay in registers: read once, all loads in flight together
                    uint32_t dv[6], nmv[6];
#pragma unroll
                    for (int x = 0; x < 6; ++x) dv[x] = o0 + x < o1 ? D[o0 + x] : 0u;
#pragma unroll
                    for (int x = 0; x < 6; ++x) nmv[x] = o0 + x < o1 ? num_of(dv[x]) : 0u;   // (class 0 does not exist: a candidate matched something)
                    uint32_t mine = 0;
#pragma unroll
                    for (int x = 0; x < 6; ++x) mine += o0 + x < o1 && nmv[x] == nstar;
                    uint32_t tot_star;
                    uint32_t before = block_excl_scan<BLOCK>(mine, misc + 40, tot_star);
                    uint32_t sel = 0; bool take[6];
#pragma unroll
                    for (int x = 0; x < 6; ++x) {
                        const bool in_r = o0 + x < o1, eq = nmv[x] == nstar;
                        take[x] = in_r && (nmv[x] > nstar || (eq && before < rstar));
                        sel += take[x]; before += in_r && eq;
                    }
                    const uint32_t inc = wave_incl_scan(sel);
                    uint32_t base = 0; if (lane == 63 && inc) base = atomicAdd((uint32_t*)&misc[S_NB], inc);
                    uint32_t at = (uint32_t)__builtin_amdgcn_readlane((int)base, 63) + inc - sel;
#pragma unroll
                    for (int x = 0; x < 6; ++x) if (take[x]) publish(at++, dv[x]);
                } else {
                uint32_t mine = 0;
                for (uint32_t o = o0; o < o1; ++o) mine += num_of(D[o]) == nstar;
                uint32_t tot_star;
                uint32_t before = block_excl_scan<BLOCK>(mine, misc + 40, tot_star);
                uint32_t sel = 0;
                for (uint32_t o = o0; o < o1; ++o) { const uint32_t nm = num_of(D[o]); sel += nm > nstar || (nm == nstar && before < rstar); before += nm == nstar; }
                uint32_t at;   // output slots: a wave's selected entries are contiguous, waves in any order (the order of the list is free)
                { const uint32_t inc = wave_incl_scan(sel);
                  uint32_t base = 0; if (lane == 63 && inc) base = atomicAdd((uint32_t*)&misc[S_NB], inc);
                  at = (uint32_t)__builtin_amdgcn_readlane((int)base, 63) + inc - sel; }
                before = before - mine;   // back to this thread's start
                for (uint32_t o = o0; o < o1; ++o) {
                    const uint32_t v = D[o], nm = num_of(v);
                    if (nm > nstar || (nm == nstar && before < rstar)) publish(at++, v);
                    before += nm == nstar;
                }
                }
            }
            __syncthreads();
            K = misc[S_NB];
            SRN_TICK(4);
        }
        } else
        if constexpr (STAGE != 3) {
        // ---- phase 1: posting lists -> session table -------------------------------------
        // All <= U lists are walked as one flattened range (lane <-> list element, coalesced).  Entries
        // below x_lo (the m-th entry of a full list) can never be among the m most recent distinct
        // sessions, so they are read but not inserted.
        if constexpr (STAGE == 2) {
            uint32_t fresh = 0, rmin = 0xFFFFFFFFu, rmax = 0;
            bool ovf = false;
            for (uint32_t g = 0; g < sh.n_shards; ++g) {
                const SlotT* list = (const SlotT*)sh.gathered + ((size_t)g * p.nq + q) * (sh.gathered_stride ? sh.gathered_stride : p.m);
                const uint32_t cnt = sh.gathered_cnt[(size_t)g * p.nq + q];
                for (uint32_t e = tid; e < cnt; e += BLOCK) {
                    const SlotT v = list[e]; const uint32_t r = (uint32_t)(v >> NB);
                    rmin = min(rmin, r); rmax = max(rmax, r);
                    const int res = sess_insert<SlotT, MASKS>(stab, smask, NB, r, (uint32_t)(v & num_mask));
                    if (res < 0) ovf = true; else fresh += (uint32_t)res;
                }
            }
            fresh = wave_sum(fresh);
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) { rmin = min(rmin, (uint32_t)__shfl_xor((int)rmin, d, 64)); rmax = max(rmax, (uint32_t)__shfl_xor((int)rmax, d, 64)); }
            if (lane == 0) { if (fresh) atomicAdd((uint32_t*)&misc[S_CNT], fresh); atomicMin((uint32_t*)&misc[S_XLO], rmin); atomicMax((uint32_t*)&misc[S_RMAX], rmax); }
            if (ovf) misc[S_OVF] = 1;
        } else {
            uint32_t fresh = 0, pos = 0;
            bool ovf = false;
            // if an m-cut is possible, count the distinct sessions per top-11-bit bin of (rank - x_lo) as they are inserted:
            // the first pass of the m-cut select comes for free
            const bool want_h1 = !GLOBAL_TABLES && P > p.m;
            const int sh1 = max(bits_for(r_max - x_lo) - SEL_BITS, 0);
            auto fetch = [&](uint32_t e0, uint32_t (&r)[4], uint32_t (&w)[4]) {   // 4 list entries of this lane (coalesced across lanes)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t e = e0 + u * BLOCK;
                    r[u] = 0; w[u] = 0;
                    if (e < P) {
                        while (e >= l_pre[pos + 1]) ++pos;
                        r[u] = ix.post_rank[l_base[pos] + (e - l_pre[pos])];
                        w[u] = MASKS ? (1u << pos) : L - pos;
                    }
                } };
            auto note_fresh = [&](uint32_t rank) { ++fresh; if (want_h1) atomicAdd(&hist[sel_word((rank - x_lo) >> sh1)], 1u); };
            auto insert4 = [&](uint32_t (&r)[4], uint32_t (&w)[4]) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (w[u] && r[u] >= x_lo) {
                        const int res = sess_insert<SlotT, MASKS>(stab, smask, NB, r[u], w[u]);
                        if (res < 0) ovf = true; else if (res) note_fresh(r[u]);
                    }
                } };
            // (measured: neither prefetching the next batch's entries nor probing the 4 buckets of a batch together helps --
            //  the loop is bound by instruction issue, not by the list loads or the LDS round trips)
            for (uint32_t e0 = tid; e0 < P; e0 += 4 * BLOCK) {
                uint32_t r[4], w[4];
                fetch(e0, r, w);
                insert4(r, w);
            }
            fresh = wave_sum(fresh);
            if (lane == 0 && fresh) atomicAdd((uint32_t*)&misc[S_CNT], fresh);
            if (ovf) misc[S_OVF] = 1;
        }
        phase_sync<GLOBAL_TABLES>();
        if (misc[S_OVF]) {   // block-uniform: hand the query to the global-table pass
            // (stages of the sharded pipeline: queued for the stage's own global-table pass where the caller provides a list -- device_shard_stage --, marked otherwise)
            if (STAGE != 0 && (GLOBAL_TABLES || aux.retry_list == nullptr)) { if (tid == 0) { if (STAGE == 1) sh.cand_cnt[q] = 0xFFFFFFFFu; else sh.nb_cnt[q] = 0xFFFFFFFFu; } }
            else if (STAGE == 0 && GLOBAL_TABLES) { if (tid == 0) { p.out_counts[q] = 0xFFFFFFFFu; if (p.stats) p.stats[(size_t)q * 8 + 7] = 3; } }
            else if (tid == 0) aux.retry_list[atomicAdd(aux.retry_cnt, 1u)] = q;
            continue;
        }
        SRN_TICK(1);
        const uint32_t Call = misc[S_CNT];
        Cm = min(Call, p.m);
        const uint32_t x_lo2 = STAGE == 2 ? (Call ? misc[S_XLO] : 0u) : x_lo, r_max2 = STAGE == 2 ? misc[S_RMAX] : r_max;

        // ---- phase 2: m-cut, k-cut, compaction -------------------------------------------
        // ONE scan of the (sparse) session table, everything else on dense lists.  m-cut: the histogram of phase 1 gives
        // the top-11-bit rank bin of the m-th most recent session; the scan moves every session above that bin to the
        // dense list D and the bin's own few sessions to S1, whose r1 most recent join D.  k-cut: histogram of D by the
        // top 11 bits of the composite (numerator, rank), the bin of the k-th best; D's sessions above that bin go to
        // the neighbour list, the bin's own to S2, whose r2 best follow.  The session table is dead after the scan, so
        // the second histogram and the lists of the k-cut live in its LDS.  Lists that do not fit (very large m, k or
        // boundary bins) and the global-table pass fall back to radix selects over the table.
        const bool mcut = Call > p.m, kcut = STAGE != 1 && Cm > p.k;
        uint32_t tau = x_lo2;
        bool legacy = STAGE == 2 || GLOBAL_TABLES;   // (stage B learns the rank range only while inserting: no phase-1 histogram)
        if constexpr (STAGE != 2 && !GLOBAL_TABLES) {
            const uint32_t rb_slots = (c.off_a - c.off_b) / (uint32_t)sizeof(SlotT);
            SlotT* dl = (SlotT*)region_b;                       // D (becomes the neighbour list)
            const uint32_t CAP_S1 = rb_slots > p.m ? rb_slots - p.m : 0u, CAP_S2 = p.m;   // S2 cannot overflow (<= |D| <= m)
            SlotT* s1 = dl + p.m;
            uint32_t* hist2 = (uint32_t*)region_a;              // after the scan
            SlotT* tmp = (SlotT*)(region_a + SEL_WORDS * 4);
            SlotT* s2 = tmp + p.k;
            if (CAP_S1 < 64 || c.region_a_bytes < SEL_WORDS * 4 + (p.k + CAP_S2) * (uint32_t)sizeof(SlotT)) legacy = true;   // launch-uniform
            if (!legacy) {
                const int rb_all = bits_for(r_max2 - x_lo2), sh1 = max(rb_all - SEL_BITS, 0);
                const int sh2 = max(bits_for(misc[S_SUMW]) + rb_all - SEL_BITS, 0);
                auto comp_of = [&](SlotT s) -> unsigned long long { return ((unsigned long long)num_of(s) << rb_all) | (unsigned long long)((uint32_t)(s >> NB) - x_lo2); };
                uint32_t b1 = 0, r1 = 0;
                if (mcut) { hist_search<BLOCK>(hist, p.m, misc); b1 = misc[S_SELD]; r1 = misc[S_SELR]; }
                for (uint32_t i0 = 0; i0 < sslots; i0 += 8 * BLOCK) {   // the scan
                    SlotT sv[8]; unsigned long long bm[8]; uint32_t total = 0, edges = 0;
#pragma unroll
                    for (int u = 0; u < 8; ++u) { const uint32_t i = i0 + u * BLOCK + tid; sv[u] = i < sslots ? stab[i] : SEMPTY; }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const bool valid = sv[u] != SEMPTY;
                        const uint32_t rb = ((uint32_t)(sv[u] >> NB) - x_lo2) >> sh1;
                        bm[u] = __ballot(valid && (!mcut || rb > b1)); total += (uint32_t)__popcll(bm[u]);
                        edges |= (mcut && valid && rb == b1) ? 1u << u : 0u;
                    }
                    if (__ballot(edges != 0) != 0ull) {   // rare: some slot of this batch lies in the boundary bin
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const bool edge = (edges >> u) & 1u;
                            const uint32_t at = wave_append(edge, (uint32_t*)&misc[S_L1]);
                            if (edge && at < CAP_S1) s1[at] = sv[u];
                        }
                    }
                    uint32_t base = 0;
                    if (total) { if (lane == 0) base = atomicAdd((uint32_t*)&misc[S_NB], total); base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base); }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if ((bm[u] >> lane) & 1ull) dl[base + (uint32_t)__popcll(bm[u] & ((1ull << lane) - 1ull))] = sv[u];
                        base += (uint32_t)__popcll(bm[u]);
                    }
                }
                __syncthreads();
                SRN_TICK(2);
                const uint32_t n1 = misc[S_L1];
                if (n1 > CAP_S1) legacy = true;   // block-uniform
                else {
                    for (uint32_t t = tid; t < n1; t += BLOCK) {   // the r1 most recent sessions of the boundary bin
                        const SlotT v = s1[t]; const uint32_t my = (uint32_t)(v >> NB); uint32_t above = 0;
                        for (uint32_t j = 0; j < n1; ++j) above += (uint32_t)(s1[j] >> NB) > my;
                        if (above < r1) dl[atomicAdd((uint32_t*)&misc[S_NB], 1u)] = v;
                    }
                    if (kcut) for (uint32_t i = tid; i < SEL_WORDS; i += BLOCK) hist2[i] = 0;   // (the table is dead)
                    __syncthreads();
                    const uint32_t nd = misc[S_NB];   // == Cm
                    uint32_t b2 = 0, r2 = 0;
                    if (kcut) {
                        for (uint32_t i = tid; i < nd; i += BLOCK) atomicAdd(&hist2[sel_word((uint32_t)(comp_of(dl[i]) >> sh2))], 1u);
                        __syncthreads();
                        hist_search<BLOCK>(hist2, p.k, misc); b2 = misc[S_SELD]; r2 = misc[S_SELR];
                        SRN_TICK(3);
                        for (uint32_t i0 = 0; i0 < nd; i0 += 8 * BLOCK) {   // 8 per lane, one atomic per wave and list
                            SlotT sv[8]; unsigned long long ba[8], be[8]; uint32_t ta = 0, te = 0;
#pragma unroll
                            for (int u = 0; u < 8; ++u) { const uint32_t i = i0 + u * BLOCK + tid; sv[u] = i < nd ? dl[i] : SEMPTY; }
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                const bool in = i0 + u * BLOCK + tid < nd;
                                const uint32_t bb = (uint32_t)(comp_of(sv[u]) >> sh2);
                                ba[u] = __ballot(in && bb > b2); be[u] = __ballot(in && bb == b2);
                                ta += (uint32_t)__popcll(ba[u]); te += (uint32_t)__popcll(be[u]);
                            }
                            uint32_t basea = 0, basee = 0;
                            if (lane == 0) { if (ta) basea = atomicAdd((uint32_t*)&misc[S_NK], ta); if (te) basee = atomicAdd((uint32_t*)&misc[S_L2], te); }   // (S_NK: S_NB is still being read as |D|)
                            basea = (uint32_t)__builtin_amdgcn_readfirstlane((int)basea); basee = (uint32_t)__builtin_amdgcn_readfirstlane((int)basee);
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                if ((ba[u] >> lane) & 1ull) tmp[basea + (uint32_t)__popcll(ba[u] & ((1ull << lane) - 1ull))] = sv[u];
                                if ((be[u] >> lane) & 1ull) { const uint32_t at2 = basee + (uint32_t)__popcll(be[u] & ((1ull << lane) - 1ull)); if (at2 < CAP_S2) s2[at2] = sv[u]; }
                                basea += (uint32_t)__popcll(ba[u]); basee += (uint32_t)__popcll(be[u]);
                            }
                        }
                        __syncthreads();
                        const uint32_t n2 = misc[S_L2];
                        if (n2 <= 128) {
                            for (uint32_t t = tid; t < n2; t += BLOCK) {   // the r2 best of the boundary bin
                                const SlotT v = s2[t]; const unsigned long long my = comp_of(v); uint32_t above = 0;
                                for (uint32_t j = 0; j < n2; ++j) above += comp_of(s2[j]) > my;
                                if (above < r2) tmp[atomicAdd((uint32_t*)&misc[S_NK], 1u)] = v;
                            }
                        } else {   // a crowded bin: radix select on the composite's remaining low bits (dense list: cheap passes)
                            const unsigned long long low = (1ull << sh2) - 1ull;
                            const unsigned long long thr = block_select_desc<BLOCK, unsigned long long>(
                                [&](uint32_t i, unsigned long long& key) { key = comp_of(s2[i]) & low; return true; }, n2, sh2, r2, hist2, misc);
                            for (uint32_t t = tid; t < n2; t += BLOCK) { const SlotT v = s2[t]; if ((comp_of(v) & low) >= thr) tmp[atomicAdd((uint32_t*)&misc[S_NK], 1u)] = v; }
                        }
                        __syncthreads();
                    }
                    if (!legacy) {   // publish: D (no k-cut) or tmp -> neighbour list in region B, its global copy, debug dump
                        const uint32_t kk = kcut ? misc[S_NK] : misc[S_NB];
                        for (uint32_t i = tid; i < kk; i += BLOCK) {
                            const SlotT v = kcut ? tmp[i] : dl[i];
                            if (STAGE == 1) ((SlotT*)sh.cand)[(size_t)q * p.m + i] = v; else { if (kcut) nbl[i] = v; if (STAGE == 0) nb_spill[i] = v; }
                            if (STAGE == 0 && dump_nb) { p.nb_rank[(size_t)q * p.k + i] = (uint32_t)(v >> NB); p.nb_num[(size_t)q * p.k + i] = num_of(v); }
                        }
                        if (kcut) { __syncthreads(); if (tid == 0) misc[S_NB] = kk; }   // every thread has read |D| long ago
                    }
                }
            }
            if (legacy) {   // start over with the radix selects
                __syncthreads();
                for (uint32_t i = tid; i < SEL_WORDS; i += BLOCK) hist[i] = 0;
                if (tid == 0) misc[S_NB] = 0;
                __syncthreads();
            }
        }
        if (legacy) {
        if (Call > p.m) {
            tau = x_lo2 + block_select_desc<BLOCK, uint32_t>(
                [&](uint32_t i, uint32_t& key) { const SlotT s = stab[i]; key = (uint32_t)(s >> NB) - x_lo2; return s != SEMPTY; },
                sslots, bits_for(r_max2 - x_lo2), p.m, hist, misc);
        }
        const int rbits = bits_for(r_max2 - tau);
        unsigned long long kappa = 0;   // composite threshold: (num << rbits) | (rank - tau)
        if (STAGE != 1 && Cm > p.k) {   // (a shard's stage A keeps all of its <= m candidates: the k-cut is global)
            const int nbits = rbits + bits_for(STAGE == 2 ? L * (L + 1) / 2 : misc[S_SUMW]);
            auto comp = [&](uint32_t i, auto& key) {
                const SlotT s = stab[i]; const uint32_t r = (uint32_t)(s >> NB);
                key = ((decltype(key + 0))num_of(s) << rbits) | (r - tau);
                return s != SEMPTY && r >= tau; };
            if (nbits <= 32) kappa = block_select_desc<BLOCK, uint32_t>(comp, sslots, nbits, p.k, hist, misc);
            else kappa = block_select_desc<BLOCK, unsigned long long>(comp, sslots, nbits, p.k, hist, misc);
        }
        for (uint32_t i0 = 0; i0 < sslots; i0 += 8 * BLOCK) {
            SlotT sv[8]; unsigned long long bm[8]; uint32_t total = 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) { const uint32_t i = i0 + u * BLOCK + tid; sv[u] = i < sslots ? stab[i] : SEMPTY; }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t r = (uint32_t)(sv[u] >> NB);
                const bool sel = sv[u] != SEMPTY && r >= tau && ((((unsigned long long)num_of(sv[u])) << rbits) | (r - tau)) >= kappa;
                bm[u] = __ballot(sel); total += (uint32_t)__popcll(bm[u]);
            }
            uint32_t base = 0;
            if (total) { if (lane == 0) base = atomicAdd((uint32_t*)&misc[S_NB], total); base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base); }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if ((bm[u] >> lane) & 1ull) {
                    const uint32_t at = base + (uint32_t)__popcll(bm[u] & ((1ull << lane) - 1ull));
                    if (STAGE == 1) ((SlotT*)sh.cand)[(size_t)q * p.m + at] = sv[u]; else { nbl[at] = sv[u]; if (STAGE == 0) nb_spill[at] = sv[u]; }
                    if (STAGE == 0 && dump_nb) { p.nb_rank[(size_t)q * p.k + at] = (uint32_t)(sv[u] >> NB); p.nb_num[(size_t)q * p.k + at] = num_of(sv[u]); }
                }
                base += (uint32_t)__popcll(bm[u]);
            }
        }
        }
        __syncthreads();
        K = misc[S_NB];
        SRN_TICK(4);
        GEN_PRIO(100);
        if constexpr (STAGE == 1) { if (tid == 0) sh.cand_cnt[q] = K; continue; }
        if constexpr (STAGE == 2) {   // same neighbour order on every shard: sort by the (unique) packed value, descending
            uint32_t n2 = 2; while (n2 < K) n2 <<= 1;
            for (uint32_t i = K + tid; i < n2; i += BLOCK) nbl[i] = 0;
            __syncthreads();
            block_sort_slots<BLOCK, SlotT>(nbl, n2);
            for (uint32_t i = tid; i < K; i += BLOCK) ((SlotT*)sh.nb)[(size_t)q * p.k + i] = nbl[i];
            if (tid == 0) { sh.nb_cnt[q] = K; sh.minpos[(size_t)q * (p.k + 1) + p.k] = cur_idx != kNone ? (int)ix.meta[cur_idx].attr : MINPOS_NONE; }
        }
        } else {   // STAGE == 3: the neighbours and their first-match positions arrive from stage B + all-reduce
            K = sh.nb_cnt[q];
            if (K == 0xFFFFFFFFu) { if (tid == 0) p.out_counts[q] = 0xFFFFFFFFu; continue; }
        }

        // ---- phase 3: neighbour rows -> item accumulators --------------------------------
        // A wave takes 64 neighbours at a time, ONE PER LANE: the lane reads its neighbour's row with 16-byte loads and
        // applies the row's weight w10 * num to every element itself -- no cross-lane traffic, no dependent LDS round
        // trips in the loop.  Two walks over the rows:
        //   walk A  idx < H (the most popular items): exact, one LDS add into the direct-mapped word;  every other
        //           element adds max(w, 0) into sketch[hash(idx)] -- an UPPER BOUND of the item's accumulator, again a
        //           fire-and-forget LDS add
        //   -- phase 4a: exact top-n of the direct-mapped items -> threshold score --
        //   walk B  an element reaches the exact hash table only if its sketch word could still beat the threshold
        //           (same integer floor as the top-n pre-filter); all elements of an item share the sketch word, so
        //           an item is either accumulated completely or not at all.  With no threshold (fewer than n valid
        //           direct-mapped items, or a non-positive threshold score) every element is inserted.
        // If the hash table overflows, the item space is split into hash partitions handled one after the other.
        const double denom = (double)(10u * U);
        const bool business = (p.flags & SRN_FLAG_BUSINESS_LOGIC) != 0;
        uint32_t cur_attr = SRN_ATTR_NONE;
        if (business) {
            if (STAGE == 3) { const int v = sh.minpos[(size_t)q * (p.k + 1) + p.k]; if (v != MINPOS_NONE) cur_attr = (uint32_t)v; }   // owner shard's byte, all-reduced
            else if (use_prep) cur_attr = hd.cur_attr;   // (looked up by the prep kernel; item-sharded lists mode: the owner shard's byte)
            else if (cur_idx != kNone) cur_attr = ix.meta[cur_idx].attr;
        }
        const uint32_t n_out = p.how_many;
        const bool use_filter = SK != 0 && p.stats == nullptr;   // the debug counters need every distinct item in the table
        // the (<= 4) most common case keeps the evolving items in registers; longer sessions scan LDS
        const uint32_t qv0 = q_idx[0], qv1 = L > 1 ? q_idx[1] : kNone, qv2 = L > 2 ? q_idx[2] : kNone, qv3 = L > 3 ? q_idx[3] : kNone;
        auto match_pos = [&](uint32_t it) -> uint32_t {   // reverse position of `it` in the evolving session, or 0xFFFF
            if (it == qv0) return 0; if (it == qv1) return 1; if (it == qv2) return 2; if (it == qv3) return 3;
            for (uint32_t pp = 4; pp < L; ++pp) if (q_idx[pp] == it) return pp;
            return 0xFFFFu; };
        // Walk the rows of this wave's neighbour groups, one row per lane: per_row(j, num, for_row) is called once per
        // group; for_row(g8) feeds the lane's row to g8(it[8]) eight items at a time (EMPTY32 = none), as often as asked.
        // Memory schedule per group, all loads unconditional (inactive lanes read slot 0) and in this order: second half
        // of this group's slots (same line as the first half), first 8 items of this group's overflow rows, first half of
        // the NEXT group's slots.  Loads return in order, so waiting for this group's data never waits for the next
        // group's HBM miss, which has a whole group's work to land.
        auto walk_rows = [&](auto nb_at, auto&& per_row) -> uint32_t {
            uint32_t isum = 0;
            if (K == 0) return 0u;
            auto slot_of = [&](uint32_t j, uint32_t& num) -> size_t {   // lanes past the last neighbour read the all-EMPTY slot n_kept
                const SlotT s = nb_at(min(j, K - 1));
                num = j < K ? (uint32_t)(s & num_mask) : 0u;
                return j < K ? (size_t)(uint32_t)(s >> NB) : (size_t)n_kept; };
            constexpr uint32_t GSTEP = NWAVES * 64;
            uint32_t num, nnum;
            // Item shards keep 16-byte FRAGMENT slots (DeviceIndex::row_frag): one quad {len, i0, i1, i2}, or {len, offset, i0, i1} + items 2.. in row_ext.
            // They walk through the same code as a 64-byte slot whose words 4..15 are empty: `inl` items inline before the overflow area.
            const bool frag = ix.row_frag != 0u;   // (launch-uniform)
            const uint32_t qs = frag ? 1u : 4u, inl = frag ? 2u : 14u;
            const RowQuad EQ{EMPTY32, EMPTY32, EMPTY32, EMPTY32};
            size_t r = slot_of(wave * 64 + lane, num), nr;
            RowQuad a = ix.row_slots[qs * r], b = frag ? EQ : ix.row_slots[4 * r + 1], na, nb;
            for (uint32_t g0 = wave * 64; g0 < K; g0 += GSTEP) {
                // unused slot words are EMPTY32 in memory, so only a long row's word 1 (its overflow offset) needs masking;
                // short rows read the EMPTY32 words at the start of row_ext
                const uint32_t len = a.x;
                const bool big = len > inl + 1u;                 // word 1 = offset of items inl.. in row_ext, items 0..inl-1 in the words from 2 on
                const RowQuad c4 = frag ? EQ : ix.row_slots[4 * r + 2], d4 = frag ? EQ : ix.row_slots[4 * r + 3];
                const uint32_t* ext = ix.row_ext + (big ? a.y : 0u);
                const RowVec e0 = *reinterpret_cast<const RowVec*>(ext), e1 = *reinterpret_cast<const RowVec*>(ext + 4);
                nr = slot_of(g0 + GSTEP + lane, nnum);
                na = ix.row_slots[qs * nr]; nb = frag ? EQ : ix.row_slots[4 * nr + 1];
                isum += len;
                auto for_row = [&](auto&& g8) {
                    { const uint32_t it7[7] = {big ? EMPTY32 : a.y, a.z, a.w, b.x, b.y, b.z, b.w};   // words 1..7
                      g8(it7); }
                    uint32_t it[8];
                    if (!frag) {
                        if (__ballot(len > 7) == 0ull) return;
                        it[0] = c4.x; it[1] = c4.y; it[2] = c4.z; it[3] = c4.w; it[4] = d4.x; it[5] = d4.y; it[6] = d4.z; it[7] = d4.w;   // words 8..15
                        g8(it);
                    }
                    if (__ballot(big) == 0ull) return;
                    it[0] = e0.x; it[1] = e0.y; it[2] = e0.z; it[3] = e0.w; it[4] = e1.x; it[5] = e1.y; it[6] = e1.z; it[7] = e1.w;   // items inl .. inl + 7
#pragma unroll
                    for (uint32_t x = 0; x < 8; ++x) if (!big || inl + x >= len) it[x] = EMPTY32;
                    g8(it);
                    for (uint32_t t = inl + 8u; __ballot(big && t < len) != 0ull; t += 8) {
#pragma unroll
                        for (int x = 0; x < 8; ++x) it[x] = EMPTY32;
                        if (big && t < len) { const RowVec v = *reinterpret_cast<const RowVec*>(ext + (t - inl)); it[0] = v.x; it[1] = v.y; it[2] = v.z; it[3] = v.w; }
                        if (big && t + 4 < len) { const RowVec v = *reinterpret_cast<const RowVec*>(ext + (t - inl) + 4); it[4] = v.x; it[5] = v.y; it[6] = v.z; it[7] = v.w; }
#pragma unroll
                        for (uint32_t x = 1; x < 8; ++x) if (t + x >= len) it[x] = EMPTY32;
                        g8(it);
                    } };
                per_row(g0 + lane, num, for_row);
                num = nnum; r = nr; a = na; b = nb;
            }
            return isum; };
        // 10 * linear_score(first match) * numerator of lane's neighbour j (exact, Q3); publishes / reads the first-match
        // position in the sharded stages
        auto row_weight = [&](uint32_t j, uint32_t num, auto&& for_row) -> int {
            const bool active = j < K;
            if (MASKS) { const int mp = num ? __ffs((int)num) - 1 : MINPOS_NONE;   // lowest set position = first match (Q4)
                         if (STAGE == 2 && active) sh.minpos[(size_t)q * (p.k + 1) + j] = mp;
                         return (mp < 99 ? 10 - (mp + 1) : 0) * (int)wlut[num]; }
            if (STAGE == 3) { const int mp = active ? sh.minpos[(size_t)q * (p.k + 1) + j] : MINPOS_NONE;
                              return (mp < 99 ? 10 - (mp + 1) : 0) * (int)num; }
            uint32_t mp = 0xFFFFu;   // first-match position against the FULL row (Q4)
            for_row([&](const auto& it) {
                constexpr int N = sizeof(it) / sizeof(it[0]);
#pragma unroll
                for (int x = 0; x < N; ++x) if (it[x] != EMPTY32) mp = min(mp, match_pos(it[x])); });
            if (STAGE == 0 && active && mp == 0xFFFFu) misc[S_ERR] = 1;   // inconsistent index (reference: unwrap panic, mod.rs:138)
            if (STAGE == 2 && active) sh.minpos[(size_t)q * (p.k + 1) + j] = mp == 0xFFFFu ? MINPOS_NONE : (int)mp;
            const int p1 = (int)mp + 1;
            return (p1 < 100 ? 10 - p1 : 0) * (int)num; };
        auto nb_lds = [&](uint32_t j) -> SlotT { return STAGE == 3 ? ((const SlotT*)sh.nb)[(size_t)q * p.k + j] : nbl[j]; };
        auto nb_glb = [&](uint32_t j) -> SlotT { return STAGE == 3 ? ((const SlotT*)sh.nb)[(size_t)q * p.k + j] : nb_spill[j]; };

        // Rows differ in length and a wave pays for its longest lane, so (MASKS, fused kernel) the rows are walked in three
        // rounds of like work: (i) every neighbour's first 7 items; neighbours with more are queued, (ii) items 7..14 of the
        // queued rows; rows that continue in the overflow area are queued again, (iii) those.  `queue` has one slot per
        // neighbour (rounded up to whole groups of 64); a wave only writes queue slots of its own groups that it has already
        // consumed, so the queue may be the neighbour list itself.  (The row weight is a function of the slot's position
        // set, so a queued slot carries everything.)  elem(it[N], w) sees N row elements of one lane's row.
        auto walk_rounds = [&](auto nb_at, SlotT* queue, auto&& elem) -> uint32_t {
            uint32_t isum = 0;
                // Rows differ in length and a wave pays for its longest lane, so the rows are walked in three rounds of
            // like work: (i) every neighbour's first 7 items; neighbours with more are queued, (ii) items 7..14 of the
            // queued rows; rows that continue in the overflow area are queued again, (iii) those.  A wave queues into
            // the neighbour-list slots of its own groups that it has already consumed: no extra LDS.  (The row weight
            // is a function of the slot's position set, so a queued slot carries everything.)
            constexpr uint32_t GSTEP = NWAVES * 64;
            auto qpos = [&](uint32_t pq) -> uint32_t { return ((wave + NWAVES * (pq >> 6)) << 6) + (pq & 63u); };
            auto weight_of = [&](uint32_t num) -> int {
                const int mp = num ? __ffs((int)num) - 1 : MINPOS_NONE;
                return (mp < 99 ? 10 - (mp + 1) : 0) * (int)wlut[num]; };
            const unsigned long long lt = (1ull << lane) - 1ull;
            uint32_t qn = 0;
            if ((uint32_t)wave * 64 < K) {
                SlotT sv = nb_at(min((uint32_t)(wave * 64 + lane), K - 1)), nsv = 0;
                size_t r = (uint32_t)(wave * 64 + lane) < K ? (size_t)(uint32_t)(sv >> NB) : (size_t)n_kept;
                RowQuad a = ix.row_slots[4 * r], b = ix.row_slots[4 * r + 1], na, nb;
                for (uint32_t g0 = wave * 64; g0 < K; g0 += GSTEP) {   // (i)
                    const uint32_t jn = g0 + GSTEP + lane;
                    nsv = nb_at(min(jn, K - 1));
                    const size_t nr = jn < K ? (size_t)(uint32_t)(nsv >> NB) : (size_t)n_kept;
                    na = ix.row_slots[4 * nr]; nb = ix.row_slots[4 * nr + 1];
                    const uint32_t len = a.x;   // (0 for the idle lanes' empty slot)
                    const int w = weight_of((uint32_t)(sv & num_mask));
                    isum += len;
                    const uint32_t it7[7] = {len > 15 ? EMPTY32 : a.y, a.z, a.w, b.x, b.y, b.z, b.w};
                    elem(it7, w);
                    const bool more = len > 7;
                    const unsigned long long bm = __ballot(more);
                    if (more) queue[qpos(qn + (uint32_t)__popcll(bm & lt))] = sv;   // (all lanes hold their slot in a register by now)
                    qn += (uint32_t)__popcll(bm);
                    sv = nsv; a = na; b = nb;
                }
            }
            uint32_t qn2 = 0;
            for (uint32_t p0 = 0; p0 < qn; p0 += 64) {   // (ii)
                const bool act = p0 + lane < qn;
                const SlotT sv = queue[qpos(min(p0 + lane, qn - 1))];
                const size_t r = act ? (size_t)(uint32_t)(sv >> NB) : (size_t)n_kept;
                const RowQuad c4 = ix.row_slots[4 * r + 2], d4 = ix.row_slots[4 * r + 3];
                const uint32_t len = ix.row_slots[4 * r].x;
                const int w = weight_of((uint32_t)(sv & num_mask));
                const uint32_t it8[8] = {c4.x, c4.y, c4.z, c4.w, d4.x, d4.y, d4.z, d4.w};
                elem(it8, w);
                const bool more = len > 15;
                const unsigned long long bm = __ballot(more);
                if (more) queue[qpos(qn2 + (uint32_t)__popcll(bm & lt))] = sv;
                qn2 += (uint32_t)__popcll(bm);
            }
            for (uint32_t p0 = 0; p0 < qn2; p0 += 64) {   // (iii)
                const bool act = p0 + lane < qn2;
                const SlotT sv = queue[qpos(min(p0 + lane, qn2 - 1))];
                const size_t r = act ? (size_t)(uint32_t)(sv >> NB) : (size_t)n_kept;
                const RowQuad a = ix.row_slots[4 * r];
                const uint32_t len = a.x; const bool big = len > 15;
                const uint32_t* ext = ix.row_ext + (big ? a.y : 0u);
                const int w = weight_of((uint32_t)(sv & num_mask));
                // the first 24 overflow items (rows of <= 38) are requested together: one HBM latency for the round, not one per
                // 8 items -- few rows come here, so the round is nothing but latency
                constexpr int MAXV = 6;
                RowVec pv[MAXV];
#pragma unroll
                for (int x = 0; x < MAXV; ++x) {
                    pv[x] = RowVec{EMPTY32, EMPTY32, EMPTY32, EMPTY32};
                    if (big && 14u + 4u * x < len) pv[x] = *reinterpret_cast<const RowVec*>(ext + 4 * x);
                }
#pragma unroll
                for (int x = 0; x < MAXV; x += 2) {
                    const uint32_t t = 14u + 4u * x;
                    if (__ballot(big && t < len) == 0ull) break;
                    uint32_t it8[8] = {pv[x].x, pv[x].y, pv[x].z, pv[x].w, pv[x + 1].x, pv[x + 1].y, pv[x + 1].z, pv[x + 1].w};
#pragma unroll
                    for (uint32_t y = 0; y < 8; ++y) if (!big || t + y >= len) it8[y] = EMPTY32;
                    elem(it8, w);
                }
                for (uint32_t t = 14 + 4 * MAXV; __ballot(big && t < len) != 0ull; t += 8) {
                    uint32_t it8[8];
#pragma unroll
                    for (int x = 0; x < 8; ++x) it8[x] = EMPTY32;
                    if (big && t < len) { const RowVec v = *reinterpret_cast<const RowVec*>(ext + (t - 14)); it8[0] = v.x; it8[1] = v.y; it8[2] = v.z; it8[3] = v.w; }
                    if (big && t + 4 < len) { const RowVec v = *reinterpret_cast<const RowVec*>(ext + (t - 14) + 4); it8[4] = v.x; it8[5] = v.y; it8[6] = v.z; it8[7] = v.w; }
#pragma unroll
                    for (uint32_t x = 1; x < 8; ++x) if (t + x >= len) it8[x] = EMPTY32;
                    elem(it8, w);
                }
            }
            return isum; };
        if constexpr (STAGE == 2) {   // stage B: first-match positions only
            walk_rows(nb_lds, [&](uint32_t j, uint32_t num, auto&& for_row) { row_weight(j, num, for_row); });
            continue;
        }
        for (uint32_t i = tid; i < H + SK; i += BLOCK) hot[i] = 0;   // direct-mapped words and the sketch are adjacent
        if (tid == 0) { misc[S_CCNT] = 0; misc[S_HAVE_T] = 0; misc[S_SORTED] = 0; misc[S_I] = 0; misc[S_LIVE] = 0; misc[S_ICNT] = 0; }
        phase_sync<GLOBAL_TABLES>();
        GEN_PRIO(10);
        SRN_TICK(8);
        {   // walk A
            // one LDS add per element: idx < H -> its direct-mapped word, anything else -> sketch word idx mod SK (sketch = hot + H).
            // A sketch word sums max(w, 0) of everything that lands in it: an upper bound of each of its items.  A direct-mapped
            // word is the exact sum; where weights can be <= 0 (no MASKS) it also carries a touch count above bit SB so that a
            // touched item with sum 0 is still seen.  Without a sketch only idx < H adds.
            auto add_items = [&](const auto& it, uint32_t whot, uint32_t wsk) {
                constexpr int N = sizeof(it) / sizeof(it[0]);
#pragma unroll
                for (int x = 0; x < N; ++x) {
                    const bool is_hot = it[x] < H;
                    if (it[x] != EMPTY32 && (is_hot || SK)) atomicAdd(&hot[min(it[x], H + (it[x] & SKM))], is_hot ? whot : wsk);
                } };
            // The word index needs no select: min(idx, H + idx mod SK) is idx itself below H and a word of the sketch otherwise
            // (idx in [H, H + SK) keeps its own position).  MASKS: weights are positive, plain sums everywhere, one value for both kinds
            // Empty element slots are not masked out (that costs an exec save / restore and two taken branches per element): they
            // add into a per-lane scratch word instead -- the last 64 words of region A, which nobody reads before they are cleared.
            uint32_t* const idle_word = wb + (wwords - 64) + lane;
            auto add_items_pos = [&](const auto& it, uint32_t wv) {
                constexpr int N = sizeof(it) / sizeof(it[0]);
#pragma unroll
                for (int x = 0; x < N; ++x) { uint32_t* at = &hot[min(it[x], H + (it[x] & SKM))]; atomicAdd(it[x] != EMPTY32 ? at : idle_word, wv); } };
            uint32_t isum = 0;
            auto walk_a_rows = [&]() -> uint32_t {
                return walk_rows(nb_lds, [&](uint32_t j, uint32_t num, auto&& for_row) {
                    const int w = row_weight(j, num, for_row);
                    const uint32_t wsk = (uint32_t)max(w, 0), whot = MASKS ? (uint32_t)w : (1u << SB) + (uint32_t)w;
                    for_row([&](const auto& it) { add_items(it, whot, wsk); }); }); };
            if constexpr (MASKS && STAGE == 0) {
                if (ix.row_frag != 0u) isum = walk_a_rows();   // (an item shard in lists mode: 16-byte fragment slots, walk_rows reads both kinds)
                else
                if (SK && !GLOBAL_TABLES) isum = walk_rounds(nb_lds, nbl, [&](const auto& it, int w) { add_items_pos(it, (uint32_t)w); });   // (region A in LDS)
                else isum = walk_rounds(nb_lds, nbl, [&](const auto& it, int w) { add_items(it, (uint32_t)w, (uint32_t)w); });   // (global-table pass: the sketch words need their sums too)
            } else isum = walk_a_rows();
            isum = wave_sum(isum);
            if (lane == 0 && p.stats && isum) atomicAdd((uint32_t*)&misc[S_I], isum);
        }
        phase_sync<GLOBAL_TABLES>();
        GEN_PRIO(1);
        SRN_TICK(9);
        uint32_t d_total = 0;
        if (p.stats && H) {   // debug counters only: distinct items = touched direct-mapped entries + hash inserts
            uint32_t touched = 0; for (uint32_t i = tid; i < H; i += BLOCK) touched += hot[i] != 0;
            touched = wave_sum(touched); if (lane == 0 && touched) atomicAdd((uint32_t*)&misc[S_ICNT], touched);
            __syncthreads();
            d_total = misc[S_ICNT];
            __syncthreads();
        }

        // ---- phase 4: scores, business rules, top-n (one running candidate set) -------------
        // harvest(e_lo, e_hi): entries [e_lo, e_hi) of (direct-mapped words | hash slots) in chunks of BLOCK; the first
        // chunk without a threshold is a sample whose n-th best score becomes one, then all other chunks are swept in
        // one barrier-free round and only candidates beating the threshold are appended.  If the buffer would overflow,
        // the round is redone chunk by chunk with a prune whenever needed (exact).  Leaves the best min(cnt, n) sorted.
        auto acc_floor_of = [&](uint64_t tk) -> int {
            // smallest accumulator that could still reach the threshold: score <= idf_hi * acc / denom, so an item needs
            // acc >= thr * denom / idf_hi; shaved by a relative 1e-9 and one unit so that rounding can only keep more
            return (int)fmin(2147483000.0, fmax(1.0, floor(key_score(tk) * denom / idf_hi * (1.0 - 1e-9)) - 1.0)); };
        // leaves the best min(cnt, n) candidates sorted at the front of the buffer, sets the threshold if n exist
        auto sort_candidates = [&](uint32_t cnt) {
            if (n_out <= 64 && cnt <= 64) {   // one wave sorts in registers, no merge levels
                if (wave == 0) {   // rank by counting: lane j's candidate is broadcast with v_readlane (no LDS crossbar round trips, unlike a
                                   // shuffle network); (key, id rank) pairs are distinct, so the ranks are a permutation
                    const uint64_t kk = lane < (int)cnt ? ckey[lane] : 0; const uint32_t ii = lane < (int)cnt ? cidx[lane] : EMPTY32;
                    const int klo = (int)(uint32_t)kk, khi = (int)(uint32_t)(kk >> 32);
                    uint32_t rank = 0;
                    for (uint32_t j = 0; j < cnt; ++j) {
                        const uint64_t kj = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(khi, (int)j) << 32) | (uint32_t)__builtin_amdgcn_readlane(klo, (int)j);
                        const uint32_t ij = (uint32_t)__builtin_amdgcn_readlane((int)ii, (int)j);
                        rank += cand_better(kj, ij, kk, ii);
                    }
                    if (lane < (int)cnt) { ckey[rank] = kk; cidx[rank] = ii; }
                }
                __syncthreads();
            } else if (n_out <= 64) block_top64<BLOCK>(ckey, cidx, cnt);   // leaves the best min(cnt, 64) sorted at the front
            else {
                uint32_t n2 = 2; while (n2 < cnt) n2 <<= 1;
                for (uint32_t t = cnt + tid; t < n2; t += BLOCK) { ckey[t] = 0; cidx[t] = EMPTY32; }
                __syncthreads();
                block_sort_candidates<BLOCK>(ckey, cidx, n2);
            }
            if (tid == 0) {
                misc[S_SORTED] = min(cnt, n_out);
                if (cnt >= n_out) { misc[S_CCNT] = n_out; misc[S_HAVE_T] = 1; misc[S_TIDX] = cidx[n_out - 1];
                                    misc[S_TKEY_LO] = (uint32_t)ckey[n_out - 1]; misc[S_TKEY_HI] = (uint32_t)(ckey[n_out - 1] >> 32); }
            }
            __syncthreads(); };
        auto harvest = [&](uint32_t e_lo, uint32_t e_hi, bool sort_at_end) {
            const uint32_t n_chunks = (e_hi - e_lo + BLOCK - 1) / BLOCK;
            uint32_t u = 0, ru = (misc[S_HAVE_T] && misc[S_CCNT] * 2 <= CAND_CAP) ? n_chunks : 1u;
            while (u < n_chunks) {
                const uint32_t cnt0 = misc[S_CCNT];
                const bool have_t = misc[S_HAVE_T] != 0;
                const uint64_t tk = ((uint64_t)misc[S_TKEY_HI] << 32) | misc[S_TKEY_LO];
                const uint32_t tix = misc[S_TIDX];
                const bool t_pos = have_t && tk > 0x8000000000000000ull;   // threshold score > 0: non-positive accumulators cannot make it
                const int acc_floor = t_pos ? acc_floor_of(tk) : 0;
                const uint32_t u_end = min(u + ru, n_chunks);
                __syncthreads();   // every wave has read the round's state before any wave appends (and moves S_CCNT)
                // gather first (2 chunks' idf loads in flight per lane), then score + append.  Once the
                // threshold score is positive, an item whose upper bound idf_hi * acc / denom (same operations and
                // rounding as the score, so monotone and safe) is below it is dropped without touching meta[].
                for (uint32_t ub = u; ub < u_end; ub += 2) {   // sub-batches of 2 chunks, no barrier in between
                    uint32_t its[2]; int accs[2]; ItemMeta metas[2];
#pragma unroll
                    for (int x = 0; x < 2; ++x) {
                        its[x] = EMPTY32; accs[x] = 0; metas[x] = ItemMeta{0.0, 0u, 0u};
                        const uint32_t i = e_lo + (ub + x) * BLOCK + tid;
                        if (ub + x < u_end && i < e_hi) {
                            uint32_t it = EMPTY32; int acc = 0;
                            if (i < H) { const uint32_t v = hot[i]; if (v) { it = i; acc = MASKS ? (int)v : (int)(v - (((v + (1u << (SB - 1))) >> SB) << SB)); } }
                            else { it = ikeys[i - H]; acc = iacc[i - H]; }
                            if (it != EMPTY32 && it != cur_idx) {   // Q6
                                const bool hopeless = t_pos & (acc < acc_floor);   // branch-free on purpose (see DESIGN.md hazards)
                                if (!hopeless) { its[x] = it; accs[x] = acc; metas[x] = ix.meta[it]; }
                            }
                        }
                    }
#pragma unroll
                    for (int x = 0; x < 2; ++x) {
                        if (ub + x < u_end) {   // block-uniform
                            const uint32_t tie = metas[x].id_rank;   // ties are broken by ascending public id
                            bool take = false; uint64_t sk = 0;
                            if (its[x] != EMPTY32 && (!business || business_ok(cur_attr, metas[x].attr))) {   // business rules
                                sk = score_key((metas[x].idf > 0.0 ? metas[x].idf : 1.0) * (double)accs[x] / denom);
                                take = !have_t || sk > tk || (sk == tk && tie < tix);
                            }
                            const uint32_t at = wave_append(take, (uint32_t*)&misc[S_CCNT]);
                            if (take) { if (at < CAND_CAP) { ckey[at] = sk; cidx[at] = tie; } else misc[S_COVF] = 1; }
                        }
                    }
                }
                __syncthreads();
                if (misc[S_COVF]) {   // block-uniform: too many survivors for one optimistic round
                    __syncthreads();
                    if (tid == 0) { misc[S_CCNT] = cnt0; misc[S_COVF] = 0; }
                    ru = 1;
                    __syncthreads();
                    continue;
                }
                u = u_end;
                const uint32_t cnt = misc[S_CCNT];
                const bool last = u >= n_chunks;
                // before a single-chunk round the buffer must have room for BLOCK appends (exact path);
                // also sort once as soon as n candidates exist, to get a threshold
                const bool must_prune = cnt + BLOCK > CAND_CAP || (!have_t && cnt >= n_out && cnt > 1);
                if ((must_prune && !last) || (last && sort_at_end && cnt > 1 && cnt != misc[S_SORTED])) sort_candidates(cnt);
                // once a threshold exists and the buffer is at most half full: ONE optimistic round over all the
                // remaining chunks (a round that overflows the buffer is redone chunk by chunk, see above)
                ru = (misc[S_HAVE_T] && misc[S_CCNT] * 2 <= CAND_CAP) ? n_chunks : 1u;
            }
            __syncthreads(); };

        // phase 4a: the direct-mapped items, exactly -> threshold.
        if (H >= (uint32_t)BLOCK && n_out <= 3 * NWAVES) {
            // First chunk (the 512 most popular items) without a block-wide sort: every wave sorts its 64 scores in registers;
            // the worst of the waves' 3rd-best scores has >= 3 * NWAVES >= n candidates at or above it, so it is a valid
            // (and tight) threshold.  Only candidates at or above it are kept at all.
            uint64_t sk = 0; uint32_t tie = EMPTY32; bool valid = false;
            { const uint32_t e = (uint32_t)lane * NWAVES + (uint32_t)wave;   // entries dealt round-robin: every wave sees the same mix of popularity
              const uint32_t v = hot[e];
              if (v && e != cur_idx) {
                  const int acc = MASKS ? (int)v : (int)(v - (((v + (1u << (SB - 1))) >> SB) << SB));
                  const ItemMeta mt = ix.meta[e];
                  if (!business || business_ok(cur_attr, mt.attr)) { sk = score_key((mt.idf > 0.0 ? mt.idf : 1.0) * (double)acc / denom); tie = mt.id_rank; valid = true; }
              } }
            // the wave's 3rd largest key, on the top 32 bits of the score key (a monotone truncation: still a valid bound)
            const uint32_t k32 = valid ? (uint32_t)(sk >> 32) : 0u;
            { uint32_t v = k32, third = 0;
#pragma unroll
              for (int t = 0; t < 3; ++t) {
                  const uint32_t mx = wave_max(v);
                  third = mx;
                  const unsigned long long bal = __ballot(v == mx);
                  if (lane == __ffsll((long long)bal) - 1) v = 0;   // take one holder of the maximum out
              }
              if (lane == 0) cidx[CAND_CAP - NWAVES + wave] = third; }   // (0 if the wave has < 3 candidates)
            __syncthreads();
            uint32_t t32 = 0xFFFFFFFFu;
#pragma unroll
            for (int w = 0; w < NWAVES; ++w) t32 = min(t32, cidx[CAND_CAP - NWAVES + w]);
            const bool take = valid && k32 >= t32;   // at or above the threshold (everything, if there is none)
            const uint32_t at = wave_append(take, (uint32_t*)&misc[S_CCNT]);
            if (take) { ckey[at] = sk; cidx[at] = tie; }
            if (tid == 0 && t32) { misc[S_HAVE_T] = 1; misc[S_TIDX] = EMPTY32; misc[S_TKEY_LO] = 0u; misc[S_TKEY_HI] = t32; }
            __syncthreads();
            if (H > (uint32_t)BLOCK) harvest(BLOCK, H, false);   // (no sort yet: the sample's threshold serves walk B, one sort at the very end)
        } else if (H) harvest(0, H, true);
        SRN_TICK(10);

        // ---- walk B + phase 4b, per item partition ------------------------------------------
        uint32_t parts = 1, part = 0;
        bool failed = false;
        for (;;) {
            // what the threshold lets through (block-uniform; all waves read it before anyone can move it again)
            const bool have_t = misc[S_HAVE_T] != 0;
            const uint64_t tk = ((uint64_t)misc[S_TKEY_HI] << 32) | misc[S_TKEY_LO];
            const bool filt = use_filter && have_t && tk > 0x8000000000000000ull;
            const uint32_t floor_b = filt ? (uint32_t)acc_floor_of(tk) : 0u;
            if (filt && part == 0 && parts == 1) {   // no sketch word reaches the floor: no other item can make the top n
                uint32_t live = 0;
                for (uint32_t i = tid; i < SK; i += BLOCK) live += sketch[i] >= floor_b;
                live = wave_sum(live);
                if (lane == 0 && live) atomicAdd((uint32_t*)&misc[S_LIVE], live);
                __syncthreads();
                if (misc[S_LIVE] == 0) break;
            }
            for (uint32_t i = tid; i < c.item_slots; i += BLOCK) { ikeys[i] = EMPTY32; iacc[i] = 0; }
            if (tid == 0) { misc[S_OVF] = 0; misc[S_ICNT] = 0; }
            phase_sync<GLOBAL_TABLES>();
            SRN_TICK(11);
            {
                uint32_t fresh = 0; bool ovf = false;
                auto insert_items = [&](const auto& it, int w) {
                    constexpr int N = sizeof(it) / sizeof(it[0]);
                    bool go[N]; bool any = false;
#pragma unroll
                    for (int x = 0; x < N; ++x) {   // (the sketch word is read for every slot, empty or not: no exec juggling; the index is always in range)
                        const uint32_t ub = filt ? hot[min(it[x], H + (it[x] & SKM))] : 0xFFFFFFFFu;
                        go[x] = (it[x] != EMPTY32) & (it[x] >= H) & (ub >= floor_b) & (parts == 1 || hash_part(it[x], parts) == part);
                        any |= go[x];
                    }
                    if (__ballot(any) == 0ull) return;   // the usual case: nothing of these 8 slots can reach the top n
#pragma unroll
                    for (int x = 0; x < N; ++x) {
                        if (go[x]) { const int res = item_insert(ikeys, iacc, inb, it[x], w); if (res < 0) ovf = true; else fresh += (uint32_t)res; }
                    } };
                bool rounds = false;
                if constexpr (MASKS && STAGE == 0 && !GLOBAL_TABLES) rounds = ix.row_frag == 0u && (size_t)(K + 64) * sizeof(SlotT) <= (size_t)H * 4;   // the direct-mapped words are dead: queue there
                if (rounds) { if constexpr (MASKS && STAGE == 0 && !GLOBAL_TABLES) walk_rounds(nb_glb, (SlotT*)hot, insert_items); }
                else walk_rows(nb_glb, [&](uint32_t j, uint32_t num, auto&& for_row) {
                    const int w = row_weight(j, num, for_row);
                    for_row([&](const auto& it) { insert_items(it, w); }); });
                fresh = wave_sum(fresh);
                if (lane == 0 && fresh) atomicAdd((uint32_t*)&misc[S_ICNT], fresh);
                if (ovf) misc[S_OVF] = 1;
            }
            phase_sync<GLOBAL_TABLES>();
            SRN_TICK(12);
            if (misc[S_OVF]) {   // block-uniform: split the item space finer; partitions already harvested stay as they are
                __syncthreads();   // (hash_part(it, 2 * parts) is 2 * hash_part(it, parts) or that + 1: a refinement)
                if (parts >= MAX_ITEM_PASSES || GLOBAL_TABLES) { failed = true; break; }
                parts *= 2; part *= 2;
                continue;
            }
            d_total += misc[S_ICNT];
            harvest(H, H + c.item_slots, true);
            SRN_TICK(13);
            if (++part >= parts) break;
        }
        if (!failed) {   // (walk B skipped, or nothing came of it: the candidates may still be unsorted)
            const uint32_t cnt = misc[S_CCNT];
            if (cnt > 1 && cnt != misc[S_SORTED]) sort_candidates(cnt);
        }
        if (failed) {
            if (GLOBAL_TABLES || (STAGE == 3 && aux.retry_list == nullptr)) { if (tid == 0) { p.out_counts[q] = 0xFFFFFFFFu; if (p.stats) p.stats[(size_t)q * 8 + 7] = 3; } }
            else if (tid == 0) aux.retry_list[atomicAdd(aux.retry_cnt, 1u)] = q;
            continue;
        }
        const uint32_t n_res = min(misc[S_CCNT], n_out);
        if (tid < n_res) {
            p.out_ids[(size_t)q * n_out + tid] = ix.id_sorted[cidx[tid]];
            p.out_scores[(size_t)q * n_out + tid] = key_score(ckey[tid]);
        }
        if (tid == 0) {
            p.out_counts[q] = n_res;
            if (p.nb_cnt) p.nb_cnt[q] = K;
            if (p.stats) { uint32_t* st = p.stats + (size_t)q * 8;
                st[0] = P; st[1] = Cm; st[2] = K; st[3] = misc[S_I]; st[4] = d_total; st[5] = n_res; st[6] = L;
                st[7] = misc[S_ERR] ? 4u : (GLOBAL_TABLES ? 1u : (parts > 1 ? 0x100u * parts : 0u)); }
        }
    }
}


// =====================================================================================
// launchers
// =====================================================================================
template <int BLOCK, bool GLOBAL_TABLES, int STAGE = 0, bool MASKS = false, int WG_PER_CU = 2>
static hipError_t launch_variant(bool slot64, dim3 grid, size_t lds, hipStream_t st, const DeviceIndex& di,
                                 const LaunchParams& p, const KernelCfg& c, const uint32_t* qlist, const uint32_t* qn,
                                 uint32_t* retry_list, uint32_t* retry_cnt, char* gs, unsigned long long gstride, char* spill,
                                 const ShardIO& sh = ShardIO{}) {
#define SRN_LAUNCH(SLOT)                                                                                             \
    do {                                                                                                                  \
        auto kern = vmis_predict_kernel<BLOCK, SLOT, GLOBAL_TABLES, STAGE, MASKS, WG_PER_CU>;                                           \
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);     \
        if (e != hipSuccess) return e;                                                                                    \
        hipLaunchKernelGGL(kern, grid, dim3(BLOCK), lds, st, di, p, c, LaunchAux{qlist, qn, retry_list, retry_cnt, gs, gstride, spill}, sh); \
        return hipGetLastError();                                                                                         \
    } while (0)
    if (!slot64) SRN_LAUNCH(uint32_t);
    SRN_LAUNCH(unsigned long long);
#undef SRN_LAUNCH
}

hipError_t launch_predict(bool masks, bool slot64, bool global_tables, int stage, dim3 grid, size_t lds, hipStream_t st, const DeviceIndex& di,
                          const LaunchParams& p, const KernelCfg& c, const uint32_t* qlist, const uint32_t* qn, uint32_t* retry_list,
                          uint32_t* retry_cnt, char* gscratch, unsigned long long gscratch_stride, char* nb_spill, const ShardIO& sh, int wg_per_cu) {
    if (wg_per_cu == 3) {   // the 80-VGPR build exists for the fused kernel with u32 slots and LDS tables only
        if (global_tables || stage != 0 || slot64) return hipErrorInvalidValue;
        return masks ? launch_variant<kBlock, false, 0, true, 3>(false, grid, lds, st, di, p, c, qlist, qn, retry_list, retry_cnt, gscratch, gscratch_stride, nb_spill, sh)
                     : launch_variant<kBlock, false, 0, false, 3>(false, grid, lds, st, di, p, c, qlist, qn, retry_list, retry_cnt, gscratch, gscratch_stride, nb_spill, sh);
    }
#define SRN_V(G, S, M) launch_variant<kBlock, G, S, M>(slot64, grid, lds, st, di, p, c, qlist, qn, retry_list, retry_cnt, gscratch, gscratch_stride, nb_spill, sh)
    if (global_tables) {   // (the stages' global-table passes exist for numerator slots only: with position sets the shard group takes the lists pipeline)
        if (stage != 0) return masks ? hipErrorInvalidValue : stage == 1 ? SRN_V(true, 1, false) : stage == 2 ? SRN_V(true, 2, false) : stage == 3 ? SRN_V(true, 3, false) : hipErrorInvalidValue;
        return masks ? SRN_V(true, 0, true) : SRN_V(true, 0, false);
    }
    switch (stage) {
        case 0: return masks ? SRN_V(false, 0, true) : SRN_V(false, 0, false);
        case 1: return masks ? SRN_V(false, 1, true) : SRN_V(false, 1, false);
        case 2: return masks ? SRN_V(false, 2, true) : SRN_V(false, 2, false);
        case 3: return masks ? SRN_V(false, 3, true) : SRN_V(false, 3, false);
        default: return hipErrorInvalidValue;
    }
#undef SRN_V
}

hipError_t launch_prep(hipStream_t st, const DeviceIndex& di, const uint64_t* items_flat, const uint32_t* q_off, uint32_t nq, uint32_t m,
                       uint32_t max_len, char* out, uint32_t stride, uint32_t* zero_a, uint32_t* zero_b, const IdSlot* loc_table, uint32_t loc_mask, unsigned long long* okeys) {
    static_assert(sizeof(PrepHead) == 72 && sizeof(PrepItem) == 24, "vmis_prep_kernel writes the record word by word");
    const uint32_t per_block = 256 / PREP_LANES;
    hipLaunchKernelGGL(vmis_prep_kernel, dim3((nq + per_block - 1) / per_block), dim3(256), 0, st, di, items_flat, q_off, nq, m, max_len, out, stride, zero_a, zero_b, loc_table, loc_mask, okeys);
    return hipGetLastError();
}

hipError_t launch_rows_to_frags(hipStream_t st, const uint64_t* row_off, const uint32_t* row_items, uint64_t n_rows, const uint32_t* block_base,
                                uint32_t* slots, uint32_t* ext) {
    hipLaunchKernelGGL(rows_to_frags_kernel, dim3((unsigned)((n_rows + 1 + 1023) / 1024)), dim3(1024), 0, st, row_off, row_items, n_rows, block_base, slots, ext);
    return hipGetLastError();
}

hipError_t launch_rows_to_slots(hipStream_t st, const uint64_t* row_off, const uint32_t* row_items, uint64_t n_rows, const uint32_t* block_base,
                                uint32_t* slots, uint32_t* ext) {
    hipLaunchKernelGGL(rows_to_slots_kernel, dim3((unsigned)((n_rows + 1 + 1023) / 1024)), dim3(1024), 0, st, row_off, row_items, n_rows, block_base, slots, ext);
    return hipGetLastError();
}

}  // namespace srn

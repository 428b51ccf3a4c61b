// =====================================================================================
// VMIS-kNN predict_next on gfx950 (MI355X): the FAST kernel -- the lean instantiation of the path for the query shape a
// production workload consists of: evolving sessions of <= 8 items of which <= 4 distinct known ones
// have a non-empty posting list above x_lo, list-set slots of 32 bits (28 rank bits + 4 list bits; 29 + 3 above 2^28 sessions), k <= 1536, m <= 2560, how_many <= 24,
// business rules on or off, no debug outputs.  Everything else -- and every query this kernel meets that does not fit -- is
// queued on a device-side list and served by vmis_predict_kernel (srn_kernels.hip) in a second launch: still on the GPU,
// same results, bit for bit.  Same algorithm as the general kernel (find_neighbors src/vmisknn/vmis_index.rs:325-415,
// predict src/vmisknn/mod.rs:118-215, canonical semantics of DESIGN.md section 1); what differs:
//
//   * the posting lists are staged with the lengths the prep kernel found (entries >= x_lo), every load of every list
//     in flight at once, and the run tables of the merge tree live in SGPRs (<= 4 runs, 2 levels);
//   * rows come from a second row array: 64-byte slots of 16-BIT LDS BYTE OFFSETS (30 items per slot), the offset of
//     an item's accumulator word -- direct-mapped for the 4096 most popular items, a sketch word for the rest -- so the
//     row walks cost one extract + one ds_add per item; unused positions hold offsets into a dump area (rows of
//     different lengths need no masking);
//   * the next query's prep record is fetched into LDS during the current query (global_load_lds), the threshold sample's constants sit in per-thread LDS
//     slots: a query's critical path has one HBM round trip in front of the merges (its lists) and none in front of the top-n;
//   * a wave's first-round row quads stay in registers between walk A and walk B; walk B only reads sketch words,
//     and an element whose word could still reach the threshold fetches its item id from the general row slots;
//   * the top-n runs in x = idf * acc space (one f64 multiply per candidate, the divide by 10 U only for the <= 160
//     final candidates), with per-chunk integer floors (popular items have small idf), survivors compacted before they
//     are scored, and a final rank-by-counting instead of a sort.
// =====================================================================================
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "srn_device.h"
#include "srn_prep.h"
#include "srn_kernels.h"

namespace srn {

// A query's finish record is written once by the kernel that served it and read once by vmis_finish_kernel; the result rows are written once: with SRN_FAST_NT these go as
// non-temporal accesses: 1 GB of records + 0.35 GB of rows per 2^20-query step that need not displace posting lists and rows in the L2 (21.98 -> 21.92 ms, twice; SRN_FAST_NT=0: plain).
#ifndef SRN_FAST_NT
#define SRN_FAST_NT 1
#endif
typedef uint32_t fin_v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void fin_store(uint4* p, const uint4& v) {
#if SRN_FAST_NT
    __builtin_nontemporal_store((fin_v4u{v.x, v.y, v.z, v.w}), reinterpret_cast<fin_v4u*>(p));
#else
    *p = v;
#endif
}
__device__ __forceinline__ uint4 fin_load(const uint4* p) {
#if SRN_FAST_NT
    const fin_v4u v = __builtin_nontemporal_load(reinterpret_cast<const fin_v4u*>(p)); return make_uint4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}
template <typename T> __device__ __forceinline__ void row_store(T* p, T v) {
#if SRN_FAST_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

// scalar words at the head of LDS
enum { FS_NB = 0, FS_FAIL, FS_SURV, FS_CCNT, FS_HITS, FS_LIVE, FS_SCAN_A = 8, FS_SCAN_B = 8, FS_W3 = 8, FS_CLS = 16, FS_TACC = 32 };   // (the three scratch areas are never live together; words 32..63: debug counters)
// behind the per-query areas: every thread's own sample constants (idf_eff of the popular item it samples in phase 4a, and its attribute byte), read from global
// memory ONCE per workgroup -- hand-made register spills into the 5 KB of LDS that three workgroups per CU leave over
static constexpr uint32_t F_SIDF = F_LDS_BYTES, F_SATTR = F_SIDF + 512 * 8, F_TOTAL = F_SATTR + 512, F_SINV = F_W10 + 384;   // (F_SINV: 1 / max idf of the 8 chunks, then of all items -- 128 bytes behind the parked record in the weight table's unused tail; 53 760 bytes in all: one more allocation granule and only two workgroups fit a CU)
static_assert(F_TOTAL * F_WG_PER_CU <= 160 * 1024, "LDS budget with the sample constants");

// per-phase cycle accounting (debug): summed per workgroup in LDS, flushed once at the end -- one global atomic per phase and
// query (as the general kernel does) serialises on 16 addresses and distorts what it measures
#ifndef SRN_FAST_RELOAD_ROWS
#define SRN_FAST_RELOAD_ROWS 0
#endif
// Wave priorities by phase (s_setprio, round 4).  The kernel is held by issue-slot contention between workgroups in DIFFERENT phases (a wave is ready but cannot
// issue 26 % of its time): the front end and the harvest are chains of dependent LDS reads -- every cycle a ready instruction of theirs waits lengthens the query -- while walk A
// is bound by the LDS pipe's atomic throughput and loses nothing by yielding the issue slot.  SRN_FAST_PRIO_LEVELS = eight decimal digits, one level (0..3) per phase: record ->
// first barrier | stage + merge tree | cuts | row requests + clears | walk A | phase 4a | live check .. resolve | wave 0's hand-off; 0 = no s_setprio at all (the kernel of rounds 2-3).
// Measured on config 3 (tools/ab_variants.sh, profiles/r04_prio_ab.txt): the levels below 22.68 ms against 23.40 without; what matters is the ORDER row requests + clears < walk A <
// front end < harvest = hand-off -- the nearer a query is to its end the higher, except that the two walk phases yield to everybody (raising either costs the whole gain).
#ifndef SRN_FAST_PRIO_LEVELS
#define SRN_FAST_PRIO_LEVELS 22201333
#endif
enum { FP_REC = 10000000, FP_FRONT = 1000000, FP_CUT = 100000, FP_REQ = 10000, FP_WALK = 1000, FP_HARV = 100, FP_WB = 10, FP_HAND = 1 };
// (two more points -- SRN_FAST_PRIO_X = two digits, 9 = leave as is: the row REQUESTS alone (the clears behind them then take FP_REQ's level), walk A's tail rounds.  Measured: the requests at
//  level 2 or 3: 23.05-23.10 ms against 22.75 -- they too must yield; the tail rounds (a second HBM / L2 round trip: latency, not throughput) at level 2: 22.65, at 3: 22.68)
#ifndef SRN_FAST_PRIO_X
#define SRN_FAST_PRIO_X 92
#endif
#define FAST_PRIO_X(div) do { if (SRN_FAST_PRIO_LEVELS && ((SRN_FAST_PRIO_X) / (div)) % 10 != 9) __builtin_amdgcn_s_setprio((short)(((SRN_FAST_PRIO_X) / (div)) % 10)); } while (0)
#ifndef SRN_MID_PRIO_LEVELS
#define SRN_MID_PRIO_LEVELS SRN_FAST_PRIO_LEVELS   // (the MID / BIG instantiations: same levels; variants measured flat, profiles/r04_prio_ab.txt)
#endif
#define FAST_PRIO(ph) do { if (SRN_FAST_PRIO_LEVELS) __builtin_amdgcn_s_setprio((short)((((MID) ? (SRN_MID_PRIO_LEVELS) : (SRN_FAST_PRIO_LEVELS)) / (ph)) % 10)); } while (0)
#ifndef SRN_FAST_STOP
#define SRN_FAST_STOP (-1)   // experiments only (tools/fast_phase_insts.sh): every query leaves after phase tick N, to count instructions per phase
#endif
#define FAST_TICK(ph) \
    do { if (ticking && tid == 0) { const long long t_ = clock64(); tacc[ph] += (unsigned long long)(t_ - t_prev); t_prev = t_; } } while (0); \
    if (SRN_FAST_STOP == (ph)) continue

// -------------------------------------------------------------------------------------
// Row layout kernel (attach time): CSR rows -> 64-byte slots of 16-bit LDS byte offsets (relative to F_HOT) + overflow
// blocks.  halfword 0 = row length, 1 = 0, 2..31 = items 0..29; rows longer than 30: halfwords 2..29 = items 0..27, word
// 15 = index of the row's first 16-byte overflow block (8 items each, items 28..).  Unused positions hold offsets
// into the dump area, spread by a hash of (row, position).  Slot n is the empty row.
// -------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t fast_offset_of(uint32_t idx, uint64_t r) {   // r: the row (recency rank) -- picks the replica of a replicated item
    if (idx < F_REP_ITEMS) return (F_DIRECT + idx * F_REP + ((uint32_t)r & (F_REP - 1u))) * 4u;
    return idx < F_DIRECT ? idx * 4u : (F_SKETCH - F_HOT) + (idx & (F_SK_WORDS - 1u)) * 4u;
}
// (round 6) WHERE the unused positions of a row slot point.  Only ~30 % of the lanes of a walk's ds_add carry a real row element; until round 5 the others added into a
// 256-word dump area inside the accumulators' map -- real LDS work, with same-address collisions among 45 lanes in 256 words.  A DS access beyond the workgroup's LDS
// allocation is dropped by the hardware (writes ignored, reads return 0) before it costs a bank cycle: tools/lds_oor_bench.hip, 24 waves per CU -- 1.57 LDS cycles per
// ds_add with the phantoms in a dump area, 1.15 with them out of range (reads: 1.91 / 1.45).  So the phantoms now point at [F_PHANTOM, + 1 KB): beyond the 53 760 bytes
// of the lean / MID / TINY / back-end forms, and inside the BIG / LONG forms' 80 KB at a place that is dead during the walks (the LONG form's numerator bytes of the
// m-cut, read for the last time in the k-cut; plain merge room in BIG) and zeroed where the dump area used to be zeroed, before walk B looks at it.
#ifndef SRN_FAST_PHANTOM_OOR
#define SRN_FAST_PHANTOM_OOR 1
#endif
static constexpr uint32_t F_PHANTOM_WORDS = 256u;
static constexpr uint32_t F_PHANTOM = SRN_FAST_PHANTOM_OOR ? F_BIG_TOTAL - (F_TOTAL - F_SIDF) - 256u * 4u - F_M_MAX : F_DUMP;   // (BIG: below the class histogram = the start of the LONG form's numerator bytes)
static_assert(!SRN_FAST_PHANTOM_OOR || (F_PHANTOM >= F_TOTAL && F_PHANTOM % 4u == 0u), "the phantom words lie beyond the 53 KB forms' LDS allocation");
static_assert(F_PHANTOM + F_PHANTOM_WORDS * 4u - F_HOT <= 65536u, "16-bit row offsets");
static_assert(!SRN_FAST_PHANTOM_OOR || F_PHANTOM_WORDS * 4u <= F_M_MAX, "inside the numerator bytes' room");
__device__ __forceinline__ uint32_t fast_phantom(uint64_t r, uint32_t j) {
    return (F_PHANTOM - F_HOT) + ((((uint32_t)r * 0x9E3779B1u + j * 0x85EBCA6Bu) >> 21) & (F_PHANTOM_WORDS - 1u)) * 4u;
}
__global__ __launch_bounds__(1024) void rows_to_packed_kernel(const uint64_t* __restrict__ row_off, const uint32_t* __restrict__ row_items, uint64_t n,
                                                              const uint32_t* __restrict__ block_base, uint32_t* __restrict__ packed, uint32_t* __restrict__ ext16) {
    __shared__ uint32_t wave_tot[16];
    const uint64_t r = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t o = 0, len = 0;
    if (r < n) { o = row_off[r]; len = row_off[r + 1] - o; }
    const uint32_t e = len > 30 ? (uint32_t)((len - 28 + 7) / 8) : 0u;   // overflow blocks of this row
    const uint32_t inc = wave_incl_scan(e);
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    uint32_t base = block_base[blockIdx.x];
    for (int w = 0; w < wave; ++w) base += wave_tot[w];
    const uint32_t eblk = base + inc - e;
    if (r > n) return;
    const uint32_t inl = len > 30 ? 28u : (uint32_t)len;
    uint32_t sl[16];
    auto half = [&](uint32_t j) -> uint32_t { return j < inl ? fast_offset_of(row_items[o + j], r) : fast_phantom(r, j); };
    sl[0] = (uint32_t)(len > 0xFFFFu ? 0xFFFFu : len);
#pragma unroll
    for (uint32_t wd = 1; wd < 16; ++wd) sl[wd] = half(2 * wd - 2) | (half(2 * wd - 1) << 16);
    if (len > 30) {
        sl[15] = eblk;
        for (uint32_t b = 0; b < e; ++b) {
            uint32_t wv[4];
#pragma unroll
            for (uint32_t x = 0; x < 4; ++x) {
                const uint64_t j0 = 28 + 8ull * b + 2 * x;
                const uint32_t lo = j0 < len ? fast_offset_of(row_items[o + j0], r) : fast_phantom(r, (uint32_t)j0);
                const uint32_t hi = j0 + 1 < len ? fast_offset_of(row_items[o + j0 + 1], r) : fast_phantom(r, (uint32_t)j0 + 1);
                wv[x] = lo | (hi << 16);
            }
            reinterpret_cast<uint4*>(ext16)[(size_t)eblk + b] = make_uint4(wv[0], wv[1], wv[2], wv[3]);
        }
    }
    uint4* dst = reinterpret_cast<uint4*>(packed + r * 16);
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) dst[qd] = make_uint4(sl[4 * qd], sl[4 * qd + 1], sl[4 * qd + 2], sl[4 * qd + 3]);
}

// The same for an item shard's row FRAGMENTS: 16-byte slots -- halfword 0 = length, 1 = 0, 2..7 = items 0..5; fragments of > 6 items: halfwords 2..5 =
// items 0..3, word 3 = index of the fragment's first overflow block (items 4..).
__global__ __launch_bounds__(1024) void rows_to_packed_frag_kernel(const uint64_t* __restrict__ row_off, const uint32_t* __restrict__ row_items, uint64_t n,
                                                                   const uint32_t* __restrict__ block_base, uint32_t* __restrict__ packed, uint32_t* __restrict__ ext16) {
    __shared__ uint32_t wave_tot[16];
    const uint64_t r = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t o = 0, len = 0;
    if (r < n) { o = row_off[r]; len = row_off[r + 1] - o; }
    const uint32_t e = len > 6 ? (uint32_t)((len - 4 + 7) / 8) : 0u;   // overflow blocks of this fragment
    const uint32_t inc = wave_incl_scan(e);
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    uint32_t base = block_base[blockIdx.x];
    for (int w = 0; w < wave; ++w) base += wave_tot[w];
    const uint32_t eblk = base + inc - e;
    if (r > n) return;
    const uint32_t inl = len > 6 ? 4u : (uint32_t)len;
    auto half = [&](uint32_t j) -> uint32_t { return j < inl ? fast_offset_of(row_items[o + j], r) : fast_phantom(r, j); };
    uint32_t sl[4];
    sl[0] = (uint32_t)(len > 0xFFFFu ? 0xFFFFu : len);
#pragma unroll
    for (uint32_t wd = 1; wd < 4; ++wd) sl[wd] = half(2 * wd - 2) | (half(2 * wd - 1) << 16);
    if (len > 6) {
        sl[3] = eblk;
        for (uint32_t b = 0; b < e; ++b) {
            uint32_t wv[4];
#pragma unroll
            for (uint32_t x = 0; x < 4; ++x) {
                const uint64_t j0 = 4 + 8ull * b + 2 * x;
                const uint32_t lo = j0 < len ? fast_offset_of(row_items[o + j0], r) : fast_phantom(r, (uint32_t)j0);
                const uint32_t hi = j0 + 1 < len ? fast_offset_of(row_items[o + j0 + 1], r) : fast_phantom(r, (uint32_t)j0 + 1);
                wv[x] = lo | (hi << 16);
            }
            reinterpret_cast<uint4*>(ext16)[(size_t)eblk + b] = make_uint4(wv[0], wv[1], wv[2], wv[3]);
        }
    }
    reinterpret_cast<uint4*>(packed)[r] = make_uint4(sl[0], sl[1], sl[2], sl[3]);
}

// -------------------------------------------------------------------------------------
// merge path over two adjacent sorted (descending) runs A = in[sa, sa + la), B = in[sa + la, sa + la + lb) -> out[sa, ...):
// team thread t (of nthr) produces outputs [t g, (t + 1) g).  All values are distinct (the position bit differs between lists).
// A thread's binary search on its diagonal costs ~11 round trips whatever g is, so a pair is merged by only as many waves as give
// every thread ~8 outputs (merge_team): the searches of the other waves would be pure overhead.
// -------------------------------------------------------------------------------------
#ifndef SRN_MERGE_G
#define SRN_MERGE_G 8   // outputs per thread a merge team is sized for
#endif
__device__ __forceinline__ uint32_t merge_team(uint32_t total) {   // log2 of the team size: 64 .. 512 threads, >= total / SRN_MERGE_G
    const uint32_t want = (total + (uint32_t)SRN_MERGE_G - 1u) / (uint32_t)SRN_MERGE_G;
    return want <= 64u ? 6u : want <= 128u ? 7u : want <= 256u ? 8u : 9u;
}
__device__ __forceinline__ void merge_pair(const uint32_t* in, uint32_t* out, uint32_t sa, uint32_t la, uint32_t lb, uint32_t ttid, uint32_t lg_nthr) {
    if ((ttid >> lg_nthr) != 0u) return;   // (whole waves)
#ifndef SRN_MERGE_ODD
#define SRN_MERGE_ODD 1
#endif
    // (round 6) a thread's share is made ODD: lane l writes out[d0 + i] with d0 = l g -- an even g puts the 64 lanes of a store on 32 / 16 / 8 of the 64 banks
    // (g = 8, a quarter of the merges: an 8-way conflict on every step); an odd stride touches all banks.  The last threads of the team get nothing: they leave at once
    const uint32_t sb = sa + la, total = la + lb, g = ((total + (1u << lg_nthr) - 1u) >> lg_nthr) | (SRN_MERGE_ODD ? 1u : 0u);
    const uint32_t d0 = min(ttid * g, total), d1 = min(d0 + g, total);
    if (d0 >= d1) return;
    uint32_t lo = d0 > lb ? d0 - lb : 0u, hi = min(d0, la);
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (in[sa + mid] > in[sb + d0 - 1 - mid]) lo = mid + 1; else hi = mid;
    }
    // sequential steps on two read pointers: one compare, one max, one dependent LDS read per output (12 VALU instructions; the index
    // form with its per-step bound checks on both runs compiled to ~20)
    const uint32_t* pa = in + sa + lo; const uint32_t* pb = in + sb + (d0 - lo);
    const uint32_t* const ea = in + sb; const uint32_t* const eb = in + sb + lb;
    uint32_t va = pa < ea ? *pa : 0u, vb = pb < eb ? *pb : 0u;   // (a staged slot is never 0)
    uint32_t* c = out + sa + d0; uint32_t* const ce = out + sa + d1;
    for (; c < ce; ++c) {
        const bool ta = va > vb;
        *c = max(va, vb);
        const uint32_t* q = (ta ? pa : pb) + 1;
        pa = ta ? q : pa; pb = ta ? pb : q;
        const uint32_t nxt = *q;   // (past a run's end: a neighbour's entry or scratch, discarded)
        const uint32_t nv = q < (ta ? ea : eb) ? nxt : 0u;
        va = ta ? nv : va; vb = ta ? vb : nv;
    }
}

// -------------------------------------------------------------------------------------
// The two cuts of find_neighbors (vmis_index.rs:332-414) in ONE pass over the merged run (round 4; before: an m-cut pass that OR-ed the copies of a session into
// a compacted list D with LDS atomics, then a k-cut pass that read D back).  F = n packed slots (rank << NB | list bit), descending: the <= 4 copies of a session
// are adjacent.  A thread takes g = ceil(n / 512) <= G consecutive entries plus three beyond them into registers, ORs every group onto its first copy (backwards carry),
// flags the firsts; block scan #1 -> index among the distinct sessions (the m most recent are kept).  No more than k of them: they ARE the neighbours, written in place.
// Otherwise the k-cut on the registers: numerator classes (<= 15) counted in packed 4-bit fields + DPP sums, the boundary class n* and its share r*, block scan #2 over
// that class, one atomic per wave for the output slots.  Returns K; the caller's barrier publishes nbl.
// -------------------------------------------------------------------------------------
#ifndef SRN_FAST_FUSED_CUT
#define SRN_FAST_FUSED_CUT 1
#endif
template <int G>
__device__ __forceinline__ uint32_t fast_cut(const uint32_t* F, uint32_t* nbl, uint32_t n, uint32_t NB, uint32_t m, uint32_t k, const uint8_t* wlut, uint32_t* misc, uint32_t tid, uint32_t lane) {
    constexpr int BLOCK = 512;
    const uint32_t NBM = (1u << NB) - 1u;
    const uint32_t g = (n + BLOCK - 1) / BLOCK, o0 = min(tid * g, n), o1 = min(o0 + g, n);
    uint32_t fv[G + 3];
#pragma unroll
    for (int x = 0; x < G + 3; ++x) fv[x] = F[min(o0 + (uint32_t)x, n - 1u)];
    const uint32_t prev0 = o0 > 0u && o0 < o1 ? F[o0 - 1u] >> NB : 0xFFFFFFFFu;   // (no rank is 0xFFFFFFFF: F[0] starts a session)
#pragma unroll
    for (int x = 0; x < G + 3; ++x) fv[x] = o0 + (uint32_t)x < n ? fv[x] : 0u;    // (past the run's end: no entry)
    {   // every group OR-ed onto its first copy
        uint32_t carry = 0u, carry_r = 0xFFFFFFFFu;
#pragma unroll
        for (int x = G + 2; x >= 0; --x) { const uint32_t r = fv[x] >> NB; carry = r == carry_r ? carry | fv[x] : fv[x]; carry_r = r; fv[x] = carry; }
    }
    uint32_t fmask = 0u, firsts = 0u;   // bit x: entry x of the chunk starts a session
    {
        uint32_t prev = prev0;
#pragma unroll
        for (int x = 0; x < G; ++x) { const bool in = o0 + (uint32_t)x < o1; const uint32_t r = fv[x] >> NB; const bool f1 = in && r != prev; fmask |= f1 ? 1u << x : 0u; firsts += f1; prev = in ? r : prev; }
    }
    uint32_t Call;
    const uint32_t idx0 = block_excl_scan<BLOCK>(firsts, misc + FS_SCAN_A, Call);   // (barrier inside)
    const uint32_t Cm = min(Call, m);
    if (Cm <= k) {   // (block-uniform) the m most recent distinct sessions are the neighbours
        uint32_t idx = idx0;
#pragma unroll
        for (int x = 0; x < G; ++x) if ((fmask >> x) & 1u) { if (idx < m) nbl[idx] = fv[x]; ++idx; }
        return Cm;
    }
    uint32_t inm = 0u; unsigned long long nmp = 0ull, acc = 0ull;   // bit x: a first among the m most recent; its class in nibble x; class counts in 4-bit fields
    {
        uint32_t idx = idx0, nmv[G];
#pragma unroll
        for (int x = 0; x < G; ++x) nmv[x] = (uint32_t)wlut[fv[x] & NBM];   // (class 0 does not exist; all reads in flight together)
#pragma unroll
        for (int x = 0; x < G; ++x) {
            const bool f1 = (fmask >> x) & 1u, in = f1 && idx < m;
            idx += f1;
            inm |= in ? 1u << x : 0u;
            nmp |= in ? (unsigned long long)nmv[x] << (4 * x) : 0ull;
            acc += in ? 1ull << (4u * nmv[x]) : 0ull;
        }
    }
    uint32_t* cls = misc + FS_CLS;
    {   // class counts: 16 fields of 4 bits per thread, spread over 4 x 64 bits with 16-bit fields, 8 DPP wave sums
        uint32_t tot[8];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const unsigned long long t = (acc >> (4 * jj)) & 0x000F000F000F000Full;
            tot[2 * jj] = wave_sum((uint32_t)t); tot[2 * jj + 1] = wave_sum((uint32_t)(t >> 32));
        }
        const uint32_t dw = (lane & 3u) * 2u + (lane >> 3);   // class = lane: word lane & 3, field lane >> 2
        uint32_t pick = tot[0];
#pragma unroll
        for (int x = 1; x < 8; ++x) pick = dw == (uint32_t)x ? tot[x] : pick;
        const uint32_t mycnt = lane < 16u ? (pick >> (16u * ((lane >> 2) & 1u))) & 0xFFFFu : 0u;
        if (mycnt) atomicAdd(&cls[lane], mycnt);
    }
    __syncthreads();
    uint32_t nstar, rstar;
    {   // lane v holds class v: suffix sums from the best class down; the boundary class is the highest one whose suffix reaches k
        const uint32_t cv = lane < 16u ? cls[lane] : 0u; const uint32_t pre = wave_incl_scan(cv);
        const uint32_t suf = (uint32_t)__builtin_amdgcn_readlane((int)pre, 63) - pre + cv;
        const unsigned long long reach = __ballot(suf >= k);
        nstar = 63u - (uint32_t)__clzll((long long)reach);
        rstar = k - ((uint32_t)__builtin_amdgcn_readlane((int)suf, (int)nstar) - (uint32_t)__builtin_amdgcn_readlane((int)cv, (int)nstar));
    }
    uint32_t mine = 0;
#pragma unroll
    for (int x = 0; x < G; ++x) mine += ((inm >> x) & 1u) && (uint32_t)((nmp >> (4 * x)) & 15ull) == nstar;
    uint32_t tot_star;
    uint32_t before = block_excl_scan<BLOCK>(mine, misc + FS_SCAN_B, tot_star);
    uint32_t sel = 0, tmask = 0;
#pragma unroll
    for (int x = 0; x < G; ++x) {
        const bool in = (inm >> x) & 1u; const uint32_t c = (uint32_t)((nmp >> (4 * x)) & 15ull);
        const bool eq = in && c == nstar, take = in && (c > nstar || (eq && before < rstar));
        tmask |= take ? 1u << x : 0u; sel += take; before += eq;
    }
    const uint32_t inc = wave_incl_scan(sel);
    uint32_t base = 0; if (lane == 63u && inc) base = atomicAdd(&misc[FS_NB], inc);
    uint32_t at = (uint32_t)__builtin_amdgcn_readlane((int)base, 63) + inc - sel;
#pragma unroll
    for (int x = 0; x < G; ++x) if ((tmask >> x) & 1u) nbl[at++] = fv[x];
    return 0xFFFFFFFFu;   // (K = misc[FS_NB] after the caller's barrier)
}

// FRAG (an item shard in lists mode): the row slots are 16-byte FRAGMENT slots -- halfword 0 = length, halfwords 2..7 = items 0..5; fragments of > 6 items:
// halfwords 2..5 = items 0..3, word 3 = the fragment's first overflow block (8 items each, items 4..) -- see rows_to_packed_frag_kernel; the general slots the
// hits are resolved from are the 16-byte fragments of DeviceIndex::row_frag.
// WIDE (an index of > 2^28 sessions): 29 rank bits + 3 list bits per slot, see NB in the kernel.
// MODE (round 4, the item-sharded index with replicated postings -- srn_group.hip, the neighbours pipeline): FM_FUSED = the whole query; FM_FRONT = find_neighbors only
// (vmis_index.rs:325-415: lists -> merge tree -> cuts), the neighbour list goes to an exchange buffer -- K, then K packed slots -- instead of into the walks, for the
// queries [f.q_base, p.nq) this rank fronts; FM_BACK = predict's scoring (mod.rs:126-214) from a neighbour list found by ANY rank: the list is read from the exchange
// buffer, the weight table rebuilt from the query's own prep record (same record on every rank: same table), then the fused kernel's walks unchanged.
// MID (round 4, the tier between this kernel and vmis_predict_kernel): evolving sessions of <= 10 items (so <= 10 lists) and numerators up to 63 -- the reference's whole
// hyper-parameter grid of last_items_in_session (1, 2, 3, 5, 10: src/hyperparameter/hyperparamgrid.rs:93-139) instead of the <= 4 lists / numerators <= 15 of the lean
// form.  Same LDS map, same walks; what differs: the queries come from a device-side list (f.mid_list, filled by the lean instantiation's hand-overs), slots carry as many
// list bits as the query has lists (ranks counted from the cut x_lo), the lists are staged four at a time, a merge tree of up to four levels with the run lengths in SGPRs, the
// two-pass cuts with a 64-bin class histogram in LDS (lane v = class v), weights from two half-set tables (lists 0..4 | 5..9), no record prefetch (parking the next record through the list's indirection measured flat: SRN_MID_PREFETCH).  A session of 10 items can see weight 0 (linear_score(10), mod.rs:110-116):
// its zero-weight neighbours add nothing, and a query whose positive-score items do not fill the top n -- the only case in which a zero-score item can be returned --
// goes to the general kernel.
enum { FM_FUSED = 0, FM_FRONT = 1, FM_BACK = 2 };
// BIG (MID only): the same kernel with 80 KB of LDS -- the merge buffers' room grows from 12 032 to 19 072 words (the per-thread sample constants move to the end), two workgroups per CU --
// for the queries MID passes on only because 2 n + 264 words do not fit the 53 KB layout (3-4 % of what it is handed on config 3); they come from a third device-side list.
// LONG (round 5, a form of MID + BIG): sessions of 11..20 items (<= 20 lists, numerators up to 255, NEGATIVE weights from the eleventh position on).  Slots are
// (rank - x_lo) << 5 | (31 - list number): the copies of a session are adjacent in the merged run, the FIRST of them (largest code = most recent position) is its first
// match (mod.rs:133-140); the m-cut stores that copy and sums the copies' numerators (L - position) into a byte per session; the k-cut's classes are those bytes (256-bin
// histogram); the neighbour list is 64-bit {slot, signed weight 10 * linear_score(first match) * numerator}.  Direct-mapped accumulators take signed adds, sketch words only the
// positive ones (they must stay upper bounds: DESIGN.md "Why the sketch filter is exact" holds with acc <= its positive part <= the word); sums <= 0 are no candidates,
// and a query whose positive scores do not fill the top n goes to the general kernel (an item of score <= 0 could then be returned).
template <int WG_PER_CU, bool FRAG, bool WIDE, int MODE = FM_FUSED, bool MID = false, bool BIG = false, bool LONG = false, bool TINY = false>
#ifndef SRN_FAST_WAVES
#define SRN_FAST_WAVES (WG_PER_CU * 2)
#endif
// (round 6) the TINY form -- one workgroup per session on the latency path, a handful of them on the whole GPU -- has no use for three workgroups per CU: at the 80-register
// cap it spilled 45-48 VGPRs, and those come back from scratch on the serial tail of a single session's latency.  SRN_TINY_WAVES waves per SIMD (2: up to 256 registers).
#ifndef SRN_TINY_WAVES
#define SRN_TINY_WAVES 2
#endif
__global__ __launch_bounds__(512, TINY ? SRN_TINY_WAVES : BIG ? 4 : SRN_FAST_WAVES) void vmis_fast_kernel(DeviceIndex ix_arg, LaunchParams p_arg, FastParams f_arg) {
#if SRN_FAST_SMALL
    // static allocation: the compiler then knows every LDS address and folds the region offsets into the instructions' offset fields
    // (with a dynamic allocation each computed address pays a v_add of the -- zero -- base: two per row item in the walks)
    static_assert(!BIG || MID, "BIG is a form of MID"); static_assert(!LONG || BIG, "LONG is a form of BIG");
    constexpr uint32_t SIDF = BIG ? F_BIG_TOTAL - (F_TOTAL - F_SIDF) : F_SIDF, SATTR = SIDF + 512 * 8;   // per-thread sample constants: behind the merge buffers' room
    constexpr uint32_t MW = (SIDF - F_WORK) / 4;   // words the merge buffers may use (= F_MERGE_WORDS in the 53 KB layout)
    __shared__ __attribute__((aligned(16))) char smem[BIG ? F_BIG_TOTAL : F_TOTAL];
#else
    extern __shared__ __attribute__((aligned(16))) char smem[];
#endif
    typedef const __attribute__((address_space(4))) char* KArg;
    const KArg ka = (KArg)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr size_t OFF_P = (sizeof(DeviceIndex) + alignof(LaunchParams) - 1) / alignof(LaunchParams) * alignof(LaunchParams);
    constexpr size_t OFF_F = (OFF_P + sizeof(LaunchParams) + alignof(FastParams) - 1) / alignof(FastParams) * alignof(FastParams);
    const __attribute__((address_space(4))) DeviceIndex& ix = *(const __attribute__((address_space(4))) DeviceIndex*)ka;
    const __attribute__((address_space(4))) LaunchParams& p = *(const __attribute__((address_space(4))) LaunchParams*)(ka + OFF_P);
    const __attribute__((address_space(4))) FastParams& f = *(const __attribute__((address_space(4))) FastParams*)(ka + OFF_F);
    constexpr int BLOCK = 512, NW = 8;
    constexpr int NL = LONG ? (int)F_LONG_LISTS : MID ? (int)F_MID_LISTS : 4;             // lists a query may have
    constexpr uint32_t SINV = F_SINV;                           // idf bounds of the integer floors
    static_assert(!MID || MODE == FM_FUSED, "MID: fused form only");
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    static_assert(!TINY || (MODE == FM_FUSED && !BIG && !FRAG), "TINY: the lean or the MID fused form over an unsharded index");
    if constexpr (TINY) {
        // The latency path's ONE launch, one workgroup per evolving session (srn_predict: one; a round of concurrent callers or a small host batch: up to SRN_TINY_FUSED_MAX = 48 by default): the workgroup
        // writes its query's prep record itself (its first eight lanes: vmis_prep_kernel's body), serves it, and wave 0 finishes it from registers (finish_inline) -- five
        // launches of 5..18 us each became one.  The launch sequence's counters start at zero (the host sees to it) and are published by the LAST workgroup to finish: what
        // this kernel hands on (general kernel, MID, > 63 entries) the host launches behind it, for that call only.
        // (the session's items ride in the kernel arguments where they fit: read from the pinned staging they are two dependent PCIe round trips -- offsets, then items)
        unsigned long long* const a_items = reinterpret_cast<unsigned long long*>(smem + F_W10 + 64); uint32_t* const a_off = reinterpret_cast<uint32_t*>(smem + F_W10 + 64 + 64);   // (where a NEXT query's record would be parked: this launch has none)
        const uint32_t alen = f.tiny_len;
        if (f.serve == nullptr) {   // (the persistent form writes the record per request, at the top of the loop below)
        if (alen) { if (tid < 8u) a_items[tid] = f.tiny_items[tid]; if (tid == 0u) { a_off[0] = 0u; a_off[1] = alen; } __syncthreads(); }
        if (tid < PREP_LANES) prep_group(ix_arg, alen ? (const uint64_t*)a_items : p.items_flat, alen ? (const uint32_t*)a_off : p.q_off, alen ? 0u : blockIdx.x, tid, p.m, p.max_len,
                                         const_cast<char*>(p.prep) + (size_t)blockIdx.x * p.prep_stride, nullptr, 0u, nullptr);
        __syncthreads();   // (workgroup-scope release / acquire: the record's words for every wave)
        }
    }

    uint32_t* misc = (uint32_t*)(smem + F_MISC);
    uint8_t* wlut = (uint8_t*)(smem + F_WLUT);            // numerator of each position set
    uint16_t* w10t = (uint16_t*)(smem + F_W10);           // 10 * linear_score(first match) * numerator of each position set
    uint32_t* nbl = (uint32_t*)(smem + F_NBL);            // neighbours; in the walks also the waves' row queues
    unsigned long long* ckey = (unsigned long long*)(smem + F_CAND);   // candidates: x = idf_eff * acc (f64 bits), then the score
    uint32_t* cidx = (uint32_t*)(smem + F_CAND + F_CAND_CAP * 8);      //             id rank (tie-break key)
    char* const acc_base = smem + F_HOT;
    uint32_t* hot = (uint32_t*)(smem + F_HOT);
    uint32_t* ikeys = (uint32_t*)(smem + F_TABLE);
    int* iacc = (int*)(smem + F_TABLE + F_TABLE_WORDS * 4);
    uint2* hits = (uint2*)(smem + F_HITS);
    uint32_t* surv = (uint32_t*)(smem + F_SURV);
    constexpr uint32_t SURV_CAP = F_SURV_WORDS - 256u;
    uint32_t* const cls_area = BIG ? (uint32_t*)(smem + SIDF) - 256 : surv + SURV_CAP;   // MID's class histogram of the k-cut: the last 256 words of the merge buffers' room
    // LONG: below it the numerator bytes of the m-cut's sessions, below those the 64-bit neighbour list (all beyond F_LDS_BYTES: the walks do not touch them)
    uint32_t* const dn_area = cls_area - F_M_MAX / 4u;
    uint2* const nb2 = reinterpret_cast<uint2*>(dn_area - F_K_MAX * 2u);
    uint32_t* thist = surv + SURV_CAP;                    // 256 bins: the sample's candidates by the top 16 bits of x, relative to the first threshold
    // numerator / 10 * linear_score(first match) * numerator of a slot's list set.  Lean: one table entry per set (16 sets).  MID (<= 10 lists: 1 024 sets): the set is split into
    // lists 0..4 and 5..9 -- wlut[0..31] / [32..63] = the halves' numerators (they add), wlut[64..95] / [96..127] = the halves' first-match positions (the low half wins)
    auto num_of = [&](uint32_t set) -> uint32_t {
        if constexpr (MID) return (uint32_t)wlut[set & 31u] + (uint32_t)wlut[32u + (set >> 5)];
        else return (uint32_t)wlut[set]; };
    auto w10_of = [&](uint32_t set) -> uint32_t {
        if constexpr (MID) { const uint32_t lo = set & 31u, hi = set >> 5; const uint32_t mp = lo ? (uint32_t)wlut[64u + lo] : (uint32_t)wlut[96u + hi]; return (9u - mp) * ((uint32_t)wlut[lo] + (uint32_t)wlut[32u + hi]); }
        else return (uint32_t)w10t[set]; };

    unsigned long long* tacc = (unsigned long long*)(smem + F_MISC + FS_TACC * 4);   // 16 debug counters, kept across the queries
    if (tid < 16u) tacc[tid] = 0ull;
    const bool ticking = p.phase_cycles != nullptr;
    const bool business = (p.flags & SRN_FLAG_BUSINESS_LOGIC) != 0u;   // (launch-uniform)
    const uint32_t n_kept = ix.n_kept;
    auto qpos = [&](uint32_t pq) -> uint32_t { return ((wave + NW * (pq >> 6)) << 6) + (pq & 63u); };   // the wave's own neighbour-list slots

    // The NEXT query's record is fetched during this one (wave 1, between walk A and the end of phase 4a) and parked in LDS: a query's critical path then starts
    // with the list loads, not with the record's HBM round trip followed by theirs.  The record's first 256 bytes go from global memory STRAIGHT into LDS
    // (global_load_lds: one dword per lane, no registers held): words 0..17 = PrepHead, words 18 + 6 l .. = item l.  (The 448 bytes behind the 16 entries
    // the weight table uses are free.)
    uint32_t* const pre = (uint32_t*)(smem + F_W10 + 64);
    bool have_pre = false;   // block-uniform
    { const ItemMeta m0 = f.meta_sample[tid];
      ((double*)(smem + SIDF))[tid] = m0.idf > 0.0 ? m0.idf : 1.0; ((uint8_t*)(smem + SATTR))[tid] = (uint8_t)m0.attr;   // (own slot only: no barrier)
      if (tid < 16u) ((double*)(smem + SINV))[tid] = tid < 8u ? f.inv_idf_hot[tid] : f.inv_idf_hi; }   // (read in phase 4a: barriers in between)
    // (round 5) the lean BACK-END form over a LIST: what the item shard's wave-per-query kernel (srn_sback.hip) could not hold -- a query with hundreds of long fragments, a hit
    // list beyond its room -- is served here, eight waves and 53 KB per query, before the general kernel gets a look; the list is the MID tier's, which a back end never has
    const bool listed = !MID && MODE == FM_BACK && f.mid_list != nullptr;   // (launch-uniform)
    const uint32_t q_end = TINY ? p.nq : LONG ? *f.long_cnt : BIG ? *f.bigq_cnt : MID || listed ? *f.mid_cnt : p.nq;   // (MID: the list is final -- the lean instantiation's launch is over)
    // The serving order (round 5, f.order; lean fused and back-end forms): the batch sorted by each query's most popular item and dealt to the XCDs chunk by chunk (ord_pos,
    // srn_device.h): one XCD's L2 sees runs of like queries, whose posting lists and neighbour rows are largely the same lines.  Without an order: query index order,
    // workgroup b serves b, b + gridDim, ...
    const bool ordered = !MID && MODE != FM_FRONT && f.order != nullptr && !listed;   // (launch-uniform)
    const uint32_t ox = blockIdx.x & 7u;
    const bool serving = TINY && f.serve != nullptr;   // (launch-uniform) the persistent form: the loop below never advances, every round is one posted session
    const uint32_t qi_step = serving ? 0u : ordered ? gridDim.x >> 3 : gridDim.x;
    uint32_t serve_seq = 0u; bool serve_have = false; unsigned long long serve_t0 = 0ull, serve_c0 = 0ull;   // (wave 0) the number of the request being served; before the first one: the number the host left in done_seq at the launch
    if constexpr (TINY) { if (serving) serve_seq = __atomic_load_n(&(f.serve + blockIdx.x)->done_seq, __ATOMIC_RELAXED); }   // (the host writes done_seq only while no kernel is resident)
    const uint32_t qi_end = ordered ? ord_count(p.nq, ox) : q_end;
    for (uint32_t qi = ordered ? blockIdx.x >> 3 : (MODE == FM_FRONT ? f.q_base : 0u) + blockIdx.x; qi < qi_end; qi += qi_step) {
        if constexpr (TINY) {
            if (serving) {
                // ---- the persistent form: answer the previous session, wait for the next one, write its record ----
                // (every path of the body ends here: the `continue`s of the hand-overs too; wave 0 finished the row from its registers before it arrived)
                uint32_t* const sv_flag = reinterpret_cast<uint32_t*>(smem + F_W10 + 64 + 192);   // (behind the posted items and their offsets; the weight table's tail is unused in this launch)
                unsigned long long* const sv_items = reinterpret_cast<unsigned long long*>(smem + F_W10 + 64); uint32_t* const sv_off = reinterpret_cast<uint32_t*>(smem + F_W10 + 64 + 128);
                ServeCtl* const c = f.serve + blockIdx.x;
                __syncthreads();
                if (wave == 0u) {
                    __threadfence_system();   // (the row's stores of every lane before the answer)
                    uint32_t leave = 0u;
                    if (serve_have && lane == 0u) {
                        // a row is final iff a finishing path wrote its count over the sentinel; what the body handed on (general kernel, MID, > 63 entries) nobody
                        // will launch: the caller takes the launch path for this session.  The hand-over counters are nobody's business here: back to 0
                        const uint32_t cnt = __atomic_load_n(&p.out_counts[blockIdx.x], __ATOMIC_RELAXED);
                        if (cnt & 0x80000000u) { (void)atomicExch(&f.slow_cnt[0], 0u); (void)atomicExch(&f.slow_cnt[1], 0u); (void)atomicExch(&f.slow_cnt[2], 0u); (void)atomicExch(&f.slow_cnt[3], 0u); }
                        const unsigned long long t_end = wall_clock64();
                        c->stamp[2] = (uint32_t)(t_end - serve_t0); c->stamp[3] = (uint32_t)((unsigned long long)clock64() - serve_c0);   // ([3]: the same span in shader cycles)
                        __atomic_store_n(&c->count, cnt, __ATOMIC_RELAXED);
                        __atomic_store_n(&c->status, (cnt & 0x80000000u) ? 1u : 0u, __ATOMIC_RELAXED);
                        __threadfence_system();
                        __atomic_store_n(&c->done_seq, serve_seq, __ATOMIC_RELAXED);
                    }
                    // the doorbell: the first 64-byte line of the control block in ONE 16-lane load -- number, length, check word, the first five items; the host wrote the
                    // number last, and the check word (number ^ length ^ the items' halves) says the line was seen whole
                    const unsigned long long t0 = wall_clock64(), idle = __builtin_nontemporal_load(&c->idle_ticks);
                    uint32_t w = 0u, s_new = 0u, n = 0u;
                    for (;;) {
                        if (lane < 16u) w = __atomic_load_n(reinterpret_cast<const uint32_t*>(c) + lane, __ATOMIC_RELAXED);
                        s_new = (uint32_t)__builtin_amdgcn_readlane((int)w, 0); n = min((uint32_t)__builtin_amdgcn_readlane((int)w, 1), 16u);
                        if (s_new != serve_seq) {
                            uint32_t x = lane >= 4u && lane < 4u + 2u * min(n, 5u) ? w : 0u;
#pragma unroll
                            for (int d = 1; d < 16; d <<= 1) x ^= __shfl_xor(x, d, 16);
                            if ((uint32_t)__builtin_amdgcn_readlane((int)x, 0) == ((uint32_t)__builtin_amdgcn_readlane((int)w, 2) ^ s_new ^ n)) break;   // (a torn line: look again)
                        }
                        if ((uint32_t)__builtin_amdgcn_readlane((int)w, 3) != 0u || wall_clock64() - t0 > idle) { leave = 1u; break; }   // (stop rides in the doorbell's line)
                    }
                    if (!leave) {
                        serve_seq = s_new; serve_have = true; serve_t0 = wall_clock64(); serve_c0 = (unsigned long long)clock64();
                        if (lane >= 4u && lane < 14u) reinterpret_cast<uint32_t*>(sv_items)[lane - 4u] = w;                                       // items 0..4 came with the doorbell
                        if (n > 5u && lane < 2u * (n - 5u)) reinterpret_cast<uint32_t*>(sv_items)[10u + lane] = __atomic_load_n(reinterpret_cast<const uint32_t*>(c->more) + lane, __ATOMIC_RELAXED);   // (a session of > 5 items: one more line)
                        if (lane == 0u) { sv_off[0] = 0u; sv_off[1] = n; p.out_counts[blockIdx.x] = 0x80000002u; c->stamp[0] = (uint32_t)(serve_t0 - t0); }   // (the sentinel: no finishing path has written this row's count)
                    }
                    if (lane == 0u) sv_flag[0] = leave;
                }
                __syncthreads();
                if (sv_flag[0] != 0u) break;   // (block-uniform)
                if (tid < PREP_LANES) prep_group(ix_arg, (const uint64_t*)sv_items, (const uint32_t*)sv_off, 0u, tid, p.m, p.max_len, const_cast<char*>(p.prep) + (size_t)blockIdx.x * p.prep_stride, nullptr, 0u, nullptr);
                __syncthreads();
                if (tid == 0u) c->stamp[1] = (uint32_t)(wall_clock64() - serve_t0);
            }
        }
        const uint32_t q = TINY ? qi : LONG ? f.long_list[qi] : BIG ? f.bigq_list[qi] : MID || listed ? f.mid_list[qi] : ordered ? (uint32_t)f.order[ord_pos(ox, qi)] : qi;   // (TINY: the launch's one query, MID form included)
        const uint32_t q_ord_next = ordered && qi + qi_step < qi_end ? (uint32_t)f.order[ord_pos(ox, qi + qi_step)]
                                  : listed && qi + qi_step < qi_end ? f.mid_list[qi + qi_step] : 0xFFFFFFFFu;   // (the query this workgroup serves next: its record is parked during this one)
#ifndef SRN_MID_PREFETCH
#define SRN_MID_PREFETCH 0   // (experiment, measured flat: 7.73 / 8.61 ms per 2^18 queries at max_items 8 / 10 with or without it, and 16 bytes of scratch with)
#endif
        const uint32_t q_after = !(MID && SRN_MID_PREFETCH) ? 0xFFFFFFFFu : qi + gridDim.x < q_end ? (BIG ? f.bigq_list : f.mid_list)[qi + gridDim.x] : 0xFFFFFFFFu;   // (MID: the query this workgroup serves next -- its record is parked during this one, like the lean form's)
        long long t_prev = ticking ? clock64() : 0;
        FAST_PRIO(FP_REC);
        // ---- phase 0: the query's prep record -> run descriptors in SGPRs ------------------------------
        const char* const rec = p.prep + (size_t)q * p.prep_stride;
        struct { uint32_t U, rmax, xlo, sumw, L, n_staged, cur_attr, unsafe; } hd;
        constexpr uint32_t HW = (uint32_t)sizeof(PrepHead) / 4u;   // record words before the items
        struct { uint32_t idx, kept; unsigned long long base; } x0{kNone, 0u, 0ull};
        if (have_pre) {
            auto uni = [&](uint32_t w) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane((int)pre[w]); };   // (SGPRs: the branches on these stay scalar)
            hd.U = uni(0); hd.rmax = uni(1); hd.xlo = uni(2); hd.sumw = uni(3); hd.L = uni(6); hd.n_staged = uni(7); hd.cur_attr = uni(16); hd.unsafe = uni(17);
            if (lane < (LONG ? 32u : MID ? 16u : 8u) && lane < hd.L) { const uint32_t* it = pre + HW + 6u * lane; x0.idx = it[0]; x0.kept = it[3]; x0.base = ((unsigned long long)it[5] << 32) | it[4]; }
        } else {
            const PrepHead h0 = *(const PrepHead*)rec;   // (uniform address)
            hd.U = h0.U; hd.rmax = h0.rmax; hd.xlo = h0.xlo; hd.sumw = h0.sumw; hd.L = h0.L; hd.n_staged = h0.n_staged; hd.cur_attr = h0.cur_attr; hd.unsafe = h0.unsafe;
            if ((MID ? lane < (LONG ? 32u : 16u) && lane < p.max_len : lane < 8u) && lane < hd.L) { const PrepItem pi = ((const PrepItem*)(rec + sizeof(PrepHead)))[lane]; x0.idx = pi.idx; x0.kept = pi.kept; x0.base = pi.base; }
        }
        have_pre = false;
        const uint32_t L = hd.L, n = hd.n_staged, U = hd.U;
        unsigned long long rm = __ballot(x0.kept > 0u);
#ifdef SRN_FAST_EXP_ONELIST   // experiment (timing only, wrong results): every query as if it had its first list alone -- no merge, no cuts: what the walks + harvest cost by themselves
        rm &= 0ull - rm;
#endif
        const uint32_t nr = (uint32_t)__popcll(rm);
        // Slot = (rank - base) << NB | set of lists.  Normally NB = 4 (WIDE, above 2^28 sessions: 3) and base = 0.  A query with MORE lists than bits
        // (4 lists on an index of > 2^28 sessions) takes NB = 4 with the ranks counted from the cut x_lo -- every staged entry is >= x_lo -- if that fits 28 bits.
        const bool rel = MID || (WIDE && nr > 3u);   // (block-uniform)
        const uint32_t NB = LONG ? F_LONG_NB : MID ? max(nr, 4u) : WIDE && !rel ? 3u : 4u, NBM = (1u << NB) - 1u, base = rel ? hd.xlo : 0u;
        // (MID: the last 256 words of the merge buffers' room hold the class histogram of the k-cut)
        // (round 6: hd.unsafe -- an incomplete posting list of a pre-built index can reach this query's neighbours: the general kernel's row pass serves it, srn_prep.h)
        const bool fits = hd.unsafe != 0u ? false : LONG ? L >= 1u && L <= F_LONG_LMAX && L <= p.max_len && hd.sumw <= F_LONG_CLASSES && nr <= F_LONG_LISTS && 2u * n + 8u + F_LONG_RES_WORDS <= MW && hd.rmax - hd.xlo < (1u << (32u - F_LONG_NB))
                        : MID ? L >= 1u && L <= F_MID_LMAX && L <= p.max_len && hd.sumw <= F_MID_CLASSES && nr <= F_MID_LISTS && 2u * n + 8u + 256u <= MW && hd.rmax - hd.xlo < (1u << (32u - NB))
                              : L >= 1u && L <= 8u && L <= p.max_len && hd.sumw <= 15u && nr <= 4u && 2u * n + 8u <= MW && (!rel || hd.rmax - hd.xlo < (1u << 28));
        uint32_t* const xq = MODE == FM_FUSED ? nullptr : f.xchg + (size_t)q * f.xchg_stride;   // this query's place in the exchange buffer: K | K slots
        if (!fits) {   // block-uniform: the general kernel takes it
            if constexpr (MODE == FM_FRONT) { if (tid == 0) xq[0] = 0xFFFFFFFFu; }   // (every rank's back end reads the marker and hands the query to its general kernel)
            else if (tid == 0) {
                // (round 5) a query of the lean shape whose merged lists alone outgrow the 53 KB layout -- the headline batches' 0.3 % -- goes straight to the BIG form (80 KB of
                // LDS: n <= 9 404), not to the general kernel: 3 209 such queries per 2^20 cost 0.27 ms there
                if (hd.unsafe != 0u) f.slow_list[atomicAdd(f.slow_cnt, 1u)] = q;
                else if (!MID && MODE == FM_FUSED && f.bigq_list != nullptr && L >= 1u && L <= 8u && L <= p.max_len && hd.sumw <= 15u && nr <= 4u && hd.rmax - hd.xlo < (1u << 28) &&
                    2u * n + 8u + 256u <= (F_BIG_TOTAL - (F_TOTAL - F_SIDF) - F_WORK) / 4u) f.bigq_list[atomicAdd(f.bigq_cnt, 1u)] = q;
                // (the MID instantiation looks at the query next, if this launch sequence has one; it decides for itself)
                else if (!MID && MODE == FM_FUSED && f.mid_list != nullptr && L >= 1u && L <= F_MID_LMAX && L <= p.max_len) f.mid_list[atomicAdd(f.mid_cnt, 1u)] = q;
                // (... and the LONG one at sessions of 11..20 items)
                else if (!MID && MODE == FM_FUSED && f.long_list != nullptr && L > F_MID_LMAX && L <= F_LONG_LMAX && L <= p.max_len) f.long_list[atomicAdd(f.long_cnt, 1u)] = q;
                else if (MID && !BIG && f.bigq_list != nullptr && L >= 1u && L <= F_MID_LMAX && L <= p.max_len && hd.sumw <= F_MID_CLASSES && nr <= F_MID_LISTS && hd.rmax - hd.xlo < (1u << (32u - NB)) &&
                         2u * n + 8u + 256u <= (F_BIG_TOTAL - (F_TOTAL - F_SIDF) - F_WORK) / 4u) f.bigq_list[atomicAdd(f.bigq_cnt, 1u)] = q;   // (only the merge buffers' room is missing: MID's BIG form has it)
                else f.slow_list[atomicAdd(f.slow_cnt, 1u)] = q;
            }
            continue;
        }
        if (n == 0u) { if (tid == 0) { if constexpr (MODE == FM_FRONT) xq[0] = 0u; else p.out_counts[q] = 0u; } continue; }   // no known item (vmis_index.rs:350): empty result
        uint32_t kp[NL], ps[NL]; const uint32_t* src[NL];
#pragma unroll
        for (int r = 0; r < NL; ++r) {
            const int l = rm ? __ffsll((long long)rm) - 1 : 0;
            kp[r] = rm ? (uint32_t)__builtin_amdgcn_readlane((int)x0.kept, l) : 0u;
            const uint32_t blo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x0.base, l), bhi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x0.base >> 32), l);
            src[r] = ix.post_rank + (((unsigned long long)bhi << 32) | blo);
            ps[r] = (uint32_t)l;
            rm &= rm - 1ull;
        }
        const uint32_t cur_idx = (uint32_t)__builtin_amdgcn_readlane((int)x0.idx, 0);
        const uint32_t cur_attr = hd.cur_attr;   // business rules (mod.rs:162-182): the current item's attribute byte, looked up by the prep kernel
        const uint32_t s1 = kp[0], s2 = s1 + kp[1], s3 = s2 + kp[2];

        // The stage loads go out BEFORE the barrier that ends the previous query: waves 1..7 get here while wave 0 still ranks that
        // query's candidates, and their share of the posting lists is in flight meanwhile.
        uint32_t v[4][5];   // every load of every list in flight at once (uniform skips; past a list's end the lanes re-read its last entry)
        uint32_t K = 0;
        uint32_t pfv[4] = {0u, 0u, 0u, 0u};   // (experiment SRN_FAST_ROW_PREFETCH=2 only)
        uint32_t xsv[3] = {0u, 0u, 0u};
        if constexpr (MODE == FM_BACK) {   // K and this lane's <= 3 neighbour slots in ONE round trip, requested before the barrier like the lists of the fused form (slots past K: stale words of
                                           // the query's own row of the exchange buffer, masked below)
            const uint32_t kv = xq[0];
#pragma unroll
            for (int t = 0; t < 3; ++t) xsv[t] = xq[1u + min(wave * 64u + lane + (uint32_t)t * BLOCK, f.xchg_stride - 2u)];
            K = (uint32_t)__builtin_amdgcn_readfirstlane((int)kv);
        } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 5; ++j) { v[r][j] = 0u; if ((uint32_t)j * BLOCK < kp[r]) v[r][j] = src[r][min(tid + j * BLOCK, kp[r] - 1u)]; }
        }
        __syncthreads();   // previous query's LDS reads are done
        FAST_PRIO(FP_FRONT);
        // (opaque copies: with a compile-time shift the packing of a staged entry is otherwise hoisted into the conditional block of its load, which then
        //  ends in s_waitcnt vmcnt(0) -- the lists' loads would go out one HBM round trip after the other instead of all together)
        if constexpr (MODE != FM_BACK) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 5; ++j) asm volatile("" : "+v"(v[r][j]));
        }
        FAST_TICK(0);
        if (tid < (uint32_t)FS_TACC) misc[tid] = 0;
        // Experiment (round 4, measured, NOT kept: 22.92 against 22.67 ms): the accumulator words ABOVE this query's merge buffers cleared here, while its posting lists are still on
        // their way -- the previous query is done with all of LDS, the merges use 2 n words from F_WORK up, a one-list query and the back-end form have no merge buffers at all --
        // so that the clears phase only clears what the merges used.  The clears then run at the front end's priority, in front of its chain, instead of yielding to everybody.
#ifndef SRN_FAST_EARLY_CLEAR
#define SRN_FAST_EARLY_CLEAR 0
#endif
        const uint32_t clr_lo = (!SRN_FAST_EARLY_CLEAR || MODE == FM_FRONT) ? F_TABLE : (MODE == FM_BACK || nr <= 1u) ? F_HOT : min(F_TABLE, max(F_HOT, (F_WORK + (2u * n + 16u) * 4u + 15u) & ~15u));   // (block-uniform)
        if (SRN_FAST_EARLY_CLEAR && MODE != FM_FRONT) {
            uint4* z = reinterpret_cast<uint4*>(smem);
            for (uint32_t i = clr_lo / 16u + tid; i < F_TABLE / 16u; i += BLOCK) z[i] = make_uint4(0u, 0u, 0u, 0u);
        }
        // A slot's low bits are the set of RUNS (not evolving positions) that hold the session: <= 4 runs, so 4 bits (3 above 2^28 sessions, see NB above) whatever the
        // session length, and 28 (29) bits for the rank.  Runs are numbered in position order, so the lowest set run is the first match (Q4).
        if constexpr (LONG) {   // by list code (31 - list number): wlut[code] = the list's numerator share L - position, wlut[32 + code] = its position
            if (tid < 32u) {
                uint32_t pr = 0u;
#pragma unroll
                for (int r = 0; r < NL; ++r) pr = tid == (uint32_t)(31 - r) ? ps[r] : pr;
                wlut[tid] = (uint8_t)(L - pr); wlut[32u + tid] = (uint8_t)pr;   // (codes of lists the query does not have are never looked up)
            }
        } else
        if constexpr (MID) {
            if (tid < 64u) {   // (two halves of 5 lists; 10 * linear_score(first match) = 9 - mp: 0 at the tenth position)
                const uint32_t half = tid >> 5, hs = tid & 31u;
                uint32_t num = 0u, mp = 0u;
                const uint32_t lo = hs ? (uint32_t)__ffs((int)hs) - 1u : 0u;
#pragma unroll
                for (int r = 0; r < 5; ++r) { const uint32_t pr = half ? ps[5 + r] : ps[r]; num += ((hs >> r) & 1u) ? L - pr : 0u; mp = lo == (uint32_t)r ? pr : mp; }
                wlut[tid] = (uint8_t)num; wlut[64u + tid] = (uint8_t)mp;
            }
        } else
        if (tid < (1u << nr)) {
            const uint32_t num = ((tid & 1u) ? L - ps[0] : 0u) + ((tid & 2u) ? L - ps[1] : 0u) + ((tid & 4u) ? L - ps[2] : 0u) + ((tid & 8u) ? L - ps[3] : 0u);
            const uint32_t lo = tid ? (uint32_t)__ffs((int)tid) - 1u : 0u;
            const uint32_t mp = lo == 0u ? ps[0] : lo == 1u ? ps[1] : lo == 2u ? ps[2] : ps[3];
            wlut[tid] = (uint8_t)num; w10t[tid] = (uint16_t)((9u - mp) * num);
        }
        if constexpr (MODE == FM_BACK) {
            if (K == 0xFFFFFFFFu) { if (tid == 0) f.slow_list[atomicAdd(f.slow_cnt, 1u)] = q; continue; }   // (the fronting rank could not take it: block-uniform)
            if (K == 0u) { if (tid == 0) p.out_counts[q] = 0u; continue; }
        } else
        if (nr == 1u) {
            // ONE list (a quarter of the queries): its entries are distinct sessions in recency order, all of one numerator class -- the candidates are
            // its first min(n, m) entries, the neighbours the first k of those.  No merge, no m-cut, no k-cut: the list goes straight into the neighbour list.
            K = min(kp[0], p.k);
#pragma unroll
            for (int j = 0; j < 3; ++j) { const uint32_t e = tid + j * BLOCK; if ((uint32_t)j * BLOCK < K && e < K) {
                if constexpr (LONG) nb2[e] = make_uint2(((v[0][j] - base) << NB) | 31u, (uint32_t)((9 - (int)ps[0]) * (int)(L - ps[0])));   // (one class, one weight)
                else nbl[e] = ((v[0][j] - base) << NB) | 1u; } }
            __syncthreads();
            FAST_TICK(1);
        } else {
        // ---- phase 1: stage the lists' kept prefixes as packed slots (rank << NB | position bit) ---------
        // (B1 is the LOWER half: the m-cut's output D lands there, on top of the neighbour list's own words -- where no k-cut follows, D is the neighbour list as it stands)
        uint32_t* const B1 = (uint32_t*)(smem + F_WORK); uint32_t* const B0 = B1 + n;
        static_assert(F_WORK == F_NBL, "the m-cut writes the neighbour list in place");
        if constexpr (MID) {
            // lists 0..3 are in flight since before the barrier; the others take one more round trip (only queries of > 4 lists pay it).  Everything is staged into ONE
            // buffer at the lists' prefix offsets, the one from which nlev merge levels end in B0.
            const uint32_t nlev = nr <= 2u ? 1u : nr <= 4u ? 2u : nr <= 8u ? 3u : nr <= 16u ? 4u : 5u;
            uint32_t* const X = (nlev & 1u) ? B1 : B0; uint32_t* const Y = (nlev & 1u) ? B0 : B1;
            uint32_t pre_r[NL];
            { uint32_t a = 0u;
#pragma unroll
              for (int r = 0; r < NL; ++r) { pre_r[r] = a; a += kp[r]; } }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 5; ++j) { const uint32_t e = tid + j * BLOCK; if ((uint32_t)j * BLOCK < kp[r] && e < kp[r]) X[pre_r[r] + e] = ((v[r][j] - base) << NB) | (LONG ? 31u - (uint32_t)r : 1u << r); }
#pragma unroll
            for (int b4 = 4; b4 < NL; b4 += 4) {
                if (nr > (uint32_t)b4) {   // (block-uniform)
                    constexpr int R4 = 4;
#pragma unroll
                    for (int r = 0; r < R4; ++r)
#pragma unroll
                        for (int j = 0; j < 5; ++j) { v[r][j] = 0u; if (b4 + r < NL) { if ((uint32_t)j * BLOCK < kp[b4 + r < NL ? b4 + r : 0]) v[r][j] = src[b4 + r < NL ? b4 + r : 0][min(tid + j * BLOCK, kp[b4 + r < NL ? b4 + r : 0] - 1u)]; } }
#pragma unroll
                    for (int r = 0; r < R4; ++r)
#pragma unroll
                        for (int j = 0; j < 5; ++j) asm volatile("" : "+v"(v[r][j]));   // (all loads out before the first is packed, as above)
#pragma unroll
                    for (int r = 0; r < R4; ++r)
#pragma unroll
                        for (int j = 0; j < 5; ++j) { if (b4 + r < NL) { const int rr = b4 + r < NL ? b4 + r : 0; const uint32_t e = tid + j * BLOCK; if ((uint32_t)j * BLOCK < kp[rr] && e < kp[rr]) X[pre_r[rr] + e] = ((v[r][j] - base) << NB) | (LONG ? 31u - (uint32_t)rr : 1u << rr); } }
                }
            }
            __syncthreads();
            FAST_TICK(1);
            // merge tree: <= 4 levels of adjacent pairs; a level's pairs go to consecutive thread teams (sized as in the lean form; teams wrap around the workgroup: a
            // wave that belongs to two teams merges one pair after the other), a run without a partner is copied (merge with an empty run)
            uint32_t len[NL];
#pragma unroll
            for (int r = 0; r < NL; ++r) len[r] = kp[r];
            uint32_t runs = nr; const uint32_t* in = X; uint32_t* out = Y;
#pragma unroll
            for (int lev = 0; lev < (LONG ? 5 : 4); ++lev) {
                if ((uint32_t)lev < nlev) {   // (block-uniform)
                    uint32_t off = 0u, start = 0u;
#pragma unroll
                    for (int j = 0; j < (((NL + (1 << lev) - 1) >> lev) + 1) / 2; ++j) {   // pairs of a level that starts with ceil(NL / 2^lev) runs
                        if (2u * (uint32_t)j < runs) {
                            const uint32_t la = len[2 * j], lb = 2u * (uint32_t)j + 1u < runs ? len[2 * j + 1] : 0u;
                            const uint32_t lg = merge_team(la + lb);
                            merge_pair(in, out, off, la, lb, (tid - start) & 511u, lg);
                            start = (start + (1u << lg)) & 511u; off += la + lb; len[j] = la + lb;
                        }
                    }
                    runs = (runs + 1u) >> 1;
                    uint32_t* const t = const_cast<uint32_t*>(in); in = out; out = t;
                    __syncthreads();
                }
            }
        } else {
        const uint32_t nl = (nr > 1u) + (nr > 2u);
        {
            uint32_t* const d01 = nl == 1u ? B1 : B0; uint32_t* const d2 = nr == 3u ? B1 : B0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                uint32_t* const dst = (r < 2 ? d01 : r == 2 ? d2 : B0) + (r == 0 ? 0u : r == 1 ? s1 : r == 2 ? s2 : s3);
#pragma unroll
                for (int j = 0; j < 5; ++j) { const uint32_t e = tid + j * BLOCK; if ((uint32_t)j * BLOCK < kp[r] && e < kp[r]) dst[e] = ((v[r][j] - base) << NB) | (1u << r); }
            }
        }
        __syncthreads();
        FAST_TICK(1);
        // ---- merge tree: <= 2 levels, the final run lands in B0 ------------------------------------------
        // (level 0 of four runs: the first pair's team counts threads from 0 up, the second pair's from 511 down -- on disjoint waves when both are small)
        if (nr == 2u) { merge_pair(B1, B0, 0u, kp[0], kp[1], tid, merge_team(s2)); __syncthreads(); }
        else if (nr == 3u) { merge_pair(B0, B1, 0u, kp[0], kp[1], tid, merge_team(s2)); __syncthreads(); merge_pair(B1, B0, 0u, s2, kp[2], tid, merge_team(n)); __syncthreads(); }
        else if (nr == 4u) { merge_pair(B0, B1, 0u, kp[0], kp[1], tid, merge_team(s2)); merge_pair(B0, B1, s2, kp[2], kp[3], 511u - tid, merge_team(n - s2)); __syncthreads();
                             merge_pair(B1, B0, 0u, s2, kp[2] + kp[3], tid, merge_team(n)); __syncthreads(); }
        }
        FAST_TICK(3);
        FAST_PRIO(FP_CUT);
        const uint32_t* F = B0; uint32_t* D = B1;
        // Experiment (round 6, measured, NOT kept: 26.5 against 21.5 ms): the CANDIDATES' row slots prefetched here, 5-6 K cycles before the cuts have named the neighbours
        // -- one dword per candidate into a dead LDS line (global_load_lds) -- so that the real requests find their lines in the L2.  The memory system moves a tenth of
        // the algorithmic bytes and the L2s run at <= 41 %, yet 3 072 more scattered requests per query (+130 %) cost a quarter of the kernel: what a CU can ISSUE in scattered
        // 64-lane gathers is the scarce thing, not bytes.  (profiles/r06_row_prefetch_ab.txt)
#ifndef SRN_FAST_ROW_PREFETCH
#define SRN_FAST_ROW_PREFETCH 0
#endif
        if constexpr (SRN_FAST_ROW_PREFETCH != 0 && !MID && MODE == FM_FUSED && !FRAG) {
#pragma unroll
            for (int j = 0; j < (SRN_FAST_ROW_PREFETCH == 1 ? 6 : 4); ++j) {
                if ((uint32_t)j * BLOCK < n) {   // (block-uniform)
                    const uint32_t sl = F[min(tid + (uint32_t)j * BLOCK, n - 1u)];
                    const char* a = reinterpret_cast<const char*>(f.row_packed) + (size_t)(base + (sl >> NB)) * 64u;
                    if (SRN_FAST_ROW_PREFETCH == 1) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)a, (__attribute__((address_space(3))) void*)pre, 4, 0, 0);
                    else if (j < 4) pfv[j] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(a));   // (2: plain loads into registers that stay live across the cuts)
                }
            }
        }
        // ---- the two cuts in one pass over the merged run (fast_cut above) where a thread's chunk is <= 6 entries (n <= 3072: four queries in five); the two-pass form below otherwise ----
        if (!MID && SRN_FAST_FUSED_CUT && n <= 6u * BLOCK) {   // (block-uniform)
            const uint32_t kc = fast_cut<6>(F, nbl, n, NB, p.m, p.k, wlut, misc, tid, lane);
            __syncthreads();
            K = kc != 0xFFFFFFFFu ? kc : misc[FS_NB];
            FAST_TICK(2);
        } else {
        // ---- m-cut: the copies of a session are adjacent in F; the first m distinct sessions, position sets OR-ed ----
        uint32_t Call, Cm;
        {
            constexpr int MC_G = 6;   // chunks of <= MC_G entries (n <= 3072) are read with all LDS reads in flight together, twice; longer ones one entry per round trip
            const uint32_t g = (n + BLOCK - 1) / BLOCK, o0 = min(tid * g, n), o1 = min(o0 + g, n);
            const uint32_t prev0 = o0 > 0u && o0 < o1 ? F[o0 - 1] >> NB : 0xFFFFFFFFu;   // (no rank is 0xFFFFFFFF: F[0] starts a session)
            uint32_t firsts = 0, prev = prev0;
            if (g <= (uint32_t)MC_G) {   // (block-uniform)
                uint32_t fv[MC_G];
#pragma unroll
                for (int x = 0; x < MC_G; ++x) fv[x] = F[min(o0 + (uint32_t)x, n - 1u)];
#pragma unroll
                for (int x = 0; x < MC_G; ++x) { const bool in = o0 + (uint32_t)x < o1; const uint32_t r = fv[x] >> NB; firsts += in && r != prev; prev = in ? r : prev; }
            } else
                for (uint32_t o = o0; o < o1; ++o) { const uint32_t r = F[o] >> NB; firsts += r != prev; prev = r; }
            for (uint32_t i = tid; i < min(n, p.m); i += BLOCK) D[i] = 0;
            if (MID && tid < (LONG ? 256u : 64u)) cls_area[tid] = 0u;   // (the k-cut's class histogram: the tail of the merge buffers' room, see `fits`)
            if constexpr (LONG) for (uint32_t i = tid; i < (min(n, p.m) + 3u) / 4u; i += BLOCK) dn_area[i] = 0u;   // (a numerator byte per kept session)
            uint32_t idx = block_excl_scan<BLOCK>(firsts, misc + FS_SCAN_A, Call);   // (barrier inside)
            prev = prev0;
            if (g <= (uint32_t)MC_G) {
                uint32_t fv[MC_G];
#pragma unroll
                for (int x = 0; x < MC_G; ++x) fv[x] = F[min(o0 + (uint32_t)x, n - 1u)];
#pragma unroll
                for (int x = 0; x < MC_G; ++x) {
                    const bool in = o0 + (uint32_t)x < o1; const uint32_t r = fv[x] >> NB; const bool f1 = in && r != prev;
                    idx += f1; prev = in ? r : prev;
                    if constexpr (LONG) {   // the group's first copy (largest code: the first match) is the session's slot; every copy adds its list's share of the numerator
                        if (in && idx - 1u < p.m) { if (f1) D[idx - 1u] = fv[x]; atomicAdd(&dn_area[(idx - 1u) >> 2], (uint32_t)wlut[fv[x] & 31u] << (8u * ((idx - 1u) & 3u))); }
                    } else
                    if (in && idx - 1u < p.m) atomicOr(&D[idx - 1u], fv[x]);
                }
            } else
                for (uint32_t o = o0; o < o1; ++o) {
                    const uint32_t v = F[o], r = v >> NB; const bool f1 = r != prev;
                    idx += f1; prev = r;
                    if constexpr (LONG) { if (idx - 1u < p.m) { if (f1) D[idx - 1u] = v; atomicAdd(&dn_area[(idx - 1u) >> 2], (uint32_t)wlut[v & 31u] << (8u * ((idx - 1u) & 3u))); } }
                    else
                    if (idx - 1u < p.m) atomicOr(&D[idx - 1u], v);
                }
            Cm = min(Call, p.m);
        }
        __syncthreads();
        FAST_TICK(2);
        // ---- k-cut: D is ordered by recency, so inside one numerator class the order is already the wanted one ----
        auto dn_of = [&](uint32_t i) -> uint32_t { return (dn_area[i >> 2] >> (8u * (i & 3u))) & 255u; };                               // LONG: numerator of the m-cut's i-th session
        auto w_long = [&](uint32_t slot, uint32_t num) -> uint32_t { return (uint32_t)((9 - (int)(uint32_t)wlut[32u + (slot & 31u)]) * (int)num); };   // 10 * linear_score(first match) * numerator, signed
        if (Cm <= p.k) {   // (block-uniform; the m-cut wrote the neighbour list)
            K = Cm;
            if constexpr (LONG) { for (uint32_t i = tid; i < Cm; i += BLOCK) { const uint32_t dvv = D[i]; nb2[i] = make_uint2(dvv, w_long(dvv, dn_of(i))); } __syncthreads(); }
        } else {
            uint32_t* cls = MID ? cls_area : misc + FS_CLS;
            const uint32_t g = (Cm + BLOCK - 1) / BLOCK, o0 = min(tid * g, Cm), o1 = min(o0 + g, Cm);   // g <= 5
            uint32_t dv[5], nmv[5];
#pragma unroll
            for (int x = 0; x < 5; ++x) dv[x] = o0 + x < o1 ? D[o0 + x] : 0u;
#pragma unroll
            for (int x = 0; x < 5; ++x) { if constexpr (LONG) nmv[x] = o0 + x < o1 ? dn_of(o0 + (uint32_t)x) : 0u; else nmv[x] = o0 + x < o1 ? num_of(dv[x] & NBM) : 0u; }   // (class 0 does not exist)
            if constexpr (MID) {   // <= 63 classes: a histogram in LDS (lane v of the scan below = class v)
#pragma unroll
                for (int x = 0; x < 5; ++x) if (o0 + x < o1) atomicAdd(&cls[nmv[x]], 1u);
            } else
            {   // class counts: 16 fields of 4 bits per thread, spread over 4 x 64 bits with 16-bit fields, 8 DPP wave sums
                unsigned long long acc = 0;
#pragma unroll
                for (int x = 0; x < 5; ++x) acc += o0 + x < o1 ? 1ull << (4u * nmv[x]) : 0ull;
                uint32_t tot[8];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const unsigned long long t = (acc >> (4 * jj)) & 0x000F000F000F000Full;
                    tot[2 * jj] = wave_sum((uint32_t)t); tot[2 * jj + 1] = wave_sum((uint32_t)(t >> 32));
                }
                const uint32_t dw = (lane & 3u) * 2u + (lane >> 3);   // class = lane: word lane & 3, field lane >> 2
                uint32_t pick = tot[0];
#pragma unroll
                for (int x = 1; x < 8; ++x) pick = dw == (uint32_t)x ? tot[x] : pick;
                const uint32_t mycnt = lane < 16u ? (pick >> (16u * ((lane >> 2) & 1u))) & 0xFFFFu : 0u;
                if (mycnt) atomicAdd(&cls[lane], mycnt);
            }
            __syncthreads();
            uint32_t nstar, rstar;
            if constexpr (LONG) {   // 256 classes: lane l holds classes 4 l .. 4 l + 3; suffix sums from the best class down, the boundary class is the highest one whose suffix reaches k
                const uint4 c4 = reinterpret_cast<const uint4*>(cls)[lane];
                const uint32_t s4 = c4.x + c4.y + c4.z + c4.w, pre4 = wave_incl_scan(s4);
                const uint32_t above = (uint32_t)__builtin_amdgcn_readlane((int)pre4, 63) - pre4;   // entries in the classes of higher lanes
                const uint32_t suf3 = above + c4.w, suf2 = suf3 + c4.z, suf1 = suf2 + c4.y, suf0 = suf1 + c4.x;
                const unsigned long long reach = __ballot(suf0 >= p.k);
                const int top = 63 - __clzll((long long)reach);
                const uint32_t jsel = suf3 >= p.k ? 3u : suf2 >= p.k ? 2u : suf1 >= p.k ? 1u : 0u;
                const uint32_t sufs = jsel == 3u ? suf3 : jsel == 2u ? suf2 : jsel == 1u ? suf1 : suf0, cnts = jsel == 3u ? c4.w : jsel == 2u ? c4.z : jsel == 1u ? c4.y : c4.x;
                nstar = (uint32_t)__builtin_amdgcn_readlane((int)(4u * lane + jsel), top);
                rstar = p.k - ((uint32_t)__builtin_amdgcn_readlane((int)sufs, top) - (uint32_t)__builtin_amdgcn_readlane((int)cnts, top));
            } else
            {   // lane v holds class v: suffix sums from the best class down; the boundary class is the highest one whose suffix reaches k
                const uint32_t cv = (MID || lane < 16u) ? cls[lane] : 0u; const uint32_t pre = wave_incl_scan(cv);
                const uint32_t suf = (uint32_t)__builtin_amdgcn_readlane((int)pre, 63) - pre + cv;
                const unsigned long long reach = __ballot(suf >= p.k);
                nstar = 63u - (uint32_t)__clzll((long long)reach);
                rstar = p.k - ((uint32_t)__builtin_amdgcn_readlane((int)suf, (int)nstar) - (uint32_t)__builtin_amdgcn_readlane((int)cv, (int)nstar));
            }
            uint32_t mine = 0;
#pragma unroll
            for (int x = 0; x < 5; ++x) mine += o0 + x < o1 && nmv[x] == nstar;
            uint32_t tot_star;
            uint32_t before = block_excl_scan<BLOCK>(mine, misc + FS_SCAN_B, tot_star);
            uint32_t sel = 0; bool take[5];
#pragma unroll
            for (int x = 0; x < 5; ++x) {
                const bool in_r = o0 + x < o1, eq = nmv[x] == nstar;
                take[x] = in_r && (nmv[x] > nstar || (eq && before < rstar));
                sel += take[x]; before += in_r && eq;
            }
            const uint32_t inc = wave_incl_scan(sel);
            uint32_t base = 0; if (lane == 63u && inc) base = atomicAdd(&misc[FS_NB], inc);
            uint32_t at = (uint32_t)__builtin_amdgcn_readlane((int)base, 63) + inc - sel;
#pragma unroll
            for (int x = 0; x < 5; ++x) if (take[x]) { if constexpr (LONG) nb2[at++] = make_uint2(dv[x], w_long(dv[x], nmv[x])); else nbl[at++] = dv[x]; }
            __syncthreads();
            K = misc[FS_NB];
        }
        }
        }

        FAST_TICK(4);
        if constexpr (SRN_FAST_ROW_PREFETCH == 2 && !MID && MODE == FM_FUSED && !FRAG) asm volatile("" :: "v"(pfv[0]), "v"(pfv[1]), "v"(pfv[2]), "v"(pfv[3]));
        FAST_PRIO(FP_REQ); FAST_PRIO_X(10);
        if constexpr (MODE == FM_FRONT) {   // the neighbour list leaves for the exchange buffer (a barrier stands between its last write and here in every branch above)
            if (tid == 0) xq[0] = K;
#pragma unroll
            for (int j = 0; j < 3; ++j) { const uint32_t e = tid + (uint32_t)j * BLOCK; if ((uint32_t)j * BLOCK < K && e < K) xq[1u + e] = nbl[e]; }
            continue;
        }

        // ---- walk A: one neighbour row per lane; the first 16 bytes (6 items) of all the wave's rows requested at once, the next 16 only where a row has them ----
        uint32_t svr[3]; uint4 rq[3], rq1[3];
        uint32_t wvr[3] = {0u, 0u, 0u};   // LONG: the neighbours' signed weights (everywhere else a weight is a table look-up on the slot's list set)
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const uint32_t j = wave * 64u + lane + (uint32_t)t * BLOCK;
            if constexpr (MODE == FM_BACK) svr[t] = j < K ? xsv[t] : 0u;   // (slots past K are stale words: the idle lanes read the empty row)
            else if constexpr (LONG) { const uint2 e2 = nb2[K ? min(j, K - 1u) : 0u]; svr[t] = K ? e2.x : 0u; wvr[t] = e2.y; }
            else
            svr[t] = K ? nbl[min(j, K - 1u)] : 0u;   // (all loads unconditional: a load inside a branch is waited for at the branch's end)
            const size_t r = (MODE == FM_BACK ? j < K : K != 0u) ? (size_t)(base + (svr[t] >> NB)) : (size_t)n_kept;   // (idle lanes re-read the last neighbour's row)
            rq[t] = *reinterpret_cast<const uint4*>(&f.row_packed[r * (FRAG ? 1 : 4)]); rq1[t] = make_uint4(0u, 0u, 0u, 0u);
        }
        if (((SRN_FAST_PRIO_X) / 10) % 10 != 9) FAST_PRIO(FP_REQ);
        {   // clear: accumulators + sketch + dump, exact table (keys EMPTY32, sums 0)
            uint4* z = reinterpret_cast<uint4*>(smem + F_HOT);
            for (uint32_t i = tid; i < (clr_lo - F_HOT) / 16u; i += BLOCK) z[i] = make_uint4(0u, 0u, 0u, 0u);   // (what the merge buffers covered: the rest was cleared at the top of the query)
            if (tid < 64u) reinterpret_cast<uint4*>(thist)[tid] = make_uint4(0u, 0u, 0u, 0u);
            if (tid < F_TABLE_WORDS / 4u) { reinterpret_cast<uint4*>(ikeys)[tid] = make_uint4(EMPTY32, EMPTY32, EMPTY32, EMPTY32);
                                            reinterpret_cast<uint4*>(iacc)[tid] = make_uint4(0u, 0u, 0u, 0u); }
        }
        __syncthreads();   // (also: every wave holds its neighbour slots in registers, the list's LDS is free for the queue)
        FAST_PRIO(FP_WALK);
        FAST_TICK(8);
        // (LONG: a negative weight goes to the exact words only -- direct-mapped and replicas: offsets below the sketch --; a sketch word sums positive parts and stays an
        //  upper bound of every item that shares it; a dump word takes anything)
        auto add2 = [&](uint32_t wd, uint32_t w) {
            if constexpr (LONG) {
                const bool neg = (int)w < 0; const uint32_t a = wd & 0xFFFFu, b = wd >> 16;
                if (!neg || a < F_SKETCH - F_HOT) atomicAdd((uint32_t*)(acc_base + a), w);
                if (!neg || b < F_SKETCH - F_HOT) atomicAdd((uint32_t*)(acc_base + b), w);
            } else {
            atomicAdd((uint32_t*)(acc_base + (wd & 0xFFFFu)), w); atomicAdd((uint32_t*)(acc_base + (wd >> 16)), w); } };
        constexpr uint32_t INL1 = FRAG ? 6u : 14u, INL2 = FRAG ? 20u : 28u;   // items served by round (i) / before the overflow loop
        // rows of > 14 items queue for round (ii) in the wave's own 192 list slots (their session slots are in registers by now)
        uint32_t cnt3 = 0;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const bool m3 = wave * 64u + lane + (uint32_t)t * BLOCK < K && (rq[t].x & 0xFFFFu) > INL1;
            const unsigned long long b3 = __ballot(m3);
            if (m3) nbl[qpos(cnt3 + (uint32_t)__popcll(b3 & lt))] = LONG ? wave * 64u + lane + (uint32_t)t * BLOCK : svr[t];   // (LONG queues the neighbour's INDEX: slot and weight are looked up in nb2)
            cnt3 += (uint32_t)__popcll(b3);
        }
#ifdef SRN_FAST_SUBTICKS
        FAST_TICK(5);   // (wave 0 has its rows' first 16 bytes)
#endif
        // The second 16 bytes (items 6..13) are asked for only now, and only by the rows that have them (1 in 4): the other lanes all read the empty
        // slot's -- one line for the lot -- so the address unit sees a quarter of the requests; the lines themselves came with the first 16 bytes.
        if constexpr (!FRAG) {
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const bool more6 = wave * 64u + lane + (uint32_t)t * BLOCK < K && (rq[t].x & 0xFFFFu) > 6u;
                const size_t r1 = more6 ? (size_t)(base + (svr[t] >> NB)) : (size_t)n_kept;
                rq1[t] = *reinterpret_cast<const uint4*>(&f.row_packed[r1 * 4 + 1]);
            }
        }
        // round (ii)'s first batch is requested BEFORE round (i)'s adds, which hide its latency (L2 hits: the lines came with round (i))
        // (FRAG: the queued fragments continue in their overflow blocks: c4 = items 4..11, d4 = items 12..19 if the fragment has that many; blk = the block the
        //  overflow loop starts at)
        auto tok_slot = [&](uint32_t tok) -> uint32_t { if constexpr (LONG) return nb2[tok].x; else return tok; };
        auto tok_w = [&](uint32_t tok, uint32_t slot) -> uint32_t { if constexpr (LONG) return nb2[tok].y; else return w10_of(slot & NBM); };
        // (tok: what the queue holds for the lane -- the slot itself, or (LONG) the neighbour's index)
        auto q3_load = [&](uint32_t p0, uint32_t& tok, uint32_t& sv, uint32_t& hdr, uint4& c4, uint4& d4, uint32_t& blk) {
            tok = nbl[qpos(min(p0 + lane, cnt3 - 1u))];
            sv = tok_slot(tok);
            if constexpr (FRAG) {
                const uint4 s4 = *reinterpret_cast<const uint4*>(f.row_packed + (size_t)(base + (sv >> NB)));
                hdr = s4.x;
                const uint4* eb = reinterpret_cast<const uint4*>(f.row_ext16) + ((hdr & 0xFFFFu) > 6u ? (size_t)s4.w : (size_t)0);   // (an idle lane's slot may be a short one)
                c4 = eb[0]; d4 = eb[1]; blk = s4.w + 2u;
            } else {
                const RowQuad* rowp = f.row_packed + (size_t)(base + (sv >> NB)) * 4;
                hdr = *reinterpret_cast<const uint32_t*>(rowp); c4 = *reinterpret_cast<const uint4*>(rowp + 2); d4 = *reinterpret_cast<const uint4*>(rowp + 3); blk = d4.w;
            } };
        uint32_t tk3 = 0, sv3 = 0, hdr3 = 0, blk3 = 0; uint4 c43 = make_uint4(0u, 0u, 0u, 0u), d43 = c43;
        if (cnt3) q3_load(0u, tk3, sv3, hdr3, c43, d43, blk3);   // (wave-uniform)
        // the next query's record: requested by wave 1 HERE -- behind its own row requests, with no other load of the wave due for thousands of cycles (loads return
        // in order: anywhere else the record's HBM round trip would sit in front of data the wave needs at once); it lands in LDS by itself, the wave
        // waits for it at the end of phase 4a, before the barrier that everybody passes on the way to the next query
        const uint32_t qn = MID ? q_after : ordered || listed ? q_ord_next : q + gridDim.x;
        const bool park = MID ? q_after != 0xFFFFFFFFu : ordered || listed ? q_ord_next != 0xFFFFFFFFu : qn < p.nq;   // (block-uniform)
        if (wave == 1u && park) {   // (a record is at most 72 + 8 * 24 = 264 bytes -- MID: 72 + 10 * 24 = 312 --: two rounds of a dword per lane)
            const char* const rn = p.prep + (size_t)qn * p.prep_stride + lane * 4u;
            if (lane * 4u < p.prep_stride) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)rn, (__attribute__((address_space(3))) void*)pre, 4, 0, 0);
            if (256u + lane * 4u < min(p.prep_stride, 320u)) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rn + 256), (__attribute__((address_space(3))) void*)(pre + 64), 4, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) {   // (i) items 0..13
            if (wave * 64u + (uint32_t)t * BLOCK < K) {
                const bool act = wave * 64u + lane + (uint32_t)t * BLOCK < K;
                const uint32_t w = LONG ? wvr[t] : w10_of(svr[t] & NBM), len = rq[t].x & 0xFFFFu;
                if constexpr (FRAG) { if (act) { add2(rq[t].y, w); add2(rq[t].z, w); if (len <= 6u) add2(rq[t].w, w); } }
                else {
                if (act) { add2(rq[t].y, w); add2(rq[t].z, w); add2(rq[t].w, w); }
                if (act && len > 6u) { add2(rq1[t].x, w); add2(rq1[t].y, w); add2(rq1[t].z, w); add2(rq1[t].w, w); }
                }
            }
        }
        auto add_tail = [&](uint32_t p0, uint32_t tok, uint32_t sv, uint32_t hdr, const uint4& c4, const uint4& d4, uint32_t blk) {   // items 14..29 (FRAG: 4..19), then the overflow blocks
            const bool act = p0 + lane < cnt3;
            const uint32_t len = hdr & 0xFFFFu, w = tok_w(tok, sv);
            const bool more = FRAG ? true : len > 30u;   // (a slot without overflow blocks keeps items in the words the others use for the block index)
            if constexpr (FRAG) {
                if (act) { add2(c4.x, w); add2(c4.y, w); add2(c4.z, w); add2(c4.w, w); }
                if (act && len > 12u) { add2(d4.x, w); add2(d4.y, w); add2(d4.z, w); add2(d4.w, w); }   // (a fragment of <= 12 items has one block: the next belongs to another row)
            } else if (act) {
                add2(c4.x, w); add2(c4.y, w); add2(c4.z, w); add2(c4.w, w);
                add2(d4.x, w); add2(d4.y, w); add2(d4.z, w);
                if (len <= 30u) add2(d4.w, w);
            }
            for (uint32_t t8 = INL2; __ballot(act && more && t8 < len) != 0ull; t8 += 8u) {
                if (act && more && t8 < len) {
                    const uint4 e4 = reinterpret_cast<const uint4*>(f.row_ext16)[(size_t)blk + ((t8 - INL2) >> 3)];
                    add2(e4.x, w); add2(e4.y, w); add2(e4.z, w); add2(e4.w, w);
                }
            } };
#ifdef SRN_FAST_SUBTICKS
        FAST_TICK(6);   // (round (i): second 16 bytes + adds issued)
#endif
        FAST_PRIO_X(1);
        if (cnt3) add_tail(0u, tk3, sv3, hdr3, c43, d43, blk3);
        for (uint32_t p0 = 64u; p0 < cnt3; p0 += 64u) { uint32_t tk, sv, hdr, blk; uint4 c4, d4; q3_load(p0, tk, sv, hdr, c4, d4, blk); add_tail(p0, tk, sv, hdr, c4, d4, blk); }
#ifdef SRN_FAST_SUBTICKS
        FAST_TICK(7);   // (wave 0's tail rounds done; what follows is the wait for the other waves)
#endif
        __syncthreads();
        FAST_PRIO(FP_HARV);
        FAST_TICK(9);
        // ---- phase 4a: the direct-mapped items, exactly -> threshold, candidates ----------------------------
        // Sample = the 512 most popular items, dealt round-robin to the waves: every wave takes the 3rd largest of its 64 values
        // of x = idf_eff * acc (top 32 bits of the f64: a monotone truncation; how_many > 24: the ceil(n / 8)-th largest); the smallest of the 8 has >= 24 >= n items at or
        // above it.  Everything is kept down to one step (2^-20 relative) BELOW it, so that what is dropped is strictly smaller
        // after the division by 10 U as well (ties at the cut included).
        uint32_t t32m1; uint32_t floor_b;
        {
            const uint32_t e = lane * NW + wave;
            uint32_t v = hot[e];
            if (F_REP_ITEMS && e < F_REP_ITEMS) {   // a replicated item: its sum is spread over F_REP words (its own word stays 0)
                const uint4* rp = reinterpret_cast<const uint4*>(hot + F_DIRECT + e * F_REP);
                const uint4 a = rp[0], b = rp[1];
                v = a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
            }
            if (!SRN_FAST_PHANTOM_OOR || BIG)   // (the 53 KB forms: out of range -- nothing there to zero, walk B's reads of it return 0)
            for (uint32_t i = tid; i < F_PHANTOM_WORDS; i += BLOCK) ((uint32_t*)(smem + F_PHANTOM))[i] = 0u;   // (walk B reads the phantom words, where the positions past a row's end point, as "cannot reach the floor")
            bool valid = (LONG ? (int)v > 0 : v != 0u) && e != cur_idx;   // (LONG: sums are signed; an item whose sum is <= 0 is no candidate and sets no threshold)
            double x = 0.0; uint32_t tie = 0;
            // idf_eff and the attribute byte come from the thread's LDS slot; only the id rank (wanted after the barrier below, for the few candidates) is a
            // global load -- nothing on the way to the wave reductions waits for memory
            tie = f.meta_sample[tid].id_rank;   // (coalesced, unconditional)
            if (business) valid = valid && business_ok(cur_attr, (uint32_t)((const uint8_t*)(smem + SATTR))[tid]);   // an item the rules exclude is no candidate and sets no threshold
            if (valid) x = ((const double*)(smem + SIDF))[tid] * (double)v;
            const uint32_t k32 = (uint32_t)((unsigned long long)__double_as_longlong(x) >> 32);
            {
                uint32_t vv = k32, third = 0;
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const uint32_t mx = wave_max(vv);
                    third = mx;
                    const unsigned long long bal = __ballot(vv == mx);
                    if ((int)lane == __ffsll((long long)bal) - 1) vv = 0;   // take one holder of the maximum out
                }
                // (round 6) how_many up to 64: the j-th largest of every wave, j = ceil(n / 8) -- the smallest of the 8 then has >= 8 j >= n items at or above it
                // (launch-uniform; the default n = 21 runs the three steps above and nothing else)
                for (uint32_t t = 3u; t < ((p.how_many + 7u) >> 3); ++t) {
                    const uint32_t mx = wave_max(vv);
                    third = mx;
                    const unsigned long long bal = __ballot(vv == mx);
                    if ((int)lane == __ffsll((long long)bal) - 1) vv = 0;
                }
                if (lane == 0u) misc[FS_W3 + wave] = third;   // (0 if the wave has < max(3, ceil(n / 8)) valid items)
            }
            __syncthreads();
            uint32_t t32 = 0xFFFFFFFFu;
#pragma unroll
            for (int w = 0; w < NW; ++w) t32 = min(t32, misc[FS_W3 + w]);
            // no threshold from the sample (t32 == 0: a small query): everything valid is a candidate, every element of a non-popular
            // item is listed; the caps of the lists decide whether the query fits (the general kernel takes it otherwise)
            t32m1 = t32 ? t32 - 1u : 0u;
            const bool take = valid && k32 >= t32m1;
            const uint32_t at = wave_append(take, &misc[FS_CCNT]);
            if (take) { if (at < F_CAND_CAP) { ckey[at] = (unsigned long long)__double_as_longlong(x); cidx[at] = tie; } else misc[FS_FAIL] = 1;   // (FS_FAIL: 1 candidates, 2 floor survivors, 4 hit list, 8 exact table)
                        const uint32_t hb = k32 >> 16, tb = t32 >> 16; if (t32) atomicAdd(&thist[hb > tb ? min(hb - tb, 255u) : 0u], 1u); }   // (>= 24 entries, few more: no pile-up)
            // integer floors: an item needs idf * acc >= x_lo, i.e. acc >= x_lo / (largest idf of its chunk); shaved so that rounding
            // can only keep more.  Lane c computes chunk c's floor (lane 8: the sketch words'), broadcast by v_readlane.
            const double x_lo = __longlong_as_double((long long)((unsigned long long)t32m1 << 32));
            const double inv = ((const double*)(smem + SINV))[min(lane, 8u)];
            const uint32_t my_floor = (uint32_t)fmin(4.0e9, fmax(1.0, floor(x_lo * inv * (1.0 - 1e-9)) - 1.0));
            floor_b = (uint32_t)__builtin_amdgcn_readlane((int)my_floor, 8);
            // The other direct-mapped words, four per thread and read: a thread's quad lies inside one chunk of 512 words and a wave's quads inside one chunk too
            // (quad u = 128 + tid: chunk 1 + wave / 2; quad 640 + tid: chunk 5 + wave / 2), so the chunk's floor is wave-uniform and the usual case -- no word of the
            // quad at its floor (1.1 survivors per query on config 3) -- costs one 16-byte read, three v_max and one compare per four items.
            {
                const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave);
                const uint4* h4 = reinterpret_cast<const uint4*>(hot);
                static_assert(F_HOT_WORDS == 4096 || F_HOT_WORDS == 3840, "quads per thread below (3840: the 40 KB timing experiment)");
                constexpr uint32_t NQ2 = F_HOT_WORDS / 4u - 128u - (uint32_t)BLOCK;   // quads of the second round (384 for 4096 words)
                const uint4 a4 = h4[128u + tid];
                const uint4 b4 = tid < NQ2 ? h4[128u + (uint32_t)BLOCK + tid] : make_uint4(0u, 0u, 0u, 0u);
                if (tid == NQ2 - 1u) hot[F_HOT_WORDS - 1u] = 0u;   // (the word walk B reads for every popular item: a replica word, summed by the sample above before the barrier)
                const uint32_t fl1 = (uint32_t)__builtin_amdgcn_readlane((int)my_floor, (int)(1u + (wv >> 1)));          // floors of this wave's two chunks (lane c of my_floor = chunk c)
                const uint32_t fl2 = (uint32_t)__builtin_amdgcn_readlane((int)my_floor, (int)min(5u + (wv >> 1), 7u));
                auto quad = [&](const uint4& q4, uint32_t e0, uint32_t fl) {
                    const uint32_t mx = LONG ? (uint32_t)max(max(max((int)q4.x, (int)q4.y), max((int)q4.z, (int)q4.w)), 0) : max(max(q4.x, q4.y), max(q4.z, q4.w));   // (LONG: signed sums)
                    if (__ballot(mx >= fl) == 0ull) return;
                    const uint32_t vv[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
                    for (uint32_t j = 0; j < 4u; ++j) {
                        const uint32_t e2 = e0 + j;
                        const bool pass = (LONG ? (int)vv[j] >= (int)fl : vv[j] >= fl) && e2 != cur_idx && e2 < F_DIRECT;   // (the words from F_DIRECT up are the replicas of the hottest items, already in the sample)
                        const uint32_t at2 = wave_append(pass, &misc[FS_SURV]);
                        if (pass) { if (at2 < SURV_CAP && vv[j] < (1u << 20)) surv[at2] = (e2 << 20) | vv[j]; else atomicOr(&misc[FS_FAIL], 2u); }   // (a survivor packs 20 bits of sum: LONG sums can pass them -- the general kernel then)
                    } };
                quad(a4, 512u + 4u * tid, fl1);
                quad(b4, 512u + 4u * ((uint32_t)BLOCK + tid), fl2);
            }
        }
        __syncthreads();
        {   // The first threshold is loose (a wave whose 3rd-best is weak drags it down) and the sketch floor with it.  Second, tight one:
            // the bin (top 16 bits of x: 1/16 octave) of the n-th best of the sample's candidates -- at least n valid items sit at
            // or above its lower edge.  Every wave scans the 256 bins itself (4 per lane, suffix sums from the top): no extra barrier.
            const uint4 h4 = reinterpret_cast<const uint4*>(thist)[lane];
            const uint32_t sum4 = h4.x + h4.y + h4.z + h4.w, pre = wave_incl_scan(sum4);
            uint32_t above = (uint32_t)__builtin_amdgcn_readlane((int)pre, 63) - pre;   // entries in bins owned by higher lanes
            const uint32_t need = p.how_many;
            const bool mine = above < need && need <= above + sum4;
            uint32_t bin = 4u * lane + 3u;
            if (above + h4.w < need) { above += h4.w; bin = 4u * lane + 2u;
                if (above + h4.z < need) { above += h4.z; bin = 4u * lane + 1u;
                    if (above + h4.y < need) bin = 4u * lane; } }
            const unsigned long long mb = __ballot(mine);   // (exactly one lane: the histogram holds >= 24 >= n entries)
            const uint32_t bsel = mb ? (uint32_t)__builtin_amdgcn_readlane((int)bin, __ffsll((long long)mb) - 1) : 0u;
            if (bsel) {
                const uint32_t t32b = (((t32m1 + 1u) >> 16) + bsel) << 16;
                t32m1 = t32b - 1u;
                const double x_lo = __longlong_as_double((long long)((unsigned long long)t32m1 << 32));
                floor_b = (uint32_t)fmin(4.0e9, fmax(1.0, floor(x_lo * ((const double*)(smem + SINV))[8] * (1.0 - 1e-9)) - 1.0));
            }
        }
        if (wave == 1u) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the record has landed
        FAST_TICK(10);
        FAST_PRIO(FP_WB);
        {
            const uint32_t ns = min(misc[FS_SURV], SURV_CAP);
            for (uint32_t i = tid; i < ns; i += BLOCK) {
                const uint32_t sv = surv[i], e = sv >> 20, v = sv & 0xFFFFFu;
                const ItemMeta mt = ix.meta[e];
                const double x = (mt.idf > 0.0 ? mt.idf : 1.0) * (double)v;
                const bool take = (uint32_t)((unsigned long long)__double_as_longlong(x) >> 32) >= t32m1 && (!business || business_ok(cur_attr, mt.attr));
                const uint32_t at = wave_append(take, &misc[FS_CCNT]);
                if (take) { if (at < F_CAND_CAP) { ckey[at] = (unsigned long long)__double_as_longlong(x); cidx[at] = mt.id_rank; } else misc[FS_FAIL] = 1; }
            }
        }
        // Is any sketch word at the floor at all?  (config 3: for 84 % of the queries none is -- no item outside the direct-mapped range can
        // make the top n -- and walk B, the resolve pass and the table scan are skipped: one 16-byte read and a v_max3 per 4 words.)
        {
            const uint4* sk4 = reinterpret_cast<const uint4*>(smem + F_SKETCH);
            uint32_t mx = 0;
            for (uint32_t i = tid; i < F_SK_WORDS / 4u; i += BLOCK) { const uint4 w4 = sk4[i]; mx = max(max(mx, w4.x), max(max(w4.y, w4.z), w4.w)); }
            if (__ballot(mx >= floor_b) != 0ull && lane == 0u) misc[FS_LIVE] = 1u;
        }
        __syncthreads();
        have_pre = park;   // (the barrier above orders wave 1's writes before anybody's next look)
        const bool live = misc[FS_LIVE] != 0u;   // block-uniform
        if (live) {
        // ---- walk B: an element reaches the exact table only if its sketch word can still reach the floor -----------
        // (all elements of an item share the word, so an item is accumulated completely or not at all; the accumulator words of
        // the popular items read 0 by now; a dump word that happens to reach the floor belongs to a position past its row's end).
        // The walk itself touches no global memory in round (i): it only LISTS the elements whose word is live, as (session slot,
        // row position) pairs; the list is resolved afterwards, one element per thread, all item-id fetches in flight together.
        {
            uint32_t dbg_hits = 0;
            // (offsets below the sketch -- popular items -- are mapped to the last direct-mapped word, which reads 0: one v_max per position
            //  buys the whole direct-mapped area for the hit list)
            auto chk2 = [&](uint32_t hm, uint32_t wd) -> uint32_t {   // two more positions, most recent in bit 0 ... appended at the top
                const uint32_t a = *(const uint32_t*)(acc_base + max(wd & 0xFFFFu, F_ZERO_OFF)), b = *(const uint32_t*)(acc_base + max(wd >> 16, F_ZERO_OFF));
                return (hm << 2) | (a >= floor_b ? 2u : 0u) | (b >= floor_b ? 1u : 0u); };   // (bit (n - 1 - i) = position i of the n checked)
            auto list_hits = [&](uint32_t hm, uint32_t npos, uint32_t sv, uint32_t j0) {   // hm: bit (npos - 1 - i) = position j0 + i
                const uint32_t c = (uint32_t)__popc(hm);
                if (__ballot(c != 0u) == 0ull) return;
                dbg_hits += c;
                const uint32_t inc = wave_incl_scan(c);
                uint32_t base = 0; if (lane == 63u) base = atomicAdd(&misc[FS_HITS], inc);
                uint32_t at = (uint32_t)__builtin_amdgcn_readlane((int)base, 63) + inc - c;
                while (hm) {
                    const uint32_t b = 31u - (uint32_t)__clz((int)hm); hm &= ~(1u << b);
                    if (at < F_HIT_CAP) hits[at] = make_uint2(sv, j0 + (npos - 1u - b)); else atomicOr(&misc[FS_FAIL], 4u);
                    ++at;
                }
            };
            // the usual case costs one v_max3 per two positions: the largest word of the lane's row; only a wave that sees a word at the
            // floor looks again, position by position
            auto mx2 = [&](uint32_t mx, uint32_t wd) -> uint32_t {
                const uint32_t a = *(const uint32_t*)(acc_base + max(wd & 0xFFFFu, F_ZERO_OFF)), b = *(const uint32_t*)(acc_base + max(wd >> 16, F_ZERO_OFF));
                return max(max(mx, a), b); };
            uint32_t tk3b = 0, sv3b = 0, hdr3b = 0, blk3b = 0; uint4 c43b = make_uint4(0u, 0u, 0u, 0u), d43b = c43b;
#if SRN_FAST_RELOAD_ROWS
            // the rows' first 32 bytes again (L2 hits), all requested at once: keeping them in registers since walk A costs 24 VGPRs
            // across phase 4a -- at the 80-register cap of three workgroups per CU that means scratch spills on the serial paths
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const size_t r = K ? (size_t)(base + (svr[t] >> NB)) : (size_t)n_kept;
                rq[t] = *reinterpret_cast<const uint4*>(&f.row_packed[r * 4]);
                rq1[t] = *reinterpret_cast<const uint4*>(&f.row_packed[r * 4 + 1]);
            }
#endif
            if (cnt3) q3_load(0u, tk3b, sv3b, hdr3b, c43b, d43b, blk3b);   // requested before round (i)
#pragma unroll
            for (int t = 0; t < 3; ++t) {   // (i) from the registers
                if (wave * 64u + (uint32_t)t * BLOCK < K) {
                    const bool act = wave * 64u + lane + (uint32_t)t * BLOCK < K, two = (rq[t].x & 0xFFFFu) > 6u;
                    uint32_t mx = 0;
                    if constexpr (FRAG) {   // positions 0..5; a fragment of > 6 items keeps only 0..3 here (word 3 = its first overflow block)
                        if (act) { mx = mx2(mx2(0u, rq[t].y), rq[t].z); if (!two) mx = mx2(mx, rq[t].w); }
                        if (__ballot(mx >= floor_b) != 0ull) {
                            uint32_t hm = 0;
                            if (act) { hm = chk2(chk2(0u, rq[t].y), rq[t].z); hm = two ? hm << 2 : chk2(hm, rq[t].w); }
                            list_hits(hm, 6u, LONG ? wave * 64u + lane + (uint32_t)t * BLOCK : svr[t], 0u);
                        }
                    } else {
                    if (act) { mx = mx2(mx2(mx2(0u, rq[t].y), rq[t].z), rq[t].w); if (two) mx = mx2(mx2(mx2(mx2(mx, rq1[t].x), rq1[t].y), rq1[t].z), rq1[t].w); }
                    if (__ballot(mx >= floor_b) != 0ull) {
                        uint32_t hm = 0;
                        if (act) {
                            hm = chk2(chk2(chk2(0u, rq[t].y), rq[t].z), rq[t].w) << 8;
                            if (two) hm = chk2(chk2(chk2(chk2(hm >> 8, rq1[t].x), rq1[t].y), rq1[t].z), rq1[t].w);
                        }
                        list_hits(hm, 14u, LONG ? wave * 64u + lane + (uint32_t)t * BLOCK : svr[t], 0u);   // (LONG lists the neighbour's INDEX: slot and weight come from nb2 at the resolve)
                    }
                    }
                }
            }
            auto chk_tail = [&](uint32_t p0, uint32_t sv, uint32_t hdr, const uint4& c4, const uint4& d4, uint32_t blk) {   // (sv: the queue's token)
                const bool act = p0 + lane < cnt3;
                const uint32_t len = hdr & 0xFFFFu;
                const bool more = FRAG ? true : len > 30u;
                uint32_t hm = 0, mx = 0;
                if constexpr (FRAG) {   // positions 4..11, then 12..19 if the fragment has a second block
                    const bool two = len > 12u;
                    if (act) { mx = mx2(mx2(mx2(mx2(0u, c4.x), c4.y), c4.z), c4.w); if (two) mx = mx2(mx2(mx2(mx2(mx, d4.x), d4.y), d4.z), d4.w); }
                    if (__ballot(mx >= floor_b) != 0ull) {
                        uint32_t hmd = 0;
                        if (act) { hm = chk2(chk2(chk2(chk2(0u, c4.x), c4.y), c4.z), c4.w); if (two) hmd = chk2(chk2(chk2(chk2(0u, d4.x), d4.y), d4.z), d4.w); }
                        list_hits(hm, 8u, sv, 4u); list_hits(hmd, 8u, sv, 12u);
                    }
                } else {
                if (act) { mx = mx2(mx2(mx2(mx2(mx2(mx2(mx2(0u, c4.x), c4.y), c4.z), c4.w), d4.x), d4.y), d4.z); if (len <= 30u) mx = mx2(mx, d4.w); }
                if (__ballot(mx >= floor_b) != 0ull) {
                if (act) {
                    hm = chk2(chk2(chk2(chk2(chk2(chk2(chk2(0u, c4.x), c4.y), c4.z), c4.w), d4.x), d4.y), d4.z);
                    if (len > 30u) hm <<= 2; else hm = chk2(hm, d4.w);
                }
                list_hits(hm, 16u, sv, 14u); }
                }
                for (uint32_t t8 = INL2; __ballot(act && more && t8 < len) != 0ull; t8 += 8u) {
                    uint32_t hm2 = 0;
                    if (act && more && t8 < len) {
                        const uint4 e4 = reinterpret_cast<const uint4*>(f.row_ext16)[(size_t)blk + ((t8 - INL2) >> 3)];
                        hm2 = chk2(chk2(chk2(chk2(0u, e4.x), e4.y), e4.z), e4.w);
                    }
                    list_hits(hm2, 8u, sv, t8);   // (rows of > 30 items are few: no fast path)
                } };
            if (cnt3) chk_tail(0u, tk3b, hdr3b, c43b, d43b, blk3b);
            for (uint32_t p0 = 64u; p0 < cnt3; p0 += 64u) { uint32_t tk, sv, hdr, blk; uint4 c4, d4; q3_load(p0, tk, sv, hdr, c4, d4, blk); chk_tail(p0, tk, hdr, c4, d4, blk); }
            if (ticking) { const uint32_t hs = wave_sum(dbg_hits); if (lane == 0u && hs) atomicAdd(&tacc[5], (unsigned long long)hs); }
        }
        __syncthreads();
        FAST_TICK(11);
        {   // resolve the listed elements: item id from the general row slot (its word 0 = the row's length: a position past the
            // end was a dump word), weight from the slot's position set, exact sums in the table
            const uint32_t nh = min(misc[FS_HITS], F_HIT_CAP);
            if (ticking && tid == 0 && nh == 0u) tacc[7] += 1ull;
            bool ovf = false;
            for (uint32_t i = tid; i < nh; i += BLOCK) {
                const uint2 h = hits[i];
                const uint32_t hslot = tok_slot(h.x);   // (LONG: h.x is the neighbour's index)
                const uint32_t* os = reinterpret_cast<const uint32_t*>(ix.row_slots + (size_t)(base + (hslot >> NB)) * (FRAG ? 1 : 4));
                const uint32_t len = os[0], j = h.y;
                uint32_t it = EMPTY32;
                if constexpr (FRAG) { if (j < len) it = len <= 3u ? os[1 + j] : (j < 2u ? os[2 + j] : ix.row_ext[os[1] + (j - 2u)]); }
                else
                if (j < len) it = len <= 15u ? os[1 + j] : (j < 14u ? os[2 + j] : ix.row_ext[os[1] + (j - 14u)]);
                if (business && it != EMPTY32 && it >= F_DIRECT && !business_ok(cur_attr, ix.meta[it].attr)) it = EMPTY32;   // (one more gather per listed element, all in flight together)
                if (it != EMPTY32 && it >= F_DIRECT && item_insert(ikeys, iacc, F_TABLE_BUCKETS, it, (int)tok_w(h.x, hslot)) < 0) ovf = true;
            }
            if (ovf) atomicOr(&misc[FS_FAIL], 8u);
        }
        __syncthreads();
        }
        FAST_TICK(12);
        if (misc[FS_FAIL]) {   // block-uniform
            if (tid == 0) { f.slow_list[atomicAdd(f.slow_cnt, 1u)] = q; if (ticking) { const uint32_t c = misc[FS_FAIL]; tacc[15] += (c & 1u) + ((unsigned long long)((c >> 1) & 1u) << 16) + ((unsigned long long)((c >> 2) & 1u) << 32) + ((unsigned long long)((c >> 3) & 1u) << 48); } }
            continue;
        }
        // ---- hand-off: wave 0 alone (the others go on to the next query's record and wait at its first barrier) ---------------
        // What is left -- the idf of the exact table's contenders, score = x / (10 U), the ranking, the public ids -- is a chain of
        // dependent gathers and a serial loop: on this workgroup's path it would cost 10 K cycles per query with seven waves idle.
        // The candidates and the table's contenders (exact sum at the floor; the others were collisions in their sketch word) are
        // written to a per-query record instead, and vmis_finish_kernel ranks all queries of the batch, one wave each.
        if (wave != 0u) continue;
        FAST_PRIO(FP_HAND);
        // (every lane-derived value of this tail is recomputed from an opaque copy of the lane id: hoisted out of the query loop, such
        //  values stay live through all phases, get spilled at the 80-register cap and come back from scratch right here, on the serial path)
        uint32_t ln = lane; asm volatile("" : "+v"(ln));
        const unsigned long long ltl = (1ull << ln) - 1ull;
        uint2* tl = hits;   // (the hit list is dead)
        uint32_t nt = 0;
        if (live)
#pragma unroll
        for (uint32_t b0 = 0; b0 < (F_TABLE_BUCKETS + 63u) / 64u; ++b0) {
            const uint32_t bk = b0 * 64u + ln;
            uint4 kq = make_uint4(EMPTY32, EMPTY32, EMPTY32, EMPTY32), aq = make_uint4(0u, 0u, 0u, 0u);
            if (bk < F_TABLE_BUCKETS) { kq = reinterpret_cast<const uint4*>(ikeys)[bk]; aq = reinterpret_cast<const uint4*>(iacc)[bk]; }
            const uint32_t kk[4] = {kq.x, kq.y, kq.z, kq.w}, aa[4] = {aq.x, aq.y, aq.z, aq.w};
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const bool in = kk[s4] != EMPTY32 && kk[s4] != cur_idx && (LONG ? (int)aa[s4] >= (int)floor_b : aa[s4] >= floor_b);   // (LONG: the table's sums are signed)
                const unsigned long long bm = __ballot(in);
                if (in) tl[nt + (uint32_t)__popcll(bm & ltl)] = make_uint2(kk[s4], aa[s4]);
                nt += (uint32_t)__popcll(bm);
            }
        }
        uint32_t cnt = misc[FS_CCNT];   // (this wave's own LDS traffic is ordered; the other waves' appends are behind the barrier above)
        // (round 6) The sample's candidates were taken at the FIRST, loose threshold; the second one (the bin of the n-th best: t32m1 now) came later.  With the default
        // n = 21 that leaves ~29 entries and nobody cares; with n = 50 the loose cut keeps 70-90 -- beyond the 63 a record holds, and the query would take the one-wave-per-
        // query finish-big kernel.  Only then (wave-uniform): the candidates below the tight threshold are dropped here, by the argument that dropped the survivors and the
        // contenders below it (>= n valid items sit at or above the bin's lower edge; everything down to one step below it is kept).
        if (cnt + nt > F_FIN_ENTRIES && cnt <= F_CAND_CAP) {
            unsigned long long xs[3]; uint32_t ti[3];
#pragma unroll
            for (uint32_t t = 0; t < 3u; ++t) { const uint32_t i = t * 64u + ln; xs[t] = i < cnt ? ckey[i] : 0ull; ti[t] = i < cnt ? cidx[i] : 0u; }
            uint32_t c2 = 0;
#pragma unroll
            for (uint32_t t = 0; t < 3u; ++t) {
                const bool keep = t * 64u + ln < cnt && (uint32_t)(xs[t] >> 32) >= t32m1;
                const unsigned long long bm = __ballot(keep);
                if (keep) { const uint32_t at = c2 + (uint32_t)__popcll(bm & ltl); ckey[at] = xs[t]; cidx[at] = ti[t]; }
                c2 += (uint32_t)__popcll(bm);
            }
            cnt = c2;
        }
        // record of query q: 1 KB at a FIXED place (vmis_finish_kernel then needs no index look-up before it can ask for the entries):
        // {M, U, 0, 0} | M <= 63 entries of 16 bytes: {x (f64 bits), id rank, 0} for a candidate, {sum, item, 1} for a contender of the table
        // A query with more entries (no threshold: a small query) puts the rest in an overflow arena and itself on the list of
        // vmis_finish_big_kernel: one 64-bit atomic hands out the list slot (high word) and the arena space (low word, entries).
        const uint32_t M = cnt + nt;
        if (MID && (LONG || L == 10u) && M < p.how_many) {   // (wave-uniform) neighbours of weight 0 may exist and the positive scores do not fill the top n: an item of score 0 can be returned (mod.rs:143-153 inserts it)
            if (ln == 0u) f.slow_list[atomicAdd(f.slow_cnt, 1u)] = q;
            continue;
        }
        if constexpr (TINY) {
            // A row of > 63 entries (no threshold from the sample: a small query whose every scored item is a candidate) finished HERE as well -- 3 % of config 3's single-session
            // calls, and each of them used to cost the call a second wait behind finish-big.  All M keys (x = idf_eff * acc; the contenders' idf gathered now) into the dead
            // sketch words; the n-th largest of their top 32 bits by a bitwise search over ballots; what is at or above ONE STEP BELOW it (so that what is dropped stays strictly
            // smaller after the division by 10 U, as everywhere) -- n entries and the ties at the cut -- compacted into the lanes and ranked by finish_inline.  More than 64
            // such entries: the record path below, as before.
            constexpr uint32_t TB_MAX = 704, TB_SLOTS = TB_MAX / 64;   // (M <= F_CAND_CAP + 4 * F_TABLE_BUCKETS = 668)
            static_assert(F_CAND_CAP + 4u * F_TABLE_BUCKETS <= TB_MAX && TB_MAX * 12u + 64u * 12u <= F_SK_WORDS * 4u, "the big row's keys fit the sketch words");
            if (M > F_FIN_ENTRIES && M <= TB_MAX && cnt <= F_CAND_CAP && M >= p.how_many) {   // (wave-uniform)
                unsigned long long* const kx = reinterpret_cast<unsigned long long*>(smem + F_SKETCH); uint32_t* const kt = reinterpret_cast<uint32_t*>(smem + F_SKETCH + TB_MAX * 8u);
                unsigned long long* const cx = reinterpret_cast<unsigned long long*>(smem + F_SKETCH + TB_MAX * 12u); uint32_t* const ct = reinterpret_cast<uint32_t*>(smem + F_SKETCH + TB_MAX * 12u + 64u * 8u);
                uint32_t hk[TB_SLOTS];
                {
                    uint2 cc[TB_SLOTS]; ItemMeta mm[TB_SLOTS];
#pragma unroll
                    for (uint32_t t = 0; t < TB_SLOTS; ++t) { const uint32_t i = t * 64u + ln; cc[t] = i >= cnt && i < M ? tl[i - cnt] : make_uint2(0u, 0u); }
#pragma unroll
                    for (uint32_t t = 0; t < TB_SLOTS; ++t) mm[t] = ix.meta[cc[t].x];   // (all gathers in flight together; a candidate's lane reads item 0's record and drops it)
#pragma unroll
                    for (uint32_t t = 0; t < TB_SLOTS; ++t) {
                        const uint32_t i = t * 64u + ln;
                        unsigned long long x = 0ull; uint32_t tie = EMPTY32;
                        if (i < cnt) { x = ckey[i]; tie = cidx[i]; }
                        else if (i < M) { x = (unsigned long long)__double_as_longlong((mm[t].idf > 0.0 ? mm[t].idf : 1.0) * (double)cc[t].y); tie = mm[t].id_rank; }
                        hk[t] = (uint32_t)(x >> 32);
                        if (i < M) { kx[i] = x; kt[i] = tie; }
                    }
                }
                uint32_t t32 = 0u;
                for (int b = 31; b >= 0; --b) {
                    const uint32_t c = t32 | (1u << b);
                    uint32_t n_ge = 0;
#pragma unroll
                    for (uint32_t t = 0; t < TB_SLOTS; ++t) n_ge += (uint32_t)__popcll(__ballot(hk[t] >= c));
                    t32 = n_ge >= p.how_many ? c : t32;
                }
                const uint32_t cut = t32 ? t32 - 1u : 0u;
                uint32_t S = 0;   // (wave-uniform) entries at or above the cut
#pragma unroll
                for (uint32_t t = 0; t < TB_SLOTS; ++t) {
                    const uint32_t i = t * 64u + ln;
                    const bool keep = i < M && hk[t] >= cut && hk[t] != 0u;
                    const unsigned long long bm = __ballot(keep);
                    const uint32_t at = S + (uint32_t)__popcll(bm & ltl);
                    if (keep && at < 64u) { cx[at] = kx[i]; ct[at] = kt[i]; }
                    S += (uint32_t)__popcll(bm);
                }
                if (S <= 64u && S >= p.how_many) {
                    __builtin_amdgcn_wave_barrier();   // (one wave, LDS in order: the compacted entries are there)
                    uint4 e = make_uint4(0u, 0u, 0u, 0u);
                    if (ln < S) { const unsigned long long x = cx[ln]; e = make_uint4((uint32_t)x, (uint32_t)(x >> 32), ct[ln], 0u); }
                    finish_inline(ix_arg, S, U, e, ln, q, p.out_ids, p.out_scores, p.out_counts, p.how_many);
                    continue;
                }
            }
        }
        uint32_t ovf_at = 0;
        if (M > F_FIN_ENTRIES) {   // (wave-uniform, rare: the atomic's round trip is paid by these queries only)
            unsigned long long tk = 0;
            if (ln == 0u) tk = atomicAdd(f.big_ticket, (1ull << 32) | (unsigned long long)(M - F_FIN_ENTRIES));
            const uint32_t slot = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(tk >> 32), 0);
            ovf_at = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)tk, 0);
            if (cnt > F_CAND_CAP || (unsigned long long)ovf_at + (M - F_FIN_ENTRIES) > f.big_cap_entries) {   // no room: the general kernel redoes the query
                if (ln == 0u) { f.big_list[slot] = 0xFFFFFFFFu; f.slow_list[atomicAdd(f.slow_cnt, 1u)] = q; }
                continue;
            }
            if (ln == 0u) f.big_list[slot] = q;
        }
        if constexpr (TINY) {
            if (M <= F_FIN_ENTRIES) {   // (wave-uniform) the row, finished here: no record, no finish kernel
                uint4 e = make_uint4(0u, 0u, 0u, 0u);
                if (ln < cnt) { const unsigned long long x = ckey[ln]; e = make_uint4((uint32_t)x, (uint32_t)(x >> 32), cidx[ln], 0u); }
                else if (ln < M) { const uint2 c = tl[ln - cnt]; e = make_uint4(c.y, 0u, c.x, 1u); }
                finish_inline(ix_arg, M, U, e, ln, q, p.out_ids, p.out_scores, p.out_counts, p.how_many);
                continue;
            }
        }
        {
            uint4* rec = reinterpret_cast<uint4*>(f.fin + (size_t)q * F_FIN_BYTES);
            uint4* ovf = reinterpret_cast<uint4*>(f.big_arena) + ovf_at;
            if (ln == 0u) { fin_store(&rec[0], make_uint4(M, U, ovf_at, 0u)); p.out_counts[q] = M > F_FIN_ENTRIES ? 0x80000001u : 0x80000000u; }   // (flags: a finish kernel completes the row)
#ifdef SRN_FAST_EXP_NOHANDOFF   // experiment (timing only, wrong results): what the serial copy of the candidates costs
            for (uint32_t i = ln; i < 0u; i += 64u) {
#else
            for (uint32_t i = ln; i < M; i += 64u) {
#endif
                uint4 e;
                if (i < cnt) { const unsigned long long x = ckey[i]; e = make_uint4((uint32_t)x, (uint32_t)(x >> 32), cidx[i], 0u); }
                else { const uint2 c = tl[i - cnt]; e = make_uint4(c.y, 0u, c.x, 1u); }
                if (i < F_FIN_ENTRIES) fin_store(&rec[1 + i], e); else ovf[i - F_FIN_ENTRIES] = e;
            }
        }
        if (ticking && ln == 0u) { tacc[6] += cnt; tacc[14] += 1ull; }
        FAST_TICK(13);
    }
    if (ticking) { __syncthreads(); if (tid < 16u && tacc[tid]) atomicAdd(&p.phase_cycles[tid], tacc[tid]); }
    if constexpr (TINY) {
        if (serving) { if (tid == 0u) { __threadfence_system(); __atomic_store_n(&(f.serve + blockIdx.x)->alive, 0u, __ATOMIC_RELAXED); } return; }
        // every append and every row of this workgroup was wave 0's (thread 0's atomics, the wave's stores): behind thread 0 in program order.  The workgroups count
        // themselves off; the last one publishes the counters and, behind a system-scope fence, the call's number -- the word the caller spins on.
        if (wave == 0u && f.host_words) __threadfence_system();   // (executed by EVERY lane that stored a row: a fence orders the executing thread's accesses -- ADVICE r5)
        if (tid == 0u && f.host_words) {
            if (atomicAdd(&f.slow_cnt[6], 1u) == gridDim.x - 1u) {
                f.host_words[1] = atomicAdd(&f.slow_cnt[0], 0u); f.host_words[2] = atomicAdd(&f.slow_cnt[1], 0u); f.host_words[3] = atomicAdd(&f.slow_cnt[4], 0u);
                f.host_words[4] = atomicAdd(&f.slow_cnt[3], 0u);
                atomicExch(&f.slow_cnt[6], 0u);   // (ready for the next call: where nothing was handed on, every counter is 0 again)
                __threadfence_system();
                __atomic_store_n(&f.host_words[5], f.host_seq, __ATOMIC_RELAXED);
            }
        }
    }
}

// -------------------------------------------------------------------------------------
// The tail of predict (mod.rs:156-214) for the queries the fast kernel served, one wave per query: idf of the exact table's
// contenders, score = idf_eff * acc / (10 U) (one multiply, one divide, in that order), top-n by (score desc, public id asc)
// -- ranks counted on the scores themselves --, public ids.  Rows of other queries are left alone.
// -------------------------------------------------------------------------------------
#ifndef SRN_FIN_QPW
#define SRN_FIN_QPW 4
#endif
constexpr uint32_t FIN_QPW = SRN_FIN_QPW;   // queries per wave: the three dependent round trips (record, contenders' idf, public ids) of FIN_QPW queries overlap
__global__ __launch_bounds__(256) void vmis_finish_kernel(DeviceIndex ix, const char* __restrict__ fin, uint64_t* __restrict__ out_ids, double* __restrict__ out_scores,
                                                          uint32_t* __restrict__ out_counts, uint32_t nq, uint32_t how_many, const uint32_t* __restrict__ cnt_slow, uint32_t* __restrict__ host_words) {
    // (the latency path: the call's path counters into pinned words -- handed to the general kernel, listed for MID, for MID's BIG form, queries with > 63 entries -- so that the
    //  host launches the kernels behind this one only for a call that has work for them)
    if (host_words && blockIdx.x == 0u && threadIdx.x == 0u) { host_words[1] = cnt_slow[0]; host_words[2] = cnt_slow[1]; host_words[3] = cnt_slow[4]; host_words[4] = cnt_slow[3]; }
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t q0 = (blockIdx.x * 4u + (threadIdx.x >> 6)) * FIN_QPW;
    if (q0 >= nq) return;
    // the records sit at fixed places: flag, header and this lane's entry of every query are requested together
    uint32_t flag[FIN_QPW]; uint4 hd[FIN_QPW], e[FIN_QPW];
#pragma unroll
    for (uint32_t u = 0; u < FIN_QPW; ++u) {
        const uint32_t q = min(q0 + u, nq - 1u);
        const uint4* rec = reinterpret_cast<const uint4*>(fin + (size_t)q * F_FIN_BYTES);
        flag[u] = q0 + u < nq ? out_counts[q] : 0u;
        hd[u] = fin_load(&rec[0]);
        e[u] = fin_load(&rec[1 + min(lane, 30u)]);   // the record's first 512 bytes: most queries have <= 31 entries (lanes past 30 re-read entry 30: same line)
    }
#pragma unroll
    for (uint32_t u = 0; u < FIN_QPW; ++u)   // (wave-uniform, rare: the second half of a long record)
        if (flag[u] == 0x80000000u && hd[u].x > 31u) { const uint4* rec = reinterpret_cast<const uint4*>(fin + (size_t)(q0 + u) * F_FIN_BYTES); e[u] = rec[1 + min(lane, F_FIN_ENTRIES - 1u)]; }
    bool valid[FIN_QPW]; double x[FIN_QPW]; uint32_t tie[FIN_QPW]; ItemMeta mt[FIN_QPW];
#pragma unroll
    for (uint32_t u = 0; u < FIN_QPW; ++u) {   // (flag != 0x80000000: not served by the fast kernel -- what was read is stale, and unused)
        valid[u] = flag[u] == 0x80000000u && lane < hd[u].x;
        mt[u] = ix.meta[valid[u] && e[u].w != 0u ? e[u].z : 0u];   // (unconditional: a load inside a branch is waited for at its end)
    }
#pragma unroll
    for (uint32_t u = 0; u < FIN_QPW; ++u) {
        x[u] = 0.0; tie[u] = EMPTY32;
        if (valid[u]) {
            if (e[u].w == 0u) { x[u] = __longlong_as_double((long long)(((unsigned long long)e[u].y << 32) | e[u].x)); tie[u] = e[u].z; }
            else { x[u] = (mt[u].idf > 0.0 ? mt[u].idf : 1.0) * (double)e[u].x; tie[u] = mt[u].id_rank; }
        }
    }
    unsigned long long pid[FIN_QPW];
#pragma unroll
    // (tried, round 5: the public ids fetched only for the entries that made the top n -- a dependent trip more, a quarter fewer requests: 0.750 against 0.757 ms for everything
    //  behind the fast kernel; 8 queries per wave: 1.02 ms; 2: 0.757 -- profiles/r05_fast_ab.txt)
    for (uint32_t u = 0; u < FIN_QPW; ++u) pid[u] = ix.id_sorted[valid[u] ? tie[u] : 0u];   // (arrive while the ranks are counted)
#pragma unroll
    for (uint32_t u = 0; u < FIN_QPW; ++u) {
        if (flag[u] != 0x80000000u) continue;   // (wave-uniform)
        const uint32_t q = q0 + u, M = hd[u].x;
        const double sc = valid[u] ? x[u] / (double)(10u * hd[u].y) : 0.0;
        const unsigned long long mk = (unsigned long long)__double_as_longlong(sc);   // (positive doubles order like their bit patterns)
        const int klo = (int)(uint32_t)mk, khi = (int)(uint32_t)(mk >> 32);
        // ranks: this kernel's time is this loop (2^20 queries x ~30 entries), so it first runs on the scores' top 32 bits alone -- entry j broadcast by
        // v_readlane, a compare and an add-with-carry each for "greater" and "equal" -- and only a wave in which two entries share those bits (near ties)
        // counts again on the full keys (score, then id rank)
        // (round 5: only "greater" is counted -- one readlane, one compare, one add-with-carry per entry; if no two valid entries share their top 32 bits the ranks are a
        //  permutation of 0 .. M - 1 and sum to M (M - 1) / 2, any tie leaves the sum short: one DPP wave sum instead of an "equal" count per entry)
        uint32_t rank = 0;
        for (uint32_t j = 0; j < M; ++j) {
            const uint32_t jh = (uint32_t)__builtin_amdgcn_readlane(khi, (int)j);
            rank += (uint32_t)(jh > (uint32_t)khi);
        }
        const uint32_t Mv = min(M, 64u);
        if (wave_sum(valid[u] ? rank : 0u) != Mv * (Mv - 1u) / 2u) {   // (wave-uniform)
            rank = 0;
            for (uint32_t j = 0; j < M; ++j) {
                const uint32_t jl = (uint32_t)__builtin_amdgcn_readlane(klo, (int)j), jh = (uint32_t)__builtin_amdgcn_readlane(khi, (int)j), ij = (uint32_t)__builtin_amdgcn_readlane((int)tie[u], (int)j);
                const unsigned long long kj = ((unsigned long long)jh << 32) | jl;
                rank += (uint32_t)(kj > mk) | ((uint32_t)(kj == mk) & (uint32_t)(ij < tie[u]));
            }
        }
        if (valid[u] && rank < how_many) { row_store(&out_ids[(size_t)q * how_many + rank], (uint64_t)pid[u]); row_store(&out_scores[(size_t)q * how_many + rank], sc); }
        if (lane == 0u) out_counts[q] = min(M, how_many);
    }
}
// The same for the few queries with more than 63 entries (no threshold from the sample: small queries whose every scored item is a
// candidate): one wave per listed query, entries staged in LDS; a threshold first -- 256-bin histogram of the scores' top 16 bits
// relative to the maximum, the bin of the n-th best -- then the ranks of what is at or above it.
__global__ __launch_bounds__(64) void vmis_finish_big_kernel(DeviceIndex ix, const char* __restrict__ fin, const uint4* __restrict__ arena, const uint32_t* __restrict__ big_list,
                                                             const unsigned long long* __restrict__ big_ticket, uint64_t* __restrict__ out_ids, double* __restrict__ out_scores,
                                                             uint32_t* __restrict__ out_counts, uint32_t how_many, const uint32_t* __restrict__ cnt_retry, const uint32_t* __restrict__ cnt_slow,
                                                             uint32_t* __restrict__ host_words) {
    constexpr uint32_t CAP = F_CAND_CAP + F_TABLE_BUCKETS * 4u;
    // the launch sequence's two counters (queries for the global-table pass, queries handed to the general kernel: both final before this kernel starts) straight into
    // the workspace's pinned words -- two 4-byte device-to-host copies cost 9 us of every call
    if (host_words && blockIdx.x == 0u && threadIdx.x == 0u) { if (cnt_retry) host_words[0] = *cnt_retry; if (cnt_slow) { host_words[1] = cnt_slow[0]; host_words[2] = cnt_slow[1]; host_words[3] = cnt_slow[4]; } }   // (cnt_slow[1]: the MID instantiation's list, [4]: its BIG form's)   // (a null source: that word is someone else's)
    __shared__ unsigned long long key[CAP];
    __shared__ uint32_t tieb[CAP];
    __shared__ uint32_t hist[256];
    const uint32_t lane = threadIdx.x;
    const uint32_t nbig = (uint32_t)(*big_ticket >> 32);
    for (uint32_t b = blockIdx.x; b < nbig; b += gridDim.x) {
        const uint32_t q = big_list[b];
        if (q == 0xFFFFFFFFu) continue;   // (handed to the general kernel)
        const uint4* rec = reinterpret_cast<const uint4*>(fin + (size_t)q * F_FIN_BYTES);
        const uint4 hd = rec[0];
        const uint32_t M = min(hd.x, CAP);
        const double denom = (double)(10u * hd.y);
        for (uint32_t i = lane; i < 256u; i += 64u) hist[i] = 0;
        uint32_t kmax = 0;
        for (uint32_t i = lane; i < M; i += 64u) {
            const uint4 e = i < F_FIN_ENTRIES ? rec[1 + i] : arena[hd.z + (i - F_FIN_ENTRIES)];
            double x; uint32_t tie;
            if (e.w == 0u) { x = __longlong_as_double((long long)(((unsigned long long)e.y << 32) | e.x)); tie = e.z; }
            else { const ItemMeta mt = ix.meta[e.z]; x = (mt.idf > 0.0 ? mt.idf : 1.0) * (double)e.x; tie = mt.id_rank; }
            const unsigned long long kb = (unsigned long long)__double_as_longlong(x / denom);
            key[i] = kb; tieb[i] = tie; kmax = max(kmax, (uint32_t)(kb >> 48));
        }
        kmax = wave_max(kmax);
        __syncthreads();   // (one wave; orders the LDS traffic)
        for (uint32_t i = lane; i < M; i += 64u) { const uint32_t hb = (uint32_t)(key[i] >> 48); atomicAdd(&hist[kmax - hb < 255u ? kmax - hb : 255u], 1u); }   // bin 0 = the best 1/16 octave
        __syncthreads();
        uint32_t cut = 255;   // first bin (from the best) at which the running count reaches n: everything in bins <= cut stays
        {
            const uint4 h4 = reinterpret_cast<const uint4*>(hist)[lane];
            const uint32_t s4 = h4.x + h4.y + h4.z + h4.w, inc = wave_incl_scan(s4);
            const uint32_t before = inc - s4;
            uint32_t mybin = 0xFFFFFFFFu;
            if (before < how_many && how_many <= inc) { mybin = 4u * lane; uint32_t c = before + h4.x; if (c < how_many) { ++mybin; c += h4.y; if (c < how_many) { ++mybin; c += h4.z; if (c < how_many) ++mybin; } } }
            const unsigned long long mb = __ballot(mybin != 0xFFFFFFFFu);
            if (mb) cut = (uint32_t)__builtin_amdgcn_readlane((int)mybin, __ffsll((long long)mb) - 1);
        }
        const uint32_t kcut = kmax >= cut && cut < 255u ? kmax - cut : 0u;   // keep entries whose top 16 bits are >= kcut (bin 255 collects everything below: keep all)
        uint32_t nk = 0;
        for (uint32_t i0 = 0; i0 < M; i0 += 64u) {   // compact the kept entries to the front (in place: the write index never passes the read index)
            const uint32_t i = i0 + lane;
            unsigned long long kb = 0; uint32_t tie = 0; bool keep = false;
            if (i < M) { kb = key[i]; tie = tieb[i]; keep = (uint32_t)(kb >> 48) >= kcut; }
            const unsigned long long bm = __ballot(keep);
            __syncthreads();
            if (keep) { const uint32_t at = nk + (uint32_t)__popcll(bm & ((1ull << lane) - 1ull)); key[at] = kb; tieb[at] = tie; }
            nk += (uint32_t)__popcll(bm);
            __syncthreads();
        }
        for (uint32_t i = lane; i < nk; i += 64u) {
            const unsigned long long mk = key[i]; const uint32_t tie = tieb[i];
            const unsigned long long pid = ix.id_sorted[tie];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < nk; j += 4u) {
                unsigned long long kj[4]; uint32_t ij[4];
#pragma unroll
                for (uint32_t u = 0; u < 4u; ++u) { const uint32_t jj = min(j + u, nk - 1u); kj[u] = key[jj]; ij[u] = tieb[jj]; }
#pragma unroll
                for (uint32_t u = 0; u < 4u; ++u) rank += (uint32_t)(j + u < nk) & ((uint32_t)(kj[u] > mk) | ((uint32_t)(kj[u] == mk) & (uint32_t)(ij[u] < tie)));
            }
            if (rank < how_many) { out_ids[(size_t)q * how_many + rank] = pid; out_scores[(size_t)q * how_many + rank] = __longlong_as_double((long long)mk); }
        }
        if (lane == 0u) out_counts[q] = min(nk, how_many);
        __syncthreads();
    }
}
hipError_t launch_finish_big(hipStream_t st, const DeviceIndex& di, const FastParams& f, uint64_t* out_ids, double* out_scores, uint32_t* out_counts, uint32_t how_many, uint32_t grid,
                             const uint32_t* cnt_retry, const uint32_t* cnt_slow, uint32_t* host_words) {
    hipLaunchKernelGGL(vmis_finish_big_kernel, dim3(grid), dim3(64), 0, st, di, (const char*)f.fin, (const uint4*)f.big_arena, (const uint32_t*)f.big_list, (const unsigned long long*)f.big_ticket,
                       out_ids, out_scores, out_counts, how_many, cnt_retry, cnt_slow, host_words);
    return hipGetLastError();
}

hipError_t launch_finish(hipStream_t st, const DeviceIndex& di, const FastParams& f, uint64_t* out_ids, double* out_scores, uint32_t* out_counts, uint32_t nq, uint32_t how_many,
                         const uint32_t* cnt_slow, uint32_t* host_words) {
    hipLaunchKernelGGL(vmis_finish_kernel, dim3((nq + 4 * FIN_QPW - 1) / (4 * FIN_QPW)), dim3(256), 0, st, di, (const char*)f.fin, out_ids, out_scores, out_counts, nq, how_many, cnt_slow, host_words);
    return hipGetLastError();
}

hipError_t launch_fast(dim3 grid, hipStream_t st, const DeviceIndex& di, const LaunchParams& p, const FastParams& f, bool debug, int mode, bool mid, bool big, bool lng, bool tiny) {
    constexpr int W = (int)F_WG_PER_CU;
    const bool wide = f.nb == 3u, frag = di.row_frag != 0u;
    if (mid && (mode != FM_FUSED || frag || (!big && !tiny && f.mid_list == nullptr))) return hipErrorInvalidValue;
    if (big && (!mid || (!lng && f.bigq_list == nullptr))) return hipErrorInvalidValue;
    if (lng && (!big || frag || f.long_list == nullptr)) return hipErrorInvalidValue;
    if (tiny && (big || mode != FM_FUSED || frag || grid.x != p.nq)) return hipErrorInvalidValue;   // (a workgroup per query: the last one to finish publishes the sequence's counters)
    void (*kern)(DeviceIndex, LaunchParams, FastParams) =
        tiny ? (mid ? vmis_fast_kernel<W, false, false, FM_FUSED, true, false, false, true> : wide ? vmis_fast_kernel<W, false, true, FM_FUSED, false, false, false, true> : vmis_fast_kernel<W, false, false, FM_FUSED, false, false, false, true>) :
        lng ? vmis_fast_kernel<W, false, false, FM_FUSED, true, true, true> :
        big ? vmis_fast_kernel<W, false, false, FM_FUSED, true, true> :
        mid ? vmis_fast_kernel<W, false, false, FM_FUSED, true> :
        mode == FM_FRONT ? (wide ? vmis_fast_kernel<W, false, true, FM_FRONT> : vmis_fast_kernel<W, false, false, FM_FRONT>)
        : mode == FM_BACK ? (wide ? (frag ? vmis_fast_kernel<W, true, true, FM_BACK> : vmis_fast_kernel<W, false, true, FM_BACK>)
                                  : (frag ? vmis_fast_kernel<W, true, false, FM_BACK> : vmis_fast_kernel<W, false, false, FM_BACK>))
                          : (wide ? (frag ? vmis_fast_kernel<W, true, true> : vmis_fast_kernel<W, false, true>)
                                  : (frag ? vmis_fast_kernel<W, true, false> : vmis_fast_kernel<W, false, false>));
    constexpr size_t dyn = SRN_FAST_SMALL ? 0 : F_TOTAL;
    if (dyn) { hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn); if (e != hipSuccess) return e; }
    static bool told = false;
    if (!told && debug) { told = true; int nb = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)kern, 512, dyn);
        fprintf(stderr, "[srn] vmis_fast_kernel: %u bytes of LDS, %d workgroups per CU (occupancy API)\n", F_TOTAL, nb); }
    hipLaunchKernelGGL(kern, grid, dim3(512), dyn, st, di, p, f);
    return hipGetLastError();
}

hipError_t launch_rows_to_packed(hipStream_t st, const uint64_t* row_off, const uint32_t* row_items, uint64_t n_rows, const uint32_t* block_base,
                                 uint32_t* packed, uint32_t* ext16, bool frag) {
    if (frag) hipLaunchKernelGGL(rows_to_packed_frag_kernel, dim3((unsigned)((n_rows + 1 + 1023) / 1024)), dim3(1024), 0, st, row_off, row_items, n_rows, block_base, packed, ext16);
    else
    hipLaunchKernelGGL(rows_to_packed_kernel, dim3((unsigned)((n_rows + 1 + 1023) / 1024)), dim3(1024), 0, st, row_off, row_items, n_rows, block_base, packed, ext16);
    return hipGetLastError();
}

}  // namespace srn

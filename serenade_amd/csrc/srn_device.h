// Device-side primitives shared by the predict kernels (srn_kernels.hip, srn_fast.hip): DPP wave scans, one-atomic-per-wave
// appends, the block scan and the exact item hash table.  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>

#include "srn_kernels.h"

namespace srn {

static constexpr int MAX_PROBES = 48;       // bucket probes (4 slots each) before a table is declared full

// passes_business_rules, src/vmisknn/mod.rs:162-182; attribute byte SRN_ATTR_NONE = None
__device__ __forceinline__ bool business_ok(uint32_t cur, uint32_t reco) {
    if (reco == SRN_ATTR_NONE) return false;
    if (reco & SRN_ATTR_FOR_SALE) {
        if (reco & SRN_ATTR_ADULT) return cur != SRN_ATTR_NONE && (cur & SRN_ATTR_ADULT);
        return true;
    }
    return false;
}
__device__ __forceinline__ uint64_t dev_mix64(uint64_t x) {   // the id table's hash (srn_index.cpp mix64)
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33; return x;
}

// Wave-wide scans and reductions on the VALU's data-parallel-primitive path (row_shr inside the rows of 16 lanes, then
// row_bcast 15 / 31 across them): ~12 VALU instructions.  The __shfl_* forms compile to ds_bpermute_b32, i.e. six DEPENDENT
// LDS round trips per scan, on a kernel whose LDS pipe is the busiest unit.
#define SRN_DPP(v, ctrl, rmask, bctl) (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), ctrl, rmask, 0xf, bctl)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {   // lane i <- v[0] + ... + v[i]
    v += SRN_DPP(v, 0x111, 0xf, true);    // row_shr:1 (0 shifted in at the row start)
    v += SRN_DPP(v, 0x112, 0xf, true);    // row_shr:2
    v += SRN_DPP(v, 0x114, 0xf, true);    // row_shr:4
    v += SRN_DPP(v, 0x118, 0xf, true);    // row_shr:8
    v += SRN_DPP(v, 0x142, 0xa, false);   // row_bcast:15 -> rows 1 and 3
    v += SRN_DPP(v, 0x143, 0xc, false);   // row_bcast:31 -> rows 2 and 3
    return v;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {   // (uniform result)
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan(v), 63);
}
__device__ __forceinline__ uint32_t wave_max(uint32_t v) {   // (uniform result)
    v = max(v, SRN_DPP(v, 0x111, 0xf, true));
    v = max(v, SRN_DPP(v, 0x112, 0xf, true));
    v = max(v, SRN_DPP(v, 0x114, 0xf, true));
    v = max(v, SRN_DPP(v, 0x118, 0xf, true));
    v = max(v, SRN_DPP(v, 0x142, 0xa, false));
    v = max(v, SRN_DPP(v, 0x143, 0xc, false));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// one LDS atomic per wave: returns this lane's slot in a shared append buffer (only meaningful if pred)
__device__ __forceinline__ uint32_t wave_append(bool pred, uint32_t* counter) {
    const unsigned long long mask = __ballot(pred);
    if (mask == 0) return 0;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)mask) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
    return base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}

// exclusive prefix sum of one value per thread over the block (thread order); `scratch` = NWAVES words of LDS that nobody
// else touches until the next barrier after the call; one barrier inside.  total = sum over the block.
template <int BLOCK>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* scratch, uint32_t& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t inc = wave_incl_scan(v);
    if (lane == 63) scratch[wave] = inc;
    __syncthreads();
    uint32_t base = 0; total = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) { const uint32_t t = scratch[w]; total += t; if (w < wave) base += t; }
    return base + inc - v;
}

// The serving order of a batch (FastParams::order, round 5): the batch sorted by each query's most popular item, dealt to the XCDs in CHUNKS of ORD_CHUNK consecutive
// positions, chunk c to XCD c % 8 (workgroup b runs on XCD b % 8; the grid is a multiple of 8).  Chunks, not eighths: the order runs from the most popular items -- the
// heaviest queries: full lists, k neighbours -- to the tail, and an eighth per XCD left the last XCD with the light eighth (25.7 ms against 22.8 unordered on config 3);
// chunk by chunk every XCD gets the same mix, heavy first, and its L2 still sees runs of like queries.  An XCD's positions, numbered u = 0, 1, ...: pos(u) below;
// workgroup b walks u = b / 8, + gridDim / 8, ... while u < ord_count(nq, b % 8).
#ifndef SRN_ORD_LG
#define SRN_ORD_LG 8
#endif
static constexpr uint32_t ORD_LG = SRN_ORD_LG, ORD_CHUNK = 1u << ORD_LG;
__device__ __forceinline__ uint32_t ord_pos(uint32_t x, uint32_t u) { return ((x + 8u * (u >> ORD_LG)) << ORD_LG) + (u & (ORD_CHUNK - 1u)); }
__device__ __forceinline__ uint32_t ord_count(uint32_t nq, uint32_t x) {
    const uint32_t nc = (nq + ORD_CHUNK - 1u) >> ORD_LG;                   // chunks in all; XCD x owns x, x + 8, ...
    if (nc <= x) return 0u;
    const uint32_t mine = (nc - x + 7u) >> 3, last = (nc - 1u) & 7u;        // the last chunk may be short
    return mine * ORD_CHUNK - (last == x ? nc * ORD_CHUNK - nq : 0u);
}

// insert-or-add into the exact item table: 4-slot buckets (one ds_read_b128 per probe), double hashing over a prime number of
// buckets, separate key / accumulator arrays; accumulators are signed.  Returns 1 if the item was new, 0 if it existed, -1 = table full
__device__ __forceinline__ int item_insert(uint32_t* ikeys, int* iacc, uint32_t nb, uint32_t it, int w) {
    uint32_t b = __umulhi(it * 0x9E3779B1u, nb);                       // nb is prime: any step in [1, nb) cycles through all buckets
    const uint32_t step = 1u + __umulhi(it * 0x85EBCA6Bu, nb - 1u);
    for (int probe = 0; probe < MAX_PROBES;) {
        const uint4 v = *reinterpret_cast<const uint4*>(&ikeys[4 * b]);
        const int hit = v.x == it ? 0 : v.y == it ? 1 : v.z == it ? 2 : v.w == it ? 3 : -1;
        if (hit >= 0) { atomicAdd(&iacc[4 * b + hit], w); return 0; }
        const int empty = v.x == EMPTY32 ? 0 : v.y == EMPTY32 ? 1 : v.z == EMPTY32 ? 2 : v.w == EMPTY32 ? 3 : -1;
        if (empty >= 0) {
            const uint32_t old = atomicCAS(&ikeys[4 * b + empty], EMPTY32, it);
            if (old == EMPTY32) { atomicAdd(&iacc[4 * b + empty], w); return 1; }
            if (old == it) { atomicAdd(&iacc[4 * b + empty], w); return 0; }
            continue;   // another item took that slot meanwhile: look at this bucket again
        }
        b += step; if (b >= nb) b -= nb;
        ++probe;
    }
    return -1;
}


// vmis_finish_kernel's work for ONE query whose <= 63 entries are still in the serving wave's registers (lane i: entry i): the latency path's fused launch (vmis_fast_kernel<TINY>) and the item shard's wave-per-query back end (srn_sback.hip).  Same
// arithmetic, same order: idf of a contender, x = idf_eff * acc, score = x / (10 U), rank = entries with a better (score desc, id rank asc) key.
__device__ __forceinline__ void finish_inline(const DeviceIndex& ix, uint32_t M, uint32_t U, const uint4& e, uint32_t ln, uint32_t q, uint64_t* out_ids, double* out_scores, uint32_t* out_counts, uint32_t how_many) {
    const bool have = ln < M;
    const ItemMeta mt = ix.meta[have && e.w != 0u ? e.z : 0u];
    double x = 0.0; uint32_t tie = EMPTY32;
    if (have) {
        if (e.w == 0u) { x = __longlong_as_double((long long)(((unsigned long long)e.y << 32) | e.x)); tie = e.z; }
        else { x = (mt.idf > 0.0 ? mt.idf : 1.0) * (double)e.x; tie = mt.id_rank; }
    }
    const unsigned long long pid = ix.id_sorted[have ? tie : 0u];
    const double sc = have ? x / (double)(10u * U) : 0.0;
    const unsigned long long mk = (unsigned long long)__double_as_longlong(sc);   // (positive doubles order like their bit patterns)
    const int klo = (int)(uint32_t)mk, khi = (int)(uint32_t)(mk >> 32);
    uint32_t rank = 0;
    for (uint32_t j = 0; j < M; ++j) {
        const uint32_t jl = (uint32_t)__builtin_amdgcn_readlane(klo, (int)j), jh = (uint32_t)__builtin_amdgcn_readlane(khi, (int)j), ij = (uint32_t)__builtin_amdgcn_readlane((int)tie, (int)j);
        const unsigned long long kj = ((unsigned long long)jh << 32) | jl;
        rank += (uint32_t)(kj > mk) | ((uint32_t)(kj == mk) & (uint32_t)(ij < tie));
    }
    if (have && rank < how_many) { out_ids[(size_t)q * how_many + rank] = pid; out_scores[(size_t)q * how_many + rank] = sc; }
    if (ln == 0u) out_counts[q] = min(M, how_many);
}


}  // namespace srn

// =====================================================================================
// The item-sharded index's BACK END of its own (round 5): predict's scoring (src/vmisknn/mod.rs:126-214) over ONE shard's row fragments, from a neighbour list that
// any rank's front end found (the shard group's neighbours pipeline, srn_group.hip) -- ONE WAVEFRONT PER EVOLVING SESSION, the north star's sketch.
//
// Why a kernel of its own.  Until round 4 the back end was an instantiation of vmis_fast_kernel (FM_BACK): a workgroup of 8 waves and 53 KB of LDS per query, sized for
// the UNSHARDED index (4 096 + 4 096 accumulator words, 38 KB of clears, a 512-item threshold sample, four barriers in the harvest).  A shard of G sees 1 / G of the items
// but ALL the queries, so every per-query fixed cost is paid G times per node: one rank's back end took 2.0 ms per 131 072 queries at G = 2, 4 AND 8 (profiles/r04_shard_rank_time.txt).
// With 1 / G of the items a query's state is small -- here 1 024 direct-mapped + 1 024 sketch words: 8 KB -- and fits a single wave: no barriers at all, 12 queries in flight
// per CU instead of 3, every per-query sweep (clear, sample, floors, sketch check) shrinks by 4-8x, and a lane's 24 neighbour slots are all requested at once.
//
// Rows: a third per-shard row array, `frag8` -- 8-byte slots addressed by recency rank, four 16-bit LDS byte offsets (the accumulator word of each of the <= 4 items of the
// fragment: direct-mapped for the shard's SB_DIRECT most popular items, a sketch word for the rest, replicated words for the 16 hottest -- the same scheme as srn_fast.hip;
// unused positions point OUT of the wave's LDS allocation (round 6: dropped by the hardware)); a fragment of > 4 items (rare from G = 8 on) keeps all its items in 16-byte overflow blocks -- and a PRESENCE BITMAP, one bit per
// session: at G = 8 about half of a query's neighbours hold no item of this shard at all, the bitmap (1.5 MB on config 3: L2-resident) is asked first and only the others
// cost a fragment fetch (tools/shard_gather_bench.hip: a random fragment fetch is an HBM-granule miss at ~50 G/s chip-wide, a bitmap hit ~15x cheaper).
//
// Same canonical semantics, same integers: the harvest is the fast kernel's -- threshold from the most popular items' exact sums, integer floors per chunk, the sketch filter
// (DESIGN.md "Why the sketch filter is exact"), walk B + exact table for what the sketch cannot exclude -- written wave-synchronously; the hand-off record and the finish
// kernels (vmis_finish_kernel / vmis_finish_big_kernel) are shared with the fast kernel, and so is the fall-back: what this kernel cannot take goes to f.slow_list and the
// general kernel behind it.
//
// Round 6 (1.60 -> 1.09 ms per 131 072 queries at G = 8, profiles/r06_sback_ab.txt): what the compiler made of the source mattered more than any memory-side redesign of
// round 5.  (1) NO SCRATCH: values made of the lane number are loop-invariant; hoisted out of the query loop they occupied registers for the wave's life, the kernel sat at its
// 168-register limit and they were spilled -- and a reload is a vector-memory load whose s_waitcnt vmcnt(0) also sits out every other request the wave has in flight
// (opaque copies of the lane number at the points of use; sample constants fetched per query).  (2) Loads the source asks for unconditionally were moved behind branches:
// the 24 slot loads below the early exits on K (a second dependent trip), loads under `cond ? load : 0` into divergent branches (each waited for there).  (3) A
// wave-uniform branch per chunk of 64 neighbours ends the scheduler's region: every chunk's LDS reads were waited for in its own block -- chunks now go in groups of four,
// walk B reads a group's sixteen words behind a scheduling barrier, one scan per query places the hits.  (4) Straight-line walks: unused positions and absent lanes add to /
// read from offsets out of the allocation.  (5) The next query's record and slots are requested before this query's resolve.
// =====================================================================================
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>

#include "srn_device.h"
#include "srn_kernels.h"

namespace srn {

#ifndef SRN_SBACK_PF
#define SRN_SBACK_PF 10   // chunks of 64 postings a wave of the streaming form keeps in flight
#endif
static constexpr uint32_t SB_PF = SRN_SBACK_PF;
// LDS map of one wave (= one workgroup), bytes
static constexpr uint32_t SB_MISC = 0, SB_WTAB = 128, SB_CAND = 192, SB_CAND_CAP = 128, SB_TABLE = SB_CAND + SB_CAND_CAP * 12 /* 1728 */, SB_BUCKETS = 61,
                          SB_TABLE_WORDS = 256 /* 61 buckets of 4 slots, padded */, SB_HOT = SB_TABLE + SB_TABLE_WORDS * 8 /* 3776 */;
static_assert(SB_H % 256 == 0 && (SB_S & (SB_S - 1)) == 0 && SB_DUMP == 64, "geometry");
static constexpr uint32_t SB_LDS = SB_HOT + (SB_H + SB_S + SB_DUMP) * 4;
static constexpr uint32_t SB_LQ_CAP = (SB_HOT - SB_CAND) / 8;   // long fragments a query may queue (448)
static constexpr uint32_t SB_LQB_CAP = 160, SB_HIT_CAP = (SB_H * 4 - SB_LQB_CAP * 12) / 8;   // walk B's hit list and its queue of long fragments live in the direct-mapped words (dead by then)
static_assert(SB_HOT % 16 == 0 && SB_TABLE % 16 == 0 && SB_CAND % 8 == 0, "alignment");
// The STREAMING form (see the kernel) reads, instead of the neighbour slots, the query's neighbours as POSITIONS in its items' posting lists (written by
// shard_nb_positions_kernel below on the rank that fronts the query): per run a bitmap over the kept prefix, then the members' list sets as nibbles in stream order.  Record
// of a query, 32-bit words: [0] K (0xFFFFFFFF: no front end took it)  [1] 0  [2 + 2 (r mw + c)] 64-bit word c of run r's bitmap, mw = ceil(m / 64), r < 4
// [2 + 8 mw + i / 8] nibble i % 8: list set of the i-th member.  In LDS the bitmaps and nibbles take the candidate buffer's and the exact table's room (dead until the harvest).
static constexpr uint32_t SB_SCR_ENTRIES = F_K_MAX + 128, SB_SCR_WORDS = 3 * SB_SCR_ENTRIES;   // a wave's scratch in HBM: the members' positions (4 B), then their fragments (8 B)
static constexpr uint32_t SB_REC_ROOM = SB_HOT - SB_CAND;   // bytes of bitmaps + nibbles a query may bring (3 584)
// the conversion kernel's own LDS: the neighbour slots as a hash table, 256 buckets of 8 slots, two candidate buckets per key (no overflow in 40 x 1 500 simulated keys), + the nibbles
static constexpr uint32_t SBP_BUCKETS = 256, SBP_CODES = SBP_BUCKETS * 32, SBP_WTAB = SBP_CODES + F_K_MAX / 2 + 32, SBP_LDS = SBP_WTAB + 64;

// -------------------------------------------------------------------------------------
// Attach time: an item shard's CSR row fragments -> frag8 slots + overflow blocks + presence bitmap
// -------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t sb_offset_of(uint32_t idx, uint64_t r) {
    if (idx < SB_REP_ITEMS) return (SB_DIRECT + idx * SB_REP + ((uint32_t)r & (SB_REP - 1u))) * 4u;
    return idx < SB_DIRECT ? idx * 4u : (SB_H + (idx & (SB_S - 1u))) * 4u;
}
// Unused positions of a fragment / an overflow block point BEYOND the wave's LDS allocation (round 6; until then: into a dump area of 64 words): a DS access out of the
// workgroup's range is dropped by the hardware before it costs a bank cycle -- adds vanish, reads return 0 (tools/lds_oor_bench.hip; the fast kernel's phantom rows, DESIGN
// section 4.5) -- so the walks add and read all four positions of every fragment WITHOUT a branch, and the 60 % of them that name no item collide with nobody.
static constexpr uint32_t SB_OOR = 0xC000u;   // (byte offset from the accumulators' base; the wave's whole allocation is 12 KB)
static_assert(SB_OOR >= 4u * SB_LDS && SB_OOR < 0xFFFFu && SB_OOR >= (SB_H + SB_S) * 4u, "out of the allocation whatever its granule");
// A fragment of > 4 items keeps its length and its first overflow block in four words that are ALL out-of-range offsets, 4-byte aligned (a misaligned DS atomic faults even
// out of range): {SB_LONG | block bits 22..31, SB_OOR | length, SB_OOR | block bits 0..10, SB_OOR | block bits 11..21}, every field shifted left by two (round 6: until
// then {0xFFFF, length, block index}, which the walks had to replace by out-of-range offsets before adding / reading them: two selects per chunk and walk)
static constexpr uint32_t SB_LONG = 0xE000u, SB_LF_BITS = 11u, SB_LF_MASK = (1u << SB_LF_BITS) - 1u;
static_assert(SB_OOR + (SB_LF_MASK << 2) < SB_LONG && SB_LONG + (0x3FFu << 2) + SB_HOT < 0x10000u, "fields do not reach the marker, and everything stays below 64 KB");
__device__ __forceinline__ bool sb_is_long(uint32_t x) { return (x & 0xFFFFu) >= SB_LONG; }
__device__ __forceinline__ uint32_t sb_long_len(uint32_t x) { return (x >> 18) & SB_LF_MASK; }
__device__ __forceinline__ uint32_t sb_long_block(uint32_t x, uint32_t y) { return ((y >> 2) & SB_LF_MASK) | (((y >> 18) & SB_LF_MASK) << SB_LF_BITS) | (((x >> 2) & 0x3FFu) << (2u * SB_LF_BITS)); }
__device__ __forceinline__ uint32_t sb_phantom(uint64_t, uint32_t) { return SB_OOR; }
__global__ __launch_bounds__(1024) void rows_to_frag8_kernel(const uint64_t* __restrict__ row_off, const uint32_t* __restrict__ row_items, uint64_t n,
                                                             const uint32_t* __restrict__ block_base, uint2* __restrict__ frag8, uint4* __restrict__ ext8, uint32_t* __restrict__ present) {
    __shared__ uint32_t wave_tot[16];
    const uint64_t r = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t o = 0, len = 0;
    if (r < n) { o = row_off[r]; len = row_off[r + 1] - o; }
    const uint32_t e = len > 4 ? (uint32_t)((len + 7) / 8) : 0u;   // overflow blocks of this fragment (ALL its items live there)
    const uint32_t inc = wave_incl_scan(e);
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    uint32_t base = block_base[blockIdx.x];
    for (int w = 0; w < wave; ++w) base += wave_tot[w];
    const uint32_t eblk = base + inc - e;
    // presence: bit (r & 31) of word r >> 5 (a wave covers two words)
    const unsigned long long pb = __ballot(len > 0);
    if ((lane & 31) == 0 && r <= n) present[r >> 5] = (uint32_t)(pb >> (lane & 32));
    if (r > n) return;
    uint32_t h[4];
    if (len <= 4) {
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) h[j] = j < len ? sb_offset_of(row_items[o + j], r) : sb_phantom(r, j);
    } else {
        h[0] = SB_LONG | ((eblk >> (2u * SB_LF_BITS)) << 2); h[1] = SB_OOR | ((uint32_t)len << 2); h[2] = SB_OOR | ((eblk & SB_LF_MASK) << 2); h[3] = SB_OOR | (((eblk >> SB_LF_BITS) & SB_LF_MASK) << 2);   // (the host admits the shard only with fragments of < 2 048 items)
        for (uint32_t b = 0; b < e; ++b) {
            uint32_t wv[4];
#pragma unroll
            for (uint32_t x = 0; x < 4; ++x) {
                const uint64_t j0 = 8ull * b + 2 * x;
                const uint32_t lo = j0 < len ? sb_offset_of(row_items[o + j0], r) : sb_phantom(r, (uint32_t)j0);
                const uint32_t hi = j0 + 1 < len ? sb_offset_of(row_items[o + j0 + 1], r) : sb_phantom(r, (uint32_t)j0 + 1);
                wv[x] = lo | (hi << 16);
            }
            ext8[(size_t)eblk + b] = make_uint4(wv[0], wv[1], wv[2], wv[3]);
        }
    }
    frag8[r] = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
}
hipError_t launch_rows_to_frag8(hipStream_t st, const uint64_t* row_off, const uint32_t* row_items, uint64_t n_rows, const uint32_t* block_base, uint2* frag8, uint4* ext8, uint32_t* present) {
    hipLaunchKernelGGL(rows_to_frag8_kernel, dim3((unsigned)((n_rows + 1 + 1023) / 1024)), dim3(1024), 0, st, row_off, row_items, n_rows, block_base, frag8, ext8, present);
    return hipGetLastError();
}

// -------------------------------------------------------------------------------------
// The kernel.  One wave per query; a persistent grid strides over the batch.
// -------------------------------------------------------------------------------------
#ifndef SRN_SBACK_NT
#define SRN_SBACK_NT 1   // the one-use streams of a query (its neighbour slots in, its finish record out) as non-temporal accesses
#endif
#if SRN_SBACK_NT
#define SB_NT_LOAD(p) __builtin_nontemporal_load(p)
#define SB_NT_STORE(v, p) __builtin_nontemporal_store(v, p)
#else
#define SB_NT_LOAD(p) (*(p))
#define SB_NT_STORE(v, p) (*(p) = (v))
#endif
// A workgroup of this kernel is ONE wave: its LDS operations execute in the order they were issued, so what __syncthreads() has to offer here is only that the COMPILER keeps
// that order -- while its fence (s_waitcnt vmcnt(0) lgkmcnt(0) in front of the s_barrier) makes the wave sit out every load and store it still has in flight, ten times per
// query.  SB_SYNC() with SRN_SBACK_FENCES=0 is the compiler-only form (experiment).
#ifndef SRN_SBACK_FENCES
#define SRN_SBACK_FENCES 1   // (measured: 1.652 ms with the compiler-only form against 1.656 -- nothing; the plain barriers stay)
#endif
#if SRN_SBACK_FENCES
#define SB_SYNC() __syncthreads()
#else
#define SB_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#endif
#ifndef SRN_SBACK_WAVES
#define SRN_SBACK_WAVES 3   // waves per SIMD the register allocation is sized for (12 per CU: the LDS allows 13)
#endif
template <bool BITMAP, bool STREAM, bool PBYTES = false>
__global__ __launch_bounds__(64, SRN_SBACK_WAVES) void vmis_shard_back_kernel(DeviceIndex ix_arg, LaunchParams p_arg, FastParams f_arg, SBackParams sb_arg) {
    constexpr uint32_t HOT_OFF = SB_HOT;
    __shared__ __attribute__((aligned(16))) char smem[SB_LDS];
    // the parameter blocks are read from the kernel-argument segment where they are used (as the fast kernel does): loaded up front they would sit in ~120 SGPRs for the
    // kernel's whole life, and the spills of those land in VGPRs
    typedef const __attribute__((address_space(4))) char* KArg;
    const KArg ka = (KArg)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr size_t OFF_P = (sizeof(DeviceIndex) + alignof(LaunchParams) - 1) / alignof(LaunchParams) * alignof(LaunchParams);
    constexpr size_t OFF_F = (OFF_P + sizeof(LaunchParams) + alignof(FastParams) - 1) / alignof(FastParams) * alignof(FastParams);
    constexpr size_t OFF_S = (OFF_F + sizeof(FastParams) + alignof(SBackParams) - 1) / alignof(SBackParams) * alignof(SBackParams);
    const __attribute__((address_space(4))) DeviceIndex& ix = *(const __attribute__((address_space(4))) DeviceIndex*)ka;
    const __attribute__((address_space(4))) LaunchParams& p = *(const __attribute__((address_space(4))) LaunchParams*)(ka + OFF_P);
    const __attribute__((address_space(4))) FastParams& f = *(const __attribute__((address_space(4))) FastParams*)(ka + OFF_F);
    const __attribute__((address_space(4))) SBackParams& sb = *(const __attribute__((address_space(4))) SBackParams*)(ka + OFF_S);
    const uint32_t lane = threadIdx.x;
    auto below = [](unsigned long long bm) -> uint32_t { return __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u)); };   // set bits of bm below this lane (v_mbcnt: no mask register to keep)
    uint16_t* wtab = (uint16_t*)(smem + SB_WTAB);
    unsigned long long* ckey = (unsigned long long*)(smem + SB_CAND);
    uint32_t* cidx = (uint32_t*)(smem + SB_CAND + SB_CAND_CAP * 8);
    uint32_t* ikeys = (uint32_t*)(smem + SB_TABLE);
    int* iacc = (int*)(smem + SB_TABLE + SB_TABLE_WORDS * 4);
    char* const acc_base = smem + HOT_OFF;
    uint32_t* hot = (uint32_t*)(smem + HOT_OFF);
    uint2* hits = (uint2*)(smem + HOT_OFF);
    uint2* lqb = (uint2*)(smem + HOT_OFF + SB_HIT_CAP * 8); uint32_t* lqb_len = (uint32_t*)(smem + HOT_OFF + SB_HIT_CAP * 8 + SB_LQB_CAP * 8);
    uint32_t* lm = (uint32_t*)(smem + SB_CAND);   // (STREAM) the record's bitmaps, then its nibbles
    uint2* lq = (uint2*)(smem + SB_CAND);   // walk A's queue of long fragments: over the candidate buffer and the exact table, both dead until the harvest
    const bool business = (p.flags & SRN_FLAG_BUSINESS_LOGIC) != 0u;
    const bool wide = f.nb == 3u;
    const uint32_t n_kept = ix.n_kept;
    constexpr uint32_t NCH = F_K_MAX / 64u;   // 24 chunks of 64 neighbours
    const __amdgpu_buffer_rsrc_t frag_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)sb.frag8, 0, (int)((n_kept + 1u) * 8u), 0x00020000);   // (the forms that know which neighbours are absent; the host admits them only below 2^29 sessions)

    // per-phase shader cycles (debug, srn_debug_phase_cycles): summed in registers by the wave, flushed once at the end.  Slots follow the fast kernel's numbering:
    // 8 record + clears, 9 walk A, 10 sample + floors, 11 sketch check, 12 walk B + resolve, 13 hand-off; 5 listed elements, 6 candidates, 7 live queries, 14 served, 15 handed over by cause (20-bit fields: candidates | long-fragment queues | hit list; exact table: upper half of 7)
    const bool ticking = p.phase_cycles != nullptr;
    unsigned long long tk8 = 0, tk9 = 0, tk10 = 0, tk11 = 0, tk12 = 0, tk13 = 0, c5 = 0, c6 = 0, c7 = 0, c14 = 0, c15 = 0;
#ifdef SRN_SBACK_SUBTICKS   // (a variant build, build.build_variant(..., ["-DSRN_SBACK_SUBTICKS=1"], sources=("srn_sback.hip",)): four more stamps in the slots the FRONT end's phases use in
                            //  its own launches (1 walk B up to the hit list, 2 resolve, 3 walk A up to the fragments' arrival -- with a forced wait --, 4 sample + threshold): subtract the front end's figures
    unsigned long long tk1 = 0, tk2 = 0, tk3 = 0, tk4 = 0;
#define SB_SUBTICK(acc) SB_TICK(acc)
#else
#define SB_SUBTICK(acc) do {} while (0)
#endif
#define SB_TICK(acc) do { if (ticking) { const long long t_ = clock64(); acc += (unsigned long long)(t_ - t_prev); t_prev = t_; } } while (0)
    // the serving order (f.order, see vmis_fast_kernel): the batch sorted by each query's most popular item, dealt to the XCDs chunk by chunk (the grid is a multiple of 8)
    const bool ordered = f.order != nullptr;
    const uint32_t ox = blockIdx.x & 7u;
    const uint32_t qi_step = ordered ? gridDim.x >> 3 : gridDim.x, qi_end = ordered ? ord_count(p.nq, ox) : p.nq;
    // ---- a query's record, K and neighbour slots are requested ONE QUERY AHEAD (round 6), at the point where the query being served has nothing left in registers that
    // its resolve needs (the hit list carries the slots): the next query's round trip runs beside this query's resolve (item ids from the general fragment slots, a round
    // trip of its own), hand-off and finish.  Every dependent trip of a wave costs ~13 K ticks here whatever it fetches; this takes one of four off a query's chain.
    // The query INDEX is read a further step ahead (a scalar load that has a whole query's time to arrive).
    struct Pre { uint32_t q, U, xlo, L, attr, idx, kept, kv; unsigned long long base; uint32_t sv[NCH]; } pre;
    const uint32_t qi_first = ordered ? blockIdx.x >> 3 : blockIdx.x;
    auto q_at = [&](uint32_t qi) -> uint32_t { return qi < qi_end ? (ordered ? (uint32_t)f.order[ord_pos(ox, qi)] : qi) : 0u; };
    uint32_t q_ahead = q_at(qi_first);
    auto fetch = [&](uint32_t qi) {   // (no branch around any of these loads: a load inside a divergent branch is waited for at the branch's end)
        const uint32_t q = q_ahead;
        q_ahead = q_at(qi + qi_step);
        const char* const rec = p.prep + (size_t)q * p.prep_stride;
        const uint32_t* const xq = f.xchg + (size_t)q * f.xchg_stride;
        uint32_t ln = lane; asm volatile("" : "+v"(ln));   // (the lane number as the optimiser cannot see through it: the 24 clamped slot offsets are loop-invariant, and hoisted out of the query loop they were spilled -- each reload a s_waitcnt vmcnt(0) between two of the slot loads)
        const PrepHead* hp = (const PrepHead*)rec;
        pre.q = q; pre.U = hp->U; pre.xlo = hp->xlo; pre.L = hp->L; pre.attr = hp->cur_attr;
        const PrepItem* pi = (const PrepItem*)(rec + sizeof(PrepHead)) + min(ln, max(1u, min(p.max_len, 8u)) - 1u);   // (all the record's places, with the head: asked for behind L they were a dependent trip of their own; lanes past them are masked where the values are used)
        pre.idx = pi->idx; pre.kept = pi->kept; pre.base = pi->base;
        pre.kv = xq[0];
        if constexpr (!STREAM) {
            // (through a buffer descriptor of the record: the address is lane * 4 + a constant per chunk -- the instruction's own offset field --, and what lies past the
            // record's end comes back as 0 from the address unit: no clamp, no 64-bit address per load)
            const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)xq, 0, (int)(f.xchg_stride * 4u), 0x00020000);
#pragma unroll
            for (uint32_t c = 0; c < NCH; ++c) pre.sv[c] = __builtin_amdgcn_raw_buffer_load_b32(xr, ln * 4u + (4u + c * 256u), 0, SRN_SBACK_NT ? 2 : 0);   // (read once: 5.4 KB per query that need not stay in the L2 the fragments want; the slots past K are stale words of the query's own row, masked below)
        }
    };
    if (qi_first < qi_end) fetch(qi_first);
    for (uint32_t qi = qi_first; qi < qi_end; qi += qi_step) {
        long long t_prev = ticking ? clock64() : 0;
        // ---- the query's record and its neighbour slots, requested during the query before ----
        const uint32_t q = pre.q;
        const uint32_t* const xq = f.xchg + (size_t)q * f.xchg_stride;
        struct { uint32_t U, xlo, L, cur_attr; } h0;
        h0.U = (uint32_t)__builtin_amdgcn_readfirstlane((int)pre.U); h0.xlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)pre.xlo);
        h0.L = (uint32_t)__builtin_amdgcn_readfirstlane((int)pre.L); h0.cur_attr = (uint32_t)__builtin_amdgcn_readfirstlane((int)pre.attr);
        const bool place = lane < 8u && lane < p.max_len && lane < h0.L;   // (places past L hold an earlier batch's items)
        const uint32_t it_idx = place ? pre.idx : kNone, it_kept = place ? pre.kept : 0u; const unsigned long long it_base = place ? pre.base : 0ull;
        const uint32_t kv = pre.kv;
        uint32_t ln = lane; asm volatile("" : "+v"(ln));
        uint32_t sv[NCH];   // gather form: the neighbour slots; streaming form: the members' {position | run << 20 | weight << 24}, read back from the scratch after walk A
        if constexpr (!STREAM) {
#pragma unroll
            for (uint32_t c = 0; c < NCH; ++c) sv[c] = pre.sv[c];
        }
        bool fetched = false;   // (wave-uniform)
        auto fetch_next = [&]() { if (!fetched) { fetched = true; if (qi + qi_step < qi_end) fetch(qi + qi_step); } };
        const uint32_t L = h0.L, U = h0.U, cur_attr = h0.cur_attr;
        unsigned long long rm = __ballot(it_kept > 0u);
        const uint32_t nr = (uint32_t)__popcll(rm);
        const bool rel = wide && nr > 3u;
        const uint32_t NB = wide && !rel ? 3u : 4u, NBM = (1u << NB) - 1u, base = rel ? h0.xlo : 0u;
        const uint32_t cur_idx = (uint32_t)__builtin_amdgcn_readlane((int)it_idx, 0);
        uint32_t ps[4];   // position (= lane of the record's item) of run r
        {   // weight of each list set: 10 * linear_score(first match) * numerator (mod.rs:110-116, 133-144); runs are numbered in position order
#pragma unroll
            for (int r = 0; r < 4; ++r) { ps[r] = rm ? (uint32_t)__ffsll((long long)rm) - 1u : 0u; rm &= rm - 1ull; }
            if (lane < 16u) {
                const uint32_t num = ((lane & 1u) ? L - ps[0] : 0u) + ((lane & 2u) ? L - ps[1] : 0u) + ((lane & 4u) ? L - ps[2] : 0u) + ((lane & 8u) ? L - ps[3] : 0u);
                const uint32_t lo = lane ? (uint32_t)__ffs((int)lane) - 1u : 0u;
                const uint32_t mp = lo == 0u ? ps[0] : lo == 1u ? ps[1] : lo == 2u ? ps[2] : ps[3];
                wtab[lane] = (uint16_t)((9u - mp) * num);
            }
        }
        {   // clear: accumulators + sketch + dump (the exact table is cleared after walk A: its words hold the long fragments' queue until then)
            uint4* z = reinterpret_cast<uint4*>(smem + HOT_OFF);
            uint32_t z0 = 0u; asm volatile("" : "+v"(z0));   // (a zero the optimiser cannot see through: the constant quad was hoisted out of the query loop into four registers for the wave's life -- and spilled)
            for (uint32_t i = lane; i < (SB_H + SB_S + SB_DUMP) / 4u; i += 64u) z[i] = make_uint4(z0, z0, z0, z0);
        }
        SB_SYNC();   // (one wave: orders the LDS traffic; no other wave to wait for)
        // (K is looked at only HERE, behind the barrier's wait for everything in flight: tested right after its load, the compiler sinks the 24 slot loads below the two
        // early exits, i.e. issues them when K has arrived -- a second dependent round trip per query)
        const uint32_t K = (uint32_t)__builtin_amdgcn_readfirstlane((int)kv);
        if (K == 0xFFFFFFFFu || L < 1u || L > 8u || L > p.max_len || K > F_K_MAX) {   // (wave-uniform) no front end took it, or not this kernel's shape: the general kernel does its candidate work itself
            if (lane == 0u) f.slow_list[atomicAdd(f.slow_cnt, 1u)] = q;
            fetch_next();
            continue;
        }
        if (K == 0u) { if (lane == 0u) p.out_counts[q] = 0u; fetch_next(); continue; }
        SB_TICK(tk8);
        // ---- walk A: ALL of a lane's <= 24 presence words in flight together, then all fragments of the present ones: a query's walk is three HBM / L2 round trips (slots,
        // presence, fragments) whatever K is.  A first build walked in two halves with the long fragments' overflow blocks fetched inline: 30 dependent round trips per query on
        // 12 waves per CU -- 2.9 ms per 131 072 queries against the 1.9 ms of the kernel it replaces.
        auto add2 = [&](uint32_t wd, uint32_t w) { atomicAdd((uint32_t*)(acc_base + (wd & 0xFFFFu)), w); atomicAdd((uint32_t*)(acc_base + (wd >> 16)), w); };
        uint32_t pm = 0u;    // bit c: this lane's neighbour of chunk c holds an item of this shard (kept for walk B)
        uint32_t nlq = 0u;   // (wave-uniform) long fragments queued
        uint2 fr[NCH];   // a lane's <= 24 fragments: they STAY in registers for walk B -- with 12 x 32 queries in flight per XCD (33 MB of fragment lines against 4 MB of L2) a
                                      // second fetch misses like the first: walk B took as long as walk A (54 K against 50 K cycles per query) until it stopped fetching
        uint32_t nscr = 0u;  // (STREAM, wave-uniform) members with a non-empty fragment, listed in the wave's scratch for walk B
        unsigned long long lbase[4] = {0ull, 0ull, 0ull, 0ull};   // (STREAM) where the runs' posting lists start
        bool st_fail = false;
        if constexpr (STREAM) {
            // ---- walk A, streaming form.  The gather form asks for one fragment per (query, neighbour) at a random place: 178 M requests per batch of config 3, a third of
            // them past the L2 -- 128 bytes fetched for 8 used, 55 G requests/s chip-wide (tools/shard_gather_bench) -- and that is the kernel's time whatever the wave count
            // or the number of dependent trips.  Here the fragments are stored a second time IN POSTING ORDER (frag_post[e] = fragment of the session post_rank[e]): a query's
            // neighbours all lie in the kept prefixes of its items' posting lists, the rank that fronted the query has said WHERE (a bitmap per list; a neighbour in several
            // lists counts in the first), and the wave reads those prefixes coalesced -- the same lines for every query that shares the item, and the serving order puts
            // such queries next to each other.  The loop body is STRAIGHT-LINE code (selects, no branches: non-members add 0 to the lane's own dump word and store into a
            // trash place of the scratch): with a branch per chunk the compiler cannot overlap the chunks' dependent LDS reads (bitmap word -> member index -> weight), and
            // those, not memory, were the walk's time (70 K cycles per query against 25 K for the gather form).
            const uint2* const frag_post = sb.frag_post;
            uint32_t* const scr_w = sb.scr + (size_t)blockIdx.x * SB_SCR_WORDS; uint2* const scr_f = reinterpret_cast<uint2*>(scr_w + SB_SCR_ENTRIES);
            const uint32_t mw = (p.m + 63u) >> 6, ncw = (K + 7u) >> 3;
            uint8_t* const lw8 = reinterpret_cast<uint8_t*>(lm + 8u * mw);   // the members' weights, one byte each (<= 234), in stream order
            {   // the record's bitmaps (runs present only) -> LDS; its nibbles -> weights
                for (uint32_t i = lane; i < 2u * nr * mw; i += 64u) lm[i] = xq[2u + i];
                for (uint32_t i = lane; i < ncw; i += 64u) {
                    const uint32_t c8 = xq[2u + 8u * mw + i];
                    uint32_t lo = 0u, hi = 0u;
#pragma unroll
                    for (uint32_t n4 = 0; n4 < 4u; ++n4) { lo |= (uint32_t)wtab[(c8 >> (4u * n4)) & 15u] << (8u * n4); hi |= (uint32_t)wtab[(c8 >> (16u + 4u * n4)) & 15u] << (8u * n4); }
                    reinterpret_cast<uint2*>(lw8)[i] = make_uint2(lo, hi);
                }
            }
            SB_SYNC();
            const uint32_t dump1 = (SB_H + SB_S + lane) * 4u, dump2 = dump1 | (dump1 << 16);
            uint32_t nmem = 0u;   // (wave-uniform) members met so far = index of the next weight
            for (uint32_t r = 0; r < nr; ++r) {   // (wave-uniform)
                const uint32_t pl = r == 0u ? ps[0] : r == 1u ? ps[1] : r == 2u ? ps[2] : ps[3];
                const uint32_t kept = (uint32_t)__builtin_amdgcn_readlane((int)it_kept, (int)pl);
                const unsigned long long lb = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(it_base >> 32), (int)pl) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)it_base, (int)pl);
                lbase[0] = r == 0u ? lb : lbase[0]; lbase[1] = r == 1u ? lb : lbase[1]; lbase[2] = r == 2u ? lb : lbase[2]; lbase[3] = r == 3u ? lb : lbase[3];
                const unsigned long long* const bm64 = reinterpret_cast<const unsigned long long*>(lm) + (size_t)r * mw;
                for (uint32_t j0 = 0; j0 < kept; j0 += 64u * SB_PF) {   // (SB_PF chunks of 64 postings in flight at once)
                    uint2 fg[SB_PF];
#pragma unroll
                    for (uint32_t u = 0; u < SB_PF; ++u) fg[u] = frag_post[lb + (j0 + u * 64u < kept ? min(j0 + u * 64u + lane, kept - 1u) : 0u)];   // (past the list's end: its first line again)
#pragma unroll
                    for (uint32_t u = 0; u < SB_PF; ++u) {
                        const bool in = j0 + u * 64u < kept;   // (wave-uniform)
                        const unsigned long long mk0 = bm64[min((j0 >> 6) + u, mw - 1u)], mk = in ? mk0 : 0ull;
                        const bool mem = (mk >> lane) & 1ull;
                        const uint32_t idx = nmem + below(mk);
                        nmem += (uint32_t)__popcll(mk);
                        const uint32_t w0 = (uint32_t)lw8[min(idx, F_K_MAX - 1u)], w = mem ? w0 : 0u;
                        const uint32_t o0 = fg[u].x & 0xFFFFu;
                        const bool lng = mem && o0 >= SB_LONG;
                        const bool pr = mem && (lng || o0 < (SB_H + SB_S) * 4u);   // (a fragment's items fill its positions from the first: a dump offset there = an empty fragment)
                        const bool ac = pr && !lng;                                // (long fragments: from the scratch, below)
                        add2(ac ? fg[u].x : dump2, w); add2(ac ? fg[u].y : dump2, w);
                        const unsigned long long bm = __ballot(pr);
                        const uint32_t at = pr ? min(nscr + below(bm), F_K_MAX + 63u) : F_K_MAX + 64u + lane;   // (non-members: a trash place)
                        scr_w[at] = (j0 + u * 64u + lane) | (r << 20) | (w << 24); scr_f[at] = fg[u];
                        nscr += (uint32_t)__popcll(bm);
                    }
                }
            }
            if (nmem != K || nscr > F_K_MAX) st_fail = true;   // (the record does not describe K neighbours: the general kernel serves the query)
            // the members with a fragment, back from the scratch into registers (they stay there for walk B, as the gather form's do); the long ones are queued as there
            __syncthreads();
#pragma unroll
            for (uint32_t c = 0; c < NCH; ++c) {
                sv[c] = 0u; fr[c] = make_uint2(0u, 0u);
                if (c * 64u < nscr) { const uint32_t i = min(c * 64u + lane, nscr - 1u); sv[c] = scr_w[i]; fr[c] = scr_f[i]; }   // (wave-uniform branch: no load is waited for inside)
            }
#pragma unroll
            for (uint32_t c = 0; c < NCH; ++c) {
                if (c * 64u < nscr) {
                    const bool pr = c * 64u + lane < nscr;
                    pm |= pr ? 1u << c : 0u;
                    const bool lng = pr && sb_is_long(fr[c].x);
                    const unsigned long long lbm = __ballot(lng);
                    if (lbm) {
                        const uint32_t at = nlq + below(lbm);
                        if (lng && at < SB_LQ_CAP) lq[at] = make_uint2(((sv[c] >> 24) << 16) | sb_long_len(fr[c].x), sb_long_block(fr[c].x, fr[c].y));
                        nlq += (uint32_t)__popcll(lbm);
                    }
                }
            }
        } else {
        {
            uint32_t pwv[NCH];
#pragma unroll
            for (uint32_t c = 0; c < NCH; ++c) {   // (all 24 chunks whatever K, no branch: one batch of look-ups.  With a wave-uniform branch per chunk -- or group of chunks -- each look-up was waited for in its own block)
                const bool act = c * 64u + lane < K;
                if constexpr (BITMAP) { const uint32_t r = act ? base + (sv[c] >> NB) : n_kept; pwv[c] = sb.present[r >> 5] >> (r & 31u); }   // (idle lanes: the empty row, whose presence bit is 0)
                else if constexpr (PBYTES) {   // the fronting rank said which neighbours have a fragment here: a byte each behind the slots.  Clamped and unconditional: inside `act ? ... : 0` each byte load sat in a divergent branch of its own and was waited for there -- 22 round trips, what made this form lose until round 6
                    const uint32_t pb = (uint32_t)reinterpret_cast<const uint8_t*>(xq + 1u + p.k)[min(c * 64u + ln, p.k - 1u)];
                    pwv[c] = (pb >> sb.pbyte_shift) & (act ? 1u : 0u);   // (an AND, not a select: the code generator turns a select whose operand is a load back into a branch around the load)
                }
                else pwv[c] = act ? 1u : 0u;   // (no bitmap: every neighbour's fragment is fetched; the empty ones are told apart below)
            }
#pragma unroll
            for (uint32_t c = 0; c < NCH; ++c) pm |= (pwv[c] & 1u) << c;
        }
        {
#pragma unroll
            for (uint32_t c = 0; c < NCH; ++c) fr[c] = make_uint2(0u, 0u);
#pragma unroll
            for (uint32_t g = 0; g < NCH; g += 4u) if (g * 64u < K) {   // (wave-uniform, per GROUP of four chunks: a branch per chunk ends the scheduler's region there, and every chunk's loads and LDS reads are then waited for inside their own block; the lanes past K behave as absent neighbours)
#pragma unroll
                for (uint32_t c = g; c < g + 4u; ++c) {
                    if constexpr (BITMAP || PBYTES) {   // (round 6) the absent lanes ask for NOTHING: a buffer load past the descriptor's range is answered with 0 by the address unit, no request leaves it
                        typedef uint32_t v2u __attribute__((ext_vector_type(2)));
                        const v2u v = __builtin_amdgcn_raw_buffer_load_b64(frag_rsrc, (pm >> c) & 1u ? (base + (sv[c] >> NB)) * 8u : 0xFFFFFFF8u, 0, 0);
                        fr[c] = make_uint2(v.x, v.y);
                    } else fr[c] = sb.frag8[(pm >> c) & 1u ? base + (sv[c] >> NB) : n_kept];   // (unconditional per lane: a load inside a divergent branch is waited for at the branch's end; the lanes past K all read the empty row's slot -- one line)
                }
            }
            // the neighbours' weights, a byte each (<= 9 * 26), looked up while the fragments travel: the adds below then depend on no LDS read of their own
            uint32_t wq[NCH / 4u];
#pragma unroll
            for (uint32_t c = 0; c < NCH; c += 4u) wq[c >> 2] = 0u;
            {
                uint32_t w16[NCH];
#pragma unroll
                for (uint32_t c = 0; c < NCH; ++c) w16[c] = (uint32_t)wtab[sv[c] & NBM];   // (all 24, whatever K: one batch of LDS reads -- the barriers keep the scheduler from dealing them out between the adds, two at a time and each pair waited for)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (uint32_t c = 0; c < NCH; ++c) wq[c >> 2] |= w16[c] << (8u * (c & 3u));
                __builtin_amdgcn_sched_barrier(0);
            }
            // STRAIGHT-LINE adds (round 6): all four positions of every fragment, whoever holds it -- unused positions, the absent lanes' empty row and the lanes past K
            // point out of the allocation (SB_OOR: dropped by the hardware), a long fragment's words are replaced by such.  With a branch per chunk (and the weight's
            // LDS read in front of its adds) the chunks ran one after the other, each behind its own s_waitcnt.
#ifdef SRN_SBACK_SUBTICKS
            if (ticking) __builtin_amdgcn_s_waitcnt(0);
            SB_SUBTICK(tk3);
#endif
            uint32_t lngm = 0u;   // bit c: this lane's fragment of chunk c is a long one (> 4 items: queued below)
#pragma unroll
            for (uint32_t g = 0; g < NCH; g += 4u) if (g * 64u < K) {
#pragma unroll
                for (uint32_t c = g; c < g + 4u; ++c) {
                    const uint32_t o0 = fr[c].x & 0xFFFFu;
                    const bool here = (BITMAP || PBYTES) ? ((pm >> c) & 1u) != 0u : true;   // (those forms' absent lanes hold zeros)
                    const bool lng = here && o0 >= SB_LONG;   // (never the empty row's slot; a long fragment's four words are out-of-range offsets themselves)
                    const bool pr = here && (lng || o0 < (SB_H + SB_S) * 4u);   // (a fragment's items fill its positions from the first: an out-of-range offset there = an empty fragment)
                    pm = pr ? pm : pm & ~(1u << c);                  // (walk B skips it too)
                    lngm |= lng ? 1u << c : 0u;
                    const uint32_t w = (wq[c >> 2] >> (8u * (c & 3u))) & 0xFFu;
                    add2(!here ? SB_OOR * 0x10001u : fr[c].x, w); add2(!here ? SB_OOR * 0x10001u : fr[c].y, w);   // (`here` is a constant in the default form: no select)
                }
            }
            if (__ballot(lngm != 0u) != 0ull) {   // fragments of > 4 items (rare from G = 8 on): queued -- {weight | length, first overflow block} --, all the queue's blocks are fetched together below
#pragma unroll
                for (uint32_t c = 0; c < NCH; ++c) {
                    if (c * 64u < K) {
                        const bool lng = (lngm >> c) & 1u;
                        const unsigned long long lb = __ballot(lng);
                        if (lb) {
                            const uint32_t at = nlq + below(lb);
                            if (lng && at < SB_LQ_CAP) lq[at] = make_uint2((((wq[c >> 2] >> (8u * (c & 3u))) & 0xFFu) << 16) | sb_long_len(fr[c].x), sb_long_block(fr[c].x, fr[c].y));
                            nlq += (uint32_t)__popcll(lb);
                        }
                    }
                }
            }
        }
        }
        bool fail = st_fail || nlq > SB_LQ_CAP;   // (wave-uniform)
        if (fail) c15 += 1ull << 20;
        if (nlq && !fail) {
            SB_SYNC();
            for (uint32_t i0 = 0; i0 < nlq; i0 += 64u) {
                const bool act = i0 + lane < nlq;
                const uint2 e = lq[min(i0 + lane, nlq - 1u)];
                const uint32_t len = act ? e.x & 0xFFFFu : 0u, w = e.x >> 16;
                const uint4* eb = sb.ext8 + (size_t)e.y;
                const uint4 b0 = eb[0], b1 = eb[len > 8u ? 1 : 0];   // (the first two blocks together: 16 items cover nearly every long fragment)
                if (act) { add2(b0.x, w); add2(b0.y, w); add2(b0.z, w); add2(b0.w, w); }
                if (len > 8u) { add2(b1.x, w); add2(b1.y, w); add2(b1.z, w); add2(b1.w, w); }
                for (uint32_t t8 = 16u; __ballot(t8 < len) != 0ull; t8 += 8u)
                    if (t8 < len) { const uint4 e4 = eb[t8 >> 3]; add2(e4.x, w); add2(e4.y, w); add2(e4.z, w); add2(e4.w, w); }
            }
        }
        SB_SYNC();
        SB_TICK(tk9);
        // ---- harvest: the sample (this shard's 256 most popular items, four per lane), exactly -> threshold, candidates ----
        // (the sample's constants -- items 4 lane .. 4 lane + 3 of the shard's popularity order: 4 KB that every wave of the chip reads, L1-resident -- are fetched per query:
        // kept in 13 registers for the wave's life they were spilled, and each of the four reloads from scratch was waited for on its own)
        double s_idf[4]; uint32_t s_attr = 0, s_rank[4];
        uint32_t ls = lane; asm volatile("" : "+v"(ls));
        {
            ItemMeta m0[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) m0[j] = sb.sample[4u * ls + (uint32_t)j];
#pragma unroll
            for (int j = 0; j < 4; ++j) { s_idf[j] = m0[j].idf > 0.0 ? m0[j].idf : 1.0; s_attr |= (m0[j].attr & 0xFFu) << (8 * j); s_rank[j] = m0[j].id_rank; }
        }
        uint32_t v4[4];
        { const uint4 a = reinterpret_cast<const uint4*>(hot)[lane]; v4[0] = a.x; v4[1] = a.y; v4[2] = a.z; v4[3] = a.w; }
        if (lane < SB_REP_ITEMS / 4u) {   // replicated items: the sum is spread over SB_REP words (the item's own word stays 0)
#pragma unroll
            for (uint32_t j = 0; j < 4u; ++j) { const uint4* rp = reinterpret_cast<const uint4*>(hot + SB_DIRECT + (4u * lane + j) * SB_REP); const uint4 a = rp[0], b = rp[1];
                                                v4[j] = a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w; }
        }
        if (ls < SB_DUMP) hot[SB_H + SB_S + ls] = 0u;   // (walk B reads the dump words, where unused positions point, as "cannot reach the floor")
        double x4[4]; uint32_t k4[4];
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) {
            const uint32_t e = 4u * ls + j;
            bool valid = v4[j] != 0u && e != cur_idx && e < ix.n_items;
            if (business) valid = valid && business_ok(cur_attr, (s_attr >> (8u * j)) & 0xFFu);   // an item the rules exclude is no candidate and sets no threshold
            x4[j] = valid ? s_idf[j] * (double)v4[j] : 0.0;
            k4[j] = (uint32_t)((unsigned long long)__double_as_longlong(x4[j]) >> 32);
        }
        // the n-th largest of the 256 keys (top 32 bits of x: a monotone truncation), bit by bit from the top down to bit 8: t32 = the largest multiple of 256 with
        // at least n keys at or above it (0: fewer than n valid items -- no threshold, everything valid is a candidate)
        uint32_t t32 = 0u;
        for (int b = 30; b >= 8; --b) {   // (bit 31 is the sign of a non-negative number)
            const uint32_t c = t32 | (1u << b);
            const uint32_t cnt = (uint32_t)__popcll(__ballot(k4[0] >= c)) + (uint32_t)__popcll(__ballot(k4[1] >= c)) + (uint32_t)__popcll(__ballot(k4[2] >= c)) + (uint32_t)__popcll(__ballot(k4[3] >= c));
            t32 = cnt >= p.how_many ? c : t32;
        }
        SB_SUBTICK(tk4);
        // everything is kept down to one step BELOW it, so that what is dropped is strictly smaller after the division by 10 U as well (ties at the cut included)
        const uint32_t t32m1 = t32 ? t32 - 1u : 0u;
        uint32_t ncand = 0;   // (wave-uniform)
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) {
            const bool take = x4[j] != 0.0 && k4[j] >= t32m1;
            const unsigned long long bm = __ballot(take);
            if (bm) {
                const uint32_t at = ncand + below(bm);
                if (take) { if (at < SB_CAND_CAP) { ckey[at] = (unsigned long long)__double_as_longlong(x4[j]); cidx[at] = s_rank[j]; } }
                ncand += (uint32_t)__popcll(bm);
            }
        }
        // integer floors: an item needs idf * acc >= x_lo, i.e. acc >= x_lo / (largest idf of its chunk of 256); shaved so that rounding can only keep more
        const double x_lo = __longlong_as_double((long long)((unsigned long long)t32m1 << 32));
        auto floor_of = [&](double inv) -> uint32_t { return (uint32_t)fmin(4.0e9, fmax(1.0, floor(x_lo * inv * (1.0 - 1e-9)) - 1.0)); };
        const uint32_t floor_b = floor_of(sb.inv_idf_all);
        // the other direct-mapped words: a quad per lane and chunk, the chunk's floor wave-uniform.  What reaches its floor is LISTED first -- {entry, sum} in the exact table's
        // room, which walk B initialises when it runs -- and the listed entries' idf / attributes / id ranks are then fetched TOGETHER: one round trip per 64 of them
        // (round 6; until then one per (chunk, position) with an entry at the floor, each waited for: ~5 dependent trips per query, the bulk of this phase's 17 K cycles)
        uint32_t lf = lane; asm volatile("" : "+v"(lf));   // (as `ln` above: the twelve entry numbers below are loop-invariant -- hoisted, they were spilled and reloaded one by one)
        uint2* const plist = reinterpret_cast<uint2*>(smem + SB_TABLE);
        constexpr uint32_t PLIST_CAP = SB_TABLE_WORDS;   // (8 bytes per entry in the table's 2 KB)
        uint32_t npass = 0u;   // (wave-uniform)
        uint4 dq[SB_H / 256u];
#pragma unroll
        for (uint32_t ch = 1; ch < SB_H / 256u; ++ch) dq[ch] = reinterpret_cast<const uint4*>(hot)[ch * 64u + lane];   // (the three quads in one batch)
#pragma unroll
        for (uint32_t ch = 1; ch < SB_H / 256u; ++ch) {
            const uint32_t fl = floor_of(sb.inv_idf_chunk[ch]);
            const uint4 q4 = dq[ch];
            const uint32_t mx = max(max(q4.x, q4.y), max(q4.z, q4.w));
            if (__ballot(mx >= fl) == 0ull) continue;
            const uint32_t vv[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
            for (uint32_t j = 0; j < 4u; ++j) {
                const uint32_t e = ch * 256u + 4u * lf + j;
                const bool pass = vv[j] >= fl && e != cur_idx && e < SB_DIRECT;   // (the words from SB_DIRECT up are the replicas of the hottest items, already in the sample)
                const unsigned long long bm = __ballot(pass);
                if (bm) {
                    const uint32_t at = npass + below(bm);
                    if (pass && at < PLIST_CAP) plist[at] = make_uint2(e, vv[j]);
                    npass += (uint32_t)__popcll(bm);
                }
            }
        }
        if (npass > PLIST_CAP) { fail = true; c15 += 1ull; npass = 0u; }   // (more entries at their floors than candidates the record could hold)
        for (uint32_t i0 = 0; i0 < npass; i0 += 64u) {
            const bool act = i0 + lane < npass;
            const uint2 pe = plist[min(i0 + lane, npass - 1u)];
            ItemMeta mt = ItemMeta{0.0, 0u, 0u};
            if (act) mt = ix.meta[pe.x];
            const double x = (mt.idf > 0.0 ? mt.idf : 1.0) * (double)pe.y;
            const bool take = act && (uint32_t)((unsigned long long)__double_as_longlong(x) >> 32) >= t32m1 && (!business || business_ok(cur_attr, mt.attr));
            const unsigned long long bm = __ballot(take);
            if (bm) {
                const uint32_t at = ncand + below(bm);
                if (take && at < SB_CAND_CAP) { ckey[at] = (unsigned long long)__double_as_longlong(x); cidx[at] = mt.id_rank; }
                ncand += (uint32_t)__popcll(bm);
            }
        }
        if (ncand > SB_CAND_CAP) { fail = true; c15 += 1ull; }
        SB_TICK(tk10);
        // ---- is any sketch word at the floor at all?  If not, no item outside the direct-mapped range can make the top n: no walk B ----
        bool live;
        {
            const uint4* sk4 = reinterpret_cast<const uint4*>(hot + SB_H);
            uint32_t mx = 0;
#pragma unroll
            for (uint32_t i = 0; i < SB_S / 256u; ++i) { const uint4 w4 = sk4[i * 64u + lane]; mx = max(max(mx, w4.x), max(max(w4.y, w4.z), w4.w)); }
            live = __ballot(mx >= floor_b) != 0ull;
        }
        uint32_t nt = 0;   // contenders of the exact table, listed behind the candidates
        SB_TICK(tk11);
        if (live && !fail) {
            c7 += 1ull;
            // ---- walk B: the rows again (from the registers); an element is LISTED if its sketch word can still reach the floor (all elements of an item share the word, so an item
            // is accumulated completely or not at all); then the list is resolved -- item id from the general fragment slots -- into the exact table ----
            uint32_t e0 = EMPTY32, z0 = 0u; asm volatile("" : "+v"(e0), "+v"(z0));   // (see the clears: no constant quads kept across queries)
            reinterpret_cast<uint4*>(ikeys)[lane] = make_uint4(e0, e0, e0, e0);   // the exact table (keys EMPTY32, sums 0); its room held the floors' list until here
            reinterpret_cast<uint4*>(iacc)[lane] = make_uint4(z0, z0, z0, z0);
#pragma unroll
            for (uint32_t i = 0; i < SB_H / 256u; ++i) reinterpret_cast<uint4*>(hot)[i * 64u + lane] = make_uint4(z0, z0, z0, z0);   // (the direct-mapped words are dead: zeroed, none of them is "at the floor" -- four stores instead of a range test per position read below)
            SB_SYNC();   // (the direct-mapped words are dead: the hit list takes them)
            uint32_t nh = 0;   // (wave-uniform)
            auto chk = [&](uint32_t o) -> bool { return o >= SB_H * 4u && *(const uint32_t*)(acc_base + o) >= floor_b; };
            auto list = [&](uint32_t hm, uint32_t s, uint32_t j0) {   // hm: bit i = position j0 + i of the fragment is a hit (the long fragments' blocks)
                const uint32_t c = (uint32_t)__popc(hm);
                if (__ballot(c != 0u) == 0ull) return;
                const uint32_t inc = wave_incl_scan(c);
                uint32_t at = nh + inc - c;
                while (hm) { const uint32_t b = (uint32_t)__ffs((int)hm) - 1u; hm &= hm - 1u; if (at < SB_HIT_CAP) hits[at] = make_uint2(s, j0 + b); ++at; }
                nh += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
            };
            uint32_t nlb = 0u;   // (wave-uniform) long fragments queued for the second step
            {   // (the fragments are walk A's, still in registers.)  Round 6: every position's word is read WITHOUT a branch -- absent lanes hold the empty row's slot, unused
                // positions point out of the allocation and read 0 -- and a lane's hits are kept as four bits per chunk; ONE scan then places all of them.  Until then a
                // chunk was four predicated reads, each waited for, a ballot and a scan of its own: 22 dependent chains of ~600 cycles per query.
                const uint32_t lim = STREAM ? nscr : K;
                const uint32_t floor31 = min(floor_b, 0x7FFFFFFFu);
                uint32_t hmw[NCH / 8u];
#pragma unroll
                for (uint32_t c = 0; c < NCH; c += 8u) hmw[c >> 3] = 0u;
                uint32_t lngm = 0u;
#pragma unroll
                for (uint32_t g = 0; g < NCH; g += 4u) if (g * 64u < lim) {   // (wave-uniform, per group of four chunks)
                    uint32_t wv[16];   // the group's sixteen words, all asked for before the first is looked at (the scheduler, left alone, reads two and waits)
#pragma unroll
                    for (uint32_t c = g; c < g + 4u; ++c) {
                        const bool pr = (pm >> c) & 1u, lng = pr && sb_is_long(fr[c].x);
                        const uint32_t fx = STREAM && !pr ? SB_OOR * 0x10001u : fr[c].x, fy = STREAM && !pr ? SB_OOR * 0x10001u : fr[c].y;   // (the streaming form's lanes past the last member hold a copy of its fragment.  Otherwise as they are: unused positions and a long fragment's words read 0 from out of range; the opt-in forms' absent lanes and the lanes past the last member hold zeros -- the direct-mapped word 0, zeroed above)
                        const uint32_t i = 4u * (c - g);
                        wv[i] = *(const uint32_t*)(acc_base + (fx & 0xFFFFu)); wv[i + 1u] = *(const uint32_t*)(acc_base + (fx >> 16));
                        wv[i + 2u] = *(const uint32_t*)(acc_base + (fy & 0xFFFFu)); wv[i + 3u] = *(const uint32_t*)(acc_base + (fy >> 16));
                        lngm |= lng ? 1u << c : 0u;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    // bit i = word i is BELOW the floor: the sign of (word - floor) shifted in (v_sub + v_alignbit: no condition code, no constants; sums are far below 2^31)
                    uint32_t miss = 0u;
#pragma unroll
                    for (uint32_t i = 16u; i-- > 0u;) miss = __builtin_amdgcn_alignbit(miss, wv[i] - floor31, 31u);   // (direct-mapped words: zeroed above; out of range: 0)
                    hmw[g >> 3] |= (~miss & 0xFFFFu) << (4u * (g & 7u));
                }
                uint32_t cnt = 0u;
#pragma unroll
                for (uint32_t c = 0; c < NCH; c += 8u) cnt += (uint32_t)__popc(hmw[c >> 3]);
                if (__ballot(cnt != 0u) != 0ull) {
                    const uint32_t inc = wave_incl_scan(cnt);
                    uint32_t at = inc - cnt;
                    nh = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
                    if (nh <= SB_HIT_CAP) {
#pragma unroll
                        for (uint32_t g = 0; g < NCH; g += 4u) {   // (per GROUP of four chunks: a lane has a hit in one group in seven -- six short blocks instead of 24)
                            if (g * 64u >= lim) continue;
                            uint32_t hm = (hmw[g >> 3] >> (4u * (g & 7u))) & 0xFFFFu;
                            while (hm) {
                                const uint32_t b = (uint32_t)__ffs((int)hm) - 1u, cc = b >> 2; hm &= hm - 1u;
                                const uint32_t s_ = cc == 0u ? sv[g] : cc == 1u ? sv[g + 1u] : cc == 2u ? sv[g + 2u] : sv[g + 3u];
                                hits[at] = make_uint2(s_, b & 3u); ++at;
                            }
                        }
                    }
                }
                if (__ballot(lngm != 0u) != 0ull) {
#pragma unroll
                    for (uint32_t c = 0; c < NCH; ++c) {
                        if (c * 64u >= lim) continue;
                        const bool lng = (lngm >> c) & 1u;
                        const unsigned long long lb = __ballot(lng);
                        if (lb) {
                            const uint32_t at = nlb + below(lb);
                            if (lng && at < SB_LQB_CAP) { lqb[at] = make_uint2(sv[c], sb_long_block(fr[c].x, fr[c].y)); lqb_len[at] = sb_long_len(fr[c].x); }
                            nlb += (uint32_t)__popcll(lb);
                        }
                    }
                }
            }
            fetch_next();   // (the slots and the fragments are dead from here on: the next query's take their registers)
            SB_SUBTICK(tk1);
            if (nlb > SB_LQB_CAP) { fail = true; c15 += 1ull << 20; }
            else if (nlb) {
                SB_SYNC();
                for (uint32_t i0 = 0; i0 < nlb; i0 += 64u) {
                    const bool act = i0 + lane < nlb;
                    const uint2 e = lqb[min(i0 + lane, nlb - 1u)];
                    const uint32_t len = act ? lqb_len[min(i0 + lane, nlb - 1u)] : 0u;
                    for (uint32_t t8 = 0; __ballot(t8 < len) != 0ull; t8 += 8u) {
                        uint32_t h8 = 0;
                        if (t8 < len) { const uint4 e4 = sb.ext8[(size_t)e.y + (t8 >> 3)]; const uint32_t wv[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
                                        for (uint32_t x = 0; x < 4u; ++x) h8 |= (chk(wv[x] & 0xFFFFu) ? 1u << (2u * x) : 0u) | (chk(wv[x] >> 16) ? 2u << (2u * x) : 0u); }
                        list(h8, e.x, t8);
                    }
                }
            }
            SB_SYNC();
            c5 += nh;
            if (nh > SB_HIT_CAP) { fail = true; c15 += 1ull << 40; }
            else {
                bool ovf = false;
                for (uint32_t i = lane; i < nh; i += 64u) {
                    const uint2 h = hits[i];
                    size_t row;
                    if constexpr (STREAM) {   // (the hit names a posting: run, position -> the session's rank, one more fetch for the ~60 hits of a query)
                        const uint32_t hr = (h.x >> 20) & 3u;
                        const unsigned long long hb = hr == 0u ? lbase[0] : hr == 1u ? lbase[1] : hr == 2u ? lbase[2] : lbase[3];
                        row = (size_t)sb.post_rank[hb + (h.x & 0xFFFFFu)];
                    } else row = (size_t)(base + (h.x >> NB));
                    const uint4 os = *reinterpret_cast<const uint4*>(ix.row_slots + row);   // the general 16-byte fragment slot, all of it in ONE load: {len, i0, i1, i2} | {len, ext offset, i0, i1}
                    const uint32_t len = os.x, j = h.y;
                    uint32_t it = EMPTY32;
                    if (j < len) it = len <= 3u ? (j == 0u ? os.y : j == 1u ? os.z : os.w) : (j < 2u ? (j == 0u ? os.z : os.w) : ix.row_ext[os.y + (j - 2u)]);   // (a position past the end names no item)
                    if (business && it != EMPTY32 && it >= SB_DIRECT && !business_ok(cur_attr, ix.meta[it].attr)) it = EMPTY32;
                    if (it != EMPTY32 && it >= SB_DIRECT && item_insert(ikeys, iacc, SB_BUCKETS, it, (int)(STREAM ? h.x >> 24 : (uint32_t)wtab[h.x & NBM])) < 0) ovf = true;
                }
                if (__ballot(ovf) != 0ull) { fail = true; c7 += 1ull << 32; }
            }
            SB_SUBTICK(tk2);
            SB_SYNC();
            if (!fail) {   // the table's contenders (exact sum at the floor; the others were collisions in their sketch word), compacted behind the candidates' room: into the hit list's words
                uint4 kq = make_uint4(EMPTY32, EMPTY32, EMPTY32, EMPTY32), aq = make_uint4(0u, 0u, 0u, 0u);
                if (lane < SB_BUCKETS) { kq = reinterpret_cast<const uint4*>(ikeys)[lane]; aq = reinterpret_cast<const uint4*>(iacc)[lane]; }
                const uint32_t kk[4] = {kq.x, kq.y, kq.z, kq.w}, aa[4] = {aq.x, aq.y, aq.z, aq.w};
                SB_SYNC();
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const bool in = kk[s4] != EMPTY32 && kk[s4] != cur_idx && aa[s4] >= floor_b;
                    const unsigned long long bm = __ballot(in);
                    if (in) hits[nt + below(bm)] = make_uint2(kk[s4], aa[s4]);
                    nt += (uint32_t)__popcll(bm);
                }
                SB_SYNC();
            }
        }
        fetch_next();   // (a query without a walk B)
        SB_TICK(tk12);
        if (fail) {   // (beyond this kernel's room: the fast kernel's back-end form takes the query if the launch sequence has one behind this kernel, else the general kernel)
            if (lane == 0u) { if (f.mid_list) f.mid_list[atomicAdd(f.mid_cnt, 1u)] = q; else f.slow_list[atomicAdd(f.slow_cnt, 1u)] = q; }
            continue;
        }
        // ---- hand-off: the query's record for vmis_finish_kernel (score = x / (10 U), ranking, public ids), as the fast kernel writes it ----
        const uint32_t M = ncand + nt;
        uint32_t lh = lane; asm volatile("" : "+v"(lh));   // (opaque, as `ln`: the LDS addresses made of the lane number below were hoisted and spilled -- and a reload's s_waitcnt vmcnt(0) here also sits out the NEXT query's requests)
        if (sb.finish_here && M <= F_FIN_ENTRIES) {   // (wave-uniform) the row finished by this wave itself, from registers: no record, nothing for the finish kernel (round 5, second half)
            uint4 e = make_uint4(0u, 0u, 0u, 0u);
            if (lh < ncand) { const unsigned long long x = ckey[lh]; e = make_uint4((uint32_t)x, (uint32_t)(x >> 32), cidx[lh], 0u); }
            else if (lh < M) { const uint2 c = hits[lh - ncand]; e = make_uint4(c.y, 0u, c.x, 1u); }
            finish_inline(ix_arg, M, U, e, lane, q, p.out_ids, p.out_scores, p.out_counts, p.how_many);
            SB_SYNC();   // (the next query clears what this one still read)
            c6 += ncand; c14 += 1ull;
            SB_TICK(tk13);
            continue;
        }
        uint32_t ovf_at = 0;
        if (M > F_FIN_ENTRIES) {   // (wave-uniform, rare)
            unsigned long long tk = 0;
            if (lane == 0u) tk = atomicAdd(f.big_ticket, (1ull << 32) | (unsigned long long)(M - F_FIN_ENTRIES));
            const uint32_t slot = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(tk >> 32), 0);
            ovf_at = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)tk, 0);
            if ((unsigned long long)ovf_at + (M - F_FIN_ENTRIES) > f.big_cap_entries) {   // no room: the general kernel redoes the query
                if (lane == 0u) { f.big_list[slot] = 0xFFFFFFFFu; f.slow_list[atomicAdd(f.slow_cnt, 1u)] = q; }
                continue;
            }
            if (lane == 0u) f.big_list[slot] = q;
        }
        {
            uint4* out = reinterpret_cast<uint4*>(f.fin + (size_t)q * F_FIN_BYTES);
            uint4* ovf = reinterpret_cast<uint4*>(f.big_arena) + ovf_at;
            if (lane == 0u) { out[0] = make_uint4(M, U, ovf_at, 0u); p.out_counts[q] = M > F_FIN_ENTRIES ? 0x80000001u : 0x80000000u; }
            for (uint32_t i = lh; i < M; i += 64u) {
                uint4 e;
                if (i < ncand) { const unsigned long long x = ckey[i]; e = make_uint4((uint32_t)x, (uint32_t)(x >> 32), cidx[i], 0u); }
                else { const uint2 c = hits[i - ncand]; e = make_uint4(c.y, 0u, c.x, 1u); }
                if (i < F_FIN_ENTRIES) { typedef uint32_t v4u __attribute__((ext_vector_type(4))); SB_NT_STORE((v4u{e.x, e.y, e.z, e.w}), reinterpret_cast<v4u*>(&out[1 + i])); } else ovf[i - F_FIN_ENTRIES] = e;
            }
        }
        SB_SYNC();   // (the next query clears what this one still read)
        c6 += ncand; c14 += 1ull;
        SB_TICK(tk13);
    }
    if (ticking && lane == 0u) {
#ifdef SRN_SBACK_SUBTICKS
        const unsigned long long v[16] = {0, tk1, tk2, tk3, tk4, c5, c6, c7, tk8, tk9, tk10, tk11, tk12, tk13, c14, c15};
        for (int i = 1; i < 16; ++i) if (v[i]) atomicAdd(&p.phase_cycles[i], v[i]);
#else
        const unsigned long long v[16] = {0, 0, 0, 0, 0, c5, c6, c7, tk8, tk9, tk10, tk11, tk12, tk13, c14, c15};
        for (int i = 5; i < 16; ++i) if (v[i]) atomicAdd(&p.phase_cycles[i], v[i]);
#endif
    }
}

hipError_t launch_shard_back(dim3 grid, hipStream_t st, const DeviceIndex& di, const LaunchParams& p, const FastParams& f, const SBackParams& sb, bool debug) {
    static bool told = false;
    if (!told && debug) { told = true; int nb = 0, ns = 0; (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)vmis_shard_back_kernel<false, false>, 64, 0);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&ns, (const void*)vmis_shard_back_kernel<false, true>, 64, 0);
        fprintf(stderr, "[srn] vmis_shard_back_kernel: %u bytes of LDS per wave; gather form %d waves per CU, streaming form %d (occupancy API)\n", SB_LDS, nb, ns); }
    if (sb.frag_post && sb.post_rank && sb.scr) hipLaunchKernelGGL((vmis_shard_back_kernel<false, true>), grid, dim3(64), 0, st, di, p, f, sb);
    else if (sb.present) hipLaunchKernelGGL((vmis_shard_back_kernel<true, false>), grid, dim3(64), 0, st, di, p, f, sb);
    else if (sb.pbyte_shift < 8u) hipLaunchKernelGGL((vmis_shard_back_kernel<false, false, true>), grid, dim3(64), 0, st, di, p, f, sb);
    else hipLaunchKernelGGL((vmis_shard_back_kernel<false, false>), grid, dim3(64), 0, st, di, p, f, sb);
    return hipGetLastError();
}
uint32_t shard_back_scratch_words() { return SB_SCR_WORDS; }

// -------------------------------------------------------------------------------------
// Neighbour slots -> positions in the posting lists (the streaming form's record, see the constants above), on the rank that fronted the query: one wave per query; the
// slots into a hash table (two candidate buckets of 8), the kept prefixes of the query's lists streamed past it (ranks, coalesced; the front end has just read them), a
// ballot per 64 entries = one word of the list's bitmap.  A neighbour found in several lists is marked in the first (the lowest bit of its list set) and only there.
// Whatever does not come out as exactly K marks goes the way of a query no front end took: marker, every rank's general kernel.
// -------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void shard_nb_positions_kernel(const char* __restrict__ prep, uint32_t prep_stride, uint32_t max_len, const uint32_t* __restrict__ xin, uint32_t in_stride,
                                                                uint32_t* __restrict__ xout, uint32_t out_stride, const uint32_t* __restrict__ post_rank, uint32_t q_lo, uint32_t q_hi,
                                                                uint32_t m, uint32_t wide) {
    __shared__ __attribute__((aligned(16))) char smem[SBP_LDS];
    uint32_t* nh = (uint32_t*)smem; uint32_t* codes = (uint32_t*)(smem + SBP_CODES);
    const uint32_t lane = threadIdx.x; const unsigned long long lt = (1ull << lane) - 1ull;
    const uint32_t mw = (m + 63u) >> 6;
    for (uint32_t q = q_lo + blockIdx.x; q < q_hi; q += gridDim.x) {
        const char* const rec = prep + (size_t)q * prep_stride;
        const PrepHead* hp = (const PrepHead*)rec;
        const uint32_t xlo = hp->xlo, L = hp->L;
        uint32_t it_kept = 0u; unsigned long long it_base = 0ull;
        if (lane < 8u && lane < L && lane < max_len) { const PrepItem pi = ((const PrepItem*)(rec + sizeof(PrepHead)))[lane]; it_kept = pi.kept; it_base = pi.base; }
        const uint32_t* const xn = xin + (size_t)q * in_stride; uint32_t* const xo = xout + (size_t)q * out_stride;
        const uint32_t K = (uint32_t)__builtin_amdgcn_readfirstlane((int)xn[0]);
        unsigned long long rm = __ballot(it_kept > 0u);
        const uint32_t nr = (uint32_t)__popcll(rm);
        if (K == 0u || K == 0xFFFFFFFFu || K > F_K_MAX || L < 1u || L > 8u || L > max_len || nr > 4u) {   // (wave-uniform)
            if (lane == 0u) xo[0] = K == 0u ? 0u : 0xFFFFFFFFu;
            continue;
        }
        const bool rel = wide && nr > 3u;
        const uint32_t NB = wide && !rel ? 3u : 4u, NBM = (1u << NB) - 1u, base = rel ? xlo : 0u;
        uint32_t ps[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { ps[r] = rm ? (uint32_t)__ffsll((long long)rm) - 1u : 0u; rm &= rm - 1ull; }
        { uint4* z = reinterpret_cast<uint4*>(nh); for (uint32_t i = lane; i < SBP_BUCKETS * 2u; i += 64u) z[i] = make_uint4(EMPTY32, EMPTY32, EMPTY32, EMPTY32);
          for (uint32_t i = lane; i < F_K_MAX / 8u + 8u; i += 64u) codes[i] = 0u; }
        __syncthreads();
        auto buckets = [&](uint32_t key, uint32_t& b1, uint32_t& b2) { b1 = (key * 0x9E3779B1u) >> 24; b2 = ((key ^ (key >> 15)) * 0x85EBCA6Bu) >> 24; };
        auto filled = [&](uint32_t b) -> uint32_t { const uint4 a = reinterpret_cast<const uint4*>(nh)[2u * b], c = reinterpret_cast<const uint4*>(nh)[2u * b + 1u];
            return (a.x != EMPTY32) + (a.y != EMPTY32) + (a.z != EMPTY32) + (a.w != EMPTY32) + (c.x != EMPTY32) + (c.y != EMPTY32) + (c.z != EMPTY32) + (c.w != EMPTY32); };
        bool bad = false;
        for (uint32_t i0 = 0; i0 < K; i0 += 64u) {   // the neighbour slots into the table: key = slot >> NB, the emptier of its two buckets, first free slot
            bool todo = i0 + lane < K;
            const uint32_t wd = xn[1u + min(i0 + lane, K - 1u)], key = wd >> NB;
            uint32_t b1, b2; buckets(key, b1, b2);
            for (int t = 0; t < 16 && __ballot(todo) != 0ull; ++t) {
                if (todo) {
                    const uint32_t n1 = filled(b1), n2 = filled(b2);
                    const bool one = n1 <= n2; const uint32_t bb = one ? b1 : b2, nn = one ? n1 : n2;   // (slots fill from the first: the first free one is slot nn, unless another lane has just taken it)
                    if (nn < 8u && atomicCAS(&nh[8u * bb + nn], EMPTY32, wd) == EMPTY32) todo = false;
                }
            }
            bad = bad || todo;
        }
        bad = __ballot(bad) != 0ull;
        __syncthreads();
        uint32_t nmem = 0u;
        if (!bad) {
            for (uint32_t r = 0; r < nr; ++r) {
                const uint32_t pl = r == 0u ? ps[0] : r == 1u ? ps[1] : r == 2u ? ps[2] : ps[3];
                const uint32_t kept = (uint32_t)__builtin_amdgcn_readlane((int)it_kept, (int)pl);
                const unsigned long long lb = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(it_base >> 32), (int)pl) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)it_base, (int)pl);
                const uint32_t lower = (1u << r) - 1u;
                unsigned long long* const bm64 = reinterpret_cast<unsigned long long*>(xo + 2u) + (size_t)r * mw;
                for (uint32_t j0 = 0; j0 < kept; j0 += 64u * SB_PF) {
                    uint32_t rk[SB_PF];
#pragma unroll
                    for (uint32_t u = 0; u < SB_PF; ++u) rk[u] = j0 + u * 64u < kept ? post_rank[lb + min(j0 + u * 64u + lane, kept - 1u)] : 0u;
#pragma unroll
                    for (uint32_t u = 0; u < SB_PF; ++u) {
                        if (j0 + u * 64u >= kept) continue;   // (wave-uniform)
                        const bool act = j0 + u * 64u + lane < kept;
                        const uint32_t key = rk[u] - base;
                        uint32_t b1, b2; buckets(key, b1, b2);
                        const uint4 q1 = reinterpret_cast<const uint4*>(nh)[2u * b1], q2 = reinterpret_cast<const uint4*>(nh)[2u * b1 + 1u], q3 = reinterpret_cast<const uint4*>(nh)[2u * b2], q4 = reinterpret_cast<const uint4*>(nh)[2u * b2 + 1u];
                        uint32_t wd = EMPTY32;
                        wd = (q1.x >> NB) == key ? q1.x : wd; wd = (q1.y >> NB) == key ? q1.y : wd; wd = (q1.z >> NB) == key ? q1.z : wd; wd = (q1.w >> NB) == key ? q1.w : wd;
                        wd = (q2.x >> NB) == key ? q2.x : wd; wd = (q2.y >> NB) == key ? q2.y : wd; wd = (q2.z >> NB) == key ? q2.z : wd; wd = (q2.w >> NB) == key ? q2.w : wd;
                        wd = (q3.x >> NB) == key ? q3.x : wd; wd = (q3.y >> NB) == key ? q3.y : wd; wd = (q3.z >> NB) == key ? q3.z : wd; wd = (q3.w >> NB) == key ? q3.w : wd;
                        wd = (q4.x >> NB) == key ? q4.x : wd; wd = (q4.y >> NB) == key ? q4.y : wd; wd = (q4.z >> NB) == key ? q4.z : wd; wd = (q4.w >> NB) == key ? q4.w : wd;
                        const uint32_t code = wd & NBM;
                        const bool first = act && wd != EMPTY32 && (code & lower) == 0u && ((code >> r) & 1u);
                        const unsigned long long mk = __ballot(first);
                        if (lane == 0u) bm64[(j0 >> 6) + u] = mk;
                        if (first) { const uint32_t idx = nmem + (uint32_t)__popcll(mk & lt); if (idx < F_K_MAX) atomicOr(&codes[idx >> 3], code << (4u * (idx & 7u))); }
                        nmem += (uint32_t)__popcll(mk);
                    }
                }
            }
        }
        __syncthreads();
        const bool ok = !bad && nmem == K;
        if (lane == 0u) { xo[0] = ok ? K : 0xFFFFFFFFu; xo[1] = 0u; }
        if (ok) for (uint32_t i = lane; i < (K + 7u) >> 3; i += 64u) xo[2u + 8u * mw + i] = codes[i];
        __syncthreads();   // (the next query clears what this one still read)
    }
}
hipError_t launch_shard_nb_positions(dim3 grid, hipStream_t st, const char* prep, uint32_t prep_stride, uint32_t max_len, const uint32_t* xin, uint32_t in_stride, uint32_t* xout, uint32_t out_stride,
                                     const uint32_t* post_rank, uint32_t q_lo, uint32_t q_hi, uint32_t m, bool wide) {
    if (q_hi > q_lo) hipLaunchKernelGGL(shard_nb_positions_kernel, grid, dim3(64), 0, st, prep, prep_stride, max_len, xin, in_stride, xout, out_stride, post_rank, q_lo, q_hi, m, wide ? 1u : 0u);
    return hipGetLastError();
}
// words of a query's position record; 0: the batch's shape has no streaming form (bitmaps + nibbles do not fit the wave's LDS room)
uint32_t shard_nb_positions_stride(uint32_t k, uint32_t m) {
    const uint32_t mw = (m + 63u) >> 6, ncw = (std::min<uint32_t>(k, F_K_MAX) + 7u) >> 3;
    if (32u * mw + 8u * ncw > SB_REC_ROOM) return 0u;   // (in LDS: the bitmaps + a weight byte per member)
    return (2u + 8u * mw + ncw + 1u) / 2u * 2u;
}

// -------------------------------------------------------------------------------------
// The neighbours' PRESENCE BYTES (round 5, second half): bit g of session r's byte = shard g holds an item of r.  Built once per group from the shards' presence bitmaps
// (all-gathered); the rank that fronts a query writes every neighbour's byte behind the neighbour slots of its exchange record, and a back end of shard g asks only for the
// fragments whose bit g is set -- half the requests at G = 8, and no look-up of its own in front of them (what the per-shard bitmap of SRN_SBACK_BITMAP cost).
// -------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void presence_bytes_kernel(const uint32_t* __restrict__ bitmaps, size_t block_words, uint32_t G, uint64_t n, uint8_t* __restrict__ out) {
    for (uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += (uint64_t)gridDim.x * 256) {
        uint32_t b = 0;
        for (uint32_t g = 0; g < G; ++g) b |= ((bitmaps[(size_t)g * block_words + (r >> 5)] >> (r & 31u)) & 1u) << g;
        out[r] = (uint8_t)b;
    }
}
hipError_t launch_presence_bytes(hipStream_t st, const uint32_t* bitmaps, size_t block_words, uint32_t G, uint64_t n, uint8_t* out) {
    if (n) hipLaunchKernelGGL(presence_bytes_kernel, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 1u << 16)), dim3(256), 0, st, bitmaps, block_words, G, n, out);
    return hipGetLastError();
}
__global__ __launch_bounds__(64) void shard_nb_presence_kernel(const char* __restrict__ prep, uint32_t prep_stride, uint32_t max_len, uint32_t* __restrict__ xchg, uint32_t stride, uint32_t k,
                                                               const uint8_t* __restrict__ pbytes, uint32_t n_kept, uint32_t q_lo, uint32_t q_hi, uint32_t wide) {
    const uint32_t lane = threadIdx.x;
    for (uint32_t q = q_lo + blockIdx.x; q < q_hi; q += gridDim.x) {
        const char* const rec = prep + (size_t)q * prep_stride;
        const PrepHead* hp = (const PrepHead*)rec;
        const uint32_t xlo = hp->xlo, L = hp->L;
        uint32_t it_kept = 0u;
        if (lane < 8u && lane < L && lane < max_len) it_kept = ((const PrepItem*)(rec + sizeof(PrepHead)))[lane].kept;
        uint32_t* const xq = xchg + (size_t)q * stride;
        const uint32_t K = (uint32_t)__builtin_amdgcn_readfirstlane((int)xq[0]);
        if (K == 0u || K == 0xFFFFFFFFu || K > k) continue;   // (wave-uniform: nothing to mark)
        const uint32_t nr = (uint32_t)__popcll(__ballot(it_kept > 0u));
        const bool rel = wide && nr > 3u;
        const uint32_t NB = wide && !rel ? 3u : 4u, base = rel ? xlo : 0u;
        uint8_t* const pb = reinterpret_cast<uint8_t*>(xq + 1u + k);
        for (uint32_t i = lane; i < K; i += 64u) pb[i] = pbytes[min(base + (xq[1u + i] >> NB), n_kept)];
    }
}
hipError_t launch_shard_nb_presence(dim3 grid, hipStream_t st, const char* prep, uint32_t prep_stride, uint32_t max_len, uint32_t* xchg, uint32_t stride, uint32_t k, const uint8_t* pbytes, uint32_t n_kept,
                                    uint32_t q_lo, uint32_t q_hi, bool wide) {
    if (q_hi > q_lo) hipLaunchKernelGGL(shard_nb_presence_kernel, grid, dim3(64), 0, st, prep, prep_stride, max_len, xchg, stride, k, pbytes, n_kept, q_lo, q_hi, wide ? 1u : 0u);
    return hipGetLastError();
}

// frag_post[e] = frag8[post_rank[e]]: the fragments once more, in posting order (streaming form)
__global__ __launch_bounds__(256) void frag_post_kernel(const uint32_t* __restrict__ post_rank, const uint2* __restrict__ frag8, uint2* __restrict__ out, uint64_t n) {
    for (uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (uint64_t)gridDim.x * 256) out[e] = frag8[post_rank[e]];
}
hipError_t launch_frag_post(hipStream_t st, const uint32_t* post_rank, const uint2* frag8, uint2* out, uint64_t n) {
    if (n) hipLaunchKernelGGL(frag_post_kernel, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 1u << 20)), dim3(256), 0, st, post_rank, frag8, out, n);
    return hipGetLastError();
}

}  // namespace srn

// Internal interface between the kernels (srn_kernels.hip) and the device runtime (srn_runtime.hip): LDS geometry, the
// prep record layout and the launchers.  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>

#include "srn_internal.h"

namespace srn {

static constexpr uint32_t EMPTY32 = 0xFFFFFFFFu;
static constexpr int CAND_CAP = 1024;       // candidate buffer of the final top-n (entries)
static constexpr int MISC_WORDS = 64;       // scalar words at the head of LDS
static constexpr int kBlock = 512;            // threads per workgroup of the predict kernel
static constexpr int SEL_BITS = 11, SEL_BINS = 1 << SEL_BITS;   // radix-select histogram: 11 bits per pass
static constexpr int SEL_WORDS = SEL_BINS + SEL_BINS / 32;

// launch-time geometry, identical for every block of a launch (all LDS offsets multiples of 16)
struct KernelCfg {
    uint32_t sess_slots, item_slots;   // session table: power of two; item hash table: 4 * item_buckets slots
    uint32_t item_buckets;             // prime number of 4-slot buckets (double hashing needs a full cycle)
    uint32_t hot_slots;                // direct-mapped accumulators for dense idx < hot_slots (idx = popularity order)
    uint32_t sketch_slots, sketch_shift;   // power-of-two upper-bound words for every other item (0 = none); 32 - log2
    uint32_t sum_bits;                 // hot accumulator = (touch count << sum_bits) + signed weight sum
    uint32_t num_bits;                 // low bits of a session slot that hold the numerator
    uint32_t q_cap;                    // capacity of the per-query item arrays (>= max_len, multiple of 4)
    uint32_t off_q, off_wave, off_b, off_a;   // LDS byte offsets: query arrays, per-wave scratch, region B, region A
    uint32_t region_a_bytes;                  // size of region A in LDS (0 when the tables live in global memory)
    uint32_t no_merge;                        // test knob: candidate sessions through the hash table even where the merge tree applies
};

// the rest of a launch's arguments: query list of a retry pass (or null), where to queue what does not fit, global-table arena, neighbour-list copies
struct LaunchAux { const uint32_t* qlist; const uint32_t* qlist_n; uint32_t* retry_list; uint32_t* retry_cnt; char* gscratch; unsigned long long gscratch_stride; char* nb_spill; };

// record written by the prep kernel for every query: PrepHead + max_len * PrepItem, positions counted from the most recent item
struct PrepHead { uint32_t U, rmax, xlo, sumw, P, nruns, L, n_staged, run_start[8], cur_attr, unsafe; };   // S_U, S_RMAX, S_XLO, S_SUMW, S_P of the query; number of non-empty lists, session length as given, sum of the lists' kept counts, where the first 8 lists start; cur_attr = the attribute byte of the current (most recent) item, SRN_ATTR_NONE if unknown
struct PrepItem { uint32_t idx, len, pre, kept; unsigned long long base; };   // dense idx | kNone, truncated list length, prefix of len, entries >= x_lo (a prefix of the list), list start

// item-sharded index, lists mode (srn_shard.hip): what a shard knows about one evolving position of a query
struct ShardPos { uint32_t idx, len; unsigned long long base; };   // dense idx | kNone; list length truncated to m (0: not this shard's item, unknown, or an older duplicate); list start

// ---- fast path (srn_fast.hip): the lean kernel for the common query shape; everything else goes to vmis_predict_kernel ----
// LDS map of vmis_fast_kernel, bytes.  The row slots of the fast path hold 16-bit byte offsets relative to F_HOT, so the layout is fixed.
#ifndef SRN_FAST_SMALL
#define SRN_FAST_SMALL 1
#endif
#if defined(SRN_FAST_EXP_LDS40)   // experiment ("a front end of its own?"): 40 KB, four workgroups per CU -- timing runs only, the walks' layout does not survive it
static constexpr uint32_t F_HOT_WORDS = 3840, F_SK_WORDS = 1024, F_DUMP_WORDS = 256, F_TABLE_BUCKETS = 127, F_TABLE_WORDS = 512, F_SURV_WORDS = 512, F_WG_PER_CU = 4;
#elif SRN_FAST_SMALL   // 48 KB: three workgroups per CU
static constexpr uint32_t F_HOT_WORDS = 4096, F_SK_WORDS = 4096, F_DUMP_WORDS = 256, F_TABLE_BUCKETS = 127, F_TABLE_WORDS = 512, F_SURV_WORDS = 512, F_WG_PER_CU = 3;
#else                // 80 KB: two workgroups per CU
static constexpr uint32_t F_HOT_WORDS = 4096, F_SK_WORDS = 8192, F_DUMP_WORDS = 1024, F_TABLE_BUCKETS = 127, F_TABLE_WORDS = 512, F_SURV_WORDS = 768, F_WG_PER_CU = 2;
#endif
static constexpr uint32_t F_MISC = 0, F_WLUT = 256, F_W10 = 512, F_NBL = 1024, F_K_MAX = 1536, F_CAND = F_NBL + F_K_MAX * 4, F_CAND_CAP = 160;
static constexpr uint32_t F_HOT = 9216, F_SKETCH = F_HOT + F_HOT_WORDS * 4, F_DUMP = F_SKETCH + F_SK_WORDS * 4, F_TABLE = F_DUMP + F_DUMP_WORDS * 4;
// walk B's hit list -- (session slot, row position) pairs -- lives in the direct-mapped words, which are dead by then (walk B maps every
// offset below the sketch to the LAST direct-mapped word, kept 0); exact table: prime number of 4-slot buckets, keys then sums
static constexpr uint32_t F_HITS = F_HOT, F_HIT_CAP = F_HOT_WORDS / 2 - 1, F_ZERO_OFF = (F_HOT_WORDS - 1) * 4;
// REPLICATED accumulators of the hottest items.  One 64-lane ds_add of walk A takes as many LDS cycles as its fullest bank has lanes, and adds to ONE address
// serialise: on config 3 a tenth of all row elements are the most popular item, a simulation of the walk over real neighbour lists gives 6.6 LDS cycles per add
// instruction against ~4.1 when the F_REP_ITEMS hottest items have F_REP words each, the word chosen by the ROW (its recency rank mod F_REP: fixed at attach time,
// baked into the row slots' offsets -- no instruction in the walk).  The replicas take the top of the direct-mapped words (items F_DIRECT.. go to the sketch
// instead), the harvest adds them up.
#ifndef SRN_FAST_REPLICAS
#define SRN_FAST_REPLICAS 1
#endif
#ifndef SRN_FAST_REP_ITEMS
#define SRN_FAST_REP_ITEMS 16   // (32 and 48 measured in round 3: see DESIGN.md)
#endif
static constexpr uint32_t F_REP_ITEMS = SRN_FAST_REPLICAS ? SRN_FAST_REP_ITEMS : 0, F_REP = 8, F_DIRECT = F_HOT_WORDS - F_REP_ITEMS * F_REP;   // items < F_DIRECT have a word of their own
static constexpr uint32_t F_SURV = F_TABLE + F_TABLE_WORDS * 8;          // survivors of the integer floors, packed (idx << 20 | acc), then the threshold histogram
static constexpr uint32_t F_LDS_BYTES = F_SURV + F_SURV_WORDS * 4;
static constexpr uint32_t F_WORK = F_NBL, F_WORK_WORDS = (F_LDS_BYTES - F_WORK) / 4;   // merge buffers: 2 * n_staged words
static constexpr uint32_t F_M_MAX = 2560;
static constexpr uint32_t F_FIN_BYTES = 1024, F_FIN_ENTRIES = F_FIN_BYTES / 16 - 1;   // a query's record: header + 63 entries
static_assert(F_CAND + F_CAND_CAP * 12 <= F_HOT, "candidate buffer overlaps the accumulators");
static constexpr uint32_t F_BIG_TOTAL = 80 * 1024;   // LDS of the MID instantiation's BIG form (two workgroups per CU): the same map with the merge buffers' room grown to 19 072 words (n <= 9 404 staged entries)
static constexpr uint32_t F_MID_LISTS = 10, F_MID_LMAX = 10, F_MID_CLASSES = 63;   // the MID instantiation of vmis_fast_kernel: lists per query, session length, similarity numerators
// the LONG instantiation (round 5): sessions of 11..20 items -- the reference's README lets last_items_in_session_range go to 20, and linear_score is NEGATIVE from the eleventh
// position on (mod.rs:110-116).  A form of MID's BIG layout (80 KB of LDS); slots carry a 5-bit list CODE (31 - list number: the first copy of a session in the merged run is its
// first match), numerators are summed per session in bytes, the neighbour list is 64-bit {slot, signed weight}
static constexpr uint32_t F_LONG_LISTS = 20, F_LONG_LMAX = 20, F_LONG_CLASSES = 255, F_LONG_NB = 5;
static constexpr uint32_t F_LONG_RES_WORDS = F_K_MAX * 2 + F_M_MAX / 4 + 256;   // the tail of the merge buffers' room it reserves: neighbour list (uint2) | numerator bytes of the m-cut | class histogram
static_assert(F_LDS_BYTES * F_WG_PER_CU <= 160 * 1024, "LDS budget");
static_assert(F_DUMP + F_DUMP_WORDS * 4 - F_HOT <= 65536, "16-bit row offsets");
// (round 6) The PERSISTENT latency path's control block: 256 bytes of pinned, device-mapped, coherent host memory per resident workgroup.  The host writes a session
// and then its number (seq); thread 0 of the workgroup polls seq (1.6 us round trip measured, tools/ring_probe.hip), the workgroup serves the session exactly like the
// one-launch form (vmis_fast_kernel<TINY>: prep record, query, row finished from registers into the pinned row) and answers with done_seq.  No launch, no completion signal.
struct ServeCtl {
    // line 0, host -> device: the doorbell and the first five items in ONE 64-byte line (the workgroup polls it with one 16-lane load); seq is written LAST;
    // check = seq ^ len ^ (xor of the 32-bit halves of items[0 .. min(len, 5))): a line seen torn is read again
    uint32_t seq, len, check, stop;   // stop != 0: leave
    unsigned long long items[5]; unsigned long long pad1;
    // line 1: items 5..12 of a longer session (read after the doorbell)
    unsigned long long more[8];
    // line 2, host -> device, written at the launch: idle_ticks: leave after this many wall-clock ticks (100 MHz) without a request -- a host that died leaves no kernel behind
    uint32_t pad2[2]; unsigned long long idle_ticks; uint32_t pad3[12];
    // line 3, device -> host: done_seq is written LAST; status 0 = the row is final (count entries), 1 = the session needs the kernels behind the fused form (the caller
    // takes the launch path); alive = 0 once the workgroup has left; stamp: 100 MHz ticks -- waited for the doorbell | doorbell -> record written | doorbell -> answer
    uint32_t done_seq, status, count, alive; uint32_t stamp[4]; uint32_t pad4[8];
};
static_assert(sizeof(ServeCtl) == 256, "ServeCtl layout");
struct FastParams {
    const RowQuad* row_packed; const uint32_t* row_ext16;   // 64-byte slots of 16-bit LDS offsets + overflow blocks (8 items per 16 bytes)
    const ItemMeta* meta_sample;   // meta[] of the 512 most popular items in the order the threshold sample reads them: entry 64 w + l = item 8 l + w
    double inv_idf_hot[8];      // 1 / max idf_eff over the dense idx [512 c, 512 c + 512): popular items have small idf, so their integer floor is much tighter
    double inv_idf_hi;          // 1 / max idf_eff over all items
    uint32_t* slow_list; uint32_t* slow_cnt;   // queries the fast kernel hands to vmis_predict_kernel
    uint32_t* bigq_list; uint32_t* bigq_cnt;   // ... and what MID passes on only because its merged lists outgrow the 53 KB layout: MID's BIG form (80 KB of LDS, two workgroups per CU) takes them before the general kernel
    uint32_t* long_list; uint32_t* long_cnt;   // sessions of 11..20 items: the LONG instantiation takes them before the general kernel (null: no such tier in this launch)
    uint32_t* mid_list; uint32_t* mid_cnt;     // queries of 5..10 lists / <= 10 items / numerators up to 63: the fast kernel's MID instantiation takes them before the general kernel (null: no such tier in this launch)
    char* fin;                  // per-query records for vmis_finish_kernel: F_FIN_BYTES each, at q * F_FIN_BYTES
    char* big_arena; uint32_t* big_list; unsigned long long* big_ticket; uint32_t big_cap_entries;   // queries with > 63 entries: overflow entries, list for vmis_finish_big_kernel,
                                                                                                      // ticket = (list length << 32 | arena entries in use)
    uint32_t nb;                // low bits of a session slot that hold the set of runs (lists) with the session: 4, or 3 when the ranks need 29 bits
    uint32_t max_runs;          // = nb (a query with more lists than that takes 4 bits and ranks relative to its cut x_lo, if they fit 28 bits)
    // neighbours pipeline of the shard group (launch_fast modes 1 and 2): [nq][xchg_stride] words, a query's row = K | K packed slots (K = 0xFFFFFFFF: not served by the
    // front end -- the general kernel takes it on every rank); the front end works on the queries [q_base, p.nq)
    uint32_t* xchg; uint32_t xchg_stride; uint32_t q_base;
    // the order the batch is served in (round 5; null: query index order): the batch's queries sorted by their most popular item, low word of entry i = the i-th query.
    // Workgroup b serves positions of the (b % 8)-th eighth of the order -- block b runs on XCD b % 8: what one XCD's L2 sees is a window of like queries
    const unsigned long long* order;
    uint64_t tiny_items[8]; uint32_t tiny_len, host_seq;   // (the TINY launch) the session's items in the kernel arguments (tiny_len = 0: read p.items_flat), the call's number
    ServeCtl* serve;            // (the TINY launch) non-null: the persistent form -- grid of ONE workgroup that serves the sessions the host posts here until told to leave
    uint32_t* host_words;       // (the TINY launch) pinned words the kernel publishes the sequence's counters in: [1] handed to the general kernel, [2] listed for MID, [3] for MID's BIG form, [4] queries with > 63 entries, and last of all [5] = host_seq: the host may read the row
};
// the batch's order keys (written by the prep kernel) -> sorted (srn_build_gpu.hip: rocPRIM radix sort on the key bits); temp == nullptr: only *temp_bytes is set
hipError_t sort_order_keys(hipStream_t st, const unsigned long long* in, unsigned long long* out, size_t n, void* temp, size_t* temp_bytes);

// ---- the item-sharded index's own back end (srn_sback.hip, round 5): one WAVE per query over a shard's row fragments -------------------------------------------------
// Geometry of a wave's accumulators (words): SB_H direct-mapped (the shard's most popular items; the top SB_REP_ITEMS * SB_REP of them are the replicated words of the
// SB_REP_ITEMS hottest), SB_S sketch words, SB_DUMP dump words -- baked into the frag8 slots' 16-bit LDS byte offsets at attach time.
static constexpr uint32_t SB_H = 1024, SB_S = 1024, SB_DUMP = 64, SB_REP_ITEMS = 16, SB_REP = 8, SB_DIRECT = SB_H - SB_REP_ITEMS * SB_REP;
struct SBackParams {
    const uint2* frag8;        // [n_kept + 1] 8-byte fragment slots by recency rank: four 16-bit offsets, or {0xFFFF, len, overflow block index} for a fragment of > 4 items
    const uint4* ext8;         // overflow blocks: 8 offsets per 16 bytes, ALL items of a long fragment
    const uint32_t* present;   // one bit per session: the fragment is not empty (null: not consulted)
    const ItemMeta* sample;    // meta[] of the shard's 256 most popular items, zero-padded
    double inv_idf_chunk[SB_H / 256];   // 1 / max idf_eff over the dense idx [256 c, 256 c + 256)
    double inv_idf_all;        // 1 / max idf_eff over all items of the shard
    // the streaming form (all three set, or the gather form runs): the fragments in POSTING order of the replicated lists the batch's records were written against
    const uint2* frag_post;    // [number of postings] frag_post[e] = frag8[post_rank[e]]
    const uint32_t* post_rank; // the replicated posting lists (recency ranks)
    uint32_t pbyte_shift;      // < 8: the exchange records carry a presence byte per neighbour behind the slots (at word 1 + k), this shard's bit is pbyte_shift; 8: they do not
    uint32_t finish_here;      // the serving wave finishes rows of <= 63 entries itself (score, ranking, public ids) instead of leaving a record for vmis_finish_kernel
    uint32_t* scr;             // shard_back_scratch_words() words per wave of the grid: the members' slots and fragments between walk A and walk B
};
hipError_t launch_rows_to_frag8(hipStream_t st, const uint64_t* row_off, const uint32_t* row_items, uint64_t n_rows, const uint32_t* block_base, uint2* frag8, uint4* ext8, uint32_t* present);   // block_base in 16-byte blocks
hipError_t launch_frag_post(hipStream_t st, const uint32_t* post_rank, const uint2* frag8, uint2* out, uint64_t n);
uint32_t shard_back_scratch_words();
hipError_t launch_presence_bytes(hipStream_t st, const uint32_t* bitmaps, size_t block_words, uint32_t G, uint64_t n, uint8_t* out);   // bit g of out[r] = bit r of bitmap g
hipError_t launch_shard_nb_presence(dim3 grid, hipStream_t st, const char* prep, uint32_t prep_stride, uint32_t max_len, uint32_t* xchg, uint32_t stride, uint32_t k, const uint8_t* pbytes, uint32_t n_kept,
                                    uint32_t q_lo, uint32_t q_hi, bool wide);
hipError_t launch_shard_nb_positions(dim3 grid, hipStream_t st, const char* prep, uint32_t prep_stride, uint32_t max_len, const uint32_t* xin, uint32_t in_stride, uint32_t* xout, uint32_t out_stride,
                                     const uint32_t* post_rank, uint32_t q_lo, uint32_t q_hi, uint32_t m, bool wide);
uint32_t shard_nb_positions_stride(uint32_t k, uint32_t m);   // words per query of the streaming form's exchange record; 0: this (k, m) has none
hipError_t launch_shard_back(dim3 grid, hipStream_t st, const DeviceIndex& di, const LaunchParams& p, const FastParams& f, const SBackParams& sb, bool debug = false);

// ---- launchers (srn_kernels.hip) -------------------------------------------------------------
// the predict kernel: stage 0 = fused, 1..3 = the item-sharded pipeline's stages A..C; global_tables = the retry pass with its
// tables in a global-memory arena (stage 0 only)
hipError_t launch_predict(bool masks, bool slot64, bool global_tables, int stage, dim3 grid, size_t lds, hipStream_t st, const DeviceIndex& di,
                          const LaunchParams& p, const KernelCfg& c, const uint32_t* qlist, const uint32_t* qn, uint32_t* retry_list,
                          uint32_t* retry_cnt, char* gscratch, unsigned long long gscratch_stride, char* nb_spill, const ShardIO& sh, int wg_per_cu = 2);
hipError_t launch_prep(hipStream_t st, const DeviceIndex& di, const uint64_t* items_flat, const uint32_t* q_off, uint32_t nq, uint32_t m,
                       uint32_t max_len, char* out, uint32_t stride, uint32_t* zero_a = nullptr, uint32_t* zero_b = nullptr,
                       const IdSlot* loc_table = nullptr, uint32_t loc_mask = 0, unsigned long long* okeys = nullptr);   // zero_a[0..7], zero_b[0]: counters cleared by the prep kernel; loc_table: the item shard's id table (the record's idx), di = the whole index's dictionary and lists
hipError_t launch_finish_big(hipStream_t st, const DeviceIndex& di, const FastParams& f, uint64_t* out_ids, double* out_scores, uint32_t* out_counts, uint32_t how_many, uint32_t grid, const uint32_t* cnt_retry = nullptr, const uint32_t* cnt_slow = nullptr, uint32_t* host_words = nullptr);
hipError_t launch_finish(hipStream_t st, const DeviceIndex& di, const FastParams& f, uint64_t* out_ids, double* out_scores, uint32_t* out_counts, uint32_t nq, uint32_t how_many,
                         const uint32_t* cnt_slow = nullptr, uint32_t* host_words = nullptr);   // (cnt_slow + host_words: the call's path counters published to pinned words, the latency path)   // scores, ranking, public ids of the rows the fast kernel served
hipError_t launch_shard_lists_head(hipStream_t st, const DeviceIndex& di, const uint64_t* items_flat, const uint32_t* q_off, uint32_t nq, uint32_t m, uint32_t max_len,
                                   ShardPos* pos_out, int* head);
hipError_t launch_shard_lists_count(hipStream_t st, const DeviceIndex& di, const uint32_t* q_off, uint32_t nq, uint32_t max_len, const ShardPos* pos_in, const int* head,
                                    uint32_t* kept, int* tot);
hipError_t launch_shard_lists_copy(hipStream_t st, const DeviceIndex& di, uint32_t nq, uint32_t max_len, const ShardPos* pos_in, const uint32_t* kept, const long long* off,
                                   uint32_t* out);
hipError_t launch_shard_prep(hipStream_t st, const uint64_t* items_flat, const uint32_t* q_off, uint32_t nq, uint32_t max_len, uint32_t n_shards, const uint32_t* kept_g,
                             const long long* off_g, unsigned long long shard_stride, const int* head, const ShardPos* pos_local, char* out, uint32_t stride,
                             const unsigned long long* shard_base = nullptr, bool direct = false);   // shard_base [n_shards] (device): where each shard's segment starts in the gathered buffer (null: g * shard_stride); direct: one shard, lists read in place
hipError_t launch_shard_max(hipStream_t st, int* dst, const int* src, size_t n);
hipError_t launch_shard_offsets(hipStream_t st, const uint32_t* kept_g, uint32_t nq, uint32_t max_len, uint32_t n_shards, long long* off_g, unsigned long long* tot_dev,
                                unsigned long long* tot_host, unsigned long long* chunk_scratch);
hipError_t launch_shard_min(hipStream_t st, int* dst, const int* src, size_t n);
hipError_t launch_shard_scrub_counts(hipStream_t st, uint32_t* cnt_g, uint32_t n_shards, uint32_t nq, uint32_t* flag);
hipError_t launch_shard_mark(hipStream_t st, const uint32_t* flag, uint32_t nq, uint32_t* out_counts);
hipError_t launch_shard_fill_i32(hipStream_t st, int* dst, int v, size_t n);
hipError_t launch_shard_merge_topn(hipStream_t st, const char* part, size_t block_bytes, uint32_t n_shards, uint32_t nq, uint32_t how_many, uint64_t* out_ids, double* out_scores,
                                   uint32_t* out_counts);
hipError_t launch_fast(dim3 grid, hipStream_t st, const DeviceIndex& di, const LaunchParams& p, const FastParams& f, bool debug = false, int mode = 0, bool mid = false, bool big = false, bool lng = false, bool tiny = false);   // tiny: ONE workgroup = one query, prep record and finish inside (the latency path)   // debug: say the occupancy once (the SRN_DEBUG knob, read by the runtime); mode: 0 fused, 1 front end only (neighbour lists -> f.xchg), 2 back end only (neighbour lists <- f.xchg); mid: the MID instantiation over f.mid_list (mode 0 only)
hipError_t launch_rows_to_packed(hipStream_t st, const uint64_t* row_off, const uint32_t* row_items, uint64_t n_rows, const uint32_t* block_base,
                                 uint32_t* packed, uint32_t* ext16, bool frag = false);   // grid = ceil((n_rows + 1) / 1024) blocks of 1024; block_base in 16-byte blocks
hipError_t launch_rows_to_frags(hipStream_t st, const uint64_t* row_off, const uint32_t* row_items, uint64_t n_rows, const uint32_t* block_base,
                                uint32_t* slots, uint32_t* ext);   // 16-byte fragment slots of an item shard; block_base in items
hipError_t launch_rows_to_slots(hipStream_t st, const uint64_t* row_off, const uint32_t* row_items, uint64_t n_rows, const uint32_t* block_base,
                                uint32_t* slots, uint32_t* ext);   // grid = ceil((n_rows + 1) / 1024) blocks of 1024

}  // namespace srn

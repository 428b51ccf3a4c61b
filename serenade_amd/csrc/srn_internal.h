// Internal declarations shared by the host index builder, the C ABI glue and the HIP kernels.
#pragma once
#include <cstddef>
#include <cstdint>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/serenade_hip.h"

namespace srn {

// ---- error plumbing (thread-local message behind srn_last_error) ---------------------------
void set_error(const std::string& msg);
int fail(int code, const std::string& msg);
std::string last_error_string();   // this thread's message (copied: a combining round hands its leader's error to every member)

// ---- u64 item id -> dense item index: open addressing, 16-byte slots (one load per probe) ----
struct IdSlot {
    uint64_t key;
    uint32_t idx;  // 0xFFFFFFFF = empty
    uint32_t pad;
};
static constexpr uint32_t kNone = 0xFFFFFFFFu;

static inline uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33; return x;
}

// ---- training sessions (host) ------------------------------------------------------------
struct Sessions {
    std::vector<uint64_t> off, items;
    std::vector<uint32_t> ts;
    std::vector<uint64_t> session_ids;  // original SessionId column, for diagnostics
};
int sessions_from_tsv(const char* path, Sessions& out);
uint64_t sessions_length_quantile(const uint64_t* off, size_t n, double q);

// ---- the flat index (host copy).  Layout and invariants: DESIGN.md "Data layout in HBM" ----
struct FlatIndex {
    uint64_t n_items = 0, n_sessions_total = 0, n_kept = 0, nnz_rows = 0, nnz_post = 0;
    uint64_t m_index = 0, max_session_len = 0, max_row_len = 0;
    double idf_weighting = 1.0;
    bool lists_complete = true;             // every posting list holds ALL sessions of its item that are at least as recent as its last entry (and is complete if
                                            // shorter than m_index): true for every index built here; a pre-built (Avro) index may not satisfy it -- then the
                                            // first-match position comes from the rows (the reference's contains() test), not from posting-list membership
    std::vector<uint32_t> viol;             // [n_items] iff !lists_complete on an index whose loader measured it (srn_avro.cpp): 1 + the most recent rank of a LISTED session whose row holds the
                                            // item but which the item's list does not name (0: none; 0xFFFFFFFF: the list names a session without the item).  A query whose items all have
                                            // viol <= its cut x_lo is exact on the position-set path; the prep kernel flags the others for the general kernel's row pass
    bool postings_only = false;             // the REPLICATED part of an item-sharded index (srn_index_postings_view): dictionary, idf / attributes and posting lists of the whole index,
                                            // no rows -- what every rank of a shard group keeps beside its shard for the neighbours pipeline; answers no predict call itself
    uint32_t shard = 0, n_shards = 1;       // item-sharded index: this shard holds the items with owner(id) == shard
    uint64_t total_pairs = 0;               // (session,item) pairs of ALL kept sessions = idf numerator (== nnz_rows when unsharded)
    std::vector<uint64_t> item_id;          // [n_items]   public id of each dense idx; idx = popularity order (count desc, id asc)
    std::vector<uint32_t> id_rank;          // [n_items]   rank of the public id in ascending order (final tie-break key)
    std::vector<double> idf;                // [n_items]
    std::vector<uint8_t> attr;              // [n_items]   SRN_ATTR_* or SRN_ATTR_NONE
    std::vector<uint64_t> post_off;         // [n_items+1]
    std::vector<uint32_t> post_rank;        // [nnz_post]  recency ranks, strictly descending per item
    std::vector<uint64_t> row_off;          // [n_kept+1]  rows addressed by recency rank
    std::vector<uint32_t> row_items;        // [nnz_rows]  item idx (row order = ascending public id, as handed in)
    std::vector<uint32_t> rank_to_session;  // [n_kept]    reference session index of each rank
    std::vector<IdSlot> id_table;           // power-of-two open addressing table
    uint32_t id_mask = 0;
    uint32_t lookup(uint64_t id) const {
        if (id_table.empty()) return kNone;
        uint32_t h = (uint32_t)mix64(id) & id_mask;
        for (;;) { const IdSlot& s = id_table[h]; if (s.idx == kNone) return kNone; if (s.key == id) return s.idx; h = (h + 1) & id_mask; }
    }
};
int build_flat_index(const srn_sessions_view_t& v, size_t m_index, size_t max_session_len, double idf_weighting,
                     uint32_t shard, uint32_t n_shards, FlatIndex& out);
static inline uint32_t item_owner(uint64_t id, uint32_t n_shards) { return (uint32_t)(mix64(id ^ 0x9E3779B97F4A7C15ull) % n_shards); }
int build_flat_index_gpu(const srn_sessions_view_t& v, size_t m_index, size_t max_session_len, double idf_weighting, int device,
                         FlatIndex& out);   // same result, built with rocPRIM sorts on the GPU (srn_build_gpu.hip)
int build_flat_index_from_avro(const char* base_path, FlatIndex& out);   // <base>/itemindex/*.avro + <base>/sessionindex/*.avro (srn_avro.cpp)
int shard_flat_index(const FlatIndex& full, uint32_t shard, uint32_t n_shards, FlatIndex& out);   // shard `shard` of an unsharded index (same bytes as build_flat_index(..., shard, n_shards))
int check_has_rows(const FlatIndex& ix, const char* what);   // SRN_EINVAL for a postings-only view (srn_index_postings_view): it has no rows to cut, save, shard or serve from
int save_flat_index(const FlatIndex& ix, const char* path);
int load_flat_index(const char* path, FlatIndex& ix);

// ---- device side ---------------------------------------------------------------------------
struct ItemMeta {   // one 16-byte gather per scored item
    double idf;
    uint32_t id_rank;   // rank of the public id (ascending): tie-break key of the final ranking
    uint32_t attr;      // SRN_ATTR_* byte
};
struct alignas(16) RowQuad { uint32_t x, y, z, w; };
struct DeviceIndex {  // pointers into HBM; passed by value to the kernels
    const IdSlot* id_table; uint32_t id_mask;
    const ItemMeta* meta;        // [n_items] by dense idx
    const uint64_t* id_sorted;   // [n_items] public ids ascending (indexed by id_rank)
    const uint64_t* post_off; const uint32_t* post_rank;
    const uint32_t* viol;        // [n_items] or null (every list complete): FlatIndex::viol -- the prep kernel sets PrepHead::unsafe from it
    // Rows in HBM: one 64-byte, 64-byte-aligned slot per session, addressed by recency rank -- ONE line fetch per
    // neighbour and no offset lookup.  word 0 = row length; length <= 15: words 1..15 hold the items; longer rows:
    // word 1 = offset (in items) of items 14.. in row_ext, words 2..15 = items 0..13.
    const RowQuad* row_slots;   // 4 quads per slot
    const uint32_t* row_ext;
    uint32_t n_items, n_kept;
    uint32_t row_frag;       // item shards (n_shards > 1): row_slots holds 16-byte FRAGMENT slots instead -- {len, i0, i1, i2}, or {len, offset in row_ext, i0, i1} with
                             // items 2.. in row_ext when the fragment has > 3 items: a shard sees ~len / n_shards items of a row, so its row memory is 16 B per session, not 64
    double idf_hi, idf_lo;   // max / min over items of (idf > 0 ? idf : 1)
};

struct LaunchParams {
    uint32_t nq, k, m, how_many, flags, max_len;
    const uint64_t* items_flat; const uint32_t* q_off;   // device
    uint64_t* out_ids; double* out_scores; uint32_t* out_counts;   // device
    uint32_t* stats;      // [nq*8] or null
    uint32_t* nb_rank; uint32_t* nb_num; uint32_t* nb_cnt;   // debug neighbour dump or null
    unsigned long long* phase_cycles;                       // [16] per-phase shader cycles (debug) or null
    const char* prep; uint32_t prep_stride;                 // per-query records of the prep kernel (or null: translate in the main kernel)
};

struct Workspace;   // per-call device scratch + events, owned by the index handle's pool
struct DeviceState;  // uploaded arrays + workspace pool

DeviceState* device_attach(const FlatIndex& ix, int device);   // nullptr on failure (error set)
void device_release(DeviceState* d);
int device_update_attr(DeviceState* d, const FlatIndex& ix);
void device_refresh_fast_bounds(DeviceState* d, const FlatIndex& ix);   // idf bounds of the fast kernel's integer floors (srn_runtime.hip)
void reload_knobs();                                                     // re-read the SRN_* test / experiment knobs from the environment
uint64_t device_bytes(const DeviceState* d);
// item-sharded index, lists mode: prep records (PrepHead + max_len * PrepItem per query) written against a gathered posting buffer
struct ExtLists { const char* prep; uint32_t prep_stride; const uint32_t* post_rank;
                  // the shard group's neighbours pipeline: mode 1 = front end only (neighbour lists of the queries [q_lo, nq) -> xchg), 2 = back end (neighbour lists <- xchg)
                  int mode = 0; uint32_t* xchg = nullptr; uint32_t xchg_stride = 0; uint32_t q_lo = 0;
                  const unsigned long long* order = nullptr;     // mode 2: the batch's serving order (device_shard_nb_prep sorted it), or null
                  bool positions = false;                        // mode 2: xchg holds position records (device_shard_nb_positions), not neighbour slots: the streaming back end
                  bool pbytes = false; };                        // mode 2: a presence byte per neighbour behind the slots (device_shard_nb_presence): the wave-per-query back end asks only for fragments that exist
int device_predict(DeviceState* d, const FlatIndex& ix, const LaunchParams& p, bool buffers_on_device, void* stream,
                   // host-pointer mode: these are host buffers copied in/out by the call
                   const uint64_t* h_items, const uint32_t* h_qoff, uint64_t* h_ids, double* h_scores,
                   uint32_t* h_counts, uint32_t* h_stats, uint32_t* h_nb_rank, uint32_t* h_nb_num, uint32_t* h_nb_cnt, const ExtLists* ext = nullptr,
                   bool reserve_only = false, bool blocking_wait = false);   // reserve_only (srn_index_reserve): size the call's workspace, enqueue nothing; blocking_wait: the latency path sleeps on an interrupt instead of spinning
struct ShardIO {
    void* cand; uint32_t* cand_cnt;                                       // A out: [nq * m] packed slots, [nq]
    const void* gathered; const uint32_t* gathered_cnt; uint32_t n_shards;   // B in: [G][nq * gathered_stride], [G][nq]
    uint32_t gathered_stride;                                                // entries per query and shard in `gathered` (0 = m)
    void* nb; uint32_t* nb_cnt; int* minpos;                             // B out / C in: [nq * k], [nq], [nq * (k + 1)]
};

bool device_shard_lists_supported(DeviceState* d, const FlatIndex& ix, const LaunchParams& p);
bool device_fast_eligible(DeviceState* d, const FlatIndex& ix, const LaunchParams& p, uint64_t max_row_len_all = 0);   // max_row_len_all != 0: from rank-invariant inputs only (srn_group.hip)
bool device_has_packed_rows(const DeviceState* d);
// the shard's fragments once more, in the posting order of `post`'s lists (the streaming form of the wave-per-query back end); optional: no room, or not a shard of that
// kernel's geometry -> SRN_OK and the gather form runs.  post == nullptr: drop them.
int device_sback_attach_postings(DeviceState* d, DeviceState* post, uint64_t n_postings);
bool device_sback_streams(const DeviceState* d);
bool device_sback_wanted(const DeviceState* d);   // this shard has the wave-per-query back end's rows and the streaming form is not switched off: set_postings expects the copy
uint64_t device_sback_launches(const DeviceState* d);
// neighbours pipeline (replicated postings `post`: the whole index's dictionary + lists): prep records of ALL queries (lists looked up in `post`, dense idx in this shard's
// table), then the front end over the queries [q_lo, q_hi) -> xchg, or the back end over all of them <- xchg
int device_shard_nb_prep(DeviceState* d, DeviceState* post, const LaunchParams& p, char* records, void* stream, char** order_buf = nullptr, size_t* order_bytes = nullptr,
                         const unsigned long long** order_out = nullptr);   // order_buf (grow-only, the caller's): the batch's serving order is sorted into it, *order_out = where (null: batch too small)
int device_shard_nb_front(DeviceState* d, const FlatIndex& ix, DeviceState* post, const LaunchParams& p, const char* records, uint32_t* xchg, uint32_t xchg_stride, uint32_t q_lo, uint32_t q_hi, void* stream);
int device_shard_nb_back(DeviceState* d, const FlatIndex& ix, DeviceState* post, const LaunchParams& p, const char* records, uint32_t* xchg, uint32_t xchg_stride, void* stream,
                         const unsigned long long* order = nullptr, bool positions = false, bool pbytes = false);
int device_shard_nb_positions(DeviceState* d, const FlatIndex& ix, DeviceState* post, const LaunchParams& p, const char* records, const uint32_t* xin, uint32_t in_stride, uint32_t* xout, uint32_t out_stride,
                              uint32_t q_lo, uint32_t q_hi, void* stream);
uint32_t device_shard_nb_positions_stride(const LaunchParams& p);
// the neighbours' presence bytes (srn_sback.hip): a shard's presence bitmap (device words, or null), and the fronting rank's pass that writes the bytes behind the slots of [q_lo, q_hi)
const uint32_t* device_sback_present(const DeviceState* d, size_t* words);
int device_shard_nb_presence(DeviceState* d, const FlatIndex& ix, const LaunchParams& p, const char* records, uint32_t* xchg, uint32_t stride, const uint8_t* pbytes, uint32_t q_lo, uint32_t q_hi, void* stream);
bool device_shard_nb_presence_wanted();   // (knob; rank-invariant)   // 0: no streaming form for this batch shape / these knobs (rank-invariant)
uint32_t device_prep_stride(uint32_t max_len);
int device_shard_lists_head(DeviceState* d, const LaunchParams& p, void* pos, int* head, void* stream);
int device_shard_lists_count(DeviceState* d, const LaunchParams& p, const void* pos, const int* head, uint32_t* kept, int* tot, void* stream);
int device_shard_lists_copy(DeviceState* d, const LaunchParams& p, const void* pos, const uint32_t* kept, const long long* off, uint32_t* out, void* stream);
int device_shard_lists_predict(DeviceState* d, const FlatIndex& ix, const LaunchParams& p, uint32_t n_shards, const uint32_t* kept_g, const long long* off_g,
                               unsigned long long shard_stride, const uint32_t* lists_g, const int* head, const void* pos_local, char* records, void* stream,
                               const unsigned long long* shard_base = nullptr, bool direct = false);   // shard_base [n_shards] (device): the shards' segment starts (null: g * shard_stride); direct: a group of one shard reads its lists in place (lists_g = null)
int device_shard_stage(DeviceState* d, const FlatIndex& ix, int stage, const LaunchParams& p, const ShardIO& sh, void* stream);
int device_slot_bytes(DeviceState* d, const FlatIndex& ix, uint32_t max_len, uint32_t* num_bits = nullptr);
// ---- the persistent latency path (round 6): resident workgroups that serve srn_predict's call shape without a launch ----
// lanes: resident workgroups of the lean form (sessions of <= 4 items); max_items >= 5: as many of the MID form (5..10 items) beside them; idle_ms: a resident workgroup
// leaves by itself after that long without a request (it is started again by the next call that wants it)
int device_serve_start(DeviceState* d, const FlatIndex& ix, uint32_t k, uint32_t m, uint32_t how_many, uint32_t flags, uint32_t lanes, uint32_t max_items, uint32_t idle_ms);
int device_serve_stop(DeviceState* d);
// 0: served, *out_n rows written; 1: not served (no free lane, another configuration, a session the fused form hands on): the caller takes the launch path
int device_serve_predict(DeviceState* d, const uint64_t* items, uint32_t len, uint32_t k, uint32_t m, uint32_t how_many, uint32_t flags, uint64_t* out_ids, double* out_scores, size_t* out_n);
int device_serve_last_stamps(DeviceState* d, uint32_t* out4);
int device_serve_stats(DeviceState* d, uint64_t* served, uint64_t* not_served, uint64_t* launches, uint32_t* lanes);
int device_last_kernel_ms(DeviceState* d, double* ms_main, double* ms_retry, uint32_t* retried);
int device_phase_cycles(DeviceState* d, int enable, unsigned long long* out16);
int device_kernel_timing(DeviceState* d, int enable);
int device_reserve(DeviceState* d, const FlatIndex& ix, const LaunchParams& p, void* stream);
int device_last_path_counts(DeviceState* d, uint32_t* nq, uint32_t* general, uint32_t* global_pass);   // debug profiling aid
int device_last_mid_count(DeviceState* d, uint32_t* listed, uint32_t* big_listed = nullptr);   // queries the last call's lean fast kernel listed for its MID instantiation
int device_kernel_times(DeviceState* d, uint32_t max_n, double* ms_main, double* ms_retry, uint32_t* out_n, double* ms_prep = nullptr, double* ms_fast = nullptr);

}  // namespace srn

namespace srn {
// concurrent srn_predict calls on one handle combine into rounds (srn_combine.cpp)
struct Combiner;
Combiner* combiner_create();
void combiner_free(Combiner* c);
void combiner_stats(const Combiner* c, uint64_t* rounds, uint64_t* requests, uint64_t* max_round);
int combiner_predict(Combiner* c, const srn_index* idx, const uint64_t* evolving, size_t len, size_t k, size_t m, size_t how_many, unsigned flags,
                     uint64_t* out_ids, double* out_scores, size_t* out_n, int lanes, size_t round_cap);
int knob_predict_lanes();   // SRN_PREDICT_LANES (0 = no combining: every call runs by itself, as in round 2)
size_t knob_tiny_max();     // SRN_TINY_MAX: largest host batch on the zero-copy latency path = largest round
}  // namespace srn

struct srn_index {
    srn::FlatIndex flat;
    srn::DeviceState* dev = nullptr;
    int device = -1;
    srn::Combiner* comb = nullptr;   // created with the device state
    // reference session index -> recency rank (kNone: not a kept session), built at the first srn_index_items_for_session call (the index is immutable: call_once)
    mutable std::vector<uint32_t> session_to_rank; mutable std::once_flag s2r_once;
};
struct srn_sessions {
    srn::Sessions s;
};

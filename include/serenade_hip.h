/* =====================================================================================
 * serenade_hip.h -- C ABI of libserenade_hip.so: the MI355X (gfx950) implementation of
 * Serenade's VMIS-kNN `predict_next` hot path.
 *
 * The reference (bolcom/serenade @ 2024-11-08, pure Rust) has no FFI of its own; its boundary for
 * this path is the Rust signature
 *
 *   vmisknn::predict<I: SimilarityComputationNew + Send + Sync>(index: &I,
 *       evolving_session: &[u64], k: usize, m: usize, how_many: usize,
 *       enable_business_logic: bool) -> BinaryHeap<ItemScore>        src/vmisknn/mod.rs:118-125
 *
 * plus the index constructors VMISIndex::new_from_csv (src/vmisknn/vmis_index.rs:38-83) and the
 * trait accessors (src/vmisknn/similarity_indexed.rs:8-24).  The entry points below are what a Rust
 * `extern "C"` block in the reference would bind to swap that path for the GPU one (INTEGRATION.md
 * shows the binding).  Plain pointers and sizes only; every function returns 0 or a negative
 * SRN_E* code and never unwinds across the boundary; srn_last_error() gives a thread-local
 * message for the last failure on the calling thread.
 *
 * There is NO CPU fallback behind this ABI: predict calls on an index without an attached device
 * fail with SRN_ENODEV.
 *
 * Result order: every predict entry point writes results score-descending (ties: ascending item
 * id), i.e. already in the `into_sorted_vec()` order all reference callers use
 * (src/bin/evaluator.rs:67-71, src/endpoints/recommend_resource.rs:58-62).
 * ===================================================================================== */
#ifndef SERENADE_HIP_H
#define SERENADE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SRN_OK 0
#define SRN_EINVAL (-1) /* null pointer, empty evolving session (reference panics: mod.rs:157), bad CSR */
#define SRN_ENOMEM (-2)
#define SRN_EHIP (-3)    /* a HIP runtime call failed; see srn_last_error() */
#define SRN_ERANGE (-4)  /* k / m / how_many / session length above the compiled kernel limits */
#define SRN_EIO (-5)
#define SRN_ENODEV (-6)  /* index has no device attached (built with device < 0) or no GPU present */
#define SRN_ESTATE (-7)  /* the handle is unusable since an earlier failure (a shard group after a failed batch): free it */
#define SRN_ETIMEOUT (-8) /* srn_shard_group_wait: the batch did not finish in time */

/* compiled kernel limits (srn_limits() reports the same numbers at run time) */
#define SRN_MAX_HOW_MANY 512
#define SRN_MAX_SESSION_LEN 255
#define SRN_MAX_K 8192

/* attribute flags, one byte per item (ProductAttributes, src/vmisknn/vmis_index.rs:23-26) */
#define SRN_ATTR_ADULT 1u
#define SRN_ATTR_FOR_SALE 2u
#define SRN_ATTR_NONE 0xFFu /* item has no attributes (find_attributes() == None, vmis_index.rs:417-419) */

/* predict flags */
#define SRN_FLAG_BUSINESS_LOGIC 1u /* enable_business_logic = true (mod.rs:162-182) */

typedef struct srn_index srn_index_t;
typedef struct srn_sessions srn_sessions_t;

/* Training sessions in the form prepare_hashmap() consumes (src/vmisknn/vmis_index.rs:422-427):
 * session i owns items[sess_off[i] .. sess_off[i+1]) (ascending, de-duplicated item ids) and
 * max_ts[i]; the session's index i is the reference's dense session id. */
typedef struct {
    const uint64_t* sess_off; /* [n_sessions + 1] */
    const uint64_t* items;    /* [sess_off[n_sessions]] */
    const uint32_t* max_ts;   /* [n_sessions] */
    size_t n_sessions;
} srn_sessions_view_t;

typedef struct {
    uint64_t n_items;          /* distinct items in kept sessions */
    uint64_t n_sessions_total; /* sessions handed to the builder */
    uint64_t n_sessions_kept;  /* sessions with len <= max_session_len (vmis_index.rs:452) */
    uint64_t nnz_rows;         /* (session,item) pairs of kept sessions = idf numerator (vmis_index.rs:509) */
    uint64_t nnz_postings;     /* posting entries after truncation to m_index (vmis_index.rs:504) */
    uint64_t m_index;
    uint64_t max_session_len;
    uint64_t max_row_len;      /* longest kept row */
    uint64_t device_bytes;     /* HBM held by the index (0 if host-only) */
    int32_t device;            /* HIP device ordinal, -1 = host-only */
    int32_t offsets_64bit;     /* row offsets stored as u64 (nnz_rows >= 2^32) */
    double idf_weighting;
    uint64_t incomplete_items; /* (round 6) items of a PRE-BUILT index (srn_index_new_from_avro) whose posting list is not "the most recent sessions that hold the item" under any
                                * tie order the loader could infer: queries such an item can affect are served by the general kernel's row pass (lists as given,
                                * vmis_index.rs:201-228), everything else by the fast kernels.  0 for every index built by this library */
} srn_index_info_t;

typedef struct {
    uint32_t max_how_many, max_session_len, max_k, reserved;
} srn_limits_t;

/* ---- training data ---------------------------------------------------------------------- */

/* TSV "SessionId\tItemId\tTime" -> sessions, with the loader semantics of read_from_file()
 * (src/vmisknn/vmis_index.rs:591-686): stable order inside a session, first-occurrence de-dup then
 * ascending sort, max-timestamp updated on non-duplicate rows only, the final row closes the current
 * session without being added and the last session is dropped. */
int srn_sessions_from_tsv(const char* path, srn_sessions_t** out);
int srn_sessions_view(const srn_sessions_t* s, srn_sessions_view_t* out);
/* exact q-quantile of the session lengths (linear interpolation, rounded): the default stand-in for
 * the reference's t-digest estimate qty_events_p99_5 (vmis_index.rs:689-716, used at :67). */
int srn_sessions_length_quantile(const srn_sessions_t* s, double q, uint64_t* out);
void srn_sessions_free(srn_sessions_t* s);

/* ---- index ------------------------------------------------------------------------------ */

/* prepare_hashmap() (src/vmisknn/vmis_index.rs:422-528) into the flat HBM layout:
 * sessions longer than max_session_len are left out, posting lists are most-recent-first (ties:
 * larger session index first) and truncated to m_index, idf = ln(pairs / sessions_with_item) *
 * idf_weighting, every item {for_sale, not adult}.  device >= 0 uploads to that GPU; device < 0
 * keeps a host-only index (build / save / inspect; predict then fails with SRN_ENODEV). */
int srn_index_build(const srn_sessions_view_t* sessions, size_t m_index, size_t max_session_len,
                    double idf_weighting, int device, srn_index_t** out);
/* The same index built on the GPU (rocPRIM radix sorts instead of host loops; < 2^32 sessions and interactions).
 * Bit-identical to srn_index_build(); 582 M interactions take seconds instead of ~45 s on one host core. */
int srn_index_build_gpu(const srn_sessions_view_t* sessions, size_t m_index, size_t max_session_len,
                        double idf_weighting, int device, srn_index_t** out);
/* VMISIndex::new_from_csv(path, m_most_recent_sessions, idf_weighting) (vmis_index.rs:38-83);
 * max_session_len = 0 selects the exact p99.5 of the session lengths. */
int srn_index_new_from_csv(const char* path, size_t m_most_recent_sessions, double idf_weighting,
                           size_t max_session_len, int device, srn_index_t** out);
/* VMISIndex::new(base_path) (src/vmisknn/vmis_index.rs:85-314): the pre-built production index, two directories of Avro
 * object-container files (codec null or snappy):
 *   <base>/itemindex/[files].avro     {ItemId: long, session_indices_time_ordered: array<int>, idf: double, ForSale, IsAdult: boolean}
 *   <base>/sessionindex/[files].avro  {SessionIndex: int, item_ids_asc: array<long>, Time: int}
 * Posting lists, idf and the product flags are taken from the files as they are (nothing is recomputed); the lists must be
 * what their name says -- each item's most recent sessions -- or the call fails with SRN_EINVAL.  device < 0: host only. */
int srn_index_new_from_avro(const char* base_path, int device, srn_index_t** out);
int srn_index_save(const srn_index_t* idx, const char* path);
int srn_index_load(const char* path, int device, srn_index_t** out);
/* replace ProductAttributes of the given items (the Avro path carries real flags, vmis_index.rs:223-228) */
int srn_index_set_attributes(srn_index_t* idx, const uint64_t* item_ids, const uint8_t* flags, size_t n);
int srn_index_info(const srn_index_t* idx, srn_index_info_t* out);
/* posting list of one item as reference session indices (most recent first) and its idf;
 * *out_len = -1 if the item is unknown.  For index-parity tests and debugging. */
int srn_index_postings(const srn_index_t* idx, uint64_t item_id, uint32_t* out_sessions, size_t cap,
                       int64_t* out_len, double* out_idf);
/* The other accessors of the trait predict() is generic over (SimilarityComputationNew, src/vmisknn/similarity_indexed.rs:9-23) -- with srn_index_postings' idf they
 * are what `impl SimilarityComputationNew for HipVMISIndex` needs:
 *   items_for_session(&u32) -> &[u64]   (vmis_index.rs:317-319)  the row of a session by its reference index, ascending item ids; *out_len = its length (items beyond
 *                                        `cap` are not written).  Rows are kept for the sessions that can be neighbours (len <= max_session_len, vmis_index.rs:452); any
 *                                        other session index: SRN_ERANGE (the reference keeps those rows too, :79, but never reads them on this path)
 *   find_attributes(&u64) -> Option<&ProductAttributes>   (vmis_index.rs:417-419)  *out_flags = SRN_ATTR_* bits, SRN_ATTR_NONE for None (unknown item or no attributes)
 *   find_neighbors(&[u64], k, m) -> BinaryHeap<SessionScore>   (vmis_index.rs:325-415)  the neighbour sessions as (reference session index, similarity = numerator / U),
 *                                        best first (similarity desc, ties: more recent first -- the reference's heap order is unspecified); room for k entries; runs on
 *                                        the GPU, canonical semantics as everywhere */
int srn_index_items_for_session(const srn_index_t* idx, uint32_t session, uint64_t* out_items, size_t cap, size_t* out_len);
int srn_index_find_attributes(const srn_index_t* idx, uint64_t item_id, uint8_t* out_flags);
/* The recency order this index serves with -- the total order behind "most recent" in find_neighbors (session_to_max_time_stamp, vmis_index.rs:33, 358-383, 404-410):
 * out_rank[s] = the recency rank of reference session s (0 = oldest; ascending max timestamp; sessions of EQUAL timestamp, which the reference leaves to its containers,
 * by SessionIndex -- or, for a pre-built Avro index, in the order its producer's list cuts imply), 0xFFFFFFFF for a session that is in no posting list and keeps no row.
 * Room for srn_index_info().n_sessions_total entries.  What a checker needs to state the canonical answer for an index with tied timestamps. */
int srn_index_session_recency(const srn_index_t* idx, uint32_t* out_rank, size_t cap);
int srn_find_neighbors(const srn_index_t* idx, const uint64_t* evolving, size_t len, size_t k, size_t m,
                       uint32_t* out_sessions, double* out_scores, size_t* out_n);
void srn_index_free(srn_index_t* idx);

/* ---- predict ---------------------------------------------------------------------------- */

/* One evolving session (oldest item first, like the reference slice).  out_ids / out_scores have
 * room for how_many entries; *out_n receives the number written.  Unknown items give *out_n = 0
 * with SRN_OK (vmis_index.rs:350); an empty session is SRN_EINVAL (the reference panics). */
int srn_predict(const srn_index_t* idx, const uint64_t* evolving, size_t len, size_t k, size_t m,
                size_t how_many, int enable_business_logic, uint64_t* out_ids, double* out_scores,
                size_t* out_n);
/* ---- the persistent latency path (round 6) ------------------------------------------------------------------------------------------------------------
 * srn_predict is the reference's call shape (one evolving session per call: src/endpoints/recommend_resource.rs:56, src/bin/evaluator.rs:58).  On the one-launch form a
 * call costs ~28 us from a C++ host, of which ~18 us are the GPU's work; the rest is queue submission and completion signalling.  srn_index_serve_start parks `lanes`
 * (<= 256) RESIDENT workgroups on the index's GPU (a CU each, ONE launch on a stream of the lowest priority; with max_items_in_session > 4 a second launch of as many
 * workgroups of the form that serves sessions of 5..10 items; sessions of more items always take the launch path): a call of
 * srn_predict with the same k / m / how_many / business-logic flag then posts its session to a free one through pinned memory and spins on the answer -- no launch.
 * A session the resident form cannot finish (0.3-3 % of config 3's: merged lists beyond its layout, > 63 scored candidates of a small query), a call with other
 * parameters, and a call that finds every resident workgroup taken run exactly as without this: the rows are the same bytes either way.
 *   idle_ms: a resident workgroup leaves by itself after that long without a request (a host that died leaves nothing behind); the next call that wants it starts it again.
 *   Anything in this library that frees device memory or synchronises the device makes the resident workgroups leave first (they come back on demand): allocate and
 *   reserve (srn_index_reserve) BEFORE serving.  APPLICATION code that calls hipFree / hipDeviceSynchronize on this device while workgroups are resident waits until they
 *   leave (idle_ms at most): call srn_index_serve_stop first.
 *   Measured on config 3 (C++ host, profiles/r06_latency_resident_cfg3.json): p50 / p90 22.6 / 27.4 us against 27.8 / 32.7 on the launch path; 16 callers on 16 resident
 *   workgroups 23.7 / 30.0 us and 655 K requests/s against 62.8 / 86.6 us and 246 K.  srn_index_serve_stop may be called while other threads are inside srn_predict.
 * srn_index_serve_stats: sessions answered by a resident workgroup | sent to the launch path | kernel launches so far (1 per form + restarts) | resident workgroups. */
int srn_index_serve_start(srn_index_t* idx, size_t k, size_t m, size_t how_many, int enable_business_logic, unsigned lanes, unsigned max_items_in_session, unsigned idle_ms);
int srn_index_serve_stop(srn_index_t* idx);
int srn_index_serve_stats(const srn_index_t* idx, uint64_t* out_served, uint64_t* out_not_served, uint64_t* out_launches, uint32_t* out_lanes);

/* srn_predict is re-entrant on one handle from any number of host threads, like predict() on the Arc<VMISIndex> the actix workers
 * share (src/bin/serving.rs:62-94).  Concurrent calls with the same (k, m, how_many, business flag) COMBINE into rounds: one batch
 * launch serves every call that arrived while the previous rounds ran (a lone caller runs alone, at once; environment
 * SRN_PREDICT_LANES = rounds in flight side by side, default 4, 0 = never combine).  Counters since the handle was created: */
int srn_predict_stats(const srn_index_t* idx, uint64_t* out_rounds, uint64_t* out_requests, uint64_t* out_max_round);

/* nq sessions in CSR form: session q = items_flat[q_off[q] .. q_off[q+1]).  Host pointers.
 * out_ids / out_scores are [nq * how_many] (row q at q * how_many), out_counts [nq].
 * Up to 256 sessions take the zero-copy latency path (pinned, device-mapped staging, no copies: ONE launch for calls of <= 48 sessions -- a workgroup per session writes
 * its query's record, serves it and finishes its row; the fast kernel's launch sequence for larger calls with a session of > 8 items; prep + general kernel -- two
 * launches -- otherwise); larger batches are cut into
 * chunks whose uploads, kernels and downloads overlap on separate streams, the results landing in the caller's buffers while
 * the next chunks run (srn_hostpipe.hip).  The buffers may be pageable. */
int srn_predict_batch(const srn_index_t* idx, const uint64_t* items_flat, const uint32_t* q_off, size_t nq,
                      size_t k, size_t m, size_t how_many, unsigned flags, uint64_t* out_ids,
                      double* out_scores, uint32_t* out_counts);

/* Same with every buffer already resident in the index's device memory, enqueued on `stream`
 * (a hipStream_t; NULL = the null stream) without host synchronisation.  max_len_hint must be
 * >= the longest session in the batch (<= SRN_MAX_SESSION_LEN).  Calls from several host threads on the SAME stream take turns inside the library (each call's
 * launches stay contiguous in the stream: they share the workspace bound to it); different streams run side by side.  A query the kernel cannot serve
 * (empty or over-long session) gets out_counts[q] = 0xFFFFFFFF.  flags: SRN_FLAG_BUSINESS_LOGIC, SRN_FLAG_INPUTS_RESIDENT
 * (the prep kernel of this call then runs beside the previous call's kernels; for back-to-back full batches that was measured as a LOSS -- the prep kernel's
 * 0.44 ms per 2^20 queries disappear but the concurrent random look-ups slow vmis_fast_kernel by 0.7 ms -- it pays where the previous call leaves the GPU idle). */
int srn_predict_batch_device(const srn_index_t* idx, const uint64_t* d_items_flat, const uint32_t* d_q_off,
                             size_t nq, size_t max_len_hint, size_t k, size_t m, size_t how_many,
                             unsigned flags, uint64_t* d_out_ids, double* d_out_scores,
                             uint32_t* d_out_counts, void* stream);

/* Debug / measurement variant of srn_predict_batch (host pointers; any of the three extra outputs
 * may be NULL):
 *   out_stats  [nq * 8]  P, C, K, I, D, H, L, status per query -- the per-query terms of the
 *                        algorithmic-bytes formula
 *   out_nb_sessions / out_nb_num [nq * k], out_nb_counts [nq]: the selected neighbour sessions
 *                        (reference session indices, unordered) and their integer similarity
 *                        numerators (similarity = num / distinct evolving items). */
int srn_predict_batch_debug(const srn_index_t* idx, const uint64_t* items_flat, const uint32_t* q_off, size_t nq,
                            size_t k, size_t m, size_t how_many, unsigned flags, uint64_t* out_ids,
                            double* out_scores, uint32_t* out_counts, uint32_t* out_stats,
                            uint32_t* out_nb_sessions, uint32_t* out_nb_num, uint32_t* out_nb_counts);

/* Sizes the per-stream workspace of srn_predict_batch_device for calls of up to nq queries with these parameters: after it, such
 * calls on `stream` allocate nothing (hipMalloc / hipFree synchronise the device, so a serving process reserves once at start-up).
 * Without it the workspace grows on demand, the first time a larger batch arrives. */
int srn_index_reserve(const srn_index_t* idx, size_t nq, size_t max_len_hint, size_t k, size_t m, size_t how_many, unsigned flags, void* stream);

/* Kernel timing on (enable != 0) or off, per index; off by default (SRN_TIMING=1 in the environment: on from the start).  When on, every batch call records HIP
 * events between its kernels, which srn_last_kernel_ms / srn_kernel_times* read back; each event idles the launch stream for ~6 us (three per call: 18 us of the
 * 250 us a 4 096-query batch takes), so a serving process leaves it off and a benchmark switches it on around the region it reports kernel times for. */
int srn_kernel_timing(srn_index_t* idx, int enable);

/* Average duration in milliseconds of the predict kernel launches enqueued by the most recent
 * srn_predict_batch* call on this thread, measured with HIP events on the launch stream (blocks
 * until that work has finished); *out_launches = number of kernel launches it covered.  SRN_EINVAL if kernel timing
 * was off for that call (srn_kernel_timing). */
int srn_last_kernel_ms(const srn_index_t* idx, double* out_ms_main, double* out_ms_retry, uint32_t* out_retried);

/* ---- item-sharded index: one shard per GPU ---------------------------------------------------
 * For indices that outgrow one GPU (BASELINE config 5), and for the north star's multi-GPU mode, the index is split by item: shard g holds postings, row
 * fragments and idf of the items with owner(id) == g (a fixed hash of the public id), while session recency ranks stay global.  A shard answers no predict
 * call by itself: a SHARD GROUP (below) drives the shards of all ranks through one batch call.  Results are bit-identical to the unsharded path
 * (tests/test_gpu_shard_group.py, tests/test_gpu_sharded.py).
 * Shard `shard` of n_shards is cut out of an UNSHARDED index -- built by srn_index_build_gpu or read by srn_index_load; the same bytes
 * srn_index_build_shard builds from the sessions, in one O(nnz) pass (an item-sharded deployment builds or loads ONE index and
 * every rank cuts its own shard out of it; the reference loads its production index: src/vmisknn/vmis_index.rs:85-314).
 * srn_index_build_shard_gpu = srn_index_build_gpu + srn_index_shard without ever attaching the full index to the device;
 * srn_index_load_shard reads a saved index (unsharded: cut; already that shard: as is). */
int srn_index_build_shard(const srn_sessions_view_t* sessions, size_t m_index, size_t max_session_len,
                          double idf_weighting, uint32_t shard, uint32_t n_shards, int device, srn_index_t** out);
int srn_index_shard(const srn_index_t* full, uint32_t shard, uint32_t n_shards, int device, srn_index_t** out);
int srn_index_build_shard_gpu(const srn_sessions_view_t* sessions, size_t m_index, size_t max_session_len, double idf_weighting,
                              uint32_t shard, uint32_t n_shards, int device, srn_index_t** out);
int srn_index_load_shard(const char* path, uint32_t shard, uint32_t n_shards, int device, srn_index_t** out);

/* ---- the shard group: the whole item-sharded batch in ONE call, collectives inside (srn_group.hip) ---------------------------------------
 * What a reference-side host (serving / evaluator, src/endpoints/recommend_resource.rs:56, src/bin/evaluator.rs:58) calls to drive an index
 * sharded over the GPUs of a node: one process (or thread group) per GPU creates its rank of the group around its shard, then every rank calls
 * srn_shard_group_predict_batch with the SAME batch; the call runs the LISTS pipeline -- head kernel, all-reduce(max) of the cuts, count kernel, all-gather of the
 * counts, variable-length exchange of the kept list prefixes, the unsharded kernels over the rank's row fragments, all-gather of the per-shard
 * top-n, merge by (score desc, id asc) -- with RCCL called from inside the library.  Results (identical on every rank, bit-identical to the
 * unsharded index) are written to the caller's device buffers, asynchronously on `stream`; the call itself blocks the host for one short
 * synchronisation on the group's own exchange stream (the per-shard list totals size the exchange).  With SRN_FLAG_INPUTS_RESIDENT a batch's
 * exchange phase overlaps the previous batch's kernels.  Batches the lists pipeline does not serve (sessions of > 8 items,
 * m > m_index, incomplete posting lists) take the three-stage pipeline inside the same call: stage A -> all-gather of the candidates -> stage B ->
 * all-reduce(min) of the first-match positions -> stage C -> all-gather of the per-shard top-n -> merge (no host synchronisation at all).
 *
 *   rank 0:   srn_shard_group_unique_id(id, sizeof id)      -- 256 opaque bytes; hand them to the other ranks (your own control plane)
 *   rank r:   srn_index_load_shard(path, r, world, device_r, &shard);  srn_shard_group_create(shard, id, r, world, &group)
 *   per batch, every rank:   srn_shard_group_predict_batch(group, d_items, d_q_off, nq, ...)
 */
#define SRN_SHARD_GROUP_ID_BYTES 256
#define SRN_FLAG_INPUTS_RESIDENT 2u /* srn_shard_group_predict_batch, srn_predict_batch_device: d_items_flat / d_q_off are complete in device memory at call time (not pending on `stream`):
                                     * the call's first small kernels may then run on a stream of the library's own, beside the previous call's kernels */
typedef struct srn_shard_group srn_shard_group_t;
int srn_shard_group_unique_id(void* out, size_t bytes);
int srn_shard_group_create(const srn_index_t* shard, const void* unique_id, int rank, int world, srn_shard_group_t** out);
/* The same group over the APPLICATION's transport instead of RCCL (MPI, sockets, a test harness): three collectives on device buffers.  Each
 * callback must order itself after the work already enqueued on `stream` and complete (or be enqueued on `stream`) before it returns;
 * channel 0 / 1 = the group's two independent sequences of collectives (exchange stream / caller's stream).  Non-zero return = failure.
 *   all_reduce_max_i32 / all_reduce_min_i32(user, channel, d_buf, count, stream)   element-wise maximum / minimum over the ranks, in place
 *   all_gather(user, channel, d_buf, block_bytes, stream)                   d_buf = world blocks; block `rank` holds this rank's data
 *   all_gather_v(user, channel, d_buf, byte_off[world], byte_cnt[world], stream)    segment r = d_buf[byte_off[r] .. + byte_cnt[r]); this rank's is in place */
typedef struct {
    void* user;
    int (*all_reduce_max_i32)(void* user, int channel, int32_t* d_buf, size_t count, void* stream);
    int (*all_gather)(void* user, int channel, void* d_buf, size_t block_bytes, void* stream);
    int (*all_gather_v)(void* user, int channel, void* d_buf, const uint64_t* byte_off, const uint64_t* byte_cnt, void* stream);
    int (*all_reduce_min_i32)(void* user, int channel, int32_t* d_buf, size_t count, void* stream);   /* may be NULL: the group then serves the lists pipeline only */
} srn_shard_comm_t;
int srn_shard_group_create_with_comm(const srn_index_t* shard, int rank, int world, const srn_shard_comm_t* comm, srn_shard_group_t** out);
/* All shards of the group in THIS process on one device (tests; capacity experiments on one GPU): the collectives degenerate to kernels. */
int srn_shard_group_create_local(const srn_index_t* const* shards, int n_shards, srn_shard_group_t** out);
/* Threads and streams: calls on one group are serialised inside (a mutex for the enqueue; a buffer slot is reused only behind the batch that used it, whatever stream
 * that batch ran on).  Over a multi-rank transport every rank must issue the group's batches in the SAME order, and a rank that alternates between streams must
 * order them itself: collectives on one communicator execute in issue order on every rank.  One stream per group is the simple way. */
int srn_shard_group_predict_batch(srn_shard_group_t* g, const uint64_t* d_items_flat, const uint32_t* d_q_off, size_t nq, size_t max_len_hint,
                                  size_t k, size_t m, size_t how_many, unsigned flags, uint64_t* d_out_ids, double* d_out_scores,
                                  uint32_t* d_out_counts, void* stream);
typedef struct {
    uint64_t n_shards, batches, queries;
    uint64_t bytes_head, bytes_counts, bytes_lists, bytes_results; /* what THIS rank contributed to each exchange, summed over the batches */
    uint64_t bytes_lists_max_rank;                                 /* the fullest rank's list segment, summed over the batches (what a padded all-gather would ship per rank) */
    uint32_t transport;                                            /* 0 in-process, 1 RCCL, 2 application callbacks */
    uint32_t overlapped;                                           /* the exchange phase runs on the group's own stream */
    uint64_t stage_batches, bytes_stage_candidates, bytes_stage_minpos; /* batches that took the three-stage pipeline, and what this rank contributed to its two big exchanges */
    uint64_t neighbour_batches, bytes_neighbours;                      /* batches that took the neighbours pipeline, and the neighbour-list block this rank contributed to its all-gather */
} srn_shard_group_stats_t;
int srn_shard_group_stats(const srn_shard_group_t* g, srn_shard_group_stats_t* out);
/* Overlap of batch i + 1's exchange (the group's own stream and first communicator) with batch i's kernels and result gather (caller's stream, second communicator).
 * on = 0: everything in issue order on the caller's stream, one collective of the group in flight at a time -- the conservative form (two communicators with collectives in
 * flight at once need both collective kernels to become resident on every rank; they do here -- the predict kernels are finite -- but a host that wants no such
 * dependence leaves it off).  Default OFF since round 5 -- the overlapped form has never run on more than one GPU; a host opts in (or SRN_GROUP_OVERLAP=1 in the environment);
 * takes effect from the next batch; every rank must use the same setting for the same batch. */
int srn_shard_group_set_overlap(srn_shard_group_t* g, int on);
/* Failure of a batch.  A srn_shard_group_predict_batch that fails after its first collective was handed to the transport (or for a reason its peers do not share: a HIP
 * error, an allocation) leaves the peers' collectives of that batch without their partner.  The group is then BROKEN on this rank: its RCCL communicators are aborted at
 * once (nothing of this rank keeps a peer's GPU waiting), and every further call on it fails with SRN_ESTATE.  The peers learn of it from their own transport (a callback
 * that returns non-zero breaks their group the same way) or from srn_shard_group_wait: a bounded wait for the group's most recent batch -- SRN_OK when its results are
 * complete, SRN_ETIMEOUT after timeout_ms, which also breaks the group (a collective whose partner died never ends; a plain stream synchronise would hang with it).
 * A broken group is freed without waiting for the device.  Refusals every rank makes alike before anything is issued (SRN_EINVAL / SRN_ERANGE) leave the group usable. */
int srn_shard_group_wait(srn_shard_group_t* g, uint64_t timeout_ms);
/* The NEIGHBOURS pipeline (round 4; SURVEY 8(e)'s replicated-postings variant, with the candidate work divided over the ranks).  The posting lists are the pruned structure
 * (<= m_index entries per item: config 3 111 MB, config 5 ~9 GB) -- every rank keeps ALL of them beside its shard of the rows: `postings` = the unsharded index itself, or
 * its rows-free view (srn_index_postings_view: dictionary, idf / attributes, lists), on the group's device.  Per batch, rank r then runs find_neighbors
 * (vmis_index.rs:325-415: lists, merge, m-cut, k-cut) for the queries [r nq / G, (r + 1) nq / G) ONLY, one all-gather ships the neighbour lists ((k + 1) * 4 bytes per
 * query, fixed-size blocks: no host synchronisation), and every rank scores ALL queries over its own row fragments (mod.rs:126-214) before the usual all-gather of the
 * per-shard top-n and the merge.  Same integers as the unsharded index on every rank: bit-identical results.  Serves what the lists pipeline serves and the fast kernel's
 * shape (k <= 1536, m <= 2560, how_many <= 24, no debug outputs); other batches take the lists / three-stage pipelines as before.  postings = NULL switches it off.
 * Every rank must make the same call; the handle must outlive the group. */
int srn_shard_group_set_postings(srn_shard_group_t* g, const srn_index_t* postings);
int srn_index_postings_view(const srn_index_t* full, int device, srn_index_t** out);
void srn_shard_group_free(srn_shard_group_t* g);

/* The same for the most recent min(max_n, 64) predict calls (oldest first): per-call duration in ms of
 * the main kernel and of the retry pass, from HIP events recorded on the launch stream around each
 * launch.  This is what bench.py reports as the kernel's live-measured launch duration.  Only the trailing run of calls
 * made with kernel timing on (srn_kernel_timing) is reported: *out_n = 0 if the last call was not timed. */
int srn_kernel_times(const srn_index_t* idx, uint32_t max_n, double* out_ms_main, double* out_ms_retry, uint32_t* out_n);

/* The same with the launches of a call told apart: prep kernel | fast kernel alone (vmis_fast_kernel, the dominant kernel;
 * 0 when the call was not eligible for it) | all predict launches (fast kernel + general kernel over the handed-over
 * queries + finish kernel: == out_ms_main of srn_kernel_times) | global-table retry pass.  Any output may be null. */
int srn_kernel_times_detail(const srn_index_t* idx, uint32_t max_n, double* out_ms_prep, double* out_ms_fast, double* out_ms_predict,
                            double* out_ms_retry, uint32_t* out_n);


/* How the queries of the most recent predict call on this handle were served: *out_nq queries in all, *out_general of them by
 * the general kernel (the fast kernel hands over what does not fit its query shape; == nq when the launch was not eligible for
 * the fast kernel at all), *out_global_pass through the global-table retry pass.  Waits for that call to finish. */
int srn_last_path_counts(const srn_index_t* idx, uint32_t* out_nq, uint32_t* out_general, uint32_t* out_global_pass);


/* ---- misc ------------------------------------------------------------------------------- */
/* ---- dynamic batching: the serving-side caller ------------------------------------------------------------------
 * The reference answers every /v1/recommend call with its own vmisknn::predict on an actix worker
 * (src/endpoints/recommend_resource.rs:56-62; workers share one Arc<VMISIndex>, src/bin/serving.rs:62-94).
 * srn_batcher_predict has predict()'s shape -- one evolving session in, <= how_many (id, score) pairs out, blocking --
 * and may be called from any number of threads; a dispatcher thread folds what is waiting (up to max_batch requests, or
 * whatever has arrived max_wait_us after the first one) into ONE srn_predict_batch launch.  k, m, how_many and the
 * business-logic switch are fixed per batcher, like the serving binary's configuration (src/config.rs). */
typedef struct srn_batcher srn_batcher_t;
int srn_batcher_create(const srn_index_t* idx, size_t max_batch, unsigned max_wait_us, size_t k, size_t m, size_t how_many,
                       int enable_business_logic, srn_batcher_t** out);
int srn_batcher_predict(srn_batcher_t* b, const uint64_t* evolving, size_t len, uint64_t* out_ids, double* out_scores, size_t* out_n);
int srn_batcher_stats(srn_batcher_t* b, uint64_t* n_requests, uint64_t* n_batches, uint64_t* max_batch_seen);
int srn_batcher_how_many(const srn_batcher_t* b, size_t* out);   /* the result capacity a caller's buffers need */
void srn_batcher_free(srn_batcher_t* b);   /* serves what is still queued, then stops the dispatcher */

/* ---- evolving-session store + the /v1/recommend handler body --------------------------------------------------------
 * Replaces RocksDBSessionStore (src/sessions/mod.rs:7-77) and the body of v1_recommend
 * (src/endpoints/recommend_resource.rs:20-65) between the web framework and predict.  The store is in memory (a visitor
 * is pinned to one pod by session_id affinity; nothing has to survive the process).  Keys are the reference's: the MD5
 * digest of the session_id string read as a big-endian u128 (recommend_resource.rs:27-28), here as (hi, lo).
 * A session idle for more than idle_secs reads as empty (mod.rs:46-52; 0 = the reference's 20 minutes); entries older than
 * ttl_secs are dropped (RocksDB TTL, src/bin/serving.rs:55-56; 0 = the reference's 30 minutes).  now_secs = 0 means the
 * system clock (seconds since the epoch, mod.rs:74-76); tests pass explicit times. */
typedef struct srn_session_store srn_session_store_t;
int srn_session_key(const char* session_id, size_t len, uint64_t* key_hi, uint64_t* key_lo);
int srn_session_store_create(uint64_t ttl_secs, uint64_t idle_secs, srn_session_store_t** out);
void srn_session_store_free(srn_session_store_t* s);
/* get_session_items (mod.rs:37-57): unknown or idle session -> *out_n = 0 */
int srn_session_store_get(srn_session_store_t* s, uint64_t key_hi, uint64_t key_lo, uint64_t now_secs,
                          uint64_t* out_items, size_t cap, size_t* out_n);
/* update_session_items (mod.rs:59-72) */
int srn_session_store_update(srn_session_store_t* s, uint64_t key_hi, uint64_t key_lo, uint64_t now_secs,
                             const uint64_t* items, size_t n);
/* drops every entry older than ttl_secs now (also done incrementally by updates); *n_live = entries kept */
int srn_session_store_sweep(srn_session_store_t* s, uint64_t now_secs, uint64_t* n_live);
/* v1_recommend's body: with user_consent, session := stored items; append item_id unless it repeats the last one; drop the
 * oldest beyond max_items_in_session; store; without consent the session is [item_id] and the store is untouched.  Then
 * predict through the batcher.  out_ids (and out_scores, may be NULL: the endpoint returns ids only) hold how_many entries. */
int srn_recommend(srn_batcher_t* b, srn_session_store_t* s, const char* session_id, size_t session_id_len, uint64_t item_id,
                  int user_consent, size_t max_items_in_session, uint64_t now_secs, uint64_t* out_ids, double* out_scores,
                  size_t* out_n);

int srn_device_count(int* out);
void srn_limits(srn_limits_t* out);
const char* srn_last_error(void);
const char* srn_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SERENADE_HIP_H */

// C ABI of libserenade_hip.so (include/serenade_hip.h): argument validation, error codes, handle
// ownership.  No compute happens here -- predict calls go to the HIP kernels or fail.
#include <algorithm>
#include <cstring>
#include <new>
#include <vector>

#include "srn_internal.h"

namespace srn { const char* last_error_cstr(); int device_count(); }
using namespace srn;

namespace {
template <typename F> int guarded(F f) {   // never let an exception cross the C boundary
    try { return f(); }
    catch (const std::bad_alloc&) { return fail(SRN_ENOMEM, "out of host memory"); }
    catch (const std::exception& e) { return fail(SRN_EINVAL, std::string("internal error: ") + e.what()); }
    catch (...) { return fail(SRN_EINVAL, "internal error"); }
}

int check_predict_args(const srn_index_t* idx, size_t k, size_t m, size_t how_many) {
    if (!idx) return fail(SRN_EINVAL, "null index");
    if (!idx->dev) return fail(SRN_ENODEV, "index has no device attached; there is no CPU fallback behind this ABI");
    if (idx->flat.postings_only) return fail(SRN_EINVAL, "a postings-only view holds no rows: it serves a shard group (srn_shard_group_set_postings), not predict calls");
    // the reference panics (peek_mut().unwrap() on an empty heap) when any of these is zero
    if (k == 0 || m == 0 || how_many == 0) return fail(SRN_EINVAL, "k, m and how_many must be > 0");
    if (how_many > SRN_MAX_HOW_MANY) return fail(SRN_ERANGE, "how_many above SRN_MAX_HOW_MANY");
    if (k > SRN_MAX_K) return fail(SRN_ERANGE, "k above SRN_MAX_K");
    if (m > 0x7FFFFFFFull) return fail(SRN_ERANGE, "m too large");
    return SRN_OK;
}
// an item shard answers only through the three stages: its rows are fragments (DeviceIndex::row_frag) and its lists a subset
static int check_not_a_shard(const srn_index_t* idx) {
    return idx && idx->flat.n_shards > 1 ? fail(SRN_EINVAL, "this index is one item shard of several: predictions go through its shard group (srn_shard_group_predict_batch)") : SRN_OK;
}

int predict_host(const srn_index_t* idx, const uint64_t* items_flat, const uint32_t* q_off, size_t nq, size_t k, size_t m,
                 size_t how_many, unsigned flags, uint64_t* out_ids, double* out_scores, uint32_t* out_counts,
                 uint32_t* out_stats, uint32_t* out_nb_sessions, uint32_t* out_nb_num, uint32_t* out_nb_counts) {
    int rc = check_predict_args(idx, k, m, how_many); if (rc) return rc;
    rc = check_not_a_shard(idx); if (rc) return rc;
    if (nq == 0) return SRN_OK;
    if (!items_flat || !q_off || !out_ids || !out_scores || !out_counts) return fail(SRN_EINVAL, "null buffer");
    if (nq > 0x7FFFFFFFull) return fail(SRN_ERANGE, "too many queries in one batch");
    if ((out_nb_sessions != nullptr) != (out_nb_num != nullptr) || (out_nb_sessions != nullptr) != (out_nb_counts != nullptr))
        return fail(SRN_EINVAL, "neighbour debug outputs must be given together");
    uint32_t max_len = 0;
    for (size_t q = 0; q < nq; ++q) {
        if (q_off[q + 1] < q_off[q]) return fail(SRN_EINVAL, "q_off not monotone");
        const uint32_t len = q_off[q + 1] - q_off[q];
        if (len == 0) return fail(SRN_EINVAL, "empty evolving session (the reference panics: src/vmisknn/mod.rs:157)");
        max_len = std::max(max_len, len);
    }
    if (max_len > SRN_MAX_SESSION_LEN) return fail(SRN_ERANGE, "evolving session longer than SRN_MAX_SESSION_LEN");
    LaunchParams p{};
    p.nq = (uint32_t)nq; p.k = (uint32_t)k; p.m = (uint32_t)m; p.how_many = (uint32_t)how_many; p.flags = flags; p.max_len = max_len;
    rc = device_predict(idx->dev, idx->flat, p, false, nullptr, items_flat, q_off, out_ids, out_scores, out_counts, out_stats,
                        out_nb_sessions, out_nb_num, out_nb_counts);
    if (rc) return rc;
    for (size_t q = 0; q < nq; ++q)
        if (out_counts[q] == 0xFFFFFFFFu) return fail(SRN_ERANGE, "a query exceeded the kernel's table limits");
    if (out_nb_sessions)   // recency rank -> reference session index
        for (size_t q = 0; q < nq; ++q)
            for (uint32_t j = 0; j < out_nb_counts[q]; ++j) {
                uint32_t& r = out_nb_sessions[q * k + j];
                r = idx->flat.rank_to_session[r];
            }
    return SRN_OK;
}
}  // namespace

extern "C" {

const char* srn_last_error(void) { return last_error_cstr(); }
const char* srn_version(void) { return "serenade_hip 0.1 (gfx950)"; }
int srn_device_count(int* out) { if (!out) return fail(SRN_EINVAL, "null argument"); *out = device_count(); return SRN_OK; }
void srn_limits(srn_limits_t* out) { if (out) *out = srn_limits_t{SRN_MAX_HOW_MANY, SRN_MAX_SESSION_LEN, SRN_MAX_K, 0}; }

int srn_sessions_from_tsv(const char* path, srn_sessions_t** out) {
    return guarded([&]() -> int {
        if (!path || !out) return fail(SRN_EINVAL, "null argument");
        *out = nullptr;
        srn_sessions* s = new srn_sessions();
        int rc = sessions_from_tsv(path, s->s);
        if (rc) { delete s; return rc; }
        *out = s; return SRN_OK; });
}
int srn_sessions_view(const srn_sessions_t* s, srn_sessions_view_t* out) {
    if (!s || !out) return fail(SRN_EINVAL, "null argument");
    out->sess_off = s->s.off.data(); out->items = s->s.items.data(); out->max_ts = s->s.ts.data(); out->n_sessions = s->s.ts.size();
    return SRN_OK;
}
int srn_sessions_length_quantile(const srn_sessions_t* s, double q, uint64_t* out) {
    if (!s || !out || !(q >= 0.0 && q <= 1.0)) return fail(SRN_EINVAL, "bad argument");
    return guarded([&]() -> int { *out = sessions_length_quantile(s->s.off.data(), s->s.ts.size(), q); return SRN_OK; });
}
void srn_sessions_free(srn_sessions_t* s) { delete s; }

int srn_index_build(const srn_sessions_view_t* sessions, size_t m_index, size_t max_session_len, double idf_weighting,
                    int device, srn_index_t** out) {
    return guarded([&]() -> int {
        if (!sessions || !out) return fail(SRN_EINVAL, "null argument");
        *out = nullptr;
        srn_index* ix = new srn_index();
        int rc = build_flat_index(*sessions, m_index, max_session_len, idf_weighting, 0, 1, ix->flat);
        if (rc == SRN_OK && device >= 0) { ix->dev = device_attach(ix->flat, device); ix->device = device; if (!ix->dev) rc = SRN_EHIP; else ix->comb = combiner_create(); }
        if (rc) { delete ix; return rc; }
        *out = ix; return SRN_OK; });
}

int srn_index_build_gpu(const srn_sessions_view_t* sessions, size_t m_index, size_t max_session_len, double idf_weighting,
                        int device, srn_index_t** out) {
    return guarded([&]() -> int {
        if (!sessions || !out) return fail(SRN_EINVAL, "null argument");
        if (device < 0) return fail(SRN_ENODEV, "the GPU index builder needs a device");
        *out = nullptr;
        srn_index* ix = new srn_index();
        int rc = build_flat_index_gpu(*sessions, m_index, max_session_len, idf_weighting, device, ix->flat);
        if (rc == SRN_OK) { ix->dev = device_attach(ix->flat, device); ix->device = device; if (!ix->dev) rc = SRN_EHIP; else ix->comb = combiner_create(); }
        if (rc) { delete ix; return rc; }
        *out = ix; return SRN_OK; });
}

int srn_index_new_from_csv(const char* path, size_t m_most_recent_sessions, double idf_weighting, size_t max_session_len,
                           int device, srn_index_t** out) {
    return guarded([&]() -> int {
        if (!path || !out) return fail(SRN_EINVAL, "null argument");
        Sessions s; int rc = sessions_from_tsv(path, s); if (rc) return rc;
        if (max_session_len == 0) max_session_len = sessions_length_quantile(s.off.data(), s.ts.size(), 0.995);
        srn_sessions_view_t v{s.off.data(), s.items.data(), s.ts.data(), s.ts.size()};
        // same bytes either way; the GPU builder covers what fits 32-bit ranks and offsets
        const bool gpu = device >= 0 && s.ts.size() < 0xFFFFFFFFull && s.items.size() < 0xFFFFFFFFull;
        return gpu ? srn_index_build_gpu(&v, m_most_recent_sessions, max_session_len, idf_weighting, device, out)
                   : srn_index_build(&v, m_most_recent_sessions, max_session_len, idf_weighting, device, out); });
}

int srn_index_new_from_avro(const char* base_path, int device, srn_index_t** out) {
    return guarded([&]() -> int {
        if (!base_path || !out) return fail(SRN_EINVAL, "null argument");
        *out = nullptr;
        srn_index* ix = new srn_index();
        int rc = build_flat_index_from_avro(base_path, ix->flat);
        if (rc == SRN_OK && device >= 0) { ix->dev = device_attach(ix->flat, device); ix->device = device; if (!ix->dev) rc = SRN_EHIP; else ix->comb = combiner_create(); }
        if (rc) { delete ix; return rc; }
        *out = ix; return SRN_OK; });
}

int srn_index_save(const srn_index_t* idx, const char* path) {
    if (!idx || !path) return fail(SRN_EINVAL, "null argument");
    return guarded([&]() -> int { int rc = check_has_rows(idx->flat, "srn_index_save"); if (rc) return rc; return save_flat_index(idx->flat, path); });
}
int srn_index_load(const char* path, int device, srn_index_t** out) {
    return guarded([&]() -> int {
        if (!path || !out) return fail(SRN_EINVAL, "null argument");
        *out = nullptr;
        srn_index* ix = new srn_index();
        int rc = load_flat_index(path, ix->flat);
        if (rc == SRN_OK && device >= 0) { ix->dev = device_attach(ix->flat, device); ix->device = device; if (!ix->dev) rc = SRN_EHIP; else ix->comb = combiner_create(); }
        if (rc) { delete ix; return rc; }
        *out = ix; return SRN_OK; });
}
int srn_index_set_attributes(srn_index_t* idx, const uint64_t* item_ids, const uint8_t* flags, size_t n) {
    if (!idx || (n && (!item_ids || !flags))) return fail(SRN_EINVAL, "null argument");
    return guarded([&]() -> int {
        for (size_t i = 0; i < n; ++i) {
            const uint32_t j = idx->flat.lookup(item_ids[i]);
            if (j == kNone) continue;   // attributes of items outside the index can never be consulted
            idx->flat.attr[j] = flags[i] == SRN_ATTR_NONE ? (uint8_t)SRN_ATTR_NONE : (uint8_t)(flags[i] & 3u);
        }
        return idx->dev ? device_update_attr(idx->dev, idx->flat) : SRN_OK; });
}
int srn_index_info(const srn_index_t* idx, srn_index_info_t* out) {
    if (!idx || !out) return fail(SRN_EINVAL, "null argument");
    const FlatIndex& f = idx->flat;
    uint64_t incomplete = 0;
    if (!f.lists_complete && f.viol.size() == f.n_items)
        for (uint64_t i = 0; i < f.n_items; ++i) { const uint64_t len = f.post_off[i + 1] - f.post_off[i];
            if (f.viol[i] != 0u && !(len == f.m_index && len != 0 && f.viol[i] <= f.post_rank[f.post_off[i + 1] - 1])) ++incomplete; }
    else if (!f.lists_complete) incomplete = f.n_items;   // (a shard cut from such an index: not measured per item)
    *out = srn_index_info_t{f.n_items, f.n_sessions_total, f.n_kept, f.nnz_rows, f.nnz_post, f.m_index, f.max_session_len,
                            f.max_row_len, device_bytes(idx->dev), idx->dev ? idx->device : -1,
                            f.nnz_rows >= 0xFFFFFFFFull ? 1 : 0, f.idf_weighting, incomplete};
    return SRN_OK;
}
int srn_index_postings(const srn_index_t* idx, uint64_t item_id, uint32_t* out_sessions, size_t cap, int64_t* out_len, double* out_idf) {
    if (!idx || !out_len) return fail(SRN_EINVAL, "null argument");
    const FlatIndex& f = idx->flat;
    const uint32_t j = f.lookup(item_id);
    if (j == kNone) { *out_len = -1; return SRN_OK; }
    const uint64_t o0 = f.post_off[j], o1 = f.post_off[j + 1];
    *out_len = (int64_t)(o1 - o0);
    if (out_sessions) for (uint64_t t = o0; t < o1 && t - o0 < cap; ++t) out_sessions[t - o0] = f.rank_to_session[f.post_rank[t]];
    if (out_idf) *out_idf = f.idf[j];
    return SRN_OK;
}
// ---- the rest of SimilarityComputationNew (src/vmisknn/similarity_indexed.rs:9-23) at the ABI: with srn_index_postings (+ idf) above, a maintainer can write
// `impl SimilarityComputationNew for HipVMISIndex`, not only swap the free function predict ----
int srn_index_items_for_session(const srn_index_t* idx, uint32_t session, uint64_t* out_items, size_t cap, size_t* out_len) {
    return guarded([&]() -> int {
        if (!idx || !out_len) return fail(SRN_EINVAL, "null argument");
        *out_len = 0;
        const FlatIndex& f = idx->flat;
        int rc = check_has_rows(f, "srn_index_items_for_session"); if (rc) return rc;
        if (f.n_shards > 1) return fail(SRN_EINVAL, "an item shard holds row FRAGMENTS (the items it owns): ask the unsharded index");
        if (session >= f.n_sessions_total) return fail(SRN_EINVAL, "no such session (the reference indexes out of bounds: vmis_index.rs:317-319)");
        std::call_once(idx->s2r_once, [&] { idx->session_to_rank.assign(f.n_sessions_total, kNone); for (uint64_t r = 0; r < f.n_kept; ++r) idx->session_to_rank[f.rank_to_session[r]] = (uint32_t)r; });
        const uint32_t r = idx->session_to_rank[session];
        // (the reference keeps the rows of ALL sessions, vmis_index.rs:79, but only sessions of <= max_session_len items enter the posting lists, :452 -- and only those can
        //  ever be neighbours, so only their rows are kept here)
        if (r == kNone) return fail(SRN_ERANGE, "this session is in no posting list (longer than max_session_len, vmis_index.rs:452, or named by no list of a pre-built index): never a neighbour, and its row is not kept");
        const uint64_t o0 = f.row_off[r], o1 = f.row_off[r + 1];
        *out_len = (size_t)(o1 - o0);
        if (out_items) for (uint64_t t = o0; t < o1 && t - o0 < cap; ++t) out_items[t - o0] = f.item_id[f.row_items[t]];   // (row order = ascending public id, as the reference's item_ids_asc)
        return SRN_OK; });
}
int srn_index_session_recency(const srn_index_t* idx, uint32_t* out_rank, size_t cap) {
    return guarded([&]() -> int {
        if (!idx || !out_rank) return fail(SRN_EINVAL, "null argument");
        const FlatIndex& f = idx->flat;
        int rc = check_has_rows(f, "srn_index_session_recency"); if (rc) return rc;
        if (cap < f.n_sessions_total) return fail(SRN_EINVAL, "room for n_sessions entries needed");
        std::fill(out_rank, out_rank + f.n_sessions_total, kNone);
        for (uint64_t r = 0; r < f.n_kept; ++r) out_rank[f.rank_to_session[r]] = (uint32_t)r;
        return SRN_OK; });
}
int srn_index_find_attributes(const srn_index_t* idx, uint64_t item_id, uint8_t* out_flags) {
    if (!idx || !out_flags) return fail(SRN_EINVAL, "null argument");
    const uint32_t j = idx->flat.lookup(item_id);
    *out_flags = j == kNone ? (uint8_t)SRN_ATTR_NONE : idx->flat.attr[j];   // (None for an unknown item, like HashMap::get, vmis_index.rs:417-419)
    return SRN_OK;
}
// find_neighbors (vmis_index.rs:325-415) by itself: the k neighbours of an evolving session as (reference session index, similarity = numerator / U), canonical
// semantics, best first -- similarity descending, ties: the more recent session first.  The GPU does the work (the general kernel with its
// neighbour dump); not a debug aid: same k / m / session-length limits as srn_predict, re-entrant.
int srn_find_neighbors(const srn_index_t* idx, const uint64_t* evolving, size_t len, size_t k, size_t m, uint32_t* out_sessions, double* out_scores, size_t* out_n) {
    return guarded([&]() -> int {
        if (!out_n) return fail(SRN_EINVAL, "null out_n");
        *out_n = 0;
        if (!evolving || len == 0) return fail(SRN_EINVAL, "empty evolving session");
        if (len > SRN_MAX_SESSION_LEN) return fail(SRN_ERANGE, "evolving session longer than SRN_MAX_SESSION_LEN");
        int rc = check_predict_args(idx, k, m, 1); if (rc) return rc;
        rc = check_not_a_shard(idx); if (rc) return rc;
        if (!out_sessions || !out_scores) return fail(SRN_EINVAL, "null buffer");
        std::vector<uint32_t> nb_rank(k), nb_num(k); uint32_t nb_cnt = 0, cnt = 0, stats[8] = {0};
        uint64_t id1 = 0; double sc1 = 0.0;
        const uint32_t q_off[2] = {0u, (uint32_t)len};
        LaunchParams p{};
        p.nq = 1; p.k = (uint32_t)k; p.m = (uint32_t)m; p.how_many = 1; p.flags = 0; p.max_len = (uint32_t)len;
        rc = device_predict(idx->dev, idx->flat, p, false, nullptr, evolving, q_off, &id1, &sc1, &cnt, stats, nb_rank.data(), nb_num.data(), &nb_cnt);
        if (rc) return rc;
        if (cnt == 0xFFFFFFFFu) return fail(SRN_ERANGE, "the query exceeded the kernel's table limits");
        size_t U = 0;   // distinct raw ids, known or not (vmis_index.rs:334-352: the decay's denominator)
        for (size_t i = 0; i < len; ++i) { bool first = true; for (size_t j = 0; j < i; ++j) first = first && evolving[j] != evolving[i]; U += first; }
        std::vector<uint32_t> ord(nb_cnt);
        for (uint32_t i = 0; i < nb_cnt; ++i) ord[i] = i;
        std::sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) { return nb_num[a] != nb_num[b] ? nb_num[a] > nb_num[b] : nb_rank[a] > nb_rank[b]; });
        for (uint32_t i = 0; i < nb_cnt; ++i) { out_sessions[i] = idx->flat.rank_to_session[nb_rank[ord[i]]]; out_scores[i] = (double)nb_num[ord[i]] / (double)U; }
        *out_n = nb_cnt;
        return SRN_OK; });
}

void srn_index_free(srn_index_t* idx) {
    if (!idx) return;
    device_release(idx->dev);
    if (idx->comb) combiner_free(idx->comb);
    delete idx;
}

int srn_predict(const srn_index_t* idx, const uint64_t* evolving, size_t len, size_t k, size_t m, size_t how_many,
                int enable_business_logic, uint64_t* out_ids, double* out_scores, size_t* out_n) {
    return guarded([&]() -> int {
        if (!out_n) return fail(SRN_EINVAL, "null out_n");
        *out_n = 0;
        if (!evolving || len == 0) return fail(SRN_EINVAL, "empty evolving session (the reference panics: src/vmisknn/mod.rs:157)");
        if (len > SRN_MAX_SESSION_LEN) return fail(SRN_ERANGE, "evolving session longer than SRN_MAX_SESSION_LEN");
        // (round 6) a resident workgroup of the persistent latency path, if the handle has them and this call is what they serve: no launch at all
        if (idx && idx->dev && out_ids && out_scores && len <= 16 && k <= 0xFFFFFFFFull && m <= 0xFFFFFFFFull && how_many <= 0xFFFFFFFFull &&
            device_serve_predict(idx->dev, evolving, (uint32_t)len, (uint32_t)k, (uint32_t)m, (uint32_t)how_many, enable_business_logic ? SRN_FLAG_BUSINESS_LOGIC : 0u, out_ids, out_scores, out_n) == 0)
            return SRN_OK;
        // concurrent calls on one handle share launches (srn_combine.cpp); a lone caller runs its own round of one at once
        const int lanes = knob_predict_lanes();
        if (lanes > 0 && idx && idx->comb) {
            int rc = check_predict_args(idx, k, m, how_many); if (rc) return rc;
            rc = check_not_a_shard(idx); if (rc) return rc;
            if (!out_ids || !out_scores) return fail(SRN_EINVAL, "null buffer");
            return combiner_predict(idx->comb, idx, evolving, len, k, m, how_many, enable_business_logic ? SRN_FLAG_BUSINESS_LOGIC : 0, out_ids, out_scores, out_n,
                                    lanes, knob_tiny_max());
        }
        const uint32_t q_off[2] = {0, (uint32_t)len}; uint32_t cnt = 0;
        int rc = predict_host(idx, evolving, q_off, 1, k, m, how_many, enable_business_logic ? SRN_FLAG_BUSINESS_LOGIC : 0, out_ids,
                              out_scores, &cnt, nullptr, nullptr, nullptr, nullptr);
        if (rc == SRN_OK) *out_n = cnt;
        return rc; });
}

// ---- the persistent latency path (srn_runtime.hip, "serve") ----
int srn_index_serve_start(srn_index_t* idx, size_t k, size_t m, size_t how_many, int enable_business_logic, unsigned lanes, unsigned max_items_in_session, unsigned idle_ms) {
    return guarded([&]() -> int {
        if (!idx) return fail(SRN_EINVAL, "null index");
        if (!idx->dev) return fail(SRN_ENODEV, "index has no device attached");
        int rc = check_predict_args(idx, k, m, how_many); if (rc) return rc;
        rc = check_not_a_shard(idx); if (rc) return rc;
        return device_serve_start(idx->dev, idx->flat, (uint32_t)k, (uint32_t)m, (uint32_t)how_many, enable_business_logic ? SRN_FLAG_BUSINESS_LOGIC : 0u, lanes, max_items_in_session, idle_ms); });
}
int srn_index_serve_stop(srn_index_t* idx) {
    return guarded([&]() -> int {
        if (!idx) return fail(SRN_EINVAL, "null index");
        return idx->dev ? device_serve_stop(idx->dev) : SRN_OK; });
}
int srn_index_serve_stats(const srn_index_t* idx, uint64_t* out_served, uint64_t* out_not_served, uint64_t* out_launches, uint32_t* out_lanes) {
    if (!idx || !idx->dev) return fail(SRN_ENODEV, "index has no device attached");
    return device_serve_stats(idx->dev, out_served, out_not_served, out_launches, out_lanes);
}

int srn_predict_stats(const srn_index_t* idx, uint64_t* out_rounds, uint64_t* out_requests, uint64_t* out_max_round) {
    if (!idx || !idx->comb) return fail(SRN_ENODEV, "index has no device attached");
    combiner_stats(idx->comb, out_rounds, out_requests, out_max_round);
    return SRN_OK;
}

int srn_predict_batch(const srn_index_t* idx, const uint64_t* items_flat, const uint32_t* q_off, size_t nq, size_t k, size_t m,
                      size_t how_many, unsigned flags, uint64_t* out_ids, double* out_scores, uint32_t* out_counts) {
    return guarded([&]() -> int { return predict_host(idx, items_flat, q_off, nq, k, m, how_many, flags, out_ids, out_scores,
                                                       out_counts, nullptr, nullptr, nullptr, nullptr); });
}

int srn_predict_batch_debug(const srn_index_t* idx, const uint64_t* items_flat, const uint32_t* q_off, size_t nq, size_t k, size_t m,
                            size_t how_many, unsigned flags, uint64_t* out_ids, double* out_scores, uint32_t* out_counts,
                            uint32_t* out_stats, uint32_t* out_nb_sessions, uint32_t* out_nb_num, uint32_t* out_nb_counts) {
    return guarded([&]() -> int { return predict_host(idx, items_flat, q_off, nq, k, m, how_many, flags, out_ids, out_scores,
                                                       out_counts, out_stats, out_nb_sessions, out_nb_num, out_nb_counts); });
}

int srn_predict_batch_device(const srn_index_t* idx, const uint64_t* d_items_flat, const uint32_t* d_q_off, size_t nq,
                             size_t max_len_hint, size_t k, size_t m, size_t how_many, unsigned flags, uint64_t* d_out_ids,
                             double* d_out_scores, uint32_t* d_out_counts, void* stream) {
    return guarded([&]() -> int {
        int rc = check_predict_args(idx, k, m, how_many); if (rc) return rc;
        rc = check_not_a_shard(idx); if (rc) return rc;
        if (nq == 0) return SRN_OK;
        if (!d_items_flat || !d_q_off || !d_out_ids || !d_out_scores || !d_out_counts) return fail(SRN_EINVAL, "null buffer");
        if (nq > 0x7FFFFFFFull) return fail(SRN_ERANGE, "too many queries in one batch");
        if (max_len_hint == 0 || max_len_hint > SRN_MAX_SESSION_LEN) return fail(SRN_ERANGE, "max_len_hint out of range");
        LaunchParams p{};
        p.nq = (uint32_t)nq; p.k = (uint32_t)k; p.m = (uint32_t)m; p.how_many = (uint32_t)how_many; p.flags = flags;
        p.max_len = (uint32_t)max_len_hint;
        p.items_flat = d_items_flat; p.q_off = d_q_off; p.out_ids = d_out_ids; p.out_scores = d_out_scores; p.out_counts = d_out_counts;
        return device_predict(idx->dev, idx->flat, p, true, stream, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                              nullptr, nullptr); });
}

int srn_index_reserve(const srn_index_t* idx, size_t nq, size_t max_len_hint, size_t k, size_t m, size_t how_many, unsigned flags, void* stream) {
    return guarded([&]() -> int {
        int rc = check_predict_args(idx, k, m, how_many); if (rc) return rc;
        if (nq == 0 || nq > 0x7FFFFFFFull) return fail(SRN_ERANGE, "nq out of range");
        if (max_len_hint == 0 || max_len_hint > SRN_MAX_SESSION_LEN) return fail(SRN_ERANGE, "max_len_hint out of range");
        LaunchParams p{};
        p.nq = (uint32_t)nq; p.k = (uint32_t)k; p.m = (uint32_t)m; p.how_many = (uint32_t)how_many; p.flags = flags; p.max_len = (uint32_t)max_len_hint;
        return device_reserve(idx->dev, idx->flat, p, stream); });
}

int srn_last_kernel_ms(const srn_index_t* idx, double* out_ms_main, double* out_ms_retry, uint32_t* out_retried) {
    if (!idx || !idx->dev) return fail(SRN_ENODEV, "index has no device attached");
    return guarded([&]() -> int { return device_last_kernel_ms(idx->dev, out_ms_main, out_ms_retry, out_retried); });
}

// ---- item-sharded index (DESIGN.md "Multi-GPU"): one shard per GPU, three kernel stages around three collectives ----
int srn_index_build_shard(const srn_sessions_view_t* sessions, size_t m_index, size_t max_session_len, double idf_weighting,
                          uint32_t shard, uint32_t n_shards, int device, srn_index_t** out) {
    return guarded([&]() -> int {
        if (!sessions || !out) return fail(SRN_EINVAL, "null argument");
        *out = nullptr;
        srn_index* ix = new srn_index();
        int rc = build_flat_index(*sessions, m_index, max_session_len, idf_weighting, shard, n_shards, ix->flat);
        if (rc == SRN_OK && device >= 0) { ix->dev = device_attach(ix->flat, device); ix->device = device; if (!ix->dev) rc = SRN_EHIP; else ix->comb = combiner_create(); }
        if (rc) { delete ix; return rc; }
        *out = ix; return SRN_OK; });
}
int srn_index_shard(const srn_index_t* full, uint32_t shard, uint32_t n_shards, int device, srn_index_t** out) {
    return guarded([&]() -> int {
        if (!full || !out) return fail(SRN_EINVAL, "null argument");
        *out = nullptr;
        { int rc0 = check_has_rows(full->flat, "srn_index_shard"); if (rc0) return rc0; }
        srn_index* ix = new srn_index();
        int rc = shard_flat_index(full->flat, shard, n_shards, ix->flat);
        if (rc == SRN_OK && device >= 0) { ix->dev = device_attach(ix->flat, device); ix->device = device; if (!ix->dev) rc = SRN_EHIP; else ix->comb = combiner_create(); }
        if (rc) { delete ix; return rc; }
        *out = ix; return SRN_OK; });
}
// The replicated part of an item-sharded index: the whole index's dictionary, idf / attributes and posting lists WITHOUT its rows (config 5: ~10 GB of the 66.6 GB)
int srn_index_postings_view(const srn_index_t* full, int device, srn_index_t** out) {
    return guarded([&]() -> int {
        if (!full || !out) return fail(SRN_EINVAL, "null argument");
        *out = nullptr;
        if (full->flat.n_shards != 1) return fail(SRN_EINVAL, "the postings view is cut from an UNSHARDED index");
        { int rc0 = check_has_rows(full->flat, "srn_index_postings_view"); if (rc0) return rc0; }
        if (device < 0) return fail(SRN_ENODEV, "a postings view lives on a device");
        srn_index* ix = new srn_index();
        const FlatIndex& f = full->flat; FlatIndex& g = ix->flat;
        g.n_items = f.n_items; g.n_sessions_total = f.n_sessions_total; g.n_kept = f.n_kept; g.nnz_rows = 0; g.nnz_post = f.nnz_post; g.m_index = f.m_index;
        g.max_session_len = f.max_session_len; g.max_row_len = f.max_row_len /* (of the index the lists belong to: the shard group derives its choice of pipeline from it, the same on every rank) */; g.idf_weighting = f.idf_weighting; g.lists_complete = f.lists_complete; g.postings_only = true;
        g.total_pairs = f.total_pairs; g.item_id = f.item_id; g.id_rank = f.id_rank; g.idf = f.idf; g.attr = f.attr; g.post_off = f.post_off; g.post_rank = f.post_rank;
        g.id_table = f.id_table; g.id_mask = f.id_mask; g.rank_to_session = f.rank_to_session;   // (srn_index_postings answers in reference session indices)
        ix->dev = device_attach(ix->flat, device); ix->device = device;
        if (!ix->dev) { delete ix; return SRN_EHIP; }
        *out = ix; return SRN_OK; });
}
int srn_index_build_shard_gpu(const srn_sessions_view_t* sessions, size_t m_index, size_t max_session_len, double idf_weighting,
                              uint32_t shard, uint32_t n_shards, int device, srn_index_t** out) {
    return guarded([&]() -> int {
        if (!sessions || !out) return fail(SRN_EINVAL, "null argument");
        if (device < 0) return fail(SRN_ENODEV, "the GPU index builder needs a device");
        *out = nullptr;
        FlatIndex full;   // built on the GPU, never attached: only the shard goes to HBM
        int rc = build_flat_index_gpu(*sessions, m_index, max_session_len, idf_weighting, device, full);
        if (rc) return rc;
        srn_index* ix = new srn_index();
        rc = shard_flat_index(full, shard, n_shards, ix->flat);
        if (rc == SRN_OK) { ix->dev = device_attach(ix->flat, device); ix->device = device; if (!ix->dev) rc = SRN_EHIP; else ix->comb = combiner_create(); }
        if (rc) { delete ix; return rc; }
        *out = ix; return SRN_OK; });
}
int srn_index_load_shard(const char* path, uint32_t shard, uint32_t n_shards, int device, srn_index_t** out) {
    return guarded([&]() -> int {
        if (!path || !out) return fail(SRN_EINVAL, "null argument");
        *out = nullptr;
        FlatIndex full;
        int rc = load_flat_index(path, full);
        if (rc) return rc;
        srn_index* ix = new srn_index();
        rc = full.n_shards == 1 ? shard_flat_index(full, shard, n_shards, ix->flat)
                                : (full.shard == shard && full.n_shards == n_shards ? (ix->flat = std::move(full), SRN_OK) : fail(SRN_EINVAL, "the file holds another shard"));
        if (rc == SRN_OK && device >= 0) { ix->dev = device_attach(ix->flat, device); ix->device = device; if (!ix->dev) rc = SRN_EHIP; else ix->comb = combiner_create(); }
        if (rc) { delete ix; return rc; }
        *out = ix; return SRN_OK; });
}
int srn_kernel_times(const srn_index_t* idx, uint32_t max_n, double* out_ms_main, double* out_ms_retry, uint32_t* out_n) {
    if (!idx || !idx->dev) return fail(SRN_ENODEV, "index has no device attached");
    if (!out_n) return fail(SRN_EINVAL, "null argument");
    return guarded([&]() -> int { return device_kernel_times(idx->dev, max_n, out_ms_main, out_ms_retry, out_n); });
}

int srn_kernel_times_detail(const srn_index_t* idx, uint32_t max_n, double* out_ms_prep, double* out_ms_fast, double* out_ms_predict,
                            double* out_ms_retry, uint32_t* out_n) {
    if (!idx || !idx->dev) return fail(SRN_ENODEV, "index has no device attached");
    if (!out_n) return fail(SRN_EINVAL, "null argument");
    return guarded([&]() -> int { return device_kernel_times(idx->dev, max_n, out_ms_predict, out_ms_retry, out_n, out_ms_prep, out_ms_fast); });
}

int srn_kernel_timing(srn_index_t* idx, int enable) {
    if (!idx || !idx->dev) return fail(SRN_ENODEV, "index has no device attached");
    return device_kernel_timing(idx->dev, enable);
}

int srn_debug_phase_cycles(const srn_index_t* idx, int enable, uint64_t* out16) {
    if (!idx || !idx->dev) return fail(SRN_ENODEV, "index has no device attached");
    return guarded([&]() -> int { return device_phase_cycles(idx->dev, enable, (unsigned long long*)out16); });
}

int srn_last_path_counts(const srn_index_t* idx, uint32_t* out_nq, uint32_t* out_general, uint32_t* out_global_pass) {
    if (!idx || !idx->dev) return fail(SRN_ENODEV, "index has no device attached");
    return guarded([&]() -> int { return device_last_path_counts(idx->dev, out_nq, out_general, out_global_pass); });
}
uint32_t srn_debug_shard_nb_positions_stride(size_t k, size_t m) {
    LaunchParams p{}; p.k = (uint32_t)k; p.m = (uint32_t)m; p.max_len = 4; p.nq = 1; p.how_many = 21;
    return device_shard_nb_positions_stride(p);
}
int srn_debug_serve_stamps(const srn_index_t* idx, uint32_t* out4) {
    if (!idx || !idx->dev || !out4) return fail(SRN_EINVAL, "null argument");
    return device_serve_last_stamps(idx->dev, out4);
}
int srn_debug_last_mid_count(const srn_index_t* idx, uint32_t* out_listed) {
    if (!idx || !idx->dev) return fail(SRN_ENODEV, "index has no device attached");
    return guarded([&]() -> int { return device_last_mid_count(idx->dev, out_listed); });
}
int srn_debug_last_big_count(const srn_index_t* idx, uint32_t* out_listed) {
    if (!idx || !idx->dev) return fail(SRN_ENODEV, "index has no device attached");
    return guarded([&]() -> int { return device_last_mid_count(idx->dev, nullptr, out_listed); });
}

int srn_debug_sback_launches(const srn_index_t* idx, uint64_t* out_launches) {
    if (!idx || !idx->dev || !out_launches) return fail(SRN_EINVAL, "null argument / no device");
    *out_launches = device_sback_launches(idx->dev); return SRN_OK;
}

void srn_debug_reload_knobs(void) { reload_knobs(); }

}  // extern "C"

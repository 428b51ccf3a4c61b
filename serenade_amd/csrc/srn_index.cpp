// Host side of the index: TSV loader, flat-index builder, binary save / load.
//
// Mirrors the *semantics* of the reference's index construction so that the GPU path consumes the
// same logical index -- read_from_file (src/vmisknn/vmis_index.rs:591-686) and prepare_hashmap
// (:422-528) -- but produces the flat CSR layout described in DESIGN.md instead of hash maps of
// vectors: item ids become dense indices in popularity order, sessions are renumbered by recency
// rank (so "more recent" == larger u32 and the timestamp gather of find_neighbors disappears),
// postings and rows are two CSR arrays.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <thread>
#include <cstdlib>

#include "srn_internal.h"

namespace srn {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
int fail(int code, const std::string& msg) { g_err = msg; return code; }
const char* last_error_cstr() { return g_err.c_str(); }
std::string last_error_string() { return g_err; }

// ---------------------------------------------------------------------------------------------
// TSV loader
// ---------------------------------------------------------------------------------------------
namespace {
struct Row { uint64_t session, item, time; };

// one "SessionId\tItemId\tTime" line; false = unparsable (the reference skips such rows, :614-616)
bool parse_line(const char* p, const char* end, Row& r) {
    auto parse_u64 = [&](uint64_t& v) {
        if (p >= end || *p < '0' || *p > '9') return false;
        v = 0; while (p < end && *p >= '0' && *p <= '9') v = v * 10 + (uint64_t)(*p++ - '0');
        return true; };
    if (!parse_u64(r.session)) return false;
    if (p >= end || *p != '\t') return false; ++p;
    if (!parse_u64(r.item)) return false;
    if (p >= end || *p != '\t') return false; ++p;
    char buf[64]; size_t n = std::min<size_t>((size_t)(end - p), sizeof buf - 1);
    memcpy(buf, p, n); buf[n] = 0;
    char* stop = nullptr; double t = strtod(buf, &stop);
    if (stop == buf) return false;
    while (*stop == ' ' || *stop == '\r') ++stop;
    if (*stop != 0 && *stop != '\t') return false;
    if (!(t >= 0)) t = 0;                 // `as usize` saturates negatives / NaN to 0
    r.time = (uint64_t)std::llround(t);   // f64::round(): half away from zero (:607-609)
    return true;
}
}  // namespace

int sessions_from_tsv(const char* path, Sessions& out) {
    FILE* f = fopen(path, "rb");
    if (!f) return fail(SRN_EIO, std::string("cannot open ") + path);
    std::vector<char> text;
    { char chunk[1 << 16]; size_t n; while ((n = fread(chunk, 1, sizeof chunk, f)) > 0) text.insert(text.end(), chunk, chunk + n); }
    fclose(f);
    std::vector<Row> rows;
    const char* p = text.data(); const char* end = p + text.size(); bool header = true;
    while (p < end) {
        const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p)); if (!nl) nl = end;
        if (header) header = false;
        else if (nl > p) { Row r; if (parse_line(p, nl, r)) rows.push_back(r); }
        p = nl + 1;
    }
    if (rows.empty()) return fail(SRN_EINVAL, "no training rows");
    // the data is unsorted: stable order by session id keeps file order inside a session (:620-621)
    std::stable_sort(rows.begin(), rows.end(), [](const Row& a, const Row& b) { return a.session < b.session; });
    out = Sessions(); out.off.push_back(0);
    std::vector<uint64_t> cur{rows[0].item}; uint64_t cur_max = rows[0].time, cur_sid = rows[0].session;
    const size_t n = rows.size();
    for (size_t i = 1; i < n; ++i) {
        const bool same = rows[i].session == rows[i - 1].session && i != n - 1;   // final row never extends (:669)
        if (same) {
            if (std::find(cur.begin(), cur.end(), rows[i].item) == cur.end()) {
                cur.push_back(rows[i].item);
                cur_max = std::max(cur_max, rows[i].time);   // only non-duplicate rows move the max (:670-673)
            }
        } else {
            std::sort(cur.begin(), cur.end());
            out.items.insert(out.items.end(), cur.begin(), cur.end());
            out.off.push_back(out.items.size()); out.ts.push_back((uint32_t)cur_max); out.session_ids.push_back(cur_sid);
            cur.assign(1, rows[i].item); cur_max = rows[i].time; cur_sid = rows[i].session;
        }
    }
    // the session open at the end of the loop is never pushed (:675-686)
    return SRN_OK;
}

uint64_t sessions_length_quantile(const uint64_t* off, size_t n, double q) {
    if (n == 0) return 0;
    std::vector<uint64_t> len(n);
    for (size_t i = 0; i < n; ++i) len[i] = off[i + 1] - off[i];
    std::sort(len.begin(), len.end());
    const double pos = q * (double)(n - 1); const size_t lo = (size_t)std::floor(pos);
    const size_t hi = std::min(n - 1, lo + 1); const double frac = pos - (double)lo;
    return (uint64_t)std::llround((double)len[lo] + frac * ((double)len[hi] - (double)len[lo]));
}

// ---------------------------------------------------------------------------------------------
// flat index builder
// ---------------------------------------------------------------------------------------------
int build_flat_index(const srn_sessions_view_t& v, size_t m_index, size_t max_session_len, double idf_weighting,
                     uint32_t shard, uint32_t n_shards, FlatIndex& ix) {
    if (!v.sess_off || !v.max_ts || (v.sess_off[v.n_sessions] && !v.items)) return fail(SRN_EINVAL, "null session arrays");
    if (m_index == 0) return fail(SRN_EINVAL, "m_index must be > 0");
    if (n_shards == 0 || shard >= n_shards) return fail(SRN_EINVAL, "bad shard / n_shards");
    if (v.n_sessions >= 0xFFFFFFFFull) return fail(SRN_ERANGE, "too many sessions");
    ix = FlatIndex();
    ix.n_sessions_total = v.n_sessions; ix.m_index = m_index; ix.max_session_len = max_session_len; ix.idf_weighting = idf_weighting;
    ix.shard = shard; ix.n_shards = n_shards;

    // 1. kept sessions in ascending recency order: (timestamp, session index) lexicographic
    std::vector<uint64_t> key; key.reserve(v.n_sessions);
    for (size_t s = 0; s < v.n_sessions; ++s) {
        if (v.sess_off[s + 1] < v.sess_off[s]) return fail(SRN_EINVAL, "sess_off not monotone");
        const uint64_t len = v.sess_off[s + 1] - v.sess_off[s];
        if (len == 0 || len > max_session_len) continue;   // :452 (an empty session contributes nothing)
        key.push_back(((uint64_t)v.max_ts[s] << 32) | (uint64_t)s);
        ix.total_pairs += len;
    }
    std::sort(key.begin(), key.end());
    ix.n_kept = key.size();
    ix.rank_to_session.resize(ix.n_kept);
    for (size_t r = 0; r < ix.n_kept; ++r) ix.rank_to_session[r] = (uint32_t)(key[r] & 0xFFFFFFFFull);
    std::vector<uint64_t>().swap(key);

    // 2. dictionary: provisional dense index in first-seen order, then renumber ascending by id
    // (item-sharded: recency ranks stay global, only the items this shard owns get postings, row fragments and idf)
    size_t cap = 1024; while (cap < ix.total_pairs / n_shards / 4 + 16) cap <<= 1;   // grows on demand below
    std::vector<IdSlot> tab(cap, IdSlot{0, kNone, 0}); size_t tmask = cap - 1;
    std::vector<uint64_t> ids; std::vector<uint32_t> cnt;
    std::vector<uint32_t> prov(ix.total_pairs);
    ix.row_off.assign(ix.n_kept + 1, 0);
    auto grow = [&]() {
        std::vector<IdSlot> nt(tab.size() * 2, IdSlot{0, kNone, 0}); const size_t nm = nt.size() - 1;
        for (const IdSlot& s : tab) if (s.idx != kNone) { size_t h = mix64(s.key) & nm; while (nt[h].idx != kNone) h = (h + 1) & nm; nt[h] = s; }
        tab.swap(nt); tmask = nm; };
    size_t w = 0;
    for (size_t r = 0; r < ix.n_kept; ++r) {
        const uint32_t s = ix.rank_to_session[r];
        uint64_t prev = 0; bool first = true;
        for (uint64_t j = v.sess_off[s]; j < v.sess_off[s + 1]; ++j) {
            const uint64_t id = v.items[j];
            if (!first && id <= prev) return fail(SRN_EINVAL, "session rows must be strictly ascending item ids");
            prev = id; first = false;
            if (n_shards > 1 && item_owner(id, n_shards) != shard) continue;
            size_t h = mix64(id) & tmask;
            for (;;) {
                IdSlot& sl = tab[h];
                if (sl.idx == kNone) {
                    if (ids.size() >= 0xFFFFFFF0ull) return fail(SRN_ERANGE, "too many items");
                    sl.key = id; sl.idx = (uint32_t)ids.size(); ids.push_back(id); cnt.push_back(0);
                    prov[w] = sl.idx;
                    if (ids.size() * 2 > tab.size()) grow();
                    break;
                }
                if (sl.key == id) { prov[w] = sl.idx; break; }
                h = (h + 1) & tmask;
            }
            ++cnt[prov[w]]; ++w;
        }
        ix.row_off[r + 1] = w;
        ix.max_row_len = std::max<uint64_t>(ix.max_row_len, w - ix.row_off[r]);
    }
    ix.nnz_rows = w; prov.resize(w);
    ix.n_items = ids.size();
    // dense idx = popularity order (most sessions first, ties by ascending id): the kernel direct-maps the accumulators of
    // the first few thousand idx, and hot items' metadata share cache lines.  Ties in the final ranking are broken by
    // ascending PUBLIC id, carried as id_rank.
    std::vector<uint32_t> order(ix.n_items); std::iota(order.begin(), order.end(), 0u);
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return cnt[a] != cnt[b] ? cnt[a] > cnt[b] : ids[a] < ids[b]; });
    std::vector<uint32_t> remap(ix.n_items);
    ix.item_id.resize(ix.n_items);
    std::vector<uint32_t> count(ix.n_items);
    for (uint32_t i = 0; i < ix.n_items; ++i) { remap[order[i]] = i; ix.item_id[i] = ids[order[i]]; count[i] = cnt[order[i]]; }
    { std::vector<uint32_t> by_id(ix.n_items); std::iota(by_id.begin(), by_id.end(), 0u);
      std::sort(by_id.begin(), by_id.end(), [&](uint32_t a, uint32_t b) { return ix.item_id[a] < ix.item_id[b]; });
      ix.id_rank.resize(ix.n_items); for (uint32_t r = 0; r < ix.n_items; ++r) ix.id_rank[by_id[r]] = r; }
    ix.row_items.resize(ix.nnz_rows);
    for (size_t i = 0; i < ix.nnz_rows; ++i) ix.row_items[i] = remap[prov[i]];
    std::vector<uint32_t>().swap(prov); std::vector<uint64_t>().swap(ids); std::vector<uint32_t>().swap(cnt);
    std::vector<IdSlot>().swap(tab);

    // 3. postings: walk ranks newest -> oldest; each item's list is filled most-recent-first and
    //    stops at m_index (:497-504).  Rank order == (timestamp desc, session index desc).
    ix.post_off.assign(ix.n_items + 1, 0);
    for (uint32_t i = 0; i < ix.n_items; ++i) ix.post_off[i + 1] = ix.post_off[i] + std::min<uint64_t>(count[i], m_index);
    ix.nnz_post = ix.post_off[ix.n_items];
    ix.post_rank.resize(ix.nnz_post);
    std::vector<uint32_t> fill(ix.n_items, 0);
    for (size_t r = ix.n_kept; r-- > 0;) {
        for (uint64_t j = ix.row_off[r]; j < ix.row_off[r + 1]; ++j) {
            const uint32_t it = ix.row_items[j];
            if (fill[it] < m_index) ix.post_rank[ix.post_off[it] + fill[it]++] = (uint32_t)r;
        }
    }

    // 4. idf = ln(total kept pairs / sessions containing the item before truncation) * weighting (:509-512)
    ix.idf.resize(ix.n_items); ix.attr.assign(ix.n_items, (uint8_t)SRN_ATTR_FOR_SALE);   // :514-517
    for (uint32_t i = 0; i < ix.n_items; ++i)
        ix.idf[i] = std::log((double)ix.total_pairs / (double)count[i]) * idf_weighting;

    // 5. public id -> idx table for the query side (load factor <= 0.5)
    size_t tcap = 16; while (tcap < ix.n_items * 2) tcap <<= 1;
    ix.id_table.assign(tcap, IdSlot{0, kNone, 0}); ix.id_mask = (uint32_t)(tcap - 1);
    for (uint32_t i = 0; i < ix.n_items; ++i) {
        uint32_t h = (uint32_t)mix64(ix.item_id[i]) & ix.id_mask;
        while (ix.id_table[h].idx != kNone) h = (h + 1) & ix.id_mask;
        ix.id_table[h] = IdSlot{ix.item_id[i], i, 0};
    }
    return SRN_OK;
}

// ---------------------------------------------------------------------------------------------
// Shard g of an UNSHARDED flat index: the items with item_owner(id) == g, everything else restated for them -- the same bytes
// build_flat_index(..., g, n_shards) produces from the sessions (tests compare the saved files), in one O(nnz) pass instead of a
// rebuild.  The full index comes from the GPU builder or from disk, so an item-sharded deployment builds ONE index and every rank
// cuts its own shard out of it (the reference loads its production index, it does not rebuild it: vmis_index.rs:85-314).
// ---------------------------------------------------------------------------------------------
// chunks of [0, n) on up to `threads` host threads (the shard cut below walks 2.3 B row items and as many posting entries at BASELINE configs[4] scale)
template <typename F> static void parallel_chunks(uint64_t n, unsigned threads, F f) {
    threads = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(threads, n / 65536 + 1));
    if (threads == 1) { f(0u, 0ull, n); return; }
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < threads; ++t) pool.emplace_back([&, t] { f(t, n * t / threads, n * (t + 1) / threads); });
    f(0u, 0ull, n / threads);
    for (auto& th : pool) th.join();
}
static unsigned cut_threads() {
    if (const char* e = getenv("SRN_BUILD_THREADS")) return (unsigned)std::max(1, atoi(e));
    return std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
}

int shard_flat_index(const FlatIndex& full, uint32_t shard, uint32_t n_shards, FlatIndex& ix) {
    if (n_shards == 0 || shard >= n_shards) return fail(SRN_EINVAL, "shard must be < n_shards");
    if (full.n_shards != 1) return fail(SRN_EINVAL, "the source index is already a shard");
    const unsigned T = cut_threads();
    ix = FlatIndex();
    ix.n_sessions_total = full.n_sessions_total; ix.n_kept = full.n_kept; ix.m_index = full.m_index; ix.max_session_len = full.max_session_len;
    ix.idf_weighting = full.idf_weighting; ix.total_pairs = full.total_pairs; ix.shard = shard; ix.n_shards = n_shards; ix.lists_complete = full.lists_complete;
    ix.rank_to_session = full.rank_to_session;
    // owned items keep their relative (popularity) order; remap[full idx] = shard idx
    std::vector<uint32_t> remap(full.n_items, kNone);
    for (uint64_t i = 0; i < full.n_items; ++i)
        if (n_shards == 1 || item_owner(full.item_id[i], n_shards) == shard) { remap[i] = (uint32_t)ix.item_id.size(); ix.item_id.push_back(full.item_id[i]); }
    ix.n_items = ix.item_id.size();
    ix.idf.resize(ix.n_items); ix.attr.resize(ix.n_items); ix.post_off.assign(ix.n_items + 1, 0);
    for (uint64_t i = 0; i < full.n_items; ++i) { const uint32_t j = remap[i]; if (j == kNone) continue;
        ix.idf[j] = full.idf[i]; ix.attr[j] = full.attr[i]; ix.post_off[j + 1] = full.post_off[i + 1] - full.post_off[i]; }
    for (uint64_t j = 0; j < ix.n_items; ++j) ix.post_off[j + 1] += ix.post_off[j];
    ix.nnz_post = ix.post_off[ix.n_items]; ix.post_rank.resize(ix.nnz_post);
    parallel_chunks(full.n_items, T, [&](unsigned, uint64_t lo, uint64_t hi) {
        for (uint64_t i = lo; i < hi; ++i) { const uint32_t j = remap[i]; if (j == kNone) continue;
            std::copy(full.post_rank.begin() + full.post_off[i], full.post_rank.begin() + full.post_off[i + 1], ix.post_rank.begin() + ix.post_off[j]); } });
    // id_rank: rank of the public id among the owned items = order of the full index's id ranks
    { std::vector<uint32_t> by(ix.n_items); std::iota(by.begin(), by.end(), 0u);
      std::sort(by.begin(), by.end(), [&](uint32_t a, uint32_t b) { return ix.item_id[a] < ix.item_id[b]; });
      ix.id_rank.resize(ix.n_items); for (uint32_t r = 0; r < ix.n_items; ++r) ix.id_rank[by[r]] = r; }
    // row fragments, in the rows' own order: count per row (parallel), prefix, fill (parallel)
    ix.row_off.assign(ix.n_kept + 1, 0);
    parallel_chunks(full.n_kept, T, [&](unsigned, uint64_t lo, uint64_t hi) {
        for (uint64_t r = lo; r < hi; ++r) { uint64_t w = 0; for (uint64_t j = full.row_off[r]; j < full.row_off[r + 1]; ++j) w += remap[full.row_items[j]] != kNone; ix.row_off[r + 1] = w; } });
    for (uint64_t r = 0; r < full.n_kept; ++r) { ix.max_row_len = std::max<uint64_t>(ix.max_row_len, ix.row_off[r + 1]); ix.row_off[r + 1] += ix.row_off[r]; }
    ix.nnz_rows = ix.row_off[ix.n_kept]; ix.row_items.resize(ix.nnz_rows);
    parallel_chunks(full.n_kept, T, [&](unsigned, uint64_t lo, uint64_t hi) {
        for (uint64_t r = lo; r < hi; ++r) { uint64_t w = ix.row_off[r];
            for (uint64_t j = full.row_off[r]; j < full.row_off[r + 1]; ++j) { const uint32_t m = remap[full.row_items[j]]; if (m != kNone) ix.row_items[w++] = m; } } });
    size_t tcap = 16; while (tcap < ix.n_items * 2) tcap <<= 1;
    ix.id_table.assign(tcap, IdSlot{0, kNone, 0}); ix.id_mask = (uint32_t)(tcap - 1);
    for (uint32_t i = 0; i < ix.n_items; ++i) {
        uint32_t h = (uint32_t)mix64(ix.item_id[i]) & ix.id_mask;
        while (ix.id_table[h].idx != kNone) h = (h + 1) & ix.id_mask;
        ix.id_table[h] = IdSlot{ix.item_id[i], i, 0};
    }
    return SRN_OK;
}

// ---------------------------------------------------------------------------------------------
// binary save / load ("SRNFLAT4": header of u64 fields, then raw arrays; "SRNFLAT3" files -- no flags word -- still load)
// ---------------------------------------------------------------------------------------------
namespace {
template <typename T> bool wr(FILE* f, const std::vector<T>& v) {
    uint64_t n = v.size();
    return fwrite(&n, 8, 1, f) == 1 && (n == 0 || fwrite(v.data(), sizeof(T), n, f) == n);
}
template <typename T> bool rd(FILE* f, std::vector<T>& v) {
    uint64_t n = 0; if (fread(&n, 8, 1, f) != 1) return false;
    const long at = ftell(f); if (at < 0 || fseek(f, 0, SEEK_END) != 0) return false;   // (a corrupted length must not turn into a huge allocation)
    const long end = ftell(f); if (end < at || fseek(f, at, SEEK_SET) != 0 || n > (uint64_t)(end - at) / sizeof(T)) return false;
    v.resize(n); return n == 0 || fread(v.data(), sizeof(T), n, f) == n;
}
}  // namespace

int check_has_rows(const FlatIndex& ix, const char* what) {
    return ix.postings_only ? fail(SRN_EINVAL, std::string(what) + ": a postings-only view (srn_index_postings_view) holds no rows -- it serves srn_shard_group_set_postings and nothing else") : SRN_OK;
}
int save_flat_index(const FlatIndex& ix, const char* path) {
    FILE* f = fopen(path, "wb"); if (!f) return fail(SRN_EIO, std::string("cannot create ") + path);
    const char magic[8] = {'S', 'R', 'N', 'F', 'L', 'A', 'T', '4'};
    uint64_t hdr[13] = {ix.n_items, ix.n_sessions_total, ix.n_kept, ix.nnz_rows, ix.nnz_post, ix.m_index, ix.max_session_len, ix.max_row_len, ix.id_mask,
                        ix.shard, ix.n_shards, ix.total_pairs, (ix.lists_complete ? 1ull : 0ull) | (ix.viol.empty() ? 0ull : 2ull)};   // (flags: bit 0 lists complete, bit 1 the per-item viol array follows)
    bool ok = fwrite(magic, 8, 1, f) == 1 && fwrite(hdr, 8, 13, f) == 13 && fwrite(&ix.idf_weighting, 8, 1, f) == 1 &&
              wr(f, ix.item_id) && wr(f, ix.id_rank) && wr(f, ix.idf) && wr(f, ix.attr) && wr(f, ix.post_off) && wr(f, ix.post_rank) &&
              wr(f, ix.row_off) && wr(f, ix.row_items) && wr(f, ix.rank_to_session) && wr(f, ix.id_table) && (ix.viol.empty() || wr(f, ix.viol));
    ok = (fclose(f) == 0) && ok;
    return ok ? SRN_OK : fail(SRN_EIO, std::string("short write to ") + path);
}

// One O(nnz) pass over a loaded index: a truncated or corrupted file must come back as SRN_EIO, not as out-of-bounds device reads
// or a probe loop that never ends.
static const char* validate_flat_index(const FlatIndex& ix) {
    if (ix.n_shards == 0 || ix.shard >= ix.n_shards) return "bad shard header";
    if (ix.id_table.empty() || (ix.id_table.size() & (ix.id_table.size() - 1)) != 0) return "id table size is not a power of two";
    if (ix.n_items >= 0xFFFFFFF0ull || ix.n_kept >= 0xFFFFFFF0ull) return "too many items or sessions";
    if (ix.post_off[0] != 0 || ix.post_off[ix.n_items] != ix.nnz_post || ix.row_off[0] != 0 || ix.row_off[ix.n_kept] != ix.nnz_rows) return "offsets do not span the arrays";
    for (uint64_t i = 0; i < ix.n_items; ++i) {
        const uint64_t a = ix.post_off[i], b = ix.post_off[i + 1];
        if (b < a || b > ix.nnz_post || b - a > ix.m_index) return "posting offsets not monotone / list longer than m_index";
        for (uint64_t j = a; j < b; ++j) { if (ix.post_rank[j] >= ix.n_kept) return "posting entry out of range"; if (j > a && ix.post_rank[j] >= ix.post_rank[j - 1]) return "posting list not strictly descending"; }
        if (ix.id_rank[i] >= ix.n_items) return "id rank out of range";
    }
    uint64_t max_len = 0;
    for (uint64_t r = 0; r < ix.n_kept; ++r) {
        const uint64_t a = ix.row_off[r], b = ix.row_off[r + 1];
        if (b < a || b > ix.nnz_rows) return "row offsets not monotone";
        max_len = std::max(max_len, b - a);
        for (uint64_t j = a; j < b; ++j) if (ix.row_items[j] >= ix.n_items) return "row item out of range";
    }
    if (max_len > ix.max_row_len) return "row longer than max_row_len";
    uint64_t used = 0;
    for (const IdSlot& sl : ix.id_table) { if (sl.idx == kNone) continue; ++used; if (sl.idx >= ix.n_items || ix.item_id[sl.idx] != sl.key) return "id table entry does not match its item"; }
    if (used != ix.n_items || used >= ix.id_table.size()) return "id table has no empty slot or misses items";
    return nullptr;
}

int load_flat_index(const char* path, FlatIndex& ix) {
    FILE* f = fopen(path, "rb"); if (!f) return fail(SRN_EIO, std::string("cannot open ") + path);
    char magic[8]; uint64_t hdr[13] = {0}; ix = FlatIndex();
    bool ok = fread(magic, 8, 1, f) == 1 && (memcmp(magic, "SRNFLAT4", 8) == 0 || memcmp(magic, "SRNFLAT3", 8) == 0);
    const bool v3 = ok && magic[7] == '3';
    hdr[12] = 1;   // (format 3 had no flags word: every index it could hold has complete lists)
    ok = ok && fread(hdr, 8, v3 ? 12 : 13, f) == (v3 ? 12u : 13u) && fread(&ix.idf_weighting, 8, 1, f) == 1;
    const char* why = nullptr;
    if (ok) {
        ix.n_items = hdr[0]; ix.n_sessions_total = hdr[1]; ix.n_kept = hdr[2]; ix.nnz_rows = hdr[3]; ix.nnz_post = hdr[4];
        ix.m_index = hdr[5]; ix.max_session_len = hdr[6]; ix.max_row_len = hdr[7]; ix.id_mask = (uint32_t)hdr[8];
        ix.shard = (uint32_t)hdr[9]; ix.n_shards = (uint32_t)hdr[10]; ix.total_pairs = hdr[11]; ix.lists_complete = (hdr[12] & 1) != 0;
        ok = rd(f, ix.item_id) && rd(f, ix.id_rank) && rd(f, ix.idf) && rd(f, ix.attr) && rd(f, ix.post_off) && rd(f, ix.post_rank) &&
             rd(f, ix.row_off) && rd(f, ix.row_items) && rd(f, ix.rank_to_session) && rd(f, ix.id_table) && ((hdr[12] & 2) == 0 || (rd(f, ix.viol) && ix.viol.size() == ix.n_items && !ix.lists_complete));
        ok = ok && ix.item_id.size() == ix.n_items && ix.id_rank.size() == ix.n_items && ix.idf.size() == ix.n_items && ix.attr.size() == ix.n_items &&
             ix.post_off.size() == ix.n_items + 1 && ix.post_rank.size() == ix.nnz_post && ix.row_off.size() == ix.n_kept + 1 &&
             ix.row_items.size() == ix.nnz_rows && ix.rank_to_session.size() == ix.n_kept && ix.id_table.size() == (size_t)ix.id_mask + 1;
        if (ok) { why = validate_flat_index(ix); ok = why == nullptr; }
    }
    fclose(f);
    return ok ? SRN_OK : fail(SRN_EIO, std::string("not a valid SRNFLAT index: ") + path + (why ? std::string(" (") + why + ")" : std::string()));
}

}  // namespace srn

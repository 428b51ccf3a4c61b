// =====================================================================================
// Evolving-session store and the /v1/recommend handler body: what sits between a request and predict().
//
// The reference keeps each visitor's evolving session in a RocksDB opened with a TTL, keyed by the u128 built from the MD5
// digest of the request's session_id string, value = (items, epoch seconds of the last event)
// (src/sessions/mod.rs:7-77); the handler reads it, appends the clicked item unless it repeats the last one, drops the
// oldest item beyond max_items_in_session, writes it back and calls predict (src/endpoints/recommend_resource.rs:20-65).
// Here the store is an in-memory table (64 independently locked stripes; a visitor is pinned to one serving pod by the
// session_id affinity the reference's handler comment describes, so nothing has to survive the process) with the same two
// clocks: a session idle for more than idle_secs reads as empty (mod.rs:47-52: 20 minutes, hard-coded there), and an
// entry older than ttl_secs is dropped (RocksDB's TTL compaction).  srn_recommend() is the handler body with the
// predict call routed through the dynamic batcher.  No HTTP here: the web framework stays the host application's.
// =====================================================================================
#include <chrono>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "srn_internal.h"

using namespace srn;

namespace {
template <typename F> int guarded(F f) {
    try { return f(); }
    catch (const std::bad_alloc&) { return fail(SRN_ENOMEM, "out of host memory"); }
    catch (const std::exception& e) { return fail(SRN_EINVAL, std::string("internal error: ") + e.what()); }
    catch (...) { return fail(SRN_EINVAL, "internal error"); }
}

// ---- MD5 (RFC 1321), for the session key only: same digest, same u128, as md5::compute + uuid::Builder::from_bytes ----
struct Md5 {
    uint32_t a = 0x67452301u, b = 0xefcdab89u, c = 0x98badcfeu, d = 0x10325476u;
    static uint32_t rotl(uint32_t x, int s) { return (x << s) | (x >> (32 - s)); }
    void block(const uint8_t* p) {
        static const uint32_t K[64] = {
            0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8, 0x8b44f7af,
            0xffff5bb1, 0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821, 0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa,
            0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8, 0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8,
            0x676f02d9, 0x8d2a4c8a, 0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70,
            0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665, 0xf4292244, 0x432aff97,
            0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d, 0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1,
            0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
        static const int S[16] = {7, 12, 17, 22, 5, 9, 14, 20, 4, 11, 16, 23, 6, 10, 15, 21};
        uint32_t w[16];
        for (int i = 0; i < 16; ++i) w[i] = (uint32_t)p[4 * i] | (uint32_t)p[4 * i + 1] << 8 | (uint32_t)p[4 * i + 2] << 16 | (uint32_t)p[4 * i + 3] << 24;
        uint32_t A = a, B = b, C = c, D = d;
        for (int i = 0; i < 64; ++i) {
            uint32_t f; int g;
            if (i < 16)      { f = (B & C) | (~B & D); g = i; }
            else if (i < 32) { f = (D & B) | (~D & C); g = (5 * i + 1) & 15; }
            else if (i < 48) { f = B ^ C ^ D;          g = (3 * i + 5) & 15; }
            else             { f = C ^ (B | ~D);       g = (7 * i) & 15; }
            const uint32_t t = D; D = C; C = B;
            B = B + rotl(A + f + K[i] + w[g], S[(i >> 4) * 4 + (i & 3)]);
            A = t;
        }
        a += A; b += B; c += C; d += D;
    }
    void digest(const uint8_t* msg, size_t n, uint8_t out[16]) {
        size_t i = 0;
        for (; i + 64 <= n; i += 64) block(msg + i);
        uint8_t tail[128] = {0};
        const size_t r = n - i;
        std::memcpy(tail, msg + i, r);
        tail[r] = 0x80;
        const size_t tl = r < 56 ? 64 : 128;
        const uint64_t bits = (uint64_t)n * 8;
        for (int j = 0; j < 8; ++j) tail[tl - 8 + j] = (uint8_t)(bits >> (8 * j));
        block(tail);
        if (tl == 128) block(tail + 64);
        const uint32_t v[4] = {a, b, c, d};
        for (int j = 0; j < 16; ++j) out[j] = (uint8_t)(v[j >> 2] >> (8 * (j & 3)));
    }
};

struct Key { uint64_t hi, lo; bool operator==(const Key& o) const { return hi == o.hi && lo == o.lo; } };
struct KeyHash { size_t operator()(const Key& k) const { return (size_t)(k.lo ^ (k.hi * 0x9E3779B97F4A7C15ull)); } };   // the key is already a digest
struct Entry { std::vector<uint64_t> items; uint64_t epoch_secs; };
constexpr int kStripes = 64;
struct Stripe { std::mutex mu; std::unordered_map<Key, Entry, KeyHash> map; uint64_t ops = 0; };

uint64_t wall_secs() {
    return (uint64_t)std::chrono::duration_cast<std::chrono::seconds>(std::chrono::system_clock::now().time_since_epoch()).count();
}
}  // namespace

struct srn_session_store {
    uint64_t ttl_secs, idle_secs;
    Stripe stripes[kStripes];
    Stripe& stripe_of(const Key& k) { return stripes[(k.lo ^ (k.lo >> 29) ^ k.hi) & (kStripes - 1)]; }
    static void sweep_locked(Stripe& s, uint64_t now, uint64_t ttl) {
        for (auto it = s.map.begin(); it != s.map.end();) it = (now > it->second.epoch_secs && now - it->second.epoch_secs > ttl) ? s.map.erase(it) : ++it;
    }
};

extern "C" {

int srn_session_key(const char* session_id, size_t len, uint64_t* key_hi, uint64_t* key_lo) {
    if (!session_id && len) return fail(SRN_EINVAL, "srn_session_key: null session_id");
    if (!key_hi || !key_lo) return fail(SRN_EINVAL, "srn_session_key: null output");
    uint8_t d[16];
    Md5().digest((const uint8_t*)session_id, len, d);
    uint64_t hi = 0, lo = 0;                       // uuid::Builder::from_bytes(..).as_u128(): the digest read big-endian
    for (int i = 0; i < 8; ++i) { hi = hi << 8 | d[i]; lo = lo << 8 | d[8 + i]; }
    *key_hi = hi; *key_lo = lo;
    return SRN_OK;
}

int srn_session_store_create(uint64_t ttl_secs, uint64_t idle_secs, srn_session_store_t** out) {
    return guarded([&] {
        if (!out) return fail(SRN_EINVAL, "srn_session_store_create: null output");
        auto* s = new srn_session_store;
        s->ttl_secs = ttl_secs ? ttl_secs : 30 * 60;   // the serving binary's default session TTL is 30 minutes (src/config.rs)
        s->idle_secs = idle_secs ? idle_secs : 20 * 60;  // src/sessions/mod.rs:35
        *out = s;
        return (int)SRN_OK;
    });
}

void srn_session_store_free(srn_session_store_t* s) { delete s; }

int srn_session_store_get(srn_session_store_t* s, uint64_t key_hi, uint64_t key_lo, uint64_t now_secs,
                          uint64_t* out_items, size_t cap, size_t* out_n) {
    return guarded([&] {
        if (!s || !out_n || (cap && !out_items)) return fail(SRN_EINVAL, "srn_session_store_get: null argument");
        const uint64_t now = now_secs ? now_secs : wall_secs();
        const Key k{key_hi, key_lo};
        Stripe& st = s->stripe_of(k);
        std::lock_guard<std::mutex> g(st.mu);
        *out_n = 0;
        auto it = st.map.find(k);
        if (it == st.map.end()) return (int)SRN_OK;
        const Entry& e = it->second;
        // mod.rs:46-52 (the reference subtracts unsigned: a clock that stepped backwards would panic there; here it reads as "fresh")
        if (now > e.epoch_secs && now - e.epoch_secs > s->idle_secs) return (int)SRN_OK;
        if (e.items.size() > cap) return fail(SRN_ERANGE, "srn_session_store_get: output buffer too small");
        std::memcpy(out_items, e.items.data(), e.items.size() * sizeof(uint64_t));
        *out_n = e.items.size();
        return (int)SRN_OK;
    });
}

int srn_session_store_update(srn_session_store_t* s, uint64_t key_hi, uint64_t key_lo, uint64_t now_secs,
                             const uint64_t* items, size_t n) {
    return guarded([&] {
        if (!s || (n && !items)) return fail(SRN_EINVAL, "srn_session_store_update: null argument");
        const uint64_t now = now_secs ? now_secs : wall_secs();
        const Key k{key_hi, key_lo};
        Stripe& st = s->stripe_of(k);
        std::lock_guard<std::mutex> g(st.mu);
        Entry& e = st.map[k];
        e.items.assign(items, items + n);
        e.epoch_secs = now;
        if ((++st.ops & 0xFFF) == 0) srn_session_store::sweep_locked(st, now, s->ttl_secs);   // amortised expiry, like a compaction
        return (int)SRN_OK;
    });
}

int srn_session_store_sweep(srn_session_store_t* s, uint64_t now_secs, uint64_t* n_live) {
    return guarded([&] {
        if (!s) return fail(SRN_EINVAL, "srn_session_store_sweep: null store");
        const uint64_t now = now_secs ? now_secs : wall_secs();
        uint64_t live = 0;
        for (Stripe& st : s->stripes) {
            std::lock_guard<std::mutex> g(st.mu);
            srn_session_store::sweep_locked(st, now, s->ttl_secs);
            live += st.map.size();
        }
        if (n_live) *n_live = live;
        return (int)SRN_OK;
    });
}

int srn_recommend(srn_batcher_t* b, srn_session_store_t* s, const char* session_id, size_t session_id_len, uint64_t item_id,
                  int user_consent, size_t max_items_in_session, uint64_t now_secs, uint64_t* out_ids, double* out_scores,
                  size_t* out_n) {
    return guarded([&] {
        if (!b || !out_ids || !out_n) return fail(SRN_EINVAL, "srn_recommend: null argument");
        if (user_consent && !s) return fail(SRN_EINVAL, "srn_recommend: user_consent needs a session store");
        if (max_items_in_session == 0) return fail(SRN_EINVAL, "srn_recommend: max_items_in_session must be > 0");
        std::vector<uint64_t> items;
        if (user_consent) {                                  // recommend_resource.rs:39-52
            uint64_t hi, lo;
            int rc = srn_session_key(session_id, session_id_len, &hi, &lo);
            if (rc != SRN_OK) return rc;
            const uint64_t now = now_secs ? now_secs : wall_secs();
            const Key k{hi, lo};
            Stripe& st = s->stripe_of(k);
            std::lock_guard<std::mutex> g(st.mu);            // read-modify-write under one lock (the reference's get + put can interleave)
            Entry& e = st.map[k];
            if (!e.items.empty() && now > e.epoch_secs && now - e.epoch_secs > s->idle_secs) e.items.clear();
            if (e.items.empty()) e.items.push_back(item_id);
            else if (e.items.back() != item_id) {
                e.items.push_back(item_id);
                if (e.items.size() > max_items_in_session) e.items.erase(e.items.begin());
            }
            e.epoch_secs = now;
            items = e.items;
            if ((++st.ops & 0xFFF) == 0) srn_session_store::sweep_locked(st, now, s->ttl_secs);
        } else {
            items.push_back(item_id);                        // recommend_resource.rs:53-55
        }
        // a stored session may be longer than max_items_in_session if the limit was lowered since: the reference passes it on as is
        std::vector<double> scores_local;
        double* sc = out_scores;
        if (!sc) { size_t hm = 0; srn_batcher_how_many(b, &hm); scores_local.resize(hm ? hm : 1); sc = scores_local.data(); }
        return srn_batcher_predict(b, items.data(), items.size(), out_ids, sc, out_n);
    });
}

}  // extern "C"

// Deterministic synthetic workload generator for the BASELINE.json configs 2-5:
// training sessions + held-out evolving-session queries shaped like the reference's production data.
//
//  * session lengths: piece-wise linear inverse CDF through the percentiles the reference hard-codes
//    for bol.com traffic (src/vmisknn/vmis_index.rs:116-126): p25=2 p50=3 p75=6 p90=10 p95=14 p99=27
//    p99.5=34, capped at 34 (so the p99.5 training-session filter keeps everything);
//  * items: Zipf(alpha) over popularity ranks (Vose alias sampling), public id = splitmix64(rank)
//    masked to 48 bits (forces the u64 id path), collisions re-salted;
//  * timestamps: a seeded permutation of [T0, T0 + n_sessions) -- unique per session, so the
//    reference's tie-dependent behaviour never triggers;
//  * queries: held-out sessions from another seed stream, every prefix 1..len-1 of each session
//    truncated to its last `max_items` items, exactly the evaluator loop (src/bin/evaluator.rs:46-56).
//
// Everything is a pure function of (seed, parameters): chunks of 4096 sessions own their RNG stream,
// so the output does not depend on the thread count.  Plain C++ (no GPU, no product or oracle code).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

namespace {

struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed) {}
    uint64_t next() { uint64_t z = (s += 0x9E3779B97F4A7C15ULL); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
                      z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); }
    double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    uint64_t below(uint64_t n) { return (uint64_t)(((unsigned __int128)next() * n) >> 64); }
};
static uint64_t splitmix_once(uint64_t x) { Rng r(x); return r.next(); }

static const double kQ[] = {0.0, 0.25, 0.50, 0.75, 0.90, 0.95, 0.99, 0.995, 1.0};
static const double kV[] = {2.0, 2.0, 3.0, 6.0, 10.0, 14.0, 27.0, 34.0, 34.0};
static uint32_t draw_length(Rng& r) {
    const double u = r.unit();
    int i = 0; while (i < 7 && u >= kQ[i + 1]) ++i;
    const double v = kV[i] + (kV[i + 1] - kV[i]) * (u - kQ[i]) / (kQ[i + 1] - kQ[i]);
    return (uint32_t)std::min(34.0, std::floor(v));
}

struct Alias { std::vector<double> prob; std::vector<uint32_t> alias; };
static void build_alias(size_t n, double alpha, Alias& a) {
    std::vector<double> w(n); double sum = 0;
    for (size_t i = 0; i < n; ++i) { w[i] = std::pow((double)(i + 1), -alpha); sum += w[i]; }
    a.prob.assign(n, 0); a.alias.assign(n, 0);
    std::vector<uint32_t> small, large; small.reserve(n); large.reserve(n);
    for (size_t i = 0; i < n; ++i) { w[i] = w[i] * (double)n / sum; (w[i] < 1.0 ? small : large).push_back((uint32_t)i); }
    while (!small.empty() && !large.empty()) {
        const uint32_t s = small.back(); small.pop_back(); const uint32_t l = large.back();
        a.prob[s] = w[s]; a.alias[s] = l; w[l] = (w[l] + w[s]) - 1.0;
        if (w[l] < 1.0) { large.pop_back(); small.push_back(l); }
    }
    for (uint32_t i : large) a.prob[i] = 1.0;
    for (uint32_t i : small) a.prob[i] = 1.0;
}
static inline uint32_t draw_item(const Alias& a, Rng& r) {
    const uint64_t x = r.next(); const size_t n = a.prob.size();
    const uint32_t i = (uint32_t)(((unsigned __int128)x * n) >> 64);
    const double u = (double)((x * 0x9E3779B97F4A7C15ULL) >> 11) * (1.0 / 9007199254740992.0);
    return u < a.prob[i] ? i : a.alias[i];
}
static void public_ids(size_t n, std::vector<uint64_t>& ids) {
    ids.resize(n);
    for (size_t i = 0; i < n; ++i) ids[i] = splitmix_once(i + 1) & ((1ULL << 48) - 1);
    for (uint64_t salt = 1;; ++salt) {   // re-salt the (rare) 48-bit collisions until all ids are distinct
        std::vector<uint32_t> o(n); for (size_t i = 0; i < n; ++i) o[i] = (uint32_t)i;
        std::sort(o.begin(), o.end(), [&](uint32_t a, uint32_t b) { return ids[a] != ids[b] ? ids[a] < ids[b] : a < b; });
        bool clash = false;
        for (size_t i = 1; i < n; ++i) if (ids[o[i]] == ids[o[i - 1]]) { clash = true; ids[o[i]] = splitmix_once((uint64_t)o[i] + 1 + salt * 0x1000000000ULL) & ((1ULL << 48) - 1); }
        if (!clash) break;
    }
}

constexpr size_t CHUNK = 4096;

struct Synth {
    std::vector<uint64_t> off, items; std::vector<uint32_t> ts;   // training sessions (rows ascending, de-duplicated)
    std::vector<uint64_t> q_items; std::vector<uint32_t> q_off;    // queries
};

static void gen_chunk(uint64_t seed, uint64_t stream, size_t chunk, const Alias& al, const std::vector<uint64_t>& ids,
                      bool training, std::vector<uint32_t>& lens, std::vector<uint64_t>& out) {
    Rng r(splitmix_once(seed ^ (stream * 0xD1B54A32D192ED03ULL)) ^ splitmix_once(chunk + 0x51ED270B));
    lens.clear(); out.clear();
    uint64_t buf[64];
    for (size_t s = 0; s < CHUNK; ++s) {
        const uint32_t len = draw_length(r); uint32_t n = 0;
        // training rows hold `len` DISTINCT items (re-draw duplicates, bounded), so the de-duplicated
        // lengths keep the target percentiles; evolving sessions keep repeats except immediate ones
        for (uint32_t tries = 0; n < len && tries < 4 * len + 16; ++tries) {
            const uint64_t id = ids[draw_item(al, r)];
            if (training) { bool dup = false; for (uint32_t t = 0; t < n; ++t) dup |= buf[t] == id; if (!dup) buf[n++] = id; }
            else if (n == 0 || buf[n - 1] != id) buf[n++] = id;   // serving drops an immediate repeat (recommend_resource.rs:42)
        }
        if (training) std::sort(buf, buf + n);
        lens.push_back(n); out.insert(out.end(), buf, buf + n);
    }
}

}  // namespace

extern "C" {

// Training sessions until the realised (de-duplicated) interaction count reaches n_interactions.
void* srn_synth_training(uint64_t seed, uint64_t n_interactions, uint64_t n_items, double alpha, uint32_t t0, int threads) {
    Synth* S = new Synth(); Alias al; build_alias(n_items, alpha, al);
    std::vector<uint64_t> ids; public_ids(n_items, ids);
    if (threads < 1) threads = 1;
    S->off.push_back(0);
    size_t chunk0 = 0;
    while (S->items.size() < n_interactions) {
        const size_t remaining = n_interactions - S->items.size();
        const size_t wave = std::max<size_t>(1, std::min<size_t>((size_t)threads * 4, remaining / (CHUNK * 5) + 1));
        std::vector<std::vector<uint32_t>> lens(wave); std::vector<std::vector<uint64_t>> outs(wave);
        std::atomic<size_t> next{0}; std::vector<std::thread> pool;
        auto work = [&]() { for (size_t c; (c = next.fetch_add(1)) < wave;) gen_chunk(seed, 0, chunk0 + c, al, ids, true, lens[c], outs[c]); };
        for (int t = 1; t < threads; ++t) pool.emplace_back(work);
        work(); for (auto& t : pool) t.join();
        for (size_t c = 0; c < wave && S->items.size() < n_interactions; ++c) {
            size_t p = 0;
            for (uint32_t l : lens[c]) {
                if (S->items.size() >= n_interactions) break;
                S->items.insert(S->items.end(), outs[c].begin() + p, outs[c].begin() + p + l); p += l;
                S->off.push_back(S->items.size());
            }
        }
        chunk0 += wave;
    }
    const size_t n = S->off.size() - 1;
    S->ts.resize(n);
    for (size_t i = 0; i < n; ++i) S->ts[i] = t0 + (uint32_t)i;
    Rng r(splitmix_once(seed ^ 0x7157A3B5));   // unique timestamps: Fisher-Yates permutation of [t0, t0+n)
    for (size_t i = n; i > 1; --i) std::swap(S->ts[i - 1], S->ts[r.below(i)]);
    return S;
}

// Evaluator-style queries from n_sessions held-out sessions (stream 1): prefixes 1..len-1, last max_items items.
void* srn_synth_queries(uint64_t seed, uint64_t n_sessions, uint64_t n_items, double alpha, uint32_t max_items, int threads) {
    Synth* S = new Synth(); Alias al; build_alias(n_items, alpha, al);
    std::vector<uint64_t> ids; public_ids(n_items, ids);
    if (threads < 1) threads = 1;
    const size_t chunks = (n_sessions + CHUNK - 1) / CHUNK;
    std::vector<std::vector<uint32_t>> lens(chunks); std::vector<std::vector<uint64_t>> outs(chunks);
    std::atomic<size_t> next{0}; std::vector<std::thread> pool;
    auto work = [&]() { for (size_t c; (c = next.fetch_add(1)) < chunks;) gen_chunk(seed, 1, c, al, ids, false, lens[c], outs[c]); };
    for (int t = 1; t < threads; ++t) pool.emplace_back(work);
    work(); for (auto& t : pool) t.join();
    S->q_off.push_back(0); size_t sess = 0;
    for (size_t c = 0; c < chunks; ++c) {
        size_t p = 0;
        for (uint32_t l : lens[c]) {
            if (sess++ >= n_sessions) break;
            const uint64_t* ev = outs[c].data() + p; p += l;
            for (uint32_t state = 1; state < l; ++state) {
                const uint32_t start = state > max_items ? state - max_items : 0;
                S->q_items.insert(S->q_items.end(), ev + start, ev + state);
                S->q_off.push_back((uint32_t)S->q_items.size());
            }
        }
    }
    return S;
}

uint64_t srn_synth_n_sessions(void* h) { return ((Synth*)h)->ts.size(); }
uint64_t srn_synth_nnz(void* h) { return ((Synth*)h)->items.size(); }
uint64_t srn_synth_n_queries(void* h) { return ((Synth*)h)->q_off.empty() ? 0 : ((Synth*)h)->q_off.size() - 1; }
uint64_t srn_synth_q_nnz(void* h) { return ((Synth*)h)->q_items.size(); }
void srn_synth_copy_training(void* h, uint64_t* off, uint64_t* items, uint32_t* ts) {
    Synth* S = (Synth*)h;
    memcpy(off, S->off.data(), S->off.size() * 8); memcpy(items, S->items.data(), S->items.size() * 8); memcpy(ts, S->ts.data(), S->ts.size() * 4);
}
void srn_synth_copy_queries(void* h, uint64_t* q_items, uint32_t* q_off) {
    Synth* S = (Synth*)h;
    memcpy(q_items, S->q_items.data(), S->q_items.size() * 8); memcpy(q_off, S->q_off.data(), S->q_off.size() * 4);
}
void srn_synth_free(void* h) { delete (Synth*)h; }

}  // extern "C"

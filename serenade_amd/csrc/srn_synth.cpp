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
#include <cstdio>
#include <cstring>
#include <string>
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
    std::vector<uint64_t> q_next;                                   // the held-out item that followed each query's prefix (evaluator.rs:75: next_items[0], what Mrr / HitRate look at)
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
                S->q_next.push_back(ev[state]);
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
void srn_synth_copy_next(void* h, uint64_t* q_next) { Synth* S = (Synth*)h; memcpy(q_next, S->q_next.data(), S->q_next.size() * 8); }
void srn_synth_free(void* h) { delete (Synth*)h; }

// ---------------------------------------------------------------------------------------------------------------------------------
// A stand-in for the OFFLINE PRODUCER of the reference's pre-built index (the Spark job behind VMISIndex::new, src/vmisknn/vmis_index.rs:85-314; its code is not in the
// reference repository): per item the m_index most recent sessions that hold it, idf, flags -- and the files in the Avro layout srn_index_new_from_avro reads
// (schemas :184-192, :249-255; codec "null").  What the producer is free to do differently from this library's builder, and what the knobs emulate:
//   * sessions of more than max_len items go to the session index only, into no list (the CSV path does the same: :79, :452);
//   * tie_mode -- the order among sessions of EQUAL timestamp when a list is cut: 0 larger SessionIndex first (this library's builder), 1 smaller first,
//     2 larger first for every other item and smaller for the rest, 3 a different pseudo-random order PER ITEM (a window function without a tie-break column).
// Workload generator code: no product or oracle code involved.
struct Producer { std::vector<uint64_t> item_ids, list_off; std::vector<uint32_t> list_sessions; std::vector<double> idf; };

void* srn_synth_producer(const uint64_t* off, const uint64_t* items, const uint32_t* ts, uint64_t n_sessions, uint64_t m_index, uint64_t max_len, double idf_weighting, int tie_mode) {
    struct Pair { uint64_t id, key; uint32_t s; };
    std::vector<Pair> pairs; uint64_t total = 0;
    for (uint64_t s = 0; s < n_sessions; ++s) if (off[s + 1] - off[s] <= max_len) total += off[s + 1] - off[s];
    pairs.reserve(total);
    for (uint64_t s = 0; s < n_sessions; ++s) {
        if (off[s + 1] - off[s] > max_len) continue;
        for (uint64_t j = off[s]; j < off[s + 1]; ++j) {
            const uint64_t id = items[j];
            uint32_t tie;
            switch (tie_mode) {
                case 0: tie = ~(uint32_t)s; break;
                case 1: tie = (uint32_t)s; break;
                case 2: tie = (splitmix_once(id) & 1) ? (uint32_t)s : ~(uint32_t)s; break;
                default: tie = (uint32_t)splitmix_once(id * 0x9E3779B97F4A7C15ULL ^ s); break;
            }
            pairs.push_back(Pair{id, ((uint64_t)(~ts[s]) << 32) | tie, (uint32_t)s});   // ascending key = most recent first
        }
    }
    std::sort(pairs.begin(), pairs.end(), [](const Pair& a, const Pair& b) { return a.id != b.id ? a.id < b.id : a.key != b.key ? a.key < b.key : a.s < b.s; });
    Producer* P = new Producer(); P->list_off.push_back(0);
    for (size_t i = 0; i < pairs.size();) {
        size_t j = i; while (j < pairs.size() && pairs[j].id == pairs[i].id) ++j;
        P->item_ids.push_back(pairs[i].id);
        for (size_t e = i; e < j && e - i < m_index; ++e) P->list_sessions.push_back(pairs[e].s);
        P->list_off.push_back(P->list_sessions.size());
        P->idf.push_back(std::log((double)total / (double)(j - i)) * idf_weighting);   // (prepare_hashmap's formula, vmis_index.rs:509-512)
        i = j;
    }
    return P;
}
uint64_t srn_synth_producer_n_items(void* h) { return ((Producer*)h)->item_ids.size(); }
uint64_t srn_synth_producer_nnz(void* h) { return ((Producer*)h)->list_sessions.size(); }
void srn_synth_producer_copy(void* h, uint64_t* item_ids, uint64_t* list_off, uint32_t* list_sessions, double* idf) {
    Producer* P = (Producer*)h;
    memcpy(item_ids, P->item_ids.data(), P->item_ids.size() * 8); memcpy(list_off, P->list_off.data(), P->list_off.size() * 8);
    memcpy(list_sessions, P->list_sessions.data(), P->list_sessions.size() * 4); memcpy(idf, P->idf.data(), P->idf.size() * 8);
}
void srn_synth_producer_free(void* h) { delete (Producer*)h; }

}  // extern "C"

namespace {
struct AvroOut {
    FILE* f = nullptr; std::vector<uint8_t> blk; uint64_t in_blk = 0; uint8_t sync[16];
    static void zz(std::vector<uint8_t>& o, int64_t v) { uint64_t n = ((uint64_t)v << 1) ^ (uint64_t)(v >> 63); while (n >= 0x80) { o.push_back((uint8_t)(n | 0x80)); n >>= 7; } o.push_back((uint8_t)n); }
    bool open(const std::string& path, const char* schema) {
        f = fopen(path.c_str(), "wb"); if (!f) return false;
        for (int i = 0; i < 16; ++i) sync[i] = (uint8_t)(i * 37 + 11);
        std::vector<uint8_t> h = {'O', 'b', 'j', 1};
        zz(h, 2);
        auto str = [&](const char* t) { const size_t n = strlen(t); zz(h, (int64_t)n); h.insert(h.end(), t, t + n); };
        str("avro.schema"); str(schema); str("avro.codec"); str("null");
        zz(h, 0); h.insert(h.end(), sync, sync + 16);
        return fwrite(h.data(), 1, h.size(), f) == h.size();
    }
    bool flush() {
        if (!in_blk) return true;
        std::vector<uint8_t> h; zz(h, (int64_t)in_blk); zz(h, (int64_t)blk.size());
        const bool ok = fwrite(h.data(), 1, h.size(), f) == h.size() && fwrite(blk.data(), 1, blk.size(), f) == blk.size() && fwrite(sync, 1, 16, f) == 16;
        blk.clear(); in_blk = 0; return ok;
    }
    bool end_record() { ++in_blk; return blk.size() < (1u << 20) || flush(); }
    bool close() { const bool ok = flush(); const bool ok2 = fclose(f) == 0; f = nullptr; return ok && ok2; }
};
}  // namespace

extern "C" {

// <base>/itemindex/part-N.avro + <base>/sessionindex/part-N.avro (the directories must exist); every session of the arrays gets a session record (SessionIndex = its position)
int srn_synth_write_avro(const char* base, void* producer, const uint64_t* off, const uint64_t* items, const uint32_t* ts, uint64_t n_sessions, int n_files) {
    const Producer* P = (const Producer*)producer;
    if (n_files < 1) n_files = 1;
    static const char* ITEM = "{\"type\": \"record\", \"name\": \"ItemIndex\", \"fields\": [{\"name\": \"ItemId\", \"type\": \"long\"}, {\"name\": \"session_indices_time_ordered\", \"type\": {\"type\": \"array\", \"items\": \"int\"}}, "
                              "{\"name\": \"idf\", \"type\": \"double\"}, {\"name\": \"ForSale\", \"type\": \"boolean\"}, {\"name\": \"IsAdult\", \"type\": \"boolean\"}]}";
    static const char* SESS = "{\"type\": \"record\", \"name\": \"SessionIndex\", \"fields\": [{\"name\": \"SessionIndex\", \"type\": \"int\"}, {\"name\": \"item_ids_asc\", \"type\": {\"type\": \"array\", \"items\": \"long\"}}, {\"name\": \"Time\", \"type\": \"int\"}]}";
    const std::string b = base;
    for (int part = 0; part < n_files; ++part) {
        AvroOut o; if (!o.open(b + "/itemindex/part-" + std::to_string(part) + ".avro", ITEM)) return -1;
        const size_t ni = P->item_ids.size();
        for (size_t i = ni * (size_t)part / n_files; i < ni * (size_t)(part + 1) / n_files; ++i) {
            AvroOut::zz(o.blk, (int64_t)P->item_ids[i]);
            const uint64_t a = P->list_off[i], e = P->list_off[i + 1];
            if (e > a) { AvroOut::zz(o.blk, (int64_t)(e - a)); for (uint64_t j = a; j < e; ++j) AvroOut::zz(o.blk, (int64_t)P->list_sessions[j]); }
            AvroOut::zz(o.blk, 0);
            uint8_t d[8]; memcpy(d, &P->idf[i], 8); o.blk.insert(o.blk.end(), d, d + 8);
            o.blk.push_back(1); o.blk.push_back(0);   // ForSale, IsAdult
            if (!o.end_record()) return -1;
        }
        if (!o.close()) return -1;
        AvroOut q; if (!q.open(b + "/sessionindex/part-" + std::to_string(part) + ".avro", SESS)) return -1;
        for (uint64_t s = n_sessions * (uint64_t)part / n_files; s < n_sessions * (uint64_t)(part + 1) / n_files; ++s) {
            AvroOut::zz(q.blk, (int64_t)s);
            if (off[s + 1] > off[s]) { AvroOut::zz(q.blk, (int64_t)(off[s + 1] - off[s])); for (uint64_t j = off[s]; j < off[s + 1]; ++j) AvroOut::zz(q.blk, (int64_t)items[j]); }
            AvroOut::zz(q.blk, 0);
            AvroOut::zz(q.blk, (int64_t)ts[s]);
            if (!q.end_record()) return -1;
        }
        if (!q.close()) return -1;
    }
    return 0;
}

}  // extern "C"

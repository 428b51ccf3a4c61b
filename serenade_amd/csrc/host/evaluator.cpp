// serenade-hip evaluator: host program over the C ABI that mirrors the reference's offline evaluation tools.
//
//   evaluator <config.toml> [--per-call]        src/bin/evaluator.rs:9-90: index from the training file, every prefix of every
//                                               test session -> predict -> 8 metrics + latency percentiles
//   evaluator --evaluate-file <train.txt> <predictions.txt>
//                                               src/bin/evaluate_file.rs:8-52: score a "recos;next_items" file (no GPU needed)
//   evaluator --metrics-selftest                the reference's metric known answers (src/metrics/*.rs test modules)
//
// The GPU does the predictions (libserenade_hip.so); this file is plain host C++: config parsing, TSV readers
// (src/io.rs), the metrics (src/metrics/*.rs) and the report format (src/metrics/evaluation_reporter.rs).
// By default all prefixes go to the GPU as ONE srn_predict_batch call (that is the point of the port); --per-call issues one
// srn_predict per prefix like the reference loop and reports per-call latency percentiles (exact, not t-digest estimates).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <set>
#include <sstream>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../../include/serenade_hip.h"

namespace {

using Items = std::vector<uint64_t>;

// ---- metrics (src/metrics/*.rs) --------------------------------------------------------------------------------
struct Reporter {   // evaluation_reporter.rs:12-117
    size_t length; long n = 0;
    double mrr = 0, ndcg = 0, hit = 0, pop = 0, prec = 0, rec = 0;
    std::unordered_map<uint64_t, int> freq; int max_freq = 0; size_t unique_training_items = 0;
    std::unordered_set<uint64_t> covered;

    Reporter(const std::vector<uint64_t>& training_items, size_t len) : length(len) {
        for (uint64_t it : training_items) { int& c = freq[it]; ++c; max_freq = std::max(max_freq, c); }   // popularity.rs:20-29
        unique_training_items = freq.size();                                                               // coverage.rs:17-25
    }
    static double dcg(const Items& top, const std::unordered_set<uint64_t>& next) {                        // ndcg.rs:13-27
        double r = 0;
        for (size_t i = 0; i < top.size(); ++i) if (next.count(top[i])) r += i == 0 ? 1.0 : 1.0 / std::log2((double)i + 1.0);
        return r;
    }
    void add(const Items& recommendations, const Items& next_items) {
        ++n;
        Items top(recommendations.begin(), recommendations.begin() + std::min(recommendations.size(), length));
        const uint64_t next_item = next_items[0];
        auto pos = std::find(top.begin(), top.end(), next_item);
        if (pos != top.end()) { mrr += 1.0 / (double)(pos - top.begin() + 1); hit += 1.0; }                // mrr.rs:24-33, hitrate.rs:24-33
        std::unordered_set<uint64_t> next_set(next_items.begin(), next_items.end()), top_set(top.begin(), top.end());
        Items top_next(next_items.begin(), next_items.begin() + std::min(next_items.size(), length));
        ndcg += dcg(top, next_set) / dcg(top_next, next_set);                                              // ndcg.rs:42-56
        size_t inter = 0; for (uint64_t x : top_set) inter += next_set.count(x);
        prec += (double)inter / (double)length;                                                            // precision.rs:31-43
        rec += (double)inter / (double)next_items.size();                                                  // recall.rs:31-44 (len incl. duplicates)
        if (!top_set.empty()) { double s = 0; for (uint64_t x : top_set) { auto f = freq.find(x); if (f != freq.end()) s += (double)f->second / (double)max_freq; }
                                pop += s / (double)top_set.size(); }                                        // popularity.rs:41-58
        for (uint64_t x : top) covered.insert(x);                                                          // coverage.rs:29-38
    }
    double avg(double s) const { return n > 0 ? s / (double)n : 0.0; }
    double f1() const { const double p = avg(prec), r = avg(rec), f = 2.0 * (p * r) / (p + r); return std::isnan(f) ? 0.0 : f; }   // f1score.rs:27-36
    double coverage() const { return unique_training_items ? (double)covered.size() / (double)unique_training_items : 0.0; }
    std::string name() const {
        char b[256]; snprintf(b, sizeof b, "qty_evaluations,Mrr@%zu,Ndcg@%zu,HitRate@%zu,Popularity@%zu,Precision@%zu,Coverage@%zu,Recall@%zu,F1score@%zu",
                              length, length, length, length, length, length, length, length); return b; }
    std::string result() const {
        char b[256]; snprintf(b, sizeof b, "%ld,%.4f,%.4f,%.4f,%.4f,%.4f,%.4f,%.4f,%.4f", n, avg(mrr), avg(ndcg), avg(hit), avg(pop), avg(prec), coverage(), avg(rec), f1());
        return b; }
};

// ---- readers (src/io.rs) ----------------------------------------------------------------------------------------
struct Row { uint32_t session; uint64_t item; long long time; };
std::vector<Row> read_training_data(const std::string& path) {   // io.rs:13-30
    std::vector<Row> rows; std::ifstream f(path); std::string line;
    if (!f) { fprintf(stderr, "cannot open %s\n", path.c_str()); exit(2); }
    std::getline(f, line);   // header
    while (std::getline(f, line)) { std::istringstream ss(line); unsigned long long s, i; double t; if (ss >> s >> i >> t) rows.push_back(Row{(uint32_t)s, i, (long long)std::llround(t)}); }
    return rows;
}
std::map<uint32_t, Items> read_test_data_evolving(const std::string& path) {   // io.rs:40-59 (events ordered by time)
    std::map<uint32_t, std::vector<std::pair<long long, uint64_t>>> g;
    for (const Row& r : read_training_data(path)) g[r.session].push_back({r.time, r.item});
    std::map<uint32_t, Items> out;
    for (auto& kv : g) { std::stable_sort(kv.second.begin(), kv.second.end(), [](auto& a, auto& b) { return a.first < b.first; });
                         Items& v = out[kv.first]; for (auto& e : kv.second) v.push_back(e.second); }
    return out;
}

// ---- config (src/config.rs): the handful of keys the evaluator reads, defaults as config.rs:155-185 ---------------
struct Config { std::map<std::string, std::string> kv;
    std::string get(const std::string& k, const std::string& d) const { auto it = kv.find(k); return it == kv.end() ? d : it->second; }
    long geti(const std::string& k, long d) const { auto it = kv.find(k); return it == kv.end() ? d : atol(it->second.c_str()); }
    bool getb(const std::string& k, bool d) const { auto it = kv.find(k); return it == kv.end() ? d : it->second == "true"; } };
Config read_toml(const std::string& path) {
    Config c; std::ifstream f(path); std::string line, section;
    if (!f) { fprintf(stderr, "cannot open config %s\n", path.c_str()); exit(2); }
    auto trim = [](std::string s) { size_t a = s.find_first_not_of(" \t\r"), b = s.find_last_not_of(" \t\r"); return a == std::string::npos ? std::string() : s.substr(a, b - a + 1); };
    while (std::getline(f, line)) {
        const size_t hash = line.find('#'); if (hash != std::string::npos) line = line.substr(0, hash);
        line = trim(line); if (line.empty()) continue;
        if (line[0] == '[') { section = trim(line.substr(1, line.find(']') - 1)); continue; }
        const size_t eq = line.find('='); if (eq == std::string::npos) continue;
        std::string k = trim(line.substr(0, eq)), v = trim(line.substr(eq + 1));
        if (v.size() >= 2 && v.front() == '"' && v.back() == '"') v = v.substr(1, v.size() - 2);   // config_processors.rs Unquote
        c.kv[section + "." + k] = v;
    }
    if (const char* e = getenv("TRAINING_DATA")) c.kv["data.training_data_path"] = e;   // config.rs:84-88
    return c;
}

void die(const char* what) { fprintf(stderr, "%s: %s\n", what, srn_last_error()); exit(1); }

int metrics_selftest() {
    int bad = 0; auto check = [&](const char* name, double got, double want) { const bool ok = std::fabs(got - want) < 1e-15; printf("%-28s %.17g %s\n", name, got, ok ? "ok" : "MISMATCH"); bad += !ok; };
    Items recs24; for (uint64_t i = 1; i <= 24; ++i) recs24.push_back(i);
    { Reporter r({}, 20); r.add(recs24, {3, 55, 88, 4}); check("ndcg.rs:76-84", r.avg(r.ndcg), 0.36121211352040195); }
    { Reporter r({}, 20); r.add(recs24, {3, 55, 3, 4}); check("mrr.rs:53-61", r.avg(r.mrr), 0.3333333333333333);
      check("precision.rs:64-74", r.avg(r.prec), 2.0 / 20); check("recall.rs:65-75", r.avg(r.rec), 0.5); }
    { Reporter r({}, 20); r.add({1, 2}, {2, 3}); check("f1score.rs:51-58", r.f1(), 0.09090909090909091); check("hitrate.rs:55-62", r.avg(r.hit), 1.0); }
    { Reporter r({}, 20); check("f1score.rs:61-64 (empty)", r.f1(), 0.0); check("hitrate.rs:65-68 (empty)", r.avg(r.hit), 0.0); }
    return bad;
}

Items parse_list(const std::string& s) { Items v; std::stringstream ss(s); std::string tok; while (std::getline(ss, tok, ',')) if (!tok.empty()) v.push_back(strtoull(tok.c_str(), nullptr, 10)); return v; }

}  // namespace

int main(int argc, char** argv) {
    if (argc >= 2 && !strcmp(argv[1], "--metrics-selftest")) return metrics_selftest();
    if (argc >= 4 && !strcmp(argv[1], "--evaluate-file")) {   // evaluate_file.rs
        std::vector<uint64_t> train_items; for (const Row& r : read_training_data(argv[2])) train_items.push_back(r.item);
        Reporter rep(train_items, 20);
        std::ifstream f(argv[3]); std::string line;
        while (std::getline(f, line)) { const size_t semi = line.find(';'); if (semi == std::string::npos) continue;
            Items next = parse_list(line.substr(semi + 1)); if (next.empty()) continue; rep.add(parse_list(line.substr(0, semi)), next); }
        printf("===============================================================\n===              EVALUATING PREDICTONS BY FILE             ====\n===============================================================\n");
        printf("training data: %s\npredictions file: %s\n%s\n%s\n", argv[2], argv[3], rep.name().c_str(), rep.result().c_str());
        return 0;
    }
    if (argc < 2) { fprintf(stderr, "usage: evaluator <config.toml> [--per-call] | --evaluate-file <train> <predictions> | --metrics-selftest\n"); return 2; }
    const bool per_call = argc >= 3 && !strcmp(argv[2], "--per-call");
    const Config cfg = read_toml(argv[1]);
    const size_t m = cfg.geti("model.m_most_recent_sessions", 500), k = cfg.geti("model.neighborhood_size_k", 500);
    const size_t how_many = cfg.geti("model.num_items_to_recommend", 21), max_items = cfg.geti("model.max_items_in_session", 2);
    const double idf_weighting = (double)cfg.geti("model.idf_weighting", 1);
    const bool business = cfg.getb("logic.enable_business_logic", false);
    const std::string train = cfg.get("data.training_data_path", ""), test = cfg.get("hyperparam.test_data_path", "");

    srn_index_t* index = nullptr;
    if (srn_index_new_from_csv(train.c_str(), m, idf_weighting, 0, 0, &index)) die("index");   // VMISIndex::new_from_csv (evaluator.rs:25-29)
    printf("test_data_file:%s\n", test.c_str());
    const auto sessions = read_test_data_evolving(test);
    std::vector<uint64_t> train_items; for (const Row& r : read_training_data(train)) train_items.push_back(r.item);
    Reporter rep(train_items, how_many);                                                          // evaluator.rs:42

    // evaluator.rs:46-56: prefixes 1..len-1, last max_items_in_session items
    std::vector<uint64_t> flat; std::vector<uint32_t> qoff{0}; std::vector<std::pair<const Items*, size_t>> truth;
    for (const auto& kv : sessions) { const Items& ev = kv.second;
        for (size_t state = 1; state < ev.size(); ++state) { const size_t start = state > max_items ? state - max_items : 0;
            flat.insert(flat.end(), ev.begin() + start, ev.begin() + state); qoff.push_back((uint32_t)flat.size()); truth.push_back({&ev, state}); } }
    const size_t nq = truth.size();
    std::vector<uint64_t> ids(nq * how_many); std::vector<double> scores(nq * how_many); std::vector<uint32_t> counts(nq);
    std::vector<double> lat_us;
    const auto t0 = std::chrono::steady_clock::now();
    if (!per_call) { if (srn_predict_batch(index, flat.data(), qoff.data(), nq, k, m, how_many, business ? SRN_FLAG_BUSINESS_LOGIC : 0, ids.data(), scores.data(), counts.data())) die("predict_batch"); }
    else for (size_t q = 0; q < nq; ++q) { size_t n = 0; const auto c0 = std::chrono::steady_clock::now();
            if (srn_predict(index, flat.data() + qoff[q], qoff[q + 1] - qoff[q], k, m, how_many, business, ids.data() + q * how_many, scores.data() + q * how_many, &n)) die("predict");
            lat_us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - c0).count()); counts[q] = (uint32_t)n; }
    const double total_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (size_t q = 0; q < nq; ++q) { Items recs(ids.begin() + q * how_many, ids.begin() + q * how_many + counts[q]);
        rep.add(recs, Items(truth[q].first->begin() + truth[q].second, truth[q].first->end())); }   // evaluator.rs:73-74
    printf("===============================================================\n===               START EVALUATING TEST FILE               ====\n===============================================================\n");
    printf("%s\n%s\nQty test evaluations: %zu\n", rep.name().c_str(), rep.result().c_str(), nq);
    if (per_call) { std::sort(lat_us.begin(), lat_us.end()); printf("Prediction latency\n");
        for (double p : {25.0, 50.0, 75.0, 90.0, 95.0, 99.5}) printf("p%g (microseconds): %.0f\n", p, lat_us[std::min(lat_us.size() - 1, (size_t)(p / 100.0 * (double)lat_us.size()))]); }
    else printf("Batch prediction: %zu queries in %.3f ms (%.0f queries/s incl. host<->device copies)\n", nq, total_s * 1e3, (double)nq / total_s);
    srn_index_free(index);
    return 0;
}

// Closed-loop load generator for the dynamic batcher (srn_batcher_*): T client threads, each sending its next evolving
// session as soon as the previous answer is back -- the shape of actix workers calling predict
// (src/endpoints/recommend_resource.rs:56-62).  Reports requests/s and the latency percentiles the reference's README quotes
// (p90, p99.5).  usage: serve_bench <index.srn> <queries.bin> <threads> <seconds> <k> <m> <how_many> <max_batch> <max_wait_us> [direct]
//   queries.bin: u64 nq, u32 off[nq+1], u64 items[off[nq]];  "direct" = every thread calls srn_predict itself (no batching); "direct resident:N" = with N resident
//   workgroups of the persistent latency path behind srn_predict (srn_index_serve_start: no kernel launch per call)
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include <sys/resource.h>

#include "../../../include/serenade_hip.h"

int main(int argc, char** argv) {
    if (argc < 10) { fprintf(stderr, "usage: %s index queries threads seconds k m how_many max_batch max_wait_us [direct]\n", argv[0]); return 2; }
    const int T = atoi(argv[3]); const double secs = atof(argv[4]);
    const size_t k = atol(argv[5]), m = atol(argv[6]), n = atol(argv[7]), max_batch = atol(argv[8]); const unsigned wait_us = (unsigned)atol(argv[9]);
    const bool direct = argc > 10 && !strcmp(argv[10], "direct");
    const unsigned resident = argc > 11 && !strncmp(argv[11], "resident:", 9) ? (unsigned)atoi(argv[11] + 9) : 0u;
    srn_index_t* idx = nullptr;
    if (srn_index_load(argv[1], 0, &idx)) { fprintf(stderr, "load: %s\n", srn_last_error()); return 1; }
    FILE* f = fopen(argv[2], "rb"); if (!f) { perror("queries"); return 1; }
    uint64_t nq = 0; if (fread(&nq, 8, 1, f) != 1) return 1;
    std::vector<uint32_t> off(nq + 1); if (fread(off.data(), 4, nq + 1, f) != nq + 1) return 1;
    std::vector<uint64_t> items(off[nq]); if (fread(items.data(), 8, items.size(), f) != items.size()) return 1;
    fclose(f);
    srn_batcher_t* b = nullptr;
    if (!direct && srn_batcher_create(idx, max_batch, wait_us, k, m, n, 0, &b)) { fprintf(stderr, "batcher: %s\n", srn_last_error()); return 1; }
    { std::vector<uint64_t> ids(n); std::vector<double> sc(n); size_t cnt;   // warm-up: first launch, workspace allocation
      for (int i = 0; i < 20; ++i) srn_predict(idx, &items[off[i]], off[i + 1] - off[i], k, m, n, 0, ids.data(), sc.data(), &cnt); }
    if (resident && srn_index_serve_start(idx, k, m, n, 0, resident, 4, 5000)) { fprintf(stderr, "serve_start: %s\n", srn_last_error()); return 1; }
    auto cpu_s = [] { rusage ru; getrusage(RUSAGE_SELF, &ru); return ru.ru_utime.tv_sec + ru.ru_stime.tv_sec + 1e-6 * (ru.ru_utime.tv_usec + ru.ru_stime.tv_usec); };
    auto throttled = [] { unsigned long long n = 0, us = 0; if (FILE* c = fopen("/sys/fs/cgroup/cpu.stat", "r")) { char key[64]; unsigned long long v; while (fscanf(c, "%63s %llu", key, &v) == 2) { if (!strcmp(key, "nr_throttled")) n = v; if (!strcmp(key, "throttled_usec")) us = v; } fclose(c); } return std::make_pair(n, us); };
    const double cpu0 = cpu_s(); const auto thr0 = throttled();
    std::atomic<bool> stop{false}; std::atomic<uint64_t> errors{0};
    std::vector<std::vector<float>> lat(T);
    std::vector<std::thread> th;
    const auto t0 = std::chrono::steady_clock::now();
    for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
        std::vector<uint64_t> ids(n); std::vector<double> sc(n); size_t cnt = 0;
        uint64_t q = (uint64_t)t * 7919 % nq;
        lat[t].reserve(1 << 16);
        while (!stop.load(std::memory_order_relaxed)) {
            const auto a = std::chrono::steady_clock::now();
            const int rc = direct ? srn_predict(idx, &items[off[q]], off[q + 1] - off[q], k, m, n, 0, ids.data(), sc.data(), &cnt)
                                  : srn_batcher_predict(b, &items[off[q]], off[q + 1] - off[q], ids.data(), sc.data(), &cnt);
            const auto z = std::chrono::steady_clock::now();
            if (rc) ++errors;
            lat[t].push_back(std::chrono::duration<float, std::micro>(z - a).count());
            q = (q + T) % nq;
        } });
    std::this_thread::sleep_for(std::chrono::duration<double>(secs));
    stop = true;
    for (auto& x : th) x.join();
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::vector<float> all; for (auto& v : lat) all.insert(all.end(), v.begin(), v.end());
    std::sort(all.begin(), all.end());
    auto pct = [&](double p) { return all.empty() ? 0.f : all[std::min(all.size() - 1, (size_t)(p * all.size()))]; };
    uint64_t nr = 0, nb = 0, mx = 0; if (b) srn_batcher_stats(b, &nr, &nb, &mx); else srn_predict_stats(idx, &nb, &nr, &mx);   // (direct: the rounds concurrent srn_predict calls combined into)
    uint64_t sv = 0, nsv = 0, lau = 0; uint32_t ln = 0; srn_index_serve_stats(idx, &sv, &nsv, &lau, &ln);
    printf("{\"resident_workgroups\": %u, \"answered_without_a_launch\": %llu, \"sent_to_the_launch_path\": %llu, ", ln, (unsigned long long)sv, (unsigned long long)nsv);
    printf("\"mode\": \"%s\", \"threads\": %d, \"seconds\": %.2f, \"requests\": %zu, \"requests_per_s\": %.1f, \"errors\": %llu, "
           "\"latency_us\": {\"p50\": %.1f, \"p90\": %.1f, \"p99\": %.1f, \"p99_5\": %.1f, \"max\": %.1f}, \"batches\": %llu, \"mean_batch\": %.1f, \"max_batch_seen\": %llu, "
           "\"max_batch\": %zu, \"max_wait_us\": %u, \"host_cpu_cores_used\": %.2f, \"cgroup_throttled_periods\": %llu, \"cgroup_throttled_ms\": %.1f}\n",
           direct ? "direct srn_predict per thread" : "srn_batcher", T, el, all.size(), all.size() / el, (unsigned long long)errors.load(),
           pct(0.5), pct(0.9), pct(0.99), pct(0.995), all.empty() ? 0.f : all.back(), (unsigned long long)nb, nb ? (double)nr / nb : 0.0, (unsigned long long)mx, max_batch, wait_us,
           (cpu_s() - cpu0) / el, throttled().first - thr0.first, (throttled().second - thr0.second) / 1e3);
    if (b) srn_batcher_free(b);
    srn_index_free(idx);
    return errors ? 1 : 0;
}

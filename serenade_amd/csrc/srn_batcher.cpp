// =====================================================================================
// Dynamic batching in front of srn_predict_batch: the serving-side caller of the hot path.
//
// The reference serves every /v1/recommend call with its own vmisknn::predict on an actix worker thread
// (src/endpoints/recommend_resource.rs:56-62, workers sharing one Arc<VMISIndex>: src/bin/serving.rs:62-94).  A GPU wants
// the concurrent calls of all workers in ONE launch: srn_batcher_predict() has predict()'s shape (one evolving session in,
// how_many (id, score) pairs out, blocking), any number of threads may call it, and a dispatcher thread collects what is
// waiting -- up to max_batch requests, or whatever has arrived max_wait_us after the first one -- into one
// srn_predict_batch.  No HTTP, no session store: those stay the host application's.
// =====================================================================================
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "srn_internal.h"

using namespace srn;

namespace {
template <typename F> int guarded(F f) {   // never let an exception cross the C boundary
    try { return f(); }
    catch (const std::bad_alloc&) { return fail(SRN_ENOMEM, "out of host memory"); }
    catch (const std::exception& e) { return fail(SRN_EINVAL, std::string("internal error: ") + e.what()); }
    catch (...) { return fail(SRN_EINVAL, "internal error"); }
}
struct Request {
    const uint64_t* evolving; size_t len;
    uint64_t* out_ids; double* out_scores; size_t* out_n;
    int rc = SRN_OK; bool done = false;
    std::string err;
    std::mutex mu; std::condition_variable cv;   // completion is signalled per request: a finished batch wakes its own callers only
};
}  // namespace

struct srn_batcher {
    const srn_index_t* idx = nullptr;
    size_t max_batch = 0, k = 0, m = 0, how_many = 0; unsigned max_wait_us = 0, flags = 0;
    std::mutex mu;
    std::condition_variable cv_work;
    std::deque<Request*> queue;
    bool stop = false;
    std::thread worker;
    uint64_t n_requests = 0, n_batches = 0, max_seen = 0;

    void run() {
        std::vector<Request*> batch;
        std::vector<uint64_t> items, ids; std::vector<uint32_t> off, counts; std::vector<double> scores;
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv_work.wait(lk, [&] { return stop || !queue.empty(); });
            if (stop && queue.empty()) return;
            if (queue.size() < max_batch && max_wait_us) {   // give the other callers a moment to join this launch
                const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(max_wait_us);
                cv_work.wait_until(lk, deadline, [&] { return stop || queue.size() >= max_batch; });
            }
            batch.clear();
            while (!queue.empty() && batch.size() < max_batch) { batch.push_back(queue.front()); queue.pop_front(); }
            lk.unlock();
            const size_t nq = batch.size();
            int rc = SRN_OK; std::string err;
            try {   // an allocation failure here must fail this batch's requests, not terminate the host process
                off.assign(nq + 1, 0); items.clear();
                for (size_t i = 0; i < nq; ++i) { items.insert(items.end(), batch[i]->evolving, batch[i]->evolving + batch[i]->len); off[i + 1] = (uint32_t)items.size(); }
                ids.assign(nq * how_many, 0); scores.assign(nq * how_many, 0.0); counts.assign(nq, 0);
                rc = srn_predict_batch(idx, items.data(), off.data(), nq, k, m, how_many, flags, ids.data(), scores.data(), counts.data());
                if (rc) err = srn_last_error();
            } catch (const std::bad_alloc&) { rc = SRN_ENOMEM; err = "out of host memory in the batch dispatcher"; }
            catch (const std::exception& e) { rc = SRN_EINVAL; err = std::string("internal error in the batch dispatcher: ") + e.what(); }
            for (size_t i = 0; i < nq; ++i) {
                Request* r = batch[i];
                if (rc == SRN_OK) { const size_t n = counts[i]; std::memcpy(r->out_ids, &ids[i * how_many], n * 8); std::memcpy(r->out_scores, &scores[i * how_many], n * 8); *r->out_n = n; }
                else { *r->out_n = 0; r->err = err; }
                r->rc = rc;
            }
            for (Request* r : batch) { std::lock_guard<std::mutex> g(r->mu); r->done = true; r->cv.notify_one(); }   // (notify under the lock: r lives on its caller's stack and is gone once the caller runs)
            lk.lock();
            n_requests += nq; ++n_batches; max_seen = std::max<uint64_t>(max_seen, nq);
        }
    }
};

extern "C" {

int srn_batcher_create(const srn_index_t* idx, size_t max_batch, unsigned max_wait_us, size_t k, size_t m, size_t how_many,
                       int enable_business_logic, srn_batcher_t** out) {
    return guarded([&]() -> int {
        if (!out) return fail(SRN_EINVAL, "null argument");
        *out = nullptr;
        if (!idx) return fail(SRN_EINVAL, "null index");
        if (!idx->dev) return fail(SRN_ENODEV, "index has no device attached; there is no CPU fallback behind this ABI");
        { int rc0 = check_has_rows(idx->flat, "srn_batcher_create"); if (rc0) return rc0; }
        if (k == 0 || m == 0 || how_many == 0) return fail(SRN_EINVAL, "k, m and how_many must be > 0");
        if (how_many > SRN_MAX_HOW_MANY || k > SRN_MAX_K || m > 0x7FFFFFFFull) return fail(SRN_ERANGE, "k, m or how_many above the limits (srn_limits)");
        // (q_off is 32-bit: max_batch sessions of SRN_MAX_SESSION_LEN items must not wrap it)
        if (max_batch == 0 || max_batch * (unsigned long long)SRN_MAX_SESSION_LEN >= 0xFFFFFFFFull) return fail(SRN_EINVAL, "max_batch must be in [1, 2^32 / SRN_MAX_SESSION_LEN)");
        srn_batcher* b = new srn_batcher();
        b->idx = idx; b->max_batch = max_batch; b->max_wait_us = max_wait_us; b->k = k; b->m = m; b->how_many = how_many;
        b->flags = enable_business_logic ? SRN_FLAG_BUSINESS_LOGIC : 0;
        b->worker = std::thread([b] { b->run(); });
        *out = b; return SRN_OK; });
}

int srn_batcher_predict(srn_batcher_t* b, const uint64_t* evolving, size_t len, uint64_t* out_ids, double* out_scores, size_t* out_n) {
    return guarded([&]() -> int {
        if (!b || !out_ids || !out_scores || !out_n) return fail(SRN_EINVAL, "null argument");
        *out_n = 0;
        // one bad request must not fail the launch it shares with others: validate here, with srn_predict's messages
        if (!evolving || len == 0) return fail(SRN_EINVAL, "empty evolving session (the reference panics: src/vmisknn/mod.rs:157)");
        if (len > SRN_MAX_SESSION_LEN) return fail(SRN_ERANGE, "evolving session longer than SRN_MAX_SESSION_LEN");
        Request r; r.evolving = evolving; r.len = len; r.out_ids = out_ids; r.out_scores = out_scores; r.out_n = out_n;
        { std::lock_guard<std::mutex> lk(b->mu);
          if (b->stop) return fail(SRN_EINVAL, "batcher is shutting down");
          b->queue.push_back(&r); }
        b->cv_work.notify_one();
        { std::unique_lock<std::mutex> lk(r.mu); r.cv.wait(lk, [&] { return r.done; }); }
        if (r.rc) return fail(r.rc, r.err);
        return SRN_OK; });
}

int srn_batcher_stats(srn_batcher_t* b, uint64_t* n_requests, uint64_t* n_batches, uint64_t* max_batch_seen) {
    if (!b) return fail(SRN_EINVAL, "null argument");
    std::lock_guard<std::mutex> lk(b->mu);
    if (n_requests) *n_requests = b->n_requests;
    if (n_batches) *n_batches = b->n_batches;
    if (max_batch_seen) *max_batch_seen = b->max_seen;
    return SRN_OK;
}

int srn_batcher_how_many(const srn_batcher_t* b, size_t* out) {
    if (!b || !out) return fail(SRN_EINVAL, "null argument");
    *out = b->how_many;
    return SRN_OK;
}

void srn_batcher_free(srn_batcher_t* b) {
    if (!b) return;
    { std::lock_guard<std::mutex> lk(b->mu); b->stop = true; }
    b->cv_work.notify_all();
    if (b->worker.joinable()) b->worker.join();   // (requests still queued are served first)
    delete b;
}

}  // extern "C"

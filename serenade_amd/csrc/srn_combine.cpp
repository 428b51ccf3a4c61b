// =====================================================================================
// srn_predict from MANY host threads on one index: combining rounds.
//
// The reference's serving binary calls vmisknn::predict once per HTTP request on `num_workers` actix threads that share one Arc<VMISIndex>
// (src/bin/serving.rs:62-94, src/endpoints/recommend_resource.rs:56).  On a GPU a call is two launches and a stream synchronise: 16 threads doing that
// side by side reach 91 K requests/s, 64 threads fall to 26 K with a 75 ms p99 (64 streams, 64 spinning waits on 16 cores: round 2's
// profiles/r02_serving_cfg3.json).  So concurrent srn_predict calls on one handle COMBINE: a caller that finds an open round joins it and sleeps on
// the round's futex; a caller that finds none opens one, waits for one of a few lanes (SRN_PREDICT_LANES, default 4: that many rounds run side by side),
// closes the round and runs it as ONE batch on the zero-copy latency path; every member copies its own rows out.  A lone caller opens, closes and
// runs its own round at once -- the latency path of before plus one uncontended mutex.  No dispatcher thread, no timer: a round is as large as
// the load that arrived while the lanes were busy.  (srn_batcher_* remains the explicit queue for hosts that want max_batch / max_wait control.)
// =====================================================================================
#include <linux/futex.h>
#include <sys/syscall.h>
#include <sched.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <climits>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "srn_internal.h"

namespace srn {

namespace {
struct Round {
    size_t k, m, how_many; unsigned flags;
    std::vector<const uint64_t*> ev; std::vector<uint32_t> len;          // members' evolving sessions (borrowed: the members are blocked in their calls)
    std::vector<uint64_t> ids; std::vector<double> scores; std::vector<uint32_t> counts;
    int rc = SRN_OK; std::string err;
    std::vector<int> member_rc; std::vector<std::string> member_err;    // filled only where a failed round was re-run member by member
    std::atomic<uint32_t> done{0};
    bool closed = false;
};
void futex_wait(std::atomic<uint32_t>* w) {
    while (w->load(std::memory_order_acquire) == 0) syscall(SYS_futex, reinterpret_cast<uint32_t*>(w), FUTEX_WAIT_PRIVATE, 0u, nullptr, nullptr, 0);
}
void futex_wake_all(std::atomic<uint32_t>* w) {
    w->store(1, std::memory_order_release);
    syscall(SYS_futex, reinterpret_cast<uint32_t*>(w), FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0);
}
}  // namespace

// The combiner's state is touched for ~100 ns per request (join a round: two push_backs), by up to hundreds of threads that a finished round releases at the
// same instant.  A sleeping mutex turns that into a convoy -- every hand-over a futex wake and a context switch: 40 us of CPU per request measured, 14 cores at
// 350 K requests/s, and with 256 callers the cgroup's CPU quota throttled the whole process (75 ms stalls) -- so: a spin lock (test-and-test-and-set, pause,
// yield after a while), and the leaders that wait for a lane sleep on a futex of their own.
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    __asm__ __volatile__("yield");
#else
    std::atomic_signal_fence(std::memory_order_seq_cst);
#endif
}
struct SpinLock {
    std::atomic<uint32_t> v{0};
    void lock() {
        for (uint32_t spins = 0;;) {
            if (v.load(std::memory_order_relaxed) == 0 && v.exchange(1, std::memory_order_acquire) == 0) return;
            if (++spins < 2000) cpu_relax(); else { sched_yield(); spins = 0; }
        }
    }
    void unlock() { v.store(0, std::memory_order_release); }
};
struct Combiner {
    SpinLock mu;
    std::atomic<uint32_t> lane_seq{0};          // bumped whenever a lane comes free: what a waiting leader sleeps on
    std::vector<std::shared_ptr<Round>> open;   // joinable rounds (one per parameter set in use: a serving process has one)
    int lanes_busy = 0, leaders_waiting = 0;
    std::atomic<uint64_t> n_rounds{0}, n_requests{0}, max_round{0};
};
Combiner* combiner_create() { Combiner* c = new Combiner(); c->open.reserve(64); return c; }   // (no allocation under the spin lock in the common case)
struct SpinGuard { SpinLock& l; bool held = true; explicit SpinGuard(SpinLock& x) : l(x) { l.lock(); } void unlock() { if (held) { l.unlock(); held = false; } } void lock() { if (!held) { l.lock(); held = true; } }
                   ~SpinGuard() { if (held) l.unlock(); } };   // released on unwind too: a bad_alloc under the lock must not leave every later caller spinning
void combiner_free(Combiner* c) { delete c; }
void combiner_stats(const Combiner* c, uint64_t* rounds, uint64_t* requests, uint64_t* max_round) {
    if (rounds) *rounds = c->n_rounds.load(); if (requests) *requests = c->n_requests.load(); if (max_round) *max_round = c->max_round.load();
}

// One evolving session through the index's combiner.  Arguments are validated by the caller (srn_predict).
int combiner_predict(Combiner* c, const srn_index* idx, const uint64_t* evolving, size_t len, size_t k, size_t m, size_t how_many, unsigned flags,
                     uint64_t* out_ids, double* out_scores, size_t* out_n, int lanes, size_t round_cap) {
    std::shared_ptr<Round> r; size_t me = 0; bool leader = false;
    {
        std::shared_ptr<Round> fresh = std::make_shared<Round>();   // (allocated outside the lock; dropped if an open round takes this call)
        fresh->k = k; fresh->m = m; fresh->how_many = how_many; fresh->flags = flags;
        fresh->ev.reserve(round_cap); fresh->len.reserve(round_cap);   // (a round never grows beyond round_cap: the members' push_backs below do not allocate)
        SpinGuard g(c->mu);
        for (auto& o : c->open)
            if (!o->closed && o->k == k && o->m == m && o->how_many == how_many && o->flags == flags && o->ev.size() < round_cap) { r = o; break; }
        if (r) { me = r->ev.size(); r->ev.push_back(evolving); r->len.push_back((uint32_t)len); g.unlock(); }
        else {
            r = std::move(fresh);
            r->ev.push_back(evolving); r->len.push_back((uint32_t)len);
            c->open.push_back(r); leader = true;
            while (c->lanes_busy >= lanes) {   // (the round stays open meanwhile: the load that arrives now rides along)
                const uint32_t seq = c->lane_seq.load(std::memory_order_relaxed);
                ++c->leaders_waiting;
                g.unlock();
                syscall(SYS_futex, reinterpret_cast<uint32_t*>(&c->lane_seq), FUTEX_WAIT_PRIVATE, seq, nullptr, nullptr, 0);
                g.lock();
                --c->leaders_waiting;
            }
            ++c->lanes_busy; r->closed = true;
            c->open.erase(std::find(c->open.begin(), c->open.end(), r));
            g.unlock();
        }
    }
    if (leader) {
        const size_t nq = r->ev.size();
        try {
            std::vector<uint64_t> flat; std::vector<uint32_t> off(nq + 1, 0); uint32_t max_len = 0;
            for (size_t i = 0; i < nq; ++i) { flat.insert(flat.end(), r->ev[i], r->ev[i] + r->len[i]); off[i + 1] = (uint32_t)flat.size(); max_len = std::max(max_len, r->len[i]); }
            r->ids.assign(nq * how_many, 0); r->scores.assign(nq * how_many, 0.0); r->counts.assign(nq, 0);
            LaunchParams p{};
            p.nq = (uint32_t)nq; p.k = (uint32_t)k; p.m = (uint32_t)m; p.how_many = (uint32_t)how_many; p.flags = flags; p.max_len = max_len;
            // a lone caller spins on its result (lowest latency); a shared round sleeps on an interrupt: its members' cores are better spent on their next requests
            r->rc = device_predict(idx->dev, idx->flat, p, false, nullptr, flat.data(), off.data(), r->ids.data(), r->scores.data(), r->counts.data(), nullptr, nullptr, nullptr, nullptr,
                                   nullptr, false, /*blocking_wait=*/nq > 1);
            if (r->rc) r->err = last_error_string();
            // The reference's predict() calls are independent (src/endpoints/recommend_resource.rs:56): one member's session must not fail the others'.  A round
            // refused for its shape (a long session pushing the batch geometry over a limit) is re-run member by member, each with its own verdict.
            if ((r->rc == SRN_EINVAL || r->rc == SRN_ERANGE) && nq > 1) {
                r->member_rc.assign(nq, SRN_OK); r->member_err.assign(nq, std::string());
                for (size_t i = 0; i < nq; ++i) {
                    const uint32_t off1[2] = {0u, r->len[i]};
                    LaunchParams p1 = p; p1.nq = 1; p1.max_len = r->len[i];
                    r->member_rc[i] = device_predict(idx->dev, idx->flat, p1, false, nullptr, r->ev[i], off1, &r->ids[i * how_many], &r->scores[i * how_many], &r->counts[i], nullptr, nullptr,
                                                     nullptr, nullptr, nullptr, false, /*blocking_wait=*/true);
                    if (r->member_rc[i]) r->member_err[i] = last_error_string();
                }
                r->rc = SRN_OK; r->err.clear();
            }
        } catch (const std::bad_alloc&) { r->rc = SRN_ENOMEM; r->err = "out of host memory in a combining round"; }
        catch (const std::exception& e) { r->rc = SRN_EINVAL; r->err = std::string("internal error in a combining round: ") + e.what(); }
        c->n_rounds.fetch_add(1, std::memory_order_relaxed); c->n_requests.fetch_add(nq, std::memory_order_relaxed);
        uint64_t mx = c->max_round.load(std::memory_order_relaxed); while (nq > mx && !c->max_round.compare_exchange_weak(mx, nq)) {}
        futex_wake_all(&r->done);
        c->mu.lock();
        --c->lanes_busy; c->lane_seq.fetch_add(1, std::memory_order_relaxed);
        const bool wake = c->leaders_waiting > 0;
        c->mu.unlock();
        if (wake) syscall(SYS_futex, reinterpret_cast<uint32_t*>(&c->lane_seq), FUTEX_WAKE_PRIVATE, 1, nullptr, nullptr, 0);
    } else futex_wait(&r->done);
    *out_n = 0;
    if (r->rc) return fail(r->rc, r->err);
    if (!r->member_rc.empty() && r->member_rc[me]) return fail(r->member_rc[me], r->member_err[me]);
    const uint32_t cnt = r->counts[me];
    if (cnt == 0xFFFFFFFFu) return fail(SRN_ERANGE, "a query exceeded the kernel's table limits");   // (this member's query only: the others' rows are complete)
    const size_t n = std::min<size_t>(cnt, how_many);
    std::memcpy(out_ids, &r->ids[me * how_many], n * 8); std::memcpy(out_scores, &r->scores[me * how_many], n * 8);
    *out_n = n;
    return SRN_OK;
}

}  // namespace srn

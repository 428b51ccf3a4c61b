// =====================================================================================
// srn_predict_batch on HOST pointers, chunked and pipelined (the entry point a reference-side binding calls: INTEGRATION.md section 2,
// callers src/bin/evaluator.rs:46-76, src/endpoints/recommend_resource.rs:56).
//
// A batch is cut into chunks; chunk c's queries go user buffer -> pinned staging -> HBM on one stream, its kernels run on one of two kernel
// streams (so that a chunk's tail overlaps the next chunk's head), its results go HBM -> pinned staging on a third stream and from there into
// the caller's (pageable) buffers on a small pool of copy threads -- while the GPU is busy with the next chunks.  What is left outside the
// kernels' shadow is the first chunk's upload and the last chunk's download + copy.  Round 2 copied the whole batch in, ran, and copied the whole
// result out through the runtime's pageable path: 64.5 ms against 26.7 ms resident for 2^20 queries.
// =====================================================================================
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "srn_runtime.h"

namespace srn {

namespace {
// ---- copy threads: pinned staging -> the caller's buffers (page faults of a fresh result buffer included) ----
struct CopyTask { void* dst; const void* src; size_t n; std::atomic<int>* pending; };
class CopyPool {
public:
    static CopyPool& get() { static CopyPool* p = new CopyPool(); return *p; }   // (leaked on purpose: no static destructor racing with threads at exit)
    void submit(void* dst, const void* src, size_t n, std::atomic<int>* pending, int slices = 0) {
        if (n == 0) return;
        const size_t nthr = slices > 0 ? (size_t)slices : threads_.size();
        if (threads_.empty() || n < (1u << 20)) { memcpy(dst, src, n); return; }   // small: the wake-up costs more than the copy
        const size_t per = ((n + nthr - 1) / nthr + 4095) / 4096 * 4096;
        std::vector<CopyTask> ts;
        for (size_t o = 0; o < n; o += per) ts.push_back(CopyTask{(char*)dst + o, (const char*)src + o, std::min(per, n - o), pending});
        pending->fetch_add((int)ts.size(), std::memory_order_relaxed);
        { std::lock_guard<std::mutex> lk(mu_); for (auto& t : ts) q_.push_back(t); }
        cv_.notify_all();
    }
    static void wait(std::atomic<int>* pending) {   // (tasks are ~100 us: spin politely)
        while (pending->load(std::memory_order_acquire) != 0) std::this_thread::yield();
    }
private:
    CopyPool() {
        unsigned n = std::thread::hardware_concurrency() / 4;
        if (const char* e = getenv("SRN_COPY_THREADS")) n = (unsigned)std::max(0, atoi(e));
        else n = std::min(8u, std::max(2u, n));
        for (unsigned i = 0; i < n; ++i) threads_.emplace_back([this] { run(); });
        for (auto& t : threads_) t.detach();
    }
    void run() {
        for (;;) {
            CopyTask t;
            { std::unique_lock<std::mutex> lk(mu_); cv_.wait(lk, [&] { return !q_.empty(); }); t = q_.front(); q_.pop_front(); }
            memcpy(t.dst, t.src, t.n);
            t.pending->fetch_sub(1, std::memory_order_release);
        }
    }
    std::mutex mu_; std::condition_variable cv_; std::deque<CopyTask> q_; std::vector<std::thread> threads_;
};
}  // namespace

// Streams: the runtime maps HIP streams onto a few hardware queues, and its device-to-host copies into pinned memory are blit KERNELS (rocprofv3 timeline,
// profiles/r03_hostpipe_timeline.txt: __amd_rocclr_copyBuffer, ~0.45 ms per 64 K queries' results).  A download enqueued on a stream of its own landed in the
// same hardware queue as one of the two kernel streams, BEHIND the next chunk's kernels: the pipeline ran [kernels c, c+1][downloads c, c+1][kernels c+2, c+3] and
// no download ever overlapped compute.  So a chunk is one in-order sequence on ONE stream -- upload, kernels, download -- and the overlap comes from the two kernel
// streams: chunk c's download runs beside chunk c + 1's kernels.  Device staging is per stream (stream order protects it), pinned result staging has four slots
// (the copy threads lag the GPU).
struct HostPipe {
    static constexpr int NOUT = 4;
    hipStream_t s_k[2] = {nullptr, nullptr};
    hipEvent_t e_in[2] = {}, e_out[NOUT] = {};
    char* pin_in[2] = {}; size_t pin_in_bytes[2] = {};
    char* pin_out[NOUT] = {}; size_t pin_out_bytes[NOUT] = {};
    char* dev_in[2] = {}; size_t dev_in_bytes[2] = {};
    char* dev_out[2] = {}; size_t dev_out_bytes[2] = {};
    std::atomic<int> pending[NOUT];
    HostPipe() { for (auto& p : pending) p.store(0); }
};

static void pipe_free(HostPipe* hp) {
    if (!hp) return;
    for (auto& p : hp->pending) CopyPool::wait(&p);
    for (int i = 0; i < 2; ++i) { if (hp->pin_in[i]) hipHostFree(hp->pin_in[i]); if (hp->dev_in[i]) hipFree(hp->dev_in[i]); if (hp->dev_out[i]) hipFree(hp->dev_out[i]);
                                  if (hp->e_in[i]) hipEventDestroy(hp->e_in[i]); if (hp->s_k[i]) hipStreamDestroy(hp->s_k[i]); }
    for (int i = 0; i < HostPipe::NOUT; ++i) { if (hp->pin_out[i]) hipHostFree(hp->pin_out[i]); if (hp->e_out[i]) hipEventDestroy(hp->e_out[i]); }
    delete hp;
}
void hostpipes_free(DeviceState* d) { for (HostPipe* hp : d->all_pipes) pipe_free(hp); d->all_pipes.clear(); d->free_pipes.clear(); }

static HostPipe* pipe_acquire(DeviceState* d) {
    { std::lock_guard<std::mutex> lk(d->mu); if (!d->free_pipes.empty()) { HostPipe* hp = d->free_pipes.back(); d->free_pipes.pop_back(); return hp; } }
    HostPipe* hp = new HostPipe();
    bool ok = true;
    for (int i = 0; i < 2 && ok; ++i) ok = hipStreamCreateWithFlags(&hp->s_k[i], hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&hp->e_in[i], hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < HostPipe::NOUT && ok; ++i) ok = hipEventCreateWithFlags(&hp->e_out[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) { pipe_free(hp); return nullptr; }
    std::lock_guard<std::mutex> lk(d->mu); d->all_pipes.push_back(hp);
    return hp;
}
static void pipe_release(DeviceState* d, HostPipe* hp) { std::lock_guard<std::mutex> lk(d->mu); d->free_pipes.push_back(hp); }

static int ensure_pinned(char** p, size_t* have, size_t need) {
    if (*have >= need) return SRN_OK;
    if (*p) HIP_TRY(hipHostFree(*p));
    *p = nullptr; *have = 0;
    need = need + need / 8 + 4096;
    HIP_TRY(hipHostMalloc((void**)p, need, hipHostMallocDefault));
    *have = need; return SRN_OK;
}

// how a host batch is cut: one chunk up to 8192 queries (a second chunk's launches cost more than its overlap buys), then chunks of ~16 K queries, at most
// 64 K (and no chunk's results above ~32 MB of pinned staging): measured on config 3, profiles/r03_host_pipe_probe.txt
uint32_t hostpipe_chunks(uint32_t nq, uint32_t how_many) {
    const Knobs kn = knobs();
    if (kn.host_chunks > 0) return (uint32_t)std::min<uint64_t>(nq, (uint64_t)kn.host_chunks);
    if (nq <= 8192) return 1;
    const uint64_t chunk_max = std::min<uint64_t>(65536, std::max<uint64_t>(1024, (32ull << 20) / ((uint64_t)how_many * 16 + 4)));
    const uint64_t by_max = (nq + chunk_max - 1) / chunk_max, by_16k = (nq + 16383) / 16384;
    return (uint32_t)std::max<uint64_t>(2, std::min<uint64_t>(by_16k, std::max<uint64_t>(4, by_max)));
}

int device_predict_host_pipelined(DeviceState* d, const FlatIndex& ix, const LaunchParams& p_in, const uint64_t* h_items, const uint32_t* h_qoff,
                                  uint64_t* h_ids, double* h_scores, uint32_t* h_counts) {
    HIP_TRY(hipSetDevice(d->device));
    HostPipe* hp = pipe_acquire(d);
    if (!hp) return fail(SRN_EHIP, "cannot create the streams / events of the host-pointer pipeline");
    bool good = false;
    struct Rel { DeviceState* d; HostPipe* hp; bool* good; ~Rel() {
        for (auto& pnd : hp->pending) CopyPool::wait(&pnd);
        if (*good) pipe_release(d, hp);
        else { hipStreamSynchronize(hp->s_k[0]); hipStreamSynchronize(hp->s_k[1]); pipe_release(d, hp); } } } rel{d, hp, &good};
    const uint32_t nq = p_in.nq, n = p_in.how_many;
    const Knobs kn = knobs();
    const auto t_start = std::chrono::steady_clock::now();
    auto now_us = [&] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_start).count(); };
    double tr_in = 0, tr_enq = 0, tr_wait_out = 0, tr_wait_copy = 0, tr_submit = 0;
    const uint32_t nchunks = hostpipe_chunks(nq, n);
    const uint32_t csz = (uint32_t)(((uint64_t)nq + nchunks - 1) / nchunks);
    CopyPool& pool = CopyPool::get();
    auto out_bytes = [&](uint32_t cq) { return (size_t)cq * n * 16 + (size_t)cq * 4; };
    auto flush = [&](uint32_t c) -> int {   // chunk c's results: pinned staging -> the caller's buffers (asynchronous: the copy threads)
        const int o = (int)(c % HostPipe::NOUT);
        const uint32_t q0 = c * csz, cq = std::min(csz, nq - q0);
        double t0 = now_us();
        HIP_TRY(hipEventSynchronize(hp->e_out[o]));
        tr_wait_out += now_us() - t0; t0 = now_us();
        const char* src = hp->pin_out[o];
        if (!kn.host_nocopy) {
            pool.submit(h_ids + (size_t)q0 * n, src, (size_t)cq * n * 8, &hp->pending[o], kn.copy_slices);
            pool.submit(h_scores + (size_t)q0 * n, src + (size_t)cq * n * 8, (size_t)cq * n * 8, &hp->pending[o], kn.copy_slices);
        }
        memcpy(h_counts + q0, src + (size_t)cq * n * 16, (size_t)cq * 4);
        tr_submit += now_us() - t0;
        return SRN_OK;
    };
    constexpr uint32_t LAG = 2;      // the host collects chunk c - LAG after it has enqueued chunk c: two chunks stay queued on the GPU while it waits
    for (uint32_t c = 0; c < nchunks; ++c) {
        const int i = (int)(c & 1u), o = (int)(c % HostPipe::NOUT);
        const uint32_t q0 = c * csz, cq = std::min(csz, nq - q0);
        if (q0 >= nq) break;
        hipStream_t sk = hp->s_k[i];   // the chunk's ONE stream: upload, kernels, download in order
        const size_t it0 = h_qoff[q0], it1 = h_qoff[q0 + cq], in_items = (it1 - it0) * 8, in_off = ((size_t)cq + 1) * 4, in_bytes = (in_items + 255) / 256 * 256 + in_off;
        // input staging of chunk c - 2 has been read by its upload
        double t0 = now_us();
        if (c >= 2) HIP_TRY(hipEventSynchronize(hp->e_in[i]));
        { int rc = ensure_pinned(&hp->pin_in[i], &hp->pin_in_bytes[i], in_bytes); if (rc) return rc; }
        memcpy(hp->pin_in[i], h_items + it0, in_items);
        memcpy(hp->pin_in[i] + (in_items + 255) / 256 * 256, h_qoff + q0, in_off);
        tr_in += now_us() - t0; t0 = now_us();
        const size_t ob = out_bytes(cq);
        if (hp->dev_in_bytes[i] < in_bytes || hp->dev_out_bytes[i] < ob) {   // (growing frees: the stream that uses the buffers must be idle)
            HIP_TRY(hipStreamSynchronize(sk));
            int rc = ensure(&hp->dev_in[i], &hp->dev_in_bytes[i], in_bytes); if (rc) return rc;
            rc = ensure(&hp->dev_out[i], &hp->dev_out_bytes[i], ob); if (rc) return rc;
        }
        HIP_TRY(hipMemcpyAsync(hp->dev_in[i], hp->pin_in[i], in_bytes, hipMemcpyHostToDevice, sk));
        HIP_TRY(hipEventRecord(hp->e_in[i], sk));
        LaunchParams p = p_in;
        p.nq = cq;
        p.items_flat = (const uint64_t*)hp->dev_in[i] - it0;   // (the chunk's offsets stay global: the base is shifted instead)
        p.q_off = (const uint32_t*)(hp->dev_in[i] + (in_items + 255) / 256 * 256);
        p.out_ids = (uint64_t*)hp->dev_out[i]; p.out_scores = (double*)(hp->dev_out[i] + (size_t)cq * n * 8); p.out_counts = (uint32_t*)(hp->dev_out[i] + (size_t)cq * n * 16);
        HIP_TRY(hipMemsetAsync(hp->dev_out[i], 0, (size_t)cq * n * 16, sk));   // the unused tail of each row reads as 0
        { int rc = device_predict(d, ix, p, true, sk, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr); if (rc) return rc; }
        // download: the pinned slot's previous contents (chunk c - NOUT) must have reached the caller's buffers
        tr_enq += now_us() - t0; t0 = now_us();
        CopyPool::wait(&hp->pending[o]);
        tr_wait_copy += now_us() - t0; t0 = now_us();
        { int rc = ensure_pinned(&hp->pin_out[o], &hp->pin_out_bytes[o], ob); if (rc) return rc; }
        HIP_TRY(hipMemcpyAsync(hp->pin_out[o], hp->dev_out[i], ob, hipMemcpyDeviceToHost, sk));
        HIP_TRY(hipEventRecord(hp->e_out[o], sk));
        tr_enq += now_us() - t0;
        if (c >= LAG) { int rc = flush(c - LAG); if (rc) return rc; }
    }
    { const uint32_t done = (nq + csz - 1) / csz; for (uint32_t c = done > LAG ? done - LAG : 0; c < done; ++c) { int rc = flush(c); if (rc) return rc; } }
    { const double t0 = now_us(); for (auto& pnd : hp->pending) CopyPool::wait(&pnd); tr_wait_copy += now_us() - t0; }
    if (kn.host_trace) fprintf(stderr, "[srn] host pipe: nq %u, %u chunks of %u: total %.0f us = input staging %.0f + enqueue %.0f + wait downloads %.0f + wait copy threads %.0f + submit %.0f\n",
                               nq, nchunks, csz, now_us(), tr_in, tr_enq, tr_wait_out, tr_wait_copy, tr_submit);
    good = true;
    return SRN_OK;
}

}  // namespace srn

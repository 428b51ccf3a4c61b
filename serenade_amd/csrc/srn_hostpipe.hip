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
#include "srn_hipsync.h"

namespace srn {

namespace {
// ---- copy threads: pinned staging -> the caller's buffers (page faults of a fresh result buffer included) ----
struct CopyTask { void* dst; const void* src; size_t n; std::atomic<int>* pending; };
class CopyPool {
public:
    static CopyPool& get() { static CopyPool* p = new CopyPool(); return *p; }   // (leaked on purpose: no static destructor racing with threads at exit)
    void submit(void* dst, const void* src, size_t n, std::atomic<int>* pending, int slices = 0, size_t inline_below = 1u << 20) {
        if (n == 0) return;
        const size_t nthr = slices > 0 ? (size_t)slices : threads_.size();
        if (threads_.empty() || n < inline_below) { memcpy(dst, src, n); return; }   // small: the wake-up costs more than the copy
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

// What the rocprofv3 timelines of three designs showed (profiles/r03_hostpipe_timeline.txt), and why the pipeline looks the way it does:
//   * the runtime maps HIP streams onto a few hardware queues, and a download on a stream of its own may land in the SAME hardware queue as a kernel stream,
//     behind the next chunk's kernels: with four streams (upload, two kernel streams, download) no download ever overlapped compute;
//   * two kernel streams with equal chunks lock in phase -- a stream alone on the GPU runs faster and catches up with the other -- so both compute for ~3 ms and
//     then both download for ~0.75 ms while no kernel runs; a staggered start re-synchronises within two chunks;
//   * a batch cut into equal small chunks pays the launch sequence's fixed cost and its tail 16 times: 16 x 1.89 ms against 26.7 ms for one 2^20-query launch.
// So: ONE kernel stream runs the chunks back to back, ONE other stream carries the downloads (two streams in all: two hardware queues), and the chunks SHRINK
// geometrically -- half the batch, a quarter, an eighth ... -- so that the big launches run at the resident rate, every download and every copy into the caller's
// buffers hides behind the kernels of the chunks that follow, and what is left exposed at the end is the smallest chunk's download and copy.
// An own download kernel with a FEW workgroups (experiment, SRN_D2H_BLOCKS = n): the runtime's device-to-host copy on the second stream is a blit kernel whose
// grid fills the chip, and the next chunk's small launches wait for its waves to drain.  Measured, not kept as the default: 16..256 workgroups of plain 16-byte
// stores into pinned memory reach 34..35 ms per 2^20 queries against 29.3 ms with hipMemcpyAsync (profiles/r03_host_pipe_probe.txt) -- the blit is the faster mover.
__global__ __launch_bounds__(256) void d2h_copy_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

struct HostPipe {
    static constexpr int NOUT = 3;
    hipStream_t s_k = nullptr, s_out = nullptr;
    static constexpr int NPIECE = 8;
    hipEvent_t e_in[2] = {}, e_k[2] = {}, e_out[NOUT] = {}, e_piece[NPIECE] = {};   // e_piece: a single-chunk batch downloads in pieces
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
                                  if (hp->e_in[i]) hipEventDestroy(hp->e_in[i]); if (hp->e_k[i]) hipEventDestroy(hp->e_k[i]); }
    for (int i = 0; i < HostPipe::NOUT; ++i) { if (hp->pin_out[i]) hipHostFree(hp->pin_out[i]); if (hp->e_out[i]) hipEventDestroy(hp->e_out[i]); }
    for (auto& e : hp->e_piece) if (e) hipEventDestroy(e);
    if (hp->s_k) hipStreamDestroy(hp->s_k); if (hp->s_out) hipStreamDestroy(hp->s_out);
    delete hp;
}
void hostpipes_free(DeviceState* d) { for (HostPipe* hp : d->all_pipes) pipe_free(hp); d->all_pipes.clear(); d->free_pipes.clear(); }

static HostPipe* pipe_acquire(DeviceState* d) {
    { std::lock_guard<std::mutex> lk(d->mu); if (!d->free_pipes.empty()) { HostPipe* hp = d->free_pipes.back(); d->free_pipes.pop_back(); return hp; } }
    HostPipe* hp = new HostPipe();
    // The download stream gets the HIGHEST priority: priority classes have hardware queues of their own, so it cannot end up in the kernel stream's queue (where the
    // download would wait behind the next chunk's kernels: seen in a process that had created a handful of other streams before -- 34.5 ms per 2^20 queries instead of
    // 28.3), and its blit kernel is dispatched ahead of the fast kernel's next workgroups.
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    bool ok = hipStreamCreateWithFlags(&hp->s_k, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithPriority(&hp->s_out, hipStreamNonBlocking, prio_hi) == hipSuccess;
    for (int i = 0; i < 2 && ok; ++i) ok = hipEventCreateWithFlags(&hp->e_in[i], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&hp->e_k[i], hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < HostPipe::NOUT && ok; ++i) ok = hipEventCreateWithFlags(&hp->e_out[i], hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < HostPipe::NPIECE && ok; ++i) ok = hipEventCreateWithFlags(&hp->e_piece[i], hipEventDisableTiming) == hipSuccess;
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
    HIP_TRY(hipHostMalloc((void**)p, need, hipHostMallocMapped));
    *have = need; return SRN_OK;
}

// How a host batch is cut (boundaries, in queries).  Up to 65 536 queries: one chunk (a second launch sequence costs more than its overlap buys: 2.69 ms in one
// chunk against 2.81 in two, profiles/r03_host_pipe_probe.txt).  Above: half the
// batch, then half of the rest (three quarters once fewer than 256 K queries are left) ... down to 8..16 K queries; no chunk's results above ~192 MB of pinned staging (large how_many).  SRN_HOST_CHUNKS = n forces n
// equal chunks (experiments).
std::vector<uint32_t> hostpipe_cuts(uint32_t nq, uint32_t how_many) {
    const Knobs kn = knobs();
    std::vector<uint32_t> starts{0};
    if (kn.host_chunks > 0) {
        const uint32_t nc = (uint32_t)std::min<uint64_t>(nq, (uint64_t)kn.host_chunks), csz = (nq + nc - 1) / nc;
        for (uint64_t q = csz; q < nq; q += csz) starts.push_back((uint32_t)q);
    } else if (kn.host_first_pct > 0 && nq >= 8192 && nq <= 65536) {
        starts.push_back((uint32_t)std::max<uint64_t>(1024, (uint64_t)nq * kn.host_first_pct / 100 / 256 * 256));
    } else if (nq > 65536) {
        const uint64_t chunk_max = std::max<uint64_t>(4096, (192ull << 20) / ((uint64_t)how_many * 16 + 4));
        uint64_t at = 0;
        while (nq - at > 16384) {   // (below 256 K queries left: three quarters at a time -- few launches matter more than a short tail there)
            const uint64_t left = nq - at;
            const uint64_t take = std::min<uint64_t>(chunk_max, std::max<uint64_t>(8192, left >= 262144 ? left / 2 : left * 3 / 4));
            at += take; starts.push_back((uint32_t)at);
        }
    }
    starts.push_back(nq);
    return starts;
}
uint32_t hostpipe_chunks(uint32_t nq, uint32_t how_many) { return (uint32_t)hostpipe_cuts(nq, how_many).size() - 1; }

int device_predict_host_pipelined(DeviceState* d, const FlatIndex& ix, const LaunchParams& p_in, const uint64_t* h_items, const uint32_t* h_qoff,
                                  uint64_t* h_ids, double* h_scores, uint32_t* h_counts) {
    HIP_TRY(hipSetDevice(d->device));
    HostPipe* hp = pipe_acquire(d);
    if (!hp) return fail(SRN_EHIP, "cannot create the streams / events of the host-pointer pipeline");
    bool good = false;
    struct Rel { DeviceState* d; HostPipe* hp; bool* good; ~Rel() {
        for (auto& pnd : hp->pending) CopyPool::wait(&pnd);
        if (!*good) { hipStreamSynchronize(hp->s_k); hipStreamSynchronize(hp->s_out); }
        pipe_release(d, hp); } } rel{d, hp, &good};
    const uint32_t nq = p_in.nq, n = p_in.how_many;
    const Knobs kn = knobs();
    const auto t_start = std::chrono::steady_clock::now();
    auto now_us = [&] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_start).count(); };
    double tr_in = 0, tr_enq = 0, tr_wait_out = 0, tr_wait_copy = 0, tr_submit = 0;
    const std::vector<uint32_t> starts = hostpipe_cuts(nq, n);
    const uint32_t nchunks = (uint32_t)starts.size() - 1;
    const bool one = nchunks == 1;   // a single chunk: everything in order on the kernel stream (no cross-stream hop: each costs ~10 us of a ~0.35 ms call)
    CopyPool& pool = CopyPool::get();
    auto out_bytes = [&](uint32_t cq) { return (size_t)cq * n * 16 + (size_t)cq * 4; };
    auto flush = [&](uint32_t c) -> int {   // chunk c's results: pinned staging -> the caller's buffers (asynchronous: the copy threads)
        const int o = (int)(c % HostPipe::NOUT);
        const uint32_t q0 = starts[c], cq = starts[c + 1] - q0;
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
    // A single-chunk batch (<= 65 536 queries) has nothing to hide its download behind -- but the download can hide the copy into the caller's buffers: it goes in
    // pieces (at most 8), and while piece j + 1 crosses PCIe, piece j moves from the pinned staging to the caller's (pageable) ids / scores / counts.
    // 65 536 queries x 21: 22 MB, 0.45 ms of download + 0.33 ms of copy in a row before.
    const size_t ob_all = out_bytes(nq);   // (below 4 MB: two pieces, copied by this thread; above: pieces of >= 2 MB, copied by the pool -- smaller ones download faster than a copy thread wakes up)
    const uint32_t npieces = !one || kn.host_nocopy || ob_all < (1u << 20) ? 1u : ob_all < (4u << 20) ? 2u : (uint32_t)std::min<size_t>(HostPipe::NPIECE, std::max<size_t>(2, ob_all >> 21));
    auto piece_at = [&](size_t ob, uint32_t j) -> size_t { return j >= npieces ? ob : (ob / npieces * j) / 4096 * 4096; };
    auto flush_pieces = [&]() -> int {
        const size_t ob = out_bytes(nq), A = (size_t)nq * n * 8, B = A * 2;   // staging layout: ids | scores | counts
        const char* src = hp->pin_out[0];
        char* const dst_of[3] = {(char*)h_ids, (char*)h_scores, (char*)h_counts};
        const size_t reg_lo[3] = {0, A, B}, reg_hi[3] = {A, B, ob};
        for (uint32_t j = 0; j < npieces; ++j) {
            const size_t lo = piece_at(ob, j), hi = piece_at(ob, j + 1);
            double t0 = now_us();
            HIP_TRY(hipEventSynchronize(hp->e_piece[j]));
            tr_wait_out += now_us() - t0; t0 = now_us();
            for (int r = 0; r < 3; ++r) {
                const size_t a = std::max(lo, reg_lo[r]), b = std::min(hi, reg_hi[r]);
                if (a < b) pool.submit(dst_of[r] + (a - reg_lo[r]), src + a, b - a, &hp->pending[0], kn.copy_slices, /*inline_below=*/ob < (4u << 20) ? ~(size_t)0 : j + 1 == npieces ? (512u << 10) : (128u << 10));   // (a small batch: this thread copies piece j itself while piece j + 1 downloads)
            }
            tr_submit += now_us() - t0;
        }
        return SRN_OK;
    };
    hipStream_t sk = hp->s_k, sout = one ? hp->s_k : hp->s_out;
    for (uint32_t c = 0; c < nchunks; ++c) {
        const int i = (int)(c & 1u), o = (int)(c % HostPipe::NOUT);
        const uint32_t q0 = starts[c], cq = starts[c + 1] - q0;
        const size_t it0 = h_qoff[q0], it1 = h_qoff[q0 + cq], in_items = (it1 - it0) * 8, in_off = ((size_t)cq + 1) * 4, in_bytes = (in_items + 255) / 256 * 256 + in_off;
        const size_t ob = out_bytes(cq);
        double t0 = now_us();
        // staging this chunk overwrites: the pinned input slot of chunk c - 2 (read by its upload), the device slots of chunk c - 2 (its kernels and its download)
        if (c >= 2) HIP_TRY(hipEventSynchronize(hp->e_in[i]));
        { int rc = ensure_pinned(&hp->pin_in[i], &hp->pin_in_bytes[i], in_bytes); if (rc) return rc; }
        memcpy(hp->pin_in[i], h_items + it0, in_items);
        memcpy(hp->pin_in[i] + (in_items + 255) / 256 * 256, h_qoff + q0, in_off);
        tr_in += now_us() - t0; t0 = now_us();
        if (hp->dev_in_bytes[i] < in_bytes || hp->dev_out_bytes[i] < ob) {   // (growing frees: whatever uses the buffers must be idle)
            HIP_TRY(hipStreamSynchronize(sk)); HIP_TRY(hipStreamSynchronize(sout));
            int rc = ensure(&hp->dev_in[i], &hp->dev_in_bytes[i], in_bytes); if (rc) return rc;
            rc = ensure(&hp->dev_out[i], &hp->dev_out_bytes[i], ob); if (rc) return rc;
        }
        HIP_TRY(hipMemcpyAsync(hp->dev_in[i], hp->pin_in[i], in_bytes, hipMemcpyHostToDevice, sk));   // (stream order: chunk c - 2's kernels are done with the slot)
        HIP_TRY(hipEventRecord(hp->e_in[i], sk));
        if (!one && c >= 2) HIP_TRY(hipStreamWaitEvent(sk, hp->e_out[(c - 2) % HostPipe::NOUT], 0));   // device result slot: chunk c - 2's download is done
        LaunchParams p = p_in;
        p.nq = cq;
        p.items_flat = (const uint64_t*)hp->dev_in[i] - it0;   // (the chunk's offsets stay global: the base is shifted instead)
        p.q_off = (const uint32_t*)(hp->dev_in[i] + (in_items + 255) / 256 * 256);
        p.out_ids = (uint64_t*)hp->dev_out[i]; p.out_scores = (double*)(hp->dev_out[i] + (size_t)cq * n * 8); p.out_counts = (uint32_t*)(hp->dev_out[i] + (size_t)cq * n * 16);
        HIP_TRY(hipMemsetAsync(hp->dev_out[i], 0, (size_t)cq * n * 16, sk));   // the unused tail of each row reads as 0
        { int rc = device_predict(d, ix, p, true, sk, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr); if (rc) return rc; }
        if (!one) HIP_TRY(hipEventRecord(hp->e_k[i], sk));
        // download: the pinned slot's previous contents (chunk c - NOUT) must have reached the caller's buffers
        tr_enq += now_us() - t0; t0 = now_us();
        CopyPool::wait(&hp->pending[o]);
        tr_wait_copy += now_us() - t0; t0 = now_us();
        { int rc = ensure_pinned(&hp->pin_out[o], &hp->pin_out_bytes[o], ob); if (rc) return rc; }
        if (!one) HIP_TRY(hipStreamWaitEvent(sout, hp->e_k[i], 0));
        if (one && npieces > 1) {   // (see flush_pieces)
            for (uint32_t j = 0; j < npieces; ++j) {
                const size_t lo = piece_at(ob, j), hi = piece_at(ob, j + 1);
                HIP_TRY(hipMemcpyAsync(hp->pin_out[o] + lo, hp->dev_out[i] + lo, hi - lo, hipMemcpyDeviceToHost, sout));
                HIP_TRY(hipEventRecord(hp->e_piece[j], sout));
            }
        }
        else if (one || kn.d2h_blocks <= 0) HIP_TRY(hipMemcpyAsync(hp->pin_out[o], hp->dev_out[i], ob, hipMemcpyDeviceToHost, sout));
        else {
            char* dst_dev = nullptr; HIP_TRY(hipHostGetDevicePointer((void**)&dst_dev, hp->pin_out[o], 0));
            hipLaunchKernelGGL(d2h_copy_kernel, dim3((unsigned)kn.d2h_blocks), dim3(256), 0, sout, (uint4*)dst_dev, (const uint4*)hp->dev_out[i], (ob + 15) / 16);
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipEventRecord(hp->e_out[o], sout));
        tr_enq += now_us() - t0;
        if (c >= 1) { int rc = flush(c - 1); if (rc) return rc; }   // (chunk c is queued behind chunk c - 1's kernels: the GPU has work while the host waits here)
    }
    { int rc = one && npieces > 1 ? flush_pieces() : flush(nchunks - 1); if (rc) return rc; }
    { const double t0 = now_us(); for (auto& pnd : hp->pending) CopyPool::wait(&pnd); tr_wait_copy += now_us() - t0; }
    if (kn.host_trace) fprintf(stderr, "[srn] host pipe: nq %u, %u chunks (first %u, last %u): total %.0f us = input staging %.0f + enqueue %.0f + wait downloads %.0f + wait copy threads %.0f + submit %.0f\n",
                               nq, nchunks, starts[1], nq - starts[nchunks - 1], now_us(), tr_in, tr_enq, tr_wait_out, tr_wait_copy, tr_submit);
    good = true;
    return SRN_OK;
}

}  // namespace srn

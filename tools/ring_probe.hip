// Round-trip latency of a host <-> persistent-kernel doorbell on this box (round 6, the latency path's design input): the host writes a sequence number, ONE resident
// wave polls it and answers in pinned host memory.  Request word (a) in pinned, device-mapped host memory (the wave's polls cross PCIe), (b) in fine-grained DEVICE memory
// written by the host through the BAR (the wave polls its own HBM), if this platform lets the host touch it.  The kernel leaves by itself after `iters` answers or ~2 s.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/_bin/ring_probe tools/ring_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <csetjmp>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void pong(volatile unsigned* req, volatile unsigned* resp, unsigned iters, unsigned long long budget_ticks) {
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = wall_clock64();
    unsigned last = 0;
    for (unsigned i = 0; i < iters;) {
        const unsigned v = __atomic_load_n((const unsigned*)req, __ATOMIC_RELAXED);   // (system-scope visibility: the buffers are fine-grained / coherent)
        if (v != last) { last = v; __atomic_store_n((unsigned*)resp, v, __ATOMIC_RELAXED); __threadfence_system(); ++i; }
        else if (wall_clock64() - t0 > budget_ticks) break;
    }
    __atomic_store_n((unsigned*)resp + 1, 0xD0DEu, __ATOMIC_RELAXED);
}

static sigjmp_buf jb;
static void on_segv(int) { siglongjmp(jb, 1); }

static void run(const char* what, volatile unsigned* req_host, unsigned* req_dev, volatile unsigned* resp_host, unsigned* resp_dev, unsigned iters) {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    *req_host = 0; resp_host[0] = 0; resp_host[1] = 0;
    hipLaunchKernelGGL(pong, dim3(1), dim3(64), 0, st, (volatile unsigned*)req_dev, (volatile unsigned*)resp_dev, iters, 200000000ull /* 2 s at 100 MHz */);
    std::vector<double> us; us.reserve(iters);
    for (unsigned i = 1; i <= iters; ++i) {
        const auto t0 = std::chrono::steady_clock::now();
        *req_host = i;
        unsigned spins = 0;
        while (resp_host[0] != i) { if (++spins > 400000000u) { fprintf(stderr, "%s: no answer to %u\n", what, i); goto out; } }
        us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
out:
    CK(hipStreamSynchronize(st));
    if (us.size() > 100) { std::sort(us.begin() + 0, us.end()); printf("%-60s round trip p50 %.2f us  p90 %.2f us  p99 %.2f us  (%zu pings)\n", what, us[us.size() / 2], us[us.size() * 9 / 10], us[us.size() * 99 / 100], us.size()); }
    CK(hipStreamDestroy(st));
}

int main() {
    const unsigned iters = 20000;
    unsigned *pin = nullptr, *pin_dev = nullptr;
    CK(hipHostMalloc((void**)&pin, 4096, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostGetDevicePointer((void**)&pin_dev, pin, 0));
    run("request + response in pinned host memory", pin, pin_dev, pin + 64, pin_dev + 64, iters);
    unsigned* fg = nullptr;
    if (hipExtMallocWithFlags((void**)&fg, 4096, hipDeviceMallocFinegrained) != hipSuccess) { printf("no fine-grained device memory\n"); return 0; }
    CK(hipMemset(fg, 0, 4096)); CK(hipDeviceSynchronize());
    signal(SIGSEGV, on_segv); signal(SIGBUS, on_segv);
    if (sigsetjmp(jb, 1) == 0) {
        volatile unsigned* h = (volatile unsigned*)fg;
        const unsigned v = h[0]; h[1] = v + 1;   // does the host reach it at all?
        printf("host can load/store fine-grained device memory (read %u)\n", v);
        run("request in fine-grained DEVICE memory, response in pinned", (volatile unsigned*)fg, fg, pin + 64, pin_dev + 64, iters);
    } else printf("host access to fine-grained device memory faults on this platform\n");
    return 0;
}

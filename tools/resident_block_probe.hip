// Which HIP calls wait for a RESIDENT (never-ending) kernel?  (round 6: the persistent latency path's hazard list.)  One wave spins on a pinned word on a stream of the
// lowest priority; the host then times allocation calls, stream creation and small kernel launches on 1..12 fresh normal-priority streams.  Anything that takes about as long
// as the resident kernel's remaining life (2 s here) waited for it.
// hipcc --offload-arch=gfx950 -O2 -o tools/_bin/resident_block_probe tools/resident_block_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void resident(volatile unsigned* stop, unsigned long long budget) {
    const unsigned long long t0 = wall_clock64();
    while (*stop == 0u && wall_clock64() - t0 < budget) {}
}
__global__ void tiny(unsigned* p) { if (threadIdx.x == 0) p[0] += 1; }
template <typename F> double ms(F f) { const auto a = std::chrono::steady_clock::now(); f(); return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); }
int main() {
    unsigned* pin; (void)hipHostMalloc((void**)&pin, 4096, hipHostMallocMapped | hipHostMallocCoherent); pin[0] = 0;
    unsigned* pin_dev; (void)hipHostGetDevicePointer((void**)&pin_dev, pin, 0);
    int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    printf("stream priority range: lowest %d, highest %d\n", lo, hi);
    for (int prio_case = 0; prio_case < 2; ++prio_case) {
        hipStream_t rs; (void)hipStreamCreateWithPriority(&rs, hipStreamNonBlocking, prio_case == 0 ? lo : 0);
        pin[0] = 0;
        hipLaunchKernelGGL(resident, dim3(1), dim3(64), 0, rs, (volatile unsigned*)pin_dev, 200000000ull /* 2 s */);
        printf("---- resident kernel on a %s-priority stream ----\n", prio_case == 0 ? "LOWEST" : "normal");
        unsigned* d = nullptr; void* h = nullptr;
        printf("hipMalloc 1 MB           %8.2f ms\n", ms([&] { (void)hipMalloc((void**)&d, 1 << 20); }));
        printf("hipHostMalloc 1 MB       %8.2f ms\n", ms([&] { (void)hipHostMalloc(&h, 1 << 20, hipHostMallocMapped); }));
        printf("hipMemset (null stream)  %8.2f ms\n", ms([&] { (void)hipMemset(d, 0, 4096); }));
        printf("hipMemcpy H2D 4 KB       %8.2f ms\n", ms([&] { (void)hipMemcpy(d, pin + 16, 4096, hipMemcpyHostToDevice); }));
        std::vector<hipStream_t> ss;
        for (int i = 0; i < 12; ++i) {
            hipStream_t s; const double c = ms([&] { (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking); });
            const double l = ms([&] { hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d); (void)hipStreamSynchronize(s); });
            printf("fresh stream %2d: create %6.2f ms, tiny kernel + sync %8.2f ms%s\n", i, c, l, l > 500 ? "   <-- waited for the resident kernel" : "");
            ss.push_back(s);
        }
        hipEvent_t e; printf("hipEventCreate           %8.2f ms\n", ms([&] { (void)hipEventCreate(&e); }));
        pin[0] = 1; (void)hipStreamSynchronize(rs);
        for (auto s : ss) (void)hipStreamDestroy(s);
        (void)hipStreamDestroy(rs); (void)hipFree(d); (void)hipHostFree(h);
    }
    return 0;
}

// Microbenchmark: random reads at different granularities on gfx950 -- does a 16-byte load per random 32-byte sector cost what a load per
// random 64-byte line costs?  Decides whether 32-byte row slots would relieve the row gathers.
// hipcc --offload-arch=gfx950 -O3 tools/rand_gran_bench.hip -o /tmp/rgb && /tmp/rgb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int D, int LOADS> __global__ void k(const uint4* __restrict__ a, uint64_t nslots, int stride16, int iters, uint32_t* out) {
    uint64_t x = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        uint4 v[D][LOADS];
#pragma unroll
        for (int d = 0; d < D; ++d) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; const uint64_t s = (x % nslots) * stride16;
#pragma unroll
            for (int l = 0; l < LOADS; ++l) v[d][l] = a[s + l]; }
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int l = 0; l < LOADS; ++l) acc += v[d][l].x + v[d][l].w;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    uint32_t* out; hipMalloc(&out, 4 * 256 * 2048);
    for (uint64_t mb : {800ull, 8000ull}) {
        const uint64_t bytes = mb << 20; uint4* a; hipMalloc(&a, bytes); hipMemset(a, 1, bytes);
        for (int slot : {32, 64, 128}) for (int loads : {1, 2}) {
            if (loads * 16 > slot) continue;
            const uint64_t nslots = bytes / slot; const int stride16 = slot / 16;
            const int threads = 256, blocks = 256 * 24 * 64 / threads, D = 4, iters = 1024;
            auto launch = [&](int it) { if (loads == 1) k<4, 1><<<blocks, threads>>>(a, nslots, stride16, it, out); else k<4, 2><<<blocks, threads>>>(a, nslots, stride16, it, out); };
            launch(16); hipDeviceSynchronize();
            hipEventRecord(e0); launch(iters); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double n = (double)blocks * threads * iters * D;
            printf("region %5llu MB, random %3d-byte slots, %d x 16 B read per slot: %7.2f G slots/s (%.2f TB/s useful)\n", (unsigned long long)mb, slot, loads, n / ms / 1e6, n * loads * 16 / ms / 1e9);
        }
        hipFree(a);
    }
    return 0;
}

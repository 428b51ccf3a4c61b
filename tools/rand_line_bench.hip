// Microbenchmark: random 64-byte-line reads (one 16-byte load per lane, every lane its own line) on gfx950.
// Answers: how many random row slots per second can the chip / one CU fetch, as a function of waves per CU and loads in flight per lane.
// hipcc --offload-arch=gfx950 -O3 tools/rand_line_bench.hip -o /tmp/rlb && /tmp/rlb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
template <int D> __global__ void k(const uint4* __restrict__ a, uint64_t nlines, int iters, uint32_t* out) {
    uint64_t x = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        uint4 v[D];
#pragma unroll
        for (int d = 0; d < D; ++d) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v[d] = a[(x % nlines) * 4]; }
#pragma unroll
        for (int d = 0; d < D; ++d) acc += v[d].x + v[d].w;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
    const uint64_t bytes = 800ull << 20, nlines = bytes / 64;
    uint4* a; uint32_t* out; hipMalloc(&a, bytes); hipMemset(a, 1, bytes); hipMalloc(&out, 4 * 256 * 2048);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves_per_cu : {4, 8, 16, 32}) for (int D : {1, 2, 4, 8}) {
        const int threads = 256, blocks = 256 * waves_per_cu * 64 / threads; const int iters = 4096 / D;
        auto launch = [&](int it) { switch (D) { case 1: k<1><<<blocks, threads>>>(a, nlines, it, out); break; case 2: k<2><<<blocks, threads>>>(a, nlines, it, out); break;
                                                 case 4: k<4><<<blocks, threads>>>(a, nlines, it, out); break; default: k<8><<<blocks, threads>>>(a, nlines, it, out); } };
        launch(16); hipDeviceSynchronize();
        hipEventRecord(e0); launch(iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double lines = (double)blocks * threads * iters * D;
        printf("waves/CU %2d  loads in flight/lane %d : %7.2f G lines/s = %6.2f TB/s of 64-B lines; per CU %.3f lines/cycle @2.4GHz\n", waves_per_cu, D, lines / ms / 1e6, lines * 64 / ms / 1e9,
               lines / ms / 1e6 / 256 / 2.4);
    }
    // few CUs active (one 1024-thread block each): what ONE CU can pull when the memory system is not saturated
    for (int ncu : {8, 32, 128}) for (int D : {1, 2, 4, 8}) {
        const int threads = 1024, blocks = ncu; const int iters = 4096 / D;
        auto launch = [&](int it) { switch (D) { case 1: k<1><<<blocks, threads>>>(a, nlines, it, out); break; case 2: k<2><<<blocks, threads>>>(a, nlines, it, out); break;
                                                 case 4: k<4><<<blocks, threads>>>(a, nlines, it, out); break; default: k<8><<<blocks, threads>>>(a, nlines, it, out); } };
        launch(16); hipDeviceSynchronize();
        hipEventRecord(e0); launch(iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double lines = (double)blocks * threads * iters * D;
        printf("%3d blocks x 16 waves, loads in flight/lane %d : %7.2f G lines/s; per block %.3f lines/cycle @2.4GHz, implied latency at that concurrency %.0f cycles\n", ncu, D, lines / ms / 1e6,
               lines / ms / 1e6 / ncu / 2.4, 1024.0 * D / (lines / ms / 1e6 / ncu / 2.4));
    }
    return 0;
}

// Microbenchmark (round 6): random-slot gather rate by load WIDTH and loads in flight -- 4 / 8 / 16 bytes per lane from random 64-byte slots, 3 or 6 independent loads in
// flight per lane, 24 waves per CU, region in the L2s (16 MB) and far beyond them (4 GB).  Is the 78 G rows/s of tools/row_fetch_bench.hip a property of the request path or of 16-byte loads?
// hipcc --offload-arch=gfx950 -O3 -o tools/_bin/gather_width_bench tools/gather_width_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int W, int D> __global__ __launch_bounds__(512) void k(const uint32_t* __restrict__ a, uint64_t nslots, int iters, uint32_t* out) {
    uint64_t x = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        uint4 v[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            const uint32_t* p = a + (x & (nslots - 1)) * 16;
            if (W == 4) v[d] = make_uint4(*p, 0, 0, 0);
            else if (W == 8) { const uint2 t = *reinterpret_cast<const uint2*>(p); v[d] = make_uint4(t.x, t.y, 0, 0); }
            else v[d] = *reinterpret_cast<const uint4*>(p);
        }
#pragma unroll
        for (int d = 0; d < D; ++d) acc += v[d].x + v[d].y + v[d].w;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int W, int D> void run(const uint32_t* a, uint64_t nslots, uint32_t* out, uint64_t mb) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 1200 / D;
    k<W, D><<<768, 512>>>(a, nslots, 8, out); hipDeviceSynchronize();
    hipEventRecord(e0); k<W, D><<<768, 512>>>(a, nslots, iters, out); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = 768.0 * 512 * iters * D;
    printf("region %5llu MB  %2d bytes per lane, %d in flight: %.1f G slots/s chip-wide (%.2f CU-cycles per slot)\n", (unsigned long long)mb, W, D, n / ms / 1e6, ms * 1e-3 * 2.4e9 * 256 / n);
}
int main() {
    uint32_t* out; hipMalloc(&out, 4 * 512 * 768);
    for (uint64_t mb : {2ull, 64ull, 4096ull}) {
        const uint64_t bytes = mb << 20, nslots = bytes / 64; uint32_t* a; hipMalloc(&a, bytes); hipMemset(a, 1, bytes);
        run<4, 3>(a, nslots, out, mb); run<8, 3>(a, nslots, out, mb); run<16, 3>(a, nslots, out, mb);
        run<4, 6>(a, nslots, out, mb); run<8, 6>(a, nslots, out, mb); run<16, 6>(a, nslots, out, mb);
        hipFree(a);
    }
    return 0;
}

// Microbenchmark: the LATENCY constants of the fast kernel's per-phase model (tools/phase_model.py): shader cycles per DEPENDENT step of
//   (a) an LDS read whose address is the previous read's result (the merge path's binary search and sequential steps),
//   (b) a global load whose address is the previous load's result, over a table far larger than L2 + MALL (a posting list / row slot round trip),
//   (c) a workgroup barrier of 8 waves,
//   (d) a block-wide exclusive scan as the kernel does it (DPP wave scan + one LDS round + barrier),
// with the workgroup alone on its CU and with three workgroups of 512 threads per CU all doing the same (what a phase sees when its neighbours are in the same phase: the
// pessimistic end), and -- for (a) -- beside two workgroups that keep the VALU busy instead (a neighbour in a different phase).
// hipcc --offload-arch=gfx950 -O3 tools/lat_bench.hip -o /tmp/latb && /tmp/latb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>

constexpr int STEPS = 512;
__global__ __launch_bounds__(512) void lds_chain(long long* cyc, uint32_t* sink, int valu_role_from) {
    __shared__ uint32_t lds[12288];   // 48 KB: three workgroups fit a CU
    for (int i = threadIdx.x; i < 12288; i += 512) lds[i] = (uint32_t)((i * 7919u + 13u) % 12288u);
    __syncthreads();
    uint32_t x = threadIdx.x;
    if ((int)(blockIdx.x % 3) >= valu_role_from) {   // a neighbour that only burns VALU issue slots (dependent integer chain, no memory)
        uint32_t y = x | 1u;
        for (int i = 0; i < STEPS * 40; ++i) y = y * 1664525u + 1013904223u;
        sink[blockIdx.x * 512 + threadIdx.x] = y;
        return;
    }
    const long long t0 = clock64();
    for (int i = 0; i < STEPS; ++i) x = lds[x];
    const long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 512 + threadIdx.x] = x;
}
__global__ __launch_bounds__(512) void hbm_chain(const uint32_t* __restrict__ tab, long long* cyc, uint32_t* sink, int steps) {
    __shared__ uint32_t pad[12288];
    pad[threadIdx.x] = 0;
    uint32_t x = (blockIdx.x * 512u + threadIdx.x) * 2654435761u % (1u << 28);
    const long long t0 = clock64();
    for (int i = 0; i < steps; ++i) x = tab[x];
    const long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 512 + threadIdx.x] = x + pad[threadIdx.x];
}
__global__ __launch_bounds__(512) void barrier_chain(long long* cyc, uint32_t* sink) {
    __shared__ uint32_t pad[12288];
    pad[threadIdx.x] = threadIdx.x;
    const long long t0 = clock64();
    for (int i = 0; i < STEPS; ++i) __syncthreads();
    const long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 512 + threadIdx.x] = pad[threadIdx.x];
}
__device__ __forceinline__ uint32_t wave_incl(uint32_t v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(v, d, 64); if ((int)(threadIdx.x & 63) >= d) v += o; }
    return v;
}
__global__ __launch_bounds__(512) void scan_chain(long long* cyc, uint32_t* sink) {
    __shared__ uint32_t pad[12288];
    __shared__ uint32_t wt[8];
    pad[threadIdx.x] = threadIdx.x;
    uint32_t v = threadIdx.x & 3u, acc = 0;
    const long long t0 = clock64();
    for (int i = 0; i < STEPS; ++i) {
        const uint32_t inc = wave_incl(v);
        if ((threadIdx.x & 63) == 63) wt[threadIdx.x >> 6] = inc;
        __syncthreads();
        uint32_t base = 0;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += wt[w];
        acc += base + inc - v; v = (acc >> 3) & 3u;
        __syncthreads();
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 512 + threadIdx.x] = acc + pad[threadIdx.x];
}
int main() {
    const int ncu = 256;
    long long* cyc; uint32_t* sink; uint32_t* tab;
    hipMalloc(&cyc, 8 * 4096); hipMalloc(&sink, 4 * 512 * 4096);
    const size_t N = 1u << 28;   // 1 GiB of 32-bit links: one random cycle (Sattolo), far beyond the 256 MB MALL
    std::vector<uint32_t> h(N); std::iota(h.begin(), h.end(), 0u);
    { std::mt19937_64 rng(7); for (size_t i = N - 1; i > 0; --i) { const size_t j = rng() % i; std::swap(h[i], h[j]); } }
    hipMalloc(&tab, N * 4); hipMemcpy(tab, h.data(), N * 4, hipMemcpyHostToDevice);
    auto avg = [&](int grid) { std::vector<long long> c(grid); hipMemcpy(c.data(), cyc, 8 * grid, hipMemcpyDeviceToHost); double a = 0; int n = 0; for (auto v : c) if (v > 0) { a += v; ++n; } return n ? a / n : 0.0; };
    for (int per_cu : {1, 3}) {
        const int grid = ncu * per_cu;
        hipMemset(cyc, 0, 8 * 4096);
        hipLaunchKernelGGL(lds_chain, dim3(grid), dim3(512), 0, 0, cyc, sink, 3); hipDeviceSynchronize();
        printf("%d workgroup(s) of 512 per CU: dependent LDS read      %7.1f cycles per step\n", per_cu, avg(grid) / STEPS);
        hipMemset(cyc, 0, 8 * 4096);
        hipLaunchKernelGGL(hbm_chain, dim3(grid), dim3(512), 0, 0, tab, cyc, sink, 64); hipDeviceSynchronize();
        printf("%d workgroup(s) of 512 per CU: dependent HBM load      %7.1f cycles per step (every lane its own random line)\n", per_cu, avg(grid) / 64);
        hipMemset(cyc, 0, 8 * 4096);
        hipLaunchKernelGGL(barrier_chain, dim3(grid), dim3(512), 0, 0, cyc, sink); hipDeviceSynchronize();
        printf("%d workgroup(s) of 512 per CU: barrier (8 waves)       %7.1f cycles\n", per_cu, avg(grid) / STEPS);
        hipMemset(cyc, 0, 8 * 4096);
        hipLaunchKernelGGL(scan_chain, dim3(grid), dim3(512), 0, 0, cyc, sink); hipDeviceSynchronize();
        printf("%d workgroup(s) of 512 per CU: block scan (2 barriers) %7.1f cycles\n", per_cu, avg(grid) / STEPS);
    }
    hipMemset(cyc, 0, 8 * 4096);
    hipLaunchKernelGGL(lds_chain, dim3(ncu * 3), dim3(512), 0, 0, cyc, sink, 1); hipDeviceSynchronize();
    printf("1 workgroup chasing LDS beside 2 that burn VALU issue slots: dependent LDS read %7.1f cycles per step\n", avg(ncu * 3) / STEPS);
    // few lanes: one wave per CU, one lane active -- the bare latencies
    hipMemset(cyc, 0, 8 * 4096);
    hipLaunchKernelGGL(hbm_chain, dim3(ncu), dim3(64), 0, 0, tab, cyc, sink, 64); hipDeviceSynchronize();
    printf("1 wave per CU: dependent HBM load %7.1f cycles per step\n", avg(ncu) / 64);
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("shader clock (hipDeviceAttributeClockRate): %d kHz; clock64() ticks per second are measured against it by tools/phase_profile.py's total\n", clk);
    return 0;
}

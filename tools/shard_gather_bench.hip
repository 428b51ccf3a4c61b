// Microbenchmark for the item-sharded back end: what does one rank's row-fragment gather cost at the chip's random-request rate, and what do a
// session-presence bitmap and smaller fragment slots buy?  One rank of a group of G serves nq * K fragment look-ups per batch whatever G is (config 3: 131 072 * 1 360 =
// 178 M); at G = 8 about half of the fragments are empty.
//   (a) random SLOT-byte reads over a region of R MB (the fragment array: config 3 cut in 8 = 199 MB of 16-byte slots), 4 independent look-ups per lane in flight
//   (b) the same behind a presence bitmap of B MB read first (one dependent L2 round trip), a fraction `dens` of the bits set
//   (c) bitmap word + prefix count (8 bytes per 32 sessions) -> index into a COMPACT array of the non-empty fragments
// hipcc --offload-arch=gfx950 -O3 tools/shard_gather_bench.hip -o /tmp/sgb && /tmp/sgb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int SLOT16> struct Vec;   // SLOT16 = slot size / 8
template <> struct Vec<1> { typedef uint2 T; };
template <> struct Vec<2> { typedef uint4 T; };

template <int D, int S8, int MODE>   // MODE 0: plain; 1: bitmap first; 2: bitmap + prefix -> compact array
__global__ __launch_bounds__(256) void k(const typename Vec<S8>::T* __restrict__ a, uint64_t nslots, const uint2* __restrict__ bm, uint64_t nsess, int iters, uint32_t* out) {
    uint64_t x = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        uint64_t s[D]; uint2 w[D];
#pragma unroll
        for (int d = 0; d < D; ++d) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; s[d] = x % (MODE ? nsess : nslots); if (MODE) w[d] = bm[s[d] >> 5]; }
        typename Vec<S8>::T v[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            v[d] = typename Vec<S8>::T{};
            if (MODE == 0) v[d] = a[s[d]];
            else if ((w[d].x >> (s[d] & 31)) & 1u) v[d] = a[MODE == 1 ? s[d] % nslots : (w[d].y + __popc(w[d].x & ((1u << (s[d] & 31)) - 1u))) % nslots];
        }
#pragma unroll
        for (int d = 0; d < D; ++d) acc += v[d].x + v[d].y;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int threads = 256, blocks = 256 * 8, D = 4, iters = 512;
    uint32_t* out; hipMalloc(&out, 4 * (size_t)blocks * threads);
    auto timeit = [&](auto launch) { launch(8); hipDeviceSynchronize(); hipEventRecord(e0); launch(iters); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
                                     return (double)blocks * threads * iters * D / ms / 1e6; };
    printf("(G look-ups/s = 1e9 fragment look-ups per second; one rank of config 3 at 131 072 queries per batch needs 178 M of them: 1 ms at 178 G/s)\n");
    for (uint64_t mb : {1ull, 4ull, 16ull, 25ull, 50ull, 100ull, 200ull, 400ull, 800ull, 7600ull}) {   // (1-4 MB: every request an L2 hit -- the ceiling of divergent requests as such)
        const uint64_t bytes = mb << 20; void* a; if (hipMalloc(&a, bytes) != hipSuccess) { printf("no room for %llu MB\n", (unsigned long long)mb); continue; } hipMemset(a, 1, bytes);
        const double r16 = timeit([&](int it) { k<D, 2, 0><<<blocks, threads>>>((const uint4*)a, bytes / 16, nullptr, 0, it, out); });
        const double r8 = timeit([&](int it) { k<D, 1, 0><<<blocks, threads>>>((const uint2*)a, bytes / 8, nullptr, 0, it, out); });
        printf("(a) region %5llu MB: random 16-byte slots %7.1f G look-ups/s, random 8-byte slots %7.1f\n", (unsigned long long)mb, r16, r8);
        hipFree(a);
    }
    // (b), (c): nsess sessions, a bitmap word + prefix per 32 sessions
    for (uint64_t nsess : {12450000ull, 477000000ull}) for (double dens : {0.5, 0.25}) {
        const uint64_t nw = (nsess + 31) / 32;
        std::vector<uint2> h(nw); uint64_t z = 88172645463325252ull; uint32_t run = 0;
        for (uint64_t i = 0; i < nw; ++i) { uint32_t bits = 0; for (int b = 0; b < 32; ++b) { z ^= z << 13; z ^= z >> 7; z ^= z << 17; if ((double)(z >> 11) / 9007199254740992.0 < dens) bits |= 1u << b; }
                                             h[i] = uint2{bits, run}; run += __builtin_popcount(bits); }
        uint2* bm; hipMalloc(&bm, nw * 8); hipMemcpy(bm, h.data(), nw * 8, hipMemcpyHostToDevice);
        void *full, *comp; hipMalloc(&full, nsess * 16); hipMemset(full, 1, nsess * 16); hipMalloc(&comp, (uint64_t)run * 16 + 64); hipMemset(comp, 1, (uint64_t)run * 16 + 64);
        const double p16 = timeit([&](int it) { k<D, 2, 0><<<blocks, threads>>>((const uint4*)full, nsess, nullptr, 0, it, out); });
        const double b16 = timeit([&](int it) { k<D, 2, 1><<<blocks, threads>>>((const uint4*)full, nsess, bm, nsess, it, out); });
        const double c16 = timeit([&](int it) { k<D, 2, 2><<<blocks, threads>>>((const uint4*)comp, run, bm, nsess, it, out); });
        const double c8 = timeit([&](int it) { k<D, 1, 2><<<blocks, threads>>>((const uint2*)comp, run, bm, nsess, it, out); });
        printf("(b/c) %4.0f M sessions, %2.0f %% non-empty (bitmap + prefix %5.1f MB, 16-byte slots %6.0f MB, compact %6.0f MB): plain %6.1f | bitmap first %6.1f | bitmap + prefix -> compact 16 B %6.1f, 8 B %6.1f G look-ups/s\n",
               nsess / 1e6, dens * 100, nw * 8 / 1048576.0, nsess * 16 / 1048576.0, run * 16.0 / 1048576.0, p16, b16, c16, c8);
        hipFree(bm); hipFree(full); hipFree(comp);
    }
    return 0;
}

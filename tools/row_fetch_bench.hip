// Microbenchmark (round 6): does a random 64-byte row slot cost one memory REQUEST or one request PER LANE that touches it?  vmis_fast_kernel fetches a neighbour's row as
// one 16-byte load per lane (and a second 16-byte load, in another instruction, for the quarter of the rows that have > 6 items): (A).  The alternative (B): two adjacent
// lanes read the two 16-byte halves of a row's first 32 bytes in ONE instruction -- 32 rows per wave-instruction.  If the texture addresser merges adjacent lanes that hit
// the same line, (B) is one request per row for 14 items.  Rows per microsecond, chip-wide, 24 waves per CU, region small enough to live in the L2s / far larger.
// hipcc --offload-arch=gfx950 -O3 -o tools/_bin/row_fetch_bench tools/row_fetch_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE> __global__ __launch_bounds__(512) void k(const uint4* __restrict__ rows, uint64_t nrows, int iters, uint32_t* out) {
    uint64_t x = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        uint4 v[6];
        if (MODE == 0) {   // A: 3 rows per lane, first quads; then second quads for 1 row in 4 (the others re-read row 0's second quad: one line for the lot)
            uint64_t r[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; r[t] = x & (nrows - 1); v[t] = rows[r[t] * 4]; }
#pragma unroll
            for (int t = 0; t < 3; ++t) { const bool more = ((r[t] >> 3) & 3u) == 0u; v[3 + t] = rows[(more ? r[t] : 0) * 4 + 1]; }
        } else {           // B: 6 rows per lane PAIR: lane 2i reads quad 0, lane 2i + 1 quad 1 of the same row
#pragma unroll
            for (int t = 0; t < 6; ++t) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; const uint64_t mine = x & (nrows - 1); const uint64_t r = __shfl(mine, (int)(lane & ~1u)); v[t] = rows[r * 4 + (lane & 1u)]; }
        }
#pragma unroll
        for (int d = 0; d < 6; ++d) acc += v[d].x + v[d].w;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
    uint32_t* out; hipMalloc(&out, 4 * 512 * 768);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (uint64_t mb : {2ull, 64ull, 4096ull}) {
        const uint64_t bytes = mb << 20, nrows = bytes / 64; uint4* a; hipMalloc(&a, bytes); hipMemset(a, 1, bytes);
        for (int rep = 0; rep < 2; ++rep) for (int mode = 0; mode < 2; ++mode) {
            const int iters = 400;
            if (mode == 0) k<0><<<768, 512>>>(a, nrows, 8, out); else k<1><<<768, 512>>>(a, nrows, 8, out);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            if (mode == 0) k<0><<<768, 512>>>(a, nrows, iters, out); else k<1><<<768, 512>>>(a, nrows, iters, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double rows_done = 768.0 * 512 * iters * 3;   // both modes: 3 rows per lane and iteration (A: 3 per lane; B: 6 per pair)
            printf("region %5llu MB  %s : %.3f ms, %.1f G rows/s chip-wide (%.2f CU-cycles per row at 2.4 GHz)\n", (unsigned long long)mb,
                   mode == 0 ? "A  a lane per row, 16 B + 16 B for 1 in 4" : "B  a lane PAIR per row, 32 B at once     ", ms, rows_done / ms / 1e6, ms * 1e-3 * 2.4e9 * 256 / rows_done);
            if (rep == 1 && mode == 0) printf("{\"row_gather_ceiling\": {\"region_mb\": %llu, \"g_rows_per_s\": %.2f}}\n", (unsigned long long)mb, rows_done / ms / 1e6);
        }
        hipFree(a);
    }
    return 0;
}

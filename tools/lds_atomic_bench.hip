// Microbenchmark: LDS op cost on gfx950 (cycles per wave-instruction) for the access patterns of the predict kernel.
// hipcc --offload-arch=gfx950 -O3 tools/lds_atomic_bench.hip -o /tmp/ldsb && /tmp/ldsb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int ITER = 2048;
template <int MODE> __global__ void k(unsigned* out, long long* cyc, unsigned seed, int slots_mask, int dup_every) {
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i <= slots_mask; i += blockDim.x) lds[i] = 0;
    __syncthreads();
    unsigned x = seed + threadIdx.x * 2654435761u + blockIdx.x * 40503u, acc = 0;
    long long t0 = clock64();
    for (int i = 0; i < ITER; ++i) {
        x = x * 1664525u + 1013904223u;
        unsigned a = (x >> 9) & slots_mask;
        if (dup_every && (threadIdx.x % dup_every) != 0) a = (a & ~63u);   // groups of lanes hit the same address
        if (MODE == 0) acc += lds[a];                                      // ds_read_b32 (value consumed)
        if (MODE == 1) atomicAdd(&lds[a], 1u);                             // ds_add_u32 (no return)
        if (MODE == 2) acc += atomicAdd(&lds[a], 1u);                      // ds_add_rtn_u32
        if (MODE == 3) acc += atomicCAS(&lds[a], 0xFFFFFFFFu, x);          // ds_cmpst_rtn_b32
        if (MODE == 4) { unsigned c = __atomic_load_n(&lds[a], __ATOMIC_RELAXED); if (c != x) atomicAdd(&lds[a ^ 1], 1u); }   // read then add (insert hit path)
        if (MODE == 5) lds[a] = x;                                         // ds_write_b32
    }
    long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
    int ncu = 256; unsigned* out; long long* cyc;
    hipMalloc(&out, 4 * 1024 * 2048); hipMalloc(&cyc, 8 * 2048);
    const char* names[] = {"ds_read_b32", "ds_add (no rtn)", "ds_add_rtn", "ds_cmpst_rtn", "read + add", "ds_write_b32"};
    for (int blocks_per_cu = 1; blocks_per_cu <= 2; ++blocks_per_cu)
        for (int threads : {64, 512}) for (int dup : {0, 4}) for (int mode = 0; mode < 6; ++mode) {
            int grid = ncu * blocks_per_cu; size_t lds = 65536;
            auto launch = [&](auto kern) { hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, 0, out, cyc, 123u, 16383, dup); };
            switch (mode) { case 0: launch(k<0>); break; case 1: launch(k<1>); break; case 2: launch(k<2>); break; case 3: launch(k<3>); break; case 4: launch(k<4>); break; default: launch(k<5>); }
            hipDeviceSynchronize();
            std::vector<long long> h(grid); hipMemcpy(h.data(), cyc, 8 * grid, hipMemcpyDeviceToHost);
            double avg = 0; for (auto v : h) avg += v; avg /= grid;
            int waves = threads / 64;
            printf("blocks/CU %d threads %4d dup %d  %-16s : %7.1f cyc per loop-iteration per wave, %6.2f cyc per wave-instr at CU level (%d waves/CU)\n",
                   blocks_per_cu, threads, dup, names[mode], avg / ITER, avg / ITER / (waves * blocks_per_cu), waves * blocks_per_cu);
        }
    return 0;
}

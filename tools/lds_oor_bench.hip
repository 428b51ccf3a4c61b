// Microbenchmark (round 6): what do the PHANTOM adds of walk A cost?  A packed row slot's unused positions hold offsets into a 256-word dump area, so that rows of different
// lengths need no masking -- but only ~30 % of the lanes of a walk's ds_add carry a real row element.  Per wave-instruction at CU level (3 workgroups of 512 threads, 48 KB
// of LDS each, as the fast kernel runs): (A) all 64 lanes scatter over 8 K words; (B) 30 % scatter, 70 % into a 256-word dump; (C) 30 % scatter, 70 % OUT OF RANGE of the
// workgroup's allocation (does the hardware drop them before bank arbitration?); (D) 30 % scatter, the others masked off by exec; (E) like C, with reads (walk B's look).
// hipcc --offload-arch=gfx950 -O3 -o tools/_bin/lds_oor_bench tools/lds_oor_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int ITER = 4096;
template <int MODE> __global__ __launch_bounds__(512) void k(unsigned* out, long long* cyc, unsigned seed) {
    __shared__ unsigned lds[12288];   // 48 KB
    for (int i = threadIdx.x; i < 12288; i += blockDim.x) lds[i] = 0;
    __syncthreads();
    // 16 addresses per lane, fixed before the timed loop: the loop is ds instructions only
    unsigned x = seed + threadIdx.x * 2654435761u + blockIdx.x * 40503u, acc = 0;
    unsigned ad[16]; bool on[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        x = x * 1664525u + 1013904223u;
        const bool real = ((x >> 3) % 10u) < 3u;
        const unsigned a_real = (x >> 9) & 8191u, a_dump = 8192u + ((x >> 9) & 255u), a_oor = 16000u + ((x >> 9) & 255u);   // words; 16 000 words = 64 000 bytes: beyond the 48 KB
        on[j] = MODE == 0 || real;
        ad[j] = 4u * (MODE == 0 ? a_real : real ? a_real : (MODE == 1 || MODE == 5) ? a_dump : a_oor);
    }
    long long t0 = clock64();
    for (int i = 0; i < ITER / 16; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (MODE <= 2) asm volatile("ds_add_u32 %0, %1" :: "v"(ad[j]), "v"(1u) : "memory");
            if (MODE == 3) { if (on[j]) asm volatile("ds_add_u32 %0, %1" :: "v"(ad[j]), "v"(1u) : "memory"); }
            if (MODE >= 4) { unsigned v; asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(ad[j]) : "memory"); acc += v; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    long long t1 = clock64();
    __syncthreads();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    unsigned chk = 0; for (int i = threadIdx.x; i < 8192; i += blockDim.x) chk += lds[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + chk;
}
int main() {
    const int grid = 256 * 3; unsigned* out; long long* cyc;
    hipMalloc(&out, 4 * 512 * grid); hipMalloc(&cyc, 8 * grid);
    const char* names[] = {"A  all lanes scatter", "B  30 % scatter + 70 % dump words", "C  30 % scatter + 70 % out of range", "D  30 % scatter, others masked", "E  reads: 30 % real + 70 % out of range", "F  reads: 30 % real + 70 % dump"};
    for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 6; ++mode) {
        switch (mode) { case 0: hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), 0, 0, out, cyc, 123u); break; case 1: hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), 0, 0, out, cyc, 123u); break;
                        case 2: hipLaunchKernelGGL(k<2>, dim3(grid), dim3(512), 0, 0, out, cyc, 123u); break; case 3: hipLaunchKernelGGL(k<3>, dim3(grid), dim3(512), 0, 0, out, cyc, 123u); break;
                        case 4: hipLaunchKernelGGL(k<4>, dim3(grid), dim3(512), 0, 0, out, cyc, 123u); break; default: hipLaunchKernelGGL(k<5>, dim3(grid), dim3(512), 0, 0, out, cyc, 123u); }
        if (hipDeviceSynchronize() != hipSuccess) { printf("%s: FAULT\n", names[mode]); return 1; }
        std::vector<long long> h(grid); hipMemcpy(h.data(), cyc, 8 * grid, hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += v; avg /= grid;
        std::vector<unsigned> o(512); hipMemcpy(o.data(), out, 4 * 512, hipMemcpyDeviceToHost);
        printf("%-44s %7.1f cycles per iteration per wave, %5.2f per wave-instruction at CU level (24 waves per CU); check %u\n", names[mode], avg / ITER, avg / ITER / 24.0, o[0]);
    }
    return 0;
}

// Calibrates rocprofv3's FETCH_SIZE on gfx950 for the access widths the predict kernel uses (MI355X_MICROARCH.md:
// "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read; other widths are uncalibrated").
// Each kernel touches a known number of bytes of a 4 GiB buffer (>> 256 MiB Infinity Cache) exactly once.
//   hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o tools/fetch_calib
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- tools/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void stream_x4(const uint4* p, size_t n, unsigned* sink) {   // 16 B per lane, coalesced
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void stream_x1(const unsigned* p, size_t n, unsigned* sink) {   // 4 B per lane, coalesced
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void gather_row(const unsigned* p, size_t n_lines, unsigned* sink) {   // 9 lanes read 9 consecutive dwords of a random 128-B line (a "row")
    unsigned acc = 0; const int lane = threadIdx.x & 63, grp = lane / 9, off = lane % 9;
    for (size_t it = blockIdx.x * (size_t)(blockDim.x / 64) + threadIdx.x / 64; it < n_lines / 7; it += (size_t)gridDim.x * (blockDim.x / 64)) {
        size_t g = it * 7 + grp;   // 7 rows per wave-instruction
        size_t line = (g * 0x9E3779B97F4A7C15ull) % n_lines;   // each line visited at most a few times overall; pseudo-random order
        if (grp < 7) acc += p[line * 32 + off];
    }
    if (acc == 0x12345678u) *sink = acc;
}
int main() {
    const size_t bytes = 4ull << 30; unsigned* buf; unsigned* sink;
    hipMalloc(&buf, bytes); hipMalloc(&sink, 4); hipMemset(buf, 1, bytes);
    hipLaunchKernelGGL(stream_x4, dim3(2048), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, sink); hipDeviceSynchronize();
    hipLaunchKernelGGL(stream_x1, dim3(2048), dim3(256), 0, 0, buf, bytes / 4, sink); hipDeviceSynchronize();
    hipLaunchKernelGGL(gather_row, dim3(2048), dim3(256), 0, 0, buf, bytes / 128, sink); hipDeviceSynchronize();
    printf("bytes touched: stream_x4 %zu, stream_x1 %zu, gather_row: %zu rows x 36 B used = %zu B in %zu distinct-ish 128-B lines (= %zu B of lines)\n",
           bytes, bytes, (bytes / 128 / 7) * 7, (bytes / 128 / 7) * 7 * 36, (bytes / 128 / 7) * 7, (bytes / 128 / 7) * 7 * 128);
    return 0;
}

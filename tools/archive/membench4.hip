// membench4.hip — why does the runtime's fill kernel write at 6.5 TB/s when a plain store loop gets 4.5-4.8?
// data value, grid shape and per-wave address pattern of a pure write stream.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
// MODE 0: constant value; 1: zero; 2: address-dependent "random" value
template <int MODE> __global__ void k_write(u32x4* a, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        u32x4 v;
        if (MODE == 0) v = (u32x4){0x3F800000u, 0x40000000u, 0x40400000u, 0x40800000u};
        else if (MODE == 1) v = (u32x4){0u, 0u, 0u, 0u};
        else { unsigned h = (unsigned)i * 2654435761u; h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; v = (u32x4){h, h * 3u + 1u, h ^ 0x9E3779B9u, h * 7u}; }
        a[i] = v;
    }
}
// every wave owns contiguous CH-byte chunks (CH/1024 stores of 1 KiB each), chunks dealt round-robin to waves
template <int CHK_KB> __global__ void k_write_chunk(u32x4* a, size_t n) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    const u32x4 v = {0x3F800000u, 0x40000000u, 0x40400000u, 0x40800000u};
    for (size_t c = wave * (CHK_KB * 64); c + CHK_KB * 64 <= n; c += nw * (CHK_KB * 64)) {
#pragma unroll
        for (int u = 0; u < CHK_KB; ++u) a[c + u * 64 + lane] = v;
    }
}
// each lane writes 64 contiguous bytes (4 stores)
__global__ void k_write_lane64(u32x4* a, size_t n) {
    const u32x4 v = {0x3F800000u, 0x40000000u, 0x40400000u, 0x40800000u};
    for (size_t i = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) * 4; i + 4 <= n; i += (size_t)gridDim.x * blockDim.x * 4) {
        a[i] = v; a[i + 1] = v; a[i + 2] = v; a[i + 3] = v;
    }
}
// block-contiguous: block b owns [b*n/G, (b+1)*n/G)
__global__ void k_write_blocked(u32x4* a, size_t n) {
    const u32x4 v = {0x3F800000u, 0x40000000u, 0x40400000u, 0x40800000u};
    const size_t per = n / gridDim.x, b0 = blockIdx.x * per;
    for (size_t i = threadIdx.x; i < per; i += blockDim.x) a[b0 + i] = v;
}
template <typename F> static double timeit(F f, int it = 10) {
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    f(); CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(a)); for (int i = 0; i < it; ++i) f(); CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
    float ms; CHK(hipEventElapsedTime(&ms, a, b)); return ms / it;
}
#define REP(name, expr) do { double ms = timeit([&] { expr; }); printf("%-46s: %.0f GB/s\n", name, bytes / 1e9 / (ms * 1e-3)); } while (0)
int main() {
    const size_t bytes = (size_t)1200 << 20, n = bytes / 16;
    u32x4* A; CHK(hipMalloc(&A, bytes)); CHK(hipMemset(A, 0, bytes));
    REP("hipMemsetAsync 0", CHK(hipMemsetAsync(A, 0, bytes, 0)));
    REP("hipMemsetAsync 0x5A", CHK(hipMemsetAsync(A, 0x5A, bytes, 0)));
    REP("hipMemsetD32Async 0x3F8CCCCD", CHK(hipMemsetD32Async((hipDeviceptr_t)A, 0x3F8CCCCD, bytes / 4, 0)));
    char nm[96];
    for (int grid : {256, 512, 1024, 2048, 8192, 65536}) for (int bs : {256, 1024}) {
        snprintf(nm, sizeof nm, "stride const  grid %5d x %4d", grid, bs);
        REP(nm, hipLaunchKernelGGL(k_write<0>, dim3(grid), dim3(bs), 0, 0, A, n));
    }
    REP("stride zero   grid  8192 x  256", hipLaunchKernelGGL(k_write<1>, dim3(8192), dim3(256), 0, 0, A, n));
    REP("stride random grid  8192 x  256", hipLaunchKernelGGL(k_write<2>, dim3(8192), dim3(256), 0, 0, A, n));
    REP("stride zero   grid  1024 x  256", hipLaunchKernelGGL(k_write<1>, dim3(1024), dim3(256), 0, 0, A, n));
    REP("stride random grid  1024 x  256", hipLaunchKernelGGL(k_write<2>, dim3(1024), dim3(256), 0, 0, A, n));
    REP("wave chunk 2 KiB  grid 4096 x 512", hipLaunchKernelGGL(k_write_chunk<2>, dim3(4096), dim3(512), 0, 0, A, n));
    REP("wave chunk 4 KiB  grid 4096 x 512", hipLaunchKernelGGL(k_write_chunk<4>, dim3(4096), dim3(512), 0, 0, A, n));
    REP("wave chunk 16 KiB grid 4096 x 512", hipLaunchKernelGGL(k_write_chunk<16>, dim3(4096), dim3(512), 0, 0, A, n));
    REP("wave chunk 4 KiB  grid 1024 x 256", hipLaunchKernelGGL(k_write_chunk<4>, dim3(1024), dim3(256), 0, 0, A, n));
    REP("lane 64 B         grid 8192 x 256", hipLaunchKernelGGL(k_write_lane64, dim3(8192), dim3(256), 0, 0, A, n));
    REP("block-contiguous  grid 8192 x 256", hipLaunchKernelGGL(k_write_blocked, dim3(8192), dim3(256), 0, 0, A, n));
    REP("block-contiguous  grid 1024 x 256", hipLaunchKernelGGL(k_write_blocked, dim3(1024), dim3(256), 0, 0, A, n));
    REP("block-contiguous  grid  256 x 1024", hipLaunchKernelGGL(k_write_blocked, dim3(256), dim3(1024), 0, 0, A, n));
    return 0;
}

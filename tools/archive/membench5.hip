// membench5.hip — 3-plane read-modify-write (the sweep's traffic) and pure read as a function of HOW the chip walks the
// address space: grid-stride with a small moving window (persistent waves) vs block-owned contiguous chunks.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
template <int U, bool NT> __device__ __forceinline__ void rmw_tiles(f32x4* base, size_t plane, size_t t, int lane) {
    f32x4 w[U], m[U], v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        f32x4* p = base + (t + u) * 64 + lane;
        if (NT) { w[u] = __builtin_nontemporal_load(p); m[u] = __builtin_nontemporal_load(p + plane); v[u] = __builtin_nontemporal_load(p + 2 * plane); }
        else { w[u] = *p; m[u] = p[plane]; v[u] = p[2 * plane]; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        f32x4* p = base + (t + u) * 64 + lane;
        m[u] = m[u] * 0.9f + w[u] * 0.1f; v[u] = v[u] * 0.999f + w[u] * w[u] * 0.001f; w[u] = w[u] - m[u] * 1e-3f;
        if (NT) { __builtin_nontemporal_store(w[u], p); __builtin_nontemporal_store(m[u], p + plane); __builtin_nontemporal_store(v[u], p + 2 * plane); }
        else { *p = w[u]; p[plane] = m[u]; p[2 * plane] = v[u]; }
    }
}
// grid-stride over tiles: wave w takes tiles [U*(w + k*nw), ...) — chip-wide window = nw*U KiB per plane
template <int U, bool NT> __global__ void __launch_bounds__(512, 4) k_rmw_gs(f32x4* base, size_t plane, size_t ntiles) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t t = wave * U; t + U <= ntiles; t += nw * U) rmw_tiles<U, NT>(base, plane, t, lane);
}
// block-owned chunk: block b owns tiles [b*CH, (b+1)*CH), its 8 waves stride through it (the sweep's decomposition)
template <int U, bool NT> __global__ void __launch_bounds__(512, 4) k_rmw_blk(f32x4* base, size_t plane, size_t ntiles, int CH) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t t0 = (size_t)blockIdx.x * CH;
    for (int t = wave * U; t + U <= CH; t += 8 * U) if (t0 + t + U <= ntiles) rmw_tiles<U, NT>(base, plane, t0 + t, lane);
}
template <bool NT> __global__ void __launch_bounds__(512, 4) k_read_gs(const f32x4* a, float* sink, size_t n) {
    f32x4 acc = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        acc += NT ? __builtin_nontemporal_load(a + i) : a[i];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}
template <typename F> static double timeit(F f, int it = 10) {
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    f(); CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(a)); for (int i = 0; i < it; ++i) f(); CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
    float ms; CHK(hipEventElapsedTime(&ms, a, b)); return ms / it;
}
int main() {
    const size_t bytes = (size_t)1200 << 20, n = bytes / 16;
    f32x4* A; float* sink; CHK(hipMalloc(&A, bytes)); CHK(hipMalloc(&sink, 64)); CHK(hipMemset(A, 0, bytes));
    const size_t pe = n / 3 / 4096 * 4096, ntiles = pe / 64;
    const double rw = 6.0 * pe * 16 / 1e9;
    for (int grid : {128, 256, 512, 1024, 2048, 4096, 16384}) {
        double a = timeit([&] { hipLaunchKernelGGL((k_rmw_gs<1, false>), dim3(grid), dim3(512), 0, 0, A, pe, ntiles); });
        double b = timeit([&] { hipLaunchKernelGGL((k_rmw_gs<2, false>), dim3(grid), dim3(512), 0, 0, A, pe, ntiles); });
        double c = timeit([&] { hipLaunchKernelGGL((k_rmw_gs<2, true>), dim3(grid), dim3(512), 0, 0, A, pe, ntiles); });
        double d = timeit([&] { hipLaunchKernelGGL((k_rmw_gs<4, true>), dim3(grid), dim3(512), 0, 0, A, pe, ntiles); });
        printf("rmw grid-stride grid %5d x 512: U1 %.0f  U2 %.0f  U2nt %.0f  U4nt %.0f GB/s (r+w)\n", grid, rw / (a * 1e-3), rw / (b * 1e-3), rw / (c * 1e-3), rw / (d * 1e-3));
    }
    for (int CH : {8, 32, 128, 512}) {
        const int grid = (int)((ntiles + CH - 1) / CH);
        double b = timeit([&] { hipLaunchKernelGGL((k_rmw_blk<2, false>), dim3(grid), dim3(512), 0, 0, A, pe, ntiles, CH); });
        double c = timeit([&] { hipLaunchKernelGGL((k_rmw_blk<2, true>), dim3(grid), dim3(512), 0, 0, A, pe, ntiles, CH); });
        printf("rmw block-owned %3d KiB chunks (grid %6d): U2 %.0f  U2nt %.0f GB/s (r+w)\n", CH, grid, rw / (b * 1e-3), rw / (c * 1e-3));
    }
    for (int grid : {256, 512, 1024, 4096, 16384}) {
        double a = timeit([&] { hipLaunchKernelGGL((k_read_gs<false>), dim3(grid), dim3(512), 0, 0, A, sink, n); });
        double b = timeit([&] { hipLaunchKernelGGL((k_read_gs<true>), dim3(grid), dim3(512), 0, 0, A, sink, n); });
        printf("read grid-stride grid %5d x 512: %.0f  nt %.0f GB/s\n", grid, bytes / 1e9 / (a * 1e-3), bytes / 1e9 / (b * 1e-3));
    }
    return 0;
}

// membench2.hip — pure read / pure write / copy ceilings (hipcc --offload-arch=gfx950 -O3 tools/membench2.hip -o tools/membench2)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
template <int NT> __global__ void k_read(const f32x4* a, float* sink, size_t n) {
    f32x4 acc = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        acc += NT ? __builtin_nontemporal_load(a + i) : a[i];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}
template <int NT> __global__ void k_write(f32x4* a, size_t n) {
    const f32x4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (NT) __builtin_nontemporal_store(v, a + i); else a[i] = v;
    }
}
template <int NT> __global__ void k_copy(const f32x4* a, f32x4* b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (NT) __builtin_nontemporal_store(__builtin_nontemporal_load(a + i), b + i); else b[i] = a[i];
    }
}
template <typename F> static double timeit(F f, int it = 10) {
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    f(); CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(a)); for (int i = 0; i < it; ++i) f(); CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
    float ms; CHK(hipEventElapsedTime(&ms, a, b)); return ms / it;
}
int main() {
    const size_t bytes = (size_t)1200 << 20, n = bytes / 16;
    f32x4 *A, *B; float* sink;
    CHK(hipMalloc(&A, bytes)); CHK(hipMalloc(&B, bytes)); CHK(hipMalloc(&sink, 64));
    CHK(hipMemset(A, 0, bytes)); CHK(hipMemset(B, 0, bytes));
    for (int grid : {2048, 8192}) {
        double ms;
        ms = timeit([&] { hipLaunchKernelGGL(k_read<0>, dim3(grid), dim3(256), 0, 0, A, sink, n); }); printf("grid %5d read       : %.0f GB/s\n", grid, bytes / 1e9 / (ms * 1e-3));
        ms = timeit([&] { hipLaunchKernelGGL(k_read<1>, dim3(grid), dim3(256), 0, 0, A, sink, n); }); printf("grid %5d read  nt   : %.0f GB/s\n", grid, bytes / 1e9 / (ms * 1e-3));
        ms = timeit([&] { hipLaunchKernelGGL(k_write<0>, dim3(grid), dim3(256), 0, 0, A, n); });      printf("grid %5d write      : %.0f GB/s\n", grid, bytes / 1e9 / (ms * 1e-3));
        ms = timeit([&] { hipLaunchKernelGGL(k_write<1>, dim3(grid), dim3(256), 0, 0, A, n); });      printf("grid %5d write nt   : %.0f GB/s\n", grid, bytes / 1e9 / (ms * 1e-3));
        ms = timeit([&] { hipLaunchKernelGGL(k_copy<0>, dim3(grid), dim3(256), 0, 0, A, B, n); });    printf("grid %5d copy       : %.0f GB/s (r+w)\n", grid, 2 * bytes / 1e9 / (ms * 1e-3));
        ms = timeit([&] { hipLaunchKernelGGL(k_copy<1>, dim3(grid), dim3(256), 0, 0, A, B, n); });    printf("grid %5d copy  nt   : %.0f GB/s (r+w)\n", grid, 2 * bytes / 1e9 / (ms * 1e-3));
    }
    return 0;
}

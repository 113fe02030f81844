// latbench.hip — what does the first global load of a freshly launched workgroup cost on MI355X?
// kernel W (many workgroups) writes a 1 MB buffer; kernel R (ONE 512-thread workgroup, like the chain) then reads it:
// cycles (s_memtime, ~2.35 GHz here... printed as ns via the wall-clock calibration below) from kernel entry to data.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__global__ void k_write(f32x4* a, size_t n, float v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = (f32x4){v, v, v, v};
}
// NLOAD independent 16 B loads per thread (stride = 8 KiB apart, like partial-sum chunks), then a dependent one
template <int NLOAD> __global__ void __launch_bounds__(512) k_read(const f32x4* a, const int* idx, long long* out, float* sink) {
    const long long t0 = __builtin_readcyclecounter();
    const int j = idx[0];                       // scalar load (the "record")
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const long long t1 = __builtin_readcyclecounter();
    f32x4 acc = {0, 0, 0, 0};
    f32x4 p[NLOAD];
#pragma unroll
    for (int u = 0; u < NLOAD; ++u) p[u] = a[(size_t)(u + j) * 512 + threadIdx.x];
#pragma unroll
    for (int u = 0; u < NLOAD; ++u) acc += p[u];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t2 = __builtin_readcyclecounter();
    const int k = ((int)acc[0]) & 1;            // dependent second level
    f32x4 q = a[(size_t)(NLOAD + 1 + k) * 512 + threadIdx.x];
    acc += q;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t3 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = t2 - t1; out[2] = t3 - t2; }
    if (acc[0] + acc[1] == 12345.f) sink[0] = acc[2];
}
template <int NLOAD> static void run(const char* what, f32x4* A, size_t n, int* idx, long long* out, float* sink, bool rewrite, int reps) {
    long long h[3], s[3] = {0, 0, 0};
    for (int r = 0; r < reps; ++r) {
        if (rewrite) hipLaunchKernelGGL(k_write, dim3(1024), dim3(256), 0, 0, A, n, (float)r);
        hipLaunchKernelGGL((k_read<NLOAD>), dim3(1), dim3(512), 0, 0, A, idx, out, sink);
        CHK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
        if (r) for (int i = 0; i < 3; ++i) s[i] += h[i];
    }
    printf("%-44s NLOAD %2d: record %6.0f  loads %6.0f  dependent load %6.0f cycles\n", what, NLOAD, s[0] / (double)(reps - 1), s[1] / (double)(reps - 1), s[2] / (double)(reps - 1));
}
int main() {
    const size_t bytes = (size_t)4 << 20, n = bytes / 16;
    f32x4* A; int* idx; long long* out; float* sink;
    CHK(hipMalloc(&A, bytes)); CHK(hipMalloc(&idx, 64)); CHK(hipMalloc(&out, 64)); CHK(hipMalloc(&sink, 64));
    CHK(hipMemset(A, 0, bytes)); CHK(hipMemset(idx, 0, 64));
    run<1>("data written by the previous kernel", A, n, idx, out, sink, true, 20);
    run<16>("data written by the previous kernel", A, n, idx, out, sink, true, 20);
    run<1>("data untouched since (re-read each launch)", A, n, idx, out, sink, false, 20);
    run<16>("data untouched since (re-read each launch)", A, n, idx, out, sink, false, 20);
    return 0;
}

// latbench2.hip — is the chain's slow entry a TLB effect?  ONE 512-thread workgroup reads 15 x 8 KiB either from one
// allocation or spread over 5 allocations (W / m / v planes, transposed arena, step buffers), after a kernel that streamed
// over 1.2 GB (what the sweep does to the translation caches).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__global__ void k_stream(f32x4* a, size_t n, float v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = a[i] * v;
}
struct Ptrs { const f32x4* p[5]; };
template <int NB> __global__ void __launch_bounds__(512) k_read(Ptrs P, long long* out, float* sink) {
    const long long t0 = __builtin_readcyclecounter();
    f32x4 q[15];
#pragma unroll
    for (int u = 0; u < 15; ++u) q[u] = P.p[u % NB][(size_t)(u / NB) * 512 + threadIdx.x];
    f32x4 acc = {0, 0, 0, 0};
    long long t[15];
#pragma unroll
    for (int u = 0; u < 15; ++u) {   // loads retire in order: time of arrival of each
        acc += q[u];
        asm volatile("" : "+v"(acc));
        if (u == 0 || u == 4 || u == 9 || u == 14) t[u] = __builtin_readcyclecounter();
    }
    if (threadIdx.x == 0) { out[0] = t[0] - t0; out[1] = t[4] - t0; out[2] = t[9] - t0; out[3] = t[14] - t0; }
    if (acc[0] + acc[1] == 12345.f) sink[0] = acc[2];
}
template <int NB> static void run(const char* what, Ptrs P, f32x4* big, size_t nbig, long long* out, float* sink, bool trash) {
    long long h[4], s[4] = {0, 0, 0, 0};
    const int reps = 20;
    for (int r = 0; r < reps; ++r) {
        if (trash) hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, 0, big, nbig, 1.0f);
        hipLaunchKernelGGL((k_read<NB>), dim3(1), dim3(512), 0, 0, P, out, sink);
        CHK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
        if (r) for (int i = 0; i < 4; ++i) s[i] += h[i];
    }
    printf("%-30s %d allocation(s), %s: load #1 %6.0f  #5 %6.0f  #10 %6.0f  #15 %6.0f cycles\n", what, NB, trash ? "after a 1.2 GB stream " : "back to back          ",
           s[0] / 19.0, s[1] / 19.0, s[2] / 19.0, s[3] / 19.0);
}
int main() {
    f32x4* big; const size_t nbig = ((size_t)1200 << 20) / 16;
    CHK(hipMalloc(&big, nbig * 16)); CHK(hipMemset(big, 0, nbig * 16));
    Ptrs P;
    for (int i = 0; i < 5; ++i) { f32x4* b; CHK(hipMalloc(&b, (size_t)64 << 20)); CHK(hipMemset(b, 0, (size_t)64 << 20)); P.p[i] = b + ((size_t)17 << 16); }
    long long* out; float* sink; CHK(hipMalloc(&out, 64)); CHK(hipMalloc(&sink, 64));
    run<1>("15 x 8 KiB", P, big, nbig, out, sink, false);
    run<5>("15 x 8 KiB", P, big, nbig, out, sink, false);
    run<1>("15 x 8 KiB", P, big, nbig, out, sink, true);
    run<5>("15 x 8 KiB", P, big, nbig, out, sink, true);
    return 0;
}

// membench.hip — what can a 3-plane read-modify-write stream (W, m, v tiles) reach on MI355X?
// Build: hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o /tmp/membench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// variant A: plain copy (read a, write b), grid-stride float4
__global__ void k_copy(const f32x4* a, f32x4* b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
// variant B: 3-plane RMW, each wave owns runs of `run` consecutive 1 KiB tiles; U tiles of loads in flight; NT = nontemporal
template <int U, int NT>
__global__ void __launch_bounds__(256) k_rmw(float* P, size_t plane, size_t ntiles, int run) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t r0 = wave * run; r0 < ntiles; r0 += nwaves * run) {
        for (int t0 = 0; t0 < run; t0 += U) {
            f32x4 w[U], m[U], v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t off = (r0 + t0 + u) * 256 + lane * 4;
                if (NT) {
                    w[u] = __builtin_nontemporal_load((const f32x4*)(P + off));
                    m[u] = __builtin_nontemporal_load((const f32x4*)(P + plane + off));
                    v[u] = __builtin_nontemporal_load((const f32x4*)(P + 2 * plane + off));
                } else {
                    w[u] = *(const f32x4*)(P + off);
                    m[u] = *(const f32x4*)(P + plane + off);
                    v[u] = *(const f32x4*)(P + 2 * plane + off);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t off = (r0 + t0 + u) * 256 + lane * 4;
                f32x4 a = w[u] * 0.999f + m[u] * 0.001f, b = m[u] * 0.9f + v[u], c = v[u] * 0.999f + w[u];
                if (NT) {
                    __builtin_nontemporal_store(a, (f32x4*)(P + off));
                    __builtin_nontemporal_store(b, (f32x4*)(P + plane + off));
                    __builtin_nontemporal_store(c, (f32x4*)(P + 2 * plane + off));
                } else {
                    *(f32x4*)(P + off) = a;
                    *(f32x4*)(P + plane + off) = b;
                    *(f32x4*)(P + 2 * plane + off) = c;
                }
            }
        }
    }
}
template <typename F> static double timeit(F f, int it = 10) {
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    f(); CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(a)); for (int i = 0; i < it; ++i) f(); CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
    float ms; CHK(hipEventElapsedTime(&ms, a, b)); return ms / it;
}
int main(int argc, char** argv) {
    const size_t mb = argc > 1 ? atoi(argv[1]) : 400;          // MB per plane
    const size_t plane = mb * 1024 * 1024 / 4, ntiles = plane / 256;
    float* P; CHK(hipMalloc(&P, plane * 4 * 3)); CHK(hipMemset(P, 0, plane * 4 * 3));
    const double gb = plane * 4.0 * 3 * 2 / 1e9;
    double ms = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, (const f32x4*)P, (f32x4*)(P + plane), plane / 4); });
    printf("copy 1 plane->1 plane          : %.0f GB/s\n", plane * 8.0 / 1e9 / (ms * 1e-3));
    for (int grid : {1024, 2048, 4096})
        for (int run : {8, 64}) {
#define RUN(U, NT) ms = timeit([&] { hipLaunchKernelGGL((k_rmw<U, NT>), dim3(grid), dim3(256), 0, 0, P, plane, ntiles, run); }); \
        printf("rmw3 grid %4d run %2d U=%d nt=%d : %.0f GB/s\n", grid, run, U, NT, gb / (ms * 1e-3));
            RUN(1, 0) RUN(2, 0) RUN(4, 0) RUN(8, 0) RUN(4, 1) RUN(8, 1)
        }
    return 0;
}

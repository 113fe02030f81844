// membench3.hip — does the store cache policy (sc0 / sc1 / nt bits of global_store on gfx950) or the load policy change
// the streaming ceilings?  pure write and 3-plane read-modify-write (the sweep's pattern), 1.2 GB working set.
// hipcc --offload-arch=gfx950 -O3 tools/membench3.hip -o tools/membench3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

#define ST(NAME, SUFFIX) __device__ __forceinline__ void NAME(f32x4* p, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off" SUFFIX :: "v"(p), "v"(v) : "memory"); }
ST(st0, "") ST(st1, " sc0") ST(st2, " sc1") ST(st3, " sc0 sc1") ST(st4, " nt") ST(st5, " sc0 nt") ST(st6, " sc1 nt") ST(st7, " sc0 sc1 nt")
#define LD(NAME, SUFFIX) __device__ __forceinline__ f32x4 NAME(const f32x4* p) { f32x4 v; asm volatile("global_load_dwordx4 %0, %1, off" SUFFIX : "=v"(v) : "v"(p) : "memory"); return v; }
LD(ld0, "") LD(ld1, " sc0") LD(ld2, " sc1") LD(ld3, " sc0 sc1") LD(ld4, " nt") LD(ld5, " sc0 nt") LD(ld6, " sc1 nt") LD(ld7, " sc0 sc1 nt")

template <int P> __device__ __forceinline__ void st(f32x4* p, f32x4 v) {
    if (P == 0) st0(p, v); else if (P == 1) st1(p, v); else if (P == 2) st2(p, v); else if (P == 3) st3(p, v);
    else if (P == 4) st4(p, v); else if (P == 5) st5(p, v); else if (P == 6) st6(p, v); else st7(p, v);
}
template <int P> __device__ __forceinline__ f32x4 ld(const f32x4* p) {
    if (P == 0) return ld0(p); else if (P == 1) return ld1(p); else if (P == 2) return ld2(p); else if (P == 3) return ld3(p);
    else if (P == 4) return ld4(p); else if (P == 5) return ld5(p); else if (P == 6) return ld6(p); else return ld7(p);
}
template <int P> __global__ void k_write(f32x4* a, size_t n) {
    const f32x4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) st<P>(a + i, v);
}
// 3 planes, tile-by-tile like the sweep: a wave owns runs of 1 KiB tiles, 2 tiles x 3 planes in flight
template <int PL, int PS> __global__ void __launch_bounds__(512, 4) k_rmw(f32x4* base, size_t plane, size_t ntiles) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t t = wave * 2; t < ntiles; t += nw * 2) {
        f32x4 w[2], m[2], v[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            f32x4* p = base + (t + u) * 64 + lane;
            w[u] = ld<PL>(p); m[u] = ld<PL>(p + plane); v[u] = ld<PL>(p + 2 * plane);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            f32x4* p = base + (t + u) * 64 + lane;
            m[u] = m[u] * 0.9f + w[u] * 0.1f; v[u] = v[u] * 0.999f + w[u] * w[u] * 0.001f; w[u] = w[u] - m[u] * 1e-3f;
            st<PS>(p, w[u]); st<PS>(p + plane, m[u]); st<PS>(p + 2 * plane, v[u]);
        }
    }
}
template <typename F> static double timeit(F f, int it = 10) {
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    f(); CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(a)); for (int i = 0; i < it; ++i) f(); CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
    float ms; CHK(hipEventElapsedTime(&ms, a, b)); return ms / it;
}
static const char* NM[8] = {"-", "sc0", "sc1", "sc0 sc1", "nt", "sc0 nt", "sc1 nt", "sc0 sc1 nt"};
template <int P> static void wr(f32x4* A, size_t n, size_t bytes) {
    double ms = timeit([&] { hipLaunchKernelGGL(k_write<P>, dim3(8192), dim3(256), 0, 0, A, n); });
    printf("write  store[%-10s]              : %.0f GB/s\n", NM[P], bytes / 1e9 / (ms * 1e-3));
}
template <int PL, int PS> static void rmw(f32x4* A, size_t plane_elems) {
    const size_t ntiles = plane_elems / 64;
    double ms = timeit([&] { hipLaunchKernelGGL((k_rmw<PL, PS>), dim3(4096), dim3(512), 0, 0, A, plane_elems, ntiles); });
    printf("rmw x3 load[%-10s] store[%-10s]: %.0f GB/s (r+w)\n", NM[PL], NM[PS], 6.0 * plane_elems * 16 / 1e9 / (ms * 1e-3));
}
int main() {
    const size_t bytes = (size_t)1200 << 20, n = bytes / 16;
    f32x4* A; CHK(hipMalloc(&A, bytes)); CHK(hipMemset(A, 0, bytes));
    { double ms = timeit([&] { CHK(hipMemsetAsync(A, 0, bytes, 0)); }); printf("hipMemsetAsync                         : %.0f GB/s\n", bytes / 1e9 / (ms * 1e-3)); }
    wr<0>(A, n, bytes); wr<1>(A, n, bytes); wr<2>(A, n, bytes); wr<3>(A, n, bytes); wr<4>(A, n, bytes); wr<5>(A, n, bytes); wr<6>(A, n, bytes); wr<7>(A, n, bytes);
    const size_t pe = n / 3 / 64 * 64;
    rmw<0, 0>(A, pe); rmw<4, 4>(A, pe); rmw<0, 4>(A, pe); rmw<4, 0>(A, pe); rmw<0, 2>(A, pe); rmw<0, 3>(A, pe); rmw<4, 6>(A, pe); rmw<4, 7>(A, pe);
    rmw<2, 2>(A, pe); rmw<3, 3>(A, pe); rmw<6, 6>(A, pe); rmw<7, 7>(A, pe); rmw<1, 1>(A, pe); rmw<5, 5>(A, pe);
    rmw<0, 0>(A, pe); rmw<4, 4>(A, pe);
    return 0;
}

// membench6.hip — round 3: (A) does a plain float4 COPY reach the 6.29 TB/s MI355X_MICROARCH.md quotes, and with which walk?
// (B) the sweep's 3-plane read-modify-write with the planes INTERLEAVED per 1 KiB tile ([W|m|v] contiguous: one 3 KiB read
// burst + one 3 KiB write burst per tile instead of six 1 KiB streams far apart) vs separate planes; (C) sensitivity to the
// distance between the planes (channel / bank aliasing of the three streams).
// hipcc --offload-arch=gfx950 -O3 tools/membench6.hip -o tools/membench6
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// ---- (A) copies
template <int U, bool NT> __global__ void k_copy_gs(const f32x4* a, f32x4* b, size_t n) {   // grid-stride, U loads in flight per lane
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(a + i + u * stride) : a[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], b + i + u * stride); else b[i + u * stride] = v[u]; }
    }
    for (; i < n; i += stride) b[i] = a[i];
}
template <int U, bool NT> __global__ void k_copy_blk(const f32x4* a, f32x4* b, size_t n, size_t per_block) {   // block-owned contiguous chunk
    const size_t lo = (size_t)blockIdx.x * per_block, hi = lo + per_block < n ? lo + per_block : n;
    for (size_t i = lo + threadIdx.x; i < hi; i += (size_t)U * blockDim.x) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) if (i + u * blockDim.x < hi) v[u] = NT ? __builtin_nontemporal_load(a + i + u * blockDim.x) : a[i + u * blockDim.x];
#pragma unroll
        for (int u = 0; u < U; ++u) if (i + u * blockDim.x < hi) { if (NT) __builtin_nontemporal_store(v[u], b + i + u * blockDim.x); else b[i + u * blockDim.x] = v[u]; }
    }
}
// ---- (B) / (C) read-modify-write of three planes: tile t of plane pl at base + (t * ts + pl * ps) * 64 float4
template <int U, bool NT> __device__ __forceinline__ void rmw(f32x4* base, size_t ts, size_t ps, size_t t, int lane) {
    f32x4 w[U], m[U], v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        f32x4* p = base + (t + u) * ts * 64 + lane;
        if (NT) { w[u] = __builtin_nontemporal_load(p); m[u] = __builtin_nontemporal_load(p + ps * 64); v[u] = __builtin_nontemporal_load(p + 2 * ps * 64); }
        else { w[u] = *p; m[u] = p[ps * 64]; v[u] = p[2 * ps * 64]; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        f32x4* p = base + (t + u) * ts * 64 + lane;
        m[u] = m[u] * 0.9f + w[u] * 0.1f; v[u] = v[u] * 0.999f + w[u] * w[u] * 0.001f; w[u] = w[u] - m[u] * 1e-3f;
        if (NT) { __builtin_nontemporal_store(w[u], p); __builtin_nontemporal_store(m[u], p + ps * 64); __builtin_nontemporal_store(v[u], p + 2 * ps * 64); }
        else { *p = w[u]; p[ps * 64] = m[u]; p[2 * ps * 64] = v[u]; }
    }
}
// block b owns tiles [b*CH, (b+1)*CH), its 8 waves stride through them (the sweep's decomposition)
template <int U, bool NT> __global__ void __launch_bounds__(512, 4) k_rmw_blk(f32x4* base, size_t ts, size_t ps, size_t ntiles, int CH) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t t0 = (size_t)blockIdx.x * CH;
    for (int t = wave * U; t + U <= CH; t += 8 * U) if (t0 + t + U <= ntiles) rmw<U, NT>(base, ts, ps, t0 + t, lane);
}
template <int U, bool NT> __global__ void __launch_bounds__(512, 4) k_rmw_gs(f32x4* base, size_t ts, size_t ps, size_t ntiles) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t t = wave * U; t + U <= ntiles; t += nw * U) rmw<U, NT>(base, ts, ps, t, lane);
}
// (D) OUT-OF-PLACE read-modify-write: planes read from `src`, results written to `dst` (ping-pong state buffers)
template <int U, bool NT> __global__ void __launch_bounds__(512, 4) k_rmw_oop(const f32x4* src, f32x4* dst, size_t ps, size_t ntiles, int CH) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t t0 = (size_t)blockIdx.x * CH;
    for (int tt = wave * U; tt + U <= CH; tt += 8 * U) {
        if (t0 + tt + U > ntiles) continue;
        f32x4 w[U], m[U], v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const f32x4* p = src + (t0 + tt + u) * 64 + lane;
            if (NT) { w[u] = __builtin_nontemporal_load(p); m[u] = __builtin_nontemporal_load(p + ps * 64); v[u] = __builtin_nontemporal_load(p + 2 * ps * 64); }
            else { w[u] = *p; m[u] = p[ps * 64]; v[u] = p[2 * ps * 64]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            f32x4* p = dst + (t0 + tt + u) * 64 + lane;
            m[u] = m[u] * 0.9f + w[u] * 0.1f; v[u] = v[u] * 0.999f + w[u] * w[u] * 0.001f; w[u] = w[u] - m[u] * 1e-3f;
            if (NT) { __builtin_nontemporal_store(w[u], p); __builtin_nontemporal_store(m[u], p + ps * 64); __builtin_nontemporal_store(v[u], p + 2 * ps * 64); }
            else { *p = w[u]; p[ps * 64] = m[u]; p[2 * ps * 64] = v[u]; }
        }
    }
}
template <typename F> static double timeit(F f, int it = 10) {
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    f(); CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(a)); for (int i = 0; i < it; ++i) f(); CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
    float ms; CHK(hipEventElapsedTime(&ms, a, b)); return ms / it;
}
int main() {
    const size_t bytes = (size_t)1536 << 20, n = bytes / 16;
    f32x4 *A, *B; CHK(hipMalloc(&A, bytes + (64 << 20))); CHK(hipMalloc(&B, bytes + (64 << 20)));
    CHK(hipMemset(A, 0, bytes)); CHK(hipMemset(B, 0, bytes));
    printf("## (A) copy of %.2f GB, GB/s = (bytes read + bytes written) / time\n", bytes / 1e9);
    { double ms = timeit([&] { CHK(hipMemcpyAsync(B, A, bytes, hipMemcpyDeviceToDevice, 0)); }); printf("hipMemcpyAsync D2D                    : %.0f GB/s (r+w)\n", 2 * bytes / 1e9 / (ms * 1e-3)); }
    for (int blk : {256, 512, 1024})
        for (int grid : {256, 512, 1024, 2048, 8192, 65536}) {
            double a = timeit([&] { hipLaunchKernelGGL((k_copy_gs<1, false>), dim3(grid), dim3(blk), 0, 0, A, B, n); });
            double b = timeit([&] { hipLaunchKernelGGL((k_copy_gs<1, true>), dim3(grid), dim3(blk), 0, 0, A, B, n); });
            double c = timeit([&] { hipLaunchKernelGGL((k_copy_gs<4, true>), dim3(grid), dim3(blk), 0, 0, A, B, n); });
            double d = timeit([&] { hipLaunchKernelGGL((k_copy_gs<8, true>), dim3(grid), dim3(blk), 0, 0, A, B, n); });
            printf("copy grid-stride %6d x %4d         : U1 %.0f  U1nt %.0f  U4nt %.0f  U8nt %.0f GB/s (r+w)\n", grid, blk, 2 * bytes / 1e9 / (a * 1e-3), 2 * bytes / 1e9 / (b * 1e-3), 2 * bytes / 1e9 / (c * 1e-3), 2 * bytes / 1e9 / (d * 1e-3));
        }
    for (size_t kib : {16, 64, 256, 1024, 4096}) {
        const size_t per = kib * 64;   // float4 per block
        const int grid = (int)((n + per - 1) / per);
        double a = timeit([&] { hipLaunchKernelGGL((k_copy_blk<1, true>), dim3(grid), dim3(256), 0, 0, A, B, n, per); });
        double b = timeit([&] { hipLaunchKernelGGL((k_copy_blk<4, true>), dim3(grid), dim3(256), 0, 0, A, B, n, per); });
        printf("copy block-owned %5zu KiB (grid %6d x 256): U1nt %.0f  U4nt %.0f GB/s (r+w)\n", kib, grid, 2 * bytes / 1e9 / (a * 1e-3), 2 * bytes / 1e9 / (b * 1e-3));
    }
    // ---- (B): 1.2 GB of state as three planes of pe float4 (ts = 1, ps = ntiles) or interleaved (ts = 3, ps = 1)
    const size_t pe = ((size_t)1200 << 20) / 16 / 3 / 4096 * 4096, ntiles = pe / 64;
    const double rw = 6.0 * pe * 16 / 1e9;
    printf("## (B) 3-plane read-modify-write of %.2f GB: separate planes vs [W|m|v] interleaved per 1 KiB tile\n", 3.0 * pe * 16 / 1e9);
    for (int CH : {8, 32, 128}) {
        const int grid = (int)((ntiles + CH - 1) / CH);
        double s2 = timeit([&] { hipLaunchKernelGGL((k_rmw_blk<2, true>), dim3(grid), dim3(512), 0, 0, A, (size_t)1, ntiles, ntiles, CH); });
        double i2 = timeit([&] { hipLaunchKernelGGL((k_rmw_blk<2, true>), dim3(grid), dim3(512), 0, 0, A, (size_t)3, (size_t)1, ntiles, CH); });
        double s4 = timeit([&] { hipLaunchKernelGGL((k_rmw_blk<4, true>), dim3(grid), dim3(512), 0, 0, A, (size_t)1, ntiles, ntiles, CH); });
        double i4 = timeit([&] { hipLaunchKernelGGL((k_rmw_blk<4, true>), dim3(grid), dim3(512), 0, 0, A, (size_t)3, (size_t)1, ntiles, CH); });
        printf("rmw block-owned %3d tiles/block (grid %6d): separate U2nt %.0f U4nt %.0f | interleaved U2nt %.0f U4nt %.0f GB/s (r+w)\n", CH, grid,
               rw / (s2 * 1e-3), rw / (s4 * 1e-3), rw / (i2 * 1e-3), rw / (i4 * 1e-3));
    }
    for (int grid : {256, 1024, 4096, 16384}) {
        double s2 = timeit([&] { hipLaunchKernelGGL((k_rmw_gs<2, true>), dim3(grid), dim3(512), 0, 0, A, (size_t)1, ntiles, ntiles); });
        double i2 = timeit([&] { hipLaunchKernelGGL((k_rmw_gs<2, true>), dim3(grid), dim3(512), 0, 0, A, (size_t)3, (size_t)1, ntiles); });
        double i4 = timeit([&] { hipLaunchKernelGGL((k_rmw_gs<4, true>), dim3(grid), dim3(512), 0, 0, A, (size_t)3, (size_t)1, ntiles); });
        printf("rmw grid-stride grid %5d x 512: separate U2nt %.0f | interleaved U2nt %.0f U4nt %.0f GB/s (r+w)\n", grid, rw / (s2 * 1e-3), rw / (i2 * 1e-3), rw / (i4 * 1e-3));
    }
    // ---- (C): separate planes, distance between planes = ntiles + extra tiles
    printf("## (C) separate planes, plane distance = %zu tiles + d (1 KiB tiles), block-owned 32 tiles, U2nt\n", ntiles);
    for (size_t d : {(size_t)0, (size_t)1, (size_t)4, (size_t)16, (size_t)64, (size_t)256, (size_t)1024, (size_t)4096, (size_t)16384}) {
        const int grid = (int)((ntiles + 31) / 32);
        double s2 = timeit([&] { hipLaunchKernelGGL((k_rmw_blk<2, true>), dim3(grid), dim3(512), 0, 0, A, (size_t)1, ntiles + d, ntiles, 32); });
        printf("  d = %6zu tiles: %.0f GB/s (r+w)\n", d, rw / (s2 * 1e-3));
    }
    printf("## (D) the same 3 planes read from buffer A, written to buffer B (out of place) vs in place; block-owned chunks\n");
    for (int CH : {8, 16, 32, 128}) {
        const int grid = (int)((ntiles + CH - 1) / CH);
        double ip = timeit([&] { hipLaunchKernelGGL((k_rmw_blk<2, true>), dim3(grid), dim3(512), 0, 0, A, (size_t)1, ntiles, ntiles, CH); });
        double o1 = timeit([&] { hipLaunchKernelGGL((k_rmw_oop<1, true>), dim3(grid), dim3(512), 0, 0, A, B, ntiles, ntiles, CH); });
        double o2 = timeit([&] { hipLaunchKernelGGL((k_rmw_oop<2, true>), dim3(grid), dim3(512), 0, 0, A, B, ntiles, ntiles, CH); });
        double o2p = timeit([&] { hipLaunchKernelGGL((k_rmw_oop<2, false>), dim3(grid), dim3(512), 0, 0, A, B, ntiles, ntiles, CH); });
        double o4 = timeit([&] { hipLaunchKernelGGL((k_rmw_oop<4, true>), dim3(grid), dim3(512), 0, 0, A, B, ntiles, ntiles, CH); });
        printf("  %3d tiles/block (grid %6d): in place U2nt %.0f | out of place U1nt %.0f U2nt %.0f U2 %.0f U4nt %.0f GB/s (r+w)\n", CH, grid,
               rw / (ip * 1e-3), rw / (o1 * 1e-3), rw / (o2 * 1e-3), rw / (o2p * 1e-3), rw / (o4 * 1e-3));
    }
    return 0;
}

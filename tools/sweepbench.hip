// sweepbench.hip — what the sweep's WORK STRUCTURE costs on the memory system, without its arithmetic (round 4).
// Three f32 planes (W | m | v) of 64 candidates x 1.04 M parameters are read-modify-written in 1 KiB tiles exactly like
// k_step<1,true,4,false> walks them (chunk = 8 row blocks x 4 k-blocks = 32 contiguous tiles per plane; wave w of a workgroup owns
// the 4 tiles of row block w; 2 tiles x 3 planes in flight per wave; nontemporal), in variants that add or remove one structural
// feature at a time:
//   A  8-wave workgroups, no staging, no barrier                      (the structure's ceiling)
//   B  A + a dependent pair of small loads -> LDS -> __syncthreads()  (what staging x / dy costs)
//   C  B + one 1 KiB partial-slab store per wave                      (the slab write)
//   D  single-wave workgroups (64 threads), as A                      (wave-granular dispatch)
//   E2 / E4  workgroups that stream 2 / 4 chunks, barrier + restage between chunks   (multi-chunk units)
//   F2 / F4  the same WITHOUT the barriers (waves run on independently)
//   U1 / U4  A with 1 / 4 tiles per plane in flight per wave
//   I  waves interleaved tile by tile through the chunk;  X  chunk order regrouped per XCD;  H  4-wave workgroups of half a chunk;
//   K  16-wave workgroups over two chunks side by side;  G2  8 waves x 8 contiguous tiles (chunk_cols 128);  P  plain (cached) accesses;  RO / WO  the read half / the write half alone (GB/s printed for the full read + write byte count: double it... no: halve it)
// usage: sweepbench [iters]   -> microseconds per pass over 3 x 266 MB (read + write = 1.6 GB), GB/s
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int U>
__device__ __forceinline__ void stream_tiles(float* W, size_t plane, size_t tile0, int ntiles, int lane, f32x4& acc) {
    for (int t0 = 0; t0 < ntiles; t0 += U) {
        f32x4 w[U], m[U], v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t off = (tile0 + t0 + u) * 256 + lane * 4;
            w[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(W + off));
            m[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(W + plane + off));
            v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(W + 2 * plane + off));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t off = (tile0 + t0 + u) * 256 + lane * 4;
            f32x4 g = w[u] * 0.001f + acc;
            m[u] = m[u] + 0.1f * (g - m[u]);
            v[u] = v[u] * 0.999f + 0.001f * g * g;
            w[u] = w[u] - 0.001f * m[u];
            acc += w[u] * 1e-9f;
            __builtin_nontemporal_store(w[u], reinterpret_cast<f32x4*>(W + off));
            __builtin_nontemporal_store(m[u], reinterpret_cast<f32x4*>(W + plane + off));
            __builtin_nontemporal_store(v[u], reinterpret_cast<f32x4*>(W + 2 * plane + off));
        }
    }
}

// MODE bit 0: staging + barrier; bit 1: slab store; NCH chunks per workgroup; BAR: barrier + restage between chunks
template <int U, int MODE, int NCH, bool BAR>
__global__ void __launch_bounds__(512, 4) k_wg(float* W, size_t plane, const int* idx, const float* rows, float* slabs) {
    __shared__ float xs[2048];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < NCH; ++c) {
        const size_t chunk = (size_t)blockIdx.x * NCH + c;
        if ((MODE & 1) && (c == 0 || BAR)) {
            if (c > 0) __syncthreads();
            const int r = idx[(chunk * 16 + (tid >> 5)) & 0xFFFF];               // dependent pair: index, then the row
            xs[tid * 4 & 2047] = rows[(size_t)r * 64 + (tid & 31) * 2];
            __syncthreads();
            acc[0] += xs[(tid * 7) & 2047] * 1e-12f;
        }
        stream_tiles<U>(W, plane, chunk * 32 + wave * 4, 4, lane, acc);
    }
    if (MODE & 2) *reinterpret_cast<f32x4*>(slabs + ((size_t)blockIdx.x * 8 + wave) * 256 + lane * 4) = acc;
    else if (acc[0] == 123.456f) slabs[0] = acc[1];
}

// I: wave w takes tiles w, w + 8, w + 16, w + 24 of the chunk (the workgroup's waves advance through the chunk side by side)
__global__ void __launch_bounds__(512, 4) k_inter(float* W, size_t plane, float* slabs) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < 4; t += 2) {
        f32x4 w[2], m[2], v[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const size_t off = ((size_t)blockIdx.x * 32 + wave + 8 * (t + u)) * 256 + lane * 4;
            w[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(W + off));
            m[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(W + plane + off));
            v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(W + 2 * plane + off));
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const size_t off = ((size_t)blockIdx.x * 32 + wave + 8 * (t + u)) * 256 + lane * 4;
            f32x4 g = w[u] * 0.001f + acc;
            m[u] = m[u] + 0.1f * (g - m[u]);
            v[u] = v[u] * 0.999f + 0.001f * g * g;
            w[u] = w[u] - 0.001f * m[u];
            acc += w[u] * 1e-9f;
            __builtin_nontemporal_store(w[u], reinterpret_cast<f32x4*>(W + off));
            __builtin_nontemporal_store(m[u], reinterpret_cast<f32x4*>(W + plane + off));
            __builtin_nontemporal_store(v[u], reinterpret_cast<f32x4*>(W + 2 * plane + off));
        }
    }
    if (acc[0] == 123.456f) slabs[0] = acc[1];
}
// X: chunk order regrouped so that the 8 XCDs each walk their own contiguous eighth of the planes (block b runs on XCD b % 8)
__global__ void __launch_bounds__(512, 4) k_xcd(float* W, size_t plane, float* slabs, size_t nchunks) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t chunk = (blockIdx.x % 8) * (nchunks / 8) + blockIdx.x / 8;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    stream_tiles<2>(W, plane, chunk * 32 + wave * 4, 4, lane, acc);
    if (acc[0] == 123.456f) slabs[0] = acc[1];
}
// H: 256-thread workgroups, each half a chunk (4 waves x 4 tiles)
__global__ void __launch_bounds__(256, 4) k_half(float* W, size_t plane, float* slabs) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    stream_tiles<2>(W, plane, (size_t)blockIdx.x * 16 + wave * 4, 4, lane, acc);
    if (acc[0] == 123.456f) slabs[0] = acc[1];
}
// P: plain (cached) loads and stores instead of nontemporal
__global__ void __launch_bounds__(512, 4) k_plain(float* W, size_t plane, float* slabs) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int t0 = 0; t0 < 4; t0 += 2) {
        f32x4 w[2], m[2], v[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const size_t off = ((size_t)blockIdx.x * 32 + wave * 4 + t0 + u) * 256 + lane * 4;
            w[u] = *reinterpret_cast<const f32x4*>(W + off);
            m[u] = *reinterpret_cast<const f32x4*>(W + plane + off);
            v[u] = *reinterpret_cast<const f32x4*>(W + 2 * plane + off);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const size_t off = ((size_t)blockIdx.x * 32 + wave * 4 + t0 + u) * 256 + lane * 4;
            f32x4 g = w[u] * 0.001f + acc;
            m[u] = m[u] + 0.1f * (g - m[u]);
            v[u] = v[u] * 0.999f + 0.001f * g * g;
            w[u] = w[u] - 0.001f * m[u];
            acc += w[u] * 1e-9f;
            *reinterpret_cast<f32x4*>(W + off) = w[u];
            *reinterpret_cast<f32x4*>(W + plane + off) = m[u];
            *reinterpret_cast<f32x4*>(W + 2 * plane + off) = v[u];
        }
    }
    if (acc[0] == 123.456f) slabs[0] = acc[1];
}
// R: read-only and W: write-only passes over the same walk (what each direction reaches alone)
__global__ void __launch_bounds__(512, 4) k_ro(float* W, size_t plane, float* slabs) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < 4; ++t) {
        const size_t off = ((size_t)blockIdx.x * 32 + wave * 4 + t) * 256 + lane * 4;
        acc += __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(W + off)) + __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(W + plane + off)) +
               __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(W + 2 * plane + off));
    }
    if (acc[0] == 123.456f) slabs[0] = acc[1];
}
__global__ void __launch_bounds__(512, 4) k_wo(float* W, size_t plane, float* slabs) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < 4; ++t) {
        const size_t off = ((size_t)blockIdx.x * 32 + wave * 4 + t) * 256 + lane * 4;
        __builtin_nontemporal_store(z, reinterpret_cast<f32x4*>(W + off));
        __builtin_nontemporal_store(z, reinterpret_cast<f32x4*>(W + plane + off));
        __builtin_nontemporal_store(z, reinterpret_cast<f32x4*>(W + 2 * plane + off));
    }
}

// K: 16-wave workgroups (1024 threads), two chunks side by side: every wave still owns 4 tiles, all 64 tiles of the pair in flight together
__global__ void __launch_bounds__(1024, 4) k_wide(float* W, size_t plane, float* slabs) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    stream_tiles<2>(W, plane, (size_t)blockIdx.x * 64 + wave * 4, 4, lane, acc);
    if (acc[0] == 123.456f) slabs[0] = acc[1];
}
// G2: 8-wave workgroups over a 64-tile chunk laid out [rb][8 kb]: every wave streams 8 contiguous tiles (the chunk_cols = 128 layout)
__global__ void __launch_bounds__(512, 4) k_cc128(float* W, size_t plane, float* slabs) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    stream_tiles<2>(W, plane, (size_t)blockIdx.x * 64 + wave * 8, 8, lane, acc);
    if (acc[0] == 123.456f) slabs[0] = acc[1];
}

template <int U>
__global__ void __launch_bounds__(64, 4) k_wave(float* W, size_t plane, float* slabs) {
    const int lane = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    stream_tiles<U>(W, plane, (size_t)blockIdx.x * 4, 4, lane, acc);
    if (acc[0] == 123.456f) slabs[0] = acc[1];
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20;
    const size_t nchunks = 64 * 127;                       // ~ 64 candidates x 127 chunk-equivalents = 8128 chunks of 32 tiles
    const size_t plane = nchunks * 32 * 256;               // floats
    float *W, *rows, *slabs;
    int* idx;
    CHK(hipMalloc(&W, plane * 3 * sizeof(float)));
    CHK(hipMemset(W, 0, plane * 3 * sizeof(float)));
    CHK(hipMalloc(&rows, 10000 * 64 * sizeof(float)));
    CHK(hipMemset(rows, 0, 10000 * 64 * sizeof(float)));
    CHK(hipMalloc(&slabs, nchunks * 8 * 256 * sizeof(float)));
    std::vector<int> h(65536);
    for (int i = 0; i < 65536; ++i) h[i] = (i * 7919) % 10000;
    CHK(hipMalloc(&idx, 65536 * sizeof(int)));
    CHK(hipMemcpy(idx, h.data(), 65536 * sizeof(int), hipMemcpyHostToDevice));
    hipEvent_t a, b;
    CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    const double bytes = (double)plane * 4 * 3 * 2;
    printf("# 3 planes x %.1f MB, read + write %.3f GB per pass, %d passes per variant\n", plane * 4 / 1e6, bytes / 1e9, iters);
#define RUN(name, launch) do { auto f_ = [&]() { launch; }; f_(); CHK(hipDeviceSynchronize()); CHK(hipEventRecord(a)); for (int i = 0; i < iters; ++i) f_(); CHK(hipEventRecord(b)); \
        CHK(hipEventSynchronize(b)); float ms; CHK(hipEventElapsedTime(&ms, a, b)); printf("%-4s %8.1f us/pass  %7.1f GB/s\n", name, ms / iters * 1e3, bytes / (ms / iters * 1e-3) / 1e9); } while (0)
    for (int rep = 0; rep < 2; ++rep) {
        RUN("A", hipLaunchKernelGGL((k_wg<2, 0, 1, false>), dim3(nchunks), dim3(512), 0, 0, W, plane, idx, rows, slabs));
        RUN("B", hipLaunchKernelGGL((k_wg<2, 1, 1, false>), dim3(nchunks), dim3(512), 0, 0, W, plane, idx, rows, slabs));
        RUN("C", hipLaunchKernelGGL((k_wg<2, 3, 1, false>), dim3(nchunks), dim3(512), 0, 0, W, plane, idx, rows, slabs));
        RUN("D", hipLaunchKernelGGL((k_wave<2>), dim3(nchunks * 8), dim3(64), 0, 0, W, plane, slabs));
        RUN("E2", hipLaunchKernelGGL((k_wg<2, 1, 2, true>), dim3(nchunks / 2), dim3(512), 0, 0, W, plane, idx, rows, slabs));
        RUN("E4", hipLaunchKernelGGL((k_wg<2, 1, 4, true>), dim3(nchunks / 4), dim3(512), 0, 0, W, plane, idx, rows, slabs));
        RUN("F2", hipLaunchKernelGGL((k_wg<2, 1, 2, false>), dim3(nchunks / 2), dim3(512), 0, 0, W, plane, idx, rows, slabs));
        RUN("F4", hipLaunchKernelGGL((k_wg<2, 1, 4, false>), dim3(nchunks / 4), dim3(512), 0, 0, W, plane, idx, rows, slabs));
        RUN("U1", hipLaunchKernelGGL((k_wg<1, 0, 1, false>), dim3(nchunks), dim3(512), 0, 0, W, plane, idx, rows, slabs));
        RUN("I", hipLaunchKernelGGL(k_inter, dim3(nchunks), dim3(512), 0, 0, W, plane, slabs));
        RUN("X", hipLaunchKernelGGL(k_xcd, dim3(nchunks), dim3(512), 0, 0, W, plane, slabs, nchunks));
        RUN("H", hipLaunchKernelGGL(k_half, dim3(nchunks * 2), dim3(256), 0, 0, W, plane, slabs));
        RUN("P", hipLaunchKernelGGL(k_plain, dim3(nchunks), dim3(512), 0, 0, W, plane, slabs));
        RUN("RO", hipLaunchKernelGGL(k_ro, dim3(nchunks), dim3(512), 0, 0, W, plane, slabs));
        RUN("WO", hipLaunchKernelGGL(k_wo, dim3(nchunks), dim3(512), 0, 0, W, plane, slabs));
        RUN("K", hipLaunchKernelGGL(k_wide, dim3(nchunks / 2), dim3(1024), 0, 0, W, plane, slabs));
        RUN("G2", hipLaunchKernelGGL(k_cc128, dim3(nchunks / 2), dim3(512), 0, 0, W, plane, slabs));
        RUN("U4", hipLaunchKernelGGL((k_wg<4, 0, 1, false>), dim3(nchunks), dim3(512), 0, 0, W, plane, idx, rows, slabs));
    }
    return 0;
}

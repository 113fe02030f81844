// cluster_handoff.hip — what a MULTI-CU chain would pay for its exchanges (round 3).
// A cluster of C workgroups shares one candidate's chain: each member owns R/C output columns of every cell, so between two
// products every member needs the other members' slices of out_i (forward) / dy_i (backward): an all-gather of C slices of
// B x R/C floats, 8 times per train step (4 forward cells + 4 backward cells at L = 4).  This bench runs exactly that exchange
// pattern with no arithmetic between the gathers, for NCL concurrent clusters, and reports microseconds per step:
//   F: 16-byte write-through (sc1) stores -> s_waitcnt vmcnt(0) -> barrier -> relaxed flag store; consumers: lane 0 polls the C flags,
//      barrier, sc1 loads of the slices (the machinery of k_step_same / k_president);
//   G: 8-byte {value, tag} granules, one sc1 store each, consumers poll the granules themselves (MI355X_MICROARCH.md handoff-1to1).
// same_xcd = 1 places a cluster's members on ONE XCD (block b runs on XCD b % 8).
// hipcc --offload-arch=gfx950 -O3 tools/cluster_handoff.hip -o tools/cluster_handoff
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, -1, 0x00020000); }
__device__ __forceinline__ uint32_t ld_relaxed(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

constexpr int B = 16, R = 128, GATHERS = 8;

// buf: per cluster, per parity (2), per gather (8): [B][R] floats (F) or [B][R] granules (G); flags: per cluster [2][8][C]
template <int C, bool GRAN>
__global__ void __launch_bounds__(512, 2) k_cluster(float* buf, uint32_t* flags, int ncl, int same_xcd, int steps, int work_cycles, uint32_t* bad) {
    const int bid = blockIdx.x, tid = threadIdx.x;
    int cl, me;
    if (same_xcd) { const int x = bid % 8, j = bid / 8; cl = x + 8 * (j / C); me = j % C; }   // members of cluster cl: blocks with the same b % 8
    else { cl = bid / C; me = bid % C; }
    if (cl >= ncl) return;
    constexpr int SL = R / C;                 // my columns
    __shared__ float full[B * R];
    __shared__ int ok;
    const size_t gstride = (size_t)B * R * (GRAN ? 2 : 1);
    float* cbuf = buf + (size_t)cl * 2 * GATHERS * gstride;
    uint32_t* cfl = flags + (size_t)cl * 2 * GATHERS * 8;
    float acc = (float)me;
    for (int t = 0; t < steps; ++t) {
        const uint32_t tag = (uint32_t)t + 1u;
        for (int g = 0; g < GATHERS; ++g) {
            float* gb = cbuf + (size_t)((t & 1) * GATHERS + g) * gstride;
            uint32_t* gf = cfl + ((t & 1) * GATHERS + g) * 8;
            if (work_cycles > 0) {            // stand-in for the member's share of the cell (MFMA + VALU), cycles
                const long long t0 = clock64();
                while (clock64() - t0 < work_cycles) { }
            }
            // ---- publish my slice: B x SL floats = B*SL/4 float4 (F) or B*SL granules (G)
            if constexpr (!GRAN) {
                for (int e = tid; e < B * SL / 4; e += 512) {
                    const int b = e / (SL / 4), c4 = e % (SL / 4);
                    f32x4 v = {acc + e, acc, acc, (float)tag};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc(gb), (int)((b * R + me * SL + c4 * 4) * 4), 0, 16);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) __hip_atomic_store(gf + me, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // ---- wait for everybody's flag, then read the whole [B][R]
                if (tid == 0) {
                    int good = 1;
                    for (int m = 0; m < C; ++m) {
                        uint32_t spins = 0;
                        while (ld_relaxed(gf + m) < tag) { __builtin_amdgcn_s_sleep(1); if (++spins > (1u << 22)) { good = 0; break; } }
                    }
                    ok = good;
                }
                __syncthreads();
                if (!ok) { if (tid == 0) atomicAdd(bad, 1u); return; }
                for (int e = tid; e < B * R / 4; e += 512) {
                    const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rsrc(gb), e * 16, 0, 16);
                    *reinterpret_cast<f32x4*>(full + e * 4) = __builtin_bit_cast(f32x4, r);
                }
                __syncthreads();
            } else {
                for (int e = tid; e < B * SL; e += 512) {
                    const int b = e / SL, c = e % SL;
                    u32x2 v = {__float_as_uint(acc + e), tag};
                    __builtin_amdgcn_raw_buffer_store_b64(v, rsrc(gb), (int)((b * R + me * SL + c) * 8), 0, 16);
                }
                // every lane polls its own granules (B*R / 512 = 4 per lane) until their tags are current
                int good = 1;
                for (int e = tid; e < B * R; e += 512) {
                    uint32_t spins = 0;
                    u32x2 r;
                    do {
                        r = __builtin_amdgcn_raw_buffer_load_b64(rsrc(gb), e * 8, 0, 16);
                        if (r.y == tag) break;
                        __builtin_amdgcn_s_sleep(1);
                    } while (++spins < (1u << 22));
                    if (r.y != tag) good = 0;
                    full[e] = __uint_as_float(r.x);
                }
                if (!good) { atomicAdd(bad, 1u); return; }
                __syncthreads();
            }
            acc += full[(tid * 7 + g) % (B * R)] * 1e-6f;
        }
    }
    if (acc == 12345.678f) buf[0] = acc;
}

template <int C, bool GRAN>
static void run(float* buf, uint32_t* flags, uint32_t* bad, int ncl, int same_xcd, int steps, int work) {
    const int grid = same_xcd ? 8 * C * ((ncl + 7) / 8) : ncl * C;
    CHK(hipMemset(flags, 0, 4096 * 16 * 8 * 4));
    CHK(hipMemset(buf, 0, (size_t)64 * 2 * GATHERS * B * R * 2 * 4));
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    hipLaunchKernelGGL((k_cluster<C, GRAN>), dim3(grid), dim3(512), 0, 0, buf, flags, ncl, same_xcd, 50, work, bad);
    CHK(hipDeviceSynchronize());
    CHK(hipMemset(flags, 0, 4096 * 16 * 8 * 4));
    CHK(hipMemset(buf, 0, (size_t)64 * 2 * GATHERS * B * R * 2 * 4));
    CHK(hipEventRecord(a));
    hipLaunchKernelGGL((k_cluster<C, GRAN>), dim3(grid), dim3(512), 0, 0, buf, flags, ncl, same_xcd, steps, work, bad);
    CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
    float ms; CHK(hipEventElapsedTime(&ms, a, b));
    uint32_t hb; CHK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
    printf("C=%d %s clusters=%2d same_xcd=%d work=%5d cyc: %7.2f us/step = %5.2f us per all-gather (net of work %5.2f)%s\n", C, GRAN ? "granules" : "flags   ", ncl, same_xcd, work,
           ms * 1e3 / steps, ms * 1e3 / steps / GATHERS, ms * 1e3 / steps / GATHERS - work / 2400.0, hb ? "  TIMEOUTS!" : "");
}

int main() {
    float* buf; uint32_t *flags, *bad;
    CHK(hipMalloc(&buf, (size_t)64 * 2 * GATHERS * B * R * 2 * 4));
    CHK(hipMalloc(&flags, 4096 * 16 * 8 * 4));
    CHK(hipMalloc(&bad, 4)); CHK(hipMemset(bad, 0, 4));
    const int steps = 2000;
    for (int ncl : {1, 6}) for (int sx : {0, 1}) {
        run<2, false>(buf, flags, bad, ncl, sx, steps, 0);
        run<4, false>(buf, flags, bad, ncl, sx, steps, 0);
        run<2, true>(buf, flags, bad, ncl, sx, steps, 0);
        run<4, true>(buf, flags, bad, ncl, sx, steps, 0);
    }
    // with the member's share of a cell between gathers (a cell is ~9,000 cycles on one CU today: 4 members -> ~2,300 each)
    run<4, false>(buf, flags, bad, 6, 1, steps, 2300);
    run<4, true>(buf, flags, bad, 6, 1, steps, 2300);
    run<2, false>(buf, flags, bad, 6, 1, steps, 4500);
    run<2, true>(buf, flags, bad, 6, 1, steps, 4500);
    return 0;
}

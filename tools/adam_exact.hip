// adam_exact — is the engine's written-out Adam arithmetic (common.hip.h: sqrt_rn / div_by, packed adam4) bit-identical to the
// library forms (sqrtf, operator/) it replaces?  Both are compiled here (namespace lib below holds the library form; the
// engine builds it under -DMFAS_ADAM_LIBRARY_FORMS) and run on the same inputs; mismatching bits of w / m / v are counted per input class.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Imfas_amd/csrc tools/adam_exact.hip -o tools/adam_exact
//   run:   tools/adam_exact [millions of elements per class, default 64]
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#pragma clang fp contract(off)
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "mfas_hip.h"
#include "common.hip.h"

namespace lib {   // the library forms, element by element
__device__ __forceinline__ void adam1(float& w, float& m, float& v, float g, const AdamC& c) {
    g = g + c.wd * w;
    m = m + c.w1 * (g - m);
    v = v * c.b2;
    v = v + (c.w2 * g) * g;
    const float denom = sqrtf(v) / c.bc2s + c.eps;
    w = w - c.ss * (m / denom);
}
}

__device__ __forceinline__ uint32_t mix(uint32_t x) { return lowbias32(x * 0x9E3779B9u + 0x7F4A7C15u); }
__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }
// log-uniform magnitude in [2^lo, 2^hi) with random mantissa and sign
__device__ __forceinline__ float logu(uint32_t x, int lo, int hi, bool sgn) {
    const uint32_t e = (uint32_t)(lo + 127) + mix(x) % (uint32_t)(hi - lo);
    const uint32_t bits = (e << 23) | (mix(x + 1) & 0x7FFFFFu) | (sgn && (mix(x + 2) & 1) ? 0x80000000u : 0u);
    return __builtin_bit_cast(float, bits);
}

struct Counts { unsigned long long n, dw, dm, dv, dw_ulp_max, nan_mismatch; };

// cls 0: training-like (|w| ~ 1e-3..1, g ~ 1e-8..1e-1, m ~ g-like, v ~ g^2-like)       -> must be bit-identical
// cls 1: decayed state (dead columns): g = 0, m in [2^-100, 2^-20), v in [2^-96, 2^-30) -> must be bit-identical
// cls 2: deep underflow: m in [2^-149, 2^-100), v in [2^-149, 2^-96)                    -> reported (<= 1 ulp of w expected, 0 in practice)
// cls 3: zeros / padding: w = m = v = g = 0, and g = 0 with v = 0, m != 0               -> must be bit-identical
// cls 4: first step from zero moments (m = v = 0, any g)                                -> must be bit-identical
// cls 5: wide: everything log-uniform over [2^-60, 2^20)                                -> must be bit-identical
// cls 6: adam_eps = 0 with v = 0 (not reachable from training: v = 0 implies m = 0): 0/0 is NaN in both forms; m/0 is +-inf in
//        the library form and NaN here -> reported (both leave the weight non-finite)
__global__ void k_cmp(Counts* out, int cls, uint64_t n, uint32_t seed, int nsteps) {
    unsigned long long dw = 0, dm = 0, dv = 0, mx = 0, nn = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t h = mix((uint32_t)i ^ seed) + (uint32_t)(i >> 32);
        float w, m, v, g;
        AdamC c;
        const int t = 1 + (int)(mix(h + 9) % (uint32_t)nsteps);
        c.w1 = 0.1f; c.b2 = 0.999f; c.w2 = 0.001f; c.eps = 1e-8f; c.wd = (mix(h + 10) & 1) ? 1e-4f : 0.0f;
        const double bc1 = 1.0 - pow(0.9, t), bc2 = 1.0 - pow(0.999, t);
        c.ss = (float)(1e-3 * (0.001 + 0.999 * u01(mix(h + 11))) / bc1);
        c.bc2s = (float)sqrt(bc2);
        if (cls == 0) {
            w = logu(h, -10, 1, true); g = logu(h + 3, -27, -3, true);
            m = g * (0.05f + u01(mix(h + 6))) * ((mix(h + 7) & 1) ? 1.f : -1.f);
            v = g * g * (0.01f + 4.0f * u01(mix(h + 8)));
        } else if (cls == 1) {
            w = logu(h, -10, 1, true); g = 0.0f; c.wd = 0.0f;
            m = logu(h + 3, -100, -20, true); v = logu(h + 6, -96, -30, false);
        } else if (cls == 2) {
            w = logu(h, -10, 1, true); g = 0.0f; c.wd = 0.0f;
            const uint32_t mb = 1u + mix(h + 3) % ((27u << 23) - 1u);      // (0, 2^-100)
            const uint32_t vb = 1u + mix(h + 6) % ((31u << 23) - 1u);      // (0, 2^-96)
            m = __builtin_bit_cast(float, mb | ((mix(h + 4) & 1) ? 0x80000000u : 0u));
            v = __builtin_bit_cast(float, vb);
        } else if (cls == 3) {
            w = (mix(h) & 1) ? 0.0f : logu(h, -10, 1, true); g = 0.0f; v = 0.0f; c.wd = 0.0f;
            m = (mix(h + 1) & 3) ? 0.0f : logu(h + 3, -60, -3, true);
        } else if (cls == 6) {                                              // eps = 0 and v = 0: 0/0 (NaN in both) and m/0
            w = logu(h, -10, 1, true); g = 0.0f; v = 0.0f; c.wd = 0.0f; c.eps = 0.0f;
            m = (mix(h + 1) & 1) ? 0.0f : logu(h + 3, -60, -3, true);
        } else if (cls == 4) {
            w = logu(h, -10, 1, true); g = logu(h + 3, -60, 3, true); m = 0.0f; v = 0.0f;
        } else {
            w = logu(h, -60, 20, true); g = logu(h + 3, -60, 20, true); m = logu(h + 6, -60, 20, true); v = logu(h + 9, -60, 20, false);
        }
        // element-wise forms
        float w1 = w, m1 = m, v1 = v, w2 = w, m2 = m, v2 = v;
        adam1(w1, m1, v1, g, c);
        lib::adam1(w2, m2, v2, g, c);
        // packed form on (x, x, x, x) must equal the scalar one
        f32x4 W4 = (f32x4)(w), M4 = (f32x4)(m), V4 = (f32x4)(v);
        adam4(W4, M4, V4, (f32x4)(g), c);
        const uint32_t bw1 = __builtin_bit_cast(uint32_t, w1), bw2 = __builtin_bit_cast(uint32_t, w2);
        const bool nan1 = w1 != w1, nan2 = w2 != w2;
        if (nan1 || nan2) { nn += nan1 != nan2; }
        else if (bw1 != bw2) {
            ++dw;
            const int32_t a = (int32_t)bw1 < 0 ? (int32_t)(0x80000000u - bw1) : (int32_t)bw1, b = (int32_t)bw2 < 0 ? (int32_t)(0x80000000u - bw2) : (int32_t)bw2;
            const unsigned long long d = (unsigned long long)(a > b ? (int64_t)a - b : (int64_t)b - a);
            mx = d > mx ? d : mx;
        }
        dm += __builtin_bit_cast(uint32_t, m1) != __builtin_bit_cast(uint32_t, m2);
        dv += __builtin_bit_cast(uint32_t, v1) != __builtin_bit_cast(uint32_t, v2);
        // packed vs scalar (own counter folded into dm's high half: must be 0)
        if (!(nan1) && (__builtin_bit_cast(uint32_t, (float)W4[2]) != bw1 || __builtin_bit_cast(uint32_t, (float)M4[1]) != __builtin_bit_cast(uint32_t, m1) ||
                        __builtin_bit_cast(uint32_t, (float)V4[3]) != __builtin_bit_cast(uint32_t, v1)))
            dm += 1ull << 40;
    }
    atomicAdd(&out->dw, dw); atomicAdd(&out->dm, dm); atomicAdd(&out->dv, dv); atomicMax(&out->dw_ulp_max, mx); atomicAdd(&out->nan_mismatch, nn);
}

// sqrt_rn against sqrtf over every non-negative float bit pattern: [0] mismatches for x >= 2^-102 (must be 0), [1] below,
// [2] the largest |sqrt_rn - sqrtf| below (as float bits), [3] NaN results below
__global__ void k_sqrt_all(unsigned long long* out) {
    unsigned long long hi = 0, lo = 0, mx = 0, nn = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= 0x7F800000ull; i += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __builtin_bit_cast(float, (uint32_t)i);
        const uint32_t a = __builtin_bit_cast(uint32_t, sqrt_rn(x)), b = __builtin_bit_cast(uint32_t, sqrtf(x));
        if (i == 0x7F800000ull) continue;                     // +inf: NaN here (rsq = 0, 0 * inf), inf in the library — a diverged state
        if (a != b) {
            if (i >= ((uint32_t)(127 - 102) << 23)) ++hi; else ++lo;
            const float fa = __builtin_bit_cast(float, a), fb = __builtin_bit_cast(float, b);
            if (fa != fa) ++nn;
            else { const unsigned long long d = __builtin_bit_cast(uint32_t, fabsf(fa - fb)); mx = d > mx ? d : mx; }
        }
    }
    atomicAdd(&out[0], hi); atomicAdd(&out[1], lo); atomicMax(&out[2], mx); atomicAdd(&out[3], nn);
}

int main(int argc, char** argv) {
    const uint64_t n = (uint64_t)(argc > 1 ? atoi(argv[1]) : 64) * 1000000ull;
    Counts* d;
    if (hipMalloc(&d, sizeof(Counts)) != hipSuccess) { fprintf(stderr, "no device\n"); return 2; }
    const char* names[7] = {"training-like", "decayed (g=0, normal-range m,v)", "deep underflow (m<2^-100, v<2^-96)", "zeros / padding", "first step (m=v=0)", "wide log-uniform", "eps=0, v=0 (inf vs NaN)"};
    int bad = 0;
    {
        unsigned long long* ds; unsigned long long hs[4];
        (void)hipMalloc(&ds, 32); (void)hipMemset(ds, 0, 32);
        k_sqrt_all<<<4096, 256>>>(ds);
        (void)hipMemcpy(hs, ds, 32, hipMemcpyDeviceToHost);
        const uint32_t mxb = (uint32_t)hs[2]; float mxf; memcpy(&mxf, &mxb, 4);
        printf("sqrt_rn vs sqrtf over all 2,139,095,040 finite non-negative floats: %llu differ at x >= 2^-102; below: %llu differ, largest |difference| %.3g, %llu NaN\n", hs[0], hs[1], (double)mxf, hs[3]);
        if (hs[0]) bad = 1;
    }
    for (int cls = 0; cls < 7; ++cls) {
        (void)hipMemset(d, 0, sizeof(Counts));
        k_cmp<<<2048, 256>>>(d, cls, n, 0xC0FFEEu + cls, 50000);
        Counts h;
        (void)hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost);
        const unsigned long long packed = h.dm >> 40, dmm = h.dm & ((1ull << 40) - 1);
        printf("class %d %-38s n=%llu  w differs %llu (max %llu ulp)  m differs %llu  v differs %llu  NaN-ness differs %llu  packed!=scalar %llu\n",
               cls, names[cls], (unsigned long long)n, h.dw, h.dw_ulp_max, dmm, h.dv, h.nan_mismatch, packed);
        if (cls != 2 && cls != 6 && (h.dw || dmm || h.dv || h.nan_mismatch)) bad = 1;
        if (packed || (cls != 6 && h.dw_ulp_max > 1)) bad = 1;
    }
    printf(bad ? "MISMATCH\n" : "IDENTICAL (classes 0,1,3,4,5 bit for bit; class 2 within 1 ulp of w; class 6 non-finite in both)\n");
    return bad;
}

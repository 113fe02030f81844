// common.hip.h — device-side records (SegDesc, TapDesc, CandDev, Geo), small math helpers and LDS row staging
// (part of the single translation unit mfas_hip.hip; see the header comment there and DESIGN.md)
#pragma once

typedef float f32x4 __attribute__((ext_vector_type(4)));
// Explicit LDS / global address spaces for pointers whose provenance the compiler cannot see (a buffer picked by a runtime
// index, "in LDS if it fits, else scratch"): a generic pointer is accessed with FLAT instructions, and every flat load is
// followed by s_waitcnt vmcnt(0) lgkmcnt(0) -- it drains all of the wave's prefetches in flight.
typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));   // (HIP's uint4 / uint2 classes have no address-space-qualified operators)
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
#define MFAS_LDS __attribute__((address_space(3)))
#define MFAS_GLB __attribute__((address_space(1)))
template <typename T> __device__ __forceinline__ MFAS_LDS T* as_lds(T* p) { return (MFAS_LDS T*)p; }
template <typename T> __device__ __forceinline__ MFAS_GLB T* as_glb(T* p) { return (MFAS_GLB T*)p; }
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

#define KIND_S 0
#define KIND_V 1
#define KIND_OUT 2
#define KIND_HEAD 3

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define HIPCHK(x)                                                                                  \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess)                                                                      \
            return fail(MFAS_EHIP, std::string(#x) + ": " + hipGetErrorString(e_));                \
    } while (0)

// ------------------------------------------------------------------------------------------------
// Device-side descriptors
// ------------------------------------------------------------------------------------------------
struct SegDesc {          // one workgroup of k_sweep / k_pack
    int32_t cand, kind, cell, tap;
    int32_t k0, cc;       // first column inside the segment, chunk columns (multiple of 16)
    int32_t rows_p, width;  // padded rows; FEAT: table row width (elements)
    int64_t w_off;        // float offset (within a plane) of this chunk: tiles [rb][kb][256]
    int64_t wt_off;       // OUT/HEAD: float offset in the transposed arena, else -1
    int32_t part_idx;     // FEAT: chunk index within the cell's partial list
    int32_t rows, cols;   // true rows (R or C) / true columns of the whole segment
    int64_t src_off;      // flat-parameter offset of the matrix this segment belongs to
    int32_t src_ld, src_col0;
    uint32_t init_seed;   // hash seed of that matrix (device init)
    float init_bound;
    int32_t rb0, seg_nrb; // row-split units: first row block of this unit inside the segment, row blocks of the whole segment
                          // (rows_p = rows of THIS unit; w_off already points at row block rb0 of the chunk)
    int32_t nsub, _pad3;  // sweep work list only: this workgroup streams `nsub` consecutive column chunks of the segment (chunks
                          // k0, k0 + cc, ...; their tiles are consecutive in memory) and keeps the forward partial sums in registers
                          // across them: ONE partial slab per unit instead of one per chunk (0 / 1: a single chunk)
};

#define TAP_MAX_ITEMS 8
struct TapDesc {          // tap-major workgroup (R < 128): one feature chunk shared by up to 8 segments of that tap
    int32_t kind, tap, k0, cc;      // modality (KIND_S / KIND_V), tap index, first column, columns
    int32_t rows_p, width, nitems, _pad;
    int32_t cand[TAP_MAX_ITEMS], cell[TAP_MAX_ITEMS], part_idx[TAP_MAX_ITEMS];
    int64_t w_off[TAP_MAX_ITEMS];   // plane offset of each item's chunk: tiles [rb][kb][256]
};

struct CandDev {
    int32_t L;
    int32_t conf[MFAS_MAX_CELLS][3];
    int64_t seg_off[MFAS_MAX_CELLS][3];   // plane offset of S / V / OUT segment of cell i (-1: none)
    int32_t seg_cc[MFAS_MAX_CELLS][3];    // chunk columns of that segment
    int32_t seg_cols[MFAS_MAX_CELLS][3];  // padded columns
    int64_t head_off;
    int64_t outT_off[MFAS_MAX_CELLS];     // transposed arena offset of cell i's OUT segment
    int64_t headT_off;
    int64_t vec_off;                      // plane offset of the vector block
    int32_t nch_s[MFAS_MAX_CELLS], nch_v[MFAS_MAX_CELLS];
    int32_t part_cell_off[MFAS_MAX_CELLS];  // first partial-slot index of cell i
    int64_t step_off;                     // float offset of this candidate's step buffers
    uint32_t drop_seed;
    int32_t gidx;                         // index of this candidate in the population (stats / status slot)
    // flat (reference state_dict order) offsets of this candidate's parameters
    int64_t f_alpha, f_W[MFAS_MAX_CELLS], f_b[MFAS_MAX_CELLS], f_bn[MFAS_MAX_CELLS], f_Wc, f_bc;
    int32_t K_in[MFAS_MAX_CELLS];   // in_features of cell i
    int32_t _pad2[4];
};

struct DevStats {
    double train_loss, dev_loss;
    long long train_corr, dev_corr;
};

struct AdamC {
    float ss, bc2s, w1, b2, w2, eps, wd;
};
// field-by-field copy out of the kernel arguments with this step's step size / bias correction: a whole-struct copy is a 28-byte
// memcpy into a stack object, which the vectoriser then reads back as <4 x float> and the object stays in scratch
__device__ __forceinline__ AdamC adam_consts(const AdamC& k, const float ss, const float bc2s) {
    AdamC c;
    c.ss = ss; c.bc2s = bc2s; c.w1 = k.w1; c.b2 = k.b2; c.w2 = k.w2; c.eps = k.eps; c.wd = k.wd;
    return c;
}

struct Geo {             // geometry shared by all candidates of a population
    int32_t R, C, Rp, Cp, nrb, ncb, B, Bp, MB;
    int32_t bn, alphas, multitask, use_drop;
    float drop_scale, bn_eps, bn_mom;
    uint32_t drop_thr;
    // per-candidate step-buffer sub-offsets (floats)
    int64_t sb_part, sb_dy, sb_xo, sb_dlog, sb_sav, sb_yf, sb_gsc, sb_size;
    int32_t vec_cell_stride;   // 5*Rp + 16
    int32_t vec_head;          // offset of head bias inside the vector block
    int32_t sw[MFAS_MAX_TAPS], vw[MFAS_MAX_TAPS];   // table row strides of the taps (width padded to 16)
    int32_t loss_mode;         // 0 softmax CE + accuracy, 1 weighted BCE + F1-samples
    float f1_th;
    int64_t order_stride;      // elements between two candidates' sample-order tables (0: one order shared by the population)
};

// vector block of a candidate (inside every plane): per cell [b | gamma | beta | rm | rv | alpha(16)], then bc[Cp]
#define VEC_B 0
#define VEC_G 1
#define VEC_BE 2
#define VEC_RM 3
#define VEC_RV 4

// the sample-order table candidate `gidx` walks (models/searchable.py:248-250: the reference shuffles per candidate and epoch)
__device__ __forceinline__ const int32_t* cand_order(const int32_t* order, const Geo& g, int gidx) {
    return order ? order + (int64_t)gidx * g.order_stride : nullptr;
}

__device__ __forceinline__ uint32_t lowbias32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7FEB352DU;
    x ^= x >> 15;
    x *= 0x846CA68BU;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float act_fwd(float y, int nl) {
    if (nl == 0) return y <= 0.0f ? 0.0f : y;   // torch.relu: NaN propagates (fmaxf would swallow it)
    if (nl == 1) return 1.0f / (1.0f + expf(-y));
    return y > 0.0f ? y : 0.01f * y;
}
__device__ __forceinline__ float act_bwd(float a, float da, int nl) {
    if (nl == 0) return a <= 0.0f ? 0.0f : da;  // threshold_backward(grad, result, 0)
    if (nl == 1) return da * (1.0f - a) * a;
    return a > 0.0f ? da : 0.01f * da;   // leaky: sign(a) == sign(y)
}

// dW = x_t^T dy runs over the batch in blocks of 4 rows (one 16x16x4 MFMA each).  The batch is padded to MB * 16 rows; with the
// search default B = 20 (MB = 2) the blocks 5..7 hold only zeros and are skipped, B <= 48 (MB = 4) skips 12..15: ONE uniform branch
// per tile between two straight-line forms (a predicate per block cost registers: scratch in k_step_same / k_president).  Every dW
// path uses this helper, so all schedules keep summing the same products in the same order.
template <int MB> __device__ __forceinline__ constexpr int dw_short_blocks() { return MB == 2 ? 5 : (MB == 4 ? 12 : MB * 4); }
#define DW_BATCH_LOOP(MB, nj, BODY)                                                    \
    if (MB > 1 && (nj) <= dw_short_blocks<MB>()) {                                      \
        _Pragma("unroll") for (int j = 0; j < dw_short_blocks<MB>(); ++j) { BODY; }     \
    } else {                                                                            \
        _Pragma("unroll") for (int j = 0; j < MB * 4; ++j) { BODY; }                    \
    }

// ---- Adam(+L2) on weights held in registers (torch.optim.Adam, foreach=False form; oracle/np_oracle.py adam_step) --------
//     g += wd*w;  m += (1-b1)*(g-m);  v = v*b2 + ((1-b2)*g)*g;  denom = sqrt(v)/sqrt(bc2) + eps;  w -= (lr/bc1) * (m/denom)
// The square root and the two divisions are written out as correctly-rounding fma sequences — sqrt: v_rsq_f32 seed, one coupled
// Newton step, exact-residual correction (Markstein); division: v_rcp_f32 + Newton step, quotient + two fma residual corrections
// (the compiler's own f32 division, minus its v_div_scale / v_div_fmas / v_div_fixup range handling) — on the 4 elements a lane
// holds of a tile, so that nearly everything is a packed instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32): ~50 VALU
// instructions per 4 elements instead of the 200 of sqrtf() and operator/.  The results are the IEEE-754 correctly rounded ones
// — bit for bit what sqrtf() and operator/ give — whenever no intermediate leaves the normal range: v >= 2^-102 (below that
// the residual x - g*g is not exact; sqrt(v) < 4.5e-16 is then far inside the rounding of + eps) and |m| >= 2^-100 (below that
// the quotient's residual underflows; the step it would contribute is < 1e-22 * lr).  sqrt(0) = 0; 0/0 -> NaN as in the library
// form; m/0 (adam_eps = 0 with v = 0, not reachable from training) gives NaN where the library form gives +-inf, and so does an
// OVERFLOWED second moment (v = +inf, |g| > 1.8e19: rsq = 0, 0 * inf) where the library form's m / inf = 0 freezes the element —
// either way the candidate has diverged, here its weights say so.
// tools/adam_exact.hip compares both forms bit by bit on the GPU — sqrt over every non-negative float, the whole update on
// 7 x 64M sampled states (profiles/r03_adam_exact.log).
__device__ __forceinline__ f32x4 vfma4(f32x4 a, f32x4 b, f32x4 c) { return __builtin_elementwise_fma(a, b, c); }
// a - b on two register pairs (the compiler lowers a <4 x float> subtraction to four v_sub_f32; v_pk_add_f32 takes neg modifiers)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 pk_sub4(const f32x4 a, const f32x4 b) {
    f32x2 lo, hi;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(lo) : "v"(a.lo), "v"(b.lo));
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(hi) : "v"(a.hi), "v"(b.hi));
    return (f32x4){lo.x, lo.y, hi.x, hi.y};
}
__device__ __forceinline__ float rcp_refined(float d) {
    const float y = __builtin_amdgcn_rcpf(d);
    return __builtin_fmaf(__builtin_fmaf(-d, y, 1.0f), y, y);
}
__device__ __forceinline__ f32x4 rcp_refined4(f32x4 d) {
    f32x4 y;
#pragma unroll
    for (int q = 0; q < 4; ++q) y[q] = __builtin_amdgcn_rcpf(d[q]);
    return vfma4(vfma4(-d, y, (f32x4)(1.0f)), y, y);
}
// n / d given y = rcp_refined(d)
__device__ __forceinline__ float div_by(float n, float d, float y) {
    float q = n * y;
    float r = __builtin_fmaf(-d, q, n);
    q = __builtin_fmaf(r, y, q);
    r = __builtin_fmaf(-d, q, n);
    return __builtin_fmaf(r, y, q);
}
__device__ __forceinline__ f32x4 div_by4(f32x4 n, f32x4 d, f32x4 y) {
    f32x4 q = n * y;
    f32x4 r = vfma4(-d, q, n);
    q = vfma4(r, y, q);
    r = vfma4(-d, q, n);
    return vfma4(r, y, q);
}
// sqrt: v_rsq_f32 seed, one coupled Newton step on (g ~ sqrt x, h ~ 1/(2 sqrt x)), then the exact-residual correction
// g + (x - g*g) * h (Markstein's sequence).  tools/adam_exact.hip checks it against sqrtf for EVERY non-negative float.
// min(., 2^64) only acts on rsq(0) = inf (sqrt(0) = 0 without a branch); rsq of any normal x is <= 2^63.
__device__ __forceinline__ float sqrt_rn(float x) {
    const float y = fminf(__builtin_amdgcn_rsqf(x), 0x1p64f);
    float g = x * y, h = 0.5f * y;
    const float r = __builtin_fmaf(-h, g, 0.5f);
    g = __builtin_fmaf(g, r, g);
    h = __builtin_fmaf(h, r, h);
    return __builtin_fmaf(__builtin_fmaf(-g, g, x), h, g);
}
__device__ __forceinline__ f32x4 sqrt_rn4(f32x4 x) {
    f32x4 y;
#pragma unroll
    for (int q = 0; q < 4; ++q) y[q] = fminf(__builtin_amdgcn_rsqf(x[q]), 0x1p64f);
    f32x4 g = x * y, h = 0.5f * y;
    const f32x4 r = vfma4(-h, g, (f32x4)(0.5f));
    g = vfma4(g, r, g);
    h = vfma4(h, r, h);
    return vfma4(vfma4(-g, g, x), h, g);
}
#ifdef MFAS_ADAM_LIBRARY_FORMS      // the reference form of the same arithmetic (tools/adam_exact.hip builds both)
__device__ __forceinline__ void adam1(float& w, float& m, float& v, float g, const AdamC& c) {
    g = g + c.wd * w;
    m = m + c.w1 * (g - m);
    v = v * c.b2;
    v = v + (c.w2 * g) * g;
    const float denom = sqrtf(v) / c.bc2s + c.eps;
    w = w - c.ss * (m / denom);
}
__device__ __forceinline__ void adam4(f32x4& w, f32x4& m, f32x4& v, const f32x4 g, const AdamC& c) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float w1 = w[q], m1 = m[q], v1 = v[q];
        adam1(w1, m1, v1, g[q], c);
        w[q] = w1; m[q] = m1; v[q] = v1;
    }
}
// (the scalar-argument form the tile loops call: the -DMFAS_ADAM_LIBRARY_FORMS build of the whole library — __graft_entry__.build_variant
//  — trains with sqrtf() and operator/ in every kernel, for fidelity runs and for the in-situ bit-identity test)
__device__ __forceinline__ void adam4(f32x4& w, f32x4& m, f32x4& v, const f32x4 g, const float ss, const float bc2s, const float w1,
                                      const float b2, const float w2, const float eps, const float wd) {
    AdamC c;
    c.ss = ss; c.bc2s = bc2s; c.w1 = w1; c.b2 = b2; c.w2 = w2; c.eps = eps; c.wd = wd;
    adam4(w, m, v, g, c);
}
#else
__device__ __forceinline__ void adam1(float& w, float& m, float& v, float g, const AdamC& c) {
    g = g + c.wd * w;
    m = m + c.w1 * (g - m);
    v = v * c.b2;
    v = v + (c.w2 * g) * g;
    const float denom = div_by(sqrt_rn(v), c.bc2s, rcp_refined(c.bc2s)) + c.eps;
    w = w - c.ss * div_by(m, denom, rcp_refined(denom));
}
// the same on the 4 elements a lane holds of a 16x16 tile (element-wise identical to adam1)
__device__ __forceinline__ void adam4(f32x4& w, f32x4& m, f32x4& v, f32x4 g, const float ss, const float bc2s, const float w1,
                                      const float b2, const float w2, const float eps, const float wd) {
    g = g + wd * w;
    m = m + w1 * pk_sub4(g, m);
    v = v * b2;
    v = v + (w2 * g) * g;
    const f32x4 denom = div_by4(sqrt_rn4(v), (f32x4)(bc2s), (f32x4)(rcp_refined(bc2s))) + eps;
    w = pk_sub4(w, ss * div_by4(m, denom, rcp_refined4(denom)));
}
// (callers read the AdamC fields into scalars OUTSIDE their tile loops and pass those: with the struct read inside the loop the
// stack object survives into code generation — 24..32 bytes of scratch per lane)
__device__ __forceinline__ void adam4(f32x4& w, f32x4& m, f32x4& v, const f32x4 g, const AdamC& c) {
    adam4(w, m, v, g, c.ss, c.bc2s, c.w1, c.b2, c.w2, c.eps, c.wd);
}
#endif

// cross entropy of one row of C logits against `lab` (one lane, serial): the constant unimodal terms of the multitask
// loss, criteria[1](output[1], label) + criteria[2](output[2], label) (train_searchable/ntu.py:60-61)
__device__ __forceinline__ float row_ce(const float* x, int C, int lab) {
    float mx = x[0];
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, x[c]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(x[c] - mx);
    return -(x[lab] - mx - logf(se));
}

__device__ __forceinline__ uint32_t ld_u32_relaxed(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// every storing wave has drained its (write-through) stores when the barrier releases; then ONE lane signals
__device__ __forceinline__ void wg_publish_barrier() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}
// Per-(candidate, cell) "dy is out" flags of the same-group fused launch (k_step_same): [candidate][8] uint32, slot i < 4: dy_i of
// cell i (published after backward cell i, i.e. when the chain no longer reads W_out_{i+1}^T / Wc^T either); the value is the
// global step index + 1 (monotonic over a train() call)
#define CELLFLAG_STRIDE 8
#define CELLFLAG_SPIN_LIMIT (1u << 21)

// sum over the 4 lane groups that share (lane & 15): column reduction of an MFMA D block
// (gfx950 row swaps instead of two ds_bpermute round trips through the LDS crossbar: v_permlane16_swap exchanges the odd
// rows of one operand with the even rows of the other, v_permlane32_swap the wave halves; with both operands = x the two
// results are "x of the even / lower partner" and "x of the odd / upper partner", whose sum is the xor-16 / xor-32 butterfly.
// own + partner or partner + own: the same f32 sum)
__device__ __forceinline__ float colsum(float x) {
    int xi = __float_as_int(x);
    const auto r16 = __builtin_amdgcn_permlane16_swap(xi, xi, false, false);
    x = __int_as_float(r16[0]) + __int_as_float(r16[1]);
    xi = __float_as_int(x);
    const auto r32 = __builtin_amdgcn_permlane32_swap(xi, xi, false, false);
    return __int_as_float(r32[0]) + __int_as_float(r32[1]);
}

__device__ __forceinline__ int64_t tile_addr(int64_t seg_off, int rows_p, int cc, int rb, int kb) {
    const int nkb_c = cc >> 4;
    const int chunk = kb / nkb_c;
    const int kbi = kb - chunk * nkb_c;
    return seg_off + (int64_t)chunk * rows_p * cc + ((int64_t)rb * nkb_c + kbi) * 256;
}

// ------------------------------------------------------------------------------------------------
// Row staging: table rows (any dtype) -> f32 LDS tile [rows][stride]
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void stage_table(float* dst, int stride, const void* tab, int dtype, int width,
                                            int col0, int ncols, const int32_t* ord, int64_t pos,
                                            int base, int nvalid, int nrows, int tid, int nthreads) {
    if (dtype == MFAS_DT_F32) {
        const int vpr = ncols >> 2;
        for (int e = tid; e < nrows * vpr; e += nthreads) {
            const int b = e / vpr, c = (e - b * vpr) << 2;
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (b < nvalid) {
                const int64_t row = ord ? (int64_t)ord[pos + b] : (int64_t)(base + b);
                val = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(tab) + row * width + col0 + c);
            }
            *as_lds(reinterpret_cast<f32x4*>(dst + b * stride + c)) = val;
        }
    } else {
        const int vpr = ncols >> 3;
        for (int e = tid; e < nrows * vpr; e += nthreads) {
            const int b = e / vpr, c = (e - b * vpr) << 3;
            f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
            if (b < nvalid) {
                const int64_t row = ord ? (int64_t)ord[pos + b] : (int64_t)(base + b);
                const uint4 raw = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(tab) +
                                                                  row * width + col0 + c);
                const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
                float f[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (dtype == MFAS_DT_BF16) {
                        f[2 * j] = __uint_as_float(w[j] << 16);
                        f[2 * j + 1] = __uint_as_float(w[j] & 0xFFFF0000U);
                    } else {
                        f[2 * j] = __half2float(__ushort_as_half((unsigned short)(w[j] & 0xFFFFU)));
                        f[2 * j + 1] = __half2float(__ushort_as_half((unsigned short)(w[j] >> 16)));
                    }
                }
                lo = (f32x4){f[0], f[1], f[2], f[3]};
                hi = (f32x4){f[4], f[5], f[6], f[7]};
            }
            *as_lds(reinterpret_cast<f32x4*>(dst + b * stride + c)) = lo;
            *as_lds(reinterpret_cast<f32x4*>(dst + b * stride + c + 4)) = hi;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Coherent (COH) accesses for data that workgroups exchange INSIDE one launch (the persistent step loop, persist.hip.h).
// gfx950 has 8 XCDs with private, mutually non-coherent L2s and per-CU L1s that other CUs' stores never refresh, so
// exchanged data is stored write-through (`sc1`) and loaded with `sc1` (L1 bypass, coherent level) — no fences needed
// (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility"; cdna_hip_programming.md G16 R1).
// 16-byte accesses go through raw-buffer builtins (the compiler counts them in its s_waitcnt bookkeeping): `base` must be
// wave-uniform (a kernel-argument pointer) and the element index must stay below 2^30 floats (checked on the host).
// With COH = false the same helpers are ordinary loads / stores: the launch-per-phase schedule pays nothing.
// ------------------------------------------------------------------------------------------------
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define MFAS_AUX_SC1 16
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk_rsrc(const float* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, -1, 0x00020000);
}
template <bool COH>
__device__ __forceinline__ f32x4 ldc4(const float* base, int64_t idx) {
    if constexpr (COH) {
        const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(mk_rsrc(base), (int)(uint32_t)(idx << 2), 0, MFAS_AUX_SC1);
        return __builtin_bit_cast(f32x4, r);
    } else {
        return *reinterpret_cast<const f32x4*>(base + idx);
    }
}
template <bool COH>
__device__ __forceinline__ void stc4(float* base, int64_t idx, f32x4 v) {
    if constexpr (COH)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), mk_rsrc(base), (int)(uint32_t)(idx << 2), 0, MFAS_AUX_SC1);
    else
        *reinterpret_cast<f32x4*>(base + idx) = v;
}
template <bool COH>
__device__ __forceinline__ void stc1(float* p, float v) {
    if constexpr (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // global_store_dword sc1
    else *p = v;
}
template <bool COH>
__device__ __forceinline__ float ldc1(const float* p) {
    if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // global_load_dword sc1
    else return *p;
}

// f32 row-major global base[idx0 + b * src_stride + c] -> LDS [nrows][stride]
template <bool COH>
__device__ __forceinline__ void stage_f32(float* dst, int stride, const float* base, int64_t idx0, int src_stride, int ncols,
                                          int nrows, int tid, int nthreads) {
    const int vpr = ncols >> 2;
    for (int e = tid; e < nrows * vpr; e += nthreads) {
        const int b = e / vpr, c = (e - b * vpr) << 2;
        *reinterpret_cast<f32x4*>(dst + b * stride + c) = ldc4<COH>(base, idx0 + (int64_t)b * src_stride + c);
    }
}

// U = tiles of W/m/v in flight per wave (x3 planes).  A workgroup has 2 waves per SIMD, so two workgroups share a CU only
// inside a 128-VGPR budget (WPE = 4 waves per SIMD).  MB == 1 always runs that way with U = 2 (U = 4 spills, no gain).
// MB == 2 has two builds: WPE = 2 (up to 256 VGPRs, one workgroup per CU, U = 2: nothing spills, best when the co-scheduled
// chain's latency bounds the launch) and WPE = 4 (U = 1, the chain code spills a little, two workgroups per CU: +10..19 %
// when the sweep bounds the launch).  Deeper batches (U = 4, 6 at WPE = 2) measured 8-12 % slower.
// (MB == 1 with U = 4 in a one-workgroup-per-CU build for the sweep-only launches of small populations was measured SLOWER:
// 90.6 vs 82.6 us/step at 6 candidates, R=128 — profiles/r02_popsweep_r128.log.)
template <int MB, int WPE> struct SweepU { static constexpr int v = (MB == 2 && WPE == 4) ? 1 : 2; };

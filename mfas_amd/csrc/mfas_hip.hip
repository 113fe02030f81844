// mfas_hip.hip — MI355X (gfx950 / CDNA4) inner candidate-training engine for MFAS.
//
// What it replaces (reference = jperezrua/mfas, pure PyTorch):
//   train_sampled_models                models/search/ntu_searchable.py:23-102
//   train_ntu_track_acc                 models/search/train_searchable/ntu.py:14-89
//   Searchable_Skeleton_Image_Net.forward (+ autograd + torch.optim.Adam)   ntu_searchable.py:206-286
//
// Design: the whole population trains in lockstep.  Per train step three kernels run for
// ALL candidates at once:
//   k_chain  (1 workgroup / candidate): the serial R-wide part — reduce feature partial sums, cell chain
//            (prev-out GEMM on f32 MFMA, activation, BN batch stats, dropout), head, CE loss, and the
//            backward chain producing dy_i for every cell;
//   k_sweep  (1 workgroup / (candidate, weight chunk)): the HBM-bound part — for every weight tile:
//            dW = x_t^T dy (f32 MFMA) -> Adam(+L2) update of W/m/v in registers -> store -> immediately
//            use the new W for the NEXT step's forward partial sums (f32 MFMA).  24 B/param/step = the
//            algorithmic minimum with state in HBM.
// Dev evaluation is row-parallel (k_eval).  Weights live in a 16x16 tile-major layout that is exactly the
// MFMA 16x16x4 f32 operand layout, so every W/m/v access is one coalesced 16 B/lane load.
//
// MFMA used: v_mfma_f32_16x16x4_f32 (exact f32 fma chain).  Layout (lane l):
//   A[i = l&15][k = l>>4],  B[k = l>>4][j = l&15],  D[i = 4*(l>>4)+reg][j = l&15].
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <algorithm>
#include <string>
#include <vector>

#include "mfas_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
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
};

// vector block of a candidate (inside every plane): per cell [b | gamma | beta | rm | rv | alpha(16)], then bc[Cp]
#define VEC_B 0
#define VEC_G 1
#define VEC_BE 2
#define VEC_RM 3
#define VEC_RV 4

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

__device__ __forceinline__ void adam1(float& w, float& m, float& v, float g, const AdamC& c) {
    g = g + c.wd * w;
    m = m + c.w1 * (g - m);
    v = v * c.b2;
    v = v + (c.w2 * g) * g;
    const float denom = sqrtf(v) / c.bc2s + c.eps;
    w = w - c.ss * (m / denom);
}

// sum over the 4 lane groups that share (lane & 15): column reduction of an MFMA D block
__device__ __forceinline__ float colsum(float x) {
    x += __shfl_xor(x, 16);
    x += __shfl_xor(x, 32);
    return x;
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
            *reinterpret_cast<f32x4*>(dst + b * stride + c) = val;
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
            *reinterpret_cast<f32x4*>(dst + b * stride + c) = lo;
            *reinterpret_cast<f32x4*>(dst + b * stride + c + 4) = hi;
        }
    }
}

// f32 row-major global [nrows][src_stride] -> LDS [nrows][stride]
__device__ __forceinline__ void stage_f32(float* dst, int stride, const float* src, int src_stride, int ncols,
                                          int nrows, int tid, int nthreads) {
    const int vpr = ncols >> 2;
    for (int e = tid; e < nrows * vpr; e += nthreads) {
        const int b = e / vpr, c = (e - b * vpr) << 2;
        *reinterpret_cast<f32x4*>(dst + b * stride + c) =
            *reinterpret_cast<const f32x4*>(src + (int64_t)b * src_stride + c);
    }
}

// U = tiles of W/m/v in flight per wave (x3 planes).  A workgroup has 2 waves per SIMD, so two workgroups share a CU only
// inside a 128-VGPR budget (WPE = 4 waves per SIMD).  MB == 1 always runs that way with U = 2 (U = 4 spills, no gain).
// MB == 2 has two builds: WPE = 2 (up to 256 VGPRs, one workgroup per CU, U = 2: nothing spills, best when the co-scheduled
// chain's latency bounds the launch) and WPE = 4 (U = 1, the chain code spills a little, two workgroups per CU: +10..19 %
// when the sweep bounds the launch).  Deeper batches (U = 4, 6 at WPE = 2) measured 8-12 % slower.
template <int MB, int WPE> struct SweepU { static constexpr int v = (MB == 2 && WPE == 4) ? 1 : 2; };

// ------------------------------------------------------------------------------------------------
// tile_run — the fused per-tile work shared by both sweep decompositions, for ONE row block `rb` over the k-blocks
// kb0, kb0+kbs, ... < nkb of a chunk: request SWEEP_U tiles of W/m/v, then per tile
//   dW^T = x_t^T dy (4*MB f32 MFMAs; D image == the tile image)  ->  Adam(+L2) on 4 elements/lane in registers  ->
//   store W, m, v (+ transposed copy T for OUT/HEAD)  ->  y_{t+1} += x_{t+1} W_new^T (4*MB f32 MFMAs) into yacc.
// ------------------------------------------------------------------------------------------------
template <int MB, bool NT, int SWEEP_U>
__device__ __forceinline__ void tile_run(float* Wp, float* Mp, float* Vp, const int rb, const int nkb, const int kb0,
                                         const int kbs, const float* xt, const int ST, const float* xn, const int SN,
                                         const float (&dyf)[MB * 4], const float gsc, const AdamC& ac, const bool upd,
                                         const bool fwd, f32x4 (&yacc)[MB], float* T, const int tstride_rb,
                                         const int lane) {
    const int l15 = lane & 15, lg = lane >> 4;
    for (int kbb = kb0; kbb < nkb; kbb += SWEEP_U * kbs) {
        f32x4 w4[SWEEP_U], m4[SWEEP_U], v4[SWEEP_U];
#pragma unroll
        for (int u = 0; u < SWEEP_U; ++u) {
            const int kb = kbb + u * kbs;
            if (kb < nkb) {
                const int64_t off = ((int64_t)rb * nkb + kb) * 256 + lane * 4;
                // state larger than the Infinity Cache is streamed once per step: nontemporal (+5 % HBM rate)
                w4[u] = NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(Wp + off))
                           : *reinterpret_cast<const f32x4*>(Wp + off);
                if (upd) {
                    m4[u] = NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(Mp + off))
                               : *reinterpret_cast<const f32x4*>(Mp + off);
                    v4[u] = NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(Vp + off))
                               : *reinterpret_cast<const f32x4*>(Vp + off);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < SWEEP_U; ++u) {
            const int kb = kbb + u * kbs;
            if (kb < nkb) {
                const int64_t off = ((int64_t)rb * nkb + kb) * 256 + lane * 4;
                if (upd) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < MB * 4; ++j)
                        acc = MFMA16(xt[(4 * j + lg) * ST + kb * 16 + l15], dyf[j], acc);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float w = w4[u][q], m = m4[u][q], v = v4[u][q];
                        adam1(w, m, v, acc[q] * gsc, ac);
                        w4[u][q] = w;
                        m4[u][q] = m;
                        v4[u][q] = v;
                    }
                    if (NT) {
                        __builtin_nontemporal_store(w4[u], reinterpret_cast<f32x4*>(Wp + off));
                        __builtin_nontemporal_store(m4[u], reinterpret_cast<f32x4*>(Mp + off));
                        __builtin_nontemporal_store(v4[u], reinterpret_cast<f32x4*>(Vp + off));
                    } else {
                        *reinterpret_cast<f32x4*>(Wp + off) = w4[u];
                        *reinterpret_cast<f32x4*>(Mp + off) = m4[u];
                        *reinterpret_cast<f32x4*>(Vp + off) = v4[u];
                    }
                    if (T) {   // keep the transposed copy used by the backward chain in step (OUT / HEAD only)
                        float* Tt = T + ((int64_t)kb * tstride_rb + rb) * 256;
                        const int base = (((l15 >> 2) * 16 + 4 * lg) << 2) + (l15 & 3);
#pragma unroll
                        for (int q = 0; q < 4; ++q) Tt[base + 4 * q] = w4[u][q];
                    }
                }
                if (fwd) {
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) {
                        const f32x4 x4 = *reinterpret_cast<const f32x4*>(xn + (mb * 16 + l15) * SN + kb * 16 + 4 * lg);
#pragma unroll
                        for (int q = 0; q < 4; ++q) yacc[mb] = MFMA16(x4[q], w4[u][q], yacc[mb]);
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_sweep — fused dW + Adam + next-step forward.  One workgroup = ALL row blocks of one weight segment
// over a chunk of `cc` columns: x_t / x_{t+1} / dy are staged ONCE in LDS, then every wave streams whole
// row blocks (contiguous 1 KiB tiles), requesting the W/m/v tiles of 4 k-blocks before consuming them.
// ------------------------------------------------------------------------------------------------
struct SweepArgs {
    const SegDesc* desc;
    const TapDesc* tdesc;   // tap-major work list (may be empty)
    int32_t ntap, _padt;
    const CandDev* cands;
    float* plane;
    int64_t plane_stride;
    float* wt;
    float* stepbuf;
    mfas_table tab;
    const int32_t* order;
    int64_t pos_t, pos_n;
    int32_t base_t, base_n, nvalid_t, nvalid_n;
    int32_t do_update, do_forward;
    AdamC ac;
    Geo g;
};

#define STEP_NW 8
#define STEP_THREADS (STEP_NW * 64)

template <int MB, bool NT, int U>
__device__ __forceinline__ void sweep_body(const SweepArgs& a, const int bid, float* lds) {
    const SegDesc d = a.desc[bid];
    const CandDev& cd = a.cands[d.cand];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    constexpr int Bp = MB * 16;
    const int cc = d.cc, rows_p = d.rows_p, nrb = rows_p >> 4, nkb = cc >> 4;
    const int ST = cc + 16;   // x_t stride: conflict-free ds_read_b32 column reads
    const int SN = cc + 4;    // x_{t+1} stride: 16 B aligned rows for ds_read_b128
    const int SD = rows_p + 16;
    float* xt = lds;
    float* xn = xt + Bp * ST;
    float* dyl = xn + Bp * SN;
    float* wred = dyl + Bp * SD;   // [8 waves][nrb][MB][256], only when the chunk is k-split over waves
    const bool feat = d.kind <= KIND_V;
    const bool upd = a.do_update != 0;
    const bool fwd = (a.do_forward != 0) && feat;
    if (!upd && !fwd) return;
    float* sb = a.stepbuf + cd.step_off;

    if (upd) {
        if (feat) {
            const void* tp = d.kind == KIND_S ? a.tab.s[d.tap] : a.tab.v[d.tap];
            stage_table(xt, ST, tp, a.tab.dtype, d.width, d.k0, cc, a.order, a.pos_t, a.base_t, a.nvalid_t, Bp, tid, STEP_THREADS);
        } else {
            const int xcell = d.kind == KIND_OUT ? d.cell - 1 : cd.L - 1;
            stage_f32(xt, ST, sb + a.g.sb_xo + (int64_t)xcell * Bp * a.g.Rp + d.k0, a.g.Rp, cc, Bp, tid, STEP_THREADS);
        }
        const float* dsrc = d.kind == KIND_HEAD ? sb + a.g.sb_dlog : sb + a.g.sb_dy + (int64_t)d.cell * Bp * a.g.Rp;
        stage_f32(dyl, SD, dsrc, rows_p, rows_p, Bp, tid, STEP_THREADS);
    }
    if (fwd) {
        const void* tp = d.kind == KIND_S ? a.tab.s[d.tap] : a.tab.v[d.tap];
        stage_table(xn, SN, tp, a.tab.dtype, d.width, d.k0, cc, a.order, a.pos_n, a.base_n, a.nvalid_n, Bp, tid, STEP_THREADS);
    }
    __syncthreads();

    float* Wp = a.plane + d.w_off;
    float* Mp = Wp + a.plane_stride;
    float* Vp = Mp + a.plane_stride;
    const AdamC ac = a.ac;
    // alpha scaling of the gradient of S / V columns (aux_models.py:103-111): sigma(alpha_t) as used by this
    // step's forward, published by k_chain (alpha itself has already been stepped); 1.0 when alphas are off
    float gsc = 1.0f;
    if (a.g.alphas && feat && upd) gsc = sb[a.g.sb_gsc + d.cell * 2 + d.kind];

    // Work split: with >= 8 row blocks every wave owns whole row blocks (streams contiguous tiles, no
    // reduction); with fewer (R < 128) the waves split the k blocks and reduce through LDS.
    const bool split_k = nrb < STEP_NW;
    const int rb0 = split_k ? 0 : wave, rbs = split_k ? 1 : STEP_NW;
    const int kb0 = split_k ? wave : 0, kbs = split_k ? STEP_NW : 1;
    float* part = sb + a.g.sb_part + (((int64_t)(cd.part_cell_off[d.cell] + d.part_idx) * nrb * MB) << 8);

    for (int rb = rb0; rb < nrb; rb += rbs) {
        float dyf[MB * 4];
#pragma unroll
        for (int j = 0; j < MB * 4; ++j) dyf[j] = upd ? dyl[(4 * j + lg) * SD + rb * 16 + l15] : 0.f;
        f32x4 yacc[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) yacc[mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        tile_run<MB, NT, U>(Wp, Mp, Vp, rb, nkb, kb0, kbs, xt, ST, xn, SN, dyf, gsc, ac, upd, fwd, yacc,
                         d.wt_off >= 0 ? a.wt + d.wt_off + (int64_t)(d.k0 >> 4) * nrb * 256 : nullptr, nrb, lane);
        if (fwd) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                if (split_k)
                    *reinterpret_cast<f32x4*>(wred + (((wave * nrb + rb) * MB + mb) << 8) + lane * 4) = yacc[mb];
                else   // partial slot in MFMA D layout [chunk][rb][mb][lane][4]
                    *reinterpret_cast<f32x4*>(part + ((rb * MB + mb) << 8) + lane * 4) = yacc[mb];
            }
        }
    }
    if (!fwd || !split_k) return;
    __syncthreads();
    // deterministic cross-wave reduction (fixed order 0..7)
    for (int e = tid; e < nrb * MB * 64; e += STEP_THREADS) {
        const int slot = e >> 6, ln = e & 63;
        f32x4 s = *reinterpret_cast<const f32x4*>(wred + (slot << 8) + ln * 4);
#pragma unroll
        for (int w = 1; w < STEP_NW; ++w)
            s += *reinterpret_cast<const f32x4*>(wred + ((w * nrb * MB + slot) << 8) + ln * 4);
        *reinterpret_cast<f32x4*>(part + (slot << 8) + ln * 4) = s;
    }
}

// ------------------------------------------------------------------------------------------------
// sweep_tap_body — the same fused dW + Adam + next-step forward for SMALL R (1, 2 or 4 row blocks): a column chunk of
// ONE feature tap is staged once and shared by up to 8/nrb segments (candidates x cells) that read this tap; every wave
// owns one (segment, row block), streams its contiguous tiles and writes its forward partial directly — no cross-wave
// reduction, and the feature staging (1/3 of the traffic at R=16) is amortised over the segments.
// ------------------------------------------------------------------------------------------------
template <int MB, bool NT, int U>
__device__ __forceinline__ void sweep_tap_body(const SweepArgs& a, const int bid, float* lds) {
    const TapDesc& d = a.tdesc[bid];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    constexpr int Bp = MB * 16;
    const int cc = d.cc, rows_p = d.rows_p, nrb = rows_p >> 4, nkb = cc >> 4;
    const int ST = cc + 16, SN = cc + 4;
    float* xt = lds;
    float* xn = xt + Bp * ST;
    const bool upd = a.do_update != 0;
    const bool fwd = a.do_forward != 0;
    if (!upd && !fwd) return;
    const void* tp = d.kind == KIND_S ? a.tab.s[d.tap] : a.tab.v[d.tap];
    if (upd) stage_table(xt, ST, tp, a.tab.dtype, d.width, d.k0, cc, a.order, a.pos_t, a.base_t, a.nvalid_t, Bp, tid, STEP_THREADS);
    if (fwd) stage_table(xn, SN, tp, a.tab.dtype, d.width, d.k0, cc, a.order, a.pos_n, a.base_n, a.nvalid_n, Bp, tid, STEP_THREADS);
    __syncthreads();
    const int item = wave / nrb, rb = wave - item * nrb;
    if (item >= d.nitems) return;
    const CandDev& cd = a.cands[d.cand[item]];
    float* sb = a.stepbuf + cd.step_off;
    const int cell = d.cell[item];
    float dyf[MB * 4];
#pragma unroll
    for (int j = 0; j < MB * 4; ++j) dyf[j] = 0.f;
    if (upd) {
        const float* dsrc = sb + a.g.sb_dy + (int64_t)cell * Bp * a.g.Rp;
#pragma unroll
        for (int j = 0; j < MB * 4; ++j) dyf[j] = dsrc[(4 * j + lg) * rows_p + rb * 16 + l15];
    }
    float gsc = 1.0f;
    if (a.g.alphas && upd) gsc = sb[a.g.sb_gsc + cell * 2 + d.kind];
    float* Wp = a.plane + d.w_off[item];
    float* Mp = Wp + a.plane_stride;
    float* Vp = Mp + a.plane_stride;
    const AdamC ac = a.ac;
    f32x4 yacc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) yacc[mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    tile_run<MB, NT, U>(Wp, Mp, Vp, rb, nkb, 0, 1, xt, ST, xn, SN, dyf, gsc, ac, upd, fwd, yacc, nullptr, nrb, lane);
    if (fwd) {
        float* part = sb + a.g.sb_part + (((int64_t)(cd.part_cell_off[cell] + d.part_idx[item]) * nrb * MB) << 8);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
            *reinterpret_cast<f32x4*>(part + ((rb * MB + mb) << 8) + lane * 4) = yacc[mb];
    }
}

// ------------------------------------------------------------------------------------------------
// k_chain — one 8-wave workgroup per candidate: forward chain, CE loss, backward chain (train step).
// Latency-bound by construction (serial in the cells), so: every wave owns one 16-column block, weight
// tiles of a product are requested in one batch before the MFMAs, saved activations live in LDS.
// ------------------------------------------------------------------------------------------------
struct ChainArgs {
    const CandDev* cands;
    float* plane;
    int64_t plane_stride;
    const float* wt;
    float* stepbuf;
    mfas_table tab;
    const int32_t* order;
    int64_t pos_t;
    int32_t base_t, nvalid;
    int32_t gstep, epoch, E;
    int32_t yf_in_lds, vec_in_lds;
    AdamC ac;
    Geo g;
    DevStats* stats;
    int32_t* status;
    const float* pos_w;   // loss_mode 1: per-class positive weights
};

#define CHAIN_NW STEP_NW
#define CHAIN_THREADS STEP_THREADS

__device__ __forceinline__ bool drop_keep(uint32_t h0, int cell, uint32_t idx, uint32_t thr) {
    // oracle/np_oracle.py:dropout_keep
    const uint32_t key = idx + (uint32_t)cell * 0x7F4A7C15U;
    return (lowbias32(key ^ h0) >> 8) >= thr;
}

// acc[mb] += X[b][0..16*nk) . tile(k)   (X in LDS row-major with stride sx; tiles: 256 floats each, stride tstride)
template <int MB>
__device__ __forceinline__ void lds_x_times_tiles(f32x4 (&acc)[MB], const float* X, int sx, const float* tiles,
                                                  int64_t tstride, int nk, int lane) {
    // same arithmetic as mma_tiles: even / odd k-blocks in two independent chains, summed at the end
    const int l15 = lane & 15, lg = lane >> 4;
    f32x4 acc2[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc2[mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < nk; k0 += 8) {
        f32x4 w8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (k0 + u < nk) w8[u] = *reinterpret_cast<const f32x4*>(tiles + (int64_t)(k0 + u) * tstride + lane * 4);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (k0 + u < nk) {
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    const f32x4 x4 = *reinterpret_cast<const f32x4*>(X + (mb * 16 + l15) * sx + (k0 + u) * 16 + 4 * lg);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (u & 1) acc2[mb] = MFMA16(x4[q], w8[u][q], acc2[mb]);
                        else acc[mb] = MFMA16(x4[q], w8[u][q], acc[mb]);
                    }
                }
            }
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[mb] += acc2[mb];
}

// Workgroup barrier for data exchanged through LDS only: waits for this wave's LDS traffic (lgkmcnt) but NOT for its
// outstanding global stores, which __syncthreads() would (s_waitcnt vmcnt(0) = a full store round trip per cell).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Software pipelining of the chain: the weight tiles of a product do not depend on the activations, so a wave
// requests the NEXT product's tiles (<= 8 tiles = 32 VGPRs) before it starts the current one.
__device__ __forceinline__ void issue_tiles(f32x4 (&w8)[8], const float* tiles, int nk, int lane) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
        if (u < nk) w8[u] = *reinterpret_cast<const f32x4*>(tiles + (int64_t)u * 256 + lane * 4);
}

// acc[mb] += X[b][0..16*nk) . w8[k]; even / odd k-blocks accumulate in two independent MFMA chains
template <int MB>
__device__ __forceinline__ void mma_tiles(f32x4 (&acc)[MB], const float* X, int sx, const f32x4 (&w8)[8], int nk, int lane) {
    const int l15 = lane & 15, lg = lane >> 4;
    f32x4 acc2[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc2[mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 8; ++u)
        if (u < nk) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const f32x4 x4 = *reinterpret_cast<const f32x4*>(X + (mb * 16 + l15) * sx + u * 16 + 4 * lg);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (u & 1) acc2[mb] = MFMA16(x4[q], w8[u][q], acc2[mb]);
                    else acc[mb] = MFMA16(x4[q], w8[u][q], acc[mb]);
                }
            }
        }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[mb] += acc2[mb];
}

// WeightedCrossEntropyWithLogits (models/central/mm_imdb.py:655-673) on the LDS logits, 4 lanes per row:
// L = mean_{b,c}[ w_c z (-log s) + (1 - z)(-log(1 - s)) ], s = sigmoid(x);  dlogit = (-w_c z (1 - s) + (1 - z) s) / (B*C).
// red[b] receives the row's share of the BATCH-MEAN loss times nvalid (so that sum_b red[b] = loss * batch size,
// train_searchable/mmimdb.py:96), red[Bp + b] = 0.
__device__ __forceinline__ void bce_rows(float* lg_l, int SC, float* red, int Bp, const int* rowidx,
                                         const float* multilabel, const float* pos_w, int C, int Cp, int nvalid, int tid) {
    const int b = tid >> 2, sub = tid & 3;
    float* row = lg_l + b * SC;
    const bool ok = b < nvalid;
    const float* z = ok ? multilabel + (int64_t)rowidx[b] * C : nullptr;
    float ls = 0.f;
    const float inv = 1.0f / ((float)nvalid * (float)C);
    for (int c = sub; c < Cp; c += 4) {
        float dl = 0.f;
        if (ok && c < C) {
            const float sg = 1.0f / (1.0f + expf(-row[c]));
            const float zz = z[c], w = pos_w[c];
            ls += w * zz * -logf(sg) + (1.0f - zz) * -logf(1.0f - sg);
            dl = (-w * zz * (1.0f - sg) + (1.0f - zz) * sg) * inv;
        }
        row[c] = dl;
    }
    ls += __shfl_xor(ls, 1);
    ls += __shfl_xor(ls, 2);
    if (sub == 0) {
        red[b] = ls / (float)C;      // sum_b red[b] / nvalid = batch-mean loss
        red[Bp + b] = 0.f;
    }
}

// Softmax cross-entropy on the LDS logits (train_searchable/ntu.py:53-61), LPR lanes per batch row: classes c = sub,
// sub+LPR, ... (<= 8 classes per lane, exp kept).  Leaves dlogits = (softmax - onehot)/nvalid in place, the row's loss in
// red[b] and its top-1 hit in red[Bp + b] (multitask: argmax of central + visual + skeleton logits).
template <int MB, int NC>
__device__ __forceinline__ void softmax_rows_nc(const ChainArgs& a, float* lg_l, const int SC, float* red_l, const int* lab_l,
                                                const int nvalid, const float nf, const int tid) {
    constexpr int Bp = MB * 16;
    constexpr int LPR = (STEP_THREADS / Bp) < 16 ? (STEP_THREADS / Bp) : 16;
    const Geo& g = a.g;
    const int C = g.C, Cp = g.Cp;
    const int b = tid / LPR, sub = tid % LPR;
    float* row = lg_l + b * SC;
    const bool ok = b < nvalid;
    const int lab = lab_l[b];
    // NC classes per lane (host guarantees Cp <= 8 * LPR; the caller picks NC = 4 when Cp <= 4 * LPR: the skipped
    // iterations only ever added 0 / compared against -3e38, so the result is bit-identical)
    float xv[NC], ev[NC];
    float mx = -3.0e38f;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int c = sub + j * LPR;
        xv[j] = c < C ? row[c] : -3.0e38f;
        mx = fmaxf(mx, xv[j]);
    }
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float se = 0.f;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int c = sub + j * LPR;
        ev[j] = c < C ? expf(xv[j] - mx) : 0.f;
        se += ev[j];
    }
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) se += __shfl_xor(se, o);
    // argmax, first max on ties (torch.max(dim=1)); multitask: central + visual + skeleton logits
    float bv = -3.0e38f;
    int bi = 0x7FFFFFFF;
    const float* vl = nullptr;
    const float* sl = nullptr;
    if (g.multitask && ok) {
        const int64_t grow = a.order ? (int64_t)a.order[a.pos_t + b] : (int64_t)(a.base_t + b);
        vl = a.tab.vlogit + grow * C;
        sl = a.tab.slogit + grow * C;
    }
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int c = sub + j * LPR;
        if (c < C) {
            float t = xv[j];
            if (vl) t = (t + vl[c]) + sl[c];
            if (t > bv) { bv = t; bi = c; }
        }
    }
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) {
        const float pv = __shfl_xor(bv, o);
        const int pi = __shfl_xor(bi, o);
        if (pv > bv || (pv == bv && pi < bi)) { bv = pv; bi = pi; }
    }
    const float lse = mx + logf(se);
    if (sub == 0) {
        red_l[b] = ok ? -(row[lab] - lse) : 0.f;
        red_l[Bp + b] = (ok && bi == lab) ? 1.f : 0.f;
    }
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int c = sub + j * LPR;
        if (c < Cp) {
            float dl = 0.f;
            if (ok && c < C) {
                dl = ev[j] / se;
                if (c == lab) dl -= 1.0f;
                dl = dl / nf;
            }
            row[c] = dl;
        }
    }

}

template <int MB>
__device__ __forceinline__ void softmax_rows(const ChainArgs& a, float* lg_l, const int SC, float* red_l, const int* lab_l,
                                             const int nvalid, const float nf, const int tid) {
    constexpr int Bp = MB * 16;
    constexpr int LPR = (STEP_THREADS / Bp) < 16 ? (STEP_THREADS / Bp) : 16;
    if (a.g.Cp <= 4 * LPR) softmax_rows_nc<MB, 4>(a, lg_l, SC, red_l, lab_l, nvalid, nf, tid);
    else softmax_rows_nc<MB, 8>(a, lg_l, SC, red_l, lab_l, nvalid, nf, tid);
}

#ifdef MFAS_CHAIN_TIMING
#define CT_STAMP(slot) do { if (threadIdx.x == 0 && bid == 0 && a.gstep == 3) a.status[64 + (slot)] = (int32_t)(__builtin_readcyclecounter() - ct0); } while (0)
#else
#define CT_STAMP(slot) do { } while (0)
#endif

template <int MB, bool PF>
__device__ __forceinline__ void chain_body(const ChainArgs& a, const int bid, float* lds) {
#ifdef MFAS_CHAIN_TIMING
    const unsigned long long ct0 = __builtin_readcyclecounter();
#endif
    const CandDev& cd = a.cands[bid];
    const Geo& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    constexpr int Bp = MB * 16;
    constexpr int LPR = (STEP_THREADS / Bp) < 16 ? (STEP_THREADS / Bp) : 16;   // softmax lanes per batch row
    const int Rp = g.Rp, nrb = g.nrb, Cp = g.Cp, ncb = g.ncb, R = g.R, C = g.C, L = cd.L;
    const int SX = Rp + 4, SC = Cp + 4;
    // LDS kept close to the sweep's so both bodies can share one launch: ping-pong activation buffers (out_i
    // going forward, reused for dy_i coming back), logits, reduced feature sums; saved activations go to L2 scratch.
    float* xo_l = lds;                       // [2][Bp][SX]  ping-pong out_i (A operand of the next cell)
    float* dy_l = xo_l;                      // backward reuses the same two buffers for dy_i
    float* lg_l = xo_l + 2 * Bp * SX;        // [Bp][SC]  logits -> dlogits
    float* rstd_l = lg_l + Bp * SC;          // [L][Rp]
    float* red_l = rstd_l + MFAS_MAX_CELLS * Rp;   // [2*Bp] loss / correct per row (+ alpha partials)
    int* lab_l = reinterpret_cast<int*>(red_l + 2 * Bp + 16);   // [Bp]
    const int64_t sav_plane = (int64_t)MFAS_MAX_CELLS * nrb * MB * 256;

    float* W = a.plane;
    float* Mv = a.plane + a.plane_stride;
    float* Vv = Mv + a.plane_stride;
    float* sb = a.stepbuf + cd.step_off;
    float* sav = sb + g.sb_sav;              // [3][L][nrb][MB][256]: act, xhat, (yS - yV)
    // per-candidate scalars the serial loops need, read ONCE: the LDS barriers are compiler memory barriers, and a field
    // of `cd` used after one is a fresh scalar load (a few hundred cycles on the critical path of every cell)
    const int64_t cvec_off = cd.vec_off;
    int nlbits = 0;
#pragma unroll
    for (int i = 0; i < MFAS_MAX_CELLS; ++i) nlbits |= (cd.conf[i][2] & 3) << (2 * i);
    const int cgidx = cd.gidx;
    // reduced feature sums [1 or 2][L][nrb][MB][256]: LDS when it fits the shared budget, else scratch
    float* yf_l = a.yf_in_lds ? reinterpret_cast<float*>(lab_l + Bp) : sb + g.sb_yf;
    // vector parameters (+ their Adam state): the standalone chain stages the candidate's whole vector block into LDS
    // once, so that no dependent global load sits inside the serial cell loops; updates are written to global only
    const int nvec = MFAS_MAX_CELLS * g.vec_cell_stride + Cp;
    const float* vecW = W + cvec_off;
    const float* vecM = Mv + cvec_off;
    const float* vecV = Vv + cvec_off;
    if (PF && a.vec_in_lds) {
        float* vl = reinterpret_cast<float*>(lab_l + Bp) + (a.yf_in_lds ? (g.alphas ? 2 : 1) * sav_plane : 0);
        for (int e = tid; e < nvec; e += CHAIN_THREADS) {
            vl[e] = vecW[e];
            vl[nvec + e] = vecM[e];
            vl[2 * nvec + e] = vecV[e];
        }
        vecW = vl; vecM = vl + nvec; vecV = vl + 2 * nvec;   // visible after the phase-0 barrier below
    }
    const int nvalid = a.nvalid;
    const float nf = (float)nvalid;
    const AdamC ac = a.ac;
    const uint32_t h0 = lowbias32(cd.drop_seed + 0x9E3779B9U * (uint32_t)(a.gstep + 1));

    if (tid < Bp) {
        int lab = 0;
        if (tid < nvalid) {
            const int64_t row = a.order ? (int64_t)a.order[a.pos_t + tid] : (int64_t)(a.base_t + tid);
            lab = g.loss_mode == 0 ? a.tab.label[row] : (int)row;   // mode 1 keeps the table row for the multi-hot targets
        }
        lab_l[tid] = lab;
    }

    // ------------------------------------------------------------------ phase 0: all 512 threads reduce the
    // sweep's column-chunk partial sums of EVERY cell (fixed order) into LDS, loads batched 8 deep
    {
        const int per_cell = nrb * MB * 64;   // float4 items per cell
        for (int e = tid; e < L * per_cell; e += CHAIN_THREADS) {
            const int i = e / per_cell, it = e - i * per_cell;
            const int ns = cd.nch_s[i], nch = ns + cd.nch_v[i];
            const float* part = sb + g.sb_part + (((int64_t)cd.part_cell_off[i] * nrb * MB) << 8) + it * 4;
            f32x4 accS = {0.f, 0.f, 0.f, 0.f}, accV = {0.f, 0.f, 0.f, 0.f};
            constexpr int PB = PF ? 16 : 8;   // partial-sum loads in flight per thread
            for (int ch0 = 0; ch0 < nch; ch0 += PB) {
                f32x4 p8[PB];
#pragma unroll
                for (int u = 0; u < PB; ++u)
                    if (ch0 + u < nch) p8[u] = *reinterpret_cast<const f32x4*>(part + (((int64_t)(ch0 + u) * nrb * MB) << 8));
#pragma unroll
                for (int u = 0; u < PB; ++u)
                    if (ch0 + u < nch) {
                        if (ch0 + u < ns) accS += p8[u]; else accV += p8[u];
                    }
            }
            if (g.alphas) {
                *reinterpret_cast<f32x4*>(yf_l + (int64_t)i * per_cell * 4 + it * 4) = accS;
                *reinterpret_cast<f32x4*>(yf_l + sav_plane + (int64_t)i * per_cell * 4 + it * 4) = accV;
            } else {
                *reinterpret_cast<f32x4*>(yf_l + (int64_t)i * per_cell * 4 + it * 4) = accS + accV;
            }
        }
    }
    __syncthreads();

    // one row block per wave and <= 8 k-blocks per product: register-prefetched tiles (wa = current, wb = next)
    const bool pf = PF && nrb <= CHAIN_NW && ncb <= CHAIN_NW;
    f32x4 wa[8], wb[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { wa[u] = (f32x4){0.f, 0.f, 0.f, 0.f}; wb[u] = wa[u]; }
    // products in order: P_1..P_{L-1} (prev-out block of cell i), head, then backward: head^T, outT_{L-1}..outT_1
    if (pf) {
        if (L > 1) { if (wave < nrb) issue_tiles(wa, W + cd.seg_off[1][2] + (int64_t)wave * nrb * 256, nrb, lane); }
        else if (wave < ncb) issue_tiles(wa, W + cd.head_off + (int64_t)wave * nrb * 256, nrb, lane);
    }

    CT_STAMP(0);
    // ------------------------------------------------------------------ forward chain
    for (int i = 0; i < L; ++i) {
        CT_STAMP(1 + i);
        if (pf && i >= 1) {   // wa holds P_i; request the NEXT product's tiles now: P_{i+1}, or the head after the last cell
            if (i + 1 < L) { if (wave < nrb) issue_tiles(wb, W + cd.seg_off[i + 1][2] + (int64_t)wave * nrb * 256, nrb, lane); }
            else if (wave < ncb) issue_tiles(wb, W + cd.head_off + (int64_t)wave * nrb * 256, nrb, lane);
        }
        const float* xprev = xo_l + ((i + 1) & 1) * Bp * SX;
        float* xcur = xo_l + (i & 1) * Bp * SX;
        const int nl = (nlbits >> (2 * i)) & 3;
        const int64_t vb = cvec_off + (int64_t)i * g.vec_cell_stride;
        const int vbl = i * g.vec_cell_stride;
        float sgS = 1.0f, sgV = 1.0f;
        if (g.alphas) {
            const float sg = 1.0f / (1.0f + expf(-vecW[vbl + 5 * Rp]));
            sgS = sg;
            sgV = 1.0f - sg;
            if (tid == 0) {
                sb[g.sb_gsc + i * 2] = sgS;
                sb[g.sb_gsc + i * 2 + 1] = sgV;
            }
        }
        for (int rb = wave; rb < nrb; rb += CHAIN_NW) {
            const int r = rb * 16 + l15;
            const bool colok = r < R;
            // independent loads first: vector parameters of this column
            const float bias = vecW[vbl + VEC_B * Rp + r];
            float gam = 1.f, bet = 0.f;
            if (g.bn) { gam = vecW[vbl + VEC_G * Rp + r]; bet = vecW[vbl + VEC_BE * Rp + r]; }
            f32x4 acc[MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int64_t o = ((((int64_t)i * nrb + rb) * MB + mb) << 8) + lane * 4;
                acc[mb] = *reinterpret_cast<const f32x4*>(yf_l + o);
                if (g.alphas) {   // keep raw S-V difference for d(alpha); scale the two modality sums
                    const f32x4 yv = *reinterpret_cast<const f32x4*>(yf_l + sav_plane + o);
                    *reinterpret_cast<f32x4*>(sav + 2 * sav_plane + o) = acc[mb] - yv;
                    acc[mb] = acc[mb] * sgS + yv * sgV;
                }
            }
            if (i > 0) {
                if (pf) mma_tiles<MB>(acc, xprev, SX, wa, nrb, lane);
                else lds_x_times_tiles<MB>(acc, xprev, SX, W + cd.seg_off[i][2] + (int64_t)rb * nrb * 256, 256, nrb, lane);
            }
            float av[MB][4];
            float s = 0.f;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int b = mb * 16 + 4 * lg + q;
                    const float v = act_fwd(acc[mb][q] + bias, nl);
                    av[mb][q] = v;
                    if (b < nvalid) s += v;
                }
            float zv[MB][4];
            if (g.bn) {
                const float mu = colsum(s) / nf;
                float s2 = 0.f;
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int b = mb * 16 + 4 * lg + q;
                        const float dlt = av[mb][q] - mu;
                        if (b < nvalid) s2 += dlt * dlt;
                    }
                const float var = colsum(s2) / nf;
                const float rstd = 1.0f / sqrtf(var + g.bn_eps);
                f32x4 xh4[MB];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float xh = (av[mb][q] - mu) * rstd;
                        xh4[mb][q] = xh;
                        zv[mb][q] = xh * gam + bet;
                    }
                if (lg == 0) {
                    rstd_l[i * Rp + r] = rstd;
                    if (colok) {   // running stats: momentum 0.1, unbiased variance
                        float rm = vecW[vbl + VEC_RM * Rp + r], rv = vecW[vbl + VEC_RV * Rp + r];
                        const float unb = var * (nf / (nf - 1.0f));
                        rm += g.bn_mom * (mu - rm);
                        rv += g.bn_mom * (unb - rv);
                        W[vb + VEC_RM * Rp + r] = rm;
                        W[vb + VEC_RV * Rp + r] = rv;
                    }
                }
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
                    *reinterpret_cast<f32x4*>(sav + sav_plane + ((((int64_t)i * nrb + rb) * MB + mb) << 8) + lane * 4) = xh4[mb];
            } else {
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) zv[mb][q] = av[mb][q];
            }
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                f32x4 a4;
#pragma unroll
                for (int q = 0; q < 4; ++q) a4[q] = av[mb][q];
                *reinterpret_cast<f32x4*>(sav + ((((int64_t)i * nrb + rb) * MB + mb) << 8) + lane * 4) = a4;
            }
            float* xo_g = sb + g.sb_xo + (int64_t)i * Bp * Rp;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int b = mb * 16 + 4 * lg + q;
                    float o = zv[mb][q];
                    if (g.use_drop)
                        o = drop_keep(h0, i, (uint32_t)(b * R + r), g.drop_thr) ? o * g.drop_scale : 0.0f;
                    if (!(colok && b < nvalid)) o = 0.0f;
                    xcur[b * SX + r] = o;
                    xo_g[b * Rp + r] = o;
                }
        }
        if (pf && i >= 1) {
#pragma unroll
            for (int u = 0; u < 8; ++u) wa[u] = wb[u];
        }
        lds_barrier();
    }

    CT_STAMP(5);
    // ------------------------------------------------------------------ head + CE loss
    {
        const float* xl = xo_l + ((L - 1) & 1) * Bp * SX;
        if (pf && wave < nrb)   // first backward product: d_out = dlogits . Wc  (transposed head tiles of this row block)
            issue_tiles(wb, a.wt + cd.headT_off + (int64_t)wave * ncb * 256, ncb, lane);
        for (int cb = wave; cb < ncb; cb += CHAIN_NW) {
            f32x4 acc[MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) acc[mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int c = cb * 16 + l15;
            const float bias = vecW[g.vec_head + c];
            if (pf) mma_tiles<MB>(acc, xl, SX, wa, nrb, lane);
            else lds_x_times_tiles<MB>(acc, xl, SX, W + cd.head_off + (int64_t)cb * nrb * 256, 256, nrb, lane);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int q = 0; q < 4; ++q) lg_l[(mb * 16 + 4 * lg + q) * SC + c] = acc[mb][q] + bias;
        }
    }
    lds_barrier();
    CT_STAMP(6);
    if (g.loss_mode == 1) {
        if (tid < 4 * Bp) bce_rows(lg_l, SC, red_l, Bp, lab_l, a.tab.multilabel, a.pos_w, C, Cp, nvalid, tid);
    } else if (tid < LPR * Bp) {
        softmax_rows<MB>(a, lg_l, SC, red_l, lab_l, nvalid, nf, tid);
    }
    lds_barrier();
    if (tid == CHAIN_THREADS - 64) {   // last wave: keeps the read-modify-write of the statistics off wave 0
        float ls = 0.f, cs = 0.f;
        for (int b = 0; b < Bp; ++b) { ls += red_l[b]; cs += red_l[Bp + b]; }
        DevStats& st = a.stats[(int64_t)cgidx * a.E + a.epoch];
        st.train_loss += (double)ls;
        st.train_corr += (long long)cs;
        if (!(fabsf(ls) <= 3.0e38f)) a.status[cgidx] = 1;
    }
    CT_STAMP(7);
    // dlogits -> global (dy operand of the HEAD segment); head-bias Adam
    {
        float* dlg = sb + g.sb_dlog;
        for (int e = tid; e < Bp * Cp; e += CHAIN_THREADS) {
            const int b = e / Cp, c = e - b * Cp;
            dlg[e] = lg_l[b * SC + c];
        }
        const int hc = tid - (CHAIN_THREADS - 256);   // head-bias columns on the upper four waves
        if (hc >= 0 && hc < C) {
            float gsum = 0.f;
            for (int b = 0; b < Bp; ++b) gsum += lg_l[b * SC + hc];
            const int64_t o = cvec_off + g.vec_head + hc;
            float w = vecW[g.vec_head + hc], m = vecM[g.vec_head + hc], v = vecV[g.vec_head + hc];
            adam1(w, m, v, gsum, ac);
            W[o] = w; Mv[o] = m; Vv[o] = v;
        }
    }

    if (pf) {
#pragma unroll
        for (int u = 0; u < 8; ++u) wa[u] = wb[u];
    }

    // ------------------------------------------------------------------ backward chain
    for (int i = L - 1; i >= 0; --i) {
        CT_STAMP(8 + (L - 1 - i));
        if (pf && i >= 1 && wave < nrb)   // next backward product (cell i-1) uses the transposed prev-out block of cell i
            issue_tiles(wb, a.wt + cd.outT_off[i] + (int64_t)wave * nrb * 256, nrb, lane);
        const int nl = (nlbits >> (2 * i)) & 3;
        const int64_t vb = cvec_off + (int64_t)i * g.vec_cell_stride;
        const int vbl = i * g.vec_cell_stride;
        const bool from_head = (i == L - 1);
        const float* src = from_head ? lg_l : dy_l + ((i + 1) & 1) * Bp * SX;
        const int sstride = from_head ? SC : SX;
        const int nkk = from_head ? ncb : nrb;
        const float* T = a.wt + (from_head ? cd.headT_off : cd.outT_off[i + 1]);
        float* dcur = dy_l + (i & 1) * Bp * SX;
        float dalpha = 0.f;
        for (int rb = wave; rb < nrb; rb += CHAIN_NW) {
            const int r = rb * 16 + l15;
            const bool colok = r < R;
            // independent loads first
            float gr = 0.f;
            if (g.bn) gr = vecW[vbl + VEC_G * Rp + r] * rstd_l[i * Rp + r];
            int64_t ob = vb + VEC_B * Rp + r, og = vb + VEC_G * Rp + r, obe = vb + VEC_BE * Rp + r;
            float pw[3] = {0.f, 0.f, 0.f}, pm[3] = {0.f, 0.f, 0.f}, pv[3] = {0.f, 0.f, 0.f};
            if (lg == 0 && colok) {
                const int lb = vbl + VEC_B * Rp + r, lgm = vbl + VEC_G * Rp + r, lbe = vbl + VEC_BE * Rp + r;
                pw[0] = vecW[lb]; pm[0] = vecM[lb]; pv[0] = vecV[lb];
                if (g.bn) {
                    pw[1] = vecW[lgm]; pm[1] = vecM[lgm]; pv[1] = vecV[lgm];
                    pw[2] = vecW[lbe]; pm[2] = vecM[lbe]; pv[2] = vecV[lbe];
                }
            }
            f32x4 a4[MB], xh4[MB], df4[MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                a4[mb] = *reinterpret_cast<const f32x4*>(sav + ((((int64_t)i * nrb + rb) * MB + mb) << 8) + lane * 4);
                xh4[mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
                df4[mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (g.bn)
                    xh4[mb] = *reinterpret_cast<const f32x4*>(sav + sav_plane + ((((int64_t)i * nrb + rb) * MB + mb) << 8) + lane * 4);
                if (g.alphas)
                    df4[mb] = *reinterpret_cast<const f32x4*>(sav + 2 * sav_plane + ((((int64_t)i * nrb + rb) * MB + mb) << 8) + lane * 4);
            }
            f32x4 acc[MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) acc[mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (pf) mma_tiles<MB>(acc, src, sstride, wa, nkk, lane);
            else lds_x_times_tiles<MB>(acc, src, sstride, T + (int64_t)rb * nkk * 256, 256, nkk, lane);
            float dz[MB][4];
            float sdz = 0.f, sdzx = 0.f;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int b = mb * 16 + 4 * lg + q;
                    float d = acc[mb][q];
                    if (g.use_drop)
                        d = drop_keep(h0, i, (uint32_t)(b * R + r), g.drop_thr) ? d * g.drop_scale : 0.0f;
                    if (!(b < nvalid)) d = 0.f;
                    dz[mb][q] = d;
                    sdz += d;
                    if (g.bn) sdzx += d * xh4[mb][q];
                }
            float dgam = 0.f, dbet = 0.f;
            if (g.bn) {
                dbet = colsum(sdz);
                dgam = colsum(sdzx);
                const float k1 = dbet / nf, k2 = dgam / nf;
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int b = mb * 16 + 4 * lg + q;
                        const float da = gr * (dz[mb][q] - k1 - xh4[mb][q] * k2);
                        dz[mb][q] = b < nvalid ? da : 0.f;
                    }
            }
            float sdy = 0.f;
            float* dy_g = sb + g.sb_dy + (int64_t)i * Bp * Rp;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int b = mb * 16 + 4 * lg + q;
                    float dy = act_bwd(a4[mb][q], dz[mb][q], nl);
                    if (!colok) dy = 0.f;
                    sdy += dy;
                    dalpha += dy * df4[mb][q];
                    dcur[b * SX + r] = dy;
                    dy_g[b * Rp + r] = dy;
                }
            const float db = colsum(sdy);
            if (lg == 0 && colok) {   // Adam on the column's vector parameters (one owner lane per column)
                adam1(pw[0], pm[0], pv[0], db, ac);
                W[ob] = pw[0]; Mv[ob] = pm[0]; Vv[ob] = pv[0];
                if (g.bn) {
                    adam1(pw[1], pm[1], pv[1], dgam, ac);
                    W[og] = pw[1]; Mv[og] = pm[1]; Vv[og] = pv[1];
                    adam1(pw[2], pm[2], pv[2], dbet, ac);
                    W[obe] = pw[2]; Mv[obe] = pm[2]; Vv[obe] = pv[2];
                }
            }
        }
        if (pf && i >= 1) {
#pragma unroll
            for (int u = 0; u < 8; ++u) wa[u] = wb[u];
        }
        if (g.alphas) {   // d(alpha_i) = sigma'(alpha) * sum_{b,r} dy[b,r] * (yS_raw - yV_raw)[b,r]
            for (int o = 32; o > 0; o >>= 1) dalpha += __shfl_xor(dalpha, o);
            if (lane == 0) red_l[2 * Bp + wave] = dalpha;
        }
        lds_barrier();
        if (g.alphas && tid == 0) {
            float tot = 0.f;
            for (int w = 0; w < CHAIN_NW; ++w) tot += red_l[2 * Bp + w];
            const int64_t o = vb + 5 * Rp;
            float w = vecW[vbl + 5 * Rp], m = vecM[vbl + 5 * Rp], v = vecV[vbl + 5 * Rp];
            const float sg = 1.0f / (1.0f + expf(-w));
            adam1(w, m, v, tot * sg * (1.0f - sg), ac);
            W[o] = w; Mv[o] = m; Vv[o] = v;
        }
        if (g.alphas) lds_barrier();
    }
    CT_STAMP(12);
}

// ------------------------------------------------------------------------------------------------
// chain_lean — the same train-step chain for ONE row block (R <= 16) and <= 4 class blocks (C <= 64): the reference's
// search defaults (inner_representation_size 16, main_searchable_ntu.py:26-45).  A 16-wide cell is a string of ~15
// dependent little steps, and in the general chain_body every one of them pays a workgroup barrier, fresh scalar loads of
// the candidate record, address arithmetic for up to 32 row blocks and a global round trip for its weight tile.  Here:
//   * everything a step needs from global memory (labels, vector block, EVERY product's weight tile, the sweep's partial
//     sums) is requested at kernel entry — one memory latency for the whole chain;
//   * wave 0 owns the single row block and runs all L cells forward (and later backward) back to back with no barrier:
//     the activations of all cells stay in LDS (xo_l / dy_l [L][Bp][20]) together with the saved activations;
//   * the other seven waves do the bulk work around it: partial-sum reduction, head / softmax, coalesced copies of
//     out_i, dy_i and dlogits to the step buffers the sweep reads, statistics, head-bias Adam.
// Arithmetic (operation order included) is that of chain_body.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 pick4(const f32x4 (&t)[MFAS_MAX_CELLS], int i) {
    switch (i) { case 0: return t[0]; case 1: return t[1]; case 2: return t[2]; default: return t[3]; }
}

template <int MB>
__device__ __forceinline__ void chain_lean(const ChainArgs& a, const int bid, float* lds) {
#ifdef MFAS_CHAIN_TIMING
    const unsigned long long ct0 = __builtin_readcyclecounter();
#endif
    const CandDev& cd = a.cands[bid];
    const Geo& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    constexpr int Bp = MB * 16;
    constexpr int LPR = (STEP_THREADS / Bp) < 16 ? (STEP_THREADS / Bp) : 16;
    constexpr int Rp = 16, SX = Rp + 4;
    const int Cp = g.Cp, ncb = g.ncb, R = g.R, C = g.C, L = cd.L, SC = Cp + 4;
    constexpr int sav_plane = MFAS_MAX_CELLS * MB * 256;
    const int nvec = MFAS_MAX_CELLS * g.vec_cell_stride + Cp;
    float* xo_l = lds;                                          // [L][Bp][SX] out_i of every cell
    float* dy_l = xo_l + MFAS_MAX_CELLS * Bp * SX;              // [L][Bp][SX] dy_i of every cell
    float* lg_l = dy_l + MFAS_MAX_CELLS * Bp * SX;              // [Bp][SC] logits -> dlogits
    float* rstd_l = lg_l + Bp * SC;                             // [L][Rp]
    float* red_l = rstd_l + MFAS_MAX_CELLS * Rp;                // [2*Bp + 16]
    int* lab_l = reinterpret_cast<int*>(red_l + 2 * Bp + 16);   // [Bp]
    float* yf_l = reinterpret_cast<float*>(lab_l + Bp);         // [1 or 2][L][MB][256] reduced feature sums
    float* vec_l = yf_l + (g.alphas ? 2 : 1) * sav_plane;       // [3][nvec] vector block + Adam state
    float* sav_a = vec_l + 3 * nvec;                            // [L][MB][256] activations
    float* sav_x = sav_a + sav_plane;                           // xhat (batchnorm only)
    float* sav_d = sav_a + (g.bn ? 2 : 1) * sav_plane;          // yS - yV (alphas only)

    float* W = a.plane;
    float* Mv = a.plane + a.plane_stride;
    float* Vv = Mv + a.plane_stride;
    float* sb = a.stepbuf + cd.step_off;
    const int64_t cvec_off = cd.vec_off;
    const int cgidx = cd.gidx;
    int nlbits = 0;
#pragma unroll
    for (int i = 0; i < MFAS_MAX_CELLS; ++i) nlbits |= (cd.conf[i][2] & 3) << (2 * i);
    const int nvalid = a.nvalid;
    const float nf = (float)nvalid;
    const AdamC ac = a.ac;
    const uint32_t h0 = lowbias32(cd.drop_seed + 0x9E3779B9U * (uint32_t)(a.gstep + 1));

    // ------------------------------------------------------------------ entry: every global read of the chain is
    // requested here, in the order the results are needed (the memory counter retires in order): the sweep's partial
    // sums first, then the vector block, the weight tiles of all products and last the labels (a dependent pair of loads
    // that nothing needs before the loss, fetched by wave 1 so that wave 0 never waits for them)
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    constexpr int per_cell = MB * 64;   // float4 partial-sum items per cell; L * per_cell <= 512: one item per thread
    constexpr int PB = 16;
    const bool has_item = tid < L * per_cell;
    const int pi = has_item ? tid / per_cell : 0, pit = tid - pi * per_cell;
    const int ns = cd.nch_s[pi], nch = has_item ? ns + cd.nch_v[pi] : 0;
    const float* part = sb + g.sb_part + (((int64_t)cd.part_cell_off[pi] * MB) << 8) + pit * 4;
    f32x4 p8[PB];
#pragma unroll
    for (int u = 0; u < PB; ++u)
        if (u < nch) p8[u] = *reinterpret_cast<const f32x4*>(part + (((int64_t)u * MB) << 8));
    float vw = 0.f, vm = 0.f, vv = 0.f;   // nvec = 4 * 96 + Cp <= 448: one element of the vector block per thread
    if (tid < nvec) { vw = W[cvec_off + tid]; vm = Mv[cvec_off + tid]; vv = Vv[cvec_off + tid]; }
    f32x4 tP[MFAS_MAX_CELLS], tT[MFAS_MAX_CELLS], tHT[4], tH = z4;   // prev-out tile of cell i, its transpose, head^T, head
#pragma unroll
    for (int i = 0; i < MFAS_MAX_CELLS; ++i) { tP[i] = z4; tT[i] = z4; tHT[i] = z4; }
    if (wave == 0) {
#pragma unroll
        for (int i = 1; i < MFAS_MAX_CELLS; ++i)
            if (i < L) {
                tP[i] = *reinterpret_cast<const f32x4*>(W + cd.seg_off[i][2] + lane * 4);
                tT[i] = *reinterpret_cast<const f32x4*>(a.wt + cd.outT_off[i] + lane * 4);
            }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (u < ncb) tHT[u] = *reinterpret_cast<const f32x4*>(a.wt + cd.headT_off + ((int64_t)u << 8) + lane * 4);
    }
    if (wave < ncb) tH = *reinterpret_cast<const f32x4*>(W + cd.head_off + ((int64_t)wave << 8) + lane * 4);
    int lab = 0;
    if (wave == 1 && lane < nvalid) {
        const int64_t row = a.order ? (int64_t)a.order[a.pos_t + lane] : (int64_t)(a.base_t + lane);
        lab = g.loss_mode == 0 ? a.tab.label[row] : (int)row;   // mode 1 keeps the table row for the multi-hot targets
    }
    // dropout keep bits of this lane's elements (wave 0 owns the row block): bit (i*MB + mb)*4 + q — computed while the
    // loads above are in flight, used by the forward AND the backward pass
    const int r = l15;
    const bool colok = r < R;
    uint32_t keep = 0xFFFFFFFFu;
    if (wave == 0 && g.use_drop) {
#pragma unroll
        for (int i = 0; i < MFAS_MAX_CELLS; ++i)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (!drop_keep(h0, i, (uint32_t)((mb * 16 + 4 * lg + q) * R + r), g.drop_thr)) keep &= ~(1u << ((i * MB + mb) * 4 + q));
    }
    // phase 0: reduce the sweep's column-chunk partial sums (fixed order) into LDS; stage the vector block
    if (has_item) {
        f32x4 accS = z4, accV = z4;
#pragma unroll
        for (int u = 0; u < PB; ++u)
            if (u < nch) {
                if (u < ns) accS += p8[u]; else accV += p8[u];
            }
        for (int ch0 = PB; ch0 < nch; ch0 += PB) {
#pragma unroll
            for (int u = 0; u < PB; ++u)
                if (ch0 + u < nch) p8[u] = *reinterpret_cast<const f32x4*>(part + (((int64_t)(ch0 + u) * MB) << 8));
#pragma unroll
            for (int u = 0; u < PB; ++u)
                if (ch0 + u < nch) {
                    if (ch0 + u < ns) accS += p8[u]; else accV += p8[u];
                }
        }
        if (g.alphas) {
            *reinterpret_cast<f32x4*>(yf_l + tid * 4) = accS;
            *reinterpret_cast<f32x4*>(yf_l + sav_plane + tid * 4) = accV;
        } else {
            *reinterpret_cast<f32x4*>(yf_l + tid * 4) = accS + accV;
        }
    }
    if (tid < nvec) { vec_l[tid] = vw; vec_l[nvec + tid] = vm; vec_l[2 * nvec + tid] = vv; }
    const float* vecW = vec_l;
    const float* vecM = vec_l + nvec;
    const float* vecV = vec_l + 2 * nvec;
    lds_barrier();
    CT_STAMP(0);

    // ------------------------------------------------------------------ forward: wave 0, all cells, no barrier
    if (wave == 0) {
        for (int i = 0; i < L; ++i) {
            CT_STAMP(1 + i);
            const int nl = (nlbits >> (2 * i)) & 3;
            const int64_t vb = cvec_off + (int64_t)i * g.vec_cell_stride;
            const int vbl = i * g.vec_cell_stride;
            const float bias = vecW[vbl + VEC_B * Rp + r];
            float gam = 1.f, bet = 0.f;
            if (g.bn) { gam = vecW[vbl + VEC_G * Rp + r]; bet = vecW[vbl + VEC_BE * Rp + r]; }
            float sgS = 1.0f, sgV = 1.0f;
            if (g.alphas) {
                const float sg = 1.0f / (1.0f + expf(-vecW[vbl + 5 * Rp]));
                sgS = sg;
                sgV = 1.0f - sg;
                if (lane == 0) {
                    sb[g.sb_gsc + i * 2] = sgS;
                    sb[g.sb_gsc + i * 2 + 1] = sgV;
                }
            }
            f32x4 acc[MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int o = ((i * MB + mb) << 8) + lane * 4;
                acc[mb] = *reinterpret_cast<const f32x4*>(yf_l + o);
                if (g.alphas) {
                    const f32x4 yv = *reinterpret_cast<const f32x4*>(yf_l + sav_plane + o);
                    *reinterpret_cast<f32x4*>(sav_d + o) = acc[mb] - yv;
                    acc[mb] = acc[mb] * sgS + yv * sgV;
                }
            }
            if (i > 0) {
                const f32x4 w = pick4(tP, i);
                const float* xprev = xo_l + (i - 1) * Bp * SX;
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    const f32x4 x4 = *reinterpret_cast<const f32x4*>(xprev + (mb * 16 + l15) * SX + 4 * lg);
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[mb] = MFMA16(x4[q], w[q], acc[mb]);
                }
            }
            float av[MB][4];
            float s = 0.f;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int b = mb * 16 + 4 * lg + q;
                    const float v = act_fwd(acc[mb][q] + bias, nl);
                    av[mb][q] = v;
                    if (b < nvalid) s += v;
                }
            float zv[MB][4];
            if (g.bn) {
                const float mu = colsum(s) / nf;
                float s2 = 0.f;
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int b = mb * 16 + 4 * lg + q;
                        const float dlt = av[mb][q] - mu;
                        if (b < nvalid) s2 += dlt * dlt;
                    }
                const float var = colsum(s2) / nf;
                const float rstd = 1.0f / sqrtf(var + g.bn_eps);
                f32x4 xh4[MB];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float xh = (av[mb][q] - mu) * rstd;
                        xh4[mb][q] = xh;
                        zv[mb][q] = xh * gam + bet;
                    }
                if (lg == 0) {
                    rstd_l[i * Rp + r] = rstd;
                    if (colok) {   // running stats: momentum 0.1, unbiased variance
                        float rm = vecW[vbl + VEC_RM * Rp + r], rv = vecW[vbl + VEC_RV * Rp + r];
                        const float unb = var * (nf / (nf - 1.0f));
                        rm += g.bn_mom * (mu - rm);
                        rv += g.bn_mom * (unb - rv);
                        W[vb + VEC_RM * Rp + r] = rm;
                        W[vb + VEC_RV * Rp + r] = rv;
                    }
                }
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
                    *reinterpret_cast<f32x4*>(sav_x + ((i * MB + mb) << 8) + lane * 4) = xh4[mb];
            } else {
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) zv[mb][q] = av[mb][q];
            }
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                f32x4 a4;
#pragma unroll
                for (int q = 0; q < 4; ++q) a4[q] = av[mb][q];
                *reinterpret_cast<f32x4*>(sav_a + ((i * MB + mb) << 8) + lane * 4) = a4;
            }
            float* xcur = xo_l + i * Bp * SX;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int b = mb * 16 + 4 * lg + q;
                    float o = zv[mb][q];
                    if (g.use_drop) o = ((keep >> ((i * MB + mb) * 4 + q)) & 1u) ? o * g.drop_scale : 0.0f;
                    if (!(colok && b < nvalid)) o = 0.0f;
                    xcur[b * SX + r] = o;
                }
        }
    }
    if (wave == 1 && lane < Bp) lab_l[lane] = lab;   // visible to the loss after the head's barrier
    lds_barrier();
    CT_STAMP(5);

    // ------------------------------------------------------------------ out_i -> step buffer (x operand of the sweep's
    // OUT / HEAD segments), coalesced, by everyone; head on waves < ncb
    {
        float* xo_g = sb + g.sb_xo;   // [L][Bp][Rp]
        for (int e = tid; e < L * Bp * Rp; e += CHAIN_THREADS) xo_g[e] = xo_l[(e >> 4) * SX + (e & 15)];
        const float* xl = xo_l + (L - 1) * Bp * SX;
        if (wave < ncb) {
            const int c = wave * 16 + l15;
            const float bias = vecW[g.vec_head + c];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                f32x4 acc = z4;
                const f32x4 x4 = *reinterpret_cast<const f32x4*>(xl + (mb * 16 + l15) * SX + 4 * lg);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc = MFMA16(x4[q], tH[q], acc);
#pragma unroll
                for (int q = 0; q < 4; ++q) lg_l[(mb * 16 + 4 * lg + q) * SC + c] = acc[q] + bias;
            }
        }
    }
    lds_barrier();
    CT_STAMP(6);
    if (g.loss_mode == 1) {
        if (tid < 4 * Bp) bce_rows(lg_l, SC, red_l, Bp, lab_l, a.tab.multilabel, a.pos_w, C, Cp, nvalid, tid);
    } else if (tid < LPR * Bp) {
        softmax_rows<MB>(a, lg_l, SC, red_l, lab_l, nvalid, nf, tid);
    }
    lds_barrier();
    CT_STAMP(7);
    if (tid == CHAIN_THREADS - 64) {
        float ls = 0.f, cs = 0.f;
        for (int b = 0; b < Bp; ++b) { ls += red_l[b]; cs += red_l[Bp + b]; }
        DevStats& st = a.stats[(int64_t)cgidx * a.E + a.epoch];
        st.train_loss += (double)ls;
        st.train_corr += (long long)cs;
        if (!(fabsf(ls) <= 3.0e38f)) a.status[cgidx] = 1;
    }
    if (wave != 0) {   // dlogits -> step buffer (dy operand of the HEAD segment); head-bias Adam
        float* dlg = sb + g.sb_dlog;
        for (int e = tid - 64; e < Bp * Cp; e += CHAIN_THREADS - 64) {
            const int b = e / Cp, c = e - b * Cp;
            dlg[e] = lg_l[b * SC + c];
        }
        const int hc = tid - (CHAIN_THREADS - 256);
        if (hc >= 0 && hc < C) {
            float gsum = 0.f;
            for (int b = 0; b < Bp; ++b) gsum += lg_l[b * SC + hc];
            const int64_t o = cvec_off + g.vec_head + hc;
            float w = vecW[g.vec_head + hc], m = vecM[g.vec_head + hc], v = vecV[g.vec_head + hc];
            adam1(w, m, v, gsum, ac);
            W[o] = w; Mv[o] = m; Vv[o] = v;
        }
    } else {
        // -------------------------------------------------------------- backward: wave 0, all cells, no barrier
        for (int i = L - 1; i >= 0; --i) {
            CT_STAMP(8 + (L - 1 - i));
            const int nl = (nlbits >> (2 * i)) & 3;
            const int64_t vb = cvec_off + (int64_t)i * g.vec_cell_stride;
            const int vbl = i * g.vec_cell_stride;
            const bool from_head = (i == L - 1);
            float gr = 0.f;
            if (g.bn) gr = vecW[vbl + VEC_G * Rp + r] * rstd_l[i * Rp + r];
            const int64_t ob = vb + VEC_B * Rp + r, og = vb + VEC_G * Rp + r, obe = vb + VEC_BE * Rp + r;
            float pw[3] = {0.f, 0.f, 0.f}, pm[3] = {0.f, 0.f, 0.f}, pv[3] = {0.f, 0.f, 0.f};
            if (lg == 0 && colok) {
                const int lb = vbl + VEC_B * Rp + r, lgm = vbl + VEC_G * Rp + r, lbe = vbl + VEC_BE * Rp + r;
                pw[0] = vecW[lb]; pm[0] = vecM[lb]; pv[0] = vecV[lb];
                if (g.bn) {
                    pw[1] = vecW[lgm]; pm[1] = vecM[lgm]; pv[1] = vecV[lgm];
                    pw[2] = vecW[lbe]; pm[2] = vecM[lbe]; pv[2] = vecV[lbe];
                }
            }
            f32x4 a4[MB], xh4[MB], df4[MB], acc[MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int o = ((i * MB + mb) << 8) + lane * 4;
                a4[mb] = *reinterpret_cast<const f32x4*>(sav_a + o);
                xh4[mb] = z4;
                df4[mb] = z4;
                if (g.bn) xh4[mb] = *reinterpret_cast<const f32x4*>(sav_x + o);
                if (g.alphas) df4[mb] = *reinterpret_cast<const f32x4*>(sav_d + o);
                acc[mb] = z4;
            }
            if (from_head) {   // d_out = dlogits . Wc: even / odd class blocks in two chains (as mma_tiles)
                f32x4 acc2[MB];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) acc2[mb] = z4;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (u < ncb) {
#pragma unroll
                        for (int mb = 0; mb < MB; ++mb) {
                            const f32x4 x4 = *reinterpret_cast<const f32x4*>(lg_l + (mb * 16 + l15) * SC + u * 16 + 4 * lg);
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                if (u & 1) acc2[mb] = MFMA16(x4[q], tHT[u][q], acc2[mb]);
                                else acc[mb] = MFMA16(x4[q], tHT[u][q], acc[mb]);
                            }
                        }
                    }
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) acc[mb] += acc2[mb];
            } else {
                const f32x4 w = pick4(tT, i + 1);
                const float* src = dy_l + (i + 1) * Bp * SX;
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    const f32x4 x4 = *reinterpret_cast<const f32x4*>(src + (mb * 16 + l15) * SX + 4 * lg);
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[mb] = MFMA16(x4[q], w[q], acc[mb]);
                }
            }
            float dz[MB][4];
            float sdz = 0.f, sdzx = 0.f;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int b = mb * 16 + 4 * lg + q;
                    float d = acc[mb][q];
                    if (g.use_drop) d = ((keep >> ((i * MB + mb) * 4 + q)) & 1u) ? d * g.drop_scale : 0.0f;
                    if (!(b < nvalid)) d = 0.f;
                    dz[mb][q] = d;
                    sdz += d;
                    if (g.bn) sdzx += d * xh4[mb][q];
                }
            float dgam = 0.f, dbet = 0.f;
            if (g.bn) {
                dbet = colsum(sdz);
                dgam = colsum(sdzx);
                const float k1 = dbet / nf, k2 = dgam / nf;
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int b = mb * 16 + 4 * lg + q;
                        const float da = gr * (dz[mb][q] - k1 - xh4[mb][q] * k2);
                        dz[mb][q] = b < nvalid ? da : 0.f;
                    }
            }
            float sdy = 0.f, dalpha = 0.f;
            float* dcur = dy_l + i * Bp * SX;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int b = mb * 16 + 4 * lg + q;
                    float dy = act_bwd(a4[mb][q], dz[mb][q], nl);
                    if (!colok) dy = 0.f;
                    sdy += dy;
                    dalpha += dy * df4[mb][q];
                    dcur[b * SX + r] = dy;
                }
            const float db = colsum(sdy);
            if (lg == 0 && colok) {   // Adam on the column's vector parameters (one owner lane per column)
                adam1(pw[0], pm[0], pv[0], db, ac);
                W[ob] = pw[0]; Mv[ob] = pm[0]; Vv[ob] = pv[0];
                if (g.bn) {
                    adam1(pw[1], pm[1], pv[1], dgam, ac);
                    W[og] = pw[1]; Mv[og] = pm[1]; Vv[og] = pv[1];
                    adam1(pw[2], pm[2], pv[2], dbet, ac);
                    W[obe] = pw[2]; Mv[obe] = pm[2]; Vv[obe] = pv[2];
                }
            }
            if (g.alphas) {   // d(alpha_i) = sigma'(alpha) * sum_{b,r} dy[b,r] * (yS_raw - yV_raw)[b,r]
                for (int o = 32; o > 0; o >>= 1) dalpha += __shfl_xor(dalpha, o);
                if (lane == 0) {
                    const float tot = dalpha;
                    const int64_t o = vb + 5 * Rp;
                    float w = vecW[vbl + 5 * Rp], m = vecM[vbl + 5 * Rp], v = vecV[vbl + 5 * Rp];
                    const float sg = 1.0f / (1.0f + expf(-w));
                    adam1(w, m, v, tot * sg * (1.0f - sg), ac);
                    W[o] = w; Mv[o] = m; Vv[o] = v;
                }
            }
        }
    }
    lds_barrier();
    CT_STAMP(12);
    {   // dy_i -> step buffer (dy operand of the sweep), coalesced
        float* dy_g = sb + g.sb_dy;   // [L][Bp][Rp]
        for (int e = tid; e < L * Bp * Rp; e += CHAIN_THREADS) dy_g[e] = dy_l[(e >> 4) * SX + (e & 15)];
    }
}

// ------------------------------------------------------------------------------------------------
// k_step — ONE launch per half-step: blocks [0, nchain) run the chain of one candidate group while the other
// blocks run the sweep of the OTHER group (candidates are independent).  The latency-bound chain hides under
// the HBM-bound sweep; kernel boundaries carry every dependency (chain(t) -> sweep(t) -> chain(t+1) of a group).
// ------------------------------------------------------------------------------------------------
struct StepArgs {
    SweepArgs sa;
    ChainArgs ca;
    int32_t nchain, _pad;
};

template <int MB, bool NT, int WPE, bool LEAN>
__global__ void __launch_bounds__(STEP_THREADS, WPE) k_step(const StepArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int bid = (int)blockIdx.x;
    if (bid < a.nchain) {
        if constexpr (LEAN) chain_lean<MB>(a.ca, bid, lds);
        else chain_body<MB, false>(a.ca, bid, lds);
    }
    else if (bid < a.nchain + a.sa.ntap) sweep_tap_body<MB, NT, SweepU<MB, WPE>::v>(a.sa, bid - a.nchain, lds);
    else sweep_body<MB, NT, SweepU<MB, WPE>::v>(a.sa, bid - a.nchain - a.sa.ntap, lds);
}

// Standalone chain launch (small populations: chain and sweep run back to back, so the chain's latency is on the
// critical path): full register budget, next-product weight tiles prefetched into registers.
template <int MB, bool LEAN>
__global__ void __launch_bounds__(STEP_THREADS, 2) k_chain(const ChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if constexpr (LEAN) chain_lean<MB>(a, (int)blockIdx.x, lds);
    else chain_body<MB, true>(a, (int)blockIdx.x, lds);
}

// ------------------------------------------------------------------------------------------------
// k_eval — eval-mode forward (BN running stats, no dropout) over a block of table rows
// ------------------------------------------------------------------------------------------------
struct EvalArgs {
    const CandDev* cands;
    const float* plane;
    mfas_table tab;
    int64_t row0, nrows;
    int32_t cand0;
    int32_t epoch, E;
    Geo g;
    float* logits;        // optional (nrows, C) for candidate cand0
    DevStats* stats;      // optional: dev_corr / dev_loss of stats[cand*E + epoch]
    long long* corr_out;  // optional single counter
    const float* pos_w;   // loss_mode 1
};

#define EVAL_CE 128   // staged feature columns per pass

template <int MBE, int NRBW>
__global__ void __launch_bounds__(256) k_eval(const EvalArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int cand = a.cand0 + blockIdx.y;
    const CandDev& cd = a.cands[cand];
    const Geo& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    constexpr int ME = MBE * 16;
    const int Rp = g.Rp, nrb = g.nrb, Cp = g.Cp, ncb = g.ncb, R = g.R, C = g.C, L = cd.L;
    const int SX = Rp + 4, SC = Cp + 4, SS = EVAL_CE + 4;
    // LDS kept to <= 80 KiB so that two workgroups share a CU (one stages features while the other runs MFMAs):
    // ONE activation buffer (extra barrier per cell) and the logits alias the feature staging tile.
    float* xs = lds;                     // [ME][SS]   feature staging tile; later the logits [ME][SC]
    float* xo_l = xs + ME * max(SS, SC); // [ME][SX]   out_{i-1} -> out_i
    float* lg_l = xs;
    const float* W = a.plane;
    const int64_t brow = a.row0 + (int64_t)blockIdx.x * ME;
    const int nvalid = (int)min((int64_t)ME, a.row0 + a.nrows - brow);

    for (int i = 0; i < L; ++i) {
        const float* xprev = xo_l;
        float* xcur = xo_l;
        const int nl = cd.conf[i][2];
        const int64_t vb = cd.vec_off + (int64_t)i * g.vec_cell_stride;
        float sgS = 1.0f, sgV = 1.0f;
        if (g.alphas) {
            const float sg = 1.0f / (1.0f + expf(-W[vb + 5 * Rp]));
            sgS = sg;
            sgV = 1.0f - sg;
        }
        f32x4 acc[NRBW][MBE];
#pragma unroll
        for (int j = 0; j < NRBW; ++j)
#pragma unroll
            for (int mb = 0; mb < MBE; ++mb) acc[j][mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int sv = 0; sv < 2; ++sv) {
            const int tap = cd.conf[i][sv];
            const void* tp = sv == 0 ? a.tab.s[tap] : a.tab.v[tap];
            const int cols = cd.seg_cols[i][sv], cc = cd.seg_cc[i][sv];
            const int tw = sv == 0 ? g.sw[tap] : g.vw[tap];
            if (g.alphas && sv == 1) {   // switch modality: fold the S sum with its scale, restart for V
#pragma unroll
                for (int j = 0; j < NRBW; ++j)
#pragma unroll
                    for (int mb = 0; mb < MBE; ++mb) acc[j][mb] = acc[j][mb] * (sgS / sgV);
            }
            for (int c0 = 0; c0 < cols; c0 += EVAL_CE) {
                const int nc = min(EVAL_CE, cols - c0);
                if constexpr (NRBW <= 2) {
                    // this chunk's weight tiles are requested BEFORE the feature staging so that their L2 latency
                    // overlaps the staging barriers (8 k-blocks x NRBW row blocks = up to 64 VGPRs)
                    f32x4 wt[EVAL_CE / 16][NRBW];
#pragma unroll
                    for (int kbl = 0; kbl < EVAL_CE / 16; ++kbl)
#pragma unroll
                        for (int j = 0; j < NRBW; ++j) {
                            const int rb = wave + 4 * j;
                            if (kbl < (nc >> 4) && rb < nrb)
                                wt[kbl][j] = *reinterpret_cast<const f32x4*>(W + tile_addr(cd.seg_off[i][sv], Rp, cc, rb, (c0 >> 4) + kbl) + lane * 4);
                        }
                    __syncthreads();
                    stage_table(xs, SS, tp, a.tab.dtype, tw, c0, nc, nullptr, 0, (int)brow, nvalid, ME, tid, 256);
                    __syncthreads();
#pragma unroll
                    for (int kbl = 0; kbl < EVAL_CE / 16; ++kbl)
                        if (kbl < (nc >> 4)) {
#pragma unroll
                            for (int j = 0; j < NRBW; ++j) {
                                const int rb = wave + 4 * j;
                                if (rb < nrb) {
#pragma unroll
                                    for (int mb = 0; mb < MBE; ++mb) {
                                        const f32x4 x4 = *reinterpret_cast<const f32x4*>(xs + (mb * 16 + l15) * SS + kbl * 16 + 4 * lg);
#pragma unroll
                                        for (int q = 0; q < 4; ++q) acc[j][mb] = MFMA16(x4[q], wt[kbl][j][q], acc[j][mb]);
                                    }
                                }
                            }
                        }
                } else {
                    __syncthreads();
                    stage_table(xs, SS, tp, a.tab.dtype, tw, c0, nc, nullptr, 0, (int)brow, nvalid, ME, tid, 256);
                    __syncthreads();
                    for (int kbl = 0; kbl < (nc >> 4); ++kbl) {
                        const int kb = (c0 >> 4) + kbl;
#pragma unroll
                        for (int j = 0; j < NRBW; ++j) {
                            const int rb = wave + 4 * j;
                            if (rb < nrb) {
                                const f32x4 w4 = *reinterpret_cast<const f32x4*>(W + tile_addr(cd.seg_off[i][sv], Rp, cc, rb, kb) + lane * 4);
#pragma unroll
                                for (int mb = 0; mb < MBE; ++mb) {
                                    const f32x4 x4 = *reinterpret_cast<const f32x4*>(xs + (mb * 16 + l15) * SS + kbl * 16 + 4 * lg);
#pragma unroll
                                    for (int q = 0; q < 4; ++q) acc[j][mb] = MFMA16(x4[q], w4[q], acc[j][mb]);
                                }
                            }
                        }
                    }
                }
            }
        }
        if (g.alphas) {
#pragma unroll
            for (int j = 0; j < NRBW; ++j)
#pragma unroll
                for (int mb = 0; mb < MBE; ++mb) acc[j][mb] = acc[j][mb] * sgV;
        }
        if (i > 0) {
#pragma unroll
            for (int j = 0; j < NRBW; ++j) {
                const int rb = wave + 4 * j;
                if (rb < nrb) {
                    for (int kb = 0; kb < nrb; ++kb) {
                        const f32x4 w4 = *reinterpret_cast<const f32x4*>(W + tile_addr(cd.seg_off[i][2], Rp, Rp, rb, kb) + lane * 4);
#pragma unroll
                        for (int mb = 0; mb < MBE; ++mb) {
                            const f32x4 x4 = *reinterpret_cast<const f32x4*>(xprev + (mb * 16 + l15) * SX + kb * 16 + 4 * lg);
#pragma unroll
                            for (int q = 0; q < 4; ++q) acc[j][mb] = MFMA16(x4[q], w4[q], acc[j][mb]);
                        }
                    }
                }
            }
            __syncthreads();   // every wave is done reading out_{i-1} before out_i overwrites it
        }
#pragma unroll
        for (int j = 0; j < NRBW; ++j) {
            const int rb = wave + 4 * j;
            if (rb < nrb) {
                const int r = rb * 16 + l15;
                const float bias = W[vb + VEC_B * Rp + r];
                float sc = 1.0f, sh = 0.0f, rm = 0.0f;
                if (g.bn) {
                    rm = W[vb + VEC_RM * Rp + r];
                    sc = 1.0f / sqrtf(W[vb + VEC_RV * Rp + r] + g.bn_eps);
                }
                const float gam = g.bn ? W[vb + VEC_G * Rp + r] : 1.0f;
                sh = g.bn ? W[vb + VEC_BE * Rp + r] : 0.0f;
#pragma unroll
                for (int mb = 0; mb < MBE; ++mb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int b = mb * 16 + 4 * lg + q;
                        float o = act_fwd(acc[j][mb][q] + bias, nl);
                        if (g.bn) o = ((o - rm) * sc) * gam + sh;
                        if (!(r < R)) o = 0.f;
                        xcur[b * SX + r] = o;
                    }
            }
        }
        __syncthreads();
    }
    {
        const float* xl = xo_l;
        for (int cb = wave; cb < ncb; cb += 4) {
            f32x4 hacc[MBE];
#pragma unroll
            for (int mb = 0; mb < MBE; ++mb) hacc[mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int kb = 0; kb < nrb; ++kb) {
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(W + tile_addr(cd.head_off, Cp, Rp, cb, kb) + lane * 4);
#pragma unroll
                for (int mb = 0; mb < MBE; ++mb) {
                    const f32x4 x4 = *reinterpret_cast<const f32x4*>(xl + (mb * 16 + l15) * SX + kb * 16 + 4 * lg);
#pragma unroll
                    for (int q = 0; q < 4; ++q) hacc[mb] = MFMA16(x4[q], w4[q], hacc[mb]);
                }
            }
            const int c = cb * 16 + l15;
            const float bias = W[cd.vec_off + g.vec_head + c];
#pragma unroll
            for (int mb = 0; mb < MBE; ++mb)
#pragma unroll
                for (int q = 0; q < 4; ++q) lg_l[(mb * 16 + 4 * lg + q) * SC + c] = hacc[mb][q] + bias;
        }
    }
    __syncthreads();
    if (a.logits) {
        for (int e = tid; e < nvalid * C; e += 256) {
            const int b = e / C, c = e - b * C;
            a.logits[(brow - a.row0 + b) * C + c] = lg_l[b * SC + c];
        }
    }
    if (tid < ME) {   // ME <= 64: exactly wave 0
        float loss = 0.f;
        long long corr = 0;
        if (tid < nvalid && g.loss_mode == 1) {
            // F1 'samples' (sklearn f1_score(average='samples')): per row 2|P&T| / (|P|+|T|), 0 when both are empty;
            // accumulated as 32.32 fixed point so that the sum is order-independent
            const float* row = lg_l + tid * SC;
            const int64_t grow = brow + tid;
            const float* z = a.tab.multilabel + grow * C;
            int tp = 0, np = 0, nt = 0;
            float ls = 0.f;
            for (int c = 0; c < C; ++c) {
                const float sg = 1.0f / (1.0f + expf(-row[c]));
                const bool pr = sg > g.f1_th, tr = z[c] > 0.5f;
                tp += (pr && tr) ? 1 : 0; np += pr ? 1 : 0; nt += tr ? 1 : 0;
                ls += a.pos_w[c] * z[c] * -logf(sg) + (1.0f - z[c]) * -logf(1.0f - sg);
            }
            loss = ls / (float)C;
            corr = (np + nt) > 0 ? (long long)((((unsigned long long)(2 * tp)) << 32) / (unsigned long long)(np + nt)) : 0;
        } else if (tid < nvalid) {
            const float* row = lg_l + tid * SC;
            const int64_t grow = brow + tid;
            const int lab = a.tab.label[grow];
            float mx = row[0];
            for (int c = 1; c < C; ++c) mx = fmaxf(mx, row[c]);
            float se = 0.f;
            for (int c = 0; c < C; ++c) se += expf(row[c] - mx);
            loss = -(row[lab] - mx - logf(se));
            int best = 0;
            float bv;
            if (g.multitask) {
                const float* vl = a.tab.vlogit + grow * C;
                const float* sl = a.tab.slogit + grow * C;
                bv = (row[0] + vl[0]) + sl[0];
                for (int c = 1; c < C; ++c) {
                    const float t = (row[c] + vl[c]) + sl[c];
                    if (t > bv) { bv = t; best = c; }
                }
            } else {
                bv = row[0];
                for (int c = 1; c < C; ++c)
                    if (row[c] > bv) { bv = row[c]; best = c; }
            }
            corr = best == lab ? 1 : 0;
        }
        for (int o = 32; o > 0; o >>= 1) {
            loss += __shfl_xor(loss, o);
            corr += __shfl_xor(corr, o);
        }
        if (tid == 0) {
            if (a.stats) {
                DevStats& st = a.stats[(int64_t)cand * a.E + a.epoch];
                atomicAdd(reinterpret_cast<unsigned long long*>(&st.dev_corr), (unsigned long long)corr);
                atomicAdd(&st.dev_loss, (double)loss);
            }
            if (a.corr_out) atomicAdd(reinterpret_cast<unsigned long long*>(a.corr_out), (unsigned long long)corr);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Parameter import / export / device init (tile-major <-> reference row-major state_dict order)
// ------------------------------------------------------------------------------------------------
#define PK_SET 0
#define PK_GET 1
#define PK_INIT 2

struct PackArgs {
    const SegDesc* desc;
    const CandDev* cands;
    float* plane;
    int64_t plane_stride;
    float* wt;
    float* flat;            // SET: source, GET: destination (one candidate)
    const uint32_t* seeds;  // INIT: per candidate
    int32_t mode, sel_plane;
    Geo g;
};

__device__ __forceinline__ uint32_t d_param_seed(uint32_t seed, uint32_t slot) {
    return (uint32_t)(((unsigned long long)seed * 1000003ULL + (unsigned long long)slot * 7919ULL + 17ULL) & 0x7FFFFFFFULL);
}
__device__ __forceinline__ uint32_t d_hash_h0(uint32_t seed) { return lowbias32(seed * 0x9E3779B9U + 0x7F4A7C15U); }
__device__ __forceinline__ float d_hash_u01(uint32_t h0, uint32_t idx) {
    return (float)(lowbias32(idx ^ h0) >> 8) * (1.0f / 16777216.0f);
}

__global__ void __launch_bounds__(256) k_pack(const PackArgs a) {
    const SegDesc d = a.desc[blockIdx.x];
    const int nkb = d.cc >> 4, nrb = d.rows_p >> 4;
    float* Wp = a.plane + d.w_off;
    uint32_t h0 = 0;
    if (a.mode == PK_INIT) h0 = d_hash_h0(d_param_seed(a.seeds[d.cand], d.init_seed));
    for (int e = threadIdx.x; e < d.rows_p * d.cc; e += 256) {
        const int tile = e >> 8, within = e & 255, lane = within >> 2, q = within & 3;
        const int rb = tile / nkb, kb = tile - rb * nkb;
        const int r = rb * 16 + (lane & 15);
        const int k = d.k0 + kb * 16 + 4 * (lane >> 4) + q;   // column inside the segment
        const bool ok = r < d.rows && k < d.cols;
        const int64_t fidx = (int64_t)r * d.src_ld + d.src_col0 + k;
        if (a.mode == PK_GET) {
            if (ok) a.flat[d.src_off + fidx] = Wp[a.sel_plane * a.plane_stride + e];
            continue;
        }
        float val = 0.f;
        if (ok) {
            if (a.mode == PK_SET) val = a.flat[d.src_off + fidx];
            else val = (d_hash_u01(h0, (uint32_t)fidx) * 2.0f - 1.0f) * d.init_bound;
        }
        Wp[e] = val;
        Wp[a.plane_stride + e] = 0.f;
        Wp[2 * a.plane_stride + e] = 0.f;
        if (d.wt_off >= 0) {
            const int l15 = lane & 15, lg = lane >> 4;
            float* T = a.wt + d.wt_off + ((int64_t)((d.k0 >> 4) + kb) * nrb + rb) * 256;
            T[((((l15 >> 2) * 16 + 4 * lg) + q) << 2) + (l15 & 3)] = val;
        }
    }
}

// vector parameters of one candidate (SET/GET) or of all candidates (INIT: blockIdx.x = candidate)
__global__ void __launch_bounds__(256) k_vec(const PackArgs a, int cand_fixed) {
    const int cand = cand_fixed >= 0 ? cand_fixed : blockIdx.x;
    const CandDev& cd = a.cands[cand];
    const Geo& g = a.g;
    float* P0 = a.plane + cd.vec_off;
    const int tid = threadIdx.x;
    const int nvec = MFAS_MAX_CELLS * g.vec_cell_stride + g.Cp;
    if (a.mode != PK_GET)
        for (int e = tid; e < nvec; e += 256) {   // zero everything first (padding, Adam state)
            P0[e] = 0.f;
            P0[a.plane_stride + e] = 0.f;
            P0[2 * a.plane_stride + e] = 0.f;
        }
    __syncthreads();
    float* P = P0 + (a.mode == PK_GET ? a.sel_plane * a.plane_stride : 0);
    const uint32_t seed = a.mode == PK_INIT ? a.seeds[cand] : 0;
    for (int i = 0; i < cd.L; ++i) {
        float* vb = P + i * g.vec_cell_stride;
        const float bound = (float)(1.0 / sqrt((double)cd.K_in[i]));
        const uint32_t hb = d_hash_h0(d_param_seed(seed, 2 * i + 1));
        for (int r = tid; r < g.R; r += 256) {
            if (a.mode == PK_SET) {
                vb[VEC_B * g.Rp + r] = a.flat[cd.f_b[i] + r];
                if (g.bn) {
                    vb[VEC_G * g.Rp + r] = a.flat[cd.f_bn[i] + r];
                    vb[VEC_BE * g.Rp + r] = a.flat[cd.f_bn[i] + g.R + r];
                    vb[VEC_RM * g.Rp + r] = a.flat[cd.f_bn[i] + 2 * g.R + r];
                    vb[VEC_RV * g.Rp + r] = a.flat[cd.f_bn[i] + 3 * g.R + r];
                }
            } else if (a.mode == PK_GET) {
                a.flat[cd.f_b[i] + r] = vb[VEC_B * g.Rp + r];
                if (g.bn) {
                    a.flat[cd.f_bn[i] + r] = vb[VEC_G * g.Rp + r];
                    a.flat[cd.f_bn[i] + g.R + r] = vb[VEC_BE * g.Rp + r];
                    // running stats exist only in plane 0
                    a.flat[cd.f_bn[i] + 2 * g.R + r] = a.sel_plane == 0 ? vb[VEC_RM * g.Rp + r] : 0.f;
                    a.flat[cd.f_bn[i] + 3 * g.R + r] = a.sel_plane == 0 ? vb[VEC_RV * g.Rp + r] : 0.f;
                }
            } else {
                vb[VEC_B * g.Rp + r] = (d_hash_u01(hb, (uint32_t)r) * 2.0f - 1.0f) * bound;
                if (g.bn) {
                    vb[VEC_G * g.Rp + r] = 1.0f;
                    vb[VEC_RV * g.Rp + r] = 1.0f;
                }
            }
        }
        if (tid == 0) {
            if (a.mode == PK_SET) vb[5 * g.Rp] = a.flat[cd.f_alpha + i];
            else if (a.mode == PK_GET) a.flat[cd.f_alpha + i] = vb[5 * g.Rp];
            else if (g.alphas) {
                const uint32_t ha = d_hash_h0(d_param_seed(seed, 40 + i));
                const float u0 = d_hash_u01(ha, 0), u1 = d_hash_u01(ha, 1), u2 = d_hash_u01(ha, 2), u3 = d_hash_u01(ha, 3);
                vb[5 * g.Rp] = ((((u0 + u1) + (u2 + u3)) - 2.0f) * 1.7320508075688772f) * 0.1f;
            }
        }
    }
    {
        float* hb_ = P + g.vec_head;
        const float bound = (float)(1.0 / sqrt((double)g.R));
        const uint32_t hh = d_hash_h0(d_param_seed(seed, 11));
        for (int c = tid; c < g.C; c += 256) {
            if (a.mode == PK_SET) hb_[c] = a.flat[cd.f_bc + c];
            else if (a.mode == PK_GET) a.flat[cd.f_bc + c] = hb_[c];
            else hb_[c] = (d_hash_u01(hh, (uint32_t)c) * 2.0f - 1.0f) * bound;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_pool — GlobalPooling2D (models/auxiliary/aux_models.py:54-64): mean over all trailing dims of a (B, C, ...) tap.
// One wave per (b, c) row of `inner` contiguous elements, 16 B per lane per load, f32 accumulation, wave shuffle
// reduction; pure HBM-bound reduction (the "step before the path" that builds the feature table).
// ------------------------------------------------------------------------------------------------
template <typename T> struct PoolVec;
template <> struct PoolVec<float> { static constexpr int N = 4; };
template <> struct PoolVec<uint16_t> { static constexpr int N = 8; };

__device__ __forceinline__ float pool_cvt(uint16_t v, int dtype) {
    return dtype == MFAS_DT_BF16 ? __uint_as_float((uint32_t)v << 16) : __half2float(__ushort_as_half(v));
}

__global__ void __launch_bounds__(256) k_pool(const void* x, int dtype, int64_t rows, int64_t inner, void* out, int out_dtype) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float acc = 0.f;
    if (dtype == MFAS_DT_F32) {
        const float* p = reinterpret_cast<const float*>(x) + row * inner;
        const int64_t nv = ((reinterpret_cast<uintptr_t>(p) & 15) == 0) ? inner / 4 : 0;
        for (int64_t i = lane; i < nv; i += 64) {
            const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p) + i);
            acc += (v[0] + v[1]) + (v[2] + v[3]);
        }
        for (int64_t i = nv * 4 + lane; i < inner; i += 64) acc += p[i];
    } else {
        const uint16_t* p = reinterpret_cast<const uint16_t*>(x) + row * inner;
        const int64_t nv = ((reinterpret_cast<uintptr_t>(p) & 15) == 0) ? inner / 8 : 0;
        for (int64_t i = lane; i < nv; i += 64) {
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 w = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p) + i);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc += pool_cvt((uint16_t)(w[j] & 0xFFFFU), dtype) + pool_cvt((uint16_t)(w[j] >> 16), dtype);
        }
        for (int64_t i = nv * 8 + lane; i < inner; i += 64) acc += pool_cvt(p[i], dtype);
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) {
        const float m = acc / (float)inner;
        if (out_dtype == MFAS_DT_F32) reinterpret_cast<float*>(out)[row] = m;
        else if (out_dtype == MFAS_DT_BF16) {
            uint32_t u = __float_as_uint(m);
            u += 0x7FFFU + ((u >> 16) & 1U);          // round to nearest even
            reinterpret_cast<uint16_t*>(out)[row] = (uint16_t)(u >> 16);
        } else reinterpret_cast<__half*>(out)[row] = __float2half(m);
    }
}

// ------------------------------------------------------------------------------------------------
// k_stream_probe — what this box's memory system gives the sweep's access pattern with NO compute: every wave
// read-modify-writes runs of 1 KiB tiles of three planes (16 B/lane, nontemporal), like W / m / v.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_stream_probe(float* P, size_t plane, size_t ntiles) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    constexpr int RUN = 8, U = 4;
    for (size_t r0 = wave * RUN; r0 < ntiles; r0 += nwaves * RUN)
        for (int t0 = 0; t0 < RUN && r0 + t0 < ntiles; t0 += U) {
            f32x4 w[U], m[U], v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t off = (r0 + t0 + u) * 256 + lane * 4;
                w[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(P + off));
                m[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(P + plane + off));
                v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(P + 2 * plane + off));
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t off = (r0 + t0 + u) * 256 + lane * 4;
                __builtin_nontemporal_store(w[u] * 0.999f + m[u] * 0.001f, reinterpret_cast<f32x4*>(P + off));
                __builtin_nontemporal_store(m[u] * 0.9f + v[u], reinterpret_cast<f32x4*>(P + plane + off));
                __builtin_nontemporal_store(v[u] * 0.999f + w[u], reinterpret_cast<f32x4*>(P + 2 * plane + off));
            }
        }
}

// ================================================================================================
// Host side: C ABI
// ================================================================================================
struct mfas_population {
    mfas_hyper hp;
    Geo g;
    int K = 0, device = 0;
    hipStream_t stream = nullptr;
    std::vector<CandDev> cands;
    std::vector<SegDesc> descs;
    std::vector<int> desc_start;    // K+1
    std::vector<int64_t> nparams;
    std::vector<int64_t> cand_plane_base, cand_plane_size;
    float* plane = nullptr;
    float* wt = nullptr;
    float* stepbuf = nullptr;
    float* best = nullptr;          // snapshot_best: copy of plane 0
    int64_t plane_stride = 0, wt_size = 0, step_total = 0;
    CandDev* d_cands = nullptr;
    SegDesc* d_descs = nullptr;
    struct Group { int c0 = 0, nc = 0, ndesc = 0, ntap = 0; SegDesc* d_descs = nullptr; TapDesc* d_taps = nullptr;
                   double alg_state = 0, alg_feat = 0; };
    std::vector<Group> groups;           // 1 or 2 contiguous candidate ranges, each with its own sweep work list
    DevStats* d_stats = nullptr;
    int32_t* d_status = nullptr;
    uint32_t* d_seeds = nullptr;
    long long* d_corr = nullptr;
    float* d_posw = nullptr;        // loss_mode 1: per-class positive weights (default 1)
    size_t lds_step = 0, lds_chain = 0, lds_eval = 0;
    bool vec_in_lds = false;
    int mbe = 4, nrbw = 1;
    bool yf_in_lds = false;
    bool lean_chain = false;        // chain_lean (R <= 16, C <= 64, B <= 32) in standalone and fused launches
    bool nontemporal = false;
    int stats_cap = 0;
    // profiling of the dominant kernel
    bool profiling = false;
    int prof_every = 16;            // HIP events bracket every prof_every-th sweep launch (event records are not free)
    double occ_bytes = 0;           // MB == 2: group state bytes/launch above which the sweep (not the chain) bounds a fused launch
    std::vector<hipEvent_t> ev;     // pairs
    int64_t prof_launches = 0;
    double prof_ms = 0.0, bytes_per_launch = 0.0, prof_bytes = 0.0;
    double alg_state_bytes = 0.0, alg_feat_elems = 0.0;
};

static inline int ceil16(int x) { return (x + 15) & ~15; }

extern "C" const char* mfas_last_error(void) { return g_err.c_str(); }
extern "C" int mfas_version(void) { return 100; }

static int pick_chunk(int cols_p, int target) {
    int best = 16;
    for (int c = 16; c <= cols_p && c <= target; c += 16)
        if (cols_p % c == 0) best = c;
    return best;
}

template <typename KT>
static hipError_t set_lds(KT kernel, size_t bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

extern "C" int mfas_population_create(const mfas_hyper* hp, const int32_t* confs, const int32_t* n_cells,
                                      const uint32_t* drop_seeds, int32_t K, int32_t device, void* hip_stream,
                                      int32_t chunk_cols, mfas_population** out) {
    if (!hp || !confs || !n_cells || !out || K <= 0) return fail(MFAS_EINVAL, "null argument or K <= 0");
    if (hp->R < 1 || hp->R > 512 || hp->C < 1 || hp->C > 256) return fail(MFAS_EINVAL, "R must be in [1,512], C in [1,256]");
    if (hp->B < 2 || hp->B > 64) return fail(MFAS_EINVAL, "batchsize must be in [2,64]");
    {
        const int bp = ((hp->B + 15) / 16 == 3 ? 4 : (hp->B + 15) / 16) * 16, lpr = std::min(16, 512 / bp);
        if (((hp->C + 15) & ~15) > 8 * lpr) return fail(MFAS_EINVAL, "num_outputs too large for this batch size (C_padded <= 8 * min(16, 512/B_padded))");
    }
    if (!(hp->drpt > 1e-10) && !hp->bn && !hp->allow_plain_cell)   // ntu_searchable.py:274-284: `op` never assigned
        return fail(MFAS_EINVAL, "illegal cell variant: drpt < 1e-10 without batchnorm (reference: UnboundLocalError)");
    if (hp->drpt >= 1.0) return fail(MFAS_EINVAL, "drpt must be < 1");
    if (hp->loss_mode == 1 && hp->multitask) return fail(MFAS_EINVAL, "multitask applies to the single-label head only");
    for (int j = 0; j < MFAS_MAX_TAPS; ++j)
        if (hp->s_sizes[j] < 0 || hp->v_sizes[j] < 0 || hp->s_sizes[j] > (1 << 20) || hp->v_sizes[j] > (1 << 20))
            return fail(MFAS_EINVAL, "tap widths must be in [0, 2^20]");
    mfas_population* p = new (std::nothrow) mfas_population();
    if (!p) return fail(MFAS_ENOMEM, "host alloc");
    p->hp = *hp;
    p->K = K;
    p->device = device;
    p->stream = reinterpret_cast<hipStream_t>(hip_stream);
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) { delete p; return fail(MFAS_EHIP, std::string("hipSetDevice: ") + hipGetErrorString(e)); }

    Geo& g = p->g;
    memset(&g, 0, sizeof(g));
    g.R = hp->R; g.C = hp->C; g.Rp = ceil16(hp->R); g.Cp = ceil16(hp->C);
    g.nrb = g.Rp / 16; g.ncb = g.Cp / 16; g.B = hp->B;
    g.MB = (hp->B + 15) / 16; if (g.MB == 3) g.MB = 4;
    g.Bp = g.MB * 16;
    g.bn = hp->bn != 0; g.alphas = hp->alphas != 0; g.multitask = hp->multitask != 0;
    g.use_drop = hp->drpt > 1e-10;
    g.drop_scale = g.use_drop ? (float)(1.0 / (1.0 - hp->drpt)) : 1.0f;
    g.drop_thr = g.use_drop ? (uint32_t)floor(hp->drpt * 16777216.0) : 0u;
    g.bn_eps = (float)hp->bn_eps; g.bn_mom = (float)hp->bn_momentum;
    g.vec_cell_stride = 5 * g.Rp + 16;
    g.vec_head = MFAS_MAX_CELLS * g.vec_cell_stride;
    for (int j = 0; j < MFAS_MAX_TAPS; ++j) { g.sw[j] = ceil16(hp->s_sizes[j]); g.vw[j] = ceil16(hp->v_sizes[j]); }
    g.loss_mode = hp->loss_mode == 1 ? 1 : 0;
    g.f1_th = (float)hp->f1_threshold;
    const int vec_size = (g.vec_head + g.Cp + 63) & ~63;

    // ---- column chunk per workgroup.  A workgroup should stream >= ~64 tiles (amortises staging / reduction and
    // keeps the number of partial-sum chunks the chain has to reduce small), the launch should still have a few
    // hundred workgroups, and x_t / x_{t+1} for the chunk must fit the LDS budget.
    int target = chunk_cols;
    if (target <= 0) {
        double tot_cols = 0;
        for (int k = 0; k < K; ++k)
            for (int i = 0; i < n_cells[k]; ++i)
                tot_cols += ceil16(hp->s_sizes[confs[(k * 4 + i) * 3] & 7]) + ceil16(hp->v_sizes[confs[(k * 4 + i) * 3 + 1] & 7]);
        int lds_max = 64;                                   // largest power of two with Bp*(2cc+20)*4 <= 72 KiB
        while ((size_t)g.Bp * (8 * lds_max + 20) * 4 <= 72 * 1024 && lds_max < 1024) lds_max <<= 1;   // test the doubled size
        target = 64;
        while (target * g.nrb < 64 * 16 && target < lds_max) target <<= 1;      // >= 64 tiles per workgroup
        while (target > 64 && tot_cols / target < 320.0) target >>= 1;           // ... but keep >= ~320 workgroups
        // R >= 128, measured on MI355X: 64-column chunks (finer, better-balanced workgroups) win once
        // the chain is hidden under the other group's sweep (K >= 24); below that fewer partial chunks matter more
        if (g.nrb >= 8) target = std::min(target, K >= 24 ? 64 : 256);   // (K >= 24 at R >= 128 <=> fused schedule)
    }
    target = std::max(16, (target / 16) * 16);
    p->cands.resize(K);
    p->desc_start.assign(K + 1, 0);
    p->nparams.resize(K);
    p->cand_plane_base.resize(K);
    p->cand_plane_size.resize(K);
    int64_t plane_off = 0, wt_off = 0, step_off = 0;
    double alg_bytes = 0.0, alg_feat = 0.0;
    int max_slots = 0;
    std::vector<int> slots(K);
    // first pass: layout
    for (int k = 0; k < K; ++k) {
        CandDev& c = p->cands[k];
        memset(&c, 0, sizeof(c));
        const int L = n_cells[k];
        if (L < 1 || L > MFAS_MAX_CELLS) { delete p; return fail(MFAS_EINVAL, "n_cells must be in [1,4]"); }
        c.L = L;
        c.drop_seed = drop_seeds ? drop_seeds[k] : (uint32_t)k;
        c.gidx = k;
        p->desc_start[k] = (int)p->descs.size();
        p->cand_plane_base[k] = plane_off;
        c.vec_off = plane_off;
        plane_off += vec_size;
        int64_t f = 0;
        c.f_alpha = f; f += L;
        int pslot = 0;
        for (int i = 0; i < L; ++i) {
            for (int j = 0; j < 3; ++j) {
                c.conf[i][j] = confs[(k * 4 + i) * 3 + j];
                c.seg_off[i][j] = -1;
            }
            if (c.conf[i][0] < 0 || c.conf[i][0] >= MFAS_MAX_TAPS || c.conf[i][1] < 0 || c.conf[i][1] >= MFAS_MAX_TAPS ||
                c.conf[i][2] < 0 || c.conf[i][2] > 2 || hp->s_sizes[c.conf[i][0]] < 1 || hp->v_sizes[c.conf[i][1]] < 1) {
                delete p; return fail(MFAS_EINVAL, "configuration entry out of range (tap index / unused tap slot / non-linearity)");
            }
            const int sw = hp->s_sizes[c.conf[i][0]], vw = hp->v_sizes[c.conf[i][1]];
            const int Kin = sw + vw + (i > 0 ? hp->R : 0);
            c.K_in[i] = Kin;
            c.f_W[i] = f; f += (int64_t)hp->R * Kin;
            c.f_b[i] = f; f += hp->R;
            c.f_bn[i] = f; if (hp->bn) f += 4 * (int64_t)hp->R;
            c.part_cell_off[i] = pslot;
            const float bound = (float)(1.0 / sqrt((double)Kin));
            const int widths[3] = {ceil16(sw), ceil16(vw), g.Rp};     // stored (padded) columns = table row stride
            const int true_w[3] = {sw, vw, hp->R};                     // reference columns
            const int col0[3] = {0, sw, sw + vw};
            for (int j = 0; j < 3; ++j) {
                if (j == 2 && i == 0) continue;
                const int cols_p = widths[j];
                const int cc = j < 2 ? pick_chunk(cols_p, target) : cols_p;
                const int nch = cols_p / cc;
                c.seg_off[i][j] = plane_off;
                c.seg_cc[i][j] = cc;
                c.seg_cols[i][j] = cols_p;
                if (j == 0) c.nch_s[i] = nch;
                if (j == 1) c.nch_v[i] = nch;
                if (j == 2) { c.outT_off[i] = wt_off; }
                for (int ch = 0; ch < nch; ++ch) {
                    SegDesc d;
                    memset(&d, 0, sizeof(d));
                    d.cand = k; d.kind = j; d.cell = i; d.tap = j < 2 ? c.conf[i][j] : 0;
                    d.k0 = ch * cc; d.cc = cc; d.rows_p = g.Rp; d.width = j < 2 ? widths[j] : g.Rp;
                    d.w_off = plane_off + (int64_t)ch * g.Rp * cc;
                    d.wt_off = j == 2 ? wt_off : -1;
                    d.part_idx = j < 2 ? (j == 0 ? ch : c.nch_s[i] + ch) : 0;
                    d.rows = hp->R; d.cols = true_w[j];
                    d.src_off = c.f_W[i]; d.src_ld = Kin; d.src_col0 = col0[j];
                    d.init_seed = 2 * i; d.init_bound = bound;
                    p->descs.push_back(d);
                }
                if (j < 2) pslot += nch;
                plane_off += (int64_t)g.Rp * cols_p;
                if (j == 2) wt_off += (int64_t)g.Rp * g.Rp;
                alg_bytes += 24.0 * hp->R * true_w[j];
                if (j < 2) alg_feat += (double)hp->B * true_w[j];   // x elements (dtype size applied at train time)
            }
        }
        c.f_Wc = f; f += (int64_t)hp->C * hp->R;
        c.f_bc = f; f += hp->C;
        p->nparams[k] = f;
        {   // head
            c.head_off = plane_off;
            c.headT_off = wt_off;
            SegDesc d;
            memset(&d, 0, sizeof(d));
            d.cand = k; d.kind = KIND_HEAD; d.cell = L - 1; d.tap = 0;
            d.k0 = 0; d.cc = g.Rp; d.rows_p = g.Cp; d.width = g.Rp;
            d.w_off = plane_off; d.wt_off = wt_off; d.part_idx = 0;
            d.rows = hp->C; d.cols = hp->R;
            d.src_off = c.f_Wc; d.src_ld = hp->R; d.src_col0 = 0;
            d.init_seed = 10; d.init_bound = (float)(1.0 / sqrt((double)hp->R));
            p->descs.push_back(d);
            plane_off += (int64_t)g.Cp * g.Rp;
            wt_off += (int64_t)g.Cp * g.Rp;
            alg_bytes += 24.0 * hp->C * hp->R;
        }
        plane_off = (plane_off + 63) & ~63LL;
        p->cand_plane_size[k] = plane_off - p->cand_plane_base[k];
        slots[k] = pslot;
        max_slots = std::max(max_slots, pslot);
    }
    p->desc_start[K] = (int)p->descs.size();
    // step buffers (same geometry for every candidate: sized for the largest)
    {
        const int64_t br = (int64_t)g.Bp * g.Rp;
        int64_t o = 0;
        g.sb_part = o; o += (int64_t)max_slots * br;
        g.sb_dy = o; o += MFAS_MAX_CELLS * br;
        g.sb_xo = o; o += MFAS_MAX_CELLS * br;
        g.sb_dlog = o; o += (int64_t)g.Bp * g.Cp;
        g.sb_sav = o; o += 3 * MFAS_MAX_CELLS * br;
        g.sb_yf = o; o += 2 * MFAS_MAX_CELLS * br;
        g.sb_gsc = o; o += 16;
        g.sb_size = (o + 63) & ~63LL;
        for (int k = 0; k < K; ++k) { p->cands[k].step_off = step_off; step_off += g.sb_size; }
    }
    p->plane_stride = plane_off;
    p->wt_size = wt_off;
    p->step_total = step_off;
    p->alg_state_bytes = alg_bytes;
    p->alg_feat_elems = alg_feat;
    p->bytes_per_launch = alg_bytes + 4.0 * alg_feat;

    // ---- LDS budgets
    {
        size_t ls = 0;
        for (const SegDesc& d : p->descs) {
            const int nrb = d.rows_p / 16;
            size_t fl = (size_t)g.Bp * (d.cc + 16) + (size_t)g.Bp * (d.cc + 4) + (size_t)g.Bp * (d.rows_p + 16);
            if (nrb < STEP_NW && d.kind <= KIND_V) fl += (size_t)STEP_NW * nrb * g.MB * 256;   // k-split reduction (forward only)
            ls = std::max(ls, fl * 4);
        }
        // chain: ping-pong activations + logits + misc (+ reduced feature sums when they fit next to the sweep's need)
        const size_t base = ((size_t)2 * g.Bp * (g.Rp + 4) + (size_t)g.Bp * (g.Cp + 4) + MFAS_MAX_CELLS * g.Rp + 3 * g.Bp + 16) * 4;
        const size_t yf = (size_t)(g.alphas ? 2 : 1) * MFAS_MAX_CELLS * g.nrb * g.MB * 256 * 4;
        p->yf_in_lds = base + yf <= std::max<size_t>(ls, 64 * 1024);
        p->lds_step = std::max(ls, base + (p->yf_in_lds ? yf : 0));
        const size_t vec = (size_t)3 * (MFAS_MAX_CELLS * g.vec_cell_stride + g.Cp) * 4;
        p->vec_in_lds = base + (p->yf_in_lds ? yf : 0) + vec <= 150 * 1024;
        p->lds_chain = base + (p->yf_in_lds ? yf : 0) + (p->vec_in_lds ? vec : 0);
        // chain_lean's LDS: out_i / dy_i of all cells, logits, misc, reduced sums, vector block, saved activations
        const size_t plane = (size_t)MFAS_MAX_CELLS * g.MB * 256;
        const size_t lean = ((size_t)2 * MFAS_MAX_CELLS * g.Bp * 20 + (size_t)g.Bp * (g.Cp + 4) + MFAS_MAX_CELLS * 16 + 3 * g.Bp + 16
                             + (g.alphas ? 2 : 1) * plane + vec / 4 + (1 + (g.bn ? 1 : 0) + (g.alphas ? 1 : 0)) * plane) * 4;
        p->lean_chain = g.nrb == 1 && g.ncb <= 4 && g.MB <= 2 && std::max(ls, lean) <= 72 * 1024 && !getenv("MFAS_NO_LEAN_CHAIN");
        if (p->lean_chain) { p->lds_chain = lean; p->lds_step = std::max(p->lds_step, lean); }
    }
    p->nrbw = (g.nrb + 3) / 4;
    if (p->nrbw == 3) p->nrbw = 4;
    for (p->mbe = 4; p->mbe >= 1; p->mbe >>= 1) {
        const int ME = p->mbe * 16;
        p->lds_eval = ((size_t)ME * std::max(EVAL_CE + 4, g.Cp + 4) + (size_t)ME * (g.Rp + 4)) * 4;
        if (p->lds_eval <= 80 * 1024) break;
    }
    if (p->mbe < 1 || p->nrbw > 8 || p->lds_step > 150 * 1024) {
        delete p;
        return fail(MFAS_EINVAL, "geometry does not fit the 160 KiB LDS (R / batchsize too large)");
    }

#define CREATE_CHK(x)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            std::string m_ = std::string(#x) + ": " + hipGetErrorString(e_);                       \
            mfas_population_destroy(p);                                                            \
            return fail(e_ == hipErrorOutOfMemory ? MFAS_ENOMEM : MFAS_EHIP, m_);                  \
        }                                                                                          \
    } while (0)
    CREATE_CHK(hipMalloc(&p->plane, sizeof(float) * 3 * (size_t)p->plane_stride));
    CREATE_CHK(hipMalloc(&p->wt, sizeof(float) * (size_t)std::max<int64_t>(p->wt_size, 64)));
    CREATE_CHK(hipMalloc(&p->stepbuf, sizeof(float) * (size_t)p->step_total));
    CREATE_CHK(hipMalloc(&p->d_cands, sizeof(CandDev) * K));
    CREATE_CHK(hipMalloc(&p->d_descs, sizeof(SegDesc) * p->descs.size()));
    CREATE_CHK(hipMalloc(&p->d_status, sizeof(int32_t) * (K + 128)));   // + debug timestamp slots (MFAS_CHAIN_TIMING builds)
    CREATE_CHK(hipMalloc(&p->d_seeds, sizeof(uint32_t) * K));
    CREATE_CHK(hipMalloc(&p->d_corr, sizeof(long long)));
    {
        std::vector<float> ones(g.Cp, 1.0f);
        CREATE_CHK(hipMalloc(&p->d_posw, sizeof(float) * g.Cp));
        CREATE_CHK(hipMemcpy(p->d_posw, ones.data(), sizeof(float) * g.Cp, hipMemcpyHostToDevice));
    }
    CREATE_CHK(hipMemcpy(p->d_cands, p->cands.data(), sizeof(CandDev) * K, hipMemcpyHostToDevice));
    CREATE_CHK(hipMemcpy(p->d_descs, p->descs.data(), sizeof(SegDesc) * p->descs.size(), hipMemcpyHostToDevice));
    {   // candidate groups: two halves balanced by work (descriptor columns), contiguous ranges
        // Two groups (the chain of one runs under the sweep of the other).  Measured on MI355X: pays from ~24 candidates
        // both at R=128 (a group's sweep then outlasts the ~50 us chain) and at R=16 (two half-size chain launches under
        // the sweeps beat chain + sweep back to back: 124 vs 105 cand/s at 50 candidates).
        int ngroups = K >= 24 ? 2 : 1;
        if (const char* e = getenv("MFAS_GROUPS")) ngroups = (atoi(e) >= 2 && K >= 2) ? 2 : 1;
        int split = K;
        if (ngroups == 2) {
            double tot = 0, run = 0;
            for (const SegDesc& d : p->descs) tot += (double)d.cc * d.rows_p;
            split = 1;
            for (int k = 0; k < K - 1; ++k) {
                for (int j = p->desc_start[k]; j < p->desc_start[k + 1]; ++j) run += (double)p->descs[j].cc * p->descs[j].rows_p;
                split = k + 1;
                if (run >= tot / 2) break;
            }
        }
        for (int gi = 0; gi < ngroups; ++gi) {
            mfas_population::Group gr;
            gr.c0 = gi == 0 ? 0 : split;
            gr.nc = gi == 0 ? split : K - split;
            std::vector<SegDesc> all(p->descs.begin() + p->desc_start[gr.c0], p->descs.begin() + p->desc_start[gr.c0 + gr.nc]);
            for (const SegDesc& d : all) {
                gr.alg_state += 24.0 * d.rows * std::max(0, std::min(d.cc, d.cols - d.k0));
                if (d.kind <= KIND_V) gr.alg_feat += (double)hp->B * d.cc;
            }
            // small R (1, 2 or 4 row blocks): feature segments are regrouped tap-major (sweep_tap_body)
            std::vector<SegDesc> sorted;
            std::vector<TapDesc> taps;
            const bool tap_major = (g.nrb == 1 || g.nrb == 2 || g.nrb == 4) && !getenv("MFAS_NO_TAP_MAJOR");
            if (tap_major) {
                const int per_wg = STEP_NW / g.nrb;
                std::vector<const SegDesc*> feat;
                for (const SegDesc& d : all) { if (d.kind <= KIND_V) feat.push_back(&d); else sorted.push_back(d); }
                std::stable_sort(feat.begin(), feat.end(), [](const SegDesc* x, const SegDesc* y) {
                    if (x->kind != y->kind) return x->kind < y->kind;
                    if (x->tap != y->tap) return x->tap < y->tap;
                    if (x->cc != y->cc) return x->cc < y->cc;
                    return x->k0 < y->k0;
                });
                for (size_t i0 = 0; i0 < feat.size();) {
                    TapDesc t;
                    memset(&t, 0, sizeof(t));
                    const SegDesc* f0 = feat[i0];
                    t.kind = f0->kind; t.tap = f0->tap; t.k0 = f0->k0; t.cc = f0->cc; t.rows_p = f0->rows_p; t.width = f0->width;
                    while (i0 < feat.size() && t.nitems < per_wg && feat[i0]->kind == t.kind && feat[i0]->tap == t.tap &&
                           feat[i0]->k0 == t.k0 && feat[i0]->cc == t.cc) {
                        t.cand[t.nitems] = feat[i0]->cand; t.cell[t.nitems] = feat[i0]->cell;
                        t.part_idx[t.nitems] = feat[i0]->part_idx; t.w_off[t.nitems] = feat[i0]->w_off;
                        ++t.nitems; ++i0;
                    }
                    taps.push_back(t);
                }
                std::stable_sort(taps.begin(), taps.end(), [](const TapDesc& x, const TapDesc& y) { return x.nitems * x.cc > y.nitems * y.cc; });
                if (taps.size() < 192 && !getenv("MFAS_FORCE_TAP_MAJOR")) {   // too few workgroups to fill 256 CUs: per-segment path
                    taps.clear();
                    sorted = all;
                }
            } else {
                sorted = all;
            }
            std::stable_sort(sorted.begin(), sorted.end(), [](const SegDesc& x, const SegDesc& y) { return x.cc * x.rows_p > y.cc * y.rows_p; });
            gr.ndesc = (int)sorted.size();
            gr.ntap = (int)taps.size();
            CREATE_CHK(hipMalloc(&gr.d_descs, sizeof(SegDesc) * std::max<size_t>(sorted.size(), 1)));
            CREATE_CHK(hipMemcpy(gr.d_descs, sorted.data(), sizeof(SegDesc) * sorted.size(), hipMemcpyHostToDevice));
            if (!taps.empty()) {
                CREATE_CHK(hipMalloc(&gr.d_taps, sizeof(TapDesc) * taps.size()));
                CREATE_CHK(hipMemcpy(gr.d_taps, taps.data(), sizeof(TapDesc) * taps.size(), hipMemcpyHostToDevice));
            }
            p->groups.push_back(gr);
        }
    }
    CREATE_CHK(hipMemsetAsync(p->plane, 0, sizeof(float) * 3 * (size_t)p->plane_stride, p->stream));
    CREATE_CHK(hipMemsetAsync(p->wt, 0, sizeof(float) * (size_t)std::max<int64_t>(p->wt_size, 64), p->stream));
    CREATE_CHK(hipMemsetAsync(p->stepbuf, 0, sizeof(float) * (size_t)p->step_total, p->stream));
    // measured crossover (MI355X, B=20): R=16 between 165 and 330 MB of group state per launch, R=128 between 300 and 600 MB
    // (the spilling chain of the occupancy build takes ~40 / ~125 us there)
    p->occ_bytes = g.nrb >= 8 ? 450e6 : 250e6;
    if (const char* e = getenv("MFAS_OCC_BYTES")) p->occ_bytes = atof(e);
#define SET_STEP(M, W, F) CREATE_CHK(set_lds((k_step<M, false, W, F>), p->lds_step)); CREATE_CHK(set_lds((k_step<M, true, W, F>), p->lds_step))
    SET_STEP(1, 4, false); SET_STEP(2, 2, false); SET_STEP(2, 4, false); SET_STEP(4, 2, false);
    SET_STEP(1, 4, true); SET_STEP(2, 2, true); SET_STEP(2, 4, true);
#undef SET_STEP
    CREATE_CHK(set_lds((k_chain<1, false>), p->lds_chain));
    CREATE_CHK(set_lds((k_chain<2, false>), p->lds_chain));
    CREATE_CHK(set_lds((k_chain<4, false>), p->lds_chain));
    CREATE_CHK(set_lds((k_chain<1, true>), p->lds_chain));
    CREATE_CHK(set_lds((k_chain<2, true>), p->lds_chain));
    // W/m/v beyond what the 256 MiB Infinity Cache can keep between steps are streamed nontemporally
    p->nontemporal = (double)p->plane_stride * 12.0 > 200.0 * 1024 * 1024;
    if (const char* e = getenv("MFAS_NT")) p->nontemporal = atoi(e) != 0;
    CREATE_CHK(hipStreamSynchronize(p->stream));
    *out = p;
    return MFAS_OK;
}

extern "C" void mfas_population_destroy(mfas_population* p) {
    if (!p) return;
    hipSetDevice(p->device);
    hipStreamSynchronize(p->stream);
    for (hipEvent_t e : p->ev) hipEventDestroy(e);
    hipFree(p->plane); hipFree(p->wt); hipFree(p->stepbuf); hipFree(p->best);
    for (auto& gr : p->groups) { hipFree(gr.d_descs); hipFree(gr.d_taps); }
    hipFree(p->d_cands); hipFree(p->d_descs); hipFree(p->d_stats); hipFree(p->d_status);
    hipFree(p->d_seeds); hipFree(p->d_corr); hipFree(p->d_posw);
    delete p;
}

extern "C" int64_t mfas_population_param_count(const mfas_population* p, int32_t k) {
    if (!p || k < 0 || k >= p->K) return fail(MFAS_EINVAL, "bad candidate index");
    return p->nparams[k];
}

static PackArgs pack_args(mfas_population* p, int mode, int plane, float* flat) {
    PackArgs a;
    memset(&a, 0, sizeof(a));
    a.desc = p->d_descs; a.cands = p->d_cands; a.plane = p->plane; a.plane_stride = p->plane_stride;
    a.wt = p->wt; a.flat = flat; a.seeds = p->d_seeds; a.mode = mode; a.sel_plane = plane; a.g = p->g;
    return a;
}

extern "C" int mfas_population_set_params(mfas_population* p, int32_t k, const float* flat) {
    if (!p || !flat || k < 0 || k >= p->K) return fail(MFAS_EINVAL, "bad argument");
    HIPCHK(hipSetDevice(p->device));
    PackArgs a = pack_args(p, PK_SET, 0, const_cast<float*>(flat));
    a.desc = p->d_descs + p->desc_start[k];
    const int n = p->desc_start[k + 1] - p->desc_start[k];
    hipLaunchKernelGGL(k_pack, dim3(n), dim3(256), 0, p->stream, a);
    hipLaunchKernelGGL(k_vec, dim3(1), dim3(256), 0, p->stream, a, (int)k);
    HIPCHK(hipGetLastError());
    return MFAS_OK;
}

extern "C" int mfas_population_get_params(mfas_population* p, int32_t k, int32_t plane, float* flat) {
    if (!p || !flat || k < 0 || k >= p->K || plane < 0 || plane > 2) return fail(MFAS_EINVAL, "bad argument");
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipMemsetAsync(flat, 0, sizeof(float) * p->nparams[k], p->stream));
    PackArgs a = pack_args(p, PK_GET, plane, flat);
    a.desc = p->d_descs + p->desc_start[k];
    const int n = p->desc_start[k + 1] - p->desc_start[k];
    hipLaunchKernelGGL(k_pack, dim3(n), dim3(256), 0, p->stream, a);
    hipLaunchKernelGGL(k_vec, dim3(1), dim3(256), 0, p->stream, a, (int)k);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(p->stream));
    return MFAS_OK;
}

extern "C" int mfas_population_init(mfas_population* p, const uint32_t* seeds) {
    if (!p || !seeds) return fail(MFAS_EINVAL, "bad argument");
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipMemcpyAsync(p->d_seeds, seeds, sizeof(uint32_t) * p->K, hipMemcpyHostToDevice, p->stream));
    PackArgs a = pack_args(p, PK_INIT, 0, nullptr);
    hipLaunchKernelGGL(k_pack, dim3((unsigned)p->descs.size()), dim3(256), 0, p->stream, a);
    hipLaunchKernelGGL(k_vec, dim3(p->K), dim3(256), 0, p->stream, a, -1);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(p->stream));   // seeds is a host buffer
    return MFAS_OK;
}

static int check_table(const mfas_population* p, const mfas_table* t, bool need_logits) {
    if (!t || t->N <= 0) return fail(MFAS_EINVAL, "table: null or empty");
    if (p->g.loss_mode == 0 && !t->label) return fail(MFAS_EINVAL, "table: labels missing");
    if (p->g.loss_mode == 1 && !t->multilabel) return fail(MFAS_EINVAL, "table: multi-hot targets missing (loss_mode 1)");
    if (t->dtype < 0 || t->dtype > 2) return fail(MFAS_EINVAL, "table: bad dtype");
    for (int j = 0; j < 4; ++j)
        if (!t->s[j] || !t->v[j]) return fail(MFAS_EINVAL, "table: null tap pointer");
    if (need_logits && (!t->vlogit || !t->slogit)) return fail(MFAS_EINVAL, "multitask needs vlogit/slogit");
    return MFAS_OK;
}

template <int MBE, int NRBW>
static hipError_t launch_eval_t(mfas_population* p, const EvalArgs& a, int ncand, hipStream_t st) {
    hipError_t e = set_lds(k_eval<MBE, NRBW>, p->lds_eval);
    if (e != hipSuccess) return e;
    const int ME = MBE * 16;
    const unsigned nblk = (unsigned)((a.nrows + ME - 1) / ME);
    hipLaunchKernelGGL((k_eval<MBE, NRBW>), dim3(nblk, ncand), dim3(256), p->lds_eval, st, a);
    return hipGetLastError();
}

static hipError_t launch_eval(mfas_population* p, const EvalArgs& a, int ncand, hipStream_t st) {
#define EV_CASE(M, N) if (p->mbe == M && p->nrbw == N) return launch_eval_t<M, N>(p, a, ncand, st);
    EV_CASE(4, 1) EV_CASE(4, 2) EV_CASE(4, 4) EV_CASE(4, 8)
    EV_CASE(2, 1) EV_CASE(2, 2) EV_CASE(2, 4) EV_CASE(2, 8)
    EV_CASE(1, 1) EV_CASE(1, 2) EV_CASE(1, 4) EV_CASE(1, 8)
#undef EV_CASE
    return hipErrorInvalidValue;
}

extern "C" int mfas_population_train(mfas_population* p, const mfas_table* train, const mfas_table* dev,
                                     const int32_t* order, const float* step_scalars, int32_t epochs,
                                     int64_t max_steps, int32_t snapshot_best, mfas_epoch_stats* stats,
                                     int32_t* status) {
    if (!p || !step_scalars || epochs <= 0 || !stats) return fail(MFAS_EINVAL, "bad argument");
    int rc = check_table(p, train, p->g.multitask);
    if (rc) return rc;
    const bool do_dev = max_steps < 0;
    if (do_dev) { rc = check_table(p, dev, p->g.multitask); if (rc) return rc; }
    HIPCHK(hipSetDevice(p->device));
    const Geo& g = p->g;
    const int K = p->K, B = g.B;
    const int64_t N = train->N;
    const int64_t nb = (N + B - 1) / B;
    if (N - (nb - 1) * B == 1 && g.bn)   // torch BatchNorm1d raises on a size-1 train batch
        return fail(MFAS_EINVAL, "final train batch of size 1 with batchnorm (reference raises ValueError)");

    if (p->stats_cap < K * epochs) {
        hipFree(p->d_stats); p->d_stats = nullptr;
        HIPCHK(hipMalloc(&p->d_stats, sizeof(DevStats) * K * epochs));
        p->stats_cap = K * epochs;
    }
    HIPCHK(hipMemsetAsync(p->d_stats, 0, sizeof(DevStats) * K * epochs, p->stream));
    HIPCHK(hipMemsetAsync(p->d_status, 0, sizeof(int32_t) * K, p->stream));
    // every call is a freshly built torch.optim.Adam (ntu_searchable.py:65; main_found_ntu.py:108,128): zero exp_avg / exp_avg_sq
    HIPCHK(hipMemsetAsync(p->plane + p->plane_stride, 0, sizeof(float) * 2 * (size_t)p->plane_stride, p->stream));
    if (snapshot_best && !p->best) HIPCHK(hipMalloc(&p->best, sizeof(float) * (size_t)p->plane_stride));
    std::vector<double> best_acc(K, 0.0);
    std::vector<DevStats> hstats((size_t)K * epochs);

    const mfas_hyper& hp = p->hp;
    AdamC ac;
    ac.w1 = (float)(1.0 - hp.beta1); ac.b2 = (float)hp.beta2; ac.w2 = (float)(1.0 - hp.beta2);
    ac.eps = (float)hp.adam_eps; ac.wd = (float)hp.wd; ac.ss = 0.f; ac.bc2s = 1.f;

    // Candidate groups A/B: every launch pairs the sweep of one group with the chain of the other (k_step).
    const int NG = (int)p->groups.size();
    StepArgs st;
    memset(&st, 0, sizeof(st));
    st.sa.cands = p->d_cands; st.sa.plane = p->plane; st.sa.plane_stride = p->plane_stride; st.sa.wt = p->wt;
    st.sa.stepbuf = p->stepbuf; st.sa.tab = *train; st.sa.order = order; st.sa.g = g; st.sa.ac = ac;
    st.ca.plane = p->plane; st.ca.plane_stride = p->plane_stride; st.ca.wt = p->wt; st.ca.stepbuf = p->stepbuf;
    st.ca.tab = *train; st.ca.order = order; st.ca.E = epochs; st.ca.g = g; st.ca.stats = p->d_stats;
    st.ca.status = p->d_status; st.ca.ac = ac; st.ca.yf_in_lds = p->yf_in_lds ? 1 : 0; st.ca.pos_w = p->d_posw;
    st.ca.vec_in_lds = p->vec_in_lds ? 1 : 0;

    const int elt = train->dtype == MFAS_DT_F32 ? 4 : 2;
    p->prof_launches = 0; p->prof_ms = 0.0; p->prof_bytes = 0.0;
    size_t ev_used = 0;
    std::vector<double> ev_bytes;
    int64_t nlaunch = 0;

    // one fused launch: sweep of group gs at step ts (gs < 0: none) + chain of group gc at step tc (gc < 0: none)
    auto step = [&](int gs, int upd, int fwd, int64_t ep, int64_t ts, int gc, int64_t tc) {
        unsigned nsw = 0, nch = 0;
        if (gs >= 0) {
            SweepArgs& s = st.sa;
            s.desc = p->groups[gs].d_descs;
            s.tdesc = p->groups[gs].d_taps; s.ntap = p->groups[gs].ntap;
            s.do_update = upd; s.do_forward = fwd;
            s.pos_t = ep * N + ts * B; s.base_t = (int)(ts * B);
            s.nvalid_t = (int)std::min<int64_t>(B, N - ts * B);
            const int64_t tn = fwd ? (upd ? ts + 1 : ts) : ts;
            s.pos_n = ep * N + tn * B; s.base_n = (int)(tn * B);
            s.nvalid_n = (int)std::min<int64_t>(B, N - tn * B);
            const int64_t gstep = ep * nb + ts;
            s.ac.ss = upd ? step_scalars[2 * gstep] : 0.f;
            s.ac.bc2s = upd ? step_scalars[2 * gstep + 1] : 1.f;
            nsw = (unsigned)(p->groups[gs].ndesc + p->groups[gs].ntap);
        }
        if (gc >= 0) {
            ChainArgs& c = st.ca;
            c.cands = p->d_cands + p->groups[gc].c0;
            c.pos_t = ep * N + tc * B; c.base_t = (int)(tc * B);
            c.nvalid = (int)std::min<int64_t>(B, N - tc * B);
            const int64_t gstep = ep * nb + tc;
            c.gstep = (int)gstep; c.epoch = (int)ep;
            c.ac.ss = step_scalars[2 * gstep]; c.ac.bc2s = step_scalars[2 * gstep + 1];
            nch = (unsigned)p->groups[gc].nc;
        }
        st.nchain = (int)nch;
        if (gs < 0) { st.sa.ntap = 0; }
        const bool prof = p->profiling && gs >= 0 && upd && fwd && ((nlaunch++ % p->prof_every) == 0);
        if (prof) {
            if (p->ev.size() < ev_used + 2) {
                hipEvent_t e0, e1;
                hipEventCreate(&e0); hipEventCreate(&e1);
                p->ev.push_back(e0); p->ev.push_back(e1);
            }
            hipEventRecord(p->ev[ev_used], p->stream);
        }
        if (nsw == 0) {   // chain only: the latency-tuned standalone kernel
#define CHAIN_LAUNCH(M, F) hipLaunchKernelGGL((k_chain<M, F>), dim3(nch), dim3(STEP_THREADS), p->lds_chain, p->stream, st.ca)
            if (p->lean_chain) { if (g.MB == 1) CHAIN_LAUNCH(1, true); else CHAIN_LAUNCH(2, true); }
            else if (g.MB == 1) CHAIN_LAUNCH(1, false);
            else if (g.MB == 2) CHAIN_LAUNCH(2, false);
            else CHAIN_LAUNCH(4, false);
#undef CHAIN_LAUNCH
            return;
        }
#define STEP_LAUNCH(M, T, W, F) hipLaunchKernelGGL((k_step<M, T, W, F>), dim3(nch + nsw), dim3(STEP_THREADS), p->lds_step, p->stream, st)
#define STEP_PICK(M, W) do { if (p->nontemporal) { if (p->lean_chain) STEP_LAUNCH(M, true, W, true); else STEP_LAUNCH(M, true, W, false); } \
                             else { if (p->lean_chain) STEP_LAUNCH(M, false, W, true); else STEP_LAUNCH(M, false, W, false); } } while (0)
        // MB == 2: the two-workgroups-per-CU build unless a co-scheduled chain would bound the launch (see SweepU)
        const bool occ = nch == 0 || p->groups[gs].alg_state > p->occ_bytes;
        if (g.MB == 1) STEP_PICK(1, 4);
        else if (g.MB == 2) { if (occ || p->lean_chain) STEP_PICK(2, 4); else STEP_PICK(2, 2); }   // chain_lean never spills
        else { if (p->nontemporal) STEP_LAUNCH(4, true, 2, false); else STEP_LAUNCH(4, false, 2, false); }
#undef STEP_PICK
#undef STEP_LAUNCH
        if (prof) {
            hipEventRecord(p->ev[ev_used + 1], p->stream);
            ev_used += 2;
            // algorithmic HBM bytes of this group's update+forward sweep: 24 B/param + the batch's taps + labels
            ev_bytes.push_back(p->groups[gs].alg_state + p->groups[gs].alg_feat * elt + 8.0 * B * p->groups[gs].nc);
        }
    };

    int64_t done = 0;   // train steps completed (max_steps bookkeeping)
    for (int ep = 0; ep < epochs; ++ep) {
        int64_t T = nb;
        if (max_steps >= 0) T = std::min<int64_t>(nb, max_steps - done);
        if (T <= 0) break;
        for (int gi = 0; gi < NG; ++gi) step(gi, 0, 1, ep, 0, -1, 0);   // prologue: forward sums of batch 0
        if (NG == 1) {
            for (int64_t t = 0; t < T; ++t) {
                step(-1, 0, 0, ep, 0, 0, t);
                step(0, 1, (t + 1 < T) ? 1 : 0, ep, t, -1, 0);
            }
        } else {
            step(-1, 0, 0, ep, 0, 0, 0);   // chain(A, 0)
            for (int64_t t = 0; t < T; ++t) {
                const int fwd = (t + 1 < T) ? 1 : 0;
                step(0, 1, fwd, ep, t, 1, t);                       // sweep(A, t)  ||  chain(B, t)
                step(1, 1, fwd, ep, t, fwd ? 0 : -1, t + 1);        // sweep(B, t)  ||  chain(A, t+1)
            }
        }
        done += T;
        HIPCHK(hipGetLastError());
        if (do_dev) {
            EvalArgs ea;
            memset(&ea, 0, sizeof(ea));
            ea.cands = p->d_cands; ea.plane = p->plane; ea.tab = *dev; ea.row0 = 0; ea.nrows = dev->N;
            ea.cand0 = 0; ea.epoch = ep; ea.E = epochs; ea.g = g; ea.stats = p->d_stats; ea.pos_w = p->d_posw;
            HIPCHK(launch_eval(p, ea, K, p->stream));
            if (snapshot_best) {
                HIPCHK(hipMemcpyAsync(hstats.data(), p->d_stats, sizeof(DevStats) * K * epochs, hipMemcpyDeviceToHost, p->stream));
                HIPCHK(hipStreamSynchronize(p->stream));
                for (int k = 0; k < K; ++k) {
                    const double acc = (double)hstats[(size_t)k * epochs + ep].dev_corr / (double)dev->N;
                    if (acc > best_acc[k]) {   // strict >, from 0 (train_searchable/ntu.py:82)
                        best_acc[k] = acc;
                        HIPCHK(hipMemcpyAsync(p->best + p->cand_plane_base[k], p->plane + p->cand_plane_base[k],
                                              sizeof(float) * p->cand_plane_size[k], hipMemcpyDeviceToDevice, p->stream));
                    }
                }
            }
        }
    }
    if (snapshot_best && do_dev) {
        for (int k = 0; k < K; ++k)
            if (best_acc[k] > 0.0)
                HIPCHK(hipMemcpyAsync(p->plane + p->cand_plane_base[k], p->best + p->cand_plane_base[k],
                                      sizeof(float) * p->cand_plane_size[k], hipMemcpyDeviceToDevice, p->stream));
    }
    HIPCHK(hipMemcpyAsync(hstats.data(), p->d_stats, sizeof(DevStats) * K * epochs, hipMemcpyDeviceToHost, p->stream));
    std::vector<int32_t> hstatus(K, 0);
    HIPCHK(hipMemcpyAsync(hstatus.data(), p->d_status, sizeof(int32_t) * K, hipMemcpyDeviceToHost, p->stream));
    HIPCHK(hipStreamSynchronize(p->stream));
    HIPCHK(hipGetLastError());
    for (size_t i = 0; i < hstats.size(); ++i) {
        stats[i].train_loss_sum = hstats[i].train_loss;
        stats[i].dev_loss_sum = hstats[i].dev_loss;
        stats[i].train_corrects = hstats[i].train_corr;
        stats[i].dev_corrects = hstats[i].dev_corr;
    }
    if (status) memcpy(status, hstatus.data(), sizeof(int32_t) * K);
#ifdef MFAS_CHAIN_TIMING
    {
        int32_t ts[16];
        if (hipMemcpy(ts, p->d_status + 64, sizeof(ts), hipMemcpyDeviceToHost) == hipSuccess) {
            fprintf(stderr, "[chain timing, shader cycles since kernel entry, candidate 0 step 3]");
            for (int i = 0; i < 13; ++i) fprintf(stderr, " %d", ts[i]);
            fprintf(stderr, "\n");
        }
    }
#endif
    if (p->profiling) {
        for (size_t i = 0; i + 1 < ev_used; i += 2) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, p->ev[i], p->ev[i + 1]) == hipSuccess) {
                p->prof_ms += ms; p->prof_launches++; p->prof_bytes += ev_bytes[i / 2];
            }
        }
        p->bytes_per_launch = p->prof_launches ? p->prof_bytes / p->prof_launches : 0.0;
    }
    return MFAS_OK;
}

extern "C" int mfas_population_forward(mfas_population* p, int32_t k, const mfas_table* tab, int64_t row0,
                                       int64_t nrows, float* logits, int64_t* corrects) {
    if (!p || k < 0 || k >= p->K || nrows <= 0 || row0 < 0) return fail(MFAS_EINVAL, "bad argument");
    int rc = check_table(p, tab, p->g.multitask);
    if (rc) return rc;
    if (row0 + nrows > tab->N) return fail(MFAS_EINVAL, "row range outside the table");
    HIPCHK(hipSetDevice(p->device));
    EvalArgs ea;
    memset(&ea, 0, sizeof(ea));
    ea.cands = p->d_cands; ea.plane = p->plane; ea.tab = *tab; ea.row0 = row0; ea.nrows = nrows;
    ea.cand0 = k; ea.epoch = 0; ea.E = 1; ea.g = p->g; ea.logits = logits; ea.pos_w = p->d_posw;
    if (corrects) {
        HIPCHK(hipMemsetAsync(p->d_corr, 0, sizeof(long long), p->stream));
        ea.corr_out = p->d_corr;
    }
    HIPCHK(launch_eval(p, ea, 1, p->stream));
    if (corrects) {
        long long h = 0;
        HIPCHK(hipMemcpyAsync(&h, p->d_corr, sizeof(long long), hipMemcpyDeviceToHost, p->stream));
        HIPCHK(hipStreamSynchronize(p->stream));
        *corrects = (int64_t)h;
    }
    return MFAS_OK;
}

extern "C" int mfas_stream_probe(int64_t bytes_per_plane, int32_t iters, double* gb_per_s) {
    if (bytes_per_plane < (1 << 20) || iters < 1 || !gb_per_s) return fail(MFAS_EINVAL, "bad argument");
    const size_t plane = (size_t)bytes_per_plane / 1024 * 256;   // floats, whole tiles
    float* P = nullptr;
    HIPCHK(hipMalloc(&P, plane * 4 * 3));
    hipError_t e = hipMemset(P, 0, plane * 4 * 3);
    hipEvent_t a, b;
    if (e == hipSuccess) e = hipEventCreate(&a);
    if (e == hipSuccess) e = hipEventCreate(&b);
    if (e != hipSuccess) { hipFree(P); return fail(MFAS_EHIP, hipGetErrorString(e)); }
    hipLaunchKernelGGL(k_stream_probe, dim3(2048), dim3(256), 0, 0, P, plane, plane / 256);
    hipEventRecord(a, 0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k_stream_probe, dim3(2048), dim3(256), 0, 0, P, plane, plane / 256);
    hipEventRecord(b, 0);
    e = hipEventSynchronize(b);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a); hipEventDestroy(b); hipFree(P);
    if (e != hipSuccess) return fail(MFAS_EHIP, hipGetErrorString(e));
    *gb_per_s = (double)plane * 4 * 3 * 2 * iters / 1e9 / (ms * 1e-3);
    return MFAS_OK;
}

extern "C" int mfas_global_pool(const void* x, int32_t dtype, int64_t rows, int64_t inner, void* out, int32_t out_dtype,
                                void* hip_stream) {
    if (!x || !out || rows <= 0 || inner <= 0 || dtype < 0 || dtype > 2 || out_dtype < 0 || out_dtype > 2)
        return fail(MFAS_EINVAL, "bad argument");
    const int64_t nblk = (rows + 3) / 4;
    if (nblk > 0x7FFFFFFF) return fail(MFAS_EINVAL, "too many rows");
    hipLaunchKernelGGL(k_pool, dim3((unsigned)nblk), dim3(256), 0, reinterpret_cast<hipStream_t>(hip_stream), x, (int)dtype,
                       rows, inner, out, (int)out_dtype);
    HIPCHK(hipGetLastError());
    return MFAS_OK;
}

extern "C" int mfas_population_set_pos_weight(mfas_population* p, const float* w) {
    if (!p || !w) return fail(MFAS_EINVAL, "null");
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipMemcpy(p->d_posw, w, sizeof(float) * p->g.C, hipMemcpyHostToDevice));
    return MFAS_OK;
}

extern "C" int mfas_population_set_profiling(mfas_population* p, int32_t on) {
    if (!p) return fail(MFAS_EINVAL, "null");
    p->profiling = on != 0;
    if (const char* e = getenv("MFAS_PROF_EVERY")) p->prof_every = std::max(1, atoi(e));
    return MFAS_OK;
}

extern "C" int mfas_population_sweep_profile(const mfas_population* p, int64_t* launches, double* total_ms,
                                             double* bytes_per_launch) {
    if (!p) return fail(MFAS_EINVAL, "null");
    if (launches) *launches = p->prof_launches;
    if (total_ms) *total_ms = p->prof_ms;
    if (bytes_per_launch) *bytes_per_launch = p->bytes_per_launch;
    return MFAS_OK;
}

// chain.hip.h — the serial train-step chain of one candidate: general (chain_body) and latency-lean small-R (chain_lean) forms
// (part of the single translation unit mfas_hip.hip; see the header comment there and DESIGN.md)
#pragma once
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
    // same-group fused launch (k_step_same): publish cellflag[candidate][i] = flag_target as soon as dy_i (slot 4: dlogits) is out
    uint32_t* cellflag;
    uint32_t flag_target, _padf;
    int32_t yf_reduced, _padr;   // the sweep already reduced the partial sums into the step buffer's yf area (sweep.hip.h)
    float* logits_out;           // train-mode FORWARD ONLY (mfas_population_forward_train): write the batch's logits (nvalid x C)
                                 // after the head and stop — batch-statistics BN (running stats updated), dropout stream of `gstep`
    const float* dlogits_in;     // BACKWARD OF AN EXTERNAL LOSS (mfas_population_backward): dL/dlogits (nvalid x C) given by the caller
                                 // takes the place of the softmax / BCE gradient; no statistics are accumulated
};

// what changes from one train step to the next (k_step / k_chain take it from the launch arguments, the persistent loop
// computes it from its step counter)
struct ChainStep {
    int64_t pos_t;
    int32_t base_t, nvalid, gstep, epoch;
    float ss, bc2s;
};
__device__ __forceinline__ ChainStep chain_step_of(const ChainArgs& a) {
    ChainStep c;
    c.pos_t = a.pos_t; c.base_t = a.base_t; c.nvalid = a.nvalid; c.gstep = a.gstep; c.epoch = a.epoch;
    c.ss = a.ac.ss; c.bc2s = a.ac.bc2s;
    return c;
}

#define CHAIN_NW STEP_NW
#define CHAIN_THREADS STEP_THREADS

__device__ __forceinline__ bool drop_keep(uint32_t h0, int cell, uint32_t idx, uint32_t thr) {
    // oracle/np_oracle.py:dropout_keep
    const uint32_t key = idx + (uint32_t)cell * 0x7F4A7C15U;
    return (lowbias32(key ^ h0) >> 8) >= thr;
}

// acc[mb] += X[b][0..16*nk) . tile(k)   (X in LDS row-major with stride sx; tiles: 256 floats each, stride tstride)
// TW = weight tiles requested before the first is consumed (a batch of TW x 4 VGPRs; even TW keeps the even / odd pairing, so
// the arithmetic does not depend on it): 8, or 4 in the 128-VGPR builds of the two-batch-block chain, which spilled 15-31 registers
template <int MB, bool COH, int TW = 8>
__device__ __forceinline__ void lds_x_times_tiles(f32x4 (&acc)[MB], const float* X, int sx, const float* tbase, int64_t tidx,
                                                  int64_t tstride, int nk, int lane) {
    // same arithmetic as mma_tiles: even / odd k-blocks in two independent chains, summed at the end
    static_assert(TW % 2 == 0, "even / odd chains are paired by the position inside a batch");
    const int l15 = lane & 15, lg = lane >> 4;
    f32x4 acc2[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc2[mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < nk; k0 += TW) {
        f32x4 w8[TW];
#pragma unroll
        for (int u = 0; u < TW; ++u)
            if (k0 + u < nk) w8[u] = ldc4<COH>(tbase, tidx + (int64_t)(k0 + u) * tstride + lane * 4);
#pragma unroll
        for (int u = 0; u < TW; ++u)
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
template <bool COH>
__device__ __forceinline__ void issue_tiles(f32x4 (&w8)[8], const float* tbase, int64_t tidx, int nk, int lane) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
        if (u < nk) w8[u] = ldc4<COH>(tbase, tidx + (int64_t)u * 256 + lane * 4);
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

// Reductions over the LPR (<= 16) lanes that share a batch row, on the DPP cross-lane path (one VALU op per stage instead of
// an LDS-crossbar ds_bpermute per __shfl_xor): stage 1 / 2 swap inside quads (quad_perm), stage 4 mirrors each half row
// (lane i <-> 7 - i), stage 8 mirrors the row (i <-> 15 - i).  Every lane ends with the reduction over its 16 (8, 4, ...)
// lanes; the operations are commutative, the pairing (hence the float summation order) is fixed.
template <int STAGE>
__device__ __forceinline__ int dpp_stage_i(int v) {
    if constexpr (STAGE == 1) return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);        // quad_perm [1,0,3,2]
    else if constexpr (STAGE == 2) return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    else if constexpr (STAGE == 4) return __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);  // row_half_mirror
    else return __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false);                            // row_mirror
}
template <int STAGE> __device__ __forceinline__ float dpp_stage_f(float v) { return __int_as_float(dpp_stage_i<STAGE>(__float_as_int(v))); }
template <int LPR> __device__ __forceinline__ float row_max(float v) {
    v = fmaxf(v, dpp_stage_f<1>(v));
    if constexpr (LPR > 2) v = fmaxf(v, dpp_stage_f<2>(v));
    if constexpr (LPR > 4) v = fmaxf(v, dpp_stage_f<4>(v));
    if constexpr (LPR > 8) v = fmaxf(v, dpp_stage_f<8>(v));
    return v;
}
template <int LPR> __device__ __forceinline__ float row_sum(float v) {
    v += dpp_stage_f<1>(v);
    if constexpr (LPR > 2) v += dpp_stage_f<2>(v);
    if constexpr (LPR > 4) v += dpp_stage_f<4>(v);
    if constexpr (LPR > 8) v += dpp_stage_f<8>(v);
    return v;
}
// first maximum (smallest class index on ties) over the row's lanes
template <int STAGE> __device__ __forceinline__ void argmax_stage(float& bv, int& bi) {
    const float pv = dpp_stage_f<STAGE>(bv);
    const int pi = dpp_stage_i<STAGE>(bi);
    if (pv > bv || (pv == bv && pi < bi)) { bv = pv; bi = pi; }
}
template <int LPR> __device__ __forceinline__ void row_argmax(float& bv, int& bi) {
    argmax_stage<1>(bv, bi);
    if constexpr (LPR > 2) argmax_stage<2>(bv, bi);
    if constexpr (LPR > 4) argmax_stage<4>(bv, bi);
    if constexpr (LPR > 8) argmax_stage<8>(bv, bi);
}

// Softmax cross-entropy on the LDS logits (train_searchable/ntu.py:53-61), LPR lanes per batch row: classes c = sub,
// sub+LPR, ... (<= 8 classes per lane, exp kept).  Leaves dlogits = (softmax - onehot)/nvalid in place, the row's loss in
// red[b] and its top-1 hit in red[Bp + b] (multitask: argmax of central + visual + skeleton logits).
template <int MB, int NC>
__device__ __forceinline__ void softmax_rows_nc(const ChainArgs& a, const ChainStep& cs, float* lg_l, const int SC, float* red_l,
                                                const int* lab_l, const int nvalid, const float nf, const int tid, const int32_t* ord) {
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
    mx = row_max<LPR>(mx);
    float se = 0.f;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int c = sub + j * LPR;
        ev[j] = c < C ? expf(xv[j] - mx) : 0.f;
        se += ev[j];
    }
    se = row_sum<LPR>(se);
    // argmax, first max on ties (torch.max(dim=1)); multitask: central + visual + skeleton logits
    float bv = -3.0e38f;
    int bi = 0x7FFFFFFF;
    const float* vl = nullptr;
    const float* sl = nullptr;
    if (g.multitask && ok) {
        const int64_t grow = ord ? (int64_t)ord[cs.pos_t + b] : (int64_t)(cs.base_t + b);
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
    row_argmax<LPR>(bv, bi);
    const float lse = mx + logf(se);
    if (sub == 0) {
        float ls = ok ? -(row[lab] - lse) : 0.f;
        if (vl) ls = (ls + row_ce(vl, C, lab)) + row_ce(sl, C, lab);   // multitask 3-term loss (ntu.py:60-61); constants w.r.t. the central parameters
        red_l[b] = ls;
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
__device__ __forceinline__ void softmax_rows(const ChainArgs& a, const ChainStep& cs, float* lg_l, const int SC, float* red_l,
                                             const int* lab_l, const int nvalid, const float nf, const int tid, const int32_t* ord) {
    constexpr int Bp = MB * 16;
    constexpr int LPR = (STEP_THREADS / Bp) < 16 ? (STEP_THREADS / Bp) : 16;
    if (a.g.Cp <= 4 * LPR) softmax_rows_nc<MB, 4>(a, cs, lg_l, SC, red_l, lab_l, nvalid, nf, tid, ord);
    else softmax_rows_nc<MB, 8>(a, cs, lg_l, SC, red_l, lab_l, nvalid, nf, tid, ord);
}

// Lean form of the same loss (round 5) for the lean chain's case — C <= 64 classes, no multitask: EIGHT lanes per batch row
// (threads [0, 8 Bp): one wave per SIMD at Bp = 32), every lane owns eight CONSECUTIVE classes (two ds_read_b128 of logits, two
// ds_write_b128 of dlogits); masks instead of exec-masked branches; independent per-class work in the lane (issue-bound, not a
// dependent chain) and three DPP stages per row reduction instead of four; exp through v_exp_f32 on (x - max) * log2(e) (absolute
// error <= 1e-7 of the probability: dlogits = p - onehot is an absolute quantity); both divisions through the correctly rounding
// sequences of common.hip.h with ONE reciprocal per row; and no arg-max: the row counts as correct iff the label's logit equals the
// row maximum and no EARLIER class does (= "first maximum is the label", torch.max(dim=1) on ties).
// softmax_rows_nc spent ~4,300 shader cycles of a ~20,000-cycle R = 16 chain here (profiles/r04_chain_phases.log): ~500 issued
// instructions per lane in one dependent string, two waves per SIMD.  Same results up to the summation order of the row's exp() terms.
template <int MB>
__device__ __forceinline__ void softmax_rows_lean(const ChainArgs& a, float* lg_l, const int SC, float* red_l, const int* lab_l,
                                                  const int nvalid, const float nf, const int tid) {
    constexpr int Bp = MB * 16;
    const int C = a.g.C;
    const int b = tid >> 3, sub = tid & 7, c0 = sub * 8;
    float* row = lg_l + b * SC;
    const bool ok = b < nvalid;
    const int lab = lab_l[b];
    const bool inrow = c0 < a.g.Cp;       // (Cp = 16 / 32 / 48 / 64: lanes beyond the padded row neither read nor write it)
    f32x4 xa = {0.f, 0.f, 0.f, 0.f}, xb = xa;
    if (inrow) {
        xa = *reinterpret_cast<const f32x4*>(row + c0);
        xb = *reinterpret_cast<const f32x4*>(row + c0 + 4);
    }
    const float xlab = row[lab];
    float xv[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        xv[q] = c0 + q < C ? xa[q] : -3.0e38f;
        xv[4 + q] = c0 + 4 + q < C ? xb[q] : -3.0e38f;
    }
    const float mx = row_max<8>(fmaxf(fmaxf(fmaxf(xv[0], xv[1]), fmaxf(xv[2], xv[3])), fmaxf(fmaxf(xv[4], xv[5]), fmaxf(xv[6], xv[7]))));
    f32x4 ea, eb;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        ea[q] = __builtin_amdgcn_exp2f((xv[q] - mx) * 1.44269504088896341f);
        eb[q] = __builtin_amdgcn_exp2f((xv[4 + q] - mx) * 1.44269504088896341f);
    }
    const float se = row_sum<8>(((ea[0] + ea[1]) + (ea[2] + ea[3])) + ((eb[0] + eb[1]) + (eb[2] + eb[3])));
    float early = 0.f;                    // an earlier class than the label attains the maximum
#pragma unroll
    for (int q = 0; q < 8; ++q) early = (c0 + q < lab && xv[q] == mx) ? 1.0f : early;
    early = row_max<8>(early);
    if (sub == 0) {
        const float lse = mx + __builtin_amdgcn_logf(se) * 0.693147180559945309f;
        red_l[b] = ok ? -(xlab - lse) : 0.f;
        red_l[Bp + b] = (ok && xlab == mx && early == 0.f) ? 1.f : 0.f;
    }
    const f32x4 se4 = (f32x4)(se), rse4 = (f32x4)(rcp_refined(se)), nf4 = (f32x4)(nf), rnf4 = (f32x4)(rcp_refined(nf));
    f32x4 da = div_by4(ea, se4, rse4), db = div_by4(eb, se4, rse4);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        da[q] = c0 + q == lab ? da[q] - 1.0f : da[q];
        db[q] = c0 + 4 + q == lab ? db[q] - 1.0f : db[q];
    }
    da = div_by4(da, nf4, rnf4);
    db = div_by4(db, nf4, rnf4);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        da[q] = (ok && c0 + q < C) ? da[q] : 0.f;
        db[q] = (ok && c0 + 4 + q < C) ? db[q] : 0.f;
    }
    if (inrow) {
        *reinterpret_cast<f32x4*>(row + c0) = da;
        *reinterpret_cast<f32x4*>(row + c0 + 4) = db;
    }
}

#ifdef MFAS_CHAIN_TIMING
#define CT_STAMP(slot) do { if (threadIdx.x == 0 && bid == 0 && cs.gstep == 3) a.status[64 + (slot)] = (int32_t)(__builtin_readcyclecounter() - ct0); } while (0)
#else
#define CT_STAMP(slot) do { } while (0)
#endif

template <int MB, bool PF, bool COH = false>
__device__ __forceinline__ void chain_body(const ChainArgs& a, const ChainStep& cs, const int bid, float* lds) {
#ifdef MFAS_CHAIN_TIMING
    const unsigned long long ct0 = __builtin_readcyclecounter();
#endif
    const CandDev& cd = a.cands[bid];
    const Geo& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    constexpr int Bp = MB * 16;
    // (co-scheduled builds of the two-batch-block chain live in 128 VGPRs: 4 tiles per batch in the same-group launch, where the chain
    //  is on the critical path, 2 under the other group's sweep — k_step<2,*,4,false> spilled 31 registers with 8, none with 2)
    constexpr int CHAIN_TW = (MB >= 2 && !PF) ? (COH ? 4 : 2) : (MB >= 4 ? 4 : 8);
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
        // (all loads of a batch of 4 strides are requested before the first LDS store: one memory round trip per batch, not
        // one per stride — the entry of the R=128 chain was 6.2 us of round trips, profiles/r02_chain_phases.log)
        for (int e0 = tid; e0 < nvec; e0 += 4 * CHAIN_THREADS) {
            float tw[4], tm[4], tv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * CHAIN_THREADS;
                const int ec = e < nvec ? e : 0;
                tw[u] = vecW[ec]; tm[u] = vecM[ec]; tv[u] = vecV[ec];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * CHAIN_THREADS;
                if (e < nvec) { vl[e] = tw[u]; vl[nvec + e] = tm[u]; vl[2 * nvec + e] = tv[u]; }
            }
        }
        vecW = vl; vecM = vl + nvec; vecV = vl + 2 * nvec;   // visible after the phase-0 barrier below
    }
    const int nvalid = cs.nvalid;
    const float nf = (float)nvalid;
    const AdamC ac = adam_consts(a.ac, cs.ss, cs.bc2s);
    const uint32_t h0 = lowbias32(cd.drop_seed + 0x9E3779B9U * (uint32_t)(cs.gstep + 1));
    const int64_t sbo = cd.step_off;   // this candidate's step buffers inside a.stepbuf (COH accesses index from the base)

    if (tid < Bp) {
        int lab = 0;
        if (tid < nvalid) {
            const int32_t* ord = cand_order(a.order, g, cd.gidx);
            const int64_t row = ord ? (int64_t)ord[cs.pos_t + tid] : (int64_t)(cs.base_t + tid);
            lab = g.loss_mode == 0 ? a.tab.label[row] : (int)row;   // mode 1 keeps the table row for the multi-hot targets
        }
        lab_l[tid] = lab;
    }

    // ------------------------------------------------------------------ phase 0: all 512 threads reduce the
    // sweep's column-chunk partial sums of EVERY cell (fixed order) into LDS, loads batched 8 deep
    if (a.yf_reduced) {   // one reduced slab per cell (two with alphas): a straight copy, same values the loop below would produce
        const int per_cell = nrb * MB * 64;
        const f32x4* src = reinterpret_cast<const f32x4*>(sb + g.sb_yf);
        if (yf_l != sb + g.sb_yf) {
            const int n = L * per_cell;
            for (int e0 = tid; e0 < n; e0 += 4 * CHAIN_THREADS) {      // 4 strides of loads in flight before the first LDS store
                f32x4 ts[4], tvv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = e0 + u * CHAIN_THREADS;
                    const int ec = e < n ? e : 0;
                    ts[u] = src[ec];
                    tvv[u] = g.alphas ? src[sav_plane / 4 + ec] : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = e0 + u * CHAIN_THREADS;
                    if (e < n) {
                        *reinterpret_cast<f32x4*>(yf_l + (int64_t)e * 4) = ts[u];
                        if (g.alphas) *reinterpret_cast<f32x4*>(yf_l + sav_plane + (int64_t)e * 4) = tvv[u];
                    }
                }
            }
        }
    } else {
        const int per_cell = nrb * MB * 64;   // float4 items per cell
        for (int e = tid; e < L * per_cell; e += CHAIN_THREADS) {
            const int i = e / per_cell, it = e - i * per_cell;
            const int ns = cd.nch_s[i], nch = ns + cd.nch_v[i];
            const int64_t part = sbo + g.sb_part + (((int64_t)cd.part_cell_off[i] * nrb * MB) << 8) + it * 4;
            f32x4 accS = {0.f, 0.f, 0.f, 0.f}, accV = {0.f, 0.f, 0.f, 0.f};
            constexpr int PB = PF ? 16 : (MB >= 2 ? 4 : 8);   // partial-sum loads in flight per thread (128-VGPR builds: fewer)
            for (int ch0 = 0; ch0 < nch; ch0 += PB) {
                f32x4 p8[PB];
#pragma unroll
                for (int u = 0; u < PB; ++u)
                    if (ch0 + u < nch) p8[u] = ldc4<COH>(a.stepbuf, part + (((int64_t)(ch0 + u) * nrb * MB) << 8));
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
    // (MB = 4, B > 32: four batch blocks of accumulators leave no room for two prefetched tile sets — 130 spilled registers with them)
    const bool pf = PF && MB < 4 && nrb <= CHAIN_NW && ncb <= CHAIN_NW;
    f32x4 wa[8], wb[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { wa[u] = (f32x4){0.f, 0.f, 0.f, 0.f}; wb[u] = wa[u]; }
    // products in order: P_1..P_{L-1} (prev-out block of cell i), head, then backward: head^T, outT_{L-1}..outT_1
    if (pf) {
        if (L > 1) { if (wave < nrb) issue_tiles<COH>(wa, W, cd.seg_off[1][2] + (int64_t)wave * nrb * 256, nrb, lane); }
        else if (wave < ncb) issue_tiles<COH>(wa, W, cd.head_off + (int64_t)wave * nrb * 256, nrb, lane);
    }

    CT_STAMP(0);
    // ------------------------------------------------------------------ forward chain
    for (int i = 0; i < L; ++i) {
        CT_STAMP(1 + i);
        if (pf && i >= 1) {   // wa holds P_i; request the NEXT product's tiles now: P_{i+1}, or the head after the last cell
            if (i + 1 < L) { if (wave < nrb) issue_tiles<COH>(wb, W, cd.seg_off[i + 1][2] + (int64_t)wave * nrb * 256, nrb, lane); }
            else if (wave < ncb) issue_tiles<COH>(wb, W, cd.head_off + (int64_t)wave * nrb * 256, nrb, lane);
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
                stc1<COH>(sb + g.sb_gsc + i * 2, sgS);
                stc1<COH>(sb + g.sb_gsc + i * 2 + 1, sgV);
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
                else lds_x_times_tiles<MB, COH, CHAIN_TW>(acc, xprev, SX, W, cd.seg_off[i][2] + (int64_t)rb * nrb * 256, 256, nrb, lane);
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
                    stc1<COH>(xo_g + b * Rp + r, o);
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
            issue_tiles<COH>(wb, a.wt, cd.headT_off + (int64_t)wave * ncb * 256, ncb, lane);
        for (int cb = wave; cb < ncb; cb += CHAIN_NW) {
            f32x4 acc[MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) acc[mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int c = cb * 16 + l15;
            const float bias = vecW[g.vec_head + c];
            if (pf) mma_tiles<MB>(acc, xl, SX, wa, nrb, lane);
            else lds_x_times_tiles<MB, COH, CHAIN_TW>(acc, xl, SX, W, cd.head_off + (int64_t)cb * nrb * 256, 256, nrb, lane);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int q = 0; q < 4; ++q) lg_l[(mb * 16 + 4 * lg + q) * SC + c] = acc[mb][q] + bias;
        }
    }
    lds_barrier();
    CT_STAMP(6);
    if (a.logits_out) {   // train-mode forward only
        for (int e = tid; e < nvalid * C; e += CHAIN_THREADS) {
            const int b = e / C, c = e - b * C;
            a.logits_out[e] = lg_l[b * SC + c];
        }
        return;
    }
    if (a.dlogits_in) {     // the caller's dL/dlogits instead of the loss gradient (rows / classes beyond the batch: 0)
        for (int e = tid; e < Bp * Cp; e += CHAIN_THREADS) {
            const int b = e / Cp, c = e - b * Cp;
            lg_l[b * SC + c] = (b < nvalid && c < C) ? a.dlogits_in[(int64_t)b * C + c] : 0.f;
        }
    } else if (g.loss_mode == 1) {
        if (tid < 4 * Bp) bce_rows(lg_l, SC, red_l, Bp, lab_l, a.tab.multilabel, a.pos_w, C, Cp, nvalid, tid);
    } else if (tid < LPR * Bp) {
        softmax_rows<MB>(a, cs, lg_l, SC, red_l, lab_l, nvalid, nf, tid, cand_order(a.order, g, cgidx));
    }
    lds_barrier();
    if (tid == CHAIN_THREADS - 64 && a.stats) {   // last wave: keeps the read-modify-write of the statistics off wave 0
        float ls = 0.f, ncor = 0.f;
        for (int b = 0; b < Bp; ++b) { ls += red_l[b]; ncor += red_l[Bp + b]; }
        DevStats& st = a.stats[(int64_t)cgidx * a.E + cs.epoch];
        st.train_loss += (double)ls;
        st.train_corr += (long long)ncor;
        if (!(fabsf(ls) <= 3.0e38f)) atomicMax(&a.status[cgidx], 1);   // (never downgrades a timeout mark 2 set by a sweep unit of the same launch)
    }
    CT_STAMP(7);
    // dlogits -> global (dy operand of the HEAD segment); head-bias Adam
    {
        float* dlg = sb + g.sb_dlog;
        for (int e = tid; e < Bp * Cp; e += CHAIN_THREADS) {
            const int b = e / Cp, c = e - b * Cp;
            stc1<COH>(dlg + e, lg_l[b * SC + c]);
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
            issue_tiles<COH>(wb, a.wt, cd.outT_off[i] + (int64_t)wave * nrb * 256, nrb, lane);
        const int nl = (nlbits >> (2 * i)) & 3;
        const int64_t vb = cvec_off + (int64_t)i * g.vec_cell_stride;
        const int vbl = i * g.vec_cell_stride;
        const bool from_head = (i == L - 1);
        const float* src = from_head ? lg_l : dy_l + ((i + 1) & 1) * Bp * SX;
        const int sstride = from_head ? SC : SX;
        const int nkk = from_head ? ncb : nrb;
        const int64_t Tidx = from_head ? cd.headT_off : cd.outT_off[i + 1];
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
            else lds_x_times_tiles<MB, COH, CHAIN_TW>(acc, src, sstride, a.wt, Tidx + (int64_t)rb * nkk * 256, 256, nkk, lane);
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
                    stc1<COH>(dy_g + b * Rp + r, dy);
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
        if constexpr (COH) {
            if (a.cellflag) {   // dy_i (and, with alphas, this step's scales) are out: cell i's sweep units of this launch may start
                wg_publish_barrier();
                if (tid == 0) __hip_atomic_store(a.cellflag + (size_t)cgidx * CELLFLAG_STRIDE + i, a.flag_target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
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
//   * the cells are ELEMENT-PARALLEL (round 2): wave w < 4*MB owns one element group of the 16 x 16 output block (batch-row
//     quad w & 3 of row block w >> 2), one output element per lane; every wave recomputes the cell's tiny product (4 MFMAs)
//     from out_{i-1} in LDS, finishes its element and keeps what the backward pass needs again (activation, x-hat, alpha
//     difference, dropout keep-bits) in REGISTERS; one workgroup barrier per cell (three with BatchNorm, whose batch statistics
//     cross the waves through 2 KB of LDS);
//   * head / softmax, coalesced copies of out_i, dy_i and dlogits to the step buffers use all eight waves; everything that is
//     not needed to publish dy — epoch statistics, bias / BatchNorm / alpha gradient sums and their Adam updates — is deferred
//     to chain_lean_tail, which the resident persistent chain runs AFTER dy is out.
// The arithmetic is that of chain_body except for the order in which bias / BatchNorm gradient sums are accumulated: the two
// chain forms agree to rounding, not bit for bit; which one runs depends only on (R, C, B) — mfas_hip.hip, `lean_chain`.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 pick4(const f32x4 (&t)[MFAS_MAX_CELLS], int i) {
    switch (i) { case 0: return t[0]; case 1: return t[1]; case 2: return t[2]; default: return t[3]; }
}

// Per-launch state of a RESIDENT lean chain (persistent schedule, persist.hip.h): the chain workgroup of a candidate keeps its
// vector block (+ Adam moments) and the weights it owns — the prev-out tiles OUT_1..OUT_{L-1} and the head tiles, with their
// moments and transposed images — in LDS for the whole epoch, and its statistics in registers.
struct LeanRes {
    double loss;        // running train loss of the epoch (thread CHAIN_THREADS - 64)
    long long corr;     // running correct count
    int bad;            // non-finite loss seen
};
#define LEAN_OWN_TILES 7   // slots 0..2: OUT_1..OUT_3, slots 3..6: head class blocks 0..3
#define LEAN_SCR 1024      // floats: BN exchange [2][8][16] | bias-gradient partials [L][8][16] | alpha partials [L][8] | dgamma / dbeta [L][2][16]

// LDS layout shared by chain_lean and the resident helpers
template <int MB>
struct LeanLds {
    static constexpr int Bp = MB * 16, Rp = 16, SX = Rp + 4, sav_plane = MFAS_MAX_CELLS * MB * 256;
    float *xo_l, *dy_l, *lg_l, *rstd_l, *red_l, *yf_l, *vec_l, *scr, *own;
    int* lab_l;
    int nvec, SC;
    __device__ __forceinline__ LeanLds(float* lds, const Geo& g) {
        SC = g.Cp + 4;
        nvec = MFAS_MAX_CELLS * g.vec_cell_stride + g.Cp;
        xo_l = lds;                                          // [L][Bp][SX] out_i of every cell
        dy_l = xo_l + MFAS_MAX_CELLS * Bp * SX;              // [L][Bp][SX] dy_i of every cell
        lg_l = dy_l + MFAS_MAX_CELLS * Bp * SX;              // [Bp][SC] logits -> dlogits
        rstd_l = lg_l + Bp * SC;                             // [L][Rp]
        red_l = rstd_l + MFAS_MAX_CELLS * Rp;                // [2*Bp + 16]
        lab_l = reinterpret_cast<int*>(red_l + 2 * Bp + 16); // [Bp]
        yf_l = reinterpret_cast<float*>(lab_l + Bp);         // [1 or 2][L][MB][256] reduced feature sums
        vec_l = yf_l + (g.alphas ? 2 : 1) * sav_plane;       // [3][nvec] vector block + Adam state
        // (activations / x-hat / alpha differences needed by the backward live in the owning lanes' registers)
        scr = vec_l + ((3 * nvec + 3) & ~3);                 // cross-wave exchange scratch (LEAN_SCR floats)
        own = scr + LEAN_SCR;                                                                 // resident: [W|M|V|T][7 tiles][256]
    }
    static __host__ __device__ constexpr int own_floats() { return 4 * LEAN_OWN_TILES * 256; }
};

// MODE 0: launch-per-phase schedule; 1: persistent, everything exchanged through memory (write-through / sc1);
// 2: persistent AND resident (LeanRes): vector block, OUT / HEAD weights and statistics live on chip, the chain's only
// global traffic per step is the sweep's partial sums in and dy (+ alpha scales) out.
template <int MB, int MODE = 0, int PB = 16>
__device__ __forceinline__ void chain_lean(const ChainArgs& a, const ChainStep& cs, const int bid, float* lds, LeanRes* rs = nullptr) {
    constexpr bool COH = MODE >= 1;
    constexpr bool RES = MODE == 2;
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
    const LeanLds<MB> ll(lds, g);
    const int nvec = ll.nvec;
    float* xo_l = ll.xo_l;
    float* dy_l = ll.dy_l;
    float* lg_l = ll.lg_l;
    float* rstd_l = ll.rstd_l;
    float* red_l = ll.red_l;
    int* lab_l = ll.lab_l;
    float* yf_l = ll.yf_l;
    float* vec_l = ll.vec_l;
    // vector-parameter updates: memory (MODE 0 / 1) or the resident LDS copy (MODE 2; flushed at the end of the launch)
    auto put_vec = [&](int64_t o, float w, float m, float v) {
        if constexpr (RES) { const int e = (int)(o - cd.vec_off); vec_l[e] = w; vec_l[nvec + e] = m; vec_l[2 * nvec + e] = v; }
        else { a.plane[o] = w; a.plane[a.plane_stride + o] = m; a.plane[2 * a.plane_stride + o] = v; }
    };

    float* W = a.plane;
    float* Mv = a.plane + a.plane_stride;
    float* Vv = Mv + a.plane_stride;
    float* sb = a.stepbuf + cd.step_off;
    const int64_t cvec_off = cd.vec_off;
    const int cgidx = cd.gidx;
    int nlbits = 0;
#pragma unroll
    for (int i = 0; i < MFAS_MAX_CELLS; ++i) nlbits |= (cd.conf[i][2] & 3) << (2 * i);
    const int nvalid = cs.nvalid;
    const float nf = (float)nvalid;
    const AdamC ac = adam_consts(a.ac, cs.ss, cs.bc2s);
    const uint32_t h0 = lowbias32(cd.drop_seed + 0x9E3779B9U * (uint32_t)(cs.gstep + 1));
    const int64_t sbo = cd.step_off;

    // ------------------------------------------------------------------ entry: every global read of the chain is
    // requested here, in the order the results are needed (the memory counter retires in order): the sweep's partial
    // sums first, then the vector block, the weight tiles of all products and last the labels (a dependent pair of loads
    // that nothing needs before the loss, fetched by wave 1 so that wave 0 never waits for them)
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    constexpr int per_cell = MB * 64;   // float4 partial-sum items per cell; L * per_cell <= 512: one item per thread
    // PB = partial-sum chunks requested per thread before any is consumed (16; 8 in the 128-VGPR co-scheduled builds of k_step)
    const bool has_item = tid < L * per_cell;
    // (per_cell = MB * 64: the cell index is wave-uniform -> scalar loads of cd.nch_* / part_cell_off)
    const int pi = __builtin_amdgcn_readfirstlane(has_item ? tid / per_cell : 0), pit = tid - pi * per_cell;
    const int ns = cd.nch_s[pi], nch = has_item ? ns + cd.nch_v[pi] : 0;
    const int64_t part = sbo + g.sb_part + (((int64_t)cd.part_cell_off[pi] * MB) << 8) + pit * 4;
    // every load below is UNCONDITIONAL (indices clamped to something valid): with a statically known number of loads in
    // flight the compiler can wait for exactly the ones it consumes instead of draining the whole queue at first use
    f32x4 p8[PB];
    if constexpr (RES) {
        // resident chain: only the slabs that exist (wave-uniform count -> scalar branches).  The unconditional form below re-requests
        // slab 0 for every missing one — 16 requests per thread for the 2-4 slabs per cell of 256...1024-column units: 128 KB of
        // requests through this one CU's memory pipeline per step for 30-60 KB of partial sums, on the critical path.
        const int nch_u = __builtin_amdgcn_readfirstlane(nch);
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            p8[u] = z4;
            if (u < nch_u) p8[u] = ldc4<COH>(a.stepbuf, part + (((int64_t)u * MB) << 8));
        }
    } else {
#pragma unroll
    for (int u = 0; u < PB; ++u) {
        const int uu = u < nch ? u : 0;
        p8[u] = ldc4<COH>(a.stepbuf, part + (((int64_t)uu * MB) << 8));
    }
    }
    const int vi = tid < nvec ? tid : 0;   // nvec = 4 * 96 + Cp <= 448: one element of the vector block per thread
    float vw = 0.f, vm = 0.f, vv = 0.f;
    if constexpr (!RES) { vw = W[cvec_off + vi]; vm = Mv[cvec_off + vi]; vv = Vv[cvec_off + vi]; }
    // weight tiles of every product (all waves fetch them — 11 KiB, uniform control flow; wave 0 / waves < ncb use them)
    f32x4 tP[MFAS_MAX_CELLS], tT[MFAS_MAX_CELLS], tHT[4], tH;   // prev-out tile of cell i, its transpose, head^T, head
    tP[0] = z4; tT[0] = z4;
    if constexpr (RES) {   // the chain owns these weights: LDS-resident images (slot i-1: OUT_i, slot 3+u: head block u), read where
                           // they are used (an LDS read costs ~100 cycles; 13 tiles held from entry cost 52 VGPRs and spilled)
#pragma unroll
        for (int i = 1; i < MFAS_MAX_CELLS; ++i) { tP[i] = z4; tT[i] = z4; }
#pragma unroll
        for (int u = 0; u < 4; ++u) tHT[u] = z4;
        tH = z4;
    } else {
#pragma unroll
        for (int i = 1; i < MFAS_MAX_CELLS; ++i) {
            // (cells beyond L: any valid address — the loads stay unconditional, their values are never used)
            tP[i] = ldc4<COH>(W, (i < L ? cd.seg_off[i][2] : cvec_off) + lane * 4);
            tT[i] = ldc4<COH>(a.wt, (i < L ? cd.outT_off[i] : cd.headT_off) + lane * 4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            tHT[u] = ldc4<COH>(a.wt, cd.headT_off + ((int64_t)(u < ncb ? u : 0) << 8) + lane * 4);
        tH = ldc4<COH>(W, cd.head_off + ((int64_t)(wave < ncb ? wave : 0) << 8) + lane * 4);
    }
    const int r = l15;
    const bool colok = r < R;
    // phase 0: reduce the sweep's column-chunk partial sums (fixed order) into LDS; stage the vector block.
    // Resident chain (round 5): only cell 0's sums are waited for here — the waves that hold the slabs of cells 1..L-1 (the cell
    // index is wave-uniform) consume theirs at the end of cell 0's forward, so the 3/4 of the slab bytes that cell 0 does not need
    // stream into this CU while cell 0 is computed instead of in front of it (the chain spent ~5,000 of ~14,000 cycles per step
    // between "units have arrived" and "sums in LDS": profiles/r04_chain_phases.log, slot 0).
// (a macro, not a lambda: with the slab registers captured by reference the closure keeps p8[] addressable and the whole array
//  lands in scratch memory — 272 bytes per lane, measured: entry 5,400 -> 15,900 cycles)
#define LEAN_CONSUME() do { \
        f32x4 accS = z4, accV = z4; \
_Pragma("unroll") \
        for (int u = 0; u < PB; ++u) \
            if (u < nch) { \
                if (u < ns) accS += p8[u]; else accV += p8[u]; \
            } \
        for (int ch0 = PB; ch0 < nch; ch0 += PB) { \
_Pragma("unroll") \
            for (int u = 0; u < PB; ++u) \
                if (ch0 + u < nch) p8[u] = ldc4<COH>(a.stepbuf, part + (((int64_t)(ch0 + u) * MB) << 8)); \
_Pragma("unroll") \
            for (int u = 0; u < PB; ++u) \
                if (ch0 + u < nch) { \
                    if (ch0 + u < ns) accS += p8[u]; else accV += p8[u]; \
                } \
        } \
        if (g.alphas) { \
            *reinterpret_cast<f32x4*>(yf_l + tid * 4) = accS; \
            *reinterpret_cast<f32x4*>(yf_l + sav_plane + tid * 4) = accV; \
        } else { \
            *reinterpret_cast<f32x4*>(yf_l + tid * 4) = accS + accV; \
        } \
    } while (0)
    // dropout keep-bits of this lane's element in every cell (a few dozen integer instructions: issued under the slab loads)
    const int ew = wave, emb = ew >> 2, eq = ew & 3;
    const bool eact = ew < MB * 4;
    const int eb = emb * 16 + 4 * lg + eq;                  // this lane's batch row
    uint32_t ekeep = 0xFu;
    if (g.use_drop) {
        ekeep = 0u;
#pragma unroll
        for (int i = 0; i < MFAS_MAX_CELLS; ++i)
            if (drop_keep(h0, i, (uint32_t)(eb * R + r), g.drop_thr)) ekeep |= 1u << i;
    }
    if constexpr (RES) {
        if (has_item && pi == 0) LEAN_CONSUME();
    } else {
        if (has_item) LEAN_CONSUME();
    }
    if constexpr (!RES) {
        if (tid < nvec) { vec_l[tid] = vw; vec_l[nvec + tid] = vm; vec_l[2 * nvec + tid] = vv; }
    }
    // labels: a dependent pair of loads that nothing needs before the loss; requested last, by wave 1, consumed after the
    // forward pass
    int lab = 0;
    if (wave == 1 && lane < nvalid) {
        const int32_t* ord = cand_order(a.order, g, cgidx);
        const int64_t row = ord ? (int64_t)ord[cs.pos_t + lane] : (int64_t)(cs.base_t + lane);
        lab = g.loss_mode == 0 ? a.tab.label[row] : (int)row;   // mode 1 keeps the table row for the multi-hot targets
    }
    const float* vecW = vec_l;
    const float* vecM = vec_l + nvec;
    const float* vecV = vec_l + 2 * nvec;
    lds_barrier();
    CT_STAMP(0);

    // ------------------------------------------------------------------ forward: ELEMENT-PARALLEL over the waves.
    // A 16-wide cell is ~150 dependent VALU / transcendental instructions per lane when one wave owns all Bp x 16 outputs (8 per
    // lane at B = 20); here wave w < 4*MB owns ONE element row group — batch rows mb*16 + 4*lg + q with (mb, q) = (w >> 2, w & 3)
    // — every wave recomputes the cell's tiny product (4 MFMAs) from the shared out_{i-1} in LDS and finishes one element per
    // lane, then one workgroup barrier hands out_i to the next cell.  BatchNorm's batch statistics and the backward's column
    // sums are exchanged through LDS (fixed order over the waves).  What each lane needs again in the backward (activation,
    // x-hat, alpha difference) stays in its registers.
    float* bnw = ll.scr;                                     // [2][8][16] cross-wave column sums
    float* gv2 = ll.scr + 256;                               // [cell][dgamma | dbeta][16]
    auto sel4 = [&](const f32x4& v4) -> float { return eq == 0 ? v4[0] : (eq == 1 ? v4[1] : (eq == 2 ? v4[2] : v4[3])); };
    float av[MFAS_MAX_CELLS], xhs[MFAS_MAX_CELLS], dsv[MFAS_MAX_CELLS];
#pragma unroll
    for (int i = 0; i < MFAS_MAX_CELLS; ++i) {
        av[i] = 0.f; xhs[i] = 0.f; dsv[i] = 0.f;
        if (i < L) {      // (workgroup-uniform)
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
                if (tid == 0) {
                    stc1<COH>(sb + g.sb_gsc + i * 2, sgS);
                    stc1<COH>(sb + g.sb_gsc + i * 2 + 1, sgV);
                }
            }
            float v = 0.f;
            if (eact) {
                const int o = ((i * MB + emb) << 8) + lane * 4;
                f32x4 acc = *reinterpret_cast<const f32x4*>(yf_l + o);
                if (g.alphas) {
                    const f32x4 yv = *reinterpret_cast<const f32x4*>(yf_l + sav_plane + o);
                    dsv[i] = sel4(acc) - sel4(yv);
                    acc = acc * sgS + yv * sgV;
                }
                if (i > 0) {
                    f32x4 w;
                    if constexpr (RES) w = *reinterpret_cast<const f32x4*>(ll.own + ((i - 1) << 8) + lane * 4);
                    else w = pick4(tP, i);
                    const f32x4 x4 = *reinterpret_cast<const f32x4*>(xo_l + (i - 1) * Bp * SX + (emb * 16 + l15) * SX + 4 * lg);
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc = MFMA16(x4[q], w[q], acc);
                }
                v = act_fwd(sel4(acc) + bias, nl);
            }
            float z = v;
            if (g.bn) {   // batch statistics over the valid rows: two exchanges (mean, then the variance of the deviations)
                float s1 = colsum((eact && eb < nvalid) ? v : 0.f);
                if (lg == 0) bnw[ew * 16 + r] = s1;
                lds_barrier();
                float tot = 0.f;
#pragma unroll
                for (int w = 0; w < MB * 4; ++w) tot += bnw[w * 16 + r];
                const float mu = tot / nf;
                const float dlt = v - mu;
                float s2 = colsum((eact && eb < nvalid) ? dlt * dlt : 0.f);
                if (lg == 0) bnw[128 + ew * 16 + r] = s2;
                lds_barrier();
                float tot2 = 0.f;
#pragma unroll
                for (int w = 0; w < MB * 4; ++w) tot2 += bnw[128 + w * 16 + r];
                const float var = tot2 / nf;
                const float rstd = 1.0f / sqrtf(var + g.bn_eps);
                const float xh = (v - mu) * rstd;
                xhs[i] = xh;
                z = xh * gam + bet;
                if (wave == 0 && lg == 0) {
                    rstd_l[i * Rp + r] = rstd;
                    if (colok) {   // running stats: momentum 0.1, unbiased variance
                        float rm = vecW[vbl + VEC_RM * Rp + r], rv = vecW[vbl + VEC_RV * Rp + r];
                        const float unb = var * (nf / (nf - 1.0f));
                        rm += g.bn_mom * (mu - rm);
                        rv += g.bn_mom * (unb - rv);
                        if constexpr (RES) { vec_l[vbl + VEC_RM * Rp + r] = rm; vec_l[vbl + VEC_RV * Rp + r] = rv; }
                        else { W[vb + VEC_RM * Rp + r] = rm; W[vb + VEC_RV * Rp + r] = rv; }
                    }
                }
            }
            av[i] = v;
            if (eact) {
                float o = z;
                if (g.use_drop) o = ((ekeep >> i) & 1u) ? o * g.drop_scale : 0.0f;
                if (!(colok && eb < nvalid)) o = 0.0f;
                xo_l[i * Bp * SX + eb * SX + r] = o;
            }
            if constexpr (RES) {
                if (i == 0 && has_item && pi > 0) LEAN_CONSUME();     // the other cells' sums (see phase 0)
            }
            if (i + 1 < L) lds_barrier();     // (the last cell's hand-off is the barrier below)
        }
    }
    if (wave == 1 && lane < Bp) lab_l[lane] = lab;   // visible to the loss after the head's barrier
    lds_barrier();
    CT_STAMP(5);

    // ------------------------------------------------------------------ out_i -> step buffer (x operand of the sweep's
    // OUT / HEAD segments), coalesced, by everyone; head on waves < ncb
    {
        if constexpr (!RES) {   // (the resident chain updates OUT / HEAD itself: no sweep reads out_i)
            float* xo_g = sb + g.sb_xo;   // [L][Bp][Rp]
            for (int e = tid; e < L * Bp * Rp; e += CHAIN_THREADS) stc1<COH>(xo_g + e, xo_l[(e >> 4) * SX + (e & 15)]);
        }
        const float* xl = xo_l + (L - 1) * Bp * SX;
        if (wave < ncb) {
            const int c = wave * 16 + l15;
            const float bias = vecW[g.vec_head + c];
            if constexpr (RES) tH = *reinterpret_cast<const f32x4*>(ll.own + ((3 + wave) << 8) + lane * 4);
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
    if constexpr (MODE == 0) {
        if (a.logits_out) {   // train-mode forward only
            for (int e = tid; e < nvalid * C; e += CHAIN_THREADS) {
                const int b = e / C, c = e - b * C;
                a.logits_out[e] = lg_l[b * SC + c];
            }
            return;
        }
    }
    if (a.dlogits_in) {     // the caller's dL/dlogits instead of the loss gradient (rows / classes beyond the batch: 0)
        for (int e = tid; e < Bp * Cp; e += CHAIN_THREADS) {
            const int b = e / Cp, c = e - b * Cp;
            lg_l[b * SC + c] = (b < nvalid && c < C) ? a.dlogits_in[(int64_t)b * C + c] : 0.f;
        }
    } else if (g.loss_mode == 1) {
        if (tid < 4 * Bp) bce_rows(lg_l, SC, red_l, Bp, lab_l, a.tab.multilabel, a.pos_w, C, Cp, nvalid, tid);
    } else if (!g.multitask) {
        if (tid < 8 * Bp) softmax_rows_lean<MB>(a, lg_l, SC, red_l, lab_l, nvalid, nf, tid);
    } else if (tid < LPR * Bp) {
        softmax_rows<MB>(a, cs, lg_l, SC, red_l, lab_l, nvalid, nf, tid, cand_order(a.order, g, cgidx));
    }
    lds_barrier();
    CT_STAMP(7);
    // ------------------------------------------------------------------ backward: element-parallel like the forward; one
    // barrier per cell (two with BatchNorm: the column sums of dz and dz * x-hat are needed before the activation gradient)
#pragma unroll
    for (int i = MFAS_MAX_CELLS - 1; i >= 0; --i) {
        if (i < L) {
            CT_STAMP(8 + (L - 1 - i));
            const int nl = (nlbits >> (2 * i)) & 3;
            const int vbl = i * g.vec_cell_stride;
            const bool from_head = (i == L - 1);
            float gr = 0.f;
            if (g.bn) gr = vecW[vbl + VEC_G * Rp + r] * rstd_l[i * Rp + r];
            if (from_head) {
                // d_out = dlogits . Wc (round 5): every wave used to run all four class blocks' products for its row block — 16 MFMAs per
                // wave, 32 per SIMD at 32 cycles of matrix pipe each, the longest phase of the backward (profiles/r04_chain_phases.log:
                // 2,236 cycles against ~1,000 for the other cells).  Now wave (row block emb, class block eq) runs ONE block's four
                // products and parks the 16 x 16 partial in the (dead) reduced-sums plane; after the barrier every element owner adds
                // the four partials of its row block, (0 + 2) + (1 + 3).
                if (eact) {
                    f32x4 pt = z4;
                    if (eq < ncb) {
                        const f32x4 x4 = *reinterpret_cast<const f32x4*>(lg_l + (emb * 16 + l15) * SC + eq * 16 + 4 * lg);
                        f32x4 wt4;
                        if constexpr (RES) wt4 = *reinterpret_cast<const f32x4*>(ll.own + ((3 * LEAN_OWN_TILES + 3 + eq) << 8) + lane * 4);
                        else wt4 = pick4(tHT, eq);
#pragma unroll
                        for (int q = 0; q < 4; ++q) pt = MFMA16(x4[q], wt4[q], pt);
                    }
                    *reinterpret_cast<f32x4*>(yf_l + (ew << 8) + lane * 4) = pt;
                }
                lds_barrier();
            }
            float d = 0.f;
            if (eact) {
                f32x4 acc = z4;
                if (from_head) {
                    const float* pp = yf_l + ((emb * 4) << 8) + lane * 4;
                    const f32x4 p0 = *reinterpret_cast<const f32x4*>(pp), p1 = *reinterpret_cast<const f32x4*>(pp + 256);
                    const f32x4 p2 = *reinterpret_cast<const f32x4*>(pp + 512), p3 = *reinterpret_cast<const f32x4*>(pp + 768);
                    acc = (p0 + p2) + (p1 + p3);
                } else {
                    f32x4 w;
                    if constexpr (RES) w = *reinterpret_cast<const f32x4*>(ll.own + ((3 * LEAN_OWN_TILES + i) << 8) + lane * 4);
                    else w = pick4(tT, i + 1);
                    const f32x4 x4 = *reinterpret_cast<const f32x4*>(dy_l + (i + 1) * Bp * SX + (emb * 16 + l15) * SX + 4 * lg);
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc = MFMA16(x4[q], w[q], acc);
                }
                d = sel4(acc);
                if (g.use_drop) d = ((ekeep >> i) & 1u) ? d * g.drop_scale : 0.0f;
                if (!(eb < nvalid)) d = 0.f;
            }
            float dz = d;
            if (g.bn) {
                const float p0 = colsum(d), p1 = colsum(d * xhs[i]);
                if (lg == 0) { bnw[ew * 16 + r] = p0; bnw[128 + ew * 16 + r] = p1; }
                lds_barrier();
                float dbet = 0.f, dgam = 0.f;
#pragma unroll
                for (int w = 0; w < MB * 4; ++w) { dbet += bnw[w * 16 + r]; dgam += bnw[128 + w * 16 + r]; }
                const float k1 = dbet / nf, k2 = dgam / nf;
                const float da = gr * (d - k1 - xhs[i] * k2);
                dz = (eact && eb < nvalid) ? da : 0.f;
                if (wave == 0 && lg == 0) { gv2[(i * 2 + 0) * 16 + r] = dgam; gv2[(i * 2 + 1) * 16 + r] = dbet; }
            }
            float dy = eact ? act_bwd(av[i], dz, nl) : 0.f;
            if (!colok) dy = 0.f;
            if (eact) {
                dy_l[i * Bp * SX + eb * SX + r] = dy;
                // d(alpha_i) needs sum_{b,r} dy[b,r] * (yS_raw - yV_raw)[b,r]: the products go to the (now dead) V plane of the
                // reduced feature sums, summed in fixed order by chain_lean_tail
                if (g.alphas) yf_l[sav_plane + (i * Bp + eb) * 16 + r] = dy * dsv[i];
            }
            lds_barrier();
        }
    }
    CT_STAMP(12);
    {   // dy_i -> step buffer (dy operand of the sweep), coalesced
        if constexpr (RES) {   // 16 B write-through stores: [L][Bp][16] = one f32x4 per thread and cell pair
            for (int e4 = tid; e4 < L * Bp * 4; e4 += CHAIN_THREADS)
                stc4<true>(a.stepbuf, sbo + g.sb_dy + (int64_t)e4 * 4, *reinterpret_cast<const f32x4*>(dy_l + (e4 >> 2) * SX + (e4 & 3) * 4));
        } else {
            float* dy_g = sb + g.sb_dy;   // [L][Bp][Rp]
            for (int e = tid; e < L * Bp * Rp; e += CHAIN_THREADS) stc1<COH>(dy_g + e, dy_l[(e >> 4) * SX + (e & 15)]);
        }
    }
    if constexpr (!RES) {   // dlogits -> step buffer (dy operand of the HEAD segment)
        float* dlg = sb + g.sb_dlog;
        for (int e = tid; e < Bp * Cp; e += CHAIN_THREADS) {
            const int b = e / Cp, c = e - b * Cp;
            stc1<COH>(dlg + e, lg_l[b * SC + c]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// chain_lean_tail — what a train step still owes after dy is out: the epoch statistics and Adam on the vector parameters
// (head bias; per cell bias, BN gamma / beta, alpha).  Gradients are summed from the step's LDS state (dy_i, dlogits, the BN
// exchange totals) in fixed order.  Launch-per-phase schedules call it right after chain_lean; the resident persistent chain
// calls it AFTER publishing dy, off the critical path.
// ------------------------------------------------------------------------------------------------
template <int MB, int MODE>
__device__ __forceinline__ void chain_lean_tail(const ChainArgs& a, const ChainStep& cs, const int bid, float* lds, LeanRes* rs = nullptr) {
    constexpr bool RES = MODE == 2;
    const CandDev& cd = a.cands[bid];
    const Geo& g = a.g;
    const LeanLds<MB> ll(lds, g);
    constexpr int Bp = MB * 16, Rp = 16, SX = Rp + 4, sav_plane = MFAS_MAX_CELLS * MB * 256;
    const int tid = threadIdx.x;
    const int C = g.C, R = g.R, L = cd.L, SC = ll.SC, nvec = ll.nvec;
    const float* lg_l = ll.lg_l;
    const float* dy_l = ll.dy_l;
    const float* red_l = ll.red_l;
    float* vec_l = ll.vec_l;
    const float* gv2 = ll.scr + 256;
    const int64_t cvec_off = cd.vec_off;
    const int cgidx = cd.gidx;
    const AdamC ac = adam_consts(a.ac, cs.ss, cs.bc2s);
    // parameters + moments: the LDS copy (MODE 2: the master copy; MODE 0 / 1: what chain_lean staged at entry — only the running
    // statistics and nothing below have been written since)
    const float* vecW = vec_l;
    const float* vecM = vec_l + nvec;
    const float* vecV = vec_l + 2 * nvec;
    auto put_vec = [&](int64_t o, float w, float m, float v) {
        if constexpr (RES) { const int e = (int)(o - cd.vec_off); vec_l[e] = w; vec_l[nvec + e] = m; vec_l[2 * nvec + e] = v; }
        else { a.plane[o] = w; a.plane[a.plane_stride + o] = m; a.plane[2 * a.plane_stride + o] = v; }
    };
    if (a.logits_out) return;       // train-mode forward only: no statistics, no update
    if (tid == CHAIN_THREADS - 64 && (RES || a.stats)) {
        float ls = 0.f, ncor = 0.f;
        for (int b = 0; b < Bp; ++b) { ls += red_l[b]; ncor += red_l[Bp + b]; }
        if constexpr (RES) {   // accumulated in registers for the epoch, flushed by lean_res_store
            rs->loss += (double)ls;
            rs->corr += (long long)ncor;
            if (!(fabsf(ls) <= 3.0e38f)) rs->bad = 1;
        } else {
            DevStats& st = a.stats[(int64_t)cgidx * a.E + cs.epoch];
            st.train_loss += (double)ls;
            st.train_corr += (long long)ncor;
            if (!(fabsf(ls) <= 3.0e38f)) atomicMax(&a.status[cgidx], 1);   // (never downgrades a timeout mark 2 set by a sweep unit of the same launch)
        }
    }
    const int hc = tid - (CHAIN_THREADS - 256);
    if (hc >= 0 && hc < C) {            // head bias: column sums of dlogits
        float gsum = 0.f;
        for (int b = 0; b < Bp; ++b) gsum += lg_l[b * SC + hc];
        const int64_t o = cvec_off + g.vec_head + hc;
        float w = vecW[g.vec_head + hc], m = vecM[g.vec_head + hc], v = vecV[g.vec_head + hc];
        adam1(w, m, v, gsum, ac);
        put_vec(o, w, m, v);
    }
    if (tid < L * 48) {                 // per cell: bias (column sums of dy_i over the batch rows, in row order), BN gamma / beta
        const int i = tid / 48, which = (tid - i * 48) >> 4, rr = tid & 15;
        if (rr < R && (which == 0 || g.bn)) {
            float gsum = 0.f;
            if (which == 0) {
                for (int b = 0; b < Bp; ++b) gsum += dy_l[i * Bp * SX + b * SX + rr];
            } else {
                gsum = gv2[(i * 2 + (which - 1)) * 16 + rr];
            }
            const int e = i * g.vec_cell_stride + (which == 0 ? VEC_B : (which == 1 ? VEC_G : VEC_BE)) * Rp + rr;
            float w = vecW[e], m = vecM[e], v = vecV[e];
            adam1(w, m, v, gsum, ac);
            put_vec(cvec_off + e, w, m, v);
        }
    } else if (g.alphas && tid >= 192 && tid < 192 + L) {   // alpha_i: sum over rows, then over columns, of dy * (yS - yV)
        const int i = tid - 192, e = i * g.vec_cell_stride + 5 * Rp;
        const float* prod = ll.yf_l + sav_plane + i * Bp * 16;
        float tot = 0.f;
        for (int rr = 0; rr < 16; ++rr) {
            float col = 0.f;
            for (int b = 0; b < Bp; ++b) col += prod[b * 16 + rr];
            tot += col;
        }
        float w = vecW[e], m = vecM[e], v = vecV[e];
        const float sg = 1.0f / (1.0f + expf(-w));
        adam1(w, m, v, tot * sg * (1.0f - sg), ac);
        put_vec(cvec_off + e, w, m, v);
    }
}

// ------------------------------------------------------------------------------------------------
// Resident lean chain, launch prologue / per-step weight update / launch epilogue (persist.hip.h, MODE 2)
// ------------------------------------------------------------------------------------------------
// own-tile plane offsets of slot s: OUT_{s+1} (s < 3), head class block s-3
__device__ __forceinline__ int64_t lean_own_off(const CandDev& cd, int s) { return s < 3 ? cd.seg_off[s + 1][2] : cd.head_off + ((int64_t)(s - 3) << 8); }
__device__ __forceinline__ int64_t lean_own_toff(const CandDev& cd, int s) { return s < 3 ? cd.outT_off[s + 1] : cd.headT_off + ((int64_t)(s - 3) << 8); }
__device__ __forceinline__ bool lean_own_live(const CandDev& cd, const Geo& g, int s) { return s < 3 ? (s + 1 < cd.L) : (s - 3 < g.ncb); }

template <int MB>
__device__ __forceinline__ void lean_res_load(const ChainArgs& a, const int bid, float* lds, LeanRes& rs) {
    const CandDev& cd = a.cands[bid];
    const LeanLds<MB> ll(lds, a.g);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < ll.nvec; e += CHAIN_THREADS)
        for (int pl = 0; pl < 3; ++pl) ll.vec_l[pl * ll.nvec + e] = a.plane[pl * a.plane_stride + cd.vec_off + e];
    if (wave < LEAN_OWN_TILES) {
        const int s = wave;
        const bool live = lean_own_live(cd, a.g, s);
        for (int pl = 0; pl < 3; ++pl) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (live) v = *reinterpret_cast<const f32x4*>(a.plane + pl * a.plane_stride + lean_own_off(cd, s) + lane * 4);
            *reinterpret_cast<f32x4*>(ll.own + (pl * LEAN_OWN_TILES + s) * 256 + lane * 4) = v;
        }
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        if (live) t = *reinterpret_cast<const f32x4*>(a.wt + lean_own_toff(cd, s) + lane * 4);
        *reinterpret_cast<f32x4*>(ll.own + (3 * LEAN_OWN_TILES + s) * 256 + lane * 4) = t;
    }
    if (tid == CHAIN_THREADS - 64) { rs.loss = 0.0; rs.corr = 0; rs.bad = 0; }     // (rs may live in LDS: one writer, the lane that accumulates)
    __syncthreads();
}

// dW + Adam of the weights the chain owns, after dy has been published (off the sweep's critical path).  Wave i (1 <= i < L)
// updates OUT_i with x = out_{i-1}, dy = dy_i; wave 4 + u updates head class block u with x = out_{L-1}, dy = dlogits.
// Arithmetic = tile_run's for these segments (MB*4 MFMAs in batch order, gradient scale 1, adam1) — bit-identical.
template <int MB>
__device__ __forceinline__ void lean_res_update(const ChainArgs& a, const ChainStep& cs, const int bid, float* lds) {
    const CandDev& cd = a.cands[bid];
    const Geo& g = a.g;
    const LeanLds<MB> ll(lds, g);
    constexpr int Bp = MB * 16, SX = 20;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int L = cd.L;
    int s = -1;
    const float* x = nullptr;
    const float* dy = nullptr;
    int sx = SX, sd = SX, dcol = 0;
    if (wave >= 1 && wave < L) { s = wave - 1; x = ll.xo_l + (wave - 1) * Bp * SX; dy = ll.dy_l + wave * Bp * SX; }
    else if (wave >= 4 && wave - 4 < g.ncb) { s = 3 + (wave - 4); x = ll.xo_l + (L - 1) * Bp * SX; dy = ll.lg_l; sd = ll.SC; dcol = (wave - 4) * 16; }
    if (s >= 0) {
        float* oW = ll.own + s * 256 + lane * 4;
        float* oM = oW + LEAN_OWN_TILES * 256;
        float* oV = oM + LEAN_OWN_TILES * 256;
        f32x4 w4 = *reinterpret_cast<const f32x4*>(oW), m4 = *reinterpret_cast<const f32x4*>(oM), v4 = *reinterpret_cast<const f32x4*>(oV);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        DW_BATCH_LOOP(MB, (g.B + 3) >> 2, acc = MFMA16(x[(4 * j + lg) * sx + l15], dy[(4 * j + lg) * sd + dcol + l15], acc))
        const float gsc = 1.0f;
        adam4(w4, m4, v4, acc * gsc, cs.ss, cs.bc2s, a.ac.w1, a.ac.b2, a.ac.w2, a.ac.eps, a.ac.wd);
        *reinterpret_cast<f32x4*>(oW) = w4;
        *reinterpret_cast<f32x4*>(oM) = m4;
        *reinterpret_cast<f32x4*>(oV) = v4;
        float* oT = ll.own + (3 * LEAN_OWN_TILES + s) * 256;        // transposed image for the backward chain
        const int base = (((l15 >> 2) * 16 + 4 * lg) << 2) + (l15 & 3);
#pragma unroll
        for (int q = 0; q < 4; ++q) oT[base + 4 * q] = w4[q];
    }
    lds_barrier();
}

template <int MB>
__device__ __forceinline__ void lean_res_store(const ChainArgs& a, const int bid, const int epoch, float* lds, const LeanRes& rs) {
    const CandDev& cd = a.cands[bid];
    const LeanLds<MB> ll(lds, a.g);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __syncthreads();
    for (int e = tid; e < ll.nvec; e += CHAIN_THREADS)
        for (int pl = 0; pl < 3; ++pl) a.plane[pl * a.plane_stride + cd.vec_off + e] = ll.vec_l[pl * ll.nvec + e];
    if (wave < LEAN_OWN_TILES && lean_own_live(cd, a.g, wave)) {
        const int s = wave;
        for (int pl = 0; pl < 3; ++pl)
            *reinterpret_cast<f32x4*>(a.plane + pl * a.plane_stride + lean_own_off(cd, s) + lane * 4) =
                *reinterpret_cast<const f32x4*>(ll.own + (pl * LEAN_OWN_TILES + s) * 256 + lane * 4);
        *reinterpret_cast<f32x4*>(const_cast<float*>(a.wt) + lean_own_toff(cd, s) + lane * 4) =
            *reinterpret_cast<const f32x4*>(ll.own + (3 * LEAN_OWN_TILES + s) * 256 + lane * 4);
    }
    if (tid == CHAIN_THREADS - 64) {
        DevStats& st = a.stats[(int64_t)cd.gidx * a.E + epoch];
        st.train_loss += rs.loss;
        st.train_corr += rs.corr;
        if (rs.bad) atomicMax(&a.status[cd.gidx], 1);   // (never downgrades a timeout mark 2 set by a sweep unit of the same launch)
    }
}

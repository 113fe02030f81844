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
    // chain_split (one candidate's chain on nsplit CUs): exchange area [candidate][parity][slot][row block][256], parity of this launch,
    // candidates of this launch (chain block b = part * ceil8(ncand) + candidate: the parts of a candidate share b % 8 = their XCD)
    float* xch;
    int32_t xpar, nsplit, ncand, _pads;
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
// debug checksums (schedule bit-identity hunts): XOR of the bits of a tile wave's register image, candidate 0, global step 0
#define CT_SUM4(slot, v4) do { if (bid == 0 && cs.gstep == 0 && is_tw) { \
        atomicXor(&a.status[96 + (slot)], __float_as_int((v4)[0]) ^ __float_as_int((v4)[1]) ^ __float_as_int((v4)[2]) ^ __float_as_int((v4)[3])); } } while (0)
#else
#define CT_STAMP(slot) do { } while (0)
#define CT_SUM4(slot, v4) do { } while (0)
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

    // labels: order entry -> label are two DEPENDENT loads; consumed at once (round 1-5 form: `lab_l[tid] = label[ord[..]]` right here) they
    // held wave 0 — and with it the phase-0 barrier — through two memory round trips before it had requested anything else.  Now the
    // order entry is requested here, the label under the last forward cell, and the LDS store sits in front of the head's barrier.
    int64_t lab_row = 0;
    int lab_r = 0;
    if (tid < Bp && tid < nvalid) {
        const int32_t* ord = cand_order(a.order, g, cd.gidx);
        lab_row = ord ? (int64_t)ord[cs.pos_t + tid] : (int64_t)(cs.base_t + tid);
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
        if (i == L - 1 && tid < Bp && tid < nvalid) lab_r = g.loss_mode == 0 ? a.tab.label[lab_row] : (int)lab_row;   // (mode 1 keeps the table row for the multi-hot targets)
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
    if (tid < Bp) lab_l[tid] = lab_r;
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
    } else if (Cp <= 64 && !g.multitask) {
        // (round 6: <= 64 classes, single task — every NTU / AV-MNIST head — take the eight-lanes-per-row form written for the lean chain:
        //  ~1/3 of the instructions on the serial path; chain_body and chain_split call the same function, so every schedule of the
        //  general chain still rounds identically)
        if (tid < 8 * Bp) softmax_rows_lean<MB>(a, lg_l, SC, red_l, lab_l, nvalid, nf, tid);
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
// search defaults (inner_representation_size 16, main_searchable_ntu.py:26-45).  A 16-wide cell is a string of dependent
// little steps; what bounds a candidate's train step here is the LATENCY of that string (one workgroup, one candidate).
//
// Round 5 form: the chain of a 16-row batch tile lives in the REGISTERS OF ONE WAVE from the first cell to the last dy.
// Every product is computed TRANSPOSED, out_i^T = W_i . out_{i-1}^T (A = the weight tile image, B = the activations): the
// 16x16x4 MFMA's D block — lane (n = batch row, group lg) holds features 4*lg .. 4*lg+3 of that row — is then exactly the B
// operand image of the next cell's four MFMAs (k = 4*lg + q for MFMA q), so a cell hands its output to the next one without a
// shuffle, an LDS round trip or a barrier.  The same holds backwards (d_out^T = W^T . dy^T) and for the head (logits^T, four
// class blocks in 16 registers of the row's four lanes; softmax / CE on those registers with two row-swap butterflies).
// Wave tw < MB ("tile wave") owns batch rows 16*tw .. 16*tw + 15; the tile waves never meet unless BatchNorm needs the batch
// statistics of both tiles (two LDS exchanges per cell forward, one backward).  Before: element-parallel cells over eight
// waves — one LDS hand-off + workgroup barrier per cell and phase, ~1,000-2,000 shader cycles each (profiles/r05_chain_phases_
// session_start.log: 18,700 cycles per step at R = 16, B = 20).
//   * everything the chain needs from global memory (the sweep's partial sums, labels; launch-per-phase: vector block and the
//     weight tiles of every product, staged in LDS) is requested at entry;
//   * the other waves sum the partial slabs of cells 1..L-1 into LDS while the tile waves already compute cell 0 (resident
//     schedule: its units write their slabs TRANSPOSED — MFMA operands swapped, persist.hip.h — so a tile wave's slab item is
//     its own register image; launch-per-phase: the reduced sums are read from LDS with transposed indices);
//   * out_i, dy_i, dlogits and the per-row loss terms are also dropped into LDS (plain writes, nobody waits for them) for what
//     runs AFTER dy is out: chain_lean_tail (statistics, bias / BatchNorm / alpha gradient sums and their Adam updates) and the
//     resident chain's own OUT / HEAD update (lean_res_update).
// The arithmetic is that of chain_body except for the order in which row / column sums (BatchNorm statistics, bias gradients,
// softmax denominators) are accumulated: the two chain forms agree to rounding, not bit for bit; which one runs depends only
// on (R, C, B) — mfas_hip.hip, `lean_chain`.  All schedules of the lean chain (launch-per-phase, fused, resident) run THIS
// code on the same operands and stay bit-identical to each other.
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
#define LEAN_BNF 384       // BatchNorm pair-exchange flags (uint32) inside the scratch: [forward sum | forward squares | backward][parity][tile wave]
#define LEAN_BNV 400       // the cells' batch variance [cell][16] for the running statistics (chain_lean_tail)
#define LEAN_SCR 1024      // floats: forward BN exchange [parity][sum | squares][tile][16] | +256: dgamma / dbeta [L][2][16] | +512: backward BN exchange | +768: [tile][cell][mean | rstd][16]

#ifndef MFAS_RES_TRANSPOSED_SLABS
#define MFAS_RES_TRANSPOSED_SLABS 1     // 0: debug build — the resident units keep the launch-per-phase operand order, the chain transposes in LDS
#endif
// LDS layout shared by chain_lean and the resident helpers
template <int MB>
struct LeanLds {
    static constexpr int Bp = MB * 16, Rp = 16, SX = Rp + 4, sav_plane = MFAS_MAX_CELLS * MB * 256;
    float *xo_l, *dy_l, *lg_l, *rstd_l, *red_l, *yf_l, *vec_l, *lgraw, *scr, *own;
    int* lab_l;
    int nvec, SC;
    __device__ __forceinline__ LeanLds(float* lds, const Geo& g) {
        SC = g.Cp + 4;
        nvec = MFAS_MAX_CELLS * g.vec_cell_stride + g.Cp;
        xo_l = lds;                                          // [L][Bp][SX] out_i of every cell
        dy_l = xo_l + MFAS_MAX_CELLS * Bp * SX;              // [L][Bp][SX] dy_i of every cell
        lg_l = dy_l + MFAS_MAX_CELLS * Bp * SX;              // [Bp][SC] logits -> dlogits
        rstd_l = lg_l + Bp * SC;                             // [L][Rp] (unused since round 5: gamma * rstd stays in registers)
        red_l = rstd_l + MFAS_MAX_CELLS * Rp;                // [2*Bp + 16]
        lab_l = reinterpret_cast<int*>(red_l + 2 * Bp + 16); // [Bp]
        yf_l = reinterpret_cast<float*>(lab_l + Bp);         // [1 or 2][L][MB][256] reduced feature sums
        vec_l = yf_l + (g.alphas ? 2 : 1) * sav_plane;       // [3][nvec] vector block + Adam state
        lgraw = vec_l + ((3 * nvec + 3) & ~3);               // [Bp][64] raw logits of the step (register softmax: the tail's statistics read them)
        scr = lgraw + Bp * 64;                               // cross-wave exchange scratch (LEAN_SCR floats)
        own = scr + LEAN_SCR;        // resident: [W|M|V|T][7 tiles][256], the master copy; otherwise [W|T][7 tiles][256] staged at entry
    }
    static __host__ __device__ constexpr int own_floats() { return 4 * LEAN_OWN_TILES * 256; }
    static __host__ __device__ constexpr int stage_floats() { return 2 * LEAN_OWN_TILES * 256; }
    // tile image of slot s (OUT_{s+1} / head block s-3) as the A operand of the forward products, and its transposed image
    template <bool RES> __device__ __forceinline__ float* wtile(int s) const { return own + s * 256; }
    template <bool RES> __device__ __forceinline__ float* ttile(int s) const { return own + ((RES ? 3 : 1) * LEAN_OWN_TILES + s) * 256; }
};

// own-tile plane offsets of slot s: OUT_{s+1} (s < 3), head class block s-3
__device__ __forceinline__ int64_t lean_own_off(const CandDev& cd, int s) { return s < 3 ? cd.seg_off[s + 1][2] : cd.head_off + ((int64_t)(s - 3) << 8); }
__device__ __forceinline__ int64_t lean_own_toff(const CandDev& cd, int s) { return s < 3 ? cd.outT_off[s + 1] : cd.headT_off + ((int64_t)(s - 3) << 8); }
__device__ __forceinline__ bool lean_own_live(const CandDev& cd, const Geo& g, int s) { return s < 3 ? (s + 1 < cd.L) : (s - 3 < g.ncb); }

// max over the 4 lane groups that share (lane & 15) (colsum's butterflies, common.hip.h)
__device__ __forceinline__ float colmax(float x) {
    int xi = __float_as_int(x);
    const auto r16 = __builtin_amdgcn_permlane16_swap(xi, xi, false, false);
    x = fmaxf(__int_as_float(r16[0]), __int_as_float(r16[1]));
    xi = __float_as_int(x);
    const auto r32 = __builtin_amdgcn_permlane32_swap(xi, xi, false, false);
    return fmaxf(__int_as_float(r32[0]), __int_as_float(r32[1]));
}
__device__ __forceinline__ f32x4 row_sum16_4(f32x4 v) {
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = row_sum<16>(v[q]);
    return v;
}
// The lean chain's activations on the 4 features a lane holds of its batch row (nl is wave-uniform: one branch, four independent
// strings).  Sigmoid: 1 / (1 + 2^(-y log2 e)) on v_exp_f32 and v_rcp_f32 — absolute error <= 1.5e-7 (an activation is an absolute
// quantity: it is summed with O(1) terms by the next product), 4 instructions per element where expf() and the IEEE division take
// ~35 in one dependent string; +-inf and NaN behave as in the library form (rcp(inf) = 0).  ReLU / LeakyReLU: act_fwd's.
__device__ __forceinline__ f32x4 act_fwd4_lean(const f32x4 y, const int nl) {
    f32x4 v;
    if (nl == 1) {
        const f32x4 t = y * -1.44269504088896341f;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t[q]));
    } else if (nl == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = y[q] <= 0.0f ? 0.0f : y[q];
    } else {
        const f32x4 s = y * 0.01f;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = y[q] > 0.0f ? y[q] : s[q];
    }
    return v;
}
__device__ __forceinline__ f32x4 act_bwd4_lean(const f32x4 a, const f32x4 da, const int nl) {
    f32x4 d;
    if (nl == 1) {
        d = da * ((f32x4)(1.0f) - a) * a;
    } else if (nl == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) d[q] = a[q] <= 0.0f ? 0.0f : da[q];
    } else {
        const f32x4 s = da * 0.01f;
#pragma unroll
        for (int q = 0; q < 4; ++q) d[q] = a[q] > 0.0f ? da[q] : s[q];
    }
    return d;
}
// dropout keep-bits of a tile-wave lane's 4 features in every cell of train step `gstep`: bit 4 * cell + q (oracle/np_oracle.py:
// dropout_keep).  The resident chain computes step t + 1's bits after it has published step t's dy (persist.hip.h).
template <int MB>
__device__ __forceinline__ uint32_t lean_keep_bits(const CandDev& cd, const Geo& g, const int gstep) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (!g.use_drop) return 0xFFFFu;
    uint32_t bits = 0u;
    if (wave < MB) {
        const uint32_t h0 = lowbias32(cd.drop_seed + 0x9E3779B9U * (uint32_t)(gstep + 1));
        const uint32_t idx0 = (uint32_t)((wave * 16 + (lane & 15)) * g.R + 4 * (lane >> 4));
#pragma unroll
        for (int i = 0; i < MFAS_MAX_CELLS; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (drop_keep(h0, i, idx0 + q, g.drop_thr)) bits |= 1u << (4 * i + q);
    }
    return bits;
}
// label (loss_mode 1: table row) of a tile-wave lane's batch row in the batch at sample-order position pos_t; the resident chain
// fetches step t + 1's after it has published step t's dy — a dependent pair of loads that must not sit in front of the chain
template <int MB>
__device__ __forceinline__ int lean_label(const ChainArgs& a, const CandDev& cd, const int64_t pos_t, const int base_t, const int nvalid) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b_row = (wave < MB ? wave : 0) * 16 + (lane & 15);
    int lab = 0;
    if (wave < MB && b_row < nvalid) {
        const int32_t* ord = cand_order(a.order, a.g, cd.gidx);
        const int64_t row = ord ? (int64_t)ord[pos_t + b_row] : (int64_t)(base_t + b_row);
        lab = a.g.loss_mode == 0 ? a.tab.label[row] : (int)row;   // mode 1 keeps the table row for the multi-hot targets
    }
    return lab;
}
// What a chain workgroup's thread needs of its candidate's record, read ONCE per launch: inside the resident step loop the compiler
// cannot hoist these loads itself (the loop stores to global memory), and as four dependent vector loads in front of the first
// slab request they cost ~2,000 cycles per step (profiles/r05_chain_phases.log).
struct LeanPre {
    int32_t L, nlbits;          // cells, 2 bits of non-linearity per cell
    int32_t pi, ns, nch;        // this wave's slab duty: cell, S chunks, S + V chunks (0: none)
    int64_t part;               // step-buffer index of this thread's first slab item
    int64_t step_off, vec_off;  // the candidate's step buffers / vector block
    int32_t gidx;
};
template <int MB>
__device__ __forceinline__ LeanPre lean_pre(const ChainArgs& a, const int bid) {
    const CandDev& cd = a.cands[bid];
    const int tid = threadIdx.x;
    constexpr int per_cell = MB * 64;   // float4 partial-sum items per cell; L * per_cell <= 512: one item per thread
    LeanPre lp;
    // (wave-uniform by construction; said so explicitly, or every `i < L` / activation switch becomes an exec-masked region)
    lp.L = __builtin_amdgcn_readfirstlane(cd.L);
    int nlbits = 0;
#pragma unroll
    for (int i = 0; i < MFAS_MAX_CELLS; ++i) nlbits |= (cd.conf[i][2] & 3) << (2 * i);
    lp.nlbits = __builtin_amdgcn_readfirstlane(nlbits);
    const bool has_item = tid < lp.L * per_cell;
    // (per_cell = MB * 64: the cell index is wave-uniform; wave w sums the slabs of cell w / MB for batch tile w % MB — the tile
    //  waves hold cell 0's)
    lp.pi = __builtin_amdgcn_readfirstlane(has_item ? tid / per_cell : 0);
    const int pit = tid - lp.pi * per_cell;
    lp.ns = __builtin_amdgcn_readfirstlane(cd.nch_s[lp.pi]);
    lp.nch = __builtin_amdgcn_readfirstlane(has_item ? lp.ns + cd.nch_v[lp.pi] : 0);
    lp.step_off = cd.step_off; lp.vec_off = cd.vec_off; lp.gidx = __builtin_amdgcn_readfirstlane(cd.gidx);
    lp.part = lp.step_off + a.g.sb_part + (((int64_t)cd.part_cell_off[lp.pi] * MB) << 8) + pit * 4;
    return lp;
}
// does this population's loss run on the tile waves' registers (softmax / CE, single task, no external gradient)?
__device__ __forceinline__ bool lean_ce_in_regs(const ChainArgs& a) { return a.g.loss_mode == 0 && !a.g.multitask && !a.dlogits_in; }

// MODE 0: launch-per-phase schedule; 2: persistent AND resident (LeanRes): vector block, OUT / HEAD weights and statistics live
// on chip, the chain's only global traffic per step is the sweep's partial sums in and dy (+ alpha scales) out; `keep_pre` = the
// step's dropout keep-bits (lean_keep_bits), computed a step ahead.
// (MODE 1 — persistent, everything exchanged through memory — went with the streaming persistent form in round 3.)
// PLAIN: the search default compiled on its own — no BatchNorm, no alphas, softmax CE, no external logits / gradients: the flags
// are compile-time false, their uniform values need no scalar registers and their branches are gone.
template <int MB, int MODE = 0, int PB = 16, int PLAIN = 0>
__device__ __forceinline__ void chain_lean(const ChainArgs& a, const ChainStep& cs, const int bid, float* lds, const LeanPre& lp,
                                           const uint32_t keep_pre = 0u, const int lab_pre = 0) {
    constexpr bool COH = MODE >= 1;
    constexpr bool RES = MODE == 2;
    constexpr bool TS = RES && (MFAS_RES_TRANSPOSED_SLABS != 0);      // the partial slabs arrive transposed (resident units, persist.hip.h)
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
    const int Cp = g.Cp, ncb = g.ncb, R = g.R, C = g.C, L = lp.L, SC = Cp + 4;
    constexpr int sav_plane = MFAS_MAX_CELLS * MB * 256;
    // (PLAIN: 0 every flag at run time; 1 the search default — no BatchNorm; 2, round 6: `--batchnorm` and nothing else — the generic form
    //  carried 10 spilled registers through the BatchNorm cells' serial path)
    const bool f_bn = PLAIN == 1 ? false : (PLAIN == 2 ? true : (g.bn != 0));
    const bool f_alphas = PLAIN ? false : (g.alphas != 0);
    const LeanLds<MB> ll(lds, g);
    const int nvec = ll.nvec;
    float* xo_l = ll.xo_l;
    float* dy_l = ll.dy_l;
    float* lg_l = ll.lg_l;
    float* red_l = ll.red_l;
    int* lab_l = ll.lab_l;
    float* yf_l = ll.yf_l;
    float* vec_l = ll.vec_l;

    float* W = a.plane;
    float* Mv = a.plane + a.plane_stride;
    float* Vv = Mv + a.plane_stride;
    float* sb = a.stepbuf + lp.step_off;
    const int64_t cvec_off = lp.vec_off;
    const int cgidx = lp.gidx;
    const int nlbits = lp.nlbits;
    const int nvalid = cs.nvalid;
    const float nf = (float)nvalid;
    const int64_t sbo = lp.step_off;
    // tile wave tw = wave < MB: batch row b_row = 16 * tw + l15, features 4 * lg + q in register q
    const bool is_tw = wave < MB;
    const int b_row = (is_tw ? wave : 0) * 16 + l15;
    const bool rowok = b_row < nvalid;
    const int r0 = 4 * lg;

    // ------------------------------------------------------------------ entry: every global read of the chain is
    // requested here, in the order the results are needed (the memory counter retires in order): the sweep's partial
    // sums first, then (launch-per-phase) the vector block and the weight tiles of all products, last the labels
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    constexpr int per_cell = MB * 64;   // float4 partial-sum items per cell; L * per_cell <= 512: one item per thread
    // PB = partial-sum chunks requested per thread before any is consumed (16; 8 in the 128-VGPR co-scheduled builds of k_step)
    const bool has_item = tid < L * per_cell;
    const int pi = lp.pi, ns = lp.ns, nch = lp.nch, pit = tid - lp.pi * per_cell;      // (wave-uniform: lean_pre)
    const int64_t part = lp.part;
    f32x4 p8[PB];
    if constexpr (RES) {
        // resident chain: only the slabs that exist.  The count is handed to the loads as a per-lane value: exec-masked loads in ONE
        // straight-line block, all requested before the first is touched (as a scalar count every load sits in its own branch, and
        // the compiler threads "slab 0 exists" through to its consumption: it waited for slab 0 before requesting slab 1)
        // (transposed slabs: a lane's item is ONE batch row — rows beyond the batch hold zeros nobody needs to move: with B = 20 the
        //  second tile's slab items are 3/4 padding, 37 % of the slab bytes that all cross this one CU's memory pipeline)
        int nch_v = (!TS || (pit >> 6) * 16 + l15 < nvalid) ? nch : 0;
        asm volatile("" : "+v"(nch_v));
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            p8[u] = z4;
            if (u < nch_v) p8[u] = ldc4<COH>(a.stepbuf, part + (((int64_t)u * MB) << 8));
        }
    } else {
        // every load UNCONDITIONAL (indices clamped to something valid): with a statically known number of loads in flight the
        // compiler waits for exactly the ones it consumes instead of draining the whole queue at first use
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            const int uu = u < nch ? u : 0;
            p8[u] = ldc4<COH>(a.stepbuf, part + (((int64_t)uu * MB) << 8));
        }
    }
    const int vi = tid < nvec ? tid : 0;   // nvec = 4 * 96 + Cp <= 448: one element of the vector block per thread
    float vw = 0.f, vm = 0.f, vv = 0.f;
    f32x4 st0 = z4, st1 = z4;              // launch-per-phase: the 14 weight tile images, two 16-byte items per thread
    if constexpr (!RES) {
        vw = W[cvec_off + vi]; vm = Mv[cvec_off + vi]; vv = Vv[cvec_off + vi];
        {
            const int s = wave;            // item tid: slot wave (0..7) = W images 0..6, T image 0
            const bool live = s < LEAN_OWN_TILES ? lean_own_live(cd, g, s) : lean_own_live(cd, g, 0);
            const float* base = s < LEAN_OWN_TILES ? W : a.wt;
            const int64_t off = s < LEAN_OWN_TILES ? lean_own_off(cd, s) : lean_own_toff(cd, 0);
            st0 = ldc4<COH>(base, (live ? off : (s < LEAN_OWN_TILES ? cvec_off : cd.headT_off)) + lane * 4);
            if (!live) st0 = z4;
        }
        {
            const int s = wave + 1;        // item 512 + tid: T images 1..6 (waves 0..5)
            const bool in = s < LEAN_OWN_TILES;
            const bool live = in && lean_own_live(cd, g, in ? s : 0);
            st1 = ldc4<COH>(a.wt, (live ? lean_own_toff(cd, s) : cd.headT_off) + lane * 4);
            if (!live) st1 = z4;
        }
    }
    // labels of this tile wave's rows (launch-per-phase: a dependent pair of loads that nothing needs before the loss)
    const int lab = RES ? lab_pre : lean_label<MB>(a, cd, cs.pos_t, cs.base_t, nvalid);
    // resident chain: the LDS operands of the step — biases, weight tile images, their transposes — are read a phase AHEAD of
    // their use, in three groups (a ds_read in front of each product cost ~120 cycles apiece, three per cell): the cells' here, under
    // the slab loads' latency; the head's under cell 0; the backward's under the softmax
    f32x4 pb4[MFAS_MAX_CELLS], pw4[MFAS_MAX_CELLS], pt4[MFAS_MAX_CELLS], phw[4], phb[4], pht[4];
    if constexpr (RES) {
        if (is_tw) {
#pragma unroll
            for (int i = 0; i < MFAS_MAX_CELLS; ++i) {
                pb4[i] = *reinterpret_cast<const f32x4*>(vec_l + i * g.vec_cell_stride + VEC_B * Rp + r0);
                pw4[i] = i > 0 ? *reinterpret_cast<const f32x4*>(ll.template wtile<RES>(i - 1) + lane * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    }
    // dropout keep-bits of this lane's 4 features in every cell (bit 4 * i + q), with "feature exists" and "row is in the batch"
    // folded in: one select per element zeroes dropped, padded and out-of-batch elements alike, forward and backward
    uint32_t ekeep = RES ? keep_pre : lean_keep_bits<MB>(cd, g, cs.gstep);
    {
        uint32_t m4 = 0u;
#pragma unroll
        for (int q = 0; q < 4; ++q) m4 |= (r0 + q < R && rowok) ? (1u << q) : 0u;
        ekeep &= m4 * 0x1111u;
    }
    const float dscale = g.use_drop ? g.drop_scale : 1.0f;
    // phase 0: sum the sweep's column-chunk partial slabs of (cell pi, tile) in fixed order: S chunks, then V chunks
    f32x4 accS = z4, accV = z4;
    // (per STEP opaque copies of the slab counts: as loop invariants, every `u < nch` below becomes a scalar-register mask computed
    //  before the resident step loop and held — spilled to vector lanes, two v_readlane each use — across it)
    int nch_c = nch, ns_c = ns;
    if constexpr (RES) asm volatile("" : "+s"(nch_c), "+s"(ns_c));
    // (acc += slab as a volatile statement: as a plain fadd, `0 + slab 0` is speculated up into the block that requests slab 0 —
    //  and the wave waits for slab 0 before it requests slab 1: one memory round trip more in front of every step)
    auto slab_add = [](f32x4& acc, const f32x4& p) {
        f32x2 lo = acc.lo, hi = acc.hi;
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(lo) : "v"(p.lo));
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(hi) : "v"(p.hi));
        acc = (f32x4){lo.x, lo.y, hi.x, hi.y};
    };
    if (has_item) {
#pragma unroll
        for (int u = 0; u < PB; ++u)
            if (u < nch_c) {
                if (u < ns_c) slab_add(accS, p8[u]); else slab_add(accV, p8[u]);
            }
        for (int ch0 = PB; ch0 < nch_c; ch0 += PB) {
#pragma unroll
            for (int u = 0; u < PB; ++u)
                if (ch0 + u < nch_c) p8[u] = ldc4<COH>(a.stepbuf, part + (((int64_t)(ch0 + u) * MB) << 8));
#pragma unroll
            for (int u = 0; u < PB; ++u)
                if (ch0 + u < nch_c) {
                    if (ch0 + u < ns_c) accS += p8[u]; else accV += p8[u];
                }
        }
        // resident: cell 0's sums stay in the tile waves' registers (transposed slabs: the item IS the lane's register image)
        if (!(TS && pi == 0)) {
            if (f_alphas) {
                *reinterpret_cast<f32x4*>(yf_l + tid * 4) = accS;
                *reinterpret_cast<f32x4*>(yf_l + sav_plane + tid * 4) = accV;
            } else {
                *reinterpret_cast<f32x4*>(yf_l + tid * 4) = accS + accV;
            }
        }
    }
    if constexpr (!RES) {
        if (tid < 16) reinterpret_cast<uint32_t*>(ll.scr + LEAN_BNF)[tid] = 0u;   // (one step per launch: LDS starts undefined; resident: lean_res_load)
        if (tid < nvec) { vec_l[tid] = vw; vec_l[nvec + tid] = vm; vec_l[2 * nvec + tid] = vv; }
        *reinterpret_cast<f32x4*>(ll.own + (wave << 8) + lane * 4) = st0;
        if (wave + 1 < LEAN_OWN_TILES) *reinterpret_cast<f32x4*>(ll.own + ((LEAN_OWN_TILES + wave + 1) << 8) + lane * 4) = st1;
        lds_barrier();
    } else if constexpr (!TS) {
        lds_barrier();
    }
    const float* vecW = vec_l;
    CT_STAMP(0);

    // reduced feature sums of (cell i, this tile) in the transposed register image: S and V parts
    auto cell_sums = [&](const int i, f32x4& yS, f32x4& yV) {
        const int o = (i * MB + wave) << 8;
        if constexpr (TS) {
            yS = *reinterpret_cast<const f32x4*>(yf_l + o + lane * 4);
            if (f_alphas) yV = *reinterpret_cast<const f32x4*>(yf_l + sav_plane + o + lane * 4);
        } else {
            // the launch-per-phase sweeps write the D image of x . W^T (lane: feature l15, batch rows 4*lg + q): read it transposed
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int at = o + ((((l15 >> 2) << 4) + r0 + q) << 2) + (l15 & 3);
                yS[q] = yf_l[at];
                if (f_alphas) yV[q] = yf_l[sav_plane + at];
            }
        }
    };

    // ------------------------------------------------------------------ forward, in the tile waves' registers
    float* bnw = ll.scr;                                     // BN exchange: [parity][sum | sum of squares][tile][16]
    // BatchNorm with TWO batch tiles (B = 17 .. 32): the column statistics need both tile waves' partial sums.  Round 6: the two waves
    // exchange them PAIRWISE — data, then a sequence number behind it (LDS executes a wave's instructions in order; release / acquire at
    // workgroup scope = a wait for the wave's own LDS operations), the partner polls the number — instead of through workgroup barriers
    // that made the tile waves wait for the six waves summing slabs: 3 barriers per cell were 12 of the 22.6 us of a BatchNorm step.
    // Parity double-buffering of the regions is enough: a wave cannot pass exchange k + 1 before its partner has finished reading k.
    uint32_t* bnf = reinterpret_cast<uint32_t*>(ll.scr + LEAN_BNF);
    const uint32_t bnseq = (uint32_t)cs.gstep * (uint32_t)MFAS_MAX_CELLS + 1u;
    // (the LDS unit executes ONE wave's instructions in issue order — data store, then number store: no wait between them; and a wave's
    //  data read is issued only after its poll has returned the number: compiler barriers keep that order, no s_waitcnt is added)
    auto pair_post = [&](float* slot, const f32x4 val, const int kind, const int i) {
        if (l15 == 0) *as_lds(reinterpret_cast<f32x4*>(slot + r0)) = val;
        asm volatile("" ::: "memory");
        __hip_atomic_store(as_lds(bnf + (kind * 2 + (i & 1)) * 2 + wave), bnseq + (uint32_t)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("" ::: "memory");
    };
    auto pair_wait = [&](const int kind, const int i) {
        const uint32_t* f = bnf + (kind * 2 + (i & 1)) * 2 + (wave ^ 1);
        while (__hip_atomic_load(as_lds(f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != bnseq + (uint32_t)i) { asm volatile("" ::: "memory"); }
        asm volatile("" ::: "memory");
    };
    static_assert(MB <= 2, "pairwise exchange: two tile waves");
    float* gv2 = ll.scr + 256;                               // [cell][dgamma | dbeta][16]
    float* bst = ll.scr + 768 + (is_tw ? wave : 0) * 128;     // this tile wave's copy of every cell's batch mean | rstd: [cell][2][16]
    f32x4 av[MFAS_MAX_CELLS];                                 // activations: all the backward needs again in registers (x-hat is recomputed
                                                              // from the batch statistics, the alpha differences wait in the dead sums plane)
    f32x4 o_prev = z4;
    bool cm[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) cm[q] = r0 + q < R;
#pragma unroll
    for (int i = 0; i < MFAS_MAX_CELLS; ++i) {
        av[i] = z4;
        if (i < L) {      // (workgroup-uniform)
            CT_STAMP(1 + i);
            const int nl = (nlbits >> (2 * i)) & 3;
            const int64_t vb = cvec_off + (int64_t)i * g.vec_cell_stride;
            const int vbl = i * g.vec_cell_stride;
            float sgS = 1.0f, sgV = 1.0f;
            if (f_alphas) {
                const float sg = 1.0f / (1.0f + expf(-vecW[vbl + 5 * Rp]));
                sgS = sg;
                sgV = 1.0f - sg;
                if (tid == 0) {
                    stc1<COH>(sb + g.sb_gsc + i * 2, sgS);
                    stc1<COH>(sb + g.sb_gsc + i * 2 + 1, sgV);
                }
            }
            if constexpr (TS) {
                if (i == 1) lds_barrier();     // the other waves' sums of cells 1..L-1 are in LDS (they got there under cell 0)
            }
            if constexpr (RES && PLAIN != 2) {
                if (i == (L > 1 ? 1 : 0) && is_tw) {      // the head's operands, a phase ahead
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) {
                        phw[cb] = *reinterpret_cast<const f32x4*>(ll.template wtile<RES>(3 + cb) + lane * 4);
                        phb[cb] = *reinterpret_cast<const f32x4*>(vec_l + g.vec_head + (cb < ncb ? cb : 0) * 16 + r0);
                    }
                }
            }
            f32x4 v = z4, s1 = z4;
            if (is_tw) {
                f32x4 yS = accS, yV = accV;
                if (!TS || i > 0) { yS = z4; yV = z4; cell_sums(i, yS, yV); }
                else if (!f_alphas) yS = accS + accV;
                f32x4 acc = yS;
                CT_SUM4(i, acc);
                if (f_alphas) {
                    // (yS - yV) for d(alpha_i): parked in this lane's slot of the (now read) sums plane until the backward
                    *reinterpret_cast<f32x4*>(yf_l + ((i * MB + wave) << 8) + lane * 4) = yS - yV;
                    acc = yS * sgS + yV * sgV;
                }
                if (i > 0) {
                    f32x4 w;
                    if constexpr (RES) w = pw4[i]; else w = *reinterpret_cast<const f32x4*>(ll.template wtile<RES>(i - 1) + lane * 4);
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc = MFMA16(w[q], o_prev[q], acc);
                }
                f32x4 bias;
                if constexpr (RES) bias = pb4[i]; else bias = *reinterpret_cast<const f32x4*>(vecW + vbl + VEC_B * Rp + r0);
                v = act_fwd4_lean(acc + bias, nl);
                if (f_bn) s1 = row_sum16_4(rowok ? v : z4);
            }
            f32x4 z = v;
            if (f_bn) {   // batch statistics over the valid rows: mean, then the variance of the deviations
                float* ex = bnw + (i & 1) * 128;
                f32x4 tot = s1;
                if constexpr (MB > 1) {
                    if (is_tw) {
                        pair_post(ex + wave * 16, s1, 0, i);
                        pair_wait(0, i);
                        tot = *reinterpret_cast<const f32x4*>(ex + r0) + *reinterpret_cast<const f32x4*>(ex + 16 + r0);
                    }
                }
                // (divisions and square root through common.hip.h's correctly rounding packed sequences: the values of operator/ and
                //  sqrtf() for a fraction of the instructions)
                const f32x4 nf4 = (f32x4)(nf), rnf4 = (f32x4)(rcp_refined(nf));
                f32x4 mu = z4, s2 = z4;
                if (is_tw) {
                    mu = div_by4(tot, nf4, rnf4);
                    const f32x4 dlt = v - mu;
                    s2 = row_sum16_4(rowok ? dlt * dlt : z4);
                }
                f32x4 tot2 = s2;
                if constexpr (MB > 1) {
                    if (is_tw) {
                        pair_post(ex + 64 + wave * 16, s2, 1, i);
                        pair_wait(1, i);
                        tot2 = *reinterpret_cast<const f32x4*>(ex + 64 + r0) + *reinterpret_cast<const f32x4*>(ex + 64 + 16 + r0);
                    }
                }
                if (is_tw) {
                    const f32x4 gam = *reinterpret_cast<const f32x4*>(vecW + vbl + VEC_G * Rp + r0);
                    const f32x4 bet = *reinterpret_cast<const f32x4*>(vecW + vbl + VEC_BE * Rp + r0);
                    const f32x4 var = div_by4(tot2, nf4, rnf4);
                    const f32x4 sd = sqrt_rn4(var + g.bn_eps);
                    const f32x4 rstd = div_by4((f32x4)(1.0f), sd, rcp_refined4(sd));
                    const f32x4 xh = (v - mu) * rstd;
                    z = xh * gam + bet;
                    if (l15 == 0) {
                        *reinterpret_cast<f32x4*>(bst + i * 32 + r0) = mu;
                        *reinterpret_cast<f32x4*>(bst + i * 32 + 16 + r0) = rstd;
                    }
                    // (running statistics: chain_lean_tail steps them from this variance and the mean above, after dy is out — as four
                    //  serial element updates on tile wave 0 they delayed BOTH tile waves of every forward cell)
                    if (wave == 0 && l15 == 0) *reinterpret_cast<f32x4*>(ll.scr + LEAN_BNV + i * 16 + r0) = var;
                }
            }
            if (is_tw) {
                av[i] = v;
                const f32x4 zs = z * dscale;
                f32x4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = (ekeep & (1u << (4 * i + q))) ? zs[q] : 0.0f;
                o_prev = o;
                CT_SUM4(4 + i, o);
                // for the updates that run after dy is out (x operand of OUT_{i+1} / HEAD): LDS, and the step buffer when the sweep owns them
                *reinterpret_cast<f32x4*>(xo_l + i * Bp * SX + b_row * SX + r0) = o;
                if constexpr (!RES) stc4<COH>(a.stepbuf, sbo + g.sb_xo + (int64_t)(i * Bp + b_row) * Rp + r0, o);
            }
        }
    }
    CT_STAMP(5);

    // ------------------------------------------------------------------ head: logits^T, class 16 * cb + 4 * lg + q of this lane's row
    f32x4 lgt[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) lgt[cb] = z4;
    if (is_tw) {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
            if (cb < ncb) {
                f32x4 w, hb;
                if constexpr (RES && PLAIN != 2) { w = phw[cb]; hb = phb[cb]; }
                else { w = *reinterpret_cast<const f32x4*>(ll.template wtile<RES>(3 + cb) + lane * 4); hb = *reinterpret_cast<const f32x4*>(vecW + g.vec_head + cb * 16 + r0); }
                f32x4 acc = z4;
#pragma unroll
                for (int q = 0; q < 4; ++q) acc = MFMA16(w[q], o_prev[q], acc);
                lgt[cb] = acc + hb;
            }
    }
    CT_STAMP(6);
    CT_SUM4(8, lgt[0]); CT_SUM4(8, lgt[1]); CT_SUM4(8, lgt[2]); CT_SUM4(8, lgt[3]);
    if constexpr (RES) {
        if (is_tw) {      // the backward's operands, under the loss
#pragma unroll
            for (int i = 0; i < MFAS_MAX_CELLS; ++i)
                pt4[i] = i < MFAS_MAX_CELLS - 1 ? *reinterpret_cast<const f32x4*>(ll.template ttile<RES>(i) + lane * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) if constexpr (PLAIN != 2) pht[cb] = *reinterpret_cast<const f32x4*>(ll.template ttile<RES>(3 + cb) + lane * 4);
        }
    }
    if constexpr (MODE == 0 && !PLAIN) {
        if (a.logits_out) {   // train-mode forward only
            if (is_tw && rowok) {
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int c = cb * 16 + r0 + q;
                        if (c < C) a.logits_out[(int64_t)b_row * C + c] = lgt[cb][q];
                    }
            }
            return;
        }
    }
    if (!PLAIN && a.dlogits_in) {     // the caller's dL/dlogits instead of the loss gradient (rows / classes beyond the batch: 0)
        if (is_tw) {
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = cb * 16 + r0 + q;
                    lgt[cb][q] = (rowok && c < C) ? a.dlogits_in[(int64_t)b_row * C + c] : 0.f;
                }
        }
    } else if (PLAIN || (g.loss_mode == 0 && !g.multitask)) {
        // softmax / cross entropy on the registers (train_searchable/ntu.py:53-61): the row's 64 class slots sit in 16 registers of
        // its four lanes; max and sum cross the lane groups with two row-swap butterflies.  Only what dy needs runs here:
        //   e_c = 2^(x_c log2 e - m),  m = rounded(max_c x_c * log2 e)  (ONE rounded m for the whole row: the probabilities
        //   e_c / sum e do not depend on its rounding; v_exp_f32: absolute error <= 1e-7 of the probability),
        //   dlogits_c = e_c * (1/sum)(1/n) - [c = label] (1/n)   (reciprocals by v_rcp_f32 + one Newton step).
        // The row's loss and top-1 hit — statistics — are computed by chain_lean_tail (after dy is out) from the raw logits parked
        // in LDS and (m, sum e) in red_l.
        if (is_tw) {
            constexpr float L2E = 1.44269504088896341f;
            f32x4 xv[4];
            // classes this lane holds below C, counted from its first: per STEP an opaque value — as a loop invariant the compiler
            // turns the 16 tests into 16 scalar-register masks held (spilled) across the whole step loop
            int nq = C - r0;
            asm volatile("" : "+v"(nq));
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                *reinterpret_cast<f32x4*>(ll.lgraw + b_row * 64 + cb * 16 + r0) = lgt[cb];
                xv[cb] = lgt[cb];
                if (cb >= ncb - 1) {      // (only the last class block has padding; the blocks behind it do not exist)
#pragma unroll
                    for (int q = 0; q < 4; ++q) xv[cb][q] = (cb * 16 + q < nq) ? lgt[cb][q] : -3.0e38f;
                }
            }
            float mxl = fmaxf(fmaxf(xv[0][0], xv[0][1]), fmaxf(xv[0][2], xv[0][3]));
#pragma unroll
            for (int cb = 1; cb < 4; ++cb) mxl = fmaxf(mxl, fmaxf(fmaxf(xv[cb][0], xv[cb][1]), fmaxf(xv[cb][2], xv[cb][3])));
            const float mx = colmax(mxl);
            const float mneg = -(mx * L2E);
            f32x4 ev[4];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                const f32x4 t = vfma4(xv[cb], (f32x4)(L2E), (f32x4)(mneg));
#pragma unroll
                for (int q = 0; q < 4; ++q) ev[cb][q] = __builtin_amdgcn_exp2f(t[q]);
            }
            const f32x4 s4 = (ev[0] + ev[1]) + (ev[2] + ev[3]);
            const float se = colsum((s4[0] + s4[1]) + (s4[2] + s4[3]));
            if (lg == 0) {
                red_l[b_row] = mneg;
                red_l[Bp + b_row] = se;
                lab_l[b_row] = lab;
            }
            const float rnf = rowok ? rcp_refined(nf) : 0.0f;
            const f32x4 k4 = (f32x4)(rcp_refined(se) * rnf);
            const int jown = (((lab >> 2) & 3) == lg) ? (((lab >> 4) << 2) | (lab & 3)) : -1;
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                f32x4 oh;
#pragma unroll
                for (int q = 0; q < 4; ++q) oh[q] = (4 * cb + q == jown) ? -rnf : 0.0f;
                lgt[cb] = vfma4(ev[cb], k4, oh);
            }
        }
    } else {
        // multitask CE / weighted BCE: the LDS forms over all eight waves (logits to LDS, gradient back to the registers)
        if (is_tw) {
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
                if (cb < ncb) *reinterpret_cast<f32x4*>(lg_l + b_row * SC + cb * 16 + r0) = lgt[cb];
            if (lg == 0) lab_l[b_row] = lab;
        }
        lds_barrier();
        if (g.loss_mode == 1) {
            if (tid < 4 * Bp) bce_rows(lg_l, SC, red_l, Bp, lab_l, a.tab.multilabel, a.pos_w, C, Cp, nvalid, tid);
        } else if (tid < LPR * Bp) {
            softmax_rows<MB>(a, cs, lg_l, SC, red_l, lab_l, nvalid, nf, tid, cand_order(a.order, g, cgidx));
        }
        lds_barrier();
        if (is_tw) {
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
                if (cb < ncb) lgt[cb] = *reinterpret_cast<const f32x4*>(lg_l + b_row * SC + cb * 16 + r0);
        }
    }
    // dlogits for the tail (head-bias gradient) and the HEAD update
    if (is_tw) {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
            if (cb < ncb) {
                *reinterpret_cast<f32x4*>(lg_l + b_row * SC + cb * 16 + r0) = lgt[cb];
                if constexpr (!RES) stc4<COH>(a.stepbuf, sbo + g.sb_dlog + (int64_t)b_row * Cp + cb * 16 + r0, lgt[cb]);
            }
    }
    CT_STAMP(7);
    CT_SUM4(9, lgt[0]); CT_SUM4(9, lgt[1]); CT_SUM4(9, lgt[2]); CT_SUM4(9, lgt[3]);
    // ------------------------------------------------------------------ backward, in the same registers
    f32x4 dy_next = z4;
#pragma unroll
    for (int i = MFAS_MAX_CELLS - 1; i >= 0; --i) {
        if (i < L) {
            CT_STAMP(8 + (L - 1 - i));
            const int nl = (nlbits >> (2 * i)) & 3;
            const bool from_head = (i == L - 1);
            f32x4 d = z4, p0 = z4, p1 = z4, xh = z4;
            if (is_tw) {
                f32x4 acc = z4;
                if (from_head) {
                    // d_out^T = Wc^T . dlogits^T: one 4-MFMA string per class block, summed (0 + 2) + (1 + 3)
                    f32x4 pt[4];
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) {
                        pt[cb] = z4;
                        if (cb < ncb) {
                            f32x4 wt4;
                            // (the BatchNorm-only build reads the head's transposed tiles where they are used: 16 registers less, no spills)
                            if constexpr (RES && PLAIN != 2) wt4 = pht[cb]; else wt4 = *reinterpret_cast<const f32x4*>(ll.template ttile<RES>(3 + cb) + lane * 4);
#pragma unroll
                            for (int q = 0; q < 4; ++q) pt[cb] = MFMA16(wt4[q], lgt[cb][q], pt[cb]);
                        }
                    }
                    acc = (pt[0] + pt[2]) + (pt[1] + pt[3]);
                } else {
                    f32x4 w;
                    if constexpr (RES) w = pt4[i]; else w = *reinterpret_cast<const f32x4*>(ll.template ttile<RES>(i) + lane * 4);
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc = MFMA16(w[q], dy_next[q], acc);
                }
                const f32x4 ds = acc * dscale;
#pragma unroll
                for (int q = 0; q < 4; ++q) d[q] = (ekeep & (1u << (4 * i + q))) ? ds[q] : 0.0f;
                if (f_bn) {
                    xh = (av[i] - *reinterpret_cast<const f32x4*>(bst + i * 32 + r0)) * *reinterpret_cast<const f32x4*>(bst + i * 32 + 16 + r0);
                    p0 = row_sum16_4(d);
                    p1 = row_sum16_4(d * xh);
                }
            }
            f32x4 dz = d;
            if (f_bn) {
                float* ex = bnw + 512 + (i & 1) * 128;     // (not the forward's region: no barrier separates the last forward cell's reads from these writes)
                f32x4 dbet = p0, dgam = p1;
                if constexpr (MB > 1) {
                    if (is_tw) {
                        if (l15 == 0) *reinterpret_cast<f32x4*>(ex + 64 + wave * 16 + r0) = p1;
                        pair_post(ex + wave * 16, p0, 2, i);
                        pair_wait(2, i);
                        dbet = *reinterpret_cast<const f32x4*>(ex + r0) + *reinterpret_cast<const f32x4*>(ex + 16 + r0);
                        dgam = *reinterpret_cast<const f32x4*>(ex + 64 + r0) + *reinterpret_cast<const f32x4*>(ex + 64 + 16 + r0);
                    }
                }
                if (is_tw) {
                    const f32x4 gr = *reinterpret_cast<const f32x4*>(vecW + i * g.vec_cell_stride + VEC_G * Rp + r0) *
                                     *reinterpret_cast<const f32x4*>(bst + i * 32 + 16 + r0);
                    const f32x4 nf4 = (f32x4)(nf), rnf4 = (f32x4)(rcp_refined(nf));
                    const f32x4 k1 = div_by4(dbet, nf4, rnf4), k2 = div_by4(dgam, nf4, rnf4);
                    const f32x4 da = gr * (d - k1 - xh * k2);
#pragma unroll
                    for (int q = 0; q < 4; ++q) dz[q] = (rowok && cm[q]) ? da[q] : 0.f;
                    if (wave == 0 && l15 == 0) {
                        *reinterpret_cast<f32x4*>(gv2 + (i * 2 + 0) * 16 + r0) = dgam;
                        *reinterpret_cast<f32x4*>(gv2 + (i * 2 + 1) * 16 + r0) = dbet;
                    }
                }
            }
            if (is_tw) {
                const f32x4 dy = act_bwd4_lean(av[i], dz, nl);
                dy_next = dy;
                CT_SUM4(10 + i, dy);
                CT_SUM4(14 + i, d);
                // dy_i -> step buffer (dy operand of the sweep) as soon as it exists; LDS copy for the tail / the own updates
                stc4<COH>(a.stepbuf, sbo + g.sb_dy + (int64_t)(i * Bp + b_row) * Rp + r0, dy);
                *reinterpret_cast<f32x4*>(dy_l + i * Bp * SX + b_row * SX + r0) = dy;
                // d(alpha_i) needs sum_{b,r} dy[b,r] * (yS_raw - yV_raw)[b,r]: the products go to the (now dead) V plane of the
                // reduced feature sums, summed in fixed order by chain_lean_tail
                if (f_alphas) *reinterpret_cast<f32x4*>(yf_l + sav_plane + (i * Bp + b_row) * 16 + r0) = dy * *reinterpret_cast<const f32x4*>(yf_l + ((i * MB + wave) << 8) + lane * 4);
            }
        }
    }
    CT_STAMP(12);
    if constexpr (!RES) lds_barrier();     // (the resident loop's publish barrier follows the call)
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
    if (g.bn && tid < L * 16 && (tid & 15) < R) {      // running statistics: momentum 0.1, unbiased variance (torch BatchNorm1d, train mode)
        const int i = tid >> 4, c = tid & 15, vbl = i * g.vec_cell_stride;
        const float mu = ll.scr[768 + i * 32 + c], var = ll.scr[LEAN_BNV + i * 16 + c], nf = (float)cs.nvalid;
        float rm = vecW[vbl + VEC_RM * Rp + c], rv = vecW[vbl + VEC_RV * Rp + c];
        const float unb = var * (nf / (nf - 1.0f));
        rm += g.bn_mom * (mu - rm);
        rv += g.bn_mom * (unb - rv);
        if constexpr (RES) { vec_l[vbl + VEC_RM * Rp + c] = rm; vec_l[vbl + VEC_RV * Rp + c] = rv; }
        else { a.plane[cvec_off + vbl + VEC_RM * Rp + c] = rm; a.plane[cvec_off + vbl + VEC_RV * Rp + c] = rv; }
    }
    if (a.logits_out) return;       // train-mode forward only: no statistics, no update
    if (lean_ce_in_regs(a)) {
        // the register softmax left (m, sum e) per row in red_l and the raw logits in LDS: row loss = ln(sum_c e^x_c) - x_label and
        // the top-1 hit (first maximum = torch.max(dim=1)), 16 lanes per row
        float* red_w = ll.red_l;
        float ls = 0.f, hit = 0.f;
        const int b = tid >> 4, sub = tid & 15;
        if (tid < Bp * 16) {
            const float* row = ll.lgraw + b * 64;
            const int lab = ll.lab_l[b];
            float bv = -3.0e38f;
            int bi = 0x7FFFFFFF;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = sub + 16 * j;
                if (c < C) {
                    const float t = row[c];
                    if (t > bv) { bv = t; bi = c; }
                }
            }
            row_argmax<16>(bv, bi);
            const bool ok = b < cs.nvalid;
            ls = ok ? (__builtin_amdgcn_logf(red_l[Bp + b]) - red_l[b]) * 0.693147180559945309f - row[lab] : 0.f;
            hit = (ok && bi == lab) ? 1.f : 0.f;
        }
        lds_barrier();
        if (tid < Bp * 16 && sub == 0) { red_w[b] = ls; red_w[Bp + b] = hit; }
        lds_barrier();
    }
    if (tid == CHAIN_THREADS - 64 && (RES || a.stats)) {
        float ls = 0.f, ncor = 0.f;
#pragma unroll 1
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
        // (rolled loops in the lean tail: unrolled, their 32 loop-invariant LDS addresses are hoisted out of the resident step loop
        //  and live — spilled — across it)
#pragma unroll 1
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
#pragma unroll 1
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
    if (tid < 16) reinterpret_cast<uint32_t*>(ll.scr + LEAN_BNF)[tid] = 0u;       // BatchNorm pair-exchange flags (chain_lean): sequence numbers start at 1
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

// ------------------------------------------------------------------------------------------------
// chain_split<NS> — the general chain of ONE candidate on NS compute units (round 6; R = 113 .. 128: eight row blocks, B <= 16: one
// batch tile, C <= 64, no alphas).  /root/reference/models/search/ntu_searchable.py:228-242 is serial in the cells, and at R = 128 a
// cell is 256 f32 MFMAs + ~300 VALU instructions per row-block wave: on ONE CU (chain_body) the eight row-block waves share four
// SIMDs and sit in the same phase between barriers — 6-9 k shader cycles per cell, 84 k per step (profiles/r05_chain_wide_r128.log).
// A cell is column-separable: Linear, activation, BatchNorm batch statistics (per column over the batch rows one wave holds),
// dropout and their backward need nothing from other columns, so workgroup `part` of the candidate owns row blocks
// part * 8 / NS .. (one wave per row block, each on its own SIMD), and only the 16 x 128 block a cell hands to the next product
// (out_i forward, dy_i backward) crosses CUs — seven exchanges per step:
//   * exchange area xch[parity][slot][row block][lane][4] (global, per candidate) in the MFMA D image: a producer's lane stores its
//     four values as ONE write-through 16-byte piece; every consumer thread polls ITS piece with sc1 loads until it no longer holds
//     the sentinel (all-ones words: not a value the arithmetic produces), then drops it into the LDS operand buffer.  No flags, no
//     store drains: a hand-off is one store -> load round trip.  Parities alternate per chain launch; a launch resets the pieces it
//     owns in the OTHER parity (the kernel boundary orders the reset before the next launch's data);
//   * the head (16 x 128 x 64) and the softmax are REPLICATED on every part (class block c on the wave pair (c, c + 4): even / odd
//     k-blocks — chain_body's two MFMA chains — four tiles per wave held in registers since entry, the odd sums cross through LDS): no
//     exchange for logits / dlogits; statistics, dlogits, the head bias belong to part 0;
//   * a product's odd-k MFMA chain runs on a HELPER wave (main wave w: even k-blocks on top of the feature sums; wave w + NRO: odd
//     k-blocks from zero; handed over through LDS at a barrier every wave of the part passes: the same two sums chain_body adds);
//   * the weight tiles of the next product are staged global -> registers -> LDS (ONE buffer: read before a product's barrier, written
//     behind it) by all eight waves while the main waves compute; saved activations / x-hat of the own row blocks stay in LDS; labels
//     and the sentinel reset sit at the END of the entry, the slab sums' loads all in flight at once; 66 KB of LDS and 128 registers,
//     so two workgroups of the launch (its dynamic LDS size is also the sweep units') still share a CU;
//   * per-cell "dy is out" flags become arrival COUNTERS (every part adds 1 behind its exchange poll — whose returned load implies
//     the wave's earlier stores were acknowledged — so no extra drain; target = NS * (step + 1)).
// The arithmetic per row block is chain_body's (same products in the same even / odd MFMA chains, same reductions): every schedule
// stays bit-identical (tests/test_gpu_parity.py::test_same_group_launch_fuzz_bit_identical, ::test_full_size_properties,
// ::test_chain_split_bit_identical).
// ------------------------------------------------------------------------------------------------
#define XCH_SLOTS 7
#define XCH_SENT 0xFFFFFFFFu
#define XCH_SPIN_LIMIT (1u << 21)
#define XCH_CAND_FLOATS (2 * XCH_SLOTS * 8 * 256)

__device__ __forceinline__ f32x4 xch_wait(const float* base, const int64_t idx, int32_t* status, const int gidx) {
    uint32_t spins = 0;
    for (;;) {
        const f32x4 v = ldc4<true>(base, idx);
        const u32x4 u = __builtin_bit_cast(u32x4, v);
        if (u[0] != XCH_SENT && u[1] != XCH_SENT && u[2] != XCH_SENT && u[3] != XCH_SENT) return v;
        if (++spins > XCH_SPIN_LIMIT) { atomicMax(&status[gidx], 2); return v; }   // (the host reports the lost dependency)
        asm volatile("" ::: "memory");          // the load is re-issued every turn
        __builtin_amdgcn_s_sleep(1);
    }
}

// LDS floats of chain_split<NS> (host: launch size)
template <int NS> __host__ __device__ constexpr size_t chain_split_lds_floats(int Rp, int Cp) {
    return (size_t)2 * 16 * (Rp + 4) + (size_t)16 * (Cp + 4) + (size_t)MFAS_MAX_CELLS * (8 / NS) * 16 + 48 + 16 +
           (size_t)MFAS_MAX_CELLS * (8 / NS) * 256 + (size_t)(MFAS_MAX_CELLS * 5 * (8 / NS) * 16 + Cp) + (size_t)(8 / NS) * 8 * 256 +
           (size_t)2 * MFAS_MAX_CELLS * (8 / NS) * 256 + (size_t)(8 / NS) * 256;
}

template <int NS>
__device__ __forceinline__ void chain_split(const ChainArgs& a, const ChainStep& cs, const int bid, const int part, float* lds) {
#ifdef MFAS_CHAIN_TIMING
    const unsigned long long ct0 = __builtin_readcyclecounter();
#define CS_STAMP(slot) do { if (threadIdx.x == 0 && bid == 0 && part == 0 && cs.gstep == 3) a.status[64 + (slot)] = (int32_t)(__builtin_readcyclecounter() - ct0); } while (0)
    // absolute time (100 MHz): entry / end of the chain in steps 3 and 4 -> the step period and what of it is the chain
    if (threadIdx.x == 0 && bid == 0 && part == 0 && (cs.gstep == 3 || cs.gstep == 4)) a.status[64 + 23 + 2 * (cs.gstep - 3)] = (int32_t)wall_clock64();
#else
#define CS_STAMP(slot) do { } while (0)
#endif
    constexpr int NRO = 8 / NS;                  // row blocks (= main waves) of this part
    constexpr int Bp = 16, MB = 1;
    const CandDev& cd = a.cands[bid];
    const Geo& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int Rp = g.Rp, nrb = 8, Cp = g.Cp, ncb = g.ncb, R = g.R, C = g.C, L = cd.L;
    const int SX = Rp + 4, SC = Cp + 4;
    constexpr int LPR = 16;
    float* xo_l = lds;                                   // [2][16][SX] out_i ping-pong; backward: dy_i
    float* dy_l = xo_l;
    float* lg_l = xo_l + 2 * Bp * SX;                    // [16][SC]
    float* rstd_l = lg_l + Bp * SC;                      // [L][own columns]
    float* red_l = rstd_l + MFAS_MAX_CELLS * (8 / NS) * 16;   // [48]
    int* lab_l = reinterpret_cast<int*>(red_l + 48);     // [16]
    float* yf_l = reinterpret_cast<float*>(lab_l + 16);  // [L][NRO][256]  this part's reduced feature sums
    constexpr int NCOL = NRO * 16;                       // this part's columns
    float* vl = yf_l + MFAS_MAX_CELLS * NRO * 256;       // [L][5 kinds][NCOL] | [Cp] head bias: the W plane (Adam moments: read where they are stepped)
    const int nvec = MFAS_MAX_CELLS * 5 * NCOL;
    const int vplane = nvec + Cp;
    // ONE tile buffer: a product's tiles are read (main + helper waves) BEFORE the product's barrier, the next product's are written
    // behind it — and everything is kept under 80 KB so that two workgroups of the launch (its dynamic LDS size is also the sweep units')
    // still share a CU
    float* tb = vl + vplane;                             // [NRO * 8][256] weight tiles of the current product
    float* sv_l = tb + NRO * 8 * 256;                    // [2: act, x-hat][L][NRO][256] saved for the backward pass (chain_body: L2 scratch)
    float* odd_l = sv_l + 2 * MFAS_MAX_CELLS * NRO * 256; // [NRO][256] the odd k-blocks' MFMA chain of a product, computed by the helper wave

    float* W = a.plane;
    float* Mv = a.plane + a.plane_stride;
    float* Vv = Mv + a.plane_stride;
    float* sb = a.stepbuf + cd.step_off;
    const int64_t cvec_off = cd.vec_off;
    int nlbits = 0;
#pragma unroll
    for (int i = 0; i < MFAS_MAX_CELLS; ++i) nlbits |= (cd.conf[i][2] & 3) << (2 * i);
    const int cgidx = cd.gidx;
    asm volatile("" :: "s"(nlbits), "s"(cgidx));
    CS_STAMP(34);
    // the candidate record's offsets the cell loops need, read ONCE (behind a barrier a field of `cd` is a fresh scalar load: a few hundred
    // cycles in front of every product's tile requests)
    const int64_t oP1 = cd.seg_off[1][2], oP2 = cd.seg_off[2][2], oP3 = cd.seg_off[3][2];
    const int64_t oT1 = cd.outT_off[1], oT2 = cd.outT_off[2], oT3 = cd.outT_off[3], oHT = cd.headT_off;
    auto offP = [&](const int i) { return i == 1 ? oP1 : (i == 2 ? oP2 : oP3); };
    auto offT = [&](const int i) { return i == 1 ? oT1 : (i == 2 ? oT2 : oT3); };
    const int nvalid = cs.nvalid;
    const float nf = (float)nvalid;
    const AdamC ac = adam_consts(a.ac, cs.ss, cs.bc2s);
    const uint32_t h0 = lowbias32(cd.drop_seed + 0x9E3779B9U * (uint32_t)(cs.gstep + 1));
    const int64_t sbo = cd.step_off;
    const int rb0 = part * NRO;                          // first own row block
    const bool main_w = wave < NRO;
    const int rb = rb0 + (main_w ? wave : 0);            // a main wave's row block
    // exchange area of this candidate: this launch's parity / the other one
    const int64_t xq = (int64_t)cgidx * XCH_CAND_FLOATS;
    const int64_t xcur_par = xq + (int64_t)a.xpar * (XCH_SLOTS * 8 * 256);
    const int64_t xoth_par = xq + (int64_t)(a.xpar ^ 1) * (XCH_SLOTS * 8 * 256);

    // ---- entry: everything that does not depend on the step's data is requested first
    // the head's tiles: class block (wave - 4) of every part, in registers until the head product
    // (wave w: class block w & 3, the even (w < 4) or odd (w >= 4) k-blocks — chain_body's two MFMA chains on two waves; four tiles =
    //  16 registers per wave: eight cost the 128-register build spills)
    f32x4 hw[4];
    const int hcb = wave & 3, hpar = wave >> 2;
    const bool head_w = hcb < ncb;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        hw[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (head_w) hw[u] = *reinterpret_cast<const f32x4*>(W + cd.head_off + ((int64_t)hcb * nrb + 2 * u + hpar) * 256 + lane * 4);
    }
    // tile staging: the NRO * 8 tiles of a product (own row blocks x k-blocks; fewer k-blocks for the head's transpose), wave w takes
    // tiles w, w + 8, ...: tile e = (own row block e / nk, k-block e % nk), source = base + (rb0 + e / nk) * nk * 256 + (e % nk) * 256
    f32x4 stg[NRO];
    auto stage_issue = [&](const float* base, const int64_t off0, const int nk) {
#pragma unroll
        for (int u = 0; u < NRO; ++u) {
            const int e = wave + 8 * u;
            const int j = e / nk, k = e - j * nk;
            if (e < NRO * nk) stg[u] = *reinterpret_cast<const f32x4*>(base + off0 + ((int64_t)(rb0 + j) * nk + k) * 256 + lane * 4);
        }
    };
    auto stage_store = [&](float* dst, const int nk) {
#pragma unroll
        for (int u = 0; u < NRO; ++u) {
            const int e = wave + 8 * u;
            const int j = e / nk, k = e - j * nk;
            if (e < NRO * nk) *as_lds(reinterpret_cast<f32x4*>(dst + (j * 8 + k) * 256 + lane * 4)) = stg[u];
        }
    };
    if (L > 1) stage_issue(W, oP1, nrb);        // P_1
    // vector block (W plane): own columns of every cell, head bias — REQUESTED here (two elements per thread cover 4 x 5 x NCOL <= 1024),
    // dropped into LDS behind the slab sums below: every load of the entry is in flight at once (one round trip, not one per loop)
    static_assert(MFAS_MAX_CELLS * 5 * (8 / NS) * 16 <= 2 * CHAIN_THREADS, "two vector elements per thread");
    float vreg[2] = {0.f, 0.f}, vbias = 0.f;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int e = tid + u * CHAIN_THREADS;
        if (e < nvec) {
            const int i = e / (5 * NCOL), r5 = e - i * (5 * NCOL), kind = r5 / NCOL, c = r5 - kind * NCOL;
            vreg[u] = W[cvec_off + (int64_t)i * g.vec_cell_stride + kind * Rp + rb0 * 16 + c];
        }
    }
    if (tid < Cp) vbias = W[cvec_off + g.vec_head + tid];
    // reduced feature sums of the own row blocks: [L][NRO][64 lanes] float4 items
    {
        const int n = L * NRO * 64;
        if (a.yf_reduced) {
            const f32x4* src = reinterpret_cast<const f32x4*>(sb + g.sb_yf);
            for (int e = tid; e < n; e += CHAIN_THREADS) {
                const int i = e / (NRO * 64), r = e - i * (NRO * 64);
                *as_lds(reinterpret_cast<f32x4*>(yf_l + (int64_t)e * 4)) = src[(i * nrb + rb0) * 64 + r];
            }
        } else {
            for (int e = tid; e < n; e += CHAIN_THREADS) {
                const int i = e / (NRO * 64), r = e - i * (NRO * 64);
                const int ns = cd.nch_s[i], nch = ns + cd.nch_v[i];
                const int64_t pbase = sbo + g.sb_part + (((int64_t)cd.part_cell_off[i] * nrb * MB) << 8) + ((int64_t)rb0 * 64 + r) * 4;
                f32x4 accS = {0.f, 0.f, 0.f, 0.f}, accV = {0.f, 0.f, 0.f, 0.f};
                constexpr int PB = 10;      // (conf 4 at 256-column chunks: <= 10 slabs per cell — ONE round of loads; 12 spill in the 128-register build)
                for (int ch0 = 0; ch0 < nch; ch0 += PB) {
                    f32x4 p8[PB];
#pragma unroll
                    for (int u = 0; u < PB; ++u)
                        if (ch0 + u < nch) p8[u] = ldc4<true>(a.stepbuf, pbase + (((int64_t)(ch0 + u) * nrb * MB) << 8));
#pragma unroll
                    for (int u = 0; u < PB; ++u)
                        if (ch0 + u < nch) { if (ch0 + u < ns) accS += p8[u]; else accV += p8[u]; }
                }
                *as_lds(reinterpret_cast<f32x4*>(yf_l + (int64_t)e * 4)) = accS + accV;
            }
        }
    }
    CS_STAMP(35);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int e = tid + u * CHAIN_THREADS;
        if (e < nvec) vl[e] = vreg[u];
    }
    if (tid < Cp) vl[nvec + tid] = vbias;
    // dropout keep-bits of this lane's four elements, bit 4 i + q for cell i: hashed a cell AHEAD (behind the exchange store, while the
    // other parts' pieces are in flight) and kept for the backward pass — same hash, same bits as chain_body's in-place calls
    uint32_t keepm = 0;
    auto keep_bits = [&](const int i) {
        if (g.use_drop && main_w) {
            const int r = rb * 16 + l15;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                keepm |= (drop_keep(h0, i, (uint32_t)((4 * lg + q) * R + r), g.drop_thr) ? 1u : 0u) << (4 * i + q);
        }
    };
    CS_STAMP(36);
    // this part's pieces of the OTHER parity back to "not written" (read by nobody in this launch)
    for (int e = tid; e < XCH_SLOTS * NRO * 64; e += CHAIN_THREADS) {
        const int s = e / (NRO * 64), r = e - s * (NRO * 64);
        stc4<true>(a.xch, xoth_par + ((int64_t)s * 8 + rb0) * 256 + r * 4, __builtin_bit_cast(f32x4, (u32x4){XCH_SENT, XCH_SENT, XCH_SENT, XCH_SENT}));
    }
    // the labels — order entry -> label: two DEPENDENT loads behind the candidate record — belong to the idle wave 7 and to the end of the
    // entry: in front of the slab requests (round 6, first form) the wait for the order entry held every later request of wave 0 back by a
    // memory round trip, and the write-through sentinel stores above sat in front of the slab loads' in-order return (entry 17 k -> 9 k cycles)
    const int lab_b = tid - (CHAIN_THREADS - 64);
    int64_t lab_row = 0;
    if (lab_b >= 0 && lab_b < Bp && lab_b < nvalid) {
        const int32_t* ord = cand_order(a.order, g, cd.gidx);
        lab_row = ord ? (int64_t)ord[cs.pos_t + lab_b] : (int64_t)(cs.base_t + lab_b);
    }
    keep_bits(0);
    if (L > 1) stage_store(tb, nrb);
    CS_STAMP(22);
    __syncthreads();
    CS_STAMP(0);
    int lab_r = 0;

    // piece of another part -> the LDS operand buffer (row-major [b][SX]); one 16-byte piece per thread: wave w takes row block
    // (rb0 + NRO + w) mod 8 — the foreign row blocks come first, so the MAIN waves poll too (their poll's returned load is also
    // what says their own earlier stores were acknowledged), and the last NRO waves would map to own row blocks: nothing to fetch
    auto fetch_others = [&](const int slot, float* dst) {
        const int prb = (rb0 + NRO + wave) & 7;
        if (wave < 8 - NRO) {
            const f32x4 v = xch_wait(a.xch, xcur_par + ((int64_t)slot * 8 + prb) * 256 + lane * 4, a.status, cgidx);
#pragma unroll
            for (int q = 0; q < 4; ++q) dst[(4 * lg + q) * SX + prb * 16 + l15] = v[q];
        }
    };
    // acc += X[16][16 nk] . tiles (LDS, [nk][256]) in chain_body's two MFMA chains — even k-blocks on top of acc, odd k-blocks from zero,
    // summed at the end — but the chains run on TWO waves: main wave w the even one, helper wave w + NRO (another SIMD) the odd one,
    // handed over through LDS at a workgroup barrier every wave of the part passes (mma_half + the barrier + `acc += odd`): same sums
    auto mma_half = [&](f32x4& acc, const float* X, const int sx, const float* tiles, const int nk, const int odd) {
#pragma unroll
        for (int u2 = 0; u2 < 4; ++u2) {
            const int u = 2 * u2 + odd;
            if (u < nk) {
                const f32x4 w4 = *as_lds(reinterpret_cast<const f32x4*>(tiles + u * 256 + lane * 4));
                const f32x4 x4 = *reinterpret_cast<const f32x4*>(X + l15 * sx + u * 16 + 4 * lg);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc = MFMA16(x4[q], w4[q], acc);
            }
        }
    };
    const bool help_w = wave >= NRO && wave < 2 * NRO;   // helper of main wave (wave - NRO)
    static_assert(2 * (8 / NS) <= 4, "main + helper waves below the head waves");
    const int vcol = (main_w ? wave : 0) * 16 + l15;     // column inside the part's vector block

    // ------------------------------------------------------------------ forward chain
    for (int i = 0; i < L; ++i) {
        CS_STAMP(1 + i);
        // next product's tiles: P_{i+1} -> tb[(i+1) & 1]; after the last cell the first backward product (head^T, ncb k-blocks) -> tb[L & 1]
        if (i == L - 1 && lab_b >= 0 && lab_b < Bp && lab_b < nvalid) lab_r = g.loss_mode == 0 ? a.tab.label[lab_row] : (int)lab_row;
        const bool st_fw = i + 1 < L && i >= 1;          // (P_1 was staged at entry)
        const bool st_hd = i + 1 == L;
        if (st_fw) stage_issue(W, offP(i + 1), nrb);
        else if (st_hd) stage_issue(a.wt, oHT, ncb);
        const float* xprev = xo_l + ((i + 1) & 1) * Bp * SX;
        float* xcur = xo_l + (i & 1) * Bp * SX;
        if (main_w) {
            const int nl = (nlbits >> (2 * i)) & 3;
            const int64_t vb = cvec_off + (int64_t)i * g.vec_cell_stride;
            const float* vc = vl + i * 5 * NCOL;
            const int r = rb * 16 + l15;
            const bool colok = r < R;
            const float bias = vc[VEC_B * NCOL + vcol];
            float gam = 1.f, bet = 0.f;
            if (g.bn) { gam = vc[VEC_G * NCOL + vcol]; bet = vc[VEC_BE * NCOL + vcol]; }
            f32x4 acc = *as_lds(reinterpret_cast<const f32x4*>(yf_l + ((i * NRO + wave) << 8) + lane * 4));
            if (i > 0) {
                mma_half(acc, xprev, SX, tb + wave * 8 * 256, nrb, 0);
                lds_barrier();
                acc += *as_lds(reinterpret_cast<const f32x4*>(odd_l + wave * 256 + lane * 4));
            }
            if (i == 1) CS_STAMP(13);
            float av[4];
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int b = 4 * lg + q;
                const float v = act_fwd(acc[q] + bias, nl);
                av[q] = v;
                if (b < nvalid) s += v;
            }
            float zv[4];
            if (g.bn) {
                const float mu = colsum(s) / nf;
                float s2 = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int b = 4 * lg + q;
                    const float dlt = av[q] - mu;
                    if (b < nvalid) s2 += dlt * dlt;
                }
                const float var = colsum(s2) / nf;
                const float rstd = 1.0f / sqrtf(var + g.bn_eps);
                f32x4 xh4;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float xh = (av[q] - mu) * rstd;
                    xh4[q] = xh;
                    zv[q] = xh * gam + bet;
                }
                if (lg == 0) {
                    rstd_l[i * NCOL + vcol] = rstd;
                    if (colok) {
                        float rm = vc[VEC_RM * NCOL + vcol], rv = vc[VEC_RV * NCOL + vcol];
                        const float unb = var * (nf / (nf - 1.0f));
                        rm += g.bn_mom * (mu - rm);
                        rv += g.bn_mom * (unb - rv);
                        W[vb + VEC_RM * Rp + r] = rm;
                        W[vb + VEC_RV * Rp + r] = rv;
                    }
                }
                *as_lds(reinterpret_cast<f32x4*>(sv_l + ((MFAS_MAX_CELLS + i) * NRO + wave) * 256 + lane * 4)) = xh4;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) zv[q] = av[q];
            }
            *as_lds(reinterpret_cast<f32x4*>(sv_l + (i * NRO + wave) * 256 + lane * 4)) = (f32x4){av[0], av[1], av[2], av[3]};
            float* xo_g = sb + g.sb_xo + (int64_t)i * Bp * Rp;
            f32x4 o4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int b = 4 * lg + q;
                float o = zv[q];
                if (g.use_drop) o = ((keepm >> (4 * i + q)) & 1u) ? o * g.drop_scale : 0.0f;
                if (!(colok && b < nvalid)) o = 0.0f;
                o4[q] = o;
            }
            stc4<true>(a.xch, xcur_par + ((int64_t)i * 8 + rb) * 256 + lane * 4, o4);     // slot i: out_i, first thing out
            if (i == 1) CS_STAMP(14);
            if (i + 1 < L) keep_bits(i + 1);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int b = 4 * lg + q;
                xcur[b * SX + r] = o4[q];
                stc1<true>(xo_g + b * Rp + r, o4[q]);
            }
        }
        if (!main_w && i > 0) {         // the odd chain of main wave (wave - NRO), or nothing: every wave meets the product's barrier
            if (help_w) {
                f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
                mma_half(acc2, xprev, SX, tb + (wave - NRO) * 8 * 256, nrb, 1);
                *as_lds(reinterpret_cast<f32x4*>(odd_l + (wave - NRO) * 256 + lane * 4)) = acc2;
            }
            lds_barrier();
        }
        if (st_fw) stage_store(tb, nrb);
        else if (st_hd) stage_store(tb, ncb);
        if (i == 1) CS_STAMP(15);
        fetch_others(i, xcur);
        if (i == 1) CS_STAMP(16);
        lds_barrier();
    }

    CS_STAMP(5);
    // ------------------------------------------------------------------ head (replicated) + loss
    {
        // logits (replicated on every part): class block hcb by the wave pair (hcb, hcb + 4) — even / odd k-blocks, chain_body's two chains;
        // the odd sums cross through the activation buffer the last cell did not write
        const float* xl = xo_l + ((L - 1) & 1) * Bp * SX;
        float* hodd = xo_l + (L & 1) * Bp * SX;          // [ncb][256]
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (head_w) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const f32x4 x4 = *reinterpret_cast<const f32x4*>(xl + l15 * SX + (2 * u + hpar) * 16 + 4 * lg);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc = MFMA16(x4[q], hw[u][q], acc);
            }
            if (hpar) *as_lds(reinterpret_cast<f32x4*>(hodd + hcb * 256 + lane * 4)) = acc;
        }
        lds_barrier();
        if (head_w && !hpar) {
            acc += *as_lds(reinterpret_cast<const f32x4*>(hodd + hcb * 256 + lane * 4));
            const int c = hcb * 16 + l15;
            const float bias = vl[nvec + c];
#pragma unroll
            for (int q = 0; q < 4; ++q) lg_l[(4 * lg + q) * SC + c] = acc[q] + bias;
        }
    }
    if (lab_b >= 0 && lab_b < Bp) lab_l[lab_b] = lab_r;
    lds_barrier();
    CS_STAMP(6);
    if (g.loss_mode == 1) {
        if (tid < 4 * Bp) bce_rows(lg_l, SC, red_l, Bp, lab_l, a.tab.multilabel, a.pos_w, C, Cp, nvalid, tid);
    } else if (Cp <= 64 && !g.multitask) {
        if (tid < 8 * Bp) softmax_rows_lean<MB>(a, lg_l, SC, red_l, lab_l, nvalid, nf, tid);
    } else if (tid < LPR * Bp) {
        softmax_rows<MB>(a, cs, lg_l, SC, red_l, lab_l, nvalid, nf, tid, cand_order(a.order, g, cgidx));
    }
    CS_STAMP(21);
    lds_barrier();
    if (part == 0) {
        if (tid == CHAIN_THREADS - 64 && a.stats) {
            float ls = 0.f, ncor = 0.f;
            for (int b = 0; b < Bp; ++b) { ls += red_l[b]; ncor += red_l[Bp + b]; }
            DevStats& st = a.stats[(int64_t)cgidx * a.E + cs.epoch];
            st.train_loss += (double)ls;
            st.train_corr += (long long)ncor;
            if (!(fabsf(ls) <= 3.0e38f)) atomicMax(&a.status[cgidx], 1);
        }
        float* dlg = sb + g.sb_dlog;
        for (int e = tid; e < Bp * Cp; e += CHAIN_THREADS) {
            const int b = e / Cp, c = e - b * Cp;
            stc1<true>(dlg + e, lg_l[b * SC + c]);
        }
        const int hc = tid - (CHAIN_THREADS - 256);
        if (hc >= 0 && hc < C) {
            float gsum = 0.f;
            for (int b = 0; b < Bp; ++b) gsum += lg_l[b * SC + hc];
            const int64_t o = cvec_off + g.vec_head + hc;
            float w = vl[nvec + hc], m = Mv[o], v = Vv[o];
            adam1(w, m, v, gsum, ac);
            W[o] = w; Mv[o] = m; Vv[o] = v;
        }
    }
    CS_STAMP(7);

    // ------------------------------------------------------------------ backward chain
    for (int i = L - 1; i >= 0; --i) {
        CS_STAMP(8 + (L - 1 - i));
        const int j = L - 1 - i;                          // backward cell j reads tb[(L + j) & 1]
        const bool st_bw = i >= 1;                        // next product: d out_{i-1} = dy_i . OUT_i (transposed tiles of cell i)
        if (st_bw) stage_issue(a.wt, offT(i), nrb);
        const bool from_head = (i == L - 1);
        const float* src = from_head ? lg_l : dy_l + ((i + 1) & 1) * Bp * SX;
        const int sstride = from_head ? SC : SX;
        const int nkk = from_head ? ncb : nrb;
        float* dcur = dy_l + (i & 1) * Bp * SX;
        if (main_w) {
            const int nl = (nlbits >> (2 * i)) & 3;
            const int64_t vb = cvec_off + (int64_t)i * g.vec_cell_stride;
            const float* vc = vl + i * 5 * NCOL;
            const int r = rb * 16 + l15;
            const bool colok = r < R;
            float gr = 0.f;
            if (g.bn) gr = vc[VEC_G * NCOL + vcol] * rstd_l[i * NCOL + vcol];
            const int64_t ob = vb + VEC_B * Rp + r, og = vb + VEC_G * Rp + r, obe = vb + VEC_BE * Rp + r;
            float pw[3] = {0.f, 0.f, 0.f}, pm[3] = {0.f, 0.f, 0.f}, pv[3] = {0.f, 0.f, 0.f};
            if (lg == 0 && colok) {
                pw[0] = vc[VEC_B * NCOL + vcol]; pm[0] = Mv[ob]; pv[0] = Vv[ob];
                if (g.bn) {
                    pw[1] = vc[VEC_G * NCOL + vcol]; pm[1] = Mv[og]; pv[1] = Vv[og];
                    pw[2] = vc[VEC_BE * NCOL + vcol]; pm[2] = Mv[obe]; pv[2] = Vv[obe];
                }
            }
            const f32x4 a4 = *as_lds(reinterpret_cast<const f32x4*>(sv_l + (i * NRO + wave) * 256 + lane * 4));
            f32x4 xh4 = {0.f, 0.f, 0.f, 0.f};
            if (g.bn) xh4 = *as_lds(reinterpret_cast<const f32x4*>(sv_l + ((MFAS_MAX_CELLS + i) * NRO + wave) * 256 + lane * 4));
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            mma_half(acc, src, sstride, tb + wave * 8 * 256, nkk, 0);
            lds_barrier();
            acc += *as_lds(reinterpret_cast<const f32x4*>(odd_l + wave * 256 + lane * 4));
            if (i == 2) CS_STAMP(17);
            float dz[4];
            float sdz = 0.f, sdzx = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int b = 4 * lg + q;
                float d = acc[q];
                if (g.use_drop) d = ((keepm >> (4 * i + q)) & 1u) ? d * g.drop_scale : 0.0f;
                if (!(b < nvalid)) d = 0.f;
                dz[q] = d;
                sdz += d;
                if (g.bn) sdzx += d * xh4[q];
            }
            float dgam = 0.f, dbet = 0.f;
            if (g.bn) {
                dbet = colsum(sdz);
                dgam = colsum(sdzx);
                const float k1 = dbet / nf, k2 = dgam / nf;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int b = 4 * lg + q;
                    const float da = gr * (dz[q] - k1 - xh4[q] * k2);
                    dz[q] = b < nvalid ? da : 0.f;
                }
            }
            float sdy = 0.f;
            float* dy_g = sb + g.sb_dy + (int64_t)i * Bp * Rp;
            f32x4 d4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float dy = act_bwd(a4[q], dz[q], nl);
                if (!colok) dy = 0.f;
                sdy += dy;
                d4[q] = dy;
            }
            if (i >= 1) stc4<true>(a.xch, xcur_par + ((int64_t)(4 + j) * 8 + rb) * 256 + lane * 4, d4);    // slot 4 + j: dy_i
            if (i == 2) CS_STAMP(18);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int b = 4 * lg + q;
                dcur[b * SX + r] = d4[q];
                stc1<true>(dy_g + b * Rp + r, d4[q]);
            }
            const float db = colsum(sdy);
            if (lg == 0 && colok) {
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
        if (!main_w) {
            if (help_w) {
                f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
                mma_half(acc2, src, sstride, tb + (wave - NRO) * 8 * 256, nkk, 1);
                *as_lds(reinterpret_cast<f32x4*>(odd_l + (wave - NRO) * 256 + lane * 4)) = acc2;
            }
            lds_barrier();
        }
        if (st_bw) stage_store(tb, nrb);
        if (i == 2) CS_STAMP(19);
        if (i >= 1) fetch_others(4 + j, dcur);
        if (i == 2) CS_STAMP(20);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // every wave's stores (dy_i, dlogits, Adam) acknowledged: free behind a poll
        lds_barrier();
        // dy_i of this part is out (a wave that polled has seen its earlier stores acknowledged; the last cell drained): arrive
        if (a.cellflag && tid == 0)
            __hip_atomic_fetch_add(a.cellflag + (size_t)cgidx * CELLFLAG_STRIDE + i, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    CS_STAMP(12);
#ifdef MFAS_CHAIN_TIMING
    if (threadIdx.x == 0 && bid == 0 && part == 0 && (cs.gstep == 3 || cs.gstep == 4)) a.status[64 + 24 + 2 * (cs.gstep - 3)] = (int32_t)wall_clock64();
#endif
#undef CS_STAMP
}

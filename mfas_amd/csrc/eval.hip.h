// eval.hip.h — eval-mode forward over table rows (dev accuracy / F1, mfas_population_forward)
// (part of the single translation unit mfas_hip.hip; see the header comment there and DESIGN.md)
#pragma once
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
    int32_t nblk, ncand;  // row tiles per candidate, candidates of this launch (the 1-D grid of the B3 build)
    int32_t wl_safe, _padw;  // WL walk: wait for EVERYTHING (vmcnt(0)) instead of counting the row requests behind the tile copies (MFAS_EVAL_NO_WL;
                             // compile-time in builds whose instruction schedule is not the -O3 one: -DMFAS_EVAL_WL_SAFE, the ASAN variant)
};

#define EVAL_CE 128   // staged feature columns per pass

// MSP: 0, or the number of row blocks (1 / 2) when the m-blocks are split over the waves.  XB (with MSP): bf16 table rows stay 16-bit
// in the LDS tile (half the store instructions, no store conflicts: 8 consecutive lanes write 128 contiguous bytes) and are widened by
// the wave that reads them (each element is read by exactly one wave under the split).
// B3 (bf16 tables, two row blocks per wave: R = 72 .. 128): the feature products run on the bf16 matrix pipe at 16x the f32 rate and
// stay exact — the table rows ARE bf16, and every f32 weight is split into three bf16 terms hi + mid + lo that sum to it exactly
// (8 + 8 + 8 significant bits, by truncation), so each of the three v_mfma_f32_16x16x32_bf16 per k-block pair accumulates exact
// products in f32 like the f32 MFMA does; only the summation order differs (hi terms, then mid, then lo, per 32 columns).
// The rows stay 16-bit in LDS and are the A operand as they lie.  Cell-to-cell and head products (f32 activations) keep the f32 MFMA.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// two f32 weight tiles (k-blocks kA, kB of one row block; lane: row l15, k = 4 lg .. 4 lg + 3 of each) -> three B operands
__device__ __forceinline__ void split3_bf16(const f32x4 wA, const f32x4 wB, u32x4& hi, u32x4& mid, u32x4& lo) {
    uint32_t h[8], m[8], l[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float w = e < 4 ? wA[e] : wB[e - 4];
        h[e] = __float_as_uint(w) & 0xFFFF0000U;
        const float r1 = w - __uint_as_float(h[e]);          // exact: <= 16 significant bits left
        m[e] = __float_as_uint(r1) & 0xFFFF0000U;
        const float r2 = r1 - __uint_as_float(m[e]);         // exact: <= 8 significant bits left = one bf16
        l[e] = __float_as_uint(r2);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {                            // element 2e in the low half, 2e + 1 in the high half
        hi[e] = (h[2 * e] >> 16) | h[2 * e + 1];
        mid[e] = (m[2 * e] >> 16) | m[2 * e + 1];
        lo[e] = (l[2 * e] >> 16) | (l[2 * e + 1] & 0xFFFF0000U);
    }
}

__device__ __forceinline__ f32x4 mfma_bf16_k32(const u32x4 a8, const u32x4 b8, const f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a8), __builtin_bit_cast(bf16x8, b8), c, 0, 0, 0);
}

template <int MBE, int NRBW, int MSP = 0, bool XB = false, bool B3 = false>
__global__ void __launch_bounds__(256, NRBW == 1 ? 4 : ((B3 && MBE <= 4) ? 2 : 1)) k_eval(const EvalArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // B3: a 1-D grid whose consecutive workgroups (dealt round-robin to the 8 XCDs) belong to 8 different candidates, so that all
    // row tiles of one candidate run on ONE XCD and its W (3.9 MB at R = 128, re-read by every tile) stays in that XCD's 4 MB L2
    // instead of six candidates' W thrashing every L2; the candidates past the last full group of 8 keep the plain order
    int cand_i = blockIdx.y, tile_i = blockIdx.x;
    if constexpr (B3) {
        const int per8 = 8 * a.nblk, id = blockIdx.x, grp = id / per8, r = id - grp * per8;
        if (grp < (a.ncand >> 3)) { cand_i = grp * 8 + (r & 7); tile_i = r >> 3; }
        else { const int rem = id - (a.ncand >> 3) * per8; cand_i = (a.ncand >> 3) * 8 + rem / a.nblk; tile_i = rem - (rem / a.nblk) * a.nblk; }
    }
    const int cand = a.cand0 + cand_i;
    const CandDev& cd = a.cands[cand];
    const Geo& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    constexpr int ME = MBE * 16;
    const int Rp = g.Rp, nrb = g.nrb, Cp = g.Cp, ncb = g.ncb, R = g.R, C = g.C, L = cd.L;
    // row strides = 8 mod 16 floats: the A-operand reads (row l15, columns 4 lg .. 4 lg + 3, ds_read_b128) are then conflict-free in
    // each of the instruction's four 16-lane groups (with +4 every group had one 2-way conflict: SQ_LDS_BANK_CONFLICT = half of
    // SQ_LDS_IDX_ACTIVE, profiles/r03_pmc_eval.log)
    const int SX = Rp + 8, SC = Cp + 4, SS = EVAL_CE + 8;
    // LDS kept to <= 80 KiB so that two workgroups share a CU (one stages features while the other runs MFMAs):
    // ONE activation buffer (extra barrier per cell) and the logits alias the feature staging tile.
    float* xs = lds;                     // [ME][SS]   feature staging tile; later the logits [ME][SC]
    constexpr bool X16 = XB || B3;       // table rows stay 16-bit in the LDS tile
    float* xo_l = xs + ME * (X16 ? max((EVAL_CE + 8) / 2, SC) : max(SS, SC)); // [ME][SX]   out_{i-1} -> out_i (16-bit rows: the row tile is half as wide)
    float* lg_l = xs;
    const float* W = a.plane;
    // WL (R <= 16, bf16 rows, m-blocks split over the waves): the four waves multiply DIFFERENT rows with the SAME weight tiles, and as
    // four private copies those tiles were two thirds of what the workgroup pulled through the CU's vector L1 (32 of 48 KB per 128-column
    // chunk; at 4 workgroups per CU the kernel sat at that port's 64 B / clock, which is why neither the MFMA pipe — 45 % busy — nor a
    // fifth / sixth workgroup per CU moved it: profiles/r05_eval_occ.log).  Now ONE copy per workgroup travels global -> LDS by LDS-DMA
    // (wave w copies tiles 2w, 2w + 1 of the NEXT chunk: a tile = 1 KB = the 64 lanes' 16-byte pieces; double-buffered, no registers)
    // and every wave reads its B operands from there (ds_read_b128, conflict-free).  Same operands, same MFMA order: bit-identical.
    constexpr bool WL = XB && MSP == 1 && NRBW == 1 && MBE == 4;
    float* wl_buf = xo_l + ME * SX;      // [2][EVAL_CE / 16][256]
    const uint32_t wl_base = (uint32_t)(uintptr_t)as_lds(wl_buf);
    // wave -> (row block, m-blocks).  R >= 64: wave w owns row blocks w, w+4, ... and all MBE m-blocks of the rows.  With one or
    // two row blocks (R <= 32) that leaves 3 (2) of the 4 waves without MFMA work, so there the m-blocks are split instead:
    // wave w owns row block w % nrb and m-blocks w / nrb, w / nrb + 4 / nrb, ...  Every output element still accumulates the same
    // products in the same order, so the logits are bit-identical under either mapping.
    static_assert(MSP == 0 || (NRBW == 1 && (MSP == 1 || MSP == 2)), "m-block split: one or two row blocks");
    static_assert(!XB || MSP > 0, "16-bit LDS rows: with the m-block split only");
    static_assert(!B3 || (NRBW == 2 && MSP == 0 && !XB), "bf16 x 3 products: the two-row-blocks-per-wave build");
    constexpr int SSH = EVAL_CE + 8;       // row stride of the 16-bit tile in elements: ds_read_b64 of (row l15, columns 4 lg ..) conflict-free
    uint16_t* xh = reinterpret_cast<uint16_t*>(lds);
    constexpr int NPI = MSP ? (MBE * MSP >= 4 ? MBE * MSP / 4 : 1) : MBE;        // accumulators (m-blocks) per wave and row block
    const int rbw = MSP ? wave % MSP : wave;                       // (NRBW == 1) this wave's row block
    const int mb0 = MSP ? wave / MSP : 0;
    constexpr int mbs = MSP ? 4 / MSP : 1;
    const int64_t brow = a.row0 + (int64_t)tile_i * ME;
    const int nvalid = (int)min((int64_t)ME, a.row0 + a.nrows - brow);

    // register-staged table rows (16-bit tables, NRBW <= 2 path): e = tid + 256 u walks [ME rows][nc / 8 vectors of 8 columns]
    // (not for the unsplit one-row-block-per-wave build, R = 48 / 64: its 128-register budget has no room for the row registers)
    const bool t16 = a.tab.dtype != MFAS_DT_F32 && !(MSP == 0 && NRBW == 1);
    const bool bf16 = a.tab.dtype == MFAS_DT_BF16;
    uint4 raw[MBE];
    bool have = false;
    auto rows_load = [&](const void* tp, const int tw, const int c0, const int nc) {
        const int vpr = nc >> 3;
#pragma unroll
        for (int u = 0; u < MBE; ++u) {
            const int e = tid + u * 256;
            int b, c;
            if (vpr == 16) { b = e >> 4; c = (e & 15) << 3; }      // full chunk: no division (uniform branch)
            else { b = e / vpr; c = (e - b * vpr) << 3; }
            raw[u] = make_uint4(0u, 0u, 0u, 0u);
            if (b < nvalid) raw[u] = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(tp) + (brow + b) * tw + c0 + c);
        }
    };
    auto rows_store = [&](const int nc) {
        const int vpr = nc >> 3;
#pragma unroll
        for (int u = 0; u < MBE; ++u) {
            const int e = tid + u * 256;
            int b, c;
            if (vpr == 16) { b = e >> 4; c = (e & 15) << 3; }
            else { b = e / vpr; c = (e - b * vpr) << 3; }
            if (b < ME) {
                if constexpr (X16) {
                    *as_lds(reinterpret_cast<u32x4*>(xh + b * SSH + c)) = (u32x4){raw[u].x, raw[u].y, raw[u].z, raw[u].w};
                    continue;
                }
                const uint32_t w[4] = {raw[u].x, raw[u].y, raw[u].z, raw[u].w};
                float f[8];
                if (bf16) {                      // (one uniform branch per 8 columns, not one per element)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        f[2 * k] = __uint_as_float(w[k] << 16);
                        f[2 * k + 1] = __uint_as_float(w[k] & 0xFFFF0000U);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        f[2 * k] = __half2float(__ushort_as_half((unsigned short)(w[k] & 0xFFFFU)));
                        f[2 * k + 1] = __half2float(__ushort_as_half((unsigned short)(w[k] >> 16)));
                    }
                }
                *as_lds(reinterpret_cast<f32x4*>(xs + b * SS + c)) = (f32x4){f[0], f[1], f[2], f[3]};
                *as_lds(reinterpret_cast<f32x4*>(xs + b * SS + c + 4)) = (f32x4){f[4], f[5], f[6], f[7]};
            }
        }
    };

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
        // sigma(alpha) rounds to exactly 1 for alpha >~ 17: the V modality then contributes v * 0 (aux_models.py:103-111);
        // (accS * sgS / sgV + accV) * sgV would be inf * 0, so that case keeps accS * sgS and skips the V columns
        const bool vdead = g.alphas && !(sgV > 0.0f);
        f32x4 acc[NRBW][NPI];
#pragma unroll
        for (int j = 0; j < NRBW; ++j)
#pragma unroll
            for (int mb = 0; mb < NPI; ++mb) acc[j][mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int sv = 0; sv < 2; ++sv) {
            const int tap = cd.conf[i][sv];
            const void* tp = sv == 0 ? a.tab.s[tap] : a.tab.v[tap];
            const int cols = (vdead && sv == 1) ? 0 : cd.seg_cols[i][sv], cc = cd.seg_cc[i][sv];   // (sigma(alpha) == 1: no V columns)
            const int tw = sv == 0 ? g.sw[tap] : g.vw[tap];
            if (g.alphas && sv == 1) {   // switch modality: fold the S sum with its scale, restart for V
#pragma unroll
                for (int j = 0; j < NRBW; ++j)
#pragma unroll
                    for (int mb = 0; mb < NPI; ++mb) acc[j][mb] = acc[j][mb] * (vdead ? sgS : sgS / sgV);
            }
            have = false;
            // (weight chunk, k-block inside it) of this eval chunk's first k-block, carried along instead of dividing per tile
            const int nkb_c = cc >> 4;
            int wch = 0, wkb = 0;
            // B3: the weight tiles of a chunk are requested a whole chunk ahead — each pair of tile registers is refilled with the
            // next chunk's pair right after its split (no second register set).  k-blocks past the segment's end re-read tile 0 and
            // a wave without a second row block (R < 128) works on its last one: no branch inside a full chunk, so that the
            // compiler's wait counts are exact and no refill is waited for before its chunk comes up
            f32x4 wt[(B3 ? EVAL_CE / 16 : 1)][NRBW];
            // (wave-uniform addressing: segment base + tile offset in scalar registers, the lane's 16 bytes as the vector offset)
            const float* wseg = W + cd.seg_off[i][sv];
            int rbo[NRBW];
#pragma unroll
            for (int j = 0; j < NRBW; ++j) rbo[j] = min(__builtin_amdgcn_readfirstlane(wave) + 4 * j, nrb - 1) * nkb_c * 256;
            // scalar offsets (row block 0) of the EVAL_CE / 16 k-block tiles that start at k-block kb_ of weight chunk ch_; n_ok of them
            // exist, the others point at tile 0 (any chunk width: the walk wraps into the next weight chunk as often as it must)
            int toff[EVAL_CE / 16];
            auto tile_offsets = [&](int ch_, int kb_, const int n_ok) {
#pragma unroll
                for (int kbl = 0; kbl < EVAL_CE / 16; ++kbl) {
                    toff[kbl] = kbl < n_ok ? ch_ * Rp * cc + kb_ * 256 : 0;
                    ++kb_;
                    if (kb_ == nkb_c) { kb_ = 0; ++ch_; }
                }
            };
            auto tiles_ld = [&](const int kbl, const int j) -> f32x4 {
                return *reinterpret_cast<const f32x4*>(wseg + toff[kbl] + rbo[j] + lane * 4);
            };
            if constexpr (B3) {
                if (cols > 0) {          // first chunk: rows and tiles, in the order the loop refills them
                    rows_load(tp, tw, 0, min(EVAL_CE, cols));
                    tile_offsets(0, 0, min(EVAL_CE, cols) >> 4);
#pragma unroll
                    for (int kp = 0; kp < EVAL_CE / 32; ++kp)
#pragma unroll
                        for (int j = 0; j < NRBW; ++j) {
                            wt[2 * kp][j] = tiles_ld(2 * kp, j);
                            wt[2 * kp + 1][j] = tiles_ld(2 * kp + 1, j);
                        }
                }
            }
            if constexpr (WL) {
                // ---- WL segment walk: the table rows of THREE chunks in flight (three register sets, used round robin: a set is refilled
                // with chunk c + 3 as soon as chunk c has gone to LDS), the weight tiles of the next chunk by LDS-DMA into the other half.
                // With one chunk in flight (the general walk below) every chunk waited ~2 us for its rows: the pass was latency-bound at
                // 4.5 us per chunk whatever the occupancy (profiles/r05_eval_occ.log).
                // The row requests are UNCONDITIONAL (rows past the table clamp to its last row and are zeroed when stored): a wave then
                // issues exactly MBE of them per chunk, which is what lets `s_waitcnt vmcnt(MBE)` wait for the tile copies — invisible to
                // the compiler — without waiting for the youngest row set behind them.
                typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                const int wave_u = __builtin_amdgcn_readfirstlane(wave);
                const int nchunks = (cols + EVAL_CE - 1) / EVAL_CE;
                auto wl_rows_load = [&](uint4 (&r)[MBE], const int c) {
                    const int cbase = c * EVAL_CE, ncc = min(EVAL_CE, cols - cbase), vpr = ncc >> 3;
#pragma unroll
                    for (int u = 0; u < MBE; ++u) {
                        const int e = tid + u * 256;
                        int b, cc8;
                        if (vpr == 16) { b = e >> 4; cc8 = (e & 15) << 3; }
                        else { b = e / vpr; cc8 = (e - b * vpr) << 3; }
                        const int bc = min(b, nvalid - 1);
                        r[u] = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(tp) + (brow + bc) * tw + cbase + cc8);
                    }
                };
                auto wl_rows_store = [&](const uint4 (&r)[MBE], const int c) {
                    const int ncc = min(EVAL_CE, cols - c * EVAL_CE), vpr = ncc >> 3;
#pragma unroll
                    for (int u = 0; u < MBE; ++u) {
                        const int e = tid + u * 256;
                        int b, cc8;
                        if (vpr == 16) { b = e >> 4; cc8 = (e & 15) << 3; }
                        else { b = e / vpr; cc8 = (e - b * vpr) << 3; }
                        if (b < ME) {
                            const bool live = b < nvalid;
                            *as_lds(reinterpret_cast<u32x4*>(xh + b * SSH + cc8)) = live ? (u32x4){r[u].x, r[u].y, r[u].z, r[u].w} : (u32x4){0u, 0u, 0u, 0u};
                        }
                    }
                };
                // tiles of chunk c: wave w copies tiles 2w, 2w + 1 into half c & 1
                auto wl_tiles = [&](const int c) {
                    const int ntile = min(EVAL_CE, cols - c * EVAL_CE) >> 4;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int kbl = 2 * wave_u + q;
                        const int kb = c * (EVAL_CE / 16) + kbl;            // k-block inside the segment
                        const int ch = kb / nkb_c, t = kb - ch * nkb_c;
                        if (kbl < ntile)
                            glds16(wseg + (int64_t)ch * Rp * cc + (int64_t)t * 256 + lane * 4,
                                   __builtin_amdgcn_readfirstlane(wl_base + (uint32_t)((((c & 1) * (EVAL_CE / 16) + kbl) * 256) << 2)));
                    }
                };
                auto wl_chunk = [&](uint4 (&r)[MBE], const int c) {
                    __syncthreads();              // every wave has left the MFMAs of chunk c - 1 (and, at c = 0, of the previous segment)
                    if (c == 0) wl_tiles(0);      // (a segment's first chunk: its tiles only now — the halves were still being read)
                    wl_rows_store(r, c);
                    // this wave's copies of chunk c's tiles have landed; only the youngest row set (chunk c + 2, requested behind them) may
                    // still be in flight — and at c = 0 the tiles are the youngest
                    // (the counted form is exact only while exactly MBE compiler-visible vector-memory operations follow a chunk's tile copies —
                    //  no spill reloads of rA / rB / rC, no load sunk past this point: true of the -O3 product build, whose resource usage
                    //  tools/build_remarks.sh prints; any other build, or MFAS_EVAL_NO_WL=1, waits for everything)
#ifdef MFAS_EVAL_WL_SAFE
                    __builtin_amdgcn_s_waitcnt(0x0F70);
#else
                    if (c > 0 && c + 2 < nchunks && !a.wl_safe) __builtin_amdgcn_s_waitcnt(0x0F70 | MBE); else __builtin_amdgcn_s_waitcnt(0x0F70);
#endif
                    __syncthreads();
                    if (c + 1 < nchunks) wl_tiles(c + 1);
                    if (c + 3 < nchunks) wl_rows_load(r, c + 3);
                    const int nkbl = min(EVAL_CE, cols - c * EVAL_CE) >> 4;
                    const float* wl = wl_buf + (c & 1) * (EVAL_CE / 16) * 256;
#pragma unroll
                    for (int kbl = 0; kbl < EVAL_CE / 16; ++kbl)
                        if (kbl < nkbl) {
                            const f32x4 w4 = *as_lds(reinterpret_cast<const f32x4*>(wl + kbl * 256 + lane * 4));
                            const u32x2 rr = *as_lds(reinterpret_cast<const u32x2*>(xh + (mb0 * 16 + l15) * SSH + kbl * 16 + 4 * lg));
                            const f32x4 x4 = (f32x4){__uint_as_float(rr[0] << 16), __uint_as_float(rr[0] & 0xFFFF0000U),
                                                     __uint_as_float(rr[1] << 16), __uint_as_float(rr[1] & 0xFFFF0000U)};
#pragma unroll
                            for (int q = 0; q < 4; ++q) acc[0][0] = MFMA16(x4[q], w4[q], acc[0][0]);
                        }
                };
                uint4 rA[MBE], rB[MBE], rC[MBE];
                if (0 < nchunks) wl_rows_load(rA, 0);
                if (1 < nchunks) wl_rows_load(rB, 1);
                if (2 < nchunks) wl_rows_load(rC, 2);
                for (int c = 0; c < nchunks; c += 3) {
                    wl_chunk(rA, c);
                    if (c + 1 < nchunks) wl_chunk(rB, c + 1);
                    if (c + 2 < nchunks) wl_chunk(rC, c + 2);
                }
            } else
            for (int c0 = 0; c0 < cols; c0 += EVAL_CE, wkb += EVAL_CE / 16) {
                while (wkb >= nkb_c) { wkb -= nkb_c; ++wch; }
                const int nc = min(EVAL_CE, cols - c0);
                if constexpr (B3) {
                    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                    __syncthreads();
                    rows_store(nc);
                    __syncthreads();
                    // the next chunk's rows (the last chunk requests its own again: no branch around the loads)
                    const bool more = c0 + EVAL_CE < cols;
                    const int nkn = more ? min(EVAL_CE, cols - c0 - EVAL_CE) >> 4 : 0;   // k-blocks of the next chunk
                    rows_load(tp, tw, more ? c0 + EVAL_CE : c0, more ? nkn << 4 : nc);
                    const int nkbl = nc >> 4;
                    int nch = wch, nkb0 = wkb + EVAL_CE / 16;            // the next chunk's first k-block
                    while (nkb0 >= nkb_c) { nkb0 -= nkb_c; ++nch; }
                    tile_offsets(nch, nkb0, nkn);
                    auto rows8 = [&](const int k0, const bool two, u32x4 (&a8)[MBE]) {
#pragma unroll
                        for (int mb = 0; mb < MBE; ++mb) {
                            const uint16_t* xr = xh + (mb * 16 + l15) * SSH + k0 * 16 + 4 * lg;
                            const u32x2 r0 = *as_lds(reinterpret_cast<const u32x2*>(xr));
                            u32x2 r1 = *as_lds(reinterpret_cast<const u32x2*>(xr + 16));     // (inside the tile's row stride either way)
                            if (!two) r1 = (u32x2){0u, 0u};                                    // an odd last k-block: zeros on both operands
                            a8[mb] = (u32x4){r0[0], r0[1], r1[0], r1[1]};
                        }
                    };
                    auto prods = [&](const int j, const u32x4 (&a8)[MBE], const u32x4 hi, const u32x4 mid, const u32x4 lo) {
#pragma unroll
                        for (int mb = 0; mb < MBE; ++mb) acc[j][mb] = mfma_bf16_k32(a8[mb], hi, acc[j][mb]);
#pragma unroll
                        for (int mb = 0; mb < MBE; ++mb) acc[j][mb] = mfma_bf16_k32(a8[mb], mid, acc[j][mb]);
#pragma unroll
                        for (int mb = 0; mb < MBE; ++mb) acc[j][mb] = mfma_bf16_k32(a8[mb], lo, acc[j][mb]);
                    };
                    if (nkbl == EVAL_CE / 16) {
#pragma unroll
                        for (int kp = 0; kp < EVAL_CE / 32; ++kp) {
                            const int k0 = 2 * kp;
                            u32x4 a8[MBE];
                            rows8(k0, true, a8);
#pragma unroll
                            for (int j = 0; j < NRBW; ++j) {
                                u32x4 hi, mid, lo;
                                split3_bf16(wt[k0][j], wt[k0 + 1][j], hi, mid, lo);
                                wt[k0][j] = tiles_ld(k0, j);
                                wt[k0 + 1][j] = tiles_ld(k0 + 1, j);
                                prods(j, a8, hi, mid, lo);
                            }
                        }
                    } else {                 // the segment's last, partial chunk: nothing to request
#pragma unroll
                        for (int kp = 0; kp < EVAL_CE / 32; ++kp) {
                            const int k0 = 2 * kp;
                            if (k0 < nkbl) {
                                const bool two = k0 + 1 < nkbl;
                                u32x4 a8[MBE];
                                rows8(k0, two, a8);
#pragma unroll
                                for (int j = 0; j < NRBW; ++j) {
                                    u32x4 hi, mid, lo;
                                    split3_bf16(wt[k0][j], two ? wt[k0 + 1][j] : (f32x4){0.f, 0.f, 0.f, 0.f}, hi, mid, lo);
                                    prods(j, a8, hi, mid, lo);
                                }
                            }
                        }
                    }
                } else if constexpr (NRBW <= 2) {
                    // this chunk's weight tiles are requested BEFORE the feature staging so that their L2 latency
                    // overlaps the staging barriers (8 k-blocks x NRBW row blocks = up to 64 VGPRs)
                    f32x4 wt[EVAL_CE / 16][NRBW];
#pragma unroll
                    for (int kbl = 0; kbl < EVAL_CE / 16; ++kbl)
#pragma unroll
                        for (int j = 0; j < NRBW; ++j) {
                            const int rb = NRBW == 1 ? rbw : wave + 4 * j;
                            if (kbl < (nc >> 4) && rb < nrb) {
                                int ch = wch, t = wkb + kbl;
                                while (t >= nkb_c) { t -= nkb_c; ++ch; }
                                wt[kbl][j] = *reinterpret_cast<const f32x4*>(W + cd.seg_off[i][sv] + (int64_t)ch * Rp * cc + ((int64_t)rb * nkb_c + t) * 256 + lane * 4);
                            }
                        }
                    if (t16) {
                        // 16-bit rows travel global -> registers -> LDS, and the NEXT chunk of this segment is requested before
                        // this chunk's MFMAs (MBE x 16 bytes per thread in flight): the staging latency leaves the critical path
                        if (!have) rows_load(tp, tw, c0, nc);
                        __syncthreads();
                        rows_store(nc);
                        __syncthreads();
                        have = c0 + EVAL_CE < cols;
                        if (have) rows_load(tp, tw, c0 + EVAL_CE, min(EVAL_CE, cols - c0 - EVAL_CE));
                    } else {
                        __syncthreads();
                        stage_table(xs, SS, tp, a.tab.dtype, tw, c0, nc, nullptr, 0, (int)brow, nvalid, ME, tid, 256);
                        __syncthreads();
                    }
                    const int nkbl = nc >> 4;
#pragma unroll
                    for (int kbl = 0; kbl < EVAL_CE / 16; ++kbl)
                        if (kbl < nkbl) {
#pragma unroll
                            for (int j = 0; j < NRBW; ++j) {
                                const int rb = NRBW == 1 ? rbw : wave + 4 * j;
                                if (rb < nrb) {
#pragma unroll
                                    for (int pi = 0; pi < NPI; ++pi) {
                                        const int mb = mb0 + mbs * pi;
                                        if (mb < MBE) {
                                            f32x4 x4;
                                            if constexpr (XB) {
                                                typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                                                const u32x2 r = *as_lds(reinterpret_cast<const u32x2*>(xh + (mb * 16 + l15) * SSH + kbl * 16 + 4 * lg));
                                                x4 = (f32x4){__uint_as_float(r[0] << 16), __uint_as_float(r[0] & 0xFFFF0000U),
                                                             __uint_as_float(r[1] << 16), __uint_as_float(r[1] & 0xFFFF0000U)};
                                            } else {
                                                x4 = *reinterpret_cast<const f32x4*>(xs + (mb * 16 + l15) * SS + kbl * 16 + 4 * lg);
                                            }
#pragma unroll
                                            for (int q = 0; q < 4; ++q) acc[j][pi] = MFMA16(x4[q], wt[kbl][j][q], acc[j][pi]);
                                        }
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
                            const int rb = NRBW == 1 ? rbw : wave + 4 * j;
                            if (rb < nrb) {
                                const f32x4 w4 = *reinterpret_cast<const f32x4*>(W + tile_addr(cd.seg_off[i][sv], Rp, cc, rb, kb) + lane * 4);
#pragma unroll
                                for (int pi = 0; pi < NPI; ++pi) {
                                    const int mb = mb0 + mbs * pi;
                                    if (mb < MBE) {
                                        const f32x4 x4 = *reinterpret_cast<const f32x4*>(xs + (mb * 16 + l15) * SS + kbl * 16 + 4 * lg);
#pragma unroll
                                        for (int q = 0; q < 4; ++q) acc[j][pi] = MFMA16(x4[q], w4[q], acc[j][pi]);
                                    }
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
                for (int mb = 0; mb < NPI; ++mb) acc[j][mb] = acc[j][mb] * (vdead ? 1.0f : sgV);
        }
        if (i > 0) {
#pragma unroll
            for (int j = 0; j < NRBW; ++j) {
                const int rb = NRBW == 1 ? rbw : wave + 4 * j;
                if (rb < nrb) {
                    for (int kb = 0; kb < nrb; ++kb) {
                        const f32x4 w4 = *reinterpret_cast<const f32x4*>(W + tile_addr(cd.seg_off[i][2], Rp, Rp, rb, kb) + lane * 4);
#pragma unroll
                        for (int pi = 0; pi < NPI; ++pi) {
                            const int mb = mb0 + mbs * pi;
                            if (mb < MBE) {
                                const f32x4 x4 = *reinterpret_cast<const f32x4*>(xprev + (mb * 16 + l15) * SX + kb * 16 + 4 * lg);
#pragma unroll
                                for (int q = 0; q < 4; ++q) acc[j][pi] = MFMA16(x4[q], w4[q], acc[j][pi]);
                            }
                        }
                    }
                }
            }
            __syncthreads();   // every wave is done reading out_{i-1} before out_i overwrites it
        }
#pragma unroll
        for (int j = 0; j < NRBW; ++j) {
            const int rb = NRBW == 1 ? rbw : wave + 4 * j;
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
                for (int pi = 0; pi < NPI; ++pi) {
                    const int mb = mb0 + mbs * pi;
                    if (mb < MBE) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int b = mb * 16 + 4 * lg + q;
                            float o = act_fwd(acc[j][pi][q] + bias, nl);
                            if (g.bn) o = ((o - rm) * sc) * gam + sh;
                            if (!(r < R)) o = 0.f;
                            xcur[b * SX + r] = o;
                        }
                    }
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
    if (tid < ME) {   // one lane per row (ME <= 64: wave 0)
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
                // 3-term multitask loss (train_searchable/ntu.py:60-61): + CE(visual logits) + CE(skeleton logits)
                loss = (loss + row_ce(vl, C, lab)) + row_ce(sl, C, lab);
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
        if (lane == 0) {
            if (a.stats) {
                DevStats& st = a.stats[(int64_t)cand * a.E + a.epoch];
                atomicAdd(reinterpret_cast<unsigned long long*>(&st.dev_corr), (unsigned long long)corr);
                atomicAdd(&st.dev_loss, (double)loss);
            }
            if (a.corr_out) atomicAdd(reinterpret_cast<unsigned long long*>(a.corr_out), (unsigned long long)corr);
        }
    }
}

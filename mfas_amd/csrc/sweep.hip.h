// sweep.hip.h — the HBM-bound sweep: dW + Adam + next-step forward per weight tile (per-segment and tap-major decompositions)
// (part of the single translation unit mfas_hip.hip; see the header comment there and DESIGN.md)
#pragma once
// ------------------------------------------------------------------------------------------------
// tile_run — the fused per-tile work shared by both sweep decompositions, for ONE row block `rb` over the k-blocks
// kb0, kb0+kbs, ... < nkb of a chunk: request SWEEP_U tiles of W/m/v, then per tile
//   dW^T = x_t^T dy (4*MB f32 MFMAs; D image == the tile image)  ->  Adam(+L2) on 4 elements/lane in registers  ->
//   store W, m, v (+ transposed copy T for OUT/HEAD)  ->  y_{t+1} += x_{t+1} W_new^T (4*MB f32 MFMAs) into yacc.
// ------------------------------------------------------------------------------------------------
template <int MB, bool NT, int SWEEP_U, bool COH = false>
__device__ __forceinline__ void tile_run(float* Wp, float* Mp, float* Vp, const int rb, const int nkb, const int kb0,
                                         const int kbs, const float* xt, const int ST, const float* xn, const int SN,
                                         const float (&dyf)[MB * 4], const float gsc, const AdamC& ac, const bool upd,
                                         const bool fwd, f32x4 (&yacc)[MB], float* T, const int tstride_rb,
                                         const int lane, const int nj) {
    // nj = ceil(B / 4): batch blocks that hold data (DW_BATCH_LOOP, common.hip.h)
    const int l15 = lane & 15, lg = lane >> 4;
    const float a_ss = ac.ss, a_bc2s = ac.bc2s, a_w1 = ac.w1, a_b2 = ac.b2, a_w2 = ac.w2, a_eps = ac.eps, a_wd = ac.wd;
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
                    // (COH = the same-group fused launch: its 128-register budget has no room for two loop bodies, so it keeps the
                    // padded blocks — they add +0, which can only turn an accumulator of -0 into +0, and Adam's m, v and w come out
                    // the same for g = -0 and g = +0: tests/test_gpu_parity.py::test_same_group_launch_fuzz_bit_identical)
                    DW_BATCH_LOOP(MB, (COH ? MB * 4 : nj), acc = MFMA16(xt[(4 * j + lg) * ST + kb * 16 + l15], dyf[j], acc))
                    { f32x4 w = w4[u], m = m4[u], v = v4[u];
                      adam4(w, m, v, acc * gsc, a_ss, a_bc2s, a_w1, a_b2, a_w2, a_eps, a_wd);
                      w4[u] = w; m4[u] = m; v4[u] = v; }
                    if (NT) {
                        __builtin_nontemporal_store(w4[u], reinterpret_cast<f32x4*>(Wp + off));
                        __builtin_nontemporal_store(m4[u], reinterpret_cast<f32x4*>(Mp + off));
                        __builtin_nontemporal_store(v4[u], reinterpret_cast<f32x4*>(Vp + off));
                    } else {
                        // (COH = the same-group launch, one launch per step: the chain that reads OUT / HEAD runs in the NEXT launch, so plain
                        //  stores do — the write-through forms the removed multi-step persistent kernel needed here cost the OUT units
                        //  their L2 residency)
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
#define STEP_NW 8
#define STEP_THREADS (STEP_NW * 64)

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
    // reduce-in-sweep (small populations, general chain): per (candidate, cell) arrival counters; the LAST feature workgroup of a
    // cell to finish sums the cell's partial slabs (fixed order = the chain's phase 0) into the step buffer's yf area, so the
    // next launch's chain reads 1 (or 2, alphas) reduced slab per cell instead of pulling every partial through its one CU
    uint32_t* red_cnt;
    // same-group fused launch: this unit's dy is published by a chain workgroup of the SAME launch (cellflag != nullptr): wait
    // for cellflag[candidate][cell] >= flag_target before reading it; flag_status[] receives an abort mark on timeout
    const uint32_t* cellflag;
    uint32_t flag_target;
    int32_t* flag_status;
    // gathered rows (two-group streaming schedule with per-candidate sample orders, see gather_body): when set, a feature unit
    // stages x_t / x_{t+1} from its candidate's gathered copy [parity][tap][Bp][width] instead of from the table through the order
    const char* gather;
    int64_t g_cand_stride, g_par_stride;   // bytes between two candidates' buffers / between the two parities of one
    int32_t g_par_t, g_par_n;              // parity that holds the rows of batch t / of batch t + 1
};

// byte offset of tap (kind, tap)'s [Bp][width] block inside one parity of a candidate's gathered rows (S taps first, then V)
__device__ __forceinline__ int64_t gather_tap_off(const Geo& g, const int kind, const int tap, const int eb) {
    int64_t o = 0;
#pragma unroll
    for (int u = 0; u < MFAS_MAX_TAPS; ++u) {
        if (kind == KIND_V || u < tap) o += g.sw[u];
        if (kind == KIND_V && u < tap) o += g.vw[u];
    }
    return o * g.Bp * eb;
}

// ------------------------------------------------------------------------------------------------
// gather_body — one workgroup per candidate of the group whose CHAIN runs in this launch: copies the candidate's OWN table rows of
// its next batch(es) (per-candidate sample order) into its gathered buffer, tap by tap for the taps its configuration uses — the
// table is read in whole-row pieces of 256 B .. 4 KB instead of the 128-byte pieces the feature units would fetch one by one
// (240 K random 128-byte reads per launch cost the HBM far more than their bytes), and every row is read from the table once
// instead of twice (as x_{t+1} of one step and x_t of the next).  The sweep of this group runs in the NEXT launch.
// ------------------------------------------------------------------------------------------------
struct GatherArgs {
    const CandDev* cands;                  // the group's candidates (one workgroup each)
    char* buf;
    int64_t cand_stride, par_stride;       // bytes
    int64_t pos[2];
    int32_t base[2], nvalid[2], par[2];
    int32_t nsets, nblocks;
};

__device__ __forceinline__ void gather_body(const GatherArgs& ga, const CandDev& cd, const Geo& g, const mfas_table& tab,
                                            const int32_t* order, const int tid) {
    const int eb = tab.dtype == MFAS_DT_F32 ? 4 : 2;
    const int32_t* ordp = cand_order(order, g, cd.gidx);
    uint32_t used = 0;            // bit kind * 4 + tap
    for (int i = 0; i < cd.L; ++i) used |= (1u << cd.conf[i][0]) | (1u << (MFAS_MAX_TAPS + cd.conf[i][1]));
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        if (s >= ga.nsets) break;
        char* dst0 = ga.buf + (int64_t)cd.gidx * ga.cand_stride + (int64_t)ga.par[s] * ga.par_stride;
        for (int kt = 0; kt < 2 * MFAS_MAX_TAPS; ++kt) {
            if (!((used >> kt) & 1u)) continue;
            const int kind = kt < MFAS_MAX_TAPS ? KIND_S : KIND_V, tap = kt & (MFAS_MAX_TAPS - 1);
            const int w = kind == KIND_S ? g.sw[tap] : g.vw[tap];
            const char* src0 = reinterpret_cast<const char*>(kind == KIND_S ? tab.s[tap] : tab.v[tap]);
            char* dst = dst0 + gather_tap_off(g, kind, tap, eb);
            const int vpr = (w * eb) >> 4, nvec = ga.nvalid[s] * vpr;      // 16-byte vectors per row (widths are multiples of 16 elements)
            for (int e0 = tid; e0 < nvec; e0 += 4 * STEP_THREADS) {
                u32x4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = e0 + u * STEP_THREADS;
                    if (e < nvec) {
                        const int b = e / vpr, c = e - b * vpr;
                        const int64_t row = ordp ? (int64_t)ordp[ga.pos[s] + b] : (int64_t)(ga.base[s] + b);
                        v[u] = *reinterpret_cast<const u32x4*>(src0 + (row * w) * eb + ((int64_t)c << 4));
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = e0 + u * STEP_THREADS;
                    if (e < nvec) {
                        const int b = e / vpr, c = e - b * vpr;
                        *reinterpret_cast<u32x4*>(dst + ((int64_t)b * w) * eb + ((int64_t)c << 4)) = v[u];
                    }
                }
            }
        }
    }
}

// what changes from one train step to the next (k_step takes it from the launch arguments, the persistent loop computes it)
struct SweepStep {
    int64_t pos_t, pos_n;
    int32_t base_t, base_n, nvalid_t, nvalid_n;
    int32_t upd, fwd;
    float ss, bc2s;
};
__device__ __forceinline__ SweepStep sweep_step_of(const SweepArgs& a) {
    SweepStep s;
    s.pos_t = a.pos_t; s.pos_n = a.pos_n; s.base_t = a.base_t; s.base_n = a.base_n;
    s.nvalid_t = a.nvalid_t; s.nvalid_n = a.nvalid_n; s.upd = a.do_update; s.fwd = a.do_forward;
    s.ss = a.ac.ss; s.bc2s = a.ac.bc2s;
    return s;
}


#ifdef MFAS_CHAIN_TIMING
// (timing build, same-group launch of step 3 with a 4-part chain: flag_target = 4 * 4) when the LAST sweep unit of the launch ended and when
// the FIRST unit of cell 0 saw its flag, 100 MHz ticks, next to the chain's absolute stamps (status slots 64 + 27 / 28)
#define SW_UNIT_STAMP(k) do { if (COH && a.cellflag && a.flag_target == 16u && threadIdx.x == 0 && d.cell == 0 && d.kind == KIND_S && d.k0 == 0 && d.cand == 0) a.flag_status[64 + 29 + (k)] = (int32_t)wall_clock64(); } while (0)
#define SW_END_STAMP() do { if (COH && a.cellflag && a.flag_target == 16u && threadIdx.x == 0) { atomicMax(reinterpret_cast<uint32_t*>(&a.flag_status[64 + 27]), (uint32_t)wall_clock64()); \
        atomicMax(reinterpret_cast<uint32_t*>(&a.flag_status[128 + d.cell * 4 + d.kind]), (uint32_t)wall_clock64()); } } while (0)
#else
#define SW_END_STAMP() do { } while (0)
#define SW_UNIT_STAMP(k) do { } while (0)
#endif
template <int MB, bool NT, int U, bool COH = false>
__device__ __forceinline__ void sweep_body(const SweepArgs& a, const SweepStep& st, const int bid, float* lds) {
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
    const bool upd = st.upd != 0;
    const bool fwd = (st.fwd != 0) && feat;
    if (!upd && !fwd) return;
    const int64_t sbo = cd.step_off;   // this candidate's step buffers inside a.stepbuf
    const bool red = a.red_cnt != nullptr;

    if (!COH && a.gather && feat) {      // the candidate's rows were gathered a launch ago (gather_body): rows 0 .. B-1 of its own copy
        const char* gb = a.gather + (int64_t)cd.gidx * a.g_cand_stride + gather_tap_off(a.g, d.kind, d.tap, a.tab.dtype == MFAS_DT_F32 ? 4 : 2);
        if (upd) stage_table(xt, ST, gb + (int64_t)a.g_par_t * a.g_par_stride, a.tab.dtype, d.width, d.k0, cc, nullptr, 0, 0, st.nvalid_t, Bp, tid, STEP_THREADS);
        if (fwd) stage_table(xn, SN, gb + (int64_t)a.g_par_n * a.g_par_stride, a.tab.dtype, d.width, d.k0, cc, nullptr, 0, 0, st.nvalid_n, Bp, tid, STEP_THREADS);
    } else {
    if (upd && feat) {
        const void* tp = d.kind == KIND_S ? a.tab.s[d.tap] : a.tab.v[d.tap];
        stage_table(xt, ST, tp, a.tab.dtype, d.width, d.k0, cc, cand_order(a.order, a.g, cd.gidx), st.pos_t, st.base_t, st.nvalid_t, Bp, tid, STEP_THREADS);
    }
    if (fwd) {
        const void* tp = d.kind == KIND_S ? a.tab.s[d.tap] : a.tab.v[d.tap];
        stage_table(xn, SN, tp, a.tab.dtype, d.width, d.k0, cc, cand_order(a.order, a.g, cd.gidx), st.pos_n, st.base_n, st.nvalid_n, Bp, tid, STEP_THREADS);
    }
    }
    if constexpr (COH) {
        if (upd && a.cellflag) {   // the table rows above are in flight while the chain of this launch gets to this cell's dy
            __shared__ int s_go;
            if (tid == 0) {
                // feature units of cell i need dy_i; OUT_i (the prev-out block of cell i) and HEAD also OVERWRITE weights the chain of
                // this launch still reads in its backward pass (W_out_i^T for d out_{i-1}, Wc^T for d out_{L-1}): they wait until
                // the chain is past that product, i.e. for dy_{i-1} / dy_{L-1}
                const int slot = d.kind == KIND_HEAD ? cd.L - 1 : (d.kind == KIND_OUT ? d.cell - 1 : d.cell);
                const uint32_t* f = a.cellflag + (size_t)cd.gidx * CELLFLAG_STRIDE + slot;
                uint32_t spins = 0;
                int go = 1;
                while (ld_u32_relaxed(f) < a.flag_target) {
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > CELLFLAG_SPIN_LIMIT) { go = 0; atomicMax(&a.flag_status[cd.gidx], 2); break; }
                }
                s_go = go;
#ifdef MFAS_CHAIN_TIMING
                if (a.flag_target == 16u && slot == 0 && d.kind <= KIND_V) atomicMin(reinterpret_cast<uint32_t*>(&a.flag_status[64 + 28]), (uint32_t)wall_clock64());
#endif
            }
            __syncthreads();
            if (!s_go) return;
            SW_UNIT_STAMP(0);
        }
    }
    if (upd) {
        if (!feat) {
            const int xcell = d.kind == KIND_OUT ? d.cell - 1 : cd.L - 1;
            stage_f32<COH>(xt, ST, a.stepbuf, sbo + a.g.sb_xo + (int64_t)xcell * Bp * a.g.Rp + d.k0, a.g.Rp, cc, Bp, tid, STEP_THREADS);
        }
        // dy of this unit's rows: columns [16*rb0, 16*rb0 + rows_p) of the segment's dy (row stride = the segment's padded rows)
        const int64_t dsrc = d.kind == KIND_HEAD ? sbo + a.g.sb_dlog : sbo + a.g.sb_dy + (int64_t)d.cell * Bp * a.g.Rp;
        stage_f32<COH>(dyl, SD, a.stepbuf, dsrc + d.rb0 * 16, d.seg_nrb * 16, rows_p, Bp, tid, STEP_THREADS);
    }
    __syncthreads();
    SW_UNIT_STAMP(1);

    float* Wp = a.plane + d.w_off;
    float* Mp = Wp + a.plane_stride;
    float* Vp = Mp + a.plane_stride;
    const AdamC ac = adam_consts(a.ac, upd ? st.ss : 0.f, upd ? st.bc2s : 1.f);
    // alpha scaling of the gradient of S / V columns (aux_models.py:103-111): sigma(alpha_t) as used by this
    // step's forward, published by k_chain (alpha itself has already been stepped); 1.0 when alphas are off
    float gsc = 1.0f;
    if (a.g.alphas && feat && upd) gsc = ldc1<COH>(a.stepbuf + sbo + a.g.sb_gsc + d.cell * 2 + d.kind);

    // Work split: with >= 8 row blocks every wave owns whole row blocks (streams contiguous tiles, no
    // reduction); with fewer (R < 128, or a row-split unit) the waves split the k blocks and reduce through LDS.
    const bool split_k = nrb < STEP_NW;
    const int rb0 = split_k ? 0 : wave, rbs = split_k ? 1 : STEP_NW;
    const int kb0 = split_k ? wave : 0, kbs = split_k ? STEP_NW : 1;
    // partial slot of this chunk: [seg_nrb][MB][256] in MFMA D layout; this unit owns row blocks rb0 .. rb0 + nrb
    const int64_t part = sbo + a.g.sb_part + (((int64_t)(cd.part_cell_off[d.cell] + d.part_idx) * d.seg_nrb * MB) << 8) +
                         (((int64_t)d.rb0 * MB) << 8);

    for (int rb = rb0; rb < nrb; rb += rbs) {
        float dyf[MB * 4];
#pragma unroll
        for (int j = 0; j < MB * 4; ++j) dyf[j] = upd ? dyl[(4 * j + lg) * SD + rb * 16 + l15] : 0.f;
        f32x4 yacc[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) yacc[mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        tile_run<MB, NT, U, COH>(Wp, Mp, Vp, rb, nkb, kb0, kbs, xt, ST, xn, SN, dyf, gsc, ac, upd, fwd, yacc,
                                 d.wt_off >= 0 ? a.wt + d.wt_off + ((int64_t)(d.k0 >> 4) * d.seg_nrb + d.rb0) * 256 : nullptr, d.seg_nrb, lane, (a.g.B + 3) >> 2);
        if (fwd) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                if (split_k)
                    *reinterpret_cast<f32x4*>(wred + (((wave * nrb + rb) * MB + mb) << 8) + lane * 4) = yacc[mb];
                else if (red) stc4<true>(a.stepbuf, part + ((rb * MB + mb) << 8) + lane * 4, yacc[mb]);
                else   // partial slot in MFMA D layout [chunk][rb][mb][lane][4]
                    stc4<COH>(a.stepbuf, part + ((rb * MB + mb) << 8) + lane * 4, yacc[mb]);
            }
        }
    }
    SW_UNIT_STAMP(2);
    if (!fwd) { SW_END_STAMP(); return; }
    if (split_k) {
        __syncthreads();
        // deterministic cross-wave reduction (fixed order 0..7)
        for (int e = tid; e < nrb * MB * 64; e += STEP_THREADS) {
            const int slot = e >> 6, ln = e & 63;
            f32x4 s = *reinterpret_cast<const f32x4*>(wred + (slot << 8) + ln * 4);
#pragma unroll
            for (int w = 1; w < STEP_NW; ++w)
                s += *reinterpret_cast<const f32x4*>(wred + ((w * nrb * MB + slot) << 8) + ln * 4);
            if (red) stc4<true>(a.stepbuf, part + (slot << 8) + ln * 4, s);
            else stc4<COH>(a.stepbuf, part + (slot << 8) + ln * 4, s);
        }
    }
    if (!red) { SW_END_STAMP(); return; }
    // ---- reduce-in-sweep: arrive on the cell's counter; the last arriver sums the cell's slabs (write-through stores + every
    // wave's drain + barrier + relaxed agent-scope counter on the producer side, sc1 loads on the reader's: G16 R1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    SW_UNIT_STAMP(3);
    int* flagw = reinterpret_cast<int*>(lds);      // (the staging tiles are dead by now)
    const int ns_c = cd.nch_s[d.cell], nch_c = ns_c + cd.nch_v[d.cell];
    if (tid == 0) {
        uint32_t* cnt = a.red_cnt + d.cand * MFAS_MAX_CELLS + d.cell;
        const uint32_t old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = old == (uint32_t)(nch_c - 1);
        if (last) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        flagw[0] = last;
    }
    __syncthreads();
    SW_UNIT_STAMP(4);
    if (!flagw[0]) { SW_END_STAMP(); return; }
    {
        const int snrb = d.seg_nrb;
        const int per_cell = snrb * MB * 64;            // float4 items of the cell's [nrb][MB][256] slab
        const int64_t part0 = sbo + a.g.sb_part + (((int64_t)cd.part_cell_off[d.cell] * snrb * MB) << 8);
        const int64_t yf0 = sbo + a.g.sb_yf + (int64_t)d.cell * per_cell * 4;
        const int64_t vplane = (int64_t)MFAS_MAX_CELLS * snrb * MB * 256;
        for (int it = tid; it < per_cell; it += STEP_THREADS) {
            f32x4 accS = {0.f, 0.f, 0.f, 0.f}, accV = {0.f, 0.f, 0.f, 0.f};
            constexpr int PB = 8;
            for (int ch0 = 0; ch0 < nch_c; ch0 += PB) {
                f32x4 p8[PB];
#pragma unroll
                for (int u = 0; u < PB; ++u)
                    if (ch0 + u < nch_c) p8[u] = ldc4<true>(a.stepbuf, part0 + (((int64_t)(ch0 + u) * snrb * MB) << 8) + it * 4);
#pragma unroll
                for (int u = 0; u < PB; ++u)
                    if (ch0 + u < nch_c) {
                        if (ch0 + u < ns_c) accS += p8[u]; else accV += p8[u];
                    }
            }
            if (a.g.alphas) {
                *reinterpret_cast<f32x4*>(a.stepbuf + yf0 + it * 4) = accS;
                *reinterpret_cast<f32x4*>(a.stepbuf + yf0 + vplane + it * 4) = accV;
            } else {
                *reinterpret_cast<f32x4*>(a.stepbuf + yf0 + it * 4) = accS + accV;
            }
        }
    }
    SW_END_STAMP();
}

// ------------------------------------------------------------------------------------------------
// sweep_multi_body — a FEATURE unit that spans `nsub` consecutive 64-column chunks of its segment (round 4; SegDesc::nsub > 1, only
// with exactly 8 row blocks: wave = row block, R in 113..128).  The memory walk is that of the one-chunk units — chunk after chunk,
// each a contiguous [rb][kb][256] block per plane — but dy is staged once, the wave's forward accumulator lives in registers
// across the chunks, and the unit writes ONE partial slab: the slab traffic (8 KB written per chunk and read back by the chain)
// and the dy staging shrink by the group factor.  Arithmetic per tile = tile_run's; only the k-summation of the forward partial
// is regrouped (as with any other chunk size).  OPT-IN (MFAS_SUBCHUNKS=n): measured slower than one-chunk units on MI355X (2 / 4 / 8
// chunks per unit: 314 / 320 / 327 us per launch against 299, profiles/r04_subchunks_preload_prio.log) although 4-7 % fewer bytes move —
// the workgroup-wide barrier between chunks makes every wave wait for the slowest, where one-chunk workgroups hand their wave slots
// to the next workgroup one wave at a time.
// ------------------------------------------------------------------------------------------------
template <int MB, bool NT, int U>
__device__ __forceinline__ void sweep_multi_body(const SweepArgs& a, const SweepStep& st, const int bid, float* lds) {
    const SegDesc& d = a.desc[bid];
    const CandDev& cd = a.cands[d.cand];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    constexpr int Bp = MB * 16;
    const int cc = d.cc, rows_p = d.rows_p, nkb = cc >> 4;
    const int ST = cc + 16, SN = cc + 4, SD = rows_p + 16;
    float* xt = lds;
    float* xn = xt + Bp * ST;
    float* dyl = xn + Bp * SN;
    const bool upd = st.upd != 0;
    const bool fwd = st.fwd != 0;
    if (!upd && !fwd) return;
    const int64_t sbo = cd.step_off;
    const int nsub = d.nsub;
    const void* tp = d.kind == KIND_S ? a.tab.s[d.tap] : a.tab.v[d.tap];
    const int32_t* ord = cand_order(a.order, a.g, cd.gidx);
    if (upd) stage_f32<false>(dyl, SD, a.stepbuf, sbo + a.g.sb_dy + (int64_t)d.cell * Bp * a.g.Rp, d.seg_nrb * 16, rows_p, Bp, tid, STEP_THREADS);
    float gsc = 1.0f;
    if (a.g.alphas && upd) gsc = a.stepbuf[sbo + a.g.sb_gsc + d.cell * 2 + d.kind];
    const AdamC ac = adam_consts(a.ac, upd ? st.ss : 0.f, upd ? st.bc2s : 1.f);
    f32x4 yacc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) yacc[mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int rb = wave;            // 8 row blocks, one per wave
    for (int sub = 0; sub < nsub; ++sub) {
        const int k0 = d.k0 + sub * cc;
        float* Wp = a.plane + d.w_off + (int64_t)sub * rows_p * cc;
        if (sub > 0) __syncthreads();                    // every wave is done with the previous chunk's rows
        if (upd) stage_table(xt, ST, tp, a.tab.dtype, d.width, k0, cc, ord, st.pos_t, st.base_t, st.nvalid_t, Bp, tid, STEP_THREADS);
        if (fwd) stage_table(xn, SN, tp, a.tab.dtype, d.width, k0, cc, ord, st.pos_n, st.base_n, st.nvalid_n, Bp, tid, STEP_THREADS);
        __syncthreads();
        float dyf[MB * 4];
#pragma unroll
        for (int j = 0; j < MB * 4; ++j) dyf[j] = upd ? dyl[(4 * j + lg) * SD + rb * 16 + l15] : 0.f;
        tile_run<MB, NT, U, false>(Wp, Wp + a.plane_stride, Wp + 2 * a.plane_stride, rb, nkb, 0, 1, xt, ST, xn, SN, dyf, gsc, ac, upd, fwd, yacc,
                                   nullptr, d.seg_nrb, lane, (a.g.B + 3) >> 2);
    }
    if (fwd) {
        const int64_t part = sbo + a.g.sb_part + (((int64_t)(cd.part_cell_off[d.cell] + d.part_idx) * d.seg_nrb * MB) << 8);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
            *reinterpret_cast<f32x4*>(a.stepbuf + part + ((rb * MB + mb) << 8) + lane * 4) = yacc[mb];
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
    const AdamC ac = adam_consts(a.ac, a.ac.ss, a.ac.bc2s);
    f32x4 yacc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) yacc[mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    tile_run<MB, NT, U>(Wp, Mp, Vp, rb, nkb, 0, 1, xt, ST, xn, SN, dyf, gsc, ac, upd, fwd, yacc, nullptr, nrb, lane, (a.g.B + 3) >> 2);
    if (fwd) {
        float* part = sb + a.g.sb_part + (((int64_t)(cd.part_cell_off[cell] + d.part_idx[item]) * nrb * MB) << 8);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
            *reinterpret_cast<f32x4*>(part + ((rb * MB + mb) << 8) + lane * 4) = yacc[mb];
    }
}

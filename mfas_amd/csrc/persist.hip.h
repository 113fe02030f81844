// persist.hip.h — the persistent step loop for SMALL populations (the regime the search really runs in: 6-16 candidates per
// GPU, models/searchable.py:90,120 issue calls of <= 50 configurations over up to 8 GPUs).
// (part of the single translation unit mfas_hip.hip; see the header comment there and DESIGN.md)
#pragma once
// ------------------------------------------------------------------------------------------------
// With a handful of candidates a train step is a latency chain, not a bandwidth problem: chain(c,t) -> units(c,t) ->
// chain(c,t+1), and the launch-per-phase schedule (k_chain / k_step) additionally serialises ALL candidates' sweeps against
// ALL chains at every kernel boundary.  k_president is ONE launch per epoch: every workgroup is resident (one per CU) and
// loops over the epoch's train steps; dependencies are per CANDIDATE:
//   * chain workgroup c (blocks [0, K)) waits until the arrival counter cnt[c] shows that every feature unit of candidate c
//     has finished step t-1 (its W/m/v update and the forward partial sums of batch t), runs the chain of step t and
//     publishes flag[c] = t + 1;
//   * unit workgroups (the remaining blocks; fixed ownership for the whole launch) wait for flag[cand(i)] >= t + 1, run the
//     unit's dW + Adam + next-step forward and arrive on cnt[cand(i)].
// So while candidate A sits in its serial chain, the CUs serve the units of the candidates whose chains have finished.
// Data exchanged inside the launch (dy from the chain; partial sums from the units) is stored write-through and loaded with
// sc1 (COH helpers, common.hip.h): no release / acquire fences, whose L2 write-back would drag the whole XCD's dirty lines
// along.  Every storing wave drains (`s_waitcnt vmcnt(0)`) before the workgroup barrier that precedes the relaxed agent-scope
// flag store / counter add; pollers are ONE lane per workgroup, relaxed loads + s_sleep, bounded (a timeout sets the abort
// word, every workgroup leaves, the host reports an error).
// Reduction orders, tile decomposition and arithmetic are those of the launch-per-phase schedule: results are bit-identical.
// (Rounds 2-3 also carried a STREAMING form — any R, units streaming W/m/v from memory like k_step, general or lean chain
//  exchanging everything through memory — that was measured 0.8-1.1x of launch-per-phase (profiles/r02_popsweep_general_chain.log,
//  DESIGN.md 5a/5b), ran only when forced, spilled, and hung on the device after an unrelated edit: removed in round 3.)
// ------------------------------------------------------------------------------------------------
struct PersistArgs {
    SweepArgs sa;              // desc = the population's sweep units (per-segment), cands = all candidates
    ChainArgs ca;              // cands = all candidates
    int32_t nchain, nitems;    // K, number of feature units
    int32_t res_wide, res_nu;  // resident units wider than 512 columns exist (16-bit staging, 8 tiles per wave); units per workgroup (1 / 2)
    int32_t nres_wg, res_buf_words;   // resident workgroups (blocks K .. K + nres_wg: unit u of workgroup w = w + u * nres_wg);
                                      // LDS words of one staged batch
    int32_t nres, res_chain;   // units [0, nres) are RESIDENT feature units: one workgroup each (blocks K .. K + nres);
                               // res_chain: the (lean) chain owns OUT / HEAD and keeps them + its vector block on chip
    int32_t lose_step, _padl;  // lose_step: -1 in the product library; the -DMFAS_TEST_HOOKS variant reads MFAS_PERSIST_TEST_LOSE_STEP: candidate 0's chain
                               // never publishes this step — the bounded waits must then end the launch with an error instead of hanging
    int32_t T, epoch;          // train steps of this launch, epoch index (statistics slot)
    int64_t N, pos0;           // N_train, epoch * N_train (position in the sample-order table)
    int32_t B, gstep0;         // batch size, epoch * batches-per-epoch (Adam / dropout step counter base)
    const float* scal;         // device [steps][2]: {lr_t/(1-beta1^t), sqrt(1-beta2^t)}
    uint32_t* sync;            // per candidate c one 256-byte record: sync[64c] = flag (steps whose dy is published), sync[64c + 32] =
                               // unit arrivals (own 128-byte line: hundreds of pollers and the arrival atomics of different candidates
                               // must not share a cache line / L2 channel); sync[64K] = abort word
    const int32_t* need;       // [K] sweep units of candidate c (= arrivals per step)
    const int32_t* role;       // [grid] XCD-aware placement (round 5): block b runs role[b] — < K: that candidate's chain, else unit workgroup role[b] - K (nullptr: b)
    unsigned long long* trace; // optional: 100 MHz timestamps of candidate 0's chain and of sweep unit 0 (steps 8..15)
};

#define PERSIST_SYNC_STRIDE 64          // uint32 words per candidate in the sync area
#define PERSIST_FLAG(sync, c) ((sync) + (size_t)(c) * PERSIST_SYNC_STRIDE)
#define PERSIST_CNT(sync, c) ((sync) + (size_t)(c) * PERSIST_SYNC_STRIDE + 32)
#define PERSIST_LDS_WORDS 32            // LDS words the loop itself uses (behind the bodies' LDS)
// step-phase timestamps (profiles/r02_persist_trace_k6_r16.log; MFAS_PERSIST_TRACE=1 in the environment allocates the buffer)
#define PTRACE(slot) do { if (a.trace && tr_on) a.trace[tr_base + (slot)] = wall_clock64(); } while (0)
#define PTRACE_UNIT(base) do { if (a.trace && un.cand == 0 && t == 12 && tid == 0 && un.index < 64) a.trace[(base) + un.index] = wall_clock64(); } while (0)
#define PERSIST_SPIN_LIMIT (1u << 22)   // a few seconds of s_sleep polls: only a lost workgroup or a bug gets here
#define PERSIST_RING_LIMIT (1u << 25)   // the same few seconds for the pipelined polls (a read every ~0.1 us instead of one per round trip)
#define PERSIST_ROLL_LIMIT 6000u        // roll call at launch start: ~5 ms of polls for every workgroup of the grid to be resident
#define PERSIST_ROLL(sync, K) ((sync) + (size_t)(K) * PERSIST_SYNC_STRIDE + 32)   // workgroups that have started (own cache line)
#define PERSIST_MAX_RELAUNCHES 40       // host: relaunches of an epoch whose roll call failed before the resident schedule is given up
#define PERSIST_ABORT_NOT_RESIDENT 2u   // abort code of a failed roll call: nothing has been modified, the host may simply relaunch

// Workgroup-wide wait until *p >= target: lane 0 polls (relaxed, sc1), everyone else parks at the barrier.
// Returns false when the launch is being aborted.  `ldsw` = one LDS word outside the bodies' LDS footprint.
__device__ __forceinline__ bool wg_wait_ge(const uint32_t* p, uint32_t target, uint32_t* abortw, int* ldsw) {
    if (threadIdx.x == 0) {
        int ok = 1;
        uint32_t spins = 0;
        // pipelined poll: four reads in flight, a short sleep apart — a read's round trip to the coherence point (~0.4 us) is then
        // not the polling interval, only its latency
        uint32_t v0 = ld_u32_relaxed(p);
        __builtin_amdgcn_s_sleep(2);
        uint32_t v1 = ld_u32_relaxed(p);
        __builtin_amdgcn_s_sleep(2);
        uint32_t v2 = ld_u32_relaxed(p);
        for (;;) {      // the OLDEST read is tested (the two younger ones stay in flight), then re-issued behind them: a ring of three
                        // registers, unrolled (rotating the registers instead makes the compiler wait for the youngest read)
            if (v0 >= target) break;
            __builtin_amdgcn_s_sleep(2);
            v0 = ld_u32_relaxed(p);
            if (v1 >= target) break;
            __builtin_amdgcn_s_sleep(2);
            v1 = ld_u32_relaxed(p);
            if (v2 >= target) break;
            __builtin_amdgcn_s_sleep(2);
            v2 = ld_u32_relaxed(p);
            if (((spins += 3) & 0x3FFu) < 3 && (spins > PERSIST_RING_LIMIT || ld_u32_relaxed(abortw) != 0)) {
                __hip_atomic_store(abortw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
        }
        *ldsw = ok;
    }
    __syncthreads();
    const int ok = *ldsw;
    __syncthreads();
    return ok != 0;
}

// ------------------------------------------------------------------------------------------------
// sweep_resident — feature units (a column chunk of one S / V segment, R <= 16: one row block) owned by ONE workgroup for
// the whole launch, with their W / m / v tiles held IN REGISTERS across the epoch's steps: the 128 MB register file of the
// chip is the parameter store of a small population (the search trains 6-16 candidates of ~125 K parameters per GPU:
// 1.5 MB of state each).  Per step a unit reads dy (2 KB, from the chain) and the batch's table rows and writes its 2 KB
// forward partial — no W/m/v traffic at all; the state is loaded at launch start and stored back at its end (the dev
// evaluation and the next epoch's launch read it from memory).  The rows of batch t+2 are staged into LDS right after a
// unit finishes step t (the sample order is known), so no table access sits on the critical path:
// wait -> dy -> dW (MFMA) -> Adam -> forward (MFMA) -> partial.
// NU = 2: a workgroup owns TWO units (of different candidates) and serves whichever candidate has published its step —
// twice the candidates fit one GPU at the price of an occasional wait behind the co-tenant.
// Tile -> wave mapping, MFMA order, Adam arithmetic and the cross-wave reduction are those of sweep_body's k-split path:
// bit-identical to the launch-per-phase schedule run on the same units.
// X16: bf16 / f16 tables are staged RAW (16-bit) and converted when read as MFMA operands (exact: the same f32 values as the
// f32 staging) — half the LDS, so a unit may span 1024 columns.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float cvt16(uint32_t u, int dtype) {
    return dtype == MFAS_DT_BF16 ? __uint_as_float(u << 16) : __half2float(__ushort_as_half((unsigned short)u));
}
// table rows of one batch -> LDS, raw 16-bit elements, row stride S16 halves (16 B aligned rows)
__device__ __forceinline__ void stage_table16(uint16_t* dst, int S16, const void* tab, int width, int col0, int ncols,
                                              const int32_t* ord, int64_t pos, int base, int nvalid, int nrows, int tid, int nthreads) {
    const int vpr = ncols >> 3;
    for (int e = tid; e < nrows * vpr; e += nthreads) {
        const int b = e / vpr, c = (e - b * vpr) << 3;
        u32x4 raw = {0u, 0u, 0u, 0u};
        if (b < nvalid) {
            const int64_t row = ord ? (int64_t)ord[pos + b] : (int64_t)(base + b);
            raw = *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint16_t*>(tab) + row * width + col0 + c);
        }
        *as_lds(reinterpret_cast<u32x4*>(dst + b * S16 + c)) = raw;
    }
}

// LDS-DMA staging of 16-bit table rows (round 4; OPT-IN: the -DMFAS_RES_DMA=1 build variant, __graft_entry__.build_variant("dma", ...)).
// Measured on MI355X (profiles/r04_resident_lds_dma_ab.log): +1-2 % at 24-28 equal-sized candidates, but -4...-10 % (and noisy) on
// mixed-depth populations of 28 — what the search issues — and nothing below 16 candidates: the default stays the cooperative
// register staging.  Each wave stages exactly
// the k-blocks IT multiplies with — one `global_load_lds_dwordx4` per k-block and batch: 32 rows x 32 bytes = the 64 lanes' 16-byte
// pieces, landing contiguously in a k-block-major LDS image [kb][row][16 halves] — so the copy needs no VGPRs, no LDS-store pass and
// NO workgroup barrier (a wave only ever reads what it wrote itself), and it is asynchronous: issued right after a unit's step, it
// lands while the workgroup already serves its other unit or waits for the chain.  Rows beyond the batch read a zeroed line.
// (M0 = the wave-uniform LDS destination; saved and restored inside the statement: cdna_hip_programming.md, LDS-DMA recipe.)
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
#ifndef MFAS_RES_DMA
#define MFAS_RES_DMA 0
#endif

// forward partial of a resident unit: operands SWAPPED -> the slab is the partial's TRANSPOSE, chain_lean's register image (the
// f32 MFMA is symmetric under the swap bit for bit: tools/mfma_swap_check.hip)
#if MFAS_RES_TRANSPOSED_SLABS
#define RES_FWD_MFMA(x, w, acc) MFMA16((w), (x), (acc))
#else
#define RES_FWD_MFMA(x, w, acc) MFMA16((x), (w), (acc))
#endif
struct ResUnit {              // wave-uniform constants of one resident unit
    int32_t valid, index, cand, cell, kind, cc, nkb, S, width, k0;
    const void* tp;
    int64_t part, dyo, gsco;  // step-buffer indices: partial slot, dy_i, alpha scale
    float *Wp, *Mp, *Vp;
    int32_t gidx;             // its candidate's population index (selects the sample order it walks)
    uint32_t *flag, *cnt;
    int32_t xb0, xbw;         // the two staged batches in LDS: word offsets xb0 and xb0 + xbw (a pointer picked at run time would be a
                              // flat pointer; a two-element ARRAY indexed at run time parks the whole record in scratch memory)
};
__device__ __forceinline__ int res_xbo(const ResUnit& un, int which) { return un.xb0 + (which ? un.xbw : 0); }

template <int MB, int NTR, bool X16, int NU>
__device__ __forceinline__ void sweep_resident(const PersistArgs& a, const int wg, const int nwg, float* lds, int* ldsw) {
    const SweepArgs& sa = a.sa;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    constexpr int Bp = MB * 16;
    const int dt = sa.tab.dtype;
    const int K = a.nchain;
    uint32_t* abortw = a.sync + (size_t)K * PERSIST_SYNC_STRIDE;
    float* wred = lds + (size_t)NU * 2 * a.res_buf_words;      // [8 waves][MB][256] cross-wave reduction of the forward partial
    int* nxt = ldsw + 8;                                          // next step of my u-th unit

    ResUnit U[NU];
    f32x4 w4[NU][NTR], m4[NU][NTR], v4[NU][NTR];
    int cur[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int ui = wg + u * nwg;
        U[u].valid = ui < a.nres;
        U[u].index = ui;
        const SegDesc d = sa.desc[U[u].valid ? ui : wg];
        const CandDev& cd = sa.cands[d.cand];
        U[u].cand = d.cand; U[u].cell = d.cell; U[u].kind = d.kind; U[u].cc = d.cc; U[u].nkb = d.cc >> 4;
        U[u].S = X16 ? d.cc + 8 : d.cc + 4;       // row stride (elements) of a staged batch: 16 B aligned rows
        U[u].width = d.width; U[u].k0 = d.k0;
        U[u].tp = d.kind == KIND_S ? sa.tab.s[d.tap] : sa.tab.v[d.tap];
        const int64_t sbo = cd.step_off;
        U[u].part = sbo + sa.g.sb_part + (((int64_t)(cd.part_cell_off[d.cell] + d.part_idx) * MB) << 8);
        U[u].dyo = sbo + sa.g.sb_dy + (int64_t)d.cell * Bp * sa.g.Rp;      // dy_i [Bp][Rp = 16]
        U[u].gsco = sbo + sa.g.sb_gsc + d.cell * 2 + d.kind;
        U[u].Wp = sa.plane + d.w_off;
        U[u].Mp = U[u].Wp + sa.plane_stride;
        U[u].Vp = U[u].Mp + sa.plane_stride;
        U[u].gidx = cd.gidx;
        U[u].flag = PERSIST_FLAG(a.sync, d.cand);
        U[u].cnt = PERSIST_CNT(a.sync, d.cand);
        U[u].xb0 = (2 * u) * a.res_buf_words;
        U[u].xbw = a.res_buf_words;
        cur[u] = 0;
        // the unit's state: wave w owns k-blocks w, w + 8, ... (as sweep_body's k-split)
#pragma unroll
        for (int s = 0; s < NTR; ++s) {
            const int kb = wave + STEP_NW * s;
            const int64_t off = (int64_t)(kb < U[u].nkb ? kb : 0) * 256 + lane * 4;
            w4[u][s] = *reinterpret_cast<const f32x4*>(U[u].Wp + off);
            m4[u][s] = *reinterpret_cast<const f32x4*>(U[u].Mp + off);
            v4[u][s] = *reinterpret_cast<const f32x4*>(U[u].Vp + off);
        }
    }
    const int nj = (sa.g.B + 3) >> 2;      // batch blocks of 4 rows that hold data (tile_run)
    const float a_w1 = sa.ac.w1, a_b2 = sa.ac.b2, a_w2 = sa.ac.w2, a_eps = sa.ac.eps, a_wd = sa.ac.wd;   // scalars, not a struct copy (common.hip.h adam4)

    // MFMA operand reads from a staged batch: one element (dW: A[i = column][k = batch row]) / four consecutive columns
    constexpr bool DMA = X16 && (MFAS_RES_DMA != 0);      // k-block-major image [kb][row][16], staged per wave by LDS-DMA
    auto x1 = [&](const float* xb, int S, int row, int col) -> float {
        if constexpr (DMA) return cvt16(as_lds(reinterpret_cast<const uint16_t*>(xb))[(((col >> 4) * Bp + row) << 4) + (col & 15)], dt);
        else if constexpr (X16) return cvt16(as_lds(reinterpret_cast<const uint16_t*>(xb))[row * S + col], dt);
        else return as_lds(xb)[row * S + col];
    };
    auto x4of = [&](const float* xb, int S, int row, int col) -> f32x4 {
        if constexpr (X16) {
            const int at = DMA ? ((((col >> 4) * Bp + row) << 4) + (col & 15)) : (row * S + col);
            const u32x2 r = *as_lds(reinterpret_cast<const u32x2*>(reinterpret_cast<const uint16_t*>(xb) + at));
            return (f32x4){cvt16(r.x & 0xFFFFu, dt), cvt16(r.x >> 16, dt), cvt16(r.y & 0xFFFFu, dt), cvt16(r.y >> 16, dt)};
        } else {
            return *as_lds(reinterpret_cast<const f32x4*>(xb + row * S + col));
        }
    };
    const uint32_t lds_base = (uint32_t)(uintptr_t)as_lds(lds);
    const void* zeros = a.sync + (size_t)K * PERSIST_SYNC_STRIDE + 48;      // 64 bytes nobody writes during the launch (zeroed before it)
    auto stage = [&](const ResUnit& un, float* dst, int t) {     // rows of batch t -> LDS
        const int nv = (int)min((int64_t)a.B, a.N - (int64_t)t * a.B);
        if constexpr (DMA) {
            // lane -> (row, 16-byte half of the row's 32-byte k-block); this wave's k-blocks only
            const int r = lane >> 1, half = lane & 1;
            const int32_t* ord = cand_order(sa.order, sa.g, un.gidx);
            int64_t row = -1;
            if (r < nv) row = ord ? (int64_t)ord[a.pos0 + (int64_t)t * a.B + r] : (int64_t)t * a.B + r;
            const uint32_t dst0 = lds_base + (uint32_t)((dst - lds) << 2);
            if (r < Bp) {
#pragma unroll
                for (int sidx = 0; sidx < NTR; ++sidx) {
                    const int kb = wave + STEP_NW * sidx;
                    if (kb < un.nkb) {
                        const void* src = row >= 0 ? static_cast<const void*>(reinterpret_cast<const uint16_t*>(un.tp) + row * un.width + un.k0 + kb * 16 + half * 8)
                                                   : zeros;
                        glds16(src, __builtin_amdgcn_readfirstlane(dst0 + (uint32_t)(kb * Bp * 32)));
                    }
                }
            }
        } else if constexpr (X16)
            stage_table16(reinterpret_cast<uint16_t*>(dst), un.S, un.tp, un.width, un.k0, un.cc, cand_order(sa.order, sa.g, un.gidx), a.pos0 + (int64_t)t * a.B, t * a.B, nv, Bp, tid, STEP_THREADS);
        else
            stage_table(dst, un.S, un.tp, sa.tab.dtype, un.width, un.k0, un.cc, cand_order(sa.order, sa.g, un.gidx), a.pos0 + (int64_t)t * a.B, t * a.B, nv, Bp, tid, STEP_THREADS);
    };
    // the slab-storing waves (0 .. MB-1) wait for their stores' acknowledgements; the LAST of them counts the unit's arrival (LDS ticket)
    auto arrive = [&](uint32_t* cnt) {
        if (tid < MB * 64) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) {
                int old = 0;
                if constexpr (MB > 1) old = __hip_atomic_fetch_add(ldsw + 4, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (old == MB - 1) {
                    if constexpr (MB > 1) __hip_atomic_store(ldsw + 4, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    };
    // cross-wave reduction of a forward partial (fixed order 0..7, as sweep_body) -> partial slot (write-through) -> arrive
    auto reduce_publish = [&](const ResUnit& un, const f32x4 (&yacc)[MB], const int nv_next) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
            *reinterpret_cast<f32x4*>(wred + ((wave * MB + mb) << 8) + lane * 4) = yacc[mb];
        __syncthreads();
        for (int e = tid; e < MB * 64; e += STEP_THREADS) {
            const int slot = e >> 6, ln = e & 63;
            f32x4 sum = *reinterpret_cast<const f32x4*>(wred + (slot << 8) + ln * 4);
#pragma unroll
            for (int w = 1; w < STEP_NW; ++w)
                sum += *reinterpret_cast<const f32x4*>(wred + ((w * MB + slot) << 8) + ln * 4);
            // (transposed slabs: lane ln holds batch row slot * 16 + (ln & 15); the chain does not read rows beyond the batch)
            if (!MFAS_RES_TRANSPOSED_SLABS || slot * 16 + (ln & 15) < nv_next) stc4<true>(sa.stepbuf, un.part + (slot << 8) + ln * 4, sum);
        }
        // Only the MB waves that stored the slab wait for the stores' acknowledgements; the LAST of them to see its own arrive counts
        // the unit's arrival (an LDS ticket).  The other waves go on — the poller (wave 7) already looks for the workgroup's next
        // unit while waves 0 .. MB-1 drain: no workgroup barrier behind the publish any more.
        arrive(un.cnt);
    };
    // (Round 5's deferred unit hand-off — the pending arrival and the LDS-DMA copy of the batch after next finished while the workgroup
    //  already serves its other unit — was a measured negative, 15.1 / 15.5 -> 16.1 / 16.4 us per step at 22 / 28 candidates, and is gone:
    //  profiles/r05_defer_ab.log, DESIGN_HISTORY.md.)
    constexpr int POLL_TID = STEP_THREADS - 64;      // the unit loop's poller: lane 0 of the last wave (waves 0 .. MB-1 drain the slab stores)
    if (tid == 0) ldsw[4] = 0;

    // ---- prologue per unit: forward partial sums of batch 0 (no update), then batch 1 staged
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        if (U[u].valid) {            // (wave-uniform, workgroup-uniform)
            stage(U[u], lds + res_xbo(U[u], 0), 0);
            if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the LDS-DMA copy is invisible to the compiler's counters)
            __syncthreads();
            f32x4 yacc[MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) yacc[mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < NTR; ++s) {
                const int kb = wave + STEP_NW * s;
                if (kb < U[u].nkb) {
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) {
                        const f32x4 x4 = x4of(lds + res_xbo(U[u], 0), U[u].S, mb * 16 + l15, kb * 16 + 4 * lg);
#pragma unroll
                        for (int q = 0; q < 4; ++q) yacc[mb] = RES_FWD_MFMA(x4[q], w4[u][s][q], yacc[mb]);
                    }
                }
            }
            reduce_publish(U[u], yacc, (int)min((int64_t)a.B, a.N));
            if (1 < a.T) stage(U[u], lds + res_xbo(U[u], 1), 1);
        }
    }
    if (tid == POLL_TID) {
#pragma unroll
        for (int u = 0; u < NU; ++u) nxt[u] = U[u].valid ? 0 : a.T;
    }
    __syncthreads();

    // ---- the epoch's train steps: serve whichever of my units' candidates has published the step the unit waits for
    int last = NU - 1;
    for (;;) {
        if (tid == POLL_TID) {
            int pick = -2;   // -2: every unit has finished its last step
            uint32_t spins = 0;
            if constexpr (NU == 1) {
                // one unit: pipelined poll of its candidate's flag (wg_wait_ge's ring of three reads in flight)
                const int tj = nxt[0];
                if (tj < a.T) {
                    const uint32_t target = (uint32_t)(tj + 1);
                    const uint32_t* fl = U[0].flag;
                    uint32_t v0 = ld_u32_relaxed(fl);
                    __builtin_amdgcn_s_sleep(2);
                    uint32_t v1 = ld_u32_relaxed(fl);
                    __builtin_amdgcn_s_sleep(2);
                    uint32_t v2 = ld_u32_relaxed(fl);
                    pick = 0;
                    for (;;) {
                        if (v0 >= target) break;
                        __builtin_amdgcn_s_sleep(2);
                        v0 = ld_u32_relaxed(fl);
                        if (v1 >= target) break;
                        __builtin_amdgcn_s_sleep(2);
                        v1 = ld_u32_relaxed(fl);
                        if (v2 >= target) break;
                        __builtin_amdgcn_s_sleep(2);
                        v2 = ld_u32_relaxed(fl);
                        if (((spins += 3) & 0x3FFu) < 3 && (spins > PERSIST_RING_LIMIT || ld_u32_relaxed(abortw) != 0)) {
                            __hip_atomic_store(abortw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            pick = -3;
                            break;
                        }
                    }
                }
            } else
            for (;;) {
                bool pending = false;
                uint32_t fv[NU];      // every unit's flag requested before the first is tested: the reads overlap
#pragma unroll
                for (int u = 0; u < NU; ++u) fv[u] = ld_u32_relaxed(U[u].flag);
#pragma unroll
                for (int q = 1; q <= NU; ++q) {
                    const int j = (last + q) % NU;
                    const int tj = nxt[j];
                    if (tj < a.T && pick < 0) {
                        pending = true;
                        uint32_t fj = fv[0];
#pragma unroll
                        for (int u = 1; u < NU; ++u) if (j == u) fj = fv[u];
                        if (fj >= (uint32_t)(tj + 1)) pick = j;
                    }
                }
                if (pick >= 0 || !pending) break;
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 0x3FFu) == 0 && (spins > PERSIST_SPIN_LIMIT || ld_u32_relaxed(abortw) != 0)) {
                    __hip_atomic_store(abortw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    pick = -3;
                    break;
                }
            }
            ldsw[0] = pick;
        }
        __syncthreads();
        const int pick = ldsw[0];
        if (pick == -3) return;
        if (pick < 0) break;
        const int t = nxt[pick];
        __syncthreads();   // everyone has read the pick / step before lane 0 can overwrite them
        last = pick;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            if (pick == u) {
                const ResUnit& un = U[u];
                const bool fwd = t + 1 < a.T;
                const float* xt = lds + res_xbo(un, cur[u]);
                const float* xn = lds + res_xbo(un, cur[u] ^ 1);
                const bool tr_on = un.index == 0 && tid == 0 && t >= 8 && t < 16;
                const int tr_base = (t - 8) * 8 + 4;
                PTRACE(1);
                PTRACE_UNIT(64);    // saw the flag
                float dyf[MB * 4];
#pragma unroll
                for (int j = 0; j < MB * 4; ++j) dyf[j] = ldc1<true>(sa.stepbuf + un.dyo + (4 * j + lg) * 16 + l15);
                float gsc = 1.0f;
                if (sa.g.alphas) gsc = ldc1<true>(sa.stepbuf + un.gsco);
                const float a_ss = a.scal[2 * (int64_t)(a.gstep0 + t)], a_bc2s = a.scal[2 * (int64_t)(a.gstep0 + t) + 1];
                // this wave's own LDS-DMA copies (batch t+1, requested after the unit's previous step) have landed: nothing else reads them
                if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                f32x4 yacc[MB];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) yacc[mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < NTR; ++s) {
                    const int kb = wave + STEP_NW * s;
                    if (kb < un.nkb) {
                        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                        DW_BATCH_LOOP(MB, nj, acc = MFMA16(x1(xt, un.S, 4 * j + lg, kb * 16 + l15), dyf[j], acc))
                        { f32x4 w = w4[u][s], m = m4[u][s], v = v4[u][s];
                          adam4(w, m, v, acc * gsc, a_ss, a_bc2s, a_w1, a_b2, a_w2, a_eps, a_wd);
                          w4[u][s] = w; m4[u][s] = m; v4[u][s] = v; }
                        if (fwd) {
#pragma unroll
                            for (int mb = 0; mb < MB; ++mb) {
                                const f32x4 x4 = x4of(xn, un.S, mb * 16 + l15, kb * 16 + 4 * lg);
#pragma unroll
                                for (int q = 0; q < 4; ++q) yacc[mb] = RES_FWD_MFMA(x4[q], w4[u][s][q], yacc[mb]);
                            }
                        }
                    }
                }
                PTRACE(2);
                PTRACE_UNIT(128);   // compute done
                if (fwd) {
                    reduce_publish(un, yacc, (int)min((int64_t)a.B, a.N - (int64_t)(t + 1) * a.B));
                } else {
                    wg_publish_barrier();
                    if (tid == 0) __hip_atomic_fetch_add(un.cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                PTRACE(3);
                PTRACE_UNIT(192);   // arrived
                if (tid == POLL_TID) nxt[u] = t + 1;
                cur[u] ^= 1;
                // batch t+2 into the buffer batch t just vacated: it lands while this unit's chain runs step t+1
                if (t + 2 < a.T) stage(un, lds + res_xbo(un, cur[u] ^ 1), t + 2);
            }
        }
    }
    // ---- state back to memory (dev evaluation, parameter export and the next epoch's launch read it there)
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        if (U[u].valid) {
#pragma unroll
            for (int s = 0; s < NTR; ++s) {
                const int kb = wave + STEP_NW * s;
                if (kb < U[u].nkb) {
                    const int64_t off = (int64_t)kb * 256 + lane * 4;
                    *reinterpret_cast<f32x4*>(U[u].Wp + off) = w4[u][s];
                    *reinterpret_cast<f32x4*>(U[u].Mp + off) = m4[u][s];
                    *reinterpret_cast<f32x4*>(U[u].Vp + off) = v4[u][s];
                }
            }
        }
    }
}

#define PERSIST_NTR 4                   // resident units, f32 staging: tiles per wave (cc <= 512 columns)
#define PERSIST_NTR16 8                 // resident units, 16-bit staging: cc <= 1024 columns

// Roll call: the loops below are only deadlock-free when EVERY workgroup of the schedule is resident at the same time.  That holds
// when the process owns the GPU (workgroups <= #CUs, one per CU); when another process's kernels hold CUs (or the device is CU-masked),
// part of the workgroups may be waiting for a slot that the resident part —
// spinning on it — never frees.  So nobody touches any state before all `total` workgroups have checked in; if that does not
// happen within ~5 ms the resident ones leave (abort code 2), the late ones see the code and leave too, and the host relaunches
// the epoch (and, when that keeps failing, falls back to the launch-per-phase schedule: mfas_hip.hip::persist_fallback).
// (count and verdict live in ONE word, so "everybody is here" and "somebody gave up" cannot both be observed)
__device__ __forceinline__ bool persist_roll_call(uint32_t* sync, const int K, const uint32_t total, int* ldsw) {
    if (threadIdx.x == 0) {
        uint32_t* roll = PERSIST_ROLL(sync, K);
        uint32_t* abortw = sync + (size_t)K * PERSIST_SYNC_STRIDE;
        constexpr uint32_t GAVE_UP = 0x80000000u;
        int ok = -1;
        if (__hip_atomic_fetch_add(roll, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & GAVE_UP) ok = 0;
        uint32_t spins = 0;
        while (ok < 0) {
            const uint32_t v = ld_u32_relaxed(roll);
            if (v & GAVE_UP) ok = 0;
            else if (v >= total) ok = 1;
            else if (++spins > PERSIST_ROLL_LIMIT) {
                uint32_t expect = v;
                if (__hip_atomic_compare_exchange_strong(roll, &expect, v | GAVE_UP, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(abortw, PERSIST_ABORT_NOT_RESIDENT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = 0;
                }
            } else __builtin_amdgcn_s_sleep(8);
        }
        ldsw[1] = ok;
    }
    __syncthreads();
    return ldsw[1] != 0;
}

// ------------------------------------------------------------------------------------------------
// k_president<MB, NTR, X16, NU, PLAIN> — the RESIDENT schedule (the default for small populations at R <= 16): ONE launch per epoch,
// blocks [0, K) = the resident lean chain of candidate blockIdx.x, blocks [K, K + nres_wg) = workgroups of resident feature
// units.  One instantiation per unit form (staging width, tiles per wave, units per workgroup): an instantiation carries exactly
// the two bodies its grid runs.  (Round 3 also ran the two roles as two kernels on two streams, each with its own register budget — the unit
// kernels then need 109-184 VGPRs and no scratch, the chain kernel 231-255: the step time did not move (15.4-15.5 us at 4-8
// candidates, profiles/r03_popsweep_split_kernels.log), and co-residency of two launches depends on the streams landing on different
// hardware queues, which HIP does not promise: after a few hundred stream creations in one process the second launch queued
// behind the first and every roll call failed.  One launch cannot be split by the runtime.)
// ------------------------------------------------------------------------------------------------
template <int MB, int NTR, bool X16, int NU, int PLAIN>
__global__ void __launch_bounds__(STEP_THREADS, 2) k_president(const PersistArgs a, const int lds_word) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int* ldsw = reinterpret_cast<int*>(lds) + lds_word;
    // blockIdx.x round-robins over the 8 XCDs: the host deals the roles so that a candidate's chain and its unit workgroups share an XCD
    // (and its L2) where they fit — placement only, every exchange stays placement-independent (sc1 / write-through)
    const int bid = a.role ? a.role[blockIdx.x] : (int)blockIdx.x, tid = threadIdx.x;
    const int K = a.nchain;
    if (!persist_roll_call(a.sync, K, gridDim.x, ldsw)) return;
    if (bid >= K) {
        sweep_resident<MB, NTR, X16, NU>(a, bid - K, a.nres_wg, lds, ldsw);
        return;
    }
    uint32_t* abortw = a.sync + (size_t)K * PERSIST_SYNC_STRIDE;
    const uint32_t need = (uint32_t)a.need[bid];
    LeanRes& rs = *reinterpret_cast<LeanRes*>(ldsw + 16);   // the epoch's running statistics: LDS words 16..21 (as registers of one lane
                                                             // they were live across the whole step loop in every wave, and spilled)
    lean_res_load<MB>(a.ca, bid, lds, rs);
    const LeanPre lpre = lean_pre<MB>(a.ca, bid);
    uint32_t keep = lean_keep_bits<MB>(a.ca.cands[bid], a.ca.g, a.gstep0);     // dropout keep-bits and labels of the step about to run
    int labp = lean_label<MB>(a.ca, a.ca.cands[bid], a.pos0, 0, (int)min((int64_t)a.B, a.N));
    for (int t = 0; t < a.T; ++t) {
        const bool tr_on = bid == 0 && tid == 0 && t >= 8 && t < 16;
        const int tr_base = (t - 8) * 8;
        PTRACE(0);
        // (the step's scalars — two loads from the Adam table among them — are requested BEFORE the wait for the units, not behind it)
        ChainStep cs;
        cs.pos_t = a.pos0 + (int64_t)t * a.B;
        cs.base_t = t * a.B;
        cs.nvalid = (int)min((int64_t)a.B, a.N - (int64_t)t * a.B);
        cs.gstep = a.gstep0 + t;
        cs.epoch = a.epoch;
        cs.ss = a.scal[2 * (int64_t)cs.gstep];
        cs.bc2s = a.scal[2 * (int64_t)cs.gstep + 1];
        if (!wg_wait_ge(PERSIST_CNT(a.sync, bid), need * (uint32_t)(t + 1), abortw, ldsw)) return;
        PTRACE(1);
        chain_lean<MB, 2, 16, PLAIN>(a.ca, cs, bid, lds, lpre, keep, labp);
        PTRACE(2);
        wg_publish_barrier();
#ifdef MFAS_TEST_HOOKS
        if (tid == 0 && !(bid == 0 && t == a.lose_step))
#else
        if (tid == 0)
#endif
            __hip_atomic_store(PERSIST_FLAG(a.sync, bid), (uint32_t)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        PTRACE(3);
        keep = lean_keep_bits<MB>(a.ca.cands[bid], a.ca.g, cs.gstep + 1);      // (next step's: ~200 integer instructions off the critical path)
        if (t + 1 < a.T) labp = lean_label<MB>(a.ca, a.ca.cands[bid], cs.pos_t + a.B, cs.base_t + a.B, (int)min((int64_t)a.B, a.N - (int64_t)(t + 1) * a.B));
        chain_lean_tail<MB, 2>(a.ca, cs, bid, lds, &rs);   // statistics + vector-parameter Adam, after dy is out
        lean_res_update<MB>(a.ca, cs, bid, lds);           // OUT / HEAD dW + Adam while the feature units run
    }
    lean_res_store<MB>(a.ca, bid, a.epoch, lds, rs);
}

// persist.hip.h — the persistent step loop for SMALL populations (the regime the search really runs in: 6-16 candidates per
// GPU, models/searchable.py:90,120 issue calls of <= 50 configurations over up to 8 GPUs).
// (part of the single translation unit mfas_hip.hip; see the header comment there and DESIGN.md)
#pragma once
// ------------------------------------------------------------------------------------------------
// With a handful of candidates a train step is a latency chain, not a bandwidth problem: chain(c,t) -> sweep(c,t) ->
// chain(c,t+1), and the launch-per-phase schedule (k_chain / k_step) additionally serialises ALL candidates' sweeps against
// ALL chains at every kernel boundary.  k_persist is ONE launch per epoch: every workgroup is resident (one per CU) and
// loops over the epoch's train steps; dependencies are per CANDIDATE:
//   * chain workgroup c (blocks [0, K)) waits until the arrival counter cnt[c] shows that every sweep unit of candidate c
//     has finished step t-1 (its W/m/v update and the forward partial sums of batch t), runs the chain of step t and
//     publishes flag[c] = t + 1;
//   * sweep workgroups (the remaining blocks; unit i is owned by workgroup i mod G, fixed for the whole launch) wait for
//     flag[cand(i)] >= t + 1, run the unit's dW + Adam + next-step forward and arrive on cnt[cand(i)].
// So while candidate A sits in its serial chain, the CUs stream the sweeps of the candidates whose chains have finished.
// Data exchanged inside the launch (dy / out_i / dlogits from the chain; partial sums and the OUT / HEAD weight tiles from
// the sweep) is stored write-through and loaded with sc1 (COH helpers, common.hip.h): no release / acquire fences, whose
// L2 write-back would drag the whole XCD's dirty W/m/v lines along.  Every storing wave drains (`s_waitcnt vmcnt(0)`) before the
// workgroup barrier that precedes the relaxed agent-scope flag store / counter add; pollers are ONE lane per workgroup,
// relaxed loads + s_sleep, bounded (a timeout sets the abort word, every workgroup leaves, the host reports an error).
// Reduction orders, tile decomposition and arithmetic are those of the launch-per-phase schedule: results are bit-identical.
// ------------------------------------------------------------------------------------------------
struct PersistArgs {
    SweepArgs sa;              // desc = the population's sweep units (per-segment), cands = all candidates
    ChainArgs ca;              // cands = all candidates
    int32_t nchain, nitems;    // K, number of sweep units
    int32_t T, epoch;          // train steps of this launch, epoch index (statistics slot)
    int64_t N, pos0;           // N_train, epoch * N_train (position in the sample-order table)
    int32_t B, gstep0;         // batch size, epoch * batches-per-epoch (Adam / dropout step counter base)
    const float* scal;         // device [steps][2]: {lr_t/(1-beta1^t), sqrt(1-beta2^t)}
    uint32_t* sync;            // [0,K) flag: steps whose dy is published; [K,2K) cnt: unit arrivals; [2K] abort word
    const int32_t* need;       // [K] sweep units of candidate c (= arrivals per step)
    unsigned long long* trace; // optional: 100 MHz timestamps of candidate 0's chain and of sweep unit 0 (steps 8..15)
};

#define PERSIST_MAX_UNITS 8             // sweep units one workgroup may own
#define PERSIST_LDS_WORDS 32            // LDS words the loop itself uses (behind the bodies' LDS)
#define PERSIST_SPIN_LIMIT (1u << 22)   // a few seconds of s_sleep polls: only a lost workgroup or a bug gets here

__device__ __forceinline__ uint32_t ld_u32_relaxed(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Workgroup-wide wait until *p >= target: lane 0 polls (relaxed, sc1), everyone else parks at the barrier.
// Returns false when the launch is being aborted.  `ldsw` = one LDS word outside the bodies' LDS footprint.
__device__ __forceinline__ bool wg_wait_ge(const uint32_t* p, uint32_t target, uint32_t* abortw, int* ldsw) {
    if (threadIdx.x == 0) {
        int ok = 1;
        uint32_t spins = 0;
        while (ld_u32_relaxed(p) < target) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 0x3FFu) == 0 && (spins > PERSIST_SPIN_LIMIT || ld_u32_relaxed(abortw) != 0)) {
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

// every storing wave has drained its (write-through) stores when the barrier releases; then ONE lane signals
__device__ __forceinline__ void wg_publish_barrier() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

#define PTRACE(slot) do { if (a.trace && tr_on) a.trace[tr_base + (slot)] = wall_clock64(); } while (0)

template <int MB, bool LEAN, int U>
__global__ void __launch_bounds__(STEP_THREADS, 2) k_persist(const PersistArgs a, const int lds_word) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int* ldsw = reinterpret_cast<int*>(lds) + lds_word;
    const int bid = (int)blockIdx.x, tid = threadIdx.x;
    const int K = a.nchain;
    uint32_t* flag = a.sync;
    uint32_t* cnt = a.sync + K;
    uint32_t* abortw = a.sync + 2 * K;

    if (bid < K) {
        // ------------------------------------------------------------------ chain workgroup of candidate `bid`
        const uint32_t need = (uint32_t)a.need[bid];
        for (int t = 0; t < a.T; ++t) {
            const bool tr_on = bid == 0 && tid == 0 && t >= 8 && t < 16;
            const int tr_base = (t - 8) * 8;
            PTRACE(0);
            if (!wg_wait_ge(cnt + bid, need * (uint32_t)(t + 1), abortw, ldsw)) return;
            PTRACE(1);
            ChainStep cs;
            cs.pos_t = a.pos0 + (int64_t)t * a.B;
            cs.base_t = t * a.B;
            cs.nvalid = (int)min((int64_t)a.B, a.N - (int64_t)t * a.B);
            cs.gstep = a.gstep0 + t;
            cs.epoch = a.epoch;
            cs.ss = a.scal[2 * (int64_t)cs.gstep];
            cs.bc2s = a.scal[2 * (int64_t)cs.gstep + 1];
            if constexpr (LEAN) chain_lean<MB, true>(a.ca, cs, bid, lds);
            else chain_body<MB, true, true>(a.ca, cs, bid, lds);
            PTRACE(2);
            wg_publish_barrier();
            if (tid == 0) __hip_atomic_store(flag + bid, (uint32_t)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            PTRACE(3);
        }
        return;
    }
    // ---------------------------------------------------------------------- sweep workgroup: owns units wg, wg + G, ...
    // Ownership is FIXED for the whole launch: a unit's W/m/v stream through plain (not write-through) stores and are only
    // ever re-read by the same CU, so they stay coherent without any fence.  A workgroup that owns several units serves
    // whichever of them is ready (its candidate's chain has published the step the unit is waiting for): units of different
    // candidates never block each other (in-order service convoys all candidates behind the slowest chain).
    const int G = (int)gridDim.x - K, wg = bid - K;
    const int n_my = wg < a.nitems ? (a.nitems - wg + G - 1) / G : 0;   // <= PERSIST_MAX_UNITS (host)
    int* nxt = ldsw + 8;                     // next step of my j-th unit (-1 = the epoch's prologue: forward of batch 0, no update)
    int* cnd = ldsw + 8 + PERSIST_MAX_UNITS; // its candidate
    if (tid < n_my) {
        nxt[tid] = -1;
        cnd[tid] = a.sa.desc[wg + tid * G].cand;
    }
    __syncthreads();
    int last = n_my - 1;
    for (;;) {
        if (tid == 0) {
            int pick = -2;   // -2: every unit has finished its last step
            uint32_t spins = 0;
            for (;;) {
                bool pending = false;
                for (int q = 1; q <= n_my; ++q) {
                    const int j = (last + q) % n_my;
                    const int tj = nxt[j];
                    if (tj >= a.T) continue;
                    pending = true;
                    if (tj < 0 || ld_u32_relaxed(flag + cnd[j]) >= (uint32_t)(tj + 1)) { pick = j; break; }
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
        if (pick < 0) return;
        const int t = nxt[pick], cand = cnd[pick], it = wg + pick * G;
        __syncthreads();   // everyone has read the pick before lane 0 can overwrite it
        last = pick;
        SweepStep st;
        const int tt = t < 0 ? 0 : t;
        st.upd = t >= 0;
        st.fwd = t + 1 < a.T;
        st.pos_t = a.pos0 + (int64_t)tt * a.B;
        st.base_t = tt * a.B;
        st.nvalid_t = (int)min((int64_t)a.B, a.N - (int64_t)tt * a.B);
        const int tn = t + 1;
        st.pos_n = a.pos0 + (int64_t)tn * a.B;
        st.base_n = tn * a.B;
        st.nvalid_n = (int)min((int64_t)a.B, a.N - (int64_t)tn * a.B);
        st.ss = t >= 0 ? a.scal[2 * (int64_t)(a.gstep0 + t)] : 0.f;
        st.bc2s = t >= 0 ? a.scal[2 * (int64_t)(a.gstep0 + t) + 1] : 1.f;
        const bool tr_on = it == 0 && tid == 0 && t >= 8 && t < 16;
        const int tr_base = (t - 8) * 8 + 4;
        PTRACE(1);
        sweep_body<MB, false, U, true>(a.sa, st, it, lds);
        PTRACE(2);
        wg_publish_barrier();
        if (tid == 0) {
            __hip_atomic_fetch_add(cnt + cand, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            nxt[pick] = t + 1;
        }
        PTRACE(3);
    }
}

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
// Source layout (ONE translation unit; this file includes the rest): common.hip.h (device records, helpers, LDS
// staging), sweep.hip.h (tile_run, sweep_body, sweep_tap_body), chain.hip.h (chain_body, chain_lean, softmax / BCE rows),
// eval.hip.h (k_eval), pack.hip.h (k_pack, k_vec, k_pool, k_stream_probe); below: k_step / k_chain and the host C ABI.
// Dev evaluation is row-parallel (k_eval).  Weights live in a 16x16 tile-major layout that is exactly the
// MFMA 16x16x4 f32 operand layout, so every W/m/v access is one coalesced 16 B/lane load.
//
// MFMA used: v_mfma_f32_16x16x4_f32 (exact f32 fma chain).  Layout (lane l):
//   A[i = l&15][k = l>>4],  B[k = l>>4][j = l&15],  D[i = 4*(l>>4)+reg][j = l&15].
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
// No implicit FMA contraction anywhere in this translation unit (see __graft_entry__.build): every schedule (k_chain / k_step /
// k_president instantiations) must round identically.  MFMA instructions are unaffected.
#pragma clang fp contract(off)
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <algorithm>
#include <string>
#include <vector>
#include <thread>
#include <chrono>
#include <dlfcn.h>

#include "mfas_hip.h"

#include "common.hip.h"
#include "sweep.hip.h"
#include "chain.hip.h"

// ------------------------------------------------------------------------------------------------
// k_step — ONE launch per half-step: blocks [0, nchain) run the chain of one candidate group while the other
// blocks run the sweep of the OTHER group (candidates are independent).  The latency-bound chain hides under
// the HBM-bound sweep; kernel boundaries carry every dependency (chain(t) -> sweep(t) -> chain(t+1) of a group).
// ------------------------------------------------------------------------------------------------
struct StepArgs {
    SweepArgs sa;
    ChainArgs ca;
    int32_t nchain, _pad;
    GatherArgs ga;          // nblocks gather workgroups (one per chain candidate) right after the chain blocks
};

// NS > 1 (its own instantiation: the headline k_step<1, true, 4, false> carries none of it): the chain blocks are chain_split parts,
// block b = part * (nchain / NS) + candidate (k_step_same below).
template <int MB, bool NT, int WPE, bool LEAN, int NS = 1>
__global__ void __launch_bounds__(STEP_THREADS, WPE) k_step(const StepArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int bid = (int)blockIdx.x;
    if (bid < a.nchain) {
        if constexpr (NS > 1) {
            const int Kp = a.nchain / NS, part = bid / Kp, c = bid - part * Kp;
            if (c < a.ca.ncand) chain_split<NS>(a.ca, chain_step_of(a.ca), c, part, lds);
        } else
        if constexpr (LEAN) { chain_lean<MB, 0, (WPE >= 4 ? 8 : 16)>(a.ca, chain_step_of(a.ca), bid, lds, lean_pre<MB>(a.ca, bid)); chain_lean_tail<MB, 0>(a.ca, chain_step_of(a.ca), bid, lds); }
        else chain_body<MB, false>(a.ca, chain_step_of(a.ca), bid, lds);
    }
    else if (bid < a.nchain + a.ga.nblocks) gather_body(a.ga, a.ga.cands[bid - a.nchain], a.sa.g, a.sa.tab, a.sa.order, (int)threadIdx.x);
    else if (bid < a.nchain + a.ga.nblocks + a.sa.ntap) sweep_tap_body<MB, NT, SweepU<MB, WPE>::v>(a.sa, bid - a.nchain - a.ga.nblocks, lds);
    else if (!LEAN && a.sa.desc[bid - a.nchain - a.ga.nblocks - a.sa.ntap].nsub > 1)      // multi-chunk feature unit (general chain, 8 row blocks)
        sweep_multi_body<MB, NT, SweepU<MB, WPE>::v>(a.sa, sweep_step_of(a.sa), bid - a.nchain - a.ga.nblocks - a.sa.ntap, lds);
    else sweep_body<MB, NT, SweepU<MB, WPE>::v>(a.sa, sweep_step_of(a.sa), bid - a.nchain - a.ga.nblocks - a.sa.ntap, lds);
}

// Same-group fused launch (small populations with the general chain, R >= 128): chain blocks AND sweep blocks of the SAME
// candidates in one launch; a sweep unit waits for its cell's dy (per-cell flags published by the chain as the backward pass
// reaches the cell) instead of for a kernel boundary, so the sweeps of cells L-1 .. 1 overlap the rest of the backward pass.
// The work list is ordered last cell first; chain blocks have the lowest block indices (dispatched first).
// NS > 1 (MB = 1, eight row blocks): every candidate's chain runs on NS workgroups (chain_split): chain block b = part * Kp + candidate
// with Kp = the candidate count rounded up to 8 — block b lands on XCD b % 8, so a candidate's parts share an XCD and its L2.
template <int MB, bool NT, int NS = 1>
__global__ void __launch_bounds__(STEP_THREADS, 4) k_step_same(const StepArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int bid = (int)blockIdx.x;
    if (bid < a.nchain) {
        if constexpr (NS > 1) {
            const int Kp = a.nchain / NS, part = bid / Kp, c = bid - part * Kp;
            if (c < a.ca.ncand) chain_split<NS>(a.ca, chain_step_of(a.ca), c, part, lds);
        } else chain_body<MB, false, true>(a.ca, chain_step_of(a.ca), bid, lds);
    } else sweep_body<MB, NT, SweepU<MB, 4>::v, true>(a.sa, sweep_step_of(a.sa), bid - a.nchain, lds);
}

// Standalone chain launch (small populations: chain and sweep run back to back, so the chain's latency is on the
// critical path): full register budget, next-product weight tiles prefetched into registers.
template <int MB, bool LEAN>
__global__ void __launch_bounds__(STEP_THREADS, 2) k_chain(const ChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if constexpr (LEAN) { chain_lean<MB>(a, chain_step_of(a), (int)blockIdx.x, lds, lean_pre<MB>(a, (int)blockIdx.x)); chain_lean_tail<MB, 0>(a, chain_step_of(a), (int)blockIdx.x, lds); }
    else chain_body<MB, true>(a, chain_step_of(a), (int)blockIdx.x, lds);
}

#include "persist.hip.h"
#include "eval.hip.h"
#include "pack.hip.h"

// ================================================================================================
// Tuning — EVERY environment switch of the library, parsed in ONE place (tuning_from_env) when a population is created or planned
// and kept in the population: nothing else in this translation unit calls getenv, so a variable that changes after create() cannot
// change the schedule of a population that was laid out (and parity-tested) without it.  An empty environment gives the defaults
// below = the configuration the test suites run; INTEGRATION.md lists the switches, mfas_tuning_describe() prints the parsed set
// (tests/test_host_cpu.py::test_empty_environment_selects_the_tested_defaults).  They are A/B and debugging aids, not API.
// ================================================================================================
struct Tuning {
    int persist = -1;               // MFAS_PERSIST            0: never take the resident schedule (k_president); unset / 1: where it fits
    int no_lean_chain = 0;          // MFAS_NO_LEAN_CHAIN      general chain_body also at R <= 16
    int persist_no_resident = 0;    // MFAS_PERSIST_NO_RESIDENT   } either one: no resident units, i.e. launch per phase
    int persist_no_res_chain = 0;   // MFAS_PERSIST_NO_RES_CHAIN  }
    int subchunks = 0;              // MFAS_SUBCHUNKS=n        multi-chunk sweep units (measured negative, opt-in)
    int subchunk_skip = 0;          // MFAS_SUBCHUNK_SKIP=n    every n-th candidate keeps one-chunk units
    int groups = 0;                 // MFAS_GROUPS=1|2         force one / two candidate groups (0: by population size)
    int same_group = -1;            // MFAS_SAME_GROUP         0: never k_step_same, 2: whatever the size (-1: by state bytes)
    int no_tap_major = 0;           // MFAS_NO_TAP_MAJOR       per-segment sweep units also at R < 128
    int force_tap_major = 0;        // MFAS_FORCE_TAP_MAJOR    tap-major units even with < 192 workgroups
    int no_red_in_sweep = 0;        // MFAS_NO_RED_IN_SWEEP    the chain reduces the partial slabs itself
    int force_red_in_sweep = 0;     // MFAS_FORCE_RED_IN_SWEEP reduce-in-sweep whatever the population size (A/B: measured negative at 128 candidates)
    double occ_bytes = -1.0;        // MFAS_OCC_BYTES          MB == 2: crossover between the 2- and 4-waves-per-SIMD sweep builds
    int no_xcd_placement = 0;       // MFAS_NO_XCD_PLACEMENT   resident launch: block b runs role b
    int n_xcd = 0;                  // MFAS_XCDS=n             XCDs the placement assumes (0: 8 — MI355X in SPX mode)
    int persist_trace = 0;          // MFAS_PERSIST_TRACE      allocate the step-phase timestamp buffer
    int nt = -1;                    // MFAS_NT=0|1             force cached / nontemporal W/m/v streaming (-1: by plane size)
    int eval_no_x16 = 0;            // MFAS_EVAL_NO_X16        dev pass: f32 row tiles also over bf16 tables
    int eval_no_msplit = 0;         // MFAS_EVAL_NO_MSPLIT     dev pass at R <= 32: one wave per row block
    int eval_no_b3 = 0;             // MFAS_EVAL_NO_B3         dev pass at R = 72..128: f32 feature products
    int eval_no_wl = 0;             // MFAS_EVAL_NO_WL         dev pass at R <= 16: every wait for the LDS weight tiles is vmcnt(0) (rows one chunk deep)
    int no_gather = 0;              // MFAS_NO_GATHER          feature units stage rows through the order table
    int gather_verbose = 0;         // MFAS_GATHER_VERBOSE
    int no_plain_chain = 0;         // MFAS_NO_PLAIN_CHAIN     resident chain: the general (BN / alphas / multitask capable) instantiation
    int persist_verbose = 0;        // MFAS_PERSIST_VERBOSE=1|2
    int prof_every = 16;            // MFAS_PROF_EVERY=n       HIP events around every n-th sweep launch when profiling is on
    int chain_split = -1;           // MFAS_CHAIN_SPLIT=0|2|4  R = 128 general chain over that many CUs per candidate (-1: the planner decides)
    // test hooks: parsed only by the -DMFAS_TEST_HOOKS build variant (__graft_entry__.build_variant("hooks", ...)); the product library
    // never reads these variables
    int test_not_resident = -1;     // MFAS_PERSIST_TEST_NOT_RESIDENT=e   every roll call from epoch e on "fails"
    int test_lose_step = -1;        // MFAS_PERSIST_TEST_LOSE_STEP=t      candidate 0's chain never publishes step t
};

static Tuning tuning_from_env() {
    Tuning t;
    auto flag = [](const char* n) { return getenv(n) ? 1 : 0; };
    auto num = [](const char* n, int dflt) { const char* e = getenv(n); return e ? atoi(e) : dflt; };
    t.persist = getenv("MFAS_PERSIST") ? (atoi(getenv("MFAS_PERSIST")) != 0 ? 1 : 0) : -1;
    t.no_lean_chain = flag("MFAS_NO_LEAN_CHAIN");
    t.persist_no_resident = flag("MFAS_PERSIST_NO_RESIDENT");
    t.persist_no_res_chain = flag("MFAS_PERSIST_NO_RES_CHAIN");
    t.subchunks = num("MFAS_SUBCHUNKS", 0);
    t.subchunk_skip = num("MFAS_SUBCHUNK_SKIP", 0);
    t.groups = num("MFAS_GROUPS", 0);
    t.same_group = num("MFAS_SAME_GROUP", -1);
    t.no_tap_major = flag("MFAS_NO_TAP_MAJOR");
    t.force_tap_major = flag("MFAS_FORCE_TAP_MAJOR");
    t.no_red_in_sweep = flag("MFAS_NO_RED_IN_SWEEP");
    t.force_red_in_sweep = flag("MFAS_FORCE_RED_IN_SWEEP");
    if (const char* e = getenv("MFAS_OCC_BYTES")) t.occ_bytes = atof(e);
    t.no_xcd_placement = flag("MFAS_NO_XCD_PLACEMENT");
    t.n_xcd = num("MFAS_XCDS", 0);
    t.persist_trace = flag("MFAS_PERSIST_TRACE");
    t.nt = getenv("MFAS_NT") ? (atoi(getenv("MFAS_NT")) != 0 ? 1 : 0) : -1;
    t.eval_no_x16 = flag("MFAS_EVAL_NO_X16");
    t.eval_no_msplit = flag("MFAS_EVAL_NO_MSPLIT");
    t.eval_no_b3 = flag("MFAS_EVAL_NO_B3");
    t.eval_no_wl = flag("MFAS_EVAL_NO_WL");
    t.no_gather = flag("MFAS_NO_GATHER");
    t.gather_verbose = flag("MFAS_GATHER_VERBOSE");
    t.no_plain_chain = flag("MFAS_NO_PLAIN_CHAIN");
    t.persist_verbose = num("MFAS_PERSIST_VERBOSE", 0);
    t.prof_every = std::max(1, num("MFAS_PROF_EVERY", 16));
    t.chain_split = num("MFAS_CHAIN_SPLIT", -1);
#ifdef MFAS_TEST_HOOKS
    t.test_not_resident = num("MFAS_PERSIST_TEST_NOT_RESIDENT", -1);
    t.test_lose_step = num("MFAS_PERSIST_TEST_LOSE_STEP", -1);
#endif
    return t;
}

// "name=value ..." of the switches as parsed from the CURRENT environment, in declaration order (hooks=1 marks the test-hook variant)
extern "C" int mfas_tuning_describe(char* buf, int32_t cap) {
    if (!buf || cap <= 0) return MFAS_EINVAL;
    const Tuning t = tuning_from_env();
    char tmp[1024];
    snprintf(tmp, sizeof(tmp),
             "persist=%d no_lean_chain=%d persist_no_resident=%d persist_no_res_chain=%d subchunks=%d subchunk_skip=%d groups=%d same_group=%d "
             "no_tap_major=%d force_tap_major=%d no_red_in_sweep=%d force_red_in_sweep=%d occ_bytes=%g no_xcd_placement=%d n_xcd=%d persist_trace=%d nt=%d eval_no_x16=%d "
             "eval_no_msplit=%d eval_no_b3=%d eval_no_wl=%d no_gather=%d gather_verbose=%d no_plain_chain=%d persist_verbose=%d prof_every=%d "
             "chain_split=%d test_not_resident=%d test_lose_step=%d hooks=%d",
             t.persist, t.no_lean_chain, t.persist_no_resident, t.persist_no_res_chain, t.subchunks, t.subchunk_skip, t.groups, t.same_group,
             t.no_tap_major, t.force_tap_major, t.no_red_in_sweep, t.force_red_in_sweep, t.occ_bytes, t.no_xcd_placement, t.n_xcd, t.persist_trace, t.nt, t.eval_no_x16,
             t.eval_no_msplit, t.eval_no_b3, t.eval_no_wl, t.no_gather, t.gather_verbose, t.no_plain_chain, t.persist_verbose, t.prof_every,
             t.chain_split, t.test_not_resident, t.test_lose_step,
#ifdef MFAS_TEST_HOOKS
             1
#else
             0
#endif
    );
    snprintf(buf, (size_t)cap, "%s", tmp);
    return MFAS_OK;
}

// ================================================================================================
// Host side: C ABI
// ================================================================================================
struct mfas_population {
    mfas_hyper hp;
    Tuning tune;                    // the environment switches as they stood when the population was created
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
    SegDesc* d_mdescs = nullptr;         // the same units as the sweep streams them (multi-chunk units merged), candidate-major
    std::vector<int> mdesc_start;        // K+1
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
    double best_threshold = 0.0;    // snapshot_best: a dev metric must exceed this to count (init_f1, mmimdb.py:18; 0 for NTU)
    // persistent step loop (persist.hip.h): one launch per epoch, per-candidate dependencies
    bool persist = false;
    int n_cus = 0;
    uint32_t* d_red_cnt = nullptr;  // reduce-in-sweep arrival counters [K][4] (small populations, general chain)
    int lp_group = 1;               // chunks per sweep unit of this layout (multi-chunk units: opt-in)
    char* d_gather = nullptr;       // gathered rows [K][2 parities][taps][Bp][width] (two-group schedule, per-candidate orders; sweep.hip.h)
    size_t gather_cap = 0;
    bool red_in_sweep = false;
    bool res_wide = false;          // resident units of more than 512 columns (16-bit staging): f32 tables cannot be trained
    bool same_group = false;        // one launch per step: chain blocks + sweep blocks of the same candidates, per-cell dy flags (k_step_same)
    uint32_t* d_cellflag = nullptr; // [K][CELLFLAG_STRIDE]
    int chain_split = 0;            // CUs per candidate chain in the same-group launch (0 / 1: chain_body on one CU; 4: chain_split<4>)
    float* d_xch = nullptr;         // chain_split's exchange area [K][XCH_CAND_FLOATS]
    size_t lds_split = 0;           // dynamic LDS of the k_step_same<1, *, 4> launches
    bool res_chain = false;         // resident lean chain: owns OUT / HEAD + vector block on chip; persistent units = feature units only
    int res_nu = 1;                 // resident units per workgroup (2: a workgroup serves units of two candidates)
    int nres_wg = 0;                // resident workgroups = ceil(nres / res_nu)
    int res_buf_words = 0;          // LDS words of one staged batch of a resident unit
    int nres = 0;                   // resident feature units (one workgroup each, W/m/v in registers): the first nres persistent units
    SegDesc* d_pdescs = nullptr;    // persistent schedule's unit list: [resident feature units | streamed units]
    int n_pdescs = 0;
    size_t lds_president = 0;       // resident form (k_president)
    int chunk_cols_req = 0;         // chunk_cols the caller asked for at creation (the fallback layout is built with the same request)
    int fell_back = 0;              // the resident schedule was given up for launch-per-phase inside a train() call (roll call never complete)
    uint32_t* d_sync = nullptr;     // [K] flags | [K] counters | abort word (zeroed before every launch)
    int32_t* d_need = nullptr;      // [K] sweep units per candidate
    int32_t* d_role = nullptr;      // [K + nres_wg] role of every workgroup of the resident launch (XCD-aware placement)
    float* d_scal = nullptr;        // device copy of the step scalars
    size_t scal_cap = 0;
    unsigned long long* d_trace = nullptr;
};

static inline int ceil16(int x) { return (x + 15) & ~15; }

// roctx ranges around a train() call and each of its epochs (rocprofv3 --marker-trace shows them next to the kernels).  The
// marker library is looked up at run time: no link-time dependency, plain no-ops where it is absent (or MFAS_NO_ROCTX is set).
namespace {
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        if (getenv("MFAS_NO_ROCTX")) return;
        void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_LAZY | RTLD_LOCAL);
        if (!h) h = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_LAZY | RTLD_LOCAL);
        if (!h) h = dlopen("libroctx64.so", RTLD_LAZY | RTLD_LOCAL);
        if (!h) return;
        push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (!push || !pop) { push = nullptr; pop = nullptr; }
    }
};
Roctx& roctx() { static Roctx r; return r; }
struct RangeGuard {      // pops on every return path
    bool on;
    explicit RangeGuard(const std::string& name) : on(roctx().push != nullptr) { if (on) roctx().push(name.c_str()); }
    ~RangeGuard() { if (on) roctx().pop(); }
};
}
extern "C" int mfas_range_push(const char* name) { if (name && roctx().push) roctx().push(name); return MFAS_OK; }
extern "C" int mfas_range_pop(void) { if (roctx().pop) roctx().pop(); return MFAS_OK; }

extern "C" const char* mfas_last_error(void) { return g_err.c_str(); }
extern "C" int mfas_version(void) { return 200; }
#ifndef MFAS_SRC_DIGEST
#define MFAS_SRC_DIGEST "unknown"
#endif
// sha256 (first 16 hex digits) of the sources this library was built from, baked in by __graft_entry__.build()
extern "C" const char* mfas_source_digest(void) { return "mfas-src-digest:" MFAS_SRC_DIGEST; }

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

// ------------------------------------------------------------------------------------------------
// Layout plan: which column chunk the feature segments are cut into and whether the population takes the RESIDENT persistent
// schedule (k_president: every chain and every feature unit resident, W/m/v in registers) — a pure function of the
// geometry, the configurations and the CU count; no allocation.  create_impl() lays the population out from it and
// mfas_population_plan() answers it on its own (the host's capacity planning, ntu_searchable._plan_rounds).
// ------------------------------------------------------------------------------------------------
struct LayoutPlan {
    bool want_persist = false;
    int target = 0, nu = 1;          // feature-column chunk (0: the launch-per-phase heuristics of create_impl decide), units per resident workgroup
    bool plan_res = false;           // the chunk was chosen for the resident schedule
    int nfeat = 0, max_fcc = 0;      // feature units and the widest of them at that chunk
    bool res_ok = false, res_wide = false, lean_ok = false;
    int nres_wg = 0;
    bool resident = false;           // the resident persistent schedule runs (before the byte-size limits, which no resident population reaches)
    int group = 1;                   // consecutive column chunks one sweep workgroup streams (SegDesc::nsub): ONE forward partial slab per group
    int group_skip = 0;              // (experiment, MFAS_SUBCHUNK_SKIP=n) every n-th candidate keeps one-chunk units: a fine-grained tail for the work list
};

static size_t plan_res_lds(const mfas_hyper* hp, const Geo& g, int cc, int nu) {
    // LDS of a resident workgroup: nu units x 2 staged batches (raw 16-bit rows when the caller promised 16-bit taps, f32 rows
    // otherwise) + the cross-wave reduction slabs + the loop's own words
    const size_t batch = hp->tap_bits == 16 ? (size_t)g.Bp * (cc + 8) * 2 : (size_t)g.Bp * (cc + 4) * 4;
    return (size_t)nu * 2 * batch + (size_t)STEP_NW * g.MB * 256 * 4 + 4 * PERSIST_LDS_WORDS + 64;
}

static void plan_layout(const mfas_hyper* hp, const Geo& g, const int32_t* confs, const int32_t* n_cells, int K, int chunk_cols,
                        int n_cus, bool allow_persist, const Tuning& tu, LayoutPlan& lp) {
    // A workgroup should stream >= ~64 tiles (amortises staging / reduction and keeps the number of partial-sum chunks the
    // chain has to reduce small), the launch should still have a few hundred workgroups, and x_t / x_{t+1} for the chunk must
    // fit the LDS budget.
    // Persistent step loop (persist.hip.h): with one row block (R <= 16) the feature units become RESIDENT (one workgroup per
    // unit, or two units per workgroup; W/m/v in registers): the column chunk is then the smallest of 128 / 256 / 512 / 1024
    // columns with which every chain and every unit workgroup gets a CU of its own.
    // Default (measured, profiles/r02_popsweep_*.log, r03_popsweep.log): ON where the resident form fits (x1.6-2.1 over the
    // launch-per-phase schedule at 4..28 candidates per GPU).  Nothing else is persistent: the streaming form of round 2 (larger R,
    // or units that do not fit; x0.8-0.9 of launch-per-phase) was removed in round 3.  MFAS_PERSIST=0 turns the schedule off.
    lp.want_persist = allow_persist && tu.persist != 0;
    // (lean-chain feasibility, same formula as the LDS budget in create_impl: resident units exist only together with the resident
    // lean chain — k_president; a population without both runs launch-per-phase)
    const size_t lean_bytes = ((size_t)2 * MFAS_MAX_CELLS * g.Bp * 20 + (size_t)g.Bp * (g.Cp + 4) + MFAS_MAX_CELLS * 16 + 3 * g.Bp + 16
                               + (size_t)(g.alphas ? 2 : 1) * MFAS_MAX_CELLS * g.MB * 256 + (size_t)3 * (MFAS_MAX_CELLS * g.vec_cell_stride + g.Cp)
                               + LEAN_SCR + 8 + LeanLds<1>::stage_floats() + (size_t)g.Bp * 64) * 4;
    lp.lean_ok = g.nrb == 1 && g.ncb <= 4 && g.MB <= 2 && lean_bytes <= 78 * 1024 && !tu.no_lean_chain;
    lp.plan_res = lp.want_persist && g.nrb == 1 && g.MB <= 2 && !tu.persist_no_resident && !tu.persist_no_res_chain && lp.lean_ok;
    auto feat_units = [&](int cc_target, int* max_cc) {
        int64_t n = 0;
        int mx = 0;
        for (int k = 0; k < K; ++k)
            for (int i = 0; i < n_cells[k] && i < MFAS_MAX_CELLS; ++i) {
                const int sw = ceil16(hp->s_sizes[confs[(k * 4 + i) * 3] & 7]), vw = ceil16(hp->v_sizes[confs[(k * 4 + i) * 3 + 1] & 7]);
                const int cs = pick_chunk(sw, cc_target), cv = pick_chunk(vw, cc_target);
                n += sw / cs + vw / cv;
                mx = std::max(mx, std::max(cs, cv));
            }
        if (max_cc) *max_cc = mx;
        return n;
    };
    auto res_fits = [&](int cc, int nu, int64_t units) {
        return (cc <= 128 * PERSIST_NTR || (hp->tap_bits == 16 && nu == 1 && cc <= 128 * PERSIST_NTR16)) &&
               plan_res_lds(hp, g, cc, nu) <= 160 * 1024 && K + (units + nu - 1) / nu <= n_cus;
    };
    lp.target = chunk_cols;
    lp.nu = 1;
    if (lp.plan_res && lp.target <= 0) {
        // smallest units first (fewest tiles per wave on the critical path); two units per workgroup before 1024-column units
        // (measured: 16 candidates, 1024-column units: 34.8 us per step)
        // (two 256-column units per workgroup before one 512-column unit: 9..15 candidates 15.7-16.2 vs 18.0-18.7 us per step)
        // (round 5: 128-column units are out — twice the partial slabs through the chain's one CU for half the tiles per wave:
        //  3 / 4 candidates 23.9 / 23.7 us per step against 18.8 / 18.8 with 256-column units, profiles/r05_popsweep_units.log)
        const int opts[5][2] = {{256, 1}, {256, 2}, {512, 1}, {512, 2}, {1024, 1}};
        int pick = -1;
        for (int o = 0; o < 5 && pick < 0; ++o)
            if (res_fits(opts[o][0], opts[o][1], feat_units(opts[o][0], nullptr))) pick = o;
        if (pick >= 0) { lp.target = opts[pick][0]; lp.nu = opts[pick][1]; }
        else lp.plan_res = false;
    } else if (lp.plan_res) {
        const int64_t units = feat_units(lp.target, nullptr);
        if (res_fits(lp.target, 1, units)) lp.nu = 1;
        else if (res_fits(lp.target, 2, units)) lp.nu = 2;
        else lp.plan_res = false;
    }
    if (lp.target <= 0) {
        double tot_cols = 0;
        for (int k = 0; k < K; ++k)
            for (int i = 0; i < n_cells[k] && i < MFAS_MAX_CELLS; ++i)
                tot_cols += ceil16(hp->s_sizes[confs[(k * 4 + i) * 3] & 7]) + ceil16(hp->v_sizes[confs[(k * 4 + i) * 3 + 1] & 7]);
        int lds_max = 64;                                   // largest power of two with Bp*(2cc+20)*4 <= 72 KiB
        while ((size_t)g.Bp * (8 * lds_max + 20) * 4 <= 72 * 1024 && lds_max < 1024) lds_max <<= 1;   // test the doubled size
        int target = 64;
        while (target * g.nrb < 64 * 16 && target < lds_max) target <<= 1;      // >= 64 tiles per workgroup
        while (target > 64 && tot_cols / target < 320.0) target >>= 1;           // ... but keep >= ~320 workgroups
        // R >= 128, measured on MI355X: 64-column chunks (finer, better-balanced workgroups) win once
        // the chain is hidden under the other group's sweep (K >= 20); below that fewer partial chunks matter more
        // (round 2: with reduce-in-sweep the number of partial slabs no longer loads the chain; 256-column chunks stay best up
        // to ~28 candidates, 64 beyond — profiles/r02_popsweep_r128.log)
        if (g.nrb >= 8) target = std::min(target, K >= 28 ? 64 : 256);
        lp.target = target;
    }
    lp.target = std::max(16, (lp.target / 16) * 16);
    // Multi-chunk units (round 4): where the planner streams 64-column chunks (R >= 128, >= 28 candidates: the finest, best-balanced
    // walk of W / m / v) a workgroup takes `group` consecutive chunks and keeps the forward partial sums in registers across them —
    // the memory walk stays that of 64-column chunks, the partial slabs (8 KB written by the unit and read back by the chain, per
    // chunk: 8 % of the algorithmic bytes of a conf-4 step at R = 128) and the dy staging shrink by the group factor.  A caller who
    // fixes chunk_cols gets exactly that decomposition (group 1).
    lp.group = 1;
    // (exactly 8 row blocks: wave = row block, one accumulator per wave; K >= 28: neither the same-group launch nor reduce-in-sweep)
    // (measured, profiles/r04_subchunks.log: uniform groups of 2 / 4 / 8 / 16 chunks are SLOWER — 314 / 320 / 327 / 327 us per launch
    //  against 299 on the same box — because 4x larger units leave ~4 units per workgroup slot and the launch's tail grows faster
    //  than the slab traffic shrinks: the default stays one chunk per unit, MFAS_SUBCHUNKS / MFAS_SUBCHUNK_SKIP select the other forms)
    if (tu.subchunks > 0 && g.nrb == STEP_NW && K >= 28 && !lp.plan_res) lp.group = std::max(1, std::min(64, tu.subchunks));
    lp.group_skip = tu.subchunk_skip;
    {
        int mx = 0;
        lp.nfeat = (int)feat_units(lp.target, &mx);
        lp.max_fcc = mx;
    }
    lp.res_ok = lp.plan_res && res_fits(lp.max_fcc, lp.nu, lp.nfeat);
    lp.res_wide = lp.res_ok && lp.max_fcc > 128 * PERSIST_NTR;      // 16-bit staging only
    lp.nres_wg = lp.res_ok ? (lp.nfeat + lp.nu - 1) / lp.nu : 0;
    lp.resident = lp.want_persist && lp.res_ok && K <= n_cus / 4 && g.MB != 4 && K + lp.nres_wg <= n_cus;
}

// Everything mfas_population_create refuses about (hp, confs, n_cells, K): shared with mfas_population_plan, so that the query never
// reports a layout for inputs create() would reject.
static int validate_inputs(const mfas_hyper* hp, const int32_t* confs, const int32_t* n_cells, int32_t K) {
    if (!hp || !confs || !n_cells || K <= 0) return fail(MFAS_EINVAL, "null argument or K <= 0");
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
    for (int k = 0; k < K; ++k) {
        const int L = n_cells[k];
        if (L < 1 || L > MFAS_MAX_CELLS) return fail(MFAS_EINVAL, "n_cells must be in [1,4]");
        for (int i = 0; i < L; ++i) {
            const int32_t* c = confs + (k * 4 + i) * 3;
            if (c[0] < 0 || c[0] >= MFAS_MAX_TAPS || c[1] < 0 || c[1] >= MFAS_MAX_TAPS || c[2] < 0 || c[2] > 2 ||
                hp->s_sizes[c[0]] < 1 || hp->v_sizes[c[1]] < 1)
                return fail(MFAS_EINVAL, "configuration entry out of range (tap index / unused tap slot / non-linearity)");
        }
    }
    return MFAS_OK;
}

// The sweep's work list: consecutive chunk descriptors of one feature segment that share a partial slab (part_idx) become ONE
// unit that streams them one after the other (SegDesc::nsub); k_pack / the resident schedule keep the per-chunk descriptors.
static std::vector<SegDesc> merge_units(const std::vector<SegDesc>& in) {
    std::vector<SegDesc> out;
    for (const SegDesc& d : in) {
        if (!out.empty()) {
            SegDesc& b = out.back();
            if (d.kind <= KIND_V && b.kind == d.kind && b.cand == d.cand && b.cell == d.cell && b.part_idx == d.part_idx && b.cc == d.cc &&
                b.k0 + b.nsub * b.cc == d.k0 && b.w_off + (int64_t)b.nsub * b.rows_p * b.cc == d.w_off) {
                b.nsub++;
                continue;
            }
        }
        out.push_back(d);
        if (out.back().nsub < 1) out.back().nsub = 1;
    }
    return out;
}

#define MFAS_RETRY_NO_PERSIST 12345   // internal: the layout was planned for the resident persistent schedule, which then did not fit

static int create_impl(const mfas_hyper* hp, const int32_t* confs, const int32_t* n_cells,
                       const uint32_t* drop_seeds, int32_t K, int32_t device, void* hip_stream,
                       int32_t chunk_cols, mfas_population** out, const bool allow_persist, const Tuning* inherit = nullptr) {
    if (!out) return fail(MFAS_EINVAL, "null argument or K <= 0");
    if (int vrc = validate_inputs(hp, confs, n_cells, K)) return vrc;
    mfas_population* p = new (std::nothrow) mfas_population();
    if (!p) return fail(MFAS_ENOMEM, "host alloc");
    p->hp = *hp;
    p->tune = inherit ? *inherit : tuning_from_env();      // (persist_fallback rebuilds a population under the switches it was created with)
    const Tuning& tu = p->tune;
    p->K = K;
    p->device = device;
    p->stream = reinterpret_cast<hipStream_t>(hip_stream);
    p->chunk_cols_req = chunk_cols;
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

    // ---- column chunk per workgroup and the schedule (plan_layout: the same pure decision mfas_population_plan answers)
    {
        int ncu = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || ncu <= 0) ncu = 256;
        p->n_cus = ncu;
    }
    LayoutPlan lp;
    plan_layout(hp, g, confs, n_cells, K, chunk_cols, p->n_cus, allow_persist, tu, lp);
    const bool want_persist = lp.want_persist;
    int target = lp.target;
    const int plan_nu = lp.nu;
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
                const int grp = (j < 2 && (lp.group_skip <= 0 || (k % lp.group_skip) != lp.group_skip - 1)) ? lp.group : 1;   // chunks per sweep unit = per partial slab
                const int nun = (nch + grp - 1) / grp;
                c.seg_off[i][j] = plane_off;
                c.seg_cc[i][j] = cc;
                c.seg_cols[i][j] = cols_p;
                if (j == 0) c.nch_s[i] = nun;
                if (j == 1) c.nch_v[i] = nun;
                if (j == 2) { c.outT_off[i] = wt_off; }
                for (int ch = 0; ch < nch; ++ch) {
                    SegDesc d;
                    memset(&d, 0, sizeof(d));
                    d.cand = k; d.kind = j; d.cell = i; d.tap = j < 2 ? c.conf[i][j] : 0;
                    d.k0 = ch * cc; d.cc = cc; d.rows_p = g.Rp; d.width = j < 2 ? widths[j] : g.Rp;
                    d.w_off = plane_off + (int64_t)ch * g.Rp * cc;
                    d.wt_off = j == 2 ? wt_off : -1;
                    d.part_idx = j < 2 ? (j == 0 ? ch / grp : c.nch_s[i] + ch / grp) : 0;
                    d.nsub = 1;
                    d.rows = hp->R; d.cols = true_w[j];
                    d.src_off = c.f_W[i]; d.src_ld = Kin; d.src_col0 = col0[j];
                    d.init_seed = 2 * i; d.init_bound = bound;
                    d.rb0 = 0; d.seg_nrb = g.nrb;
                    p->descs.push_back(d);
                }
                if (j < 2) pslot += nun;
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
            d.rb0 = 0; d.seg_nrb = g.ncb; d.nsub = 1;
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
        // resident feature units (persistent schedule) do not go through sweep_body: their LDS need is separate
        int nfeat = 0, max_fcc = 0;
        for (const SegDesc& d : p->descs)
            if (d.kind <= KIND_V) { ++nfeat; max_fcc = std::max(max_fcc, d.cc); }
        if (nfeat != lp.nfeat || max_fcc != lp.max_fcc) { delete p; return fail(MFAS_EINVAL, "internal: layout plan and descriptors disagree"); }
        const size_t lds_res = plan_res_lds(hp, g, max_fcc, plan_nu);
        const bool res_ok = lp.res_ok;
        p->res_wide = lp.res_wide;
        p->nres = res_ok ? nfeat : 0;
        p->res_nu = plan_nu;
        p->nres_wg = lp.nres_wg;
        p->res_buf_words = (int)((hp->tap_bits == 16 ? (size_t)g.Bp * (max_fcc + 8) * 2 : (size_t)g.Bp * (max_fcc + 4) * 4) / 4);
        size_t ls = 0;
        for (const SegDesc& d : p->descs) {
            if (res_ok && d.kind <= KIND_V) continue;
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
                             + (g.alphas ? 2 : 1) * plane + vec / 4 + LEAN_SCR + 8 + LeanLds<1>::stage_floats() + (size_t)g.Bp * 64) * 4;
        // (the chain form must not depend on the sweep's chunk size: since round 2 the lean chain sums bias gradients and BN
        // statistics in its own — element-parallel — order, so lean and general chains agree to rounding, not bit for bit)
        p->lean_chain = g.nrb == 1 && g.ncb <= 4 && g.MB <= 2 && lean <= 78 * 1024 && !tu.no_lean_chain;
        if (p->lean_chain) { p->lds_chain = lean; p->lds_step = std::max(p->lds_step, lean); }
        p->res_chain = res_ok && p->lean_chain;
        if (res_ok && !p->lean_chain) {   // (cannot happen while lean_ok_early mirrors the formula above)
            delete p;
            return MFAS_RETRY_NO_PERSIST;
        }
        const size_t lds_rchain = p->res_chain ? p->lds_chain + 16 + 4 * (size_t)(LeanLds<1>::own_floats() - LeanLds<1>::stage_floats()) : 0;
        p->lds_president = ((std::max(lds_rchain, lds_res) + 15) & ~(size_t)15) + 4 * PERSIST_LDS_WORDS;
    }
    p->nrbw = (g.nrb + 3) / 4;
    if (p->nrbw == 3) p->nrbw = 4;
    for (p->mbe = 4; p->mbe >= 1; p->mbe >>= 1) {
        const int ME = p->mbe * 16;
        p->lds_eval = ((size_t)ME * std::max(EVAL_CE + 8, g.Cp + 4) + (size_t)ME * (g.Rp + 8)) * 4;   // strides: eval.hip.h
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
    CREATE_CHK(hipMalloc(&p->d_status, sizeof(int32_t) * (K + 256)));   // + debug timestamp slots (MFAS_CHAIN_TIMING builds)
    CREATE_CHK(hipMemset(p->d_status, 0, sizeof(int32_t) * (K + 256)));
    CREATE_CHK(hipMalloc(&p->d_seeds, sizeof(uint32_t) * K));
    CREATE_CHK(hipMalloc(&p->d_corr, sizeof(long long)));
    {
        std::vector<float> ones(g.Cp, 1.0f);
        CREATE_CHK(hipMalloc(&p->d_posw, sizeof(float) * g.Cp));
        CREATE_CHK(hipMemcpy(p->d_posw, ones.data(), sizeof(float) * g.Cp, hipMemcpyHostToDevice));
    }
    CREATE_CHK(hipMemcpy(p->d_cands, p->cands.data(), sizeof(CandDev) * K, hipMemcpyHostToDevice));
    CREATE_CHK(hipMemcpy(p->d_descs, p->descs.data(), sizeof(SegDesc) * p->descs.size(), hipMemcpyHostToDevice));
    {
        std::vector<SegDesc> merged;
        p->mdesc_start.assign(K + 1, 0);
        for (int k = 0; k < K; ++k) {
            p->mdesc_start[k] = (int)merged.size();
            std::vector<SegDesc> one(p->descs.begin() + p->desc_start[k], p->descs.begin() + p->desc_start[k + 1]);
            if (lp.group > 1) one = merge_units(one);
            merged.insert(merged.end(), one.begin(), one.end());
        }
        p->mdesc_start[K] = (int)merged.size();
        CREATE_CHK(hipMalloc(&p->d_mdescs, sizeof(SegDesc) * merged.size()));
        CREATE_CHK(hipMemcpy(p->d_mdescs, merged.data(), sizeof(SegDesc) * merged.size(), hipMemcpyHostToDevice));
    }
    {   // candidate groups: two halves balanced by work (descriptor columns), contiguous ranges
        // Two groups (the chain of one runs under the sweep of the other).  Measured on MI355X (cand/s, unfused vs fused):
        // general chain, R=128: 16 candidates 104 vs 96, 20: 103 vs 110, 32: 119 vs 142 -> fused from 20;
        // lean chain, R=16 (18 us, cheap enough to run as its own launch over all CUs): 32: 348 vs 307, 40: 361 vs 364,
        // 50: 430 vs 475, 100: 582 vs 677, 200: 566 vs 600, 256: 630 vs 619, 512: 685 vs 641 -> fused only for 40 <= K < 224.
        // round 2, general chain with reduce-in-sweep (chain 48 -> 38 us at R=128): fused from 8 candidates
        // (R=128 cand/s unfused+reduce vs fused+reduce: 12 candidates 18.1 vs 20.2, 16: 21.0 vs 23.6, 24: 22.9 (old default) vs 26.6)
        int ngroups = p->lean_chain ? ((K >= 40 && K < 224) ? 2 : 1) : (K >= 8 ? 2 : 1);
        if (tu.groups > 0) ngroups = (tu.groups >= 2 && K >= 2) ? 2 : 1;
        {   // persistent step loop: small populations (one workgroup per CU must hold every chain + a useful number of sweep workgroups)
            const bool want = want_persist;
            const bool fits = K <= p->n_cus / 4 && g.MB != 4 && K + p->nres_wg <= p->n_cus;
            p->persist = want && fits && p->res_chain;
            if (p->persist) ngroups = 1;
            else if (p->nres > 0) {   // units and LDS budgets were laid out for resident units: start over without them
                mfas_population_destroy(p);
                return MFAS_RETRY_NO_PERSIST;
            }
        }
        // same-group fused launch (k_step_same): general chain, one group, R >= 128 (no tap-major units), launch-per-phase
        {
            const int sgenv = tu.same_group;     // 0: never, 2: whatever the size (A/B runs)
            // measured (MI355X, conf 4, B=16): pays while the population's W/m/v stream is <= ~260 MB per step — R=128: 1 / 3 / 6 / 8 / 12
            // candidates 56 / 65 / 76 / 81 / 91 -> 50 / 54 / 63 / 70 / 88 us per step (16: equal); R=64: 6 / 12 / 16: 52 / 61 / 64 -> 45 / 55 / 61
            // (24: 74 -> 79); R=32: 6 / 12 / 32: 45 / 54 / 68 -> 36 / 41 / 60 (64: 85 -> 90)
            double state_bytes = 0;
            for (const SegDesc& d : p->descs) state_bytes += 24.0 * d.cc * d.rows_p;
            const bool two_forced = tu.groups >= 2;      // (tests: the two-group fused schedule)
            p->same_group = !p->persist && !p->lean_chain && g.MB <= 2 && (state_bytes <= 260e6 || sgenv == 2) && sgenv != 0 && !two_forced &&
                            lp.group == 1;      // (multi-chunk units exist in k_step's sweep only)
        }
        if (p->same_group) ngroups = 1;
        // the chain of one candidate over 4 CUs (chain.hip.h, chain_split): eight row blocks, one batch tile, <= 4 class blocks, no alphas
        // — in the same-group launch, and in the two-group launches while the chain bounds them (< 28 candidates: sweep(8 candidates) = 36 us
        // against a 47 us chain; beyond, the chain hides under the other group's sweep and 4 x 64 chain workgroups would only take CUs from it)
        p->chain_split = ((p->same_group || (ngroups == 2 && K < 28)) && !p->lean_chain && !p->persist && g.MB == 1 && g.nrb == 8 && g.ncb <= 4 && !g.alphas &&
                          tu.chain_split != 0 && tu.chain_split != 1) ? 4 : 0;
        p->lp_group = lp.group;
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
            if (lp.group > 1) all = merge_units(all);
            // small R (1, 2 or 4 row blocks): feature segments are regrouped tap-major (sweep_tap_body)
            std::vector<SegDesc> sorted;
            std::vector<TapDesc> taps;
            // (tap-major workgroups stage a batch's rows ONCE for several candidates: not with per-candidate sample orders)
            const bool tap_major = (g.nrb == 1 || g.nrb == 2 || g.nrb == 4) && !tu.no_tap_major && !p->persist && !p->same_group &&
                                   !hp->order_per_candidate;
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
                if (taps.size() < 192 && !tu.force_tap_major) {   // too few workgroups to fill 256 CUs: per-segment path
                    taps.clear();
                    sorted = all;
                }
            } else {
                sorted = all;
            }
            std::stable_sort(sorted.begin(), sorted.end(), [](const SegDesc& x, const SegDesc& y) {
                return (int64_t)x.cc * x.rows_p * std::max(1, x.nsub) > (int64_t)y.cc * y.rows_p * std::max(1, y.nsub); });
            if (p->same_group) {
                // OUT / HEAD units one ROW BLOCK each (round 6): as ONE workgroup per 128 x 128 segment every wave walked its row block's eight
                // tiles in four dependent load -> Adam -> store rounds of ~2.5 us behind the dy it waits for — OUT_1, released by the LAST dy of
                // the step, ended 12.7 us after the chain where the cell-0 feature units end after 6.2 (profiles/r06_chain_split_r128.log).
                // Row-split units (SegDesc::rb0 / seg_nrb, one tile per wave: the k-split walk) update the same tiles with the same arithmetic.
                std::vector<SegDesc> fine;
                for (const SegDesc& d : sorted) {
                    const int nrb_d = d.rows_p / 16;
                    if (d.kind <= KIND_V || nrb_d <= 1) { fine.push_back(d); continue; }
                    for (int r0 = 0; r0 < nrb_d; ++r0) {
                        SegDesc u = d;
                        u.rb0 = r0; u.rows_p = 16; u.seg_nrb = nrb_d;
                        u.w_off = d.w_off + (int64_t)r0 * (d.cc / 16) * 256;
                        fine.push_back(u);
                    }
                }
                sorted.swap(fine);
            }
            if (p->same_group)   // the order the chain releases the units in
                std::stable_sort(sorted.begin(), sorted.end(), [](const SegDesc& x, const SegDesc& y) {
                    // (the slot each unit waits for: feature units of cell i -> i, OUT_i -> i - 1, HEAD -> the last cell; highest slot first)
                    auto slot = [](const SegDesc& d) { return d.kind == KIND_HEAD ? MFAS_MAX_CELLS : (d.kind == KIND_OUT ? d.cell - 1 : d.cell); };
                    return slot(x) > slot(y);
                });
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
    // reduce-in-sweep: one group (the chain is on the critical path), general chain, per-segment units only
    // (beyond ~28 candidates the co-scheduled chain is hidden anyway and the extra write-through traffic costs: 29.0 vs 26.7 cand/s at 32)
    // (not with chain_split: the reducing unit's drain + arrival + summing pass behind the LAST dy of the step ends the launch 3.5 us later,
    //  while the chain's four parts sum their own row blocks of the slabs at entry, every load in flight at once; measured, K = 1: 41.3 with
    //  the reduction in the sweep, 38.4 without, 43.0 with a hybrid — cells >= 1 in the sweep, cell 0 in the chain — profiles/r06_chain_split_r128.log)
    p->red_in_sweep = (K < 28 || tu.force_red_in_sweep) && !p->lean_chain && !p->persist && !tu.no_red_in_sweep && !(p->chain_split && p->same_group);
    for (const auto& gr : p->groups) if (gr.ntap != 0) p->red_in_sweep = false;     // (tap-major workgroups serve several candidates)
    if (p->red_in_sweep) {
        CREATE_CHK(hipMalloc(&p->d_red_cnt, sizeof(uint32_t) * K * MFAS_MAX_CELLS));
        CREATE_CHK(hipMemset(p->d_red_cnt, 0, sizeof(uint32_t) * K * MFAS_MAX_CELLS));
    }
    if (p->same_group) {
        CREATE_CHK(hipMalloc(&p->d_cellflag, sizeof(uint32_t) * K * CELLFLAG_STRIDE));
        CREATE_CHK(set_lds((k_step_same<1, false>), p->lds_step)); CREATE_CHK(set_lds((k_step_same<1, true>), p->lds_step));
        CREATE_CHK(set_lds((k_step_same<2, false>), p->lds_step)); CREATE_CHK(set_lds((k_step_same<2, true>), p->lds_step));
    }
    if (p->chain_split) {
        p->lds_split = std::max(p->lds_step, chain_split_lds_floats<4>(g.Rp, g.Cp) * 4);
        CREATE_CHK(set_lds((k_step_same<1, false, 4>), p->lds_split)); CREATE_CHK(set_lds((k_step_same<1, true, 4>), p->lds_split));
        CREATE_CHK(set_lds((k_step<1, false, 4, false, 4>), p->lds_split)); CREATE_CHK(set_lds((k_step<1, true, 4, false, 4>), p->lds_split));
        CREATE_CHK(hipMalloc(&p->d_xch, sizeof(float) * (size_t)K * XCH_CAND_FLOATS));
        if ((size_t)K * XCH_CAND_FLOATS >= (1ull << 30)) { mfas_population_destroy(p); return fail(MFAS_EINVAL, "internal: exchange area beyond the 32-bit buffer offsets"); }
    }
    CREATE_CHK(hipMemsetAsync(p->plane, 0, sizeof(float) * 3 * (size_t)p->plane_stride, p->stream));
    CREATE_CHK(hipMemsetAsync(p->wt, 0, sizeof(float) * (size_t)std::max<int64_t>(p->wt_size, 64), p->stream));
    CREATE_CHK(hipMemsetAsync(p->stepbuf, 0, sizeof(float) * (size_t)p->step_total, p->stream));
    // measured crossover (MI355X, B=20): R=16 between 165 and 330 MB of group state per launch, R=128 between 300 and 600 MB
    // (the spilling chain of the occupancy build takes ~40 / ~125 us there)
    p->occ_bytes = g.nrb >= 8 ? 450e6 : 250e6;
    if (tu.occ_bytes >= 0) p->occ_bytes = tu.occ_bytes;
#define SET_STEP(M, W, F) CREATE_CHK(set_lds((k_step<M, false, W, F>), p->lds_step)); CREATE_CHK(set_lds((k_step<M, true, W, F>), p->lds_step))
    SET_STEP(1, 4, false); SET_STEP(2, 2, false); SET_STEP(2, 4, false); SET_STEP(4, 2, false);
    SET_STEP(1, 4, true); SET_STEP(2, 2, true); SET_STEP(2, 4, true);
#undef SET_STEP
    CREATE_CHK(set_lds((k_chain<1, false>), p->lds_chain));
    CREATE_CHK(set_lds((k_chain<2, false>), p->lds_chain));
    CREATE_CHK(set_lds((k_chain<4, false>), p->lds_chain));
    CREATE_CHK(set_lds((k_chain<1, true>), p->lds_chain));
    CREATE_CHK(set_lds((k_chain<2, true>), p->lds_chain));
    if (p->persist) {
        std::vector<int> res_cand;      // candidate of every resident unit, in unit order
        {   // unit list of the resident schedule: the feature units (the resident lean chain updates OUT / HEAD itself)
            std::vector<SegDesc> res;
            for (const SegDesc& d : p->descs)
                if (d.kind <= KIND_V) { res.push_back(d); res_cand.push_back(d.cand); }
            p->n_pdescs = (int)res.size();
            CREATE_CHK(hipMalloc(&p->d_pdescs, sizeof(SegDesc) * res.size()));
            CREATE_CHK(hipMemcpy(p->d_pdescs, res.data(), sizeof(SegDesc) * res.size(), hipMemcpyHostToDevice));
        }
        std::vector<int32_t> need(K, 0);
        for (const SegDesc& d : p->descs)
            if (d.kind <= KIND_V) need[d.cand]++;
        CREATE_CHK(hipMalloc(&p->d_need, sizeof(int32_t) * K));
        CREATE_CHK(hipMemcpy(p->d_need, need.data(), sizeof(int32_t) * K, hipMemcpyHostToDevice));
        if (!tu.no_xcd_placement) {
            // XCD-aware placement (round 5): consecutive workgroups of a launch are dealt round-robin to the 8 XCDs (block b -> XCD b % 8,
            // MI355X_MICROARCH.md), each with its own L2.  A candidate's chain and the workgroups that hold its units exchange 60 KB of
            // slabs and 8 KB of dy per step: deal the roles so that they share an XCD wherever its 32 slots allow (greedy, candidate by
            // candidate; two-unit workgroups are grouped by their FIRST unit's candidate, and the chain of a candidate that only ever
            // comes second goes where most of its units are).  Placement only: the exchanges do not depend on it.
            // (NX: 8 XCDs on MI355X in SPX mode, block b -> XCD b % 8; MFAS_XCDS=n for another partition mode.  A wrong NX costs
            //  only the co-location.  prim / sec below mirror sweep_resident's unit mapping — unit u of workgroup w is unit w + u * nwg,
            //  persist.hip.h `const int ui = wg + u * nwg` — and must change with it.)
            const int nwg = p->nres_wg, G = K + nwg, NX = tu.n_xcd > 0 ? std::min(tu.n_xcd, 64) : 8;
            std::vector<std::vector<int>> slots(NX);
            for (int b = G - 1; b >= 0; --b) slots[b % NX].push_back(b);       // (pop_back hands out the lowest block of an XCD first)
            std::vector<int32_t> role(G, -1);
            std::vector<int> chain_xcd(K, -1);
            std::vector<char> wg_done(nwg, 0);
            auto take = [&](int x, int item) { role[slots[x].back()] = item; slots[x].pop_back(); };
            auto roomiest = [&]() { int bx = 0; for (int x = 1; x < NX; ++x) if (slots[x].size() > slots[bx].size()) bx = x; return bx; };
            std::vector<int> prim(nwg), sec(nwg, -1);
            for (int w = 0; w < nwg; ++w) {
                prim[w] = res_cand[w];
                if (p->res_nu == 2 && w + nwg < (int)res_cand.size()) sec[w] = res_cand[w + nwg];
            }
            for (int c = 0; c < K; ++c) {                   // candidates that come first in some workgroup: chain + those workgroups
                bool any = false;
                for (int w = 0; w < nwg; ++w) any = any || prim[w] == c;
                if (!any) continue;
                const int x = roomiest();
                if (!slots[x].empty()) { take(x, c); chain_xcd[c] = x; }
                for (int w = 0; w < nwg; ++w)
                    if (prim[w] == c && !wg_done[w] && !slots[x].empty()) { take(x, K + w); wg_done[w] = 1; }
            }
            std::vector<int> wg_xcd(nwg, -1);
            for (int b = 0; b < G; ++b) if (role[b] >= K) wg_xcd[role[b] - K] = b % NX;
            for (int c = 0; c < K; ++c) {                   // chains not placed yet: where most of the candidate's units are
                if (chain_xcd[c] >= 0) continue;
                std::vector<int> votes(NX, 0);
                for (int w = 0; w < nwg; ++w) if ((prim[w] == c || sec[w] == c) && wg_xcd[w] >= 0) votes[wg_xcd[w]]++;
                int bx = -1;
                for (int x = 0; x < NX; ++x) if (!slots[x].empty() && (bx < 0 || votes[x] > votes[bx])) bx = x;
                if (bx >= 0) { take(bx, c); chain_xcd[c] = bx; }
            }
            for (int w = 0; w < nwg; ++w)                   // whatever did not fit its XCD
                if (!wg_done[w]) { const int x = roomiest(); take(x, K + w); wg_done[w] = 1; }
            bool ok = true;
            for (int b = 0; b < G; ++b) ok = ok && role[b] >= 0;
            if (ok) {
                CREATE_CHK(hipMalloc(&p->d_role, sizeof(int32_t) * G));
                CREATE_CHK(hipMemcpy(p->d_role, role.data(), sizeof(int32_t) * G, hipMemcpyHostToDevice));
            }
        }
        CREATE_CHK(hipMalloc(&p->d_sync, sizeof(uint32_t) * ((size_t)K * PERSIST_SYNC_STRIDE + 64)));
        if (tu.persist_trace) {
            CREATE_CHK(hipMalloc(&p->d_trace, sizeof(unsigned long long) * 256));
            CREATE_CHK(hipMemset(p->d_trace, 0, sizeof(unsigned long long) * 256));
        }
#define SET_RES(M, P) CREATE_CHK(set_lds((k_president<M, PERSIST_NTR, false, 1, P>), p->lds_president)); CREATE_CHK(set_lds((k_president<M, PERSIST_NTR, false, 2, P>), p->lds_president)); \
                   CREATE_CHK(set_lds((k_president<M, PERSIST_NTR16, true, 1, P>), p->lds_president)); CREATE_CHK(set_lds((k_president<M, PERSIST_NTR, true, 1, P>), p->lds_president)); \
                   CREATE_CHK(set_lds((k_president<M, PERSIST_NTR, true, 2, P>), p->lds_president))
        SET_RES(1, 0); SET_RES(2, 0); SET_RES(1, 1); SET_RES(2, 1); SET_RES(1, 2); SET_RES(2, 2);
#undef SET_RES
    }
    // W/m/v beyond what the 256 MiB Infinity Cache can keep between steps are streamed nontemporally
    p->nontemporal = (double)p->plane_stride * 12.0 > 200.0 * 1024 * 1024;
    if (tu.nt >= 0) p->nontemporal = tu.nt != 0;
    CREATE_CHK(hipStreamSynchronize(p->stream));
    *out = p;
    return MFAS_OK;
}

extern "C" int mfas_population_create(const mfas_hyper* hp, const int32_t* confs, const int32_t* n_cells,
                                      const uint32_t* drop_seeds, int32_t K, int32_t device, void* hip_stream,
                                      int32_t chunk_cols, mfas_population** out) {
    int rc = create_impl(hp, confs, n_cells, drop_seeds, K, device, hip_stream, chunk_cols, out, true);
    if (rc == MFAS_RETRY_NO_PERSIST) rc = create_impl(hp, confs, n_cells, drop_seeds, K, device, hip_stream, chunk_cols, out, false);
    return rc;
}

// The layout / schedule decision of mfas_population_create for these configurations WITHOUT creating anything (no allocation, no
// launch): the host's capacity planning (how many candidates one resident round can hold) asks this instead of building and
// destroying populations.
extern "C" int mfas_population_plan(const mfas_hyper* hp, const int32_t* confs, const int32_t* n_cells, int32_t K, int32_t device,
                                    int32_t chunk_cols, int32_t info[8]) {
    if (!info) return fail(MFAS_EINVAL, "null argument or K <= 0");
    if (int vrc = validate_inputs(hp, confs, n_cells, K)) return vrc;
    Geo g;
    memset(&g, 0, sizeof(g));
    g.R = hp->R; g.C = hp->C; g.Rp = ceil16(hp->R); g.Cp = ceil16(hp->C);
    g.nrb = g.Rp / 16; g.ncb = g.Cp / 16; g.B = hp->B;
    g.MB = (hp->B + 15) / 16; if (g.MB == 3) g.MB = 4;
    g.Bp = g.MB * 16;
    g.alphas = hp->alphas != 0;
    g.vec_cell_stride = 5 * g.Rp + 16;
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || ncu <= 0) ncu = 256;
    LayoutPlan lp;
    const Tuning tu = tuning_from_env();
    plan_layout(hp, g, confs, n_cells, K, chunk_cols, ncu, true, tu, lp);
    if (!lp.resident && lp.plan_res)      // the resident chunking does not stand: what create would fall back to (launch per phase)
        plan_layout(hp, g, confs, n_cells, K, chunk_cols, ncu, false, tu, lp);
    info[0] = lp.resident ? 1 : 0; info[1] = lp.resident ? lp.nfeat : 0; info[2] = lp.resident ? lp.nres_wg : 0; info[3] = lp.nu;
    info[4] = lp.target; info[5] = lp.lean_ok ? 1 : 0; info[6] = ncu; info[7] = K;
    return MFAS_OK;
}

extern "C" void mfas_population_destroy(mfas_population* p) {
    if (!p) return;
    hipSetDevice(p->device);
    hipStreamSynchronize(p->stream);
    for (hipEvent_t e : p->ev) hipEventDestroy(e);
    hipFree(p->plane); hipFree(p->wt); hipFree(p->stepbuf); hipFree(p->best);
    for (auto& gr : p->groups) { hipFree(gr.d_descs); hipFree(gr.d_taps); }
    hipFree(p->d_cands); hipFree(p->d_descs); hipFree(p->d_mdescs); hipFree(p->d_stats); hipFree(p->d_status);
    hipFree(p->d_seeds); hipFree(p->d_corr); hipFree(p->d_posw);
    hipFree(p->d_red_cnt);
    hipFree(p->d_gather);
    hipFree(p->d_cellflag);
    hipFree(p->d_xch);
    hipFree(p->d_sync); hipFree(p->d_need); hipFree(p->d_role); hipFree(p->d_scal); hipFree(p->d_trace); hipFree(p->d_pdescs);
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

// torch.manual_seed(seeds[k]) + the module's construction draws for every candidate, on the device (k_mt_uniform, pack.hip.h).
// bounds: per candidate 2 * (MFAS_MAX_CELLS + 1) floats — per cell {weight bound, bias bound}, then the classifier's — as the host
// computed them (kaiming_uniform_(a = sqrt 5) / 1 / sqrt(fan_in), nn.Linear.reset_parameters); alphas ~ N(alpha_mean, alpha_std)
// drawn LAST like Searchable_Skeleton_Image_Net.__init__ does (ntu_searchable.py:202-204), from the stream's next raw outputs with
// at::normal_distribution<double>'s arithmetic (Box-Muller: r = sqrt(-2 log1p(-u2)), theta = 2 pi u1; the sine sample is cached
// for the next draw) in host double precision / libm, exactly what torch's CPU path evaluates.
extern "C" int mfas_population_init_torch_streams(mfas_population* p, const uint64_t* seeds, const float* bounds, double alpha_mean,
                                                  double alpha_std) {
    if (!p || !seeds || !bounds) return fail(MFAS_EINVAL, "bad argument");
    HIPCHK(hipSetDevice(p->device));
    const int K = p->K, R = p->hp.R, C = p->hp.C, NB = 2 * (MFAS_MAX_CELLS + 1);
    int64_t maxp = 0;
    for (int k = 0; k < K; ++k) maxp = std::max(maxp, p->nparams[k]);
    const int batch = (int)std::max<int64_t>(1, std::min<int64_t>(K, (64LL << 20) / std::max<int64_t>(maxp, 1)));     // <= 256 MB of flat scratch
    float* flat = nullptr;
    MtCand* d_mt = nullptr;
    uint32_t* d_tail = nullptr;
    auto cleanup = [&]() { hipFree(flat); hipFree(d_mt); hipFree(d_tail); };
    hipError_t e = hipMalloc(&flat, sizeof(float) * (size_t)maxp * batch);
    if (e == hipSuccess) e = hipMalloc(&d_mt, sizeof(MtCand) * batch);
    if (e == hipSuccess) e = hipMalloc(&d_tail, sizeof(uint32_t) * MT_TAIL * batch);
    if (e != hipSuccess) { cleanup(); return fail(MFAS_ENOMEM, std::string("init_torch_streams: ") + hipGetErrorString(e)); }
    std::vector<MtCand> mt(batch);
    std::vector<uint32_t> tails((size_t)MT_TAIL * batch);
    std::vector<float> alpha((size_t)MFAS_MAX_CELLS * batch);
    for (int k0 = 0; k0 < K && e == hipSuccess; k0 += batch) {
        const int nb = std::min(batch, K - k0);
        for (int j = 0; j < nb; ++j) {
            const int k = k0 + j;
            const CandDev& c = p->cands[k];
            MtCand& m = mt[j];
            memset(&m, 0, sizeof(m));
            m.seed = (uint32_t)(seeds[k] & 0xffffffffULL);
            m.flat_off = (int64_t)j * maxp;
            int64_t pos = 0;
            auto seg = [&](int64_t dst, int64_t n, float b) {
                m.start[m.nseg] = pos; m.dst[m.nseg] = dst; m.lo[m.nseg] = -b; m.hi[m.nseg] = b;
                pos += n; ++m.nseg;
            };
            for (int i = 0; i < c.L; ++i) {
                seg(c.f_W[i], (int64_t)R * c.K_in[i], bounds[k * NB + 2 * i]);
                seg(c.f_b[i], R, bounds[k * NB + 2 * i + 1]);
            }
            seg(c.f_Wc, (int64_t)C * R, bounds[k * NB + 2 * MFAS_MAX_CELLS]);
            seg(c.f_bc, C, bounds[k * NB + 2 * MFAS_MAX_CELLS + 1]);
            m.start[m.nseg] = pos;
            m.total = pos;
        }
        e = hipMemcpyAsync(d_mt, mt.data(), sizeof(MtCand) * nb, hipMemcpyHostToDevice, p->stream);
        if (e == hipSuccess) e = hipMemsetAsync(flat, 0, sizeof(float) * (size_t)maxp * nb, p->stream);
        if (e != hipSuccess) break;
        hipLaunchKernelGGL(k_mt_uniform, dim3(nb), dim3(256), 0, p->stream, d_mt, flat, d_tail);
        e = hipMemcpyAsync(tails.data(), d_tail, sizeof(uint32_t) * MT_TAIL * nb, hipMemcpyDeviceToHost, p->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
        if (e != hipSuccess) break;
        for (int j = 0; j < nb && e == hipSuccess; ++j) {
            const int k = k0 + j;
            const CandDev& c = p->cands[k];
            // BatchNorm defaults (gamma = 1, running_var = 1) and the alphas, then the usual repacking of a flat vector
            const uint32_t* t = tails.data() + (size_t)j * MT_TAIL;
            int used = 0;
            bool cached = false;
            double cache = 0.0;
            auto u53 = [&]() {      // uniform_real_distribution<double>: random64() = (first << 32) | second, 53 bits
                const uint64_t hi = t[used], lo = t[used + 1];
                used += 2;
                return (double)(((hi << 32) | lo) & ((1ULL << 53) - 1)) * (1.0 / 9007199254740992.0);
            };
            for (int i = 0; i < c.L; ++i) {
                double z;
                if (cached) { z = cache; cached = false; }
                else {
                    const double u1 = u53(), u2 = u53();
                    const double r = ::sqrt(-2.0 * ::log1p(-u2)), theta = 2.0 * 3.14159265358979323846 * u1;
                    cache = r * ::sin(theta);
                    cached = true;
                    z = r * ::cos(theta);
                }
                alpha[(size_t)j * MFAS_MAX_CELLS + i] = (float)(z * alpha_std + alpha_mean);
            }
            float* fk = flat + (int64_t)j * maxp;
            e = hipMemcpyAsync(fk + c.f_alpha, alpha.data() + (size_t)j * MFAS_MAX_CELLS, sizeof(float) * c.L, hipMemcpyHostToDevice, p->stream);
            if (e != hipSuccess) break;
            if (p->hp.bn) {
                for (int i = 0; i < c.L && e == hipSuccess; ++i) {
                    hipLaunchKernelGGL(k_fill, dim3(1), dim3(256), 0, p->stream, fk + c.f_bn[i], 1.0f, (int64_t)R);             // gamma
                    hipLaunchKernelGGL(k_fill, dim3(1), dim3(256), 0, p->stream, fk + c.f_bn[i] + 3 * (int64_t)R, 1.0f, (int64_t)R);   // running_var
                }
            }
            const int rc = mfas_population_set_params(p, k, fk);
            if (rc) { cleanup(); return rc; }
        }
        if (e == hipSuccess) e = hipStreamSynchronize(p->stream);       // the scratch is reused by the next batch
    }
    cleanup();
    if (e != hipSuccess) return fail(MFAS_EHIP, std::string("init_torch_streams: ") + hipGetErrorString(e));
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

template <int MBE, int NRBW, int MSP = 0, bool XB = false, bool B3 = false>
static hipError_t launch_eval_t(mfas_population* p, const EvalArgs& a, int ncand, hipStream_t st) {
    const int ME = MBE * 16;
    // (16-bit row tile: half the width, more workgroups per CU)
    size_t lds = (XB || B3) ? ((size_t)ME * std::max((EVAL_CE + 8) / 2, p->g.Cp + 4) + (size_t)ME * (p->g.Rp + 8)) * 4 : p->lds_eval;
    if (XB && MSP == 1 && NRBW == 1 && MBE == 4) lds += (size_t)2 * (EVAL_CE / 16) * 256 * 4;      // the workgroup's weight tiles, double-buffered (eval.hip.h, WL)
    hipError_t e = set_lds(k_eval<MBE, NRBW, MSP, XB, B3>, lds);
    if (e != hipSuccess) return e;
    const unsigned nblk = (unsigned)((a.nrows + ME - 1) / ME);
    EvalArgs b = a;
    b.nblk = (int32_t)nblk;
    b.ncand = ncand;
    b.wl_safe = p->tune.eval_no_wl;
    hipLaunchKernelGGL((k_eval<MBE, NRBW, MSP, XB, B3>), B3 ? dim3(nblk * (unsigned)ncand) : dim3(nblk, ncand), dim3(256), lds, st, b);
    return hipGetLastError();
}

static hipError_t launch_eval(mfas_population* p, const EvalArgs& a, int ncand, hipStream_t st) {
    // one or two row blocks (R <= 32): the m-blocks of a row tile are split over the four waves (eval.hip.h)
    // (bf16 tables: the rows stay 16-bit in LDS)
    const bool xb = a.tab.dtype == MFAS_DT_BF16 && !p->tune.eval_no_x16;
#define EV_SPLIT(M, S) if (p->mbe == M && p->nrbw == 1 && p->g.nrb == S && !p->tune.eval_no_msplit) \
        return xb ? launch_eval_t<M, 1, S, true>(p, a, ncand, st) : launch_eval_t<M, 1, S, false>(p, a, ncand, st);
    EV_SPLIT(4, 1) EV_SPLIT(4, 2) EV_SPLIT(2, 1) EV_SPLIT(2, 2) EV_SPLIT(1, 1) EV_SPLIT(1, 2)
#undef EV_SPLIT
    // two row blocks per wave (R = 72 .. 128), bf16 tables: exact bf16 x 3 feature products on the bf16 matrix pipe
    if (a.tab.dtype == MFAS_DT_BF16 && p->nrbw == 2 && !p->tune.eval_no_b3) {
        if (p->mbe == 4) return launch_eval_t<4, 2, 0, false, true>(p, a, ncand, st);
        if (p->mbe == 2) return launch_eval_t<2, 2, 0, false, true>(p, a, ncand, st);
        if (p->mbe == 1) return launch_eval_t<1, 2, 0, false, true>(p, a, ncand, st);
    }
#define EV_CASE(M, N) if (p->mbe == M && p->nrbw == N) return launch_eval_t<M, N>(p, a, ncand, st);
    EV_CASE(4, 1) EV_CASE(4, 2) EV_CASE(4, 4) EV_CASE(4, 8)
    EV_CASE(2, 1) EV_CASE(2, 2) EV_CASE(2, 4) EV_CASE(2, 8)
    EV_CASE(1, 1) EV_CASE(1, 2) EV_CASE(1, 4) EV_CASE(1, 8)
#undef EV_CASE
    return hipErrorInvalidValue;
}

// The resident persistent schedule needs every workgroup of its two launches on the GPU at the same time.  When that cannot be
// had — another process keeps CUs busy for good, the device is CU-masked, a tool serialises the two launches — the roll call fails
// BEFORE anything of the epoch has run (abort code 2), so the state in memory is that of the last completed epoch: rebuild the
// population in its launch-per-phase layout, carry W / m / v (+ the best-epoch snapshot) across through the reference's flat
// parameter order, and go on from the same epoch.  The handle keeps its identity: the two records swap contents.
static int persist_fallback(mfas_population* p) {
    const int K = p->K;
    std::vector<int32_t> confs((size_t)K * 12, 0), ncells(K);
    std::vector<uint32_t> seeds(K);
    int64_t maxp = 0;
    for (int k = 0; k < K; ++k) {
        const CandDev& c = p->cands[k];
        ncells[k] = c.L;
        seeds[k] = c.drop_seed;
        for (int i = 0; i < c.L; ++i)
            for (int j = 0; j < 3; ++j) confs[(k * 4 + i) * 3 + j] = c.conf[i][j];
        maxp = std::max(maxp, p->nparams[k]);
    }
    mfas_population* q = nullptr;
    int rc = create_impl(&p->hp, confs.data(), ncells.data(), seeds.data(), K, p->device, p->stream, p->chunk_cols_req, &q, false, &p->tune);
    if (rc) return rc;
    float* flat = nullptr;
    hipError_t e = hipMalloc(&flat, sizeof(float) * (size_t)maxp);
    if (e != hipSuccess) { mfas_population_destroy(q); return fail(MFAS_ENOMEM, "persist_fallback: scratch"); }
    if (p->best && !q->best) {
        e = hipMalloc(&q->best, sizeof(float) * (size_t)q->plane_stride);
        if (e == hipSuccess) e = hipMemsetAsync(q->best, 0, sizeof(float) * (size_t)q->plane_stride, p->stream);
        if (e != hipSuccess) { hipFree(flat); mfas_population_destroy(q); return fail(MFAS_ENOMEM, "persist_fallback: snapshot"); }
    }
    auto move = [&](int k, float* src_plane, int src_sel, float* dst_plane, int dst_sel, int mode, bool with_wt) {
        PackArgs a = pack_args(p, PK_GET, src_sel, flat);
        a.plane = src_plane;
        a.desc = p->d_descs + p->desc_start[k];
        hipMemsetAsync(flat, 0, sizeof(float) * p->nparams[k], p->stream);
        hipLaunchKernelGGL(k_pack, dim3(p->desc_start[k + 1] - p->desc_start[k]), dim3(256), 0, p->stream, a);
        hipLaunchKernelGGL(k_vec, dim3(1), dim3(256), 0, p->stream, a, k);
        PackArgs b = pack_args(q, mode, dst_sel, flat);
        b.plane = dst_plane;
        if (!with_wt) b.wt = nullptr;
        b.desc = q->d_descs + q->desc_start[k];
        hipLaunchKernelGGL(k_pack, dim3(q->desc_start[k + 1] - q->desc_start[k]), dim3(256), 0, p->stream, b);
        hipLaunchKernelGGL(k_vec, dim3(1), dim3(256), 0, p->stream, b, k);
    };
    for (int k = 0; k < K; ++k) {
        move(k, p->plane, 0, q->plane, 0, PK_SET, true);        // W (+ transposed OUT / HEAD images), zeroes m / v
        move(k, p->plane, 1, q->plane, 1, PK_PUT, false);       // Adam first moment
        move(k, p->plane, 2, q->plane, 2, PK_PUT, false);       // Adam second moment
        if (p->best) move(k, p->best, 0, q->best, 0, PK_PUT, false);
    }
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(q->d_posw, p->d_posw, sizeof(float) * p->g.Cp, hipMemcpyDeviceToDevice, p->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(q->d_status, p->d_status, sizeof(int32_t) * K, hipMemcpyDeviceToDevice, p->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
    hipFree(flat);
    if (e != hipSuccess) { mfas_population_destroy(q); return fail(MFAS_EHIP, std::string("persist_fallback: ") + hipGetErrorString(e)); }
    std::swap(q->d_stats, p->d_stats);
    std::swap(q->stats_cap, p->stats_cap);
    std::swap(q->d_scal, p->d_scal);
    std::swap(q->scal_cap, p->scal_cap);
    q->best_threshold = p->best_threshold;
    q->profiling = p->profiling;
    q->prof_every = p->prof_every;
    q->ev.swap(p->ev);
    q->fell_back = 1;
    std::swap(*p, *q);
    mfas_population_destroy(q);      // the resident layout
    return MFAS_OK;
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
    if (p->persist && p->nres > 0 && p->hp.tap_bits == 16 && train->dtype == MFAS_DT_F32)
        return fail(MFAS_EINVAL, "this population was created for 16-bit feature tables (mfas_hyper.tap_bits = 16); f32 tables need tap_bits = 32 or 0");
    if (N - (nb - 1) * B == 1 && g.bn)   // torch BatchNorm1d raises on a size-1 train batch
        return fail(MFAS_EINVAL, "final train batch of size 1 with batchnorm (reference raises ValueError)");

    if (p->stats_cap < K * epochs) {
        hipFree(p->d_stats); p->d_stats = nullptr;
        HIPCHK(hipMalloc(&p->d_stats, sizeof(DevStats) * K * epochs));
        p->stats_cap = K * epochs;
    }
    HIPCHK(hipMemsetAsync(p->d_stats, 0, sizeof(DevStats) * K * epochs, p->stream));
    HIPCHK(hipMemsetAsync(p->d_status, 0, sizeof(int32_t) * K, p->stream));
#ifdef MFAS_CHAIN_TIMING
    HIPCHK(hipMemsetAsync(p->d_status + 64 + 27, 0, sizeof(int32_t), p->stream));
    HIPCHK(hipMemsetAsync(p->d_status + 128, 0, 16 * sizeof(int32_t), p->stream));
    HIPCHK(hipMemsetAsync(p->d_status + 64 + 28, 0xFF, sizeof(int32_t), p->stream));
#endif
    // every call is a freshly built torch.optim.Adam (ntu_searchable.py:65; main_found_ntu.py:108,128): zero exp_avg / exp_avg_sq
    HIPCHK(hipMemsetAsync(p->plane + p->plane_stride, 0, sizeof(float) * 2 * (size_t)p->plane_stride, p->stream));
    if (snapshot_best && !p->best) HIPCHK(hipMalloc(&p->best, sizeof(float) * (size_t)p->plane_stride));
    // best_model_sd starts as a copy of the INITIAL state_dict (train_searchable/ntu.py:17) and is what the model is
    // left with if no epoch's dev metric beats the starting threshold (0 for accuracy, init_f1 for F1)
    if (snapshot_best && max_steps < 0)
        HIPCHK(hipMemcpyAsync(p->best, p->plane, sizeof(float) * (size_t)p->plane_stride, hipMemcpyDeviceToDevice, p->stream));
    std::vector<double> best_acc(K, p->best_threshold);
    const double metric_scale = g.loss_mode == 1 ? 1.0 / 4294967296.0 : 1.0;   // F1 sums are 32.32 fixed point
    std::vector<DevStats> hstats((size_t)K * epochs);

    RangeGuard call_range("mfas_population_train K=" + std::to_string(K) + " R=" + std::to_string(g.R) + " B=" + std::to_string(B) +
                          " E=" + std::to_string(epochs) + (p->persist ? " resident" : " launch-per-phase"));
    const mfas_hyper& hp = p->hp;
    AdamC ac;
    ac.w1 = (float)(1.0 - hp.beta1); ac.b2 = (float)hp.beta2; ac.w2 = (float)(1.0 - hp.beta2);
    ac.eps = (float)hp.adam_eps; ac.wd = (float)hp.wd; ac.ss = 0.f; ac.bc2s = 1.f;

    // Candidate groups A/B: every launch pairs the sweep of one group with the chain of the other (k_step).
    int NG = 0;
    int64_t split_launches[2] = {0, 0}; // chain_split launches of this call per candidate group (exchange parity)
    StepArgs st;
    auto init_args = [&]() -> hipError_t {     // (again after persist_fallback: the population's buffers and layout have changed)
        NG = (int)p->groups.size();
        p->g.order_stride = (p->hp.order_per_candidate && order) ? (int64_t)epochs * N : 0;    // order: [K][epochs][N_train]
        memset(&st, 0, sizeof(st));
        st.sa.cands = p->d_cands; st.sa.plane = p->plane; st.sa.plane_stride = p->plane_stride; st.sa.wt = p->wt;
        st.sa.stepbuf = p->stepbuf; st.sa.tab = *train; st.sa.order = order; st.sa.g = p->g; st.sa.ac = ac;
        st.ca.plane = p->plane; st.ca.plane_stride = p->plane_stride; st.ca.wt = p->wt; st.ca.stepbuf = p->stepbuf;
        st.ca.tab = *train; st.ca.order = order; st.ca.E = epochs; st.ca.g = p->g; st.ca.stats = p->d_stats;
        st.ca.status = p->d_status; st.ca.ac = ac; st.ca.yf_in_lds = p->yf_in_lds ? 1 : 0; st.ca.pos_w = p->d_posw;
        st.ca.vec_in_lds = p->vec_in_lds ? 1 : 0;
        st.sa.red_cnt = p->red_in_sweep ? p->d_red_cnt : nullptr;
        st.ca.yf_reduced = p->red_in_sweep ? 1 : 0;
        hipError_t e_ = hipSuccess;
        if (p->red_in_sweep) e_ = hipMemsetAsync(p->d_red_cnt, 0, sizeof(uint32_t) * K * MFAS_MAX_CELLS, p->stream);
        if (e_ == hipSuccess && p->same_group) e_ = hipMemsetAsync(p->d_cellflag, 0, sizeof(uint32_t) * K * CELLFLAG_STRIDE, p->stream);
        // chain_split: every piece of both parities "not written" (all-ones words), parity counter back to 0
        if (e_ == hipSuccess && p->chain_split) e_ = hipMemsetAsync(p->d_xch, 0xFF, sizeof(float) * (size_t)K * XCH_CAND_FLOATS, p->stream);
        st.ca.xch = p->d_xch; st.ca.nsplit = p->chain_split; st.ca.xpar = 0;
        split_launches[0] = split_launches[1] = 0;
        return e_;
    };
    HIPCHK(init_args());

    const int elt = train->dtype == MFAS_DT_F32 ? 4 : 2;
    p->prof_launches = 0; p->prof_ms = 0.0; p->prof_bytes = 0.0;
    size_t ev_used = 0;
    std::vector<double> ev_bytes;
    int64_t nlaunch = 0;

    // Gathered rows (sweep.hip.h, gather_body): two-group streaming schedule + per-candidate sample orders.  The rows of batch
    // t + 1 of group g are gathered by the launch that carries chain(g, t) (t >= 1) — the launch BEFORE sweep(g, t), which stages
    // them as x_{t+1} and, a step later, as x_t; batches 0 and 1 are gathered by the group's forward-only prologue launch.
    int64_t Tcur = 0;
    bool use_gather = false;
    int64_t g_par_stride = 0, g_cand_stride = 0;
    auto setup_gather = [&]() -> hipError_t {
        use_gather = NG == 2 && !p->persist && order && p->g.order_stride > 0 && p->lp_group == 1 && !p->tune.no_gather;
        if (!use_gather) return hipSuccess;
        int64_t totw = 0;
        for (int u = 0; u < MFAS_MAX_TAPS; ++u) totw += p->g.sw[u] + p->g.vw[u];
        g_par_stride = totw * p->g.Bp * (train->dtype == MFAS_DT_F32 ? 4 : 2);
        g_cand_stride = 2 * g_par_stride;
        const size_t need = (size_t)g_cand_stride * K;
        if (p->gather_cap < need) {
            hipFree(p->d_gather); p->d_gather = nullptr; p->gather_cap = 0;
            hipError_t e = hipMalloc(&p->d_gather, need);
            if (e != hipSuccess) { use_gather = false; (void)hipGetLastError(); return hipSuccess; }   // an optimisation: train without it
            p->gather_cap = need;
        }
        if (p->tune.gather_verbose) fprintf(stderr, "[gather] on: %d candidates, %.1f MB of gathered rows\n", K, (double)need / 1e6);
        return hipSuccess;
    };
    HIPCHK(setup_gather());
    auto gather_set = [&](GatherArgs& ga, int s, int64_t ep, int64_t t) {
        ga.pos[s] = ep * N + t * B; ga.base[s] = (int)(t * B);
        ga.nvalid[s] = (int)std::min<int64_t>(B, N - t * B); ga.par[s] = (int)(t & 1);
    };

    // one fused launch: sweep of group gs at step ts (gs < 0: none) + chain of group gc at step tc (gc < 0: none)
    auto step = [&](int gs, int upd, int fwd, int64_t ep, int64_t ts, int gc, int64_t tc) {
        unsigned nsw = 0, nch = 0;
        st.ga.nblocks = 0; st.ga.nsets = 0; st.sa.gather = nullptr;
        if (use_gather) {
            GatherArgs& ga = st.ga;
            ga.buf = p->d_gather; ga.cand_stride = g_cand_stride; ga.par_stride = g_par_stride;
            int gg = -1;
            if (gs >= 0 && !upd && fwd && ts == 0) {                    // prologue of group gs: batches 0 and 1
                gg = gs;
                gather_set(ga, 0, ep, 0); ga.nsets = 1;
                if (Tcur > 1) { gather_set(ga, 1, ep, 1); ga.nsets = 2; }
            } else if (gc >= 0 && gs >= 0 && tc >= 1 && tc + 1 < Tcur) { // chain(gc, tc) rides with a sweep: batch tc + 1 of group gc
                gg = gc;
                gather_set(ga, 0, ep, tc + 1); ga.nsets = 1;
            }
            if (gg >= 0) { ga.cands = p->d_cands + p->groups[gg].c0; ga.nblocks = p->groups[gg].nc; }
            if (gs >= 0 && upd) {
                st.sa.gather = p->d_gather; st.sa.g_cand_stride = g_cand_stride; st.sa.g_par_stride = g_par_stride;
                st.sa.g_par_t = (int)(ts & 1); st.sa.g_par_n = (int)((ts + 1) & 1);
            }
        }
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
        if (p->same_group && gs >= 0 && gc == gs && upd) {   // chain(g, t) and sweep(g, t) in ONE launch, per-cell flags
            st.sa.cellflag = p->d_cellflag; st.ca.cellflag = p->d_cellflag;
            st.sa.flag_target = st.ca.flag_target = (uint32_t)st.ca.gstep + 1u;
            st.sa.flag_status = p->d_status;
            if (p->chain_split) {      // NS parts per candidate: the per-cell flags count arrivals, chain blocks = NS * ceil8(candidates)
                const int NS = p->chain_split, Kp = (int)((nch + 7) & ~7u);
                st.sa.flag_target = st.ca.flag_target = (uint32_t)NS * ((uint32_t)st.ca.gstep + 1u);
                st.ca.ncand = (int)nch; st.ca.xpar = (int)(split_launches[0]++ & 1);
                st.nchain = NS * Kp;
                if (p->nontemporal) hipLaunchKernelGGL((k_step_same<1, true, 4>), dim3(st.nchain + nsw), dim3(STEP_THREADS), p->lds_split, p->stream, st);
                else hipLaunchKernelGGL((k_step_same<1, false, 4>), dim3(st.nchain + nsw), dim3(STEP_THREADS), p->lds_split, p->stream, st);
            } else
            if (g.MB == 1) { if (p->nontemporal) hipLaunchKernelGGL((k_step_same<1, true>), dim3(nch + st.ga.nblocks + nsw), dim3(STEP_THREADS), p->lds_step, p->stream, st);
                             else hipLaunchKernelGGL((k_step_same<1, false>), dim3(nch + st.ga.nblocks + nsw), dim3(STEP_THREADS), p->lds_step, p->stream, st); }
            else { if (p->nontemporal) hipLaunchKernelGGL((k_step_same<2, true>), dim3(nch + st.ga.nblocks + nsw), dim3(STEP_THREADS), p->lds_step, p->stream, st);
                   else hipLaunchKernelGGL((k_step_same<2, false>), dim3(nch + st.ga.nblocks + nsw), dim3(STEP_THREADS), p->lds_step, p->stream, st); }
            st.sa.cellflag = nullptr; st.ca.cellflag = nullptr;
        } else if (p->chain_split && nch > 0) {      // two-group launch whose chain blocks are chain_split parts (no flags: the kernel boundary)
            const int NS = p->chain_split, Kp = (int)((nch + 7) & ~7u);
            st.ca.ncand = (int)nch; st.ca.xpar = (int)(split_launches[gc & 1]++ & 1);
            st.nchain = NS * Kp;
            if (p->nontemporal) hipLaunchKernelGGL((k_step<1, true, 4, false, 4>), dim3(st.nchain + st.ga.nblocks + nsw), dim3(STEP_THREADS), p->lds_split, p->stream, st);
            else hipLaunchKernelGGL((k_step<1, false, 4, false, 4>), dim3(st.nchain + st.ga.nblocks + nsw), dim3(STEP_THREADS), p->lds_split, p->stream, st);
        } else {
#define STEP_LAUNCH(M, T, W, F) hipLaunchKernelGGL((k_step<M, T, W, F>), dim3(nch + st.ga.nblocks + nsw), dim3(STEP_THREADS), p->lds_step, p->stream, st)
#define STEP_PICK(M, W) do { if (p->nontemporal) { if (p->lean_chain) STEP_LAUNCH(M, true, W, true); else STEP_LAUNCH(M, true, W, false); } \
                             else { if (p->lean_chain) STEP_LAUNCH(M, false, W, true); else STEP_LAUNCH(M, false, W, false); } } while (0)
        // MB == 2: the two-workgroups-per-CU build unless a co-scheduled chain would bound the launch (see SweepU)
        const bool occ = nch == 0 || p->groups[gs].alg_state > p->occ_bytes;
        if (g.MB == 1) STEP_PICK(1, 4);
        // (lean chain: the 128-VGPR build spills 8 registers of the element-parallel chain to scratch and is still the faster
        //  one — R=16, B=20, 50 / 128 / 512 candidates: 47.2 / 99.2 / 418 us per step against 50.5 / 117.4 / 447 for the 2-workgroup build)
        else if (g.MB == 2) { if (occ || p->lean_chain) STEP_PICK(2, 4); else STEP_PICK(2, 2); }
        else { if (p->nontemporal) STEP_LAUNCH(4, true, 2, false); else STEP_LAUNCH(4, false, 2, false); }
#undef STEP_PICK
#undef STEP_LAUNCH
        }
        if (prof) {
            hipEventRecord(p->ev[ev_used + 1], p->stream);
            ev_used += 2;
            // algorithmic HBM bytes of this group's update+forward sweep: 24 B/param + the batch's taps + labels
            ev_bytes.push_back(p->groups[gs].alg_state + p->groups[gs].alg_feat * elt + 8.0 * B * p->groups[gs].nc);
        }
    };

    std::vector<uint32_t> aborts(epochs, 0u);
    if (p->persist) {   // the step scalars live on the device: the kernel walks the steps itself
        const size_t nsc = (size_t)epochs * nb * 2;
        if (p->scal_cap < nsc) {
            hipFree(p->d_scal); p->d_scal = nullptr;
            HIPCHK(hipMalloc(&p->d_scal, sizeof(float) * nsc));
            p->scal_cap = nsc;
        }
        const size_t have = (size_t)(max_steps >= 0 ? std::min<int64_t>(max_steps, (int64_t)epochs * nb) : (int64_t)epochs * nb) * 2;
        HIPCHK(hipMemcpyAsync(p->d_scal, step_scalars, sizeof(float) * have, hipMemcpyHostToDevice, p->stream));
    }
    // one persistent launch = all train steps of one epoch (persist.hip.h)
    const int test_not_resident = p->tune.test_not_resident;   // (-1 in the product library; the MFAS_TEST_HOOKS variant: from this epoch on
                                                               //  every roll call "fails" — nothing is launched)
    auto persist_epoch_once = [&](int ep, int64_t T) -> hipError_t {
        if (test_not_resident >= 0 && ep >= test_not_resident) { aborts[ep] = PERSIST_ABORT_NOT_RESIDENT; return hipSuccess; }
        hipError_t e = hipMemsetAsync(p->d_sync, 0, sizeof(uint32_t) * ((size_t)K * PERSIST_SYNC_STRIDE + 64), p->stream);
        if (e != hipSuccess) return e;
        PersistArgs pa;
        memset(&pa, 0, sizeof(pa));
        pa.sa = st.sa; pa.ca = st.ca;
        pa.sa.desc = p->d_pdescs; pa.sa.tdesc = nullptr; pa.sa.ntap = 0;
        pa.ca.cands = p->d_cands;
        pa.nchain = K; pa.nitems = p->n_pdescs; pa.nres = p->nres; pa.res_chain = p->res_chain ? 1 : 0; pa.res_wide = p->res_wide ? 1 : 0;
        pa.res_nu = p->res_nu; pa.nres_wg = p->nres_wg; pa.res_buf_words = p->res_buf_words;
        pa.T = (int)T; pa.epoch = ep;
        pa.lose_step = p->tune.test_lose_step;
        pa.N = N; pa.pos0 = (int64_t)ep * N;
        pa.B = B; pa.gstep0 = (int)((int64_t)ep * nb);
        pa.scal = p->d_scal; pa.sync = p->d_sync; pa.need = p->d_need; pa.role = p->d_role; pa.trace = p->d_trace;
        const unsigned grid = (unsigned)(K + pa.nres_wg);
        if ((int)grid > p->n_cus) return hipErrorInvalidConfiguration;
        const bool prof = p->profiling;
        if (prof) {
            if (p->ev.size() < ev_used + 2) {
                hipEvent_t e0, e1;
                hipEventCreate(&e0); hipEventCreate(&e1);
                p->ev.push_back(e0); p->ev.push_back(e1);
            }
            hipEventRecord(p->ev[ev_used], p->stream);
        }
        {      // one instantiation per unit form
            const int lw = (int)(p->lds_president / 4) - PERSIST_LDS_WORDS;
            // the search default — no BatchNorm, no alphas, single-task softmax CE — runs the chain compiled for exactly that (chain_lean PLAIN)
            // (round 6: and `--batchnorm` alone, /root/reference/main_searchable_ntu.py:48, the chain compiled for exactly THAT — PLAIN = 2)
            const bool simple = !g.alphas && !g.multitask && g.loss_mode == 0 && !p->tune.no_plain_chain;
            const int plain = simple ? (g.bn ? 2 : 1) : 0;
#define RES_LAUNCH(M, NTR, X, NU) do { if (plain == 1) hipLaunchKernelGGL((k_president<M, NTR, X, NU, 1>), dim3(grid), dim3(STEP_THREADS), p->lds_president, p->stream, pa, lw); \
                                       else if (plain == 2) hipLaunchKernelGGL((k_president<M, NTR, X, NU, 2>), dim3(grid), dim3(STEP_THREADS), p->lds_president, p->stream, pa, lw); \
                                       else hipLaunchKernelGGL((k_president<M, NTR, X, NU, 0>), dim3(grid), dim3(STEP_THREADS), p->lds_president, p->stream, pa, lw); } while (0)
#define RES_PICK(M) do { if (train->dtype == MFAS_DT_F32) { if (pa.res_nu == 2) RES_LAUNCH(M, PERSIST_NTR, false, 2); else RES_LAUNCH(M, PERSIST_NTR, false, 1); } \
                         else if (pa.res_wide) RES_LAUNCH(M, PERSIST_NTR16, true, 1); \
                         else if (pa.res_nu == 2) RES_LAUNCH(M, PERSIST_NTR, true, 2); else RES_LAUNCH(M, PERSIST_NTR, true, 1); } while (0)
            if (g.MB == 1) RES_PICK(1); else RES_PICK(2);
#undef RES_PICK
#undef RES_LAUNCH
        }
        if (prof) {
            hipEventRecord(p->ev[ev_used + 1], p->stream);
            ev_used += 2;
            // algorithmic bytes of the launch: T update+forward sweeps of every candidate
            ev_bytes.push_back((double)T * (p->groups[0].alg_state + p->groups[0].alg_feat * elt + 8.0 * B * K));
        }
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        return hipMemcpyAsync(&aborts[ep], p->d_sync + (size_t)K * PERSIST_SYNC_STRIDE, sizeof(uint32_t), hipMemcpyDeviceToHost, p->stream);
    };
    // The launch is only valid when its whole grid is resident at once (roll call, persist.hip.h).  When another process holds
    // part of the GPU the roll call fails BEFORE anything is modified (abort code 2): wait a little (jittered, so that two
    // processes that collided do not collide again in lockstep) and launch the epoch again — up to PERSIST_MAX_RELAUNCHES times
    // (~0.3 s of trying); after that the caller gives the resident schedule up for this population (persist_fallback).
    auto persist_epoch = [&](int ep, int64_t T) -> hipError_t {
        for (int attempt = 0;; ++attempt) {
            hipError_t e = persist_epoch_once(ep, T);
            if (e != hipSuccess) return e;
            e = hipStreamSynchronize(p->stream);
            if (e != hipSuccess) return e;
            if (p->tune.persist_verbose >= 2)
                fprintf(stderr, "[persist] epoch %d attempt %d: abort word %u\n", ep, attempt, aborts[ep]);
            if (aborts[ep] != PERSIST_ABORT_NOT_RESIDENT || attempt >= PERSIST_MAX_RELAUNCHES || test_not_resident >= 0) {
                if (attempt && p->tune.persist_verbose) fprintf(stderr, "[persist] epoch %d: grid not resident at once, relaunched %d time(s)\n", ep, attempt);
                return hipSuccess;
            }
            if (p->profiling && ev_used >= 2) { ev_used -= 2; ev_bytes.pop_back(); }       // the failed attempt is not a measurement
            aborts[ep] = 0;
            std::this_thread::sleep_for(std::chrono::microseconds(200 + (uint64_t)((reinterpret_cast<uintptr_t>(p) >> 6) * 2654435761u % 1800u) + 50u * (attempt % 16)));
        }
    };

    int64_t done = 0;   // train steps completed (max_steps bookkeeping)
    for (int ep = 0; ep < epochs; ++ep) {
        int64_t T = nb;
        if (max_steps >= 0) T = std::min<int64_t>(nb, max_steps - done);
        if (T <= 0) break;
        Tcur = T;
        RangeGuard epoch_range("epoch " + std::to_string(ep));
        if (p->persist) {
            HIPCHK(persist_epoch(ep, T));
            if (aborts[ep] == PERSIST_ABORT_NOT_RESIDENT) {
                // every attempt failed its roll call: nothing of this epoch has run.  Train it — and the rest — launch per phase.
                if (p->tune.persist_verbose) fprintf(stderr, "[persist] epoch %d: the resident grid never became resident; falling back to launch-per-phase\n", ep);
                rc = persist_fallback(p);
                if (rc) return rc;
                HIPCHK(init_args());
                HIPCHK(setup_gather());
                aborts[ep] = 0;
            } else if (aborts[ep]) {
                HIPCHK(hipStreamSynchronize(p->stream));
                return fail(MFAS_EHIP, "persistent step loop: a workgroup timed out waiting for its dependency (abort code 1: the epoch was "
                                       "abandoned half way, this population's parameters are not usable)");
            }
        }
        if (!p->persist) {
        for (int gi = 0; gi < NG; ++gi) step(gi, 0, 1, ep, 0, -1, 0);   // prologue: forward sums of batch 0
        if (NG == 1 && p->same_group) {
            for (int64_t t = 0; t < T; ++t) step(0, 1, (t + 1 < T) ? 1 : 0, ep, t, 0, t);
        } else if (NG == 1) {
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
                    const double acc = (double)hstats[(size_t)k * epochs + ep].dev_corr * metric_scale / (double)dev->N;
                    if (acc > best_acc[k]) {   // strict >, from 0 (train_searchable/ntu.py:82) / init_f1 (mmimdb.py:18)
                        best_acc[k] = acc;
                        HIPCHK(hipMemcpyAsync(p->best + p->cand_plane_base[k], p->plane + p->cand_plane_base[k],
                                              sizeof(float) * p->cand_plane_size[k], hipMemcpyDeviceToDevice, p->stream));
                    }
                }
            }
        }
    }
    if (snapshot_best && do_dev) {   // model.load_state_dict(best_model_sd) (:86), unconditionally
        HIPCHK(hipMemcpyAsync(p->plane, p->best, sizeof(float) * (size_t)p->plane_stride, hipMemcpyDeviceToDevice, p->stream));
        // the transposed OUT / HEAD tiles the backward chain reads still hold the last epoch's weights: re-derive them
        PackArgs pa = pack_args(p, PK_WT, 0, nullptr);
        hipLaunchKernelGGL(k_pack, dim3((unsigned)p->descs.size()), dim3(256), 0, p->stream, pa);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipMemcpyAsync(hstats.data(), p->d_stats, sizeof(DevStats) * K * epochs, hipMemcpyDeviceToHost, p->stream));
    std::vector<int32_t> hstatus(K, 0);
    HIPCHK(hipMemcpyAsync(hstatus.data(), p->d_status, sizeof(int32_t) * K, hipMemcpyDeviceToHost, p->stream));
    HIPCHK(hipStreamSynchronize(p->stream));
    HIPCHK(hipGetLastError());
    for (uint32_t ab : aborts)
        if (ab) return fail(MFAS_EHIP, ab == PERSIST_ABORT_NOT_RESIDENT ? "persistent step loop: the grid never became resident (abort code 2)"
                                                                         : "persistent step loop: a workgroup timed out waiting for its dependency (abort code 1)");
    for (int32_t sv : hstatus)
        if (sv == 2) return fail(MFAS_EHIP, "same-group fused launch: a sweep unit timed out waiting for its cell's dy");
    if (p->d_trace && p->persist) {
        unsigned long long tr[256];
        if (hipMemcpy(tr, p->d_trace, sizeof(tr), hipMemcpyDeviceToHost) == hipSuccess) {
            // step 12 of candidate 0: chain published at tr[4*8+3]; per resident unit: saw-flag / compute-done / arrived, relative to it
            const long long pub = (long long)tr[4 * 8 + 3];
            fprintf(stderr, "[persist trace step 12, candidate 0 units, ticks after the chain published: saw-flag done arrived]");
            for (int u = 0; u < 64; ++u)
                if (tr[64 + u]) fprintf(stderr, " u%d:%lld/%lld/%lld", u, (long long)tr[64 + u] - pub, (long long)tr[128 + u] - pub, (long long)tr[192 + u] - pub);
            fprintf(stderr, "\n[chain ready for step 13 at +%lld]\n", (long long)tr[5 * 8 + 1] - pub);
            fprintf(stderr, "[persist trace, 10 ns ticks; per step: chain wait0 ready done published | sweep-unit-0 wait0 ready done arrived]\n");
            for (int t = 0; t < 8; ++t) {
                fprintf(stderr, "  step %2d:", t + 8);
                for (int j = 0; j < 8; ++j) fprintf(stderr, " %lld", (long long)(tr[t * 8 + j] - tr[0]));
                fprintf(stderr, "\n");
            }
        }
    }
    for (size_t i = 0; i < hstats.size(); ++i) {
        stats[i].train_loss_sum = hstats[i].train_loss;
        stats[i].dev_loss_sum = hstats[i].dev_loss;
        stats[i].train_corrects = hstats[i].train_corr;
        stats[i].dev_corrects = hstats[i].dev_corr;
    }
    if (status) memcpy(status, hstatus.data(), sizeof(int32_t) * K);
#ifdef MFAS_CHAIN_TIMING
    {
        int32_t ts[40];
        if (hipMemcpy(ts, p->d_status + 64, sizeof(ts), hipMemcpyDeviceToHost) == hipSuccess) {
            fprintf(stderr, "[chain timing, shader cycles since kernel entry, candidate 0 step 3]");
            for (int i = 0; i < 13; ++i) fprintf(stderr, " %d", ts[i]);
            if (p->chain_split) {      // chain_split's extra stamps: forward cell 1 product done | out sent | tail done | fetched;  backward cell 2 the same;  softmax done;  entry staged
                fprintf(stderr, "  | split:");
                for (int i = 13; i < 23; ++i) fprintf(stderr, " %d", ts[i]);
                fprintf(stderr, "  | entry: record here %d, slabs summed %d, vector block in LDS %d", ts[34], ts[35], ts[36]);
                fprintf(stderr, "  | 10 ns ticks: chain of step 3 %d, end of chain 3 -> entry of chain 4 %d, chain of step 4 %d", ts[24] - ts[23], ts[25] - ts[24], ts[26] - ts[25]);
                fprintf(stderr, "; end of chain 3 -> first cell-0 unit sees its flag %d -> last sweep unit of the launch ends %d -> entry of chain 4 %d", (int32_t)((uint32_t)ts[28] - (uint32_t)ts[24]), (int32_t)((uint32_t)ts[27] - (uint32_t)ts[28]), (int32_t)((uint32_t)ts[25] - (uint32_t)ts[27]));
                fprintf(stderr, "; unit (cell 0, S, chunk 0) after the end of chain 3: flag seen %d, dy staged %d, tiles done %d, slab drained %d, arrival counted %d",
                        ts[29] - ts[24], ts[30] - ts[24], ts[31] - ts[24], ts[32] - ts[24], ts[33] - ts[24]);
                int32_t ue[16];
                if (hipMemcpy(ue, p->d_status + 128, sizeof(ue), hipMemcpyDeviceToHost) == hipSuccess) {
                    fprintf(stderr, "; last unit end after the end of chain 3, per cell [S V OUT HEAD]:");
                    for (int i = 0; i < 16; ++i) fprintf(stderr, "%s%d", (i & 3) ? " " : " | ", ue[i] ? (int32_t)((uint32_t)ue[i] - (uint32_t)ts[24]) : 0);
                }
            }
            fprintf(stderr, "\n");
            int32_t cs[24];
            if (hipMemcpy(cs, p->d_status + 96, sizeof(cs), hipMemcpyDeviceToHost) == hipSuccess) {
                fprintf(stderr, "[chain checksums, candidate 0 global step 0: sums x4, out x4, logits, dlogits, dy x4, d x4]");
                for (int i = 0; i < 18; ++i) fprintf(stderr, " %08x", (unsigned)cs[i]);
                fprintf(stderr, "\n");
            }
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

// One batch through candidate k in TRAIN mode: forward only (logits out), or forward + backward of an external loss
// (dlogits in): then every parameter's Adam first-moment slot receives its exact GRADIENT and nothing else changes — the step
// runs with beta1 = 0 (m <- m + 1 * (g - m) = g), weight decay 0 and learning rate 0 (w <- w - 0 * m / denom = w).
static int single_batch(mfas_population* p, int32_t k, const mfas_table* tab, int64_t row0, int32_t nrows, int32_t step_index,
                        float* logits, const float* dlogits) {
    if (!p || (!logits && !dlogits) || k < 0 || k >= p->K || row0 < 0) return fail(MFAS_EINVAL, "bad argument");
    int rc = check_table(p, tab, false);
    if (rc) return rc;
    const Geo& g = p->g;
    if (nrows < 1 || nrows > g.B) return fail(MFAS_EINVAL, "train-mode forward: 1 <= rows <= the population's batch size");
    if (nrows == 1 && g.bn) return fail(MFAS_EINVAL, "train-mode BatchNorm needs more than 1 row (reference: ValueError)");
    if (row0 + nrows > tab->N) return fail(MFAS_EINVAL, "row range outside the table");
    HIPCHK(hipSetDevice(p->device));
    AdamC ac;
    ac.w1 = 1.0f; ac.b2 = (float)p->hp.beta2; ac.w2 = (float)(1.0 - p->hp.beta2); ac.eps = (float)p->hp.adam_eps; ac.wd = 0.f; ac.ss = 0.f; ac.bc2s = 1.f;
    StepArgs st;
    memset(&st, 0, sizeof(st));
    st.sa.cands = p->d_cands; st.sa.plane = p->plane; st.sa.plane_stride = p->plane_stride; st.sa.wt = p->wt;
    st.sa.stepbuf = p->stepbuf; st.sa.tab = *tab; st.sa.order = nullptr; st.sa.g = g; st.sa.g.order_stride = 0;
    st.sa.desc = p->d_mdescs + p->mdesc_start[k]; st.sa.tdesc = nullptr; st.sa.ntap = 0;
    st.sa.do_update = 0; st.sa.do_forward = 1;
    st.sa.pos_n = row0; st.sa.base_n = (int)row0; st.sa.nvalid_n = nrows;
    st.sa.pos_t = row0; st.sa.base_t = (int)row0; st.sa.nvalid_t = nrows;
    st.sa.ac = ac;
    st.nchain = 0;
    const unsigned nsw = (unsigned)(p->mdesc_start[k + 1] - p->mdesc_start[k]);
    size_t lds_need = p->lds_step;   // (a population laid out for resident units budgets its streaming LDS without them)
    for (int j = p->desc_start[k]; j < p->desc_start[k + 1]; ++j) {
        const SegDesc& d = p->descs[j];
        size_t fl = (size_t)g.Bp * (d.cc + 16) + (size_t)g.Bp * (d.cc + 4) + (size_t)g.Bp * (d.rows_p + 16);
        if (d.rows_p / 16 < STEP_NW && d.kind <= KIND_V) fl += (size_t)STEP_NW * (d.rows_p / 16) * g.MB * 256;
        lds_need = std::max(lds_need, fl * 4);
    }
    if (lds_need > 150 * 1024) return fail(MFAS_EINVAL, "train-mode forward: this population's units are too wide for the streaming kernels");
    if (lds_need > p->lds_step) {
        if (g.MB == 1) HIPCHK(set_lds((k_step<1, false, 4, false>), lds_need));
        else if (g.MB == 2) HIPCHK(set_lds((k_step<2, false, 2, false>), lds_need));
        else HIPCHK(set_lds((k_step<4, false, 2, false>), lds_need));
    }
    auto sweep = [&]() {
        if (g.MB == 1) hipLaunchKernelGGL((k_step<1, false, 4, false>), dim3(nsw), dim3(STEP_THREADS), lds_need, p->stream, st);
        else if (g.MB == 2) hipLaunchKernelGGL((k_step<2, false, 2, false>), dim3(nsw), dim3(STEP_THREADS), lds_need, p->stream, st);
        else hipLaunchKernelGGL((k_step<4, false, 2, false>), dim3(nsw), dim3(STEP_THREADS), lds_need, p->stream, st);
    };
    if (dlogits) {
        // The gradient lands in the first-moment slot as m <- m + 1 * (g - m): exact only from m = 0 (1 + (1e-9 - 1) cancels to 0),
        // and a stale second moment would turn the zero-step's 0 * (m / denom) into 0 * inf.  Whatever this handle has trained
        // before, candidate k's m and v planes start from zero here (the header documents them as scratch after this call).
        HIPCHK(hipMemsetAsync(p->plane + p->plane_stride + p->cand_plane_base[k], 0, sizeof(float) * (size_t)p->cand_plane_size[k], p->stream));
        HIPCHK(hipMemsetAsync(p->plane + 2 * p->plane_stride + p->cand_plane_base[k], 0, sizeof(float) * (size_t)p->cand_plane_size[k], p->stream));
    }
    // 1. forward partial sums of the batch (no update): the sweep's forward half over this candidate's units
    sweep();
    // 2. the chain: batch-statistics BN (running statistics move like in any train-mode forward), dropout stream of step_index;
    //    forward only: stops at the logits; backward: continues from the caller's dL/dlogits and leaves dy_i for the sweep
    ChainArgs& c = st.ca;
    c.cands = p->d_cands + k; c.plane = p->plane; c.plane_stride = p->plane_stride; c.wt = p->wt; c.stepbuf = p->stepbuf;
    c.tab = *tab; c.order = nullptr; c.pos_t = row0; c.base_t = (int)row0; c.nvalid = nrows;
    c.gstep = step_index; c.epoch = 0; c.E = 1; c.g = st.sa.g; c.stats = nullptr; c.status = p->d_status;
    c.yf_in_lds = p->yf_in_lds ? 1 : 0; c.vec_in_lds = p->vec_in_lds ? 1 : 0; c.pos_w = p->d_posw;
    c.yf_reduced = 0; c.logits_out = dlogits ? nullptr : logits; c.dlogits_in = dlogits; c.ac = ac;
#define CHAIN_LAUNCH(M, F) hipLaunchKernelGGL((k_chain<M, F>), dim3(1), dim3(STEP_THREADS), p->lds_chain, p->stream, st.ca)
    if (p->lean_chain) { if (g.MB == 1) CHAIN_LAUNCH(1, true); else CHAIN_LAUNCH(2, true); }
    else if (g.MB == 1) CHAIN_LAUNCH(1, false);
    else if (g.MB == 2) CHAIN_LAUNCH(2, false);
    else CHAIN_LAUNCH(4, false);
#undef CHAIN_LAUNCH
    if (dlogits) {   // 3. dW of every matrix into its m slot (see the header comment); W, v-scaled-by-lr-0 steps leave W as it was
        st.sa.do_update = 1; st.sa.do_forward = 0;
        sweep();
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(p->stream));
    return MFAS_OK;
}

extern "C" int mfas_population_forward_train(mfas_population* p, int32_t k, const mfas_table* tab, int64_t row0, int32_t nrows,
                                             int32_t step_index, float* logits) {
    if (!logits) return fail(MFAS_EINVAL, "bad argument");
    return single_batch(p, k, tab, row0, nrows, step_index, logits, nullptr);
}

extern "C" int mfas_population_backward(mfas_population* p, int32_t k, const mfas_table* tab, int64_t row0, int32_t nrows,
                                        int32_t step_index, const float* dlogits) {
    if (!dlogits) return fail(MFAS_EINVAL, "bad argument");
    return single_batch(p, k, tab, row0, nrows, step_index, nullptr, dlogits);
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

extern "C" int mfas_population_set_best_threshold(mfas_population* p, double threshold) {
    if (!p) return fail(MFAS_EINVAL, "null");
    p->best_threshold = threshold;
    return MFAS_OK;
}

extern "C" int mfas_population_set_profiling(mfas_population* p, int32_t on) {
    if (!p) return fail(MFAS_EINVAL, "null");
    p->profiling = on != 0;
    p->prof_every = p->tune.prof_every;
    return MFAS_OK;
}

extern "C" int mfas_population_schedule(const mfas_population* p, int32_t info[8]) {
    if (!p || !info) return fail(MFAS_EINVAL, "null");
    info[0] = p->persist ? 1 : 0; info[1] = p->nres; info[2] = p->nres_wg; info[3] = p->res_nu;
    info[4] = p->res_chain ? 1 : 0; info[5] = (p->lean_chain ? 1 : 0) | (std::max(1, p->chain_split) << 8); info[6] = p->same_group ? -1 : (int32_t)p->groups.size(); info[7] = p->K;
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

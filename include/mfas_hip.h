/* mfas_hip.h — C ABI of the MI355X-native MFAS inner candidate-training engine.
 *
 * The reference (jperezrua/mfas) has no native/FFI layer: its hot path is Python calling PyTorch.
 * These entry points are what a binding for that path replaces (reference file:line cited per
 * function).  Plain pointers and sizes only; every tensor pointer is a DEVICE pointer owned by the
 * caller unless stated otherwise; functions return 0 on success or a negative MFAS_E* code and never
 * throw across the boundary.  One population handle is driven by one host thread on one HIP stream.
 */
#ifndef MFAS_HIP_H
#define MFAS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MFAS_OK 0
#define MFAS_EINVAL -1   /* bad argument / unsupported hyper-parameter combination */
#define MFAS_EHIP -2     /* a HIP runtime call failed; see mfas_last_error() */
#define MFAS_ENOMEM -3

#define MFAS_MAX_CELLS 4 /* max_fusions (main_searchable_ntu.py:39) */
#define MFAS_MAX_TAPS 8  /* tap slots per modality (NTU uses 4+4, AV-MNIST 5+3, MM-IMDB 2+4) */

#define MFAS_DT_F32 0
#define MFAS_DT_BF16 1
#define MFAS_DT_F16 2

/* The `args` fields the path reads (models/search/ntu_searchable.py:28-84,200-292) plus the Adam
 * constants fixed at ntu_searchable.py:65 (weight_decay=1e-4, torch defaults otherwise). */
typedef struct mfas_hyper {
    int32_t R;          /* args.inner_representation_size */
    int32_t C;          /* args.num_outputs */
    int32_t B;          /* args.batchsize */
    int32_t bn;         /* args.batchnorm */
    int32_t alphas;     /* args.alphas */
    int32_t multitask;  /* args.multitask: argmax over central+visual+skeleton logits */
    double drpt;        /* args.drpt (<=1e-10: no Dropout module) */
    double wd, beta1, beta2, adam_eps, bn_eps, bn_momentum;
    int32_t s_sizes[MFAS_MAX_TAPS]; /* first-modality tap widths (skeleton: ntu_searchable.py:291; audio:
                                     * avmnist_searchable.py:290-292); 0 = unused slot.  Any width >= 1: tables store
                                     * each row padded with zeros to a multiple of 16 elements. */
    int32_t v_sizes[MFAS_MAX_TAPS]; /* second-modality tap widths (visual: ntu_searchable.py:292) */
    /* Head loss / dev metric.  0: CrossEntropyLoss + top-1 accuracy (NTU, ntu_searchable.py:31).
     * 1: multi-label WeightedCrossEntropyWithLogits (models/central/mm_imdb.py:655-673) + F1 'samples' of
     *    sigmoid(logits) > f1_threshold (models/search/train_searchable/mmimdb.py:86,105). */
    int32_t loss_mode;
    int32_t allow_plain_cell; /* 1: [Linear, nl] cells (no BN, no Dropout) are legal (avmnist_searchable.py:276-285); the
                               * NTU searchable leaves that case undefined (ntu_searchable.py:274-284) */
    double f1_threshold; /* th_fscore, mmimdb.py:16 (0.3) */
    int32_t tap_bits;    /* hint: element size of the feature tables this population will train on (16: bf16 / f16 taps, the
                          * layout north_star prescribes; 32: f32; 0: unknown = any).  With 16 the engine may size resident
                          * feature units for 16-bit staging (up to 1024 columns each); training such a population on f32
                          * tables is refused with MFAS_EINVAL. */
    int32_t order_per_candidate; /* 0: one sample order per epoch shared by the whole population (lockstep: `order` of
                          * mfas_population_train holds [epochs][N_train]); 1: every candidate walks its OWN permutations — the
                          * reference draws a fresh DataLoader(shuffle=True) order per candidate and epoch
                          * (models/searchable.py:248-250, train_searchable/ntu.py:35) — `order` then holds [K][epochs][N_train]. */
} mfas_hyper;

/* Pooled feature table = what Visual/Skeleton.forward + GlobalPooling2D hand to the fusion net
 * (models/central/ntu.py:35-50,129-183; ntu_searchable.py:211-225) for N samples, plus labels
 * (datasets/ntu.py:254). */
typedef struct mfas_table {
    const void* s[MFAS_MAX_TAPS]; /* (N, ceil16(width)) row-major, zero padded */
    const void* v[MFAS_MAX_TAPS];
    const float* vlogit; /* (N, C) unimodal logits for multitask, or NULL */
    const float* slogit;
    const int32_t* label; /* (N) in [0, C); loss_mode 0 */
    const float* multilabel; /* (N, C) 0/1 targets; loss_mode 1 (else NULL) */
    int64_t N;
    int32_t dtype; /* MFAS_DT_* of s[]/v[] */
    int32_t _pad;
} mfas_table;

/* Per-epoch statistics train_ntu_track_acc accumulates (train_searchable/ntu.py:72-79). */
typedef struct mfas_epoch_stats {
    double train_loss_sum; /* sum over samples of the CE loss (running_loss) */
    double dev_loss_sum;
    int64_t train_corrects; /* running_corrects (loss_mode 0) */
    int64_t dev_corrects;   /* loss_mode 0: correct predictions; loss_mode 1: round(2^32 * sum over samples of F1) */
} mfas_epoch_stats;

typedef struct mfas_population mfas_population; /* opaque; owns its device workspace */

const char* mfas_last_error(void);
int mfas_version(void);
/* "mfas-src-digest:<16 hex>": sha256 prefix of the sources (the csrc .hip / .hip.h files and this header) the loaded library
 * was compiled from; __graft_entry__.build() rebuilds when it differs from the tree and tests assert it matches. */
const char* mfas_source_digest(void);

/* Replaces the model/optimizer construction half of train_sampled_models' population loop
 * (ntu_searchable.py:38-72): K candidates, confs[k][cell][{ske_tap, vis_tap, nonlinearity}],
 * n_cells[k] rows used.  drop_seeds[k] seeds candidate k's dropout stream (NULL: 0..K-1).
 * chunk_cols: feature-column chunk per workgroup (0 = auto). */
int mfas_population_create(const mfas_hyper* hp, const int32_t* confs /* K*4*3 */,
                           const int32_t* n_cells /* K */, const uint32_t* drop_seeds, int32_t K,
                           int32_t device, void* hip_stream, int32_t chunk_cols,
                           mfas_population** out);
void mfas_population_destroy(mfas_population* pop);

/* Number of floats of candidate k's central parameters in reference state_dict order:
 * alphas.i.alpha_x (L), then per cell fusion_layers.i.0.weight (R x K_i row-major), .0.bias (R),
 * [.2.weight, .2.bias, .2.running_mean, .2.running_var (R each) if bn], then
 * central_classifier.weight (C x R), central_classifier.bias (C)   (ntu_searchable.py:191-200). */
int64_t mfas_population_param_count(const mfas_population* pop, int32_t k);

/* Load / read back candidate k's parameters (device float buffers in the order above).
 * plane 0 = parameters, 1 = Adam exp_avg, 2 = Adam exp_avg_sq.  set() also zeroes the Adam state
 * and the step counter (a fresh torch.optim.Adam, ntu_searchable.py:65). */
int mfas_population_set_params(mfas_population* pop, int32_t k, const float* flat);
int mfas_population_get_params(mfas_population* pop, int32_t k, int32_t plane, float* flat);

/* Device-side PyTorch-default-shaped init (U(+-1/sqrt(fan_in)), BN gamma=1/beta=0/rm=0/rv=1,
 * alpha = 0.1*noise) from the hash generator documented in oracle/np_oracle.py:init_params. */
int mfas_population_init(mfas_population* pop, const uint32_t* seeds /* K, host */);

/* (new, round 4) The reference's OWN initial parameters, drawn on the device: what `searchable_type(args, conf)` — the model
 * construction at ntu_searchable.py:44 (nn.Linear.reset_parameters per cell and for the classifier, then alpha ~ N(mean, std),
 * ntu_searchable.py:202-204) — draws from torch's CPU generator after torch.manual_seed(seeds[k]): at::mt19937 and
 * uniform_real_distribution<float> run per candidate on the GPU, bit for bit the numbers of the host path (the Python mirror checks
 * that once per process against torch itself and falls back to mfas_population_set_params otherwise).
 * seeds: HOST uint64[K].  bounds: HOST float[K][2 * (MFAS_MAX_CELLS + 1)] = per cell {weight bound, bias bound} (unused cells: any),
 * then the classifier's {weight bound, bias bound}: Tensor.uniform_(-bound, bound).  BatchNorm starts at its defaults; Adam state
 * is zeroed (as mfas_population_set_params does). */
int mfas_population_init_torch_streams(mfas_population* pop, const uint64_t* seeds, const float* bounds, double alpha_mean,
                                       double alpha_std);

/* Replaces train_ntu_track_acc (train_searchable/ntu.py:14-89) for the whole population in lockstep:
 * for each epoch: train over `train` in the given sample order (order: device int32 [epochs][N_train],
 * NULL = sequential), then evaluate `dev`.  step_scalars: HOST float2 per train step
 * {lr_t/(1-beta1^t), sqrt(1-beta2^t)} (scheduler.py:25-46 + torch Adam bias corrections).
 * max_steps >= 0 stops after that many train steps of the first epoch (debug/known-answer tests;
 * dev evaluation is skipped).  stats: HOST [K][epochs].  status: HOST [K], 1 = non-finite loss seen.
 * snapshot_best != 0 keeps the best-dev-epoch parameters and restores them at the end (:82-86).
 * Every call starts from zeroed Adam moments and step count (a freshly constructed optimizer). */
int mfas_population_train(mfas_population* pop, const mfas_table* train, const mfas_table* dev,
                          const int32_t* order, const float* step_scalars, int32_t epochs,
                          int64_t max_steps, int32_t snapshot_best, mfas_epoch_stats* stats,
                          int32_t* status);

/* Replaces Searchable_Skeleton_Image_Net.forward in eval mode (ntu_searchable.py:206-247) for
 * candidate k on rows [row0, row0+nrows) of a table: writes logits (nrows, C) (device, row-major);
 * and test_ntu_track_acc (train_searchable/ntu.py:92-125) when corrects != NULL (HOST int64). */
int mfas_population_forward(mfas_population* pop, int32_t k, const mfas_table* tab, int64_t row0,
                            int64_t nrows, float* logits, int64_t* corrects);

/* Replaces Searchable_Skeleton_Image_Net.forward in TRAIN mode (ntu_searchable.py:206-247 with model.train(True)) for
 * candidate k on ONE batch = rows [row0, row0+nrows) of a table, nrows <= hyper.B: batch-statistics BatchNorm (the running
 * statistics are updated, as any train-mode forward does), dropout from the candidate's stream at `step_index`; writes
 * logits (nrows, C) (device).  No gradient is produced: the train loop (forward + backward + Adam) is mfas_population_train. */
int mfas_population_forward_train(mfas_population* pop, int32_t k, const mfas_table* tab, int64_t row0, int32_t nrows,
                                  int32_t step_index, float* logits);

/* (new, round 3) The autograd half of that forward: the gradients of an EXTERNAL loss of the same batch with respect to every
 * central parameter (loss.backward() on the logits of Searchable_Skeleton_Image_Net.forward under model.train(True),
 * /root/reference/models/search/ntu_searchable.py:206-247, for callers that write their own training loop).  The batch is run
 * again in train mode with the same dropout stream (`step_index`) and batch statistics, `dlogits` (DEVICE, nrows x C float32
 * = dL/dlogits) takes the place of the loss gradient, and each parameter's gradient is left in its Adam first-moment slot:
 * read it with mfas_population_get_params(pop, k, 1, flat).  Parameters are not changed.  Candidate k's Adam moment slots are
 * ZEROED at entry (the gradient is accumulated into the first-moment slot from 0, whatever the handle trained before) and are
 * scratch afterwards, like the BN running statistics: the optimizer state of candidate k does not survive this call — use a
 * population kept for the purpose (the Python mirror builds one per call). */
int mfas_population_backward(mfas_population* pop, int32_t k, const mfas_table* table, int64_t row0, int32_t nrows,
                             int32_t step_index, const float* dlogits);

/* Timing of the dominant kernel (fused cell sweep) over the last train() call, measured with HIP
 * events on the population's stream: number of launches, summed milliseconds, and the algorithmic
 * HBM bytes of one update+forward launch (24*P_tiles + feature bytes). */
int mfas_population_sweep_profile(const mfas_population* pop, int64_t* launches, double* total_ms,
                                  double* bytes_per_launch);
int mfas_population_set_profiling(mfas_population* pop, int32_t on);

/* (new) The step schedule this population was laid out for, so that a measurement can name the kernel it
 * timed: info[0] = 1 persistent step loop (k_president: resident units + resident chain, one launch per epoch) / 0 launch per phase (k_step / k_chain);
 * info[1] = feature units resident in registers; info[2] = their workgroups; info[3] = units per resident workgroup;
 * info[4] = 1 when the resident lean chain owns OUT/HEAD; info[5] = bit 0: lean chain (R <= 16), bits 8..15: compute units one
 * candidate's general chain runs on in the same-group launch (round 6, chain_split; 1 otherwise); info[6] = candidate groups of the
 * launch-per-phase schedule (2 = fused A/B launches, 1 = chain and sweep back to back, -1 = one launch per step holding the chain
 * AND the sweep of the same candidates, released cell by cell through per-cell flags); info[7] = candidates. */
int mfas_population_schedule(const mfas_population* pop, int32_t info[8]);

/* (new, round 3) The same decision WITHOUT creating a population — a pure query, nothing is allocated or launched: would
 * mfas_population_create(hp, confs, n_cells, ..., K, device, ..., chunk_cols) take the resident persistent schedule (every
 * chain and every feature unit resident on its own CU slot, W/m/v in registers, one launch per epoch)?  The host plans
 * resident ROUNDS with it (a share of a train_sampled_models call — /root/reference/models/search/ntu_searchable.py:38-94 trains
 * the configurations one after the other — that is too large for one resident population is trained as several).
 * info[0] = 1 resident persistent schedule / 0 launch per phase; info[1] = resident feature units; info[2] = their workgroups;
 * info[3] = units per resident workgroup; info[4] = feature-column chunk; info[5] = 1 lean chain (R <= 16, C <= 64, B <= 32);
 * info[6] = compute units of the device; info[7] = K. */
int mfas_population_plan(const mfas_hyper* hp, const int32_t* confs, const int32_t* n_cells, int32_t K, int32_t device,
                         int32_t chunk_cols, int32_t info[8]);

/* (new) Streaming ceiling of this device for the sweep's access pattern: three planes of `bytes_per_plane` are
 * read-modify-written in 1 KiB tiles with nontemporal 16 B/lane accesses and no compute; returns GB/s (read + write). */
int mfas_stream_probe(int64_t bytes_per_plane, int32_t iters, double* gb_per_s);

/* Replaces GlobalPooling2D.forward (models/auxiliary/aux_models.py:54-64; applied at ntu_searchable.py:224-225): mean over
 * the `inner` trailing elements of each of the rows = B*C contiguous rows of a backbone tap (B, C, ...) -> out[rows].
 * x / out are device pointers of dtype MFAS_DT_* ; f32 accumulation.  The step that builds an mfas_table from raw taps. */
int mfas_global_pool(const void* x, int32_t dtype, int64_t rows, int64_t inner, void* out, int32_t out_dtype,
                     void* hip_stream);

/* (new, round 4) Profiler ranges (roctx markers; no-ops when the marker library is not loadable): mfas_population_train opens one
 * range per call and one per epoch itself; the host side brackets each train_sampled_models call
 * (/root/reference/models/search/ntu_searchable.py:23-102) with these.  Always return MFAS_OK. */
int mfas_range_push(const char* name);
int mfas_range_pop(void);

/* (new, round 6) The library's environment switches (A/B and debugging aids; INTEGRATION.md lists them) as parsed from the CURRENT
 * environment: "name=value name=value ..." in declaration order, NUL-terminated, truncated to `cap`.  The library parses them in ONE
 * place, when a population is created or planned (the reference has no counterpart: its knobs are the argparse flags of
 * /root/reference/main_searchable_ntu.py:16-63, mirrored by mfas_hyper); a population keeps the set it was created under.  An empty
 * environment yields the defaults the test suites run ("hooks=0": the product library parses no test hook). */
int mfas_tuning_describe(char* buf, int32_t cap);

/* snapshot_best bookkeeping: a dev metric must EXCEED `threshold` to replace the kept parameters (best_acc = 0,
 * train_searchable/ntu.py:18; best_f1 = init_f1, train_searchable/mmimdb.py:18).  Default 0.  If no epoch exceeds it the
 * INITIAL parameters are restored, like the reference's best_model_sd (ntu.py:17,86). */
int mfas_population_set_best_threshold(mfas_population* pop, double threshold);

/* loss_mode 1: per-class positive weights of WeightedCrossEntropyWithLogits (HOST float[C]; default all 1). */
int mfas_population_set_pos_weight(mfas_population* pop, const float* pos_weight);

#ifdef __cplusplus
}
#endif
#endif /* MFAS_HIP_H */

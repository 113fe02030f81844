"""Parity at BASELINE configs[1] FULL size (conf 4, R=128, BN, B=16, N_train=10,000, N_dev=5,600, bf16-stored taps,
deterministic mode, 3 epochs) against the unchanged reference (goldens G13, G14).

At this size the reference is not reproducible against ITSELF: changing only the BLAS thread count moves its best dev
accuracy over 0.9830..0.9861 (G13: per-step losses differ by 1e-4 after ONE Adam step, whose update is sign-like, and by
1e-2 after ten), and a 1e-7 relative perturbation of the initial weight matrices moves its epoch-0 train loss over
1.94..2.08 and dev accuracy over 0.940..0.956 (G14).  A single trajectory can only be gated on that envelope;
the sharp statement is distributional: the reference, the oracle and the engine, run from the SAME perturbed starts, must
agree in the MEAN of every epoch statistic within the standard error (accuracies: + the north_star's 0.1 %).
The remedy for chaos is samples, not a wider gate: G14 holds 64 starts of the reference and the engine runs all 64
(3 s.e. = 0.05 % on the best dev accuracy, so the gate is +-0.15 %); G14m repeats the experiment in the SENSITIVE regime
(snr 0.10: dev accuracy 0.69 instead of a saturated 0.98); G18c is the same experiment WITH dropout 0.5 (the reference's dropout
modules swapped for the engine's masks, shuffled fixed order, bench.py's snr 0.12): 512 reference starts against 2,048 engine
starts, gate 3 s.e. + 0.1 % <= 0.2 %; and G15 is the bench workload itself with the reference's OWN dropout / shuffle streams
(E=10, snr 0.12) gated on the mean best dev accuracy over 4,096 engine seeds vs 1,024 reference seeds (round 6; 1,024 vs 256 in
round 5): the reference's own seed sigma there is 0.87 %, so that gate is 3 s.e. + 0.1 % = +-0.19 % (asserted <= 0.205 % in the test).  G19a / G19b are the same two experiments in the SEARCH-DEFAULT regime (R=16, no batchnorm, B=20: BASELINE
configs[2] at full size) — pointwise +-0.1 % and, round 6, +-0.2 % on the population mean (128 reference calls)."""
import numpy as np
import pytest

from oracle import np_oracle as O
from tests.helpers import CONFS, engine_hyper, etas_for, golden

TOL = 0.001      # +-0.1 % top-1 (north_star) ON TOP of 3 standard errors of the reference's own seed-to-seed spread: the gates below are
                 # +-0.15 % (G14, 64 reference starts), +-0.2 % (G18c), +-0.29 % (G15) in effect, and say so where they are applied
CONF = np.array(CONFS["c4"])
HP = O.Hyper(R=128, B=16, bn=True, drpt=0.0, epochs=3)
NAMES = [f"{ph} {q} e{e}" for e in range(3) for ph in ("train", "dev") for q in ("loss", "acc")] + ["best dev acc"]


VARIANTS = {"saturated": ("g14_fullsize_envelope.npz", 0.15), "midrange": ("g14m_fullsize_midrange.npz", 0.10)}
_TABLES = {}


def tables_for(variant):
    snr = VARIANTS[variant][1]
    if snr not in _TABLES:
        _TABLES[snr] = (O.synth_table(10000, 1, snr=snr, quant="bf16"), O.synth_table(5600, 2, snr=snr, quant="bf16"))
    return _TABLES[snr]


def g14_stats(variant="saturated"):
    g = golden(VARIANTS[variant][0])
    assert float(g["snr"][0]) == VARIANTS[variant][1]
    H = g["hist"]                                   # [trial][2*epoch + phase] = (phase, loss, acc)
    return np.concatenate([H[:, :, 1:].reshape(len(H), -1), g["best_acc"][:, None]], 1)   # [trial][13], order = NAMES


def row_of(train_loss, train_acc, dev_loss, dev_acc, best):
    r = []
    for e in range(3):
        r += [train_loss[e], train_acc[e], dev_loss[e], dev_acc[e]]
    return r + [best]


def check_ensemble(mine, ref):
    """mine, ref: [trial][13].  Means within 3 standard errors (+0.1 % on accuracies), spreads within x3, every sample
    within 5.5 sigma of the reference's mean (one straggler per 64 samples)."""
    for j, nm in enumerate(NAMES):
        sr, sm = ref[:, j].std(ddof=1), mine[:, j].std(ddof=1)
        se = np.sqrt(sr ** 2 / len(ref) + sm ** 2 / len(mine))
        budget = 3.0 * se + (TOL if "acc" in nm else 0.0)
        assert abs(mine[:, j].mean() - ref[:, j].mean()) <= budget, (nm, mine[:, j].mean(), ref[:, j].mean(), se)
        assert sm <= 3.0 * sr + 1e-4, (nm, "spread", sm, sr)
        # every sample within 5.5 sigma of the reference's distribution (64 samples: a range-based envelope is an extreme-value
        # statistic and fails by construction once there are enough of them)
        dev = np.abs(mine[:, j] - ref[:, j].mean())
        far = dev > 5.5 * sr + (TOL if "acc" in nm else 1e-4)
        # (the epoch right after a warm restart of the cosine schedule — epoch 1 here — has a heavier tail than 64 reference
        # samples resolve: at most ONE straggler per statistic, and never beyond 12 sigma)
        assert far.sum() <= (max(1, len(mine) // 64) if len(mine) >= 32 else 0), (nm, mine[:, j][far], ref[:, j].mean(), sr)
        # (G18c: 3 of the reference's own 512 starts sit beyond 4 sigma in the dev statistics of that epoch, one at 9.4 — a tail of that
        #  weight puts a sample or two of 2048 beyond 12 sigma: one allowed per 1024 samples, none in smaller ensembles)
        assert (dev > 12.0 * sr + (TOL if "acc" in nm else 1e-4)).sum() <= len(mine) // 1024, (nm, mine[:, j][dev.argmax()], ref[:, j].mean(), sr)


def test_reference_is_not_reproducible_against_itself():
    """G13 (thread counts) and G14 (1e-7 weight perturbations): the envelope the gates below are built from."""
    g13 = golden("g13_fullsize.npz")
    assert np.ptp(g13["best_acc"]) > 0.002                      # 4 thread counts: 0.9830 .. 0.9861
    r = g14_stats()
    assert r.shape == (64, 13)
    assert np.ptp(r[:, 3]) > 0.005 and np.ptp(r[:, 12]) > 0.002  # epoch-0 dev acc, best dev acc
    np.testing.assert_allclose(r[0, :12].reshape(6, 2), g13["hist"][2][:, 1:], atol=2e-3)   # trial 0 == G13's 4-thread run
    # with 64 + 64 samples the gate on the best dev accuracy is 3 s.e. + 0.1 % < 0.2 %
    assert 3.0 * r[:, 12].std(ddof=1) * np.sqrt(2.0 / 64) + TOL < 0.002
    m = g14_stats("midrange")
    assert m.shape == (64, 13) and 0.5 < m[:, 12].mean() < 0.8    # the sensitive regime


@pytest.mark.parametrize("variant,ntrial", [("saturated", 3), ("midrange", 2)])
def test_oracle_fullsize_vs_reference(variant, ntrial):
    ttr, tdv = tables_for(variant)
    rows = []
    for trial in range(ntrial):
        hist = []
        best = O.train_candidate(CONF, HP, O.perturb_params(O.init_params(CONF, HP, 77), trial), ttr, tdv, history=hist)
        rows.append(row_of([h["train_loss"] for h in hist], [h["train_acc"] for h in hist],
                           [h["dev_loss"] for h in hist], [h["dev_acc"] for h in hist], best))
    check_ensemble(np.array(rows), g14_stats(variant))


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["saturated", "midrange"])
def test_engine_fullsize_vs_reference(variant):
    torch = pytest.importorskip("torch")
    import mfas_amd as M
    ttr, tdv = tables_for(variant)
    dev = torch.device("cuda:0")
    ta, tb = M.FeatureTable.from_numpy(ttr, dev, torch.bfloat16), M.FeatureTable.from_numpy(tdv, dev, torch.bfloat16)
    NT = 64
    # all starts train as ONE lock-step population (each candidate is independent of its neighbours)
    pop = M.Population(engine_hyper(HP), [CONF] * NT, dev)
    for trial in range(NT):
        pop.set_state_dict(trial, O.perturb_params(O.init_params(CONF, HP, 77), trial))
    stats, status = pop.train(ta, tb, 3, etas_for(HP, 10000))
    assert not status.any()
    rows = [row_of(s["train_loss_sum"] / 10000, s["train_corrects"] / 10000, s["dev_loss_sum"] / 5600,
                   s["dev_corrects"] / 5600, M.best_dev_accuracy(s, 5600)) for s in stats]
    pop.close()
    check_ensemble(np.array(rows), g14_stats(variant))


# ------------------------------------------------------------------ G18c: the same experiment WITH dropout (injected masks)
HP_DROP = O.Hyper(R=128, B=16, bn=True, drpt=0.5, epochs=3)
G18C_ENGINE_STARTS = 2048


def g18c():
    g = golden("g18c_dropout_envelope.npz")
    N, Nd, snr, R, B, E, bn, drpt, init_seed, drop_seed, order_seed = g["meta"]
    assert (int(N), int(Nd), int(R), int(B), int(E), int(bn), float(drpt)) == (10000, 5600, 128, 16, 3, 1, 0.5)
    assert float(snr) == 0.12                                       # bench.py's tables
    H = g["hist"]
    stats = np.concatenate([H[:, :, 1:].reshape(len(H), -1), g["best_acc"][:, None]], 1)
    rng = np.random.default_rng(int(order_seed))
    order = np.stack([rng.permutation(int(N)) for _ in range(int(E))])
    return stats, order, int(init_seed), int(drop_seed), float(snr)


def test_dropout_envelope_gate_is_tight():
    """The reference's own spread with dropout ON (same masks, same order, 1e-7 perturbed starts) and what it makes of the gate:
    3 standard errors of (reference starts vs G18C_ENGINE_STARTS engine starts) + the north_star's 0.1 % must stay <= 0.2 %
    top-1 on the returned quantity (best dev accuracy)."""
    r, order, _, _, _ = g18c()
    assert len(r) >= 256 and order.shape == (3, 10000)
    assert 0.5 < r[:, 12].mean() < 0.9                              # not saturated
    s = r[:, 12].std(ddof=1)
    assert 3.0 * s * np.sqrt(1.0 / len(r) + 1.0 / G18C_ENGINE_STARTS) + TOL <= 0.002, (s, len(r))


def test_oracle_fullsize_dropout_vs_reference():
    r, order, init_seed, drop_seed, snr = g18c()
    ttr, tdv = O.synth_table(10000, 1, snr=snr, quant="bf16"), O.synth_table(5600, 2, snr=snr, quant="bf16")
    rows = []
    for trial in range(2):
        hist = []
        best = O.train_candidate(CONF, HP_DROP, O.perturb_params(O.init_params(CONF, HP_DROP, init_seed), trial), ttr, tdv,
                                 order=order, seed=drop_seed, history=hist)
        rows.append(row_of([h["train_loss"] for h in hist], [h["train_acc"] for h in hist],
                           [h["dev_loss"] for h in hist], [h["dev_acc"] for h in hist], best))
    check_ensemble(np.array(rows), r)


@pytest.mark.gpu
def test_engine_fullsize_dropout_vs_reference():
    """G18c on the engine: 2048 starts (16 lock-step populations of 128), every candidate with the reference's masks (same
    drop seed) and the reference's order.  This is the +-0.2 % gate on the path bench.py and the search actually run."""
    torch = pytest.importorskip("torch")
    import mfas_amd as M
    r, order, init_seed, drop_seed, snr = g18c()
    ttr, tdv = O.synth_table(10000, 1, snr=snr, quant="bf16"), O.synth_table(5600, 2, snr=snr, quant="bf16")
    dev = torch.device("cuda:0")
    ta, tb = M.FeatureTable.from_numpy(ttr, dev, torch.bfloat16), M.FeatureTable.from_numpy(tdv, dev, torch.bfloat16)
    dorder = torch.from_numpy(order.astype(np.int32)).to(dev)
    base = O.init_params(CONF, HP_DROP, init_seed)
    rows = []
    for grp in range(G18C_ENGINE_STARTS // 128):
        pop = M.Population(engine_hyper(HP_DROP), [CONF] * 128, dev, drop_seeds=[drop_seed] * 128)
        for j in range(128):
            pop.set_state_dict(j, O.perturb_params(base, 128 * grp + j))
        stats, status = pop.train(ta, tb, 3, etas_for(HP_DROP, 10000), order=dorder)
        assert not status.any()
        rows += [row_of(s["train_loss_sum"] / 10000, s["train_corrects"] / 10000, s["dev_loss_sum"] / 5600,
                        s["dev_corrects"] / 5600, M.best_dev_accuracy(s, 5600)) for s in stats]
        pop.close()
    rows = np.array(rows)
    check_ensemble(rows, r)
    se = np.sqrt(r[:, 12].std(ddof=1) ** 2 / len(r) + rows[:, 12].std(ddof=1) ** 2 / len(rows))
    assert abs(rows[:, 12].mean() - r[:, 12].mean()) <= 3.0 * se + TOL <= 0.002      # the gate that was asked for


@pytest.mark.gpu
def test_engine_bench_workload_vs_reference():
    """G15: BASELINE configs[1] exactly as bench.py runs it (conf 4, R=128, BN, drpt 0.5, shuffled, B=16, E=10,
    N=10,000/5,600, bf16-rounded taps at snr 0.12 — bench.py's tables) through the unchanged reference with its OWN dropout
    (Philox) and shuffle streams for 1,024 seeds (train_searchable/ntu.py:14-89).  The engine's streams are its own, so the gate
    is statistical: mean best dev accuracy over 4,096 engine seeds (64 populations of 64, each with its own epoch orders; the
    first 1,024 initial states are the reference's own) within 3 s.e. + 0.1 % = +-0.19 % (asserted <= 0.205 %; round 6: 1,024 reference seeds, 4,096 engine seeds); the per-epoch dev accuracies likewise.  (The
    pointwise pin of the dropout path is G18a/b/c, where the reference runs with the engine's masks.)"""
    torch = pytest.importorskip("torch")
    import mfas_amd as M
    g = golden("g15_bench_workload.npz")
    N, Nd, snr, R, B, E, bn, drpt = g["meta"]
    N, Nd, R, B, E = int(N), int(Nd), int(R), int(B), int(E)
    assert float(snr) == 0.12 and len(g["best_acc"]) >= 1000 and E == 10     # the bench workload, not a neighbour of it
    hp = O.Hyper(R=R, B=B, bn=bool(bn), drpt=float(drpt), epochs=E)
    ttr, tdv = O.synth_table(N, 1, snr=float(snr), quant="bf16"), O.synth_table(Nd, 2, snr=float(snr), quant="bf16")
    dev = torch.device("cuda:0")
    ta, tb = M.FeatureTable.from_numpy(ttr, dev, torch.bfloat16), M.FeatureTable.from_numpy(tdv, dev, torch.bfloat16)
    best, per_epoch = [], []
    for grp in range(64):          # round 6: 4,096 engine seeds (64 populations of 64; 1,024 before) against the extended reference fixture
        seeds = list(range(64 * grp, 64 * grp + 64))
        pop = M.Population(engine_hyper(hp), [CONF] * 64, dev, drop_seeds=[7000 + s for s in seeds])
        pop.init([3000 + 10 * s for s in seeds])          # == O.init_params(conf, hp, 3000 + 10 * seed): the reference's starts
        order = M.ntu_searchable.make_order(N, E, True, 900 + grp, dev)
        stats, status = pop.train(ta, tb, E, etas_for(hp, N), order=order)
        assert not status.any()
        best += [M.best_dev_accuracy(s, Nd) for s in stats]
        per_epoch += [s["dev_corrects"] / Nd for s in stats]
        pop.close()
    best, per_epoch = np.array(best), np.array(per_epoch)
    ref_best = g["best_acc"]
    ref_epoch = g["hist"][:, 1::2, 2]                      # [seed][epoch] dev accuracy as printed (4 decimals)
    assert 0.5 < ref_best.mean() < 0.9                     # not saturated

    def iqr(x):
        return float(np.subtract(*np.percentile(x, [75, 25])))

    outliers = []

    def gate(mine, ref, what):
        se = np.sqrt(ref.std(ddof=1) ** 2 / len(ref) + mine.std(ddof=1) ** 2 / len(mine))
        assert abs(mine.mean() - ref.mean()) <= 3.0 * se + TOL, (what, mine.mean(), ref.mean(), se)
        # spread: interquartile ranges (robust: right after a warm restart of the cosine schedule — epochs 1, 3, 7 — a
        # candidate's dev accuracy occasionally collapses for one epoch while its BN running statistics catch up with the
        # jump in the weights; 16 reference seeds cannot resolve a 1-in-64 tail, so such epochs are counted, not gated on sigma)
        assert iqr(mine) <= 2.5 * iqr(ref) + 2e-3, (what, "spread", iqr(mine), iqr(ref))
        far = np.abs(mine - ref.mean()) > 6.0 * ref.std(ddof=1) + TOL
        outliers.append((what, int(far.sum())))
        assert far.sum() <= 2 * (len(mine) // 64), (what, "outliers", mine[far])

    gate(best, ref_best, "best dev acc")
    se_best = np.sqrt(ref_best.std(ddof=1) ** 2 / len(ref_best) + best.std(ddof=1) ** 2 / len(best))
    print(f"G15: engine {best.mean():.5f} ({len(best)} seeds) reference {ref_best.mean():.5f} ({len(ref_best)} seeds) gate {3.0 * se_best + TOL:.5f}")
    # the gate that is claimed (reference sigma 0.90 %): round 6 bought reference seeds with CPU time — >= 1,000 of them make it 0.2 %
    assert 3.0 * se_best + TOL <= (0.00205 if len(ref_best) >= 1000 else 0.0023), (3.0 * se_best + TOL, len(ref_best))
    for e in range(E):
        gate(per_epoch[:, e], ref_epoch[:, e], f"dev acc epoch {e}")
    assert outliers[0][1] == 0                      # the returned quantity itself has no stragglers
    assert sum(n for _, n in outliers) <= 4 * (len(best) // 64), outliers


# ------------------------------------------------------------------ G19: the SEARCH-DEFAULT regime at full size (BASELINE configs[2])
# What _epnas actually issues (main_searchable_ntu.py:26-47,56; models/searchable.py:90,120): R=16, no batchnorm, drpt 0.5, B=20.
# G19b = ONE train_sampled_models call on bench.py's configs[2] population (the 16 np.random.seed(0) L=4 confs, E=10,
# N=10,000/5,600, bf16-rounded taps at snr 0.12) through the unchanged reference with its OWN dropout / shuffle streams, 128 seeds (64 in round 5).
# The reference's seed-to-seed spread of one candidate's best dev accuracy is 0.4 ... 3.3 % here (chance is 1.7 %, the best conf
# reaches 24 %), so a per-conf gate of 3 s.e. + 0.1 % is +-0.3 ... 1.6 % with 64 reference seeds and would need > 1,500 reference
# seeds (11 CPU-hours each conf) to reach 0.3 % everywhere; the quantity the driver's line carries, `small_pop.c2.mean_best_dev_acc`
# (the population mean of ONE call), has a spread of 0.29 % and IS gated at <= 0.2 % (round 6; asserted below).
G19_ENGINE_CALLS = 256


def g19b():
    g = golden("g19b_search_default_streams.npz")
    N, Nd, snr, R, B, E, bn, drpt = g["meta"]
    assert (int(N), int(Nd), int(R), int(B), int(E), int(bn), float(drpt)) == (10000, 5600, 16, 20, 10, 0, 0.5)
    assert float(snr) == 0.12                                       # bench.py's tables
    return g["best_acc"], g["dev_acc"], g["confs"]


def bench_c2_confs():
    """bench.py's `sampled_l4(16)`: np.random.seed(0), rows of get_possible_layer_configurations(0) (bench.py:361-364)."""
    from mfas_amd import ntu_searchable as NS
    np.random.seed(0)
    layer = NS.get_possible_layer_configurations(0)
    return [np.array([layer[i] for i in np.random.choice(len(layer), 4)]) for _ in range(16)]


def test_search_default_reference_fixture():
    """G19b is the workload bench.py calls configs[2] (same 16 confs, same sizes, same snr), holds >= 64 reference seeds, is in
    the regime the metric is sensitive in, and makes the gate on the population mean (what `small_pop.c2.mean_best_dev_acc`
    reports) at most 0.2 % top-1 with G19_ENGINE_CALLS engine calls."""
    best, dev_acc, confs = g19b()
    assert best.shape[0] >= 128 and best.shape[1] == 16 and dev_acc.shape == best.shape + (10,)
    assert np.array_equal(confs, np.array(bench_c2_confs()))
    np.testing.assert_allclose(dev_acc.max(axis=2), best, atol=6e-5)          # best = max over epochs (printed to 4 decimals)
    pm = best.mean(axis=1)                                                    # one call's population mean
    assert 0.08 < pm.mean() < 0.14 and best.mean(axis=0).max() > 0.2 and best.mean(axis=0).min() > 1.5 / 60
    s = pm.std(ddof=1)
    assert 3.0 * s * np.sqrt(1.0 / len(pm) + 1.0 / G19_ENGINE_CALLS) + TOL <= 0.002, (s, len(pm))       # round 6: 128 reference calls (64 before: 0.0024)


@pytest.mark.gpu
def test_engine_search_default_population_vs_reference():
    """BASELINE configs[2] at FULL size through the boundary the search calls — train_sampled_models with the engine's defaults
    (torch-stream initialisation, per-candidate shuffles, its own dropout stream), G19_ENGINE_CALLS calls of the 16-conf
    population under different torch seeds — against 128 calls of the unchanged reference (G19b).  Gates: the population mean of
    the best dev accuracy (the number bench.py reports for configs[2]) within 3 s.e. + 0.1 %, that gate <= 0.2 %; every single
    conf within 3 s.e. + 0.1 % of its own reference mean (0.3 ... 1.6 % by the reference's own spread) and spread within x2.5."""
    torch = pytest.importorskip("torch")
    from types import SimpleNamespace
    import mfas_amd as M
    ref_best, _, confs = g19b()
    dev = torch.device("cuda:0")
    ttr, tdv = O.synth_table(10000, 1, snr=0.12, quant="bf16"), O.synth_table(5600, 2, snr=0.12, quant="bf16")
    ta, tb = M.FeatureTable.from_numpy(ttr, dev, torch.bfloat16), M.FeatureTable.from_numpy(tdv, dev, torch.bfloat16)
    loaders = {"train": M.FeatureLoader(ta, 20, shuffle=True), "dev": M.FeatureLoader(tb, 20, shuffle=False)}
    args = SimpleNamespace(vid_len=(8, 32), num_outputs=60, drpt=0.5, inner_representation_size=16, batchnorm=False, alphas=False,
                           multitask=False, weightsharing=False, batchsize=20, eta_max=1e-3, eta_min=1e-6, Ti=1, Tm=2,
                           use_dataparallel=False, verbose=False, epochs=10)
    mine = []
    for call in range(G19_ENGINE_CALLS):
        torch.manual_seed(71000 + call)
        mine.append([float(a) for a in M.train_sampled_models([np.array(c) for c in confs], M.Searchable_Skeleton_Image_Net,
                                                               loaders, args, dev)])
    mine = np.array(mine)
    pm, pr = mine.mean(axis=1), ref_best.mean(axis=1)
    se = np.sqrt(pr.std(ddof=1) ** 2 / len(pr) + pm.std(ddof=1) ** 2 / len(pm))
    gate = 3.0 * se + TOL
    print(f"G19b population mean: engine {pm.mean():.5f} (sd {pm.std(ddof=1):.5f}, {len(pm)} calls) reference {pr.mean():.5f} "
          f"(sd {pr.std(ddof=1):.5f}, {len(pr)} calls) gate {gate:.5f}")
    assert gate <= 0.00205            # round 6: 3 s.e. + 0.1 % <= 0.2 % top-1 (128 reference calls x 256 engine calls; 0.24 % in round 5)
    assert abs(pm.mean() - pr.mean()) <= gate, (pm.mean(), pr.mean(), se)
    assert pm.std(ddof=1) <= 2.5 * pr.std(ddof=1) + 1e-4
    for j in range(16):
        a, r = mine[:, j], ref_best[:, j]
        sej = np.sqrt(r.std(ddof=1) ** 2 / len(r) + a.std(ddof=1) ** 2 / len(a))
        print(f"  conf {j}: engine {a.mean():.4f} reference {r.mean():.4f} gate {3.0 * sej + TOL:.4f}")
        assert abs(a.mean() - r.mean()) <= 3.0 * sej + TOL, (j, a.mean(), r.mean(), sej)
        assert a.std(ddof=1) <= 2.5 * r.std(ddof=1) + 2e-3, (j, a.std(ddof=1), r.std(ddof=1))


# ------------------------------------------------------------------ G19a: the search-default regime, masks and order injected
# The G18c experiment at R=16 / no batchnorm / B=20 (three confs of the configs[2] population, full size, 3 epochs): the reference
# run from 128 starts that differ by 1e-7 relative perturbations of the initial weight matrices, every start under the SAME
# (engine-generated) dropout masks and sample order.  Unlike the R=128 / batchnorm workload (G14 / G18c), this regime is NOT
# chaotic at that scale: all 128 reference trajectories print the SAME losses and accuracies (4 decimals = half a sample).
# So the pin is pointwise: oracle and engine, from the same start with the same masks and order, must reproduce the reference's
# per-epoch statistics — accuracies within +-0.1 % top-1 (the north_star's number, no standard-error allowance), losses to 1e-3.
HP_SEARCH = O.Hyper(R=16, B=20, bn=False, drpt=0.5, epochs=3)
SNAMES = [f"{ph} {q} e{e}" for e in range(3) for ph in ("train", "dev") for q in ("loss", "acc")]


def g19a():
    g = golden("g19a_search_default_envelope.npz")
    N, Nd, snr, R, B, E, bn, drpt, init_seed, drop_seed, order_seed = g["meta"]
    assert (int(N), int(Nd), int(R), int(B), int(E), int(bn), float(drpt)) == (10000, 5600, 16, 20, 3, 0, 0.5)
    assert float(snr) == 0.12
    H = g["hist"]                                                   # [trial][conf][2 * epoch + phase] = (phase, loss, acc)
    rng = np.random.default_rng(int(order_seed))
    order = np.stack([rng.permutation(int(N)) for _ in range(int(E))])
    return H[:, :, :, 1:].reshape(H.shape[0], H.shape[1], -1), g["best_acc"], g["confs"], order, int(init_seed), int(drop_seed)


def test_search_default_reference_is_reproducible_at_print_precision():
    """128 perturbed starts x 3 confs: zero spread in every printed statistic (so no statistical allowance is needed below), the
    confs are members of bench.py's configs[2] population, and at least one of them learns within the 3 epochs."""
    stats, best, confs, order, _, _ = g19a()
    assert stats.shape == (128, 3, 12) and order.shape == (3, 10000)
    assert np.ptp(stats, axis=0).max() == 0.0 and np.ptp(best, axis=0).max() == 0.0
    allc = np.array(bench_c2_confs())
    assert all(any(np.array_equal(c, a) for a in allc) for c in confs)
    assert best[0].max() > 0.08


def check_search_default_row(row, ref, what):
    for j, nm in enumerate(SNAMES):
        tol = TOL if "acc" in nm else 1e-3
        assert abs(row[j] - ref[j]) <= tol + 5e-5, (what, nm, row[j], ref[j])      # (+ the reference's print rounding)


def test_oracle_search_default_vs_reference():
    stats, _, confs, order, init_seed, drop_seed = g19a()
    ttr, tdv = O.synth_table(10000, 1, snr=0.12, quant="bf16"), O.synth_table(5600, 2, snr=0.12, quant="bf16")
    ci = 2                                                          # the conf that learns
    hist = []
    O.train_candidate(confs[ci], HP_SEARCH, O.init_params(confs[ci], HP_SEARCH, init_seed), ttr, tdv, order=order, seed=drop_seed,
                      history=hist)
    row = []
    for h in hist:
        row += [h["train_loss"], h["train_acc"], h["dev_loss"], h["dev_acc"]]
    check_search_default_row(row, stats[0, ci], "oracle")


@pytest.mark.gpu
def test_engine_search_default_vs_reference_pointwise():
    """Every conf of G19a on the engine from the reference's start (and from 7 perturbed ones: the engine must be as insensitive
    as the reference), the reference's masks (same drop seed) and order: per-epoch train / dev loss and accuracy against the
    reference's printed values, accuracies within +-0.1 % top-1."""
    torch = pytest.importorskip("torch")
    import mfas_amd as M
    stats, _, confs, order, init_seed, drop_seed = g19a()
    ttr, tdv = O.synth_table(10000, 1, snr=0.12, quant="bf16"), O.synth_table(5600, 2, snr=0.12, quant="bf16")
    dev = torch.device("cuda:0")
    ta, tb = M.FeatureTable.from_numpy(ttr, dev, torch.bfloat16), M.FeatureTable.from_numpy(tdv, dev, torch.bfloat16)
    dorder = torch.from_numpy(order.astype(np.int32)).to(dev)
    worst = 0.0
    for ci, conf in enumerate(confs):
        NT = 8
        pop = M.Population(engine_hyper(HP_SEARCH), [conf] * NT, dev, drop_seeds=[drop_seed] * NT)
        base = O.init_params(conf, HP_SEARCH, init_seed)
        for j in range(NT):
            pop.set_state_dict(j, O.perturb_params(base, j))
        st, status = pop.train(ta, tb, 3, etas_for(HP_SEARCH, 10000), order=dorder)
        assert not status.any()
        for j, s in enumerate(st):
            row = []
            for e in range(3):
                row += [s["train_loss_sum"][e] / 10000, s["train_corrects"][e] / 10000, s["dev_loss_sum"][e] / 5600, s["dev_corrects"][e] / 5600]
            check_search_default_row(row, stats[0, ci], f"engine conf {ci} start {j}")
            worst = max(worst, max(abs(row[k] - stats[0, ci][k]) for k in range(12) if "acc" in SNAMES[k]))
        pop.close()
    print(f"G19a: worst |accuracy - reference| over 3 confs x 8 starts x 6 statistics = {worst:.5f}")


@pytest.mark.gpu
def test_full_search_schedule_configs3_is_seeded_and_deterministic():
    """BASELINE configs[3] END TO END at full size (main_searchable_ntu.py --num_samples 50 --search_iterations 5 --max_fusions 4,
    R=16, B=20, drpt 0.5, no batchnorm, E=10, N = 10,000 / 5,600: models/searchable.py:48-137 — 20 train_sampled_models calls, 982
    candidates) through the engine with the seeded controller (surrogate on the CPU), twice: the call structure is the reference's,
    the first call (the 32 single-layer configurations, drawn before any accuracy exists) has its fixed digest, and the whole decision
    stream — every configuration list the controller asked for, which depends on every accuracy the engine returned — is the same
    in both runs bit for bit.  (The digest of the stream itself moves whenever the engine's rounding does; bench.py prints it as
    config.search_c3.decision_digest.)"""
    import torch
    import main_searchable_ntu as MS
    import mfas_amd as M
    from mfas_amd.search import NTUSearcher
    from mfas_amd.search.searcher import timed_search
    dev = torch.device("cuda:0")
    train = M.FeatureTable.synthetic(10000, 1, dev, torch.bfloat16, snr=0.12)
    devl = M.FeatureTable.synthetic(5600, 2, dev, torch.bfloat16, snr=0.12)
    sa = MS.parse_args(["--num_samples", "50", "--search_iterations", "5", "--max_fusions", "4", "--epochs", "10", "--no-verbose"])
    reps = []
    nthr = torch.get_num_threads()
    torch.set_num_threads(max(1, sa.controller_threads))      # (the 81k-parameter surrogate trains fastest on a few threads: bench.py does the same)
    try:
        for _ in range(2):
            _, rep = timed_search(NTUSearcher(sa, dev, {"train": train, "dev": devl}), seed=0)
            reps.append(rep)
    finally:
        torch.set_num_threads(nthr)
    a, b = reps
    assert a["candidates"] == 982 and a["calls"] == 20 and a["call_sizes"] == [32] + [50] * 19
    assert a["first_call_digest"] == "001d9faf4bfd802c"
    assert a["decision_digest"] == b["decision_digest"] and a["best_dev_acc"] == b["best_dev_acc"]
    assert 0.05 < a["best_dev_acc"] < 1.0 and a["cand_per_s"] > 50

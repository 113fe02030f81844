"""Parity at BASELINE configs[1] FULL size (conf 4, R=128, BN, B=16, N_train=10,000, N_dev=5,600, bf16-stored taps,
deterministic mode, 3 epochs) against the unchanged reference (goldens G13, G14).

At this size the reference is not reproducible against ITSELF: changing only the BLAS thread count moves its best dev
accuracy over 0.9830..0.9861 (G13: per-step losses differ by 1e-4 after ONE Adam step, whose update is sign-like, and by
1e-2 after ten), and a 1e-7 relative perturbation of the initial weight matrices moves its epoch-0 train loss over
1.94..2.08 and dev accuracy over 0.940..0.956 (G14, 16 starts).  A single trajectory can only be gated on that envelope;
the sharp statement is distributional: the reference, the oracle and the engine, run from the SAME perturbed starts, must
agree in the MEAN of every epoch statistic within the standard error (accuracies: + the north_star's 0.1 %)."""
import numpy as np
import pytest

from oracle import np_oracle as O
from tests.helpers import CONFS, engine_hyper, etas_for, golden

TOL = 0.001      # +-0.1 % top-1 (north_star)
CONF = np.array(CONFS["c4"])
HP = O.Hyper(R=128, B=16, bn=True, drpt=0.0, epochs=3)
NAMES = [f"{ph} {q} e{e}" for e in range(3) for ph in ("train", "dev") for q in ("loss", "acc")] + ["best dev acc"]


@pytest.fixture(scope="module")
def tables():
    return O.synth_table(10000, 1, snr=0.15, quant="bf16"), O.synth_table(5600, 2, snr=0.15, quant="bf16")


def g14_stats():
    g = golden("g14_fullsize_envelope.npz")
    H = g["hist"]                                   # [trial][2*epoch + phase] = (phase, loss, acc)
    return np.concatenate([H[:, :, 1:].reshape(len(H), -1), g["best_acc"][:, None]], 1)   # [trial][13], order = NAMES


def row_of(train_loss, train_acc, dev_loss, dev_acc, best):
    r = []
    for e in range(3):
        r += [train_loss[e], train_acc[e], dev_loss[e], dev_acc[e]]
    return r + [best]


def check_ensemble(mine, ref):
    """mine, ref: [trial][13].  Means within 3.5 standard errors (+0.1 % on accuracies), spreads within x3, every sample
    inside the reference's range widened by its own width."""
    for j, nm in enumerate(NAMES):
        sr, sm = ref[:, j].std(ddof=1), mine[:, j].std(ddof=1)
        se = np.sqrt(sr ** 2 / len(ref) + sm ** 2 / len(mine))
        budget = 3.5 * se + (TOL if "acc" in nm else 0.0)
        assert abs(mine[:, j].mean() - ref[:, j].mean()) <= budget, (nm, mine[:, j].mean(), ref[:, j].mean(), se)
        assert sm <= 3.0 * sr + 1e-4, (nm, "spread", sm, sr)
        lo, hi = ref[:, j].min(), ref[:, j].max()
        w = (hi - lo) + (TOL if "acc" in nm else 0.0)
        assert (mine[:, j] >= lo - w).all() and (mine[:, j] <= hi + w).all(), (nm, mine[:, j], lo, hi)


def test_reference_is_not_reproducible_against_itself():
    """G13 (thread counts) and G14 (1e-7 weight perturbations): the envelope the gates below are built from."""
    g13 = golden("g13_fullsize.npz")
    assert np.ptp(g13["best_acc"]) > 0.002                      # 4 thread counts: 0.9830 .. 0.9861
    r = g14_stats()
    assert r.shape == (16, 13)
    assert np.ptp(r[:, 3]) > 0.005 and np.ptp(r[:, 12]) > 0.002  # epoch-0 dev acc, best dev acc
    np.testing.assert_allclose(r[0, :12].reshape(6, 2), g13["hist"][2][:, 1:], atol=2e-3)   # trial 0 == G13's 4-thread run


def test_oracle_fullsize_vs_reference(tables):
    ttr, tdv = tables
    rows = []
    for trial in range(6):
        hist = []
        best = O.train_candidate(CONF, HP, O.perturb_params(O.init_params(CONF, HP, 77), trial), ttr, tdv, history=hist)
        rows.append(row_of([h["train_loss"] for h in hist], [h["train_acc"] for h in hist],
                           [h["dev_loss"] for h in hist], [h["dev_acc"] for h in hist], best))
    check_ensemble(np.array(rows), g14_stats())


@pytest.mark.gpu
def test_engine_fullsize_vs_reference(tables):
    torch = pytest.importorskip("torch")
    import mfas_amd as M
    ttr, tdv = tables
    dev = torch.device("cuda:0")
    ta, tb = M.FeatureTable.from_numpy(ttr, dev, torch.bfloat16), M.FeatureTable.from_numpy(tdv, dev, torch.bfloat16)
    NT = 16
    # all starts train as ONE lock-step population (each candidate is independent of its neighbours)
    pop = M.Population(engine_hyper(HP), [CONF] * NT, dev)
    for trial in range(NT):
        pop.set_state_dict(trial, O.perturb_params(O.init_params(CONF, HP, 77), trial))
    stats, status = pop.train(ta, tb, 3, etas_for(HP, 10000))
    assert not status.any()
    rows = [row_of(s["train_loss_sum"] / 10000, s["train_corrects"] / 10000, s["dev_loss_sum"] / 5600,
                   s["dev_corrects"] / 5600, M.best_dev_accuracy(s, 5600)) for s in stats]
    pop.close()
    check_ensemble(np.array(rows), g14_stats())

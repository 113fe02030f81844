"""Shared helpers for the GPU parity tests: oracle <-> engine plumbing."""
import os

import numpy as np

from oracle import np_oracle as O

CONFS = {
    "c4": [[3, 1, 1], [1, 3, 0], [1, 1, 1], [3, 3, 0]],
    "c0": [[2, 2, 0], [1, 0, 1], [3, 2, 0], [3, 1, 1]],
    "l1": [[0, 0, 0]],
    "l2": [[2, 3, 1], [0, 2, 2]],
    "l3": [[1, 0, 2], [3, 3, 1], [0, 1, 0]],
}
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def engine_hyper(ohp: "O.Hyper"):
    from mfas_amd import Hyper
    return Hyper(R=ohp.R, C=ohp.C, B=ohp.B, bn=ohp.bn, drpt=ohp.drpt, alphas=ohp.alphas,
                 multitask=ohp.multitask, wd=ohp.wd, beta1=ohp.beta1, beta2=ohp.beta2,
                 adam_eps=ohp.adam_eps, bn_eps=ohp.bn_eps, bn_momentum=ohp.bn_momentum,
                 s_sizes=tuple(ohp.s_sizes), v_sizes=tuple(ohp.v_sizes), loss_mode=ohp.loss_mode,
                 f1_threshold=ohp.f1_threshold, allow_plain_cell=ohp.allow_plain_cell)


def etas_for(ohp, n_train, epochs=None):
    epochs = ohp.epochs if epochs is None else epochs
    nb = -(-n_train // ohp.B)
    return O.eta_sequence(ohp.eta_max, ohp.eta_min, ohp.Ti, ohp.Tm, n_train / ohp.B, epochs * nb)


def rel_err(got, want):
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    scale = max(float(np.abs(want).max()), 1e-30)
    return float(np.abs(got - want).max()) / scale


def frac_bad(got, want, rtol, atol):
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    return float((np.abs(got - want) > atol + rtol * np.abs(want)).mean())


def oracle_steps(conf, ohp, params, train, nsteps, seed=0, order=None):
    """Run nsteps train steps of the oracle; returns (params, AdamState, losses)."""
    keys = O.trainable_keys(conf, ohp)
    st = O.AdamState()
    N = len(train["label"])
    nb = -(-N // ohp.B)
    etas = etas_for(ohp, N, epochs=-(-nsteps // nb))
    losses = []
    for g in range(nsteps):
        ep, bi = divmod(g, nb)
        perm = np.arange(N) if order is None else np.asarray(order[ep])
        idx = perm[bi * ohp.B:(bi + 1) * ohp.B]
        feats = {k: v[idx] for k, v in train.items() if k != "label"}
        logits, cache = O.forward(params, conf, ohp, feats, True, seed=seed, step=g)
        loss, dlog, _ = O.ce_loss(logits, train["label"][idx])
        grads = O.backward(params, ohp, cache, dlog)
        O.bn_update_running(params, ohp, cache)
        O.adam_step(params, grads, st, float(etas[g]), ohp, keys)
        losses.append(float(loss))
    return params, st, losses

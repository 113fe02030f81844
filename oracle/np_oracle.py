"""CPU oracle for the MFAS inner candidate-training path (numpy, float32).

TEST INFRASTRUCTURE ONLY.  Nothing under ``mfas_amd/`` may import this module; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it,
and there only as the checker / reported baseline, never as the thing measured or shipped.

It restates, with explicit forward / backward / Adam arithmetic, the algorithm of

* ``Searchable_Skeleton_Image_Net.forward``      /root/reference/models/search/ntu_searchable.py:206-247
* ``_create_fc_layers`` (cell = Linear -> nl -> [BN] -> [Dropout])   ntu_searchable.py:258-286
* ``AlphaScalarMultiplication.forward``          models/auxiliary/aux_models.py:103-111
* ``train_ntu_track_acc``                        models/search/train_searchable/ntu.py:14-89
* ``train_sampled_models``                       ntu_searchable.py:23-102
* ``LRCosineAnnealingScheduler``                 models/auxiliary/scheduler.py:12-46
* ``get_possible_layer_configurations``          ntu_searchable.py:105-119

The arithmetic itself lives in PyTorch in the reference (ATen addmm / batch_norm /
log_softmax+nll_loss / autograd / torch.optim.Adam single-tensor path; no version pin in the
reference tree).  Parity pin: golden vectors generated HERE by importing the unchanged
reference under torch 2.10 (tests/golden/make_golden.py -> tests/golden/*.npz); see
tests/test_oracle_golden.py.

Dropout: torch's Philox stream cannot be reproduced; this oracle and the HIP engine share the
counter-based mask ``dropout_keep`` below (bit-exact between the two); against the reference
the dropout path is compared statistically only.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

F32 = np.float32

S_SIZES = (128, 256, 1024, 512)     # ntu_searchable.py:291 (vid_len[1]*32 == 1024)
V_SIZES = (512, 1024, 2048, 2048)   # ntu_searchable.py:292


# --------------------------------------------------------------------------- hyper-parameters
@dataclass
class Hyper:
    """The ``args`` fields the path reads (ntu_searchable.py:28-84,200-292)."""
    R: int = 16                 # inner_representation_size
    C: int = 60                 # num_outputs
    B: int = 20                 # batchsize
    bn: bool = False            # --batchnorm
    drpt: float = 0.5
    alphas: bool = False
    multitask: bool = False
    eta_max: float = 1e-3
    eta_min: float = 1e-6
    Ti: float = 1
    Tm: float = 2
    epochs: int = 3
    wd: float = 1e-4            # ntu_searchable.py:65
    beta1: float = 0.9
    beta2: float = 0.999
    adam_eps: float = 1e-8
    bn_eps: float = 1e-5
    bn_momentum: float = 0.1
    s_sizes: Sequence[int] = S_SIZES
    v_sizes: Sequence[int] = V_SIZES
    # head loss / dev metric: 0 = CrossEntropy + top-1 (NTU); 1 = WeightedCrossEntropyWithLogits + F1-samples
    # (models/central/mm_imdb.py:655-673; models/search/train_searchable/mmimdb.py:15-137)
    loss_mode: int = 0
    f1_threshold: float = 0.3
    pos_weight: Optional[Sequence[float]] = None
    allow_plain_cell: bool = False   # AV-MNIST: [Linear, nl] without BN / Dropout (avmnist_searchable.py:276-285)

    @property
    def use_dropout(self) -> bool:
        return self.drpt > 1e-10

    def check(self):
        # ntu_searchable.py:274-284: with drpt<1e-10 and no batchnorm `op` is never assigned.
        if not self.use_dropout and not self.bn and not self.allow_plain_cell:
            raise ValueError("illegal cell variant: drpt<1e-10 without batchnorm "
                             "(reference raises UnboundLocalError, ntu_searchable.py:274-284)")


# --------------------------------------------------------------------------- scheduler (A9)
def eta_sequence(eta_max, eta_min, Ti, Tm, nbpe, n) -> np.ndarray:
    """LR used by train step 0..n-1 (scheduler.py:25-46; float64 like the reference)."""
    out = np.empty(n, np.float64)
    Tcur, counter = 0.0, 0.0
    Ti = Ti
    for i in range(n):
        Tcur = counter / nbpe
        counter = counter + 1.0
        eta = eta_min + 0.5 * (eta_max - eta_min) * (1 + np.cos(np.pi * Tcur / Ti))
        if eta <= eta_min + 1e-10:
            Tcur = 0
            Ti = Ti * Tm
            counter = 0
        out[i] = eta
    return out


def get_possible_layer_configurations(progression_index=0) -> List[List[int]]:
    """ntu_searchable.py:105-119 — (4,4,2) grid, non-linearity fastest."""
    return [[t, v, n] for t in range(4) for v in range(4) for n in range(2)]


# --------------------------------------------------------------------------- dropout mask
def _lowbias32(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16)
    x = (x * np.uint32(0x7FEB352D)).astype(np.uint32)
    x ^= x >> np.uint32(15)
    x = (x * np.uint32(0x846CA68B)).astype(np.uint32)
    x ^= x >> np.uint32(16)
    return x


def dropout_keep(seed: int, step: int, cell: int, B: int, R: int, p: float) -> np.ndarray:
    """Counter-based keep-mask shared (bit-exact) with the HIP engine.

    keep[b, r] = (hash(seed, step, cell, b*R + r) >> 8) >= floor(p * 2^24)
    """
    with np.errstate(over="ignore"):
        k0 = np.uint32((seed + 0x9E3779B9 * (step + 1)) & 0xFFFFFFFF)
        h0 = _lowbias32(np.array([k0], np.uint32))[0]
        idx = (np.arange(B * R, dtype=np.uint64) + np.uint64(cell) * np.uint64(0x7F4A7C15))
        idx = (idx & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        h = _lowbias32(idx ^ h0)
    thr = np.uint32(int(math.floor(p * (1 << 24))))
    return ((h >> np.uint32(8)) >= thr).reshape(B, R)


# --------------------------------------------------------------------------- parameters
def cell_in_features(conf, i, hp: Hyper) -> int:
    F = hp.s_sizes[int(conf[i][0])] + hp.v_sizes[int(conf[i][1])]
    return F + (hp.R if i > 0 else 0)


def hash_u01(seed: int, n: int, offset: int = 0) -> np.ndarray:
    """n uniforms in [0,1) with 24-bit resolution from a pure 32-bit integer hash (bit-reproducible
    on any platform; also implemented in the HIP engine for device-side init)."""
    with np.errstate(over="ignore"):
        h0 = _lowbias32(np.array([(seed * 0x9E3779B9 + 0x7F4A7C15) & 0xFFFFFFFF], np.uint32))[0]
        idx = ((np.arange(n, dtype=np.uint64) + np.uint64(offset)) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        h = _lowbias32(idx ^ h0)
    return ((h >> np.uint32(8)).astype(F32) * F32(1.0 / (1 << 24))).astype(F32)


def hash_noise(seed: int, n: int) -> np.ndarray:
    """Unit-variance, zero-mean Irwin-Hall(4) noise from 4 hash uniforms (adds only: exact in f32)."""
    u = hash_u01(seed, 4 * n).reshape(n, 4)
    return (((u[:, 0] + u[:, 1]) + (u[:, 2] + u[:, 3]) - F32(2.0)) * F32(math.sqrt(3.0))).astype(F32)


def _uniform_pm(seed, shape, bound):
    n = int(np.prod(shape))
    return ((hash_u01(seed, n) * F32(2.0) - F32(1.0)) * F32(bound)).astype(F32).reshape(shape)


def param_seed(seed: int, slot: int) -> int:
    return (seed * 1000003 + slot * 7919 + 17) & 0x7FFFFFFF


def init_params(conf, hp: Hyper, seed: int, perturb_bn: bool = False) -> Dict[str, np.ndarray]:
    """PyTorch-default-shaped init (U(+-1/sqrt(fan_in)) for Linear W and b, BN gamma=1 beta=0
    rm=0 rv=1; alpha_x = 0.1*noise, cf. N(0,0.1) at ntu_searchable.py:202-204), drawn from the hash
    generator so that tests, fixtures and the HIP device-side init agree bit-for-bit.
    Keys follow the reference state_dict.  perturb_bn: non-trivial BN affine/running stats (tests)."""
    p: Dict[str, np.ndarray] = {}
    L = len(conf)
    for i in range(L):
        p[f"alphas.{i}.alpha_x"] = (hash_noise(param_seed(seed, 40 + i), 1) * F32(0.1)).astype(F32) \
            if hp.alphas else np.zeros(1, F32)
    for i in range(L):
        K = cell_in_features(conf, i, hp)
        bound = 1.0 / math.sqrt(K)
        p[f"fusion_layers.{i}.0.weight"] = _uniform_pm(param_seed(seed, 2 * i), (hp.R, K), bound)
        p[f"fusion_layers.{i}.0.bias"] = _uniform_pm(param_seed(seed, 2 * i + 1), (hp.R,), bound)
        if hp.bn:
            if perturb_bn:
                u = hash_u01(param_seed(seed, 20 + i), 4 * hp.R).reshape(4, hp.R)
                p[f"fusion_layers.{i}.2.weight"] = (F32(0.5) + u[0]).astype(F32)
                p[f"fusion_layers.{i}.2.bias"] = (F32(0.4) * u[1] - F32(0.2)).astype(F32)
                p[f"fusion_layers.{i}.2.running_mean"] = (F32(0.5) * u[2]).astype(F32)
                p[f"fusion_layers.{i}.2.running_var"] = (F32(0.5) + u[3]).astype(F32)
            else:
                p[f"fusion_layers.{i}.2.weight"] = np.ones(hp.R, F32)
                p[f"fusion_layers.{i}.2.bias"] = np.zeros(hp.R, F32)
                p[f"fusion_layers.{i}.2.running_mean"] = np.zeros(hp.R, F32)
                p[f"fusion_layers.{i}.2.running_var"] = np.ones(hp.R, F32)
    bound = 1.0 / math.sqrt(hp.R)
    p["central_classifier.weight"] = _uniform_pm(param_seed(seed, 10), (hp.C, hp.R), bound)
    p["central_classifier.bias"] = _uniform_pm(param_seed(seed, 11), (hp.C,), bound)
    return p


def perturb_params(p: Dict[str, np.ndarray], trial: int, rel: float = 1e-7) -> Dict[str, np.ndarray]:
    """Round-off-sized multiplicative noise on the weight matrices (trial 0 = unchanged).  Used to sample the
    reproducibility envelope of a long trajectory: the reference, this oracle and the engine are each run from the SAME
    perturbed starts (golden G14, tests/test_fullsize.py)."""
    if trial == 0:
        return p
    rng = np.random.default_rng(1000 + trial)
    q = dict(p)
    for k in sorted(p):
        if p[k].dtype == F32 and p[k].ndim == 2:
            q[k] = (p[k] * (1.0 + rel * rng.standard_normal(p[k].shape))).astype(F32)
    return q


def trainable_keys(conf, hp: Hyper) -> List[str]:
    """central_params() membership/order (ntu_searchable.py:249-256).  alphas are in the optimizer
    even when unused; then their grad is None and Adam skips them."""
    keys = []
    L = len(conf)
    if hp.alphas:
        keys += [f"alphas.{i}.alpha_x" for i in range(L)]
    for i in range(L):
        keys += [f"fusion_layers.{i}.0.weight", f"fusion_layers.{i}.0.bias"]
        if hp.bn:
            keys += [f"fusion_layers.{i}.2.weight", f"fusion_layers.{i}.2.bias"]
    keys += ["central_classifier.weight", "central_classifier.bias"]
    return keys


# --------------------------------------------------------------------------- forward / backward
def _sigmoid(x):
    return (F32(1.0) / (F32(1.0) + np.exp(-x, dtype=F32))).astype(F32)


def _act(y, nl):
    if nl == 0:
        return np.maximum(y, F32(0))
    if nl == 1:
        return _sigmoid(y)
    if nl == 2:
        return np.where(y > 0, y, F32(0.01) * y).astype(F32)
    raise ValueError(nl)


def forward(params, conf, hp: Hyper, feats, train: bool, seed=0, step=0, masks=None):
    """ntu_searchable.py:206-247 on already-pooled taps.  feats: dict s0..s3, v0..v3 (B,width) f32,
    optionally vlogit/slogit.  Returns (logits, cache)."""
    L = len(conf)
    cache = {"cells": [], "conf": conf}
    out = None
    for i in range(L):
        s = feats[f"s{int(conf[i][0])}"].astype(F32, copy=False)
        v = feats[f"v{int(conf[i][1])}"].astype(F32, copy=False)
        c = {}
        if hp.alphas:                                   # aux_models.py:103-111
            sg = _sigmoid(params[f"alphas.{i}.alpha_x"])[0]
            c["sg"] = sg
            c["s_raw"], c["v_raw"] = s, v
            s = s * sg
            v = v * (F32(1.0) - sg)
        x = np.concatenate([s, v] if i == 0 else [s, v, out], axis=1)   # :235-240 [ske|vis|prev]
        W = params[f"fusion_layers.{i}.0.weight"]
        b = params[f"fusion_layers.{i}.0.bias"]
        y = (x @ W.T + b).astype(F32)
        nl = int(conf[i][2])
        a = _act(y, nl)
        c.update(x=x, y=y, a=a, nl=nl)
        z = a
        if hp.bn:
            g = params[f"fusion_layers.{i}.2.weight"]
            be = params[f"fusion_layers.{i}.2.bias"]
            if train:
                mu = a.mean(axis=0, dtype=F32)
                var = ((a - mu) ** 2).mean(axis=0, dtype=F32)           # biased
                rstd = (F32(1.0) / np.sqrt(var + F32(hp.bn_eps))).astype(F32)
                xhat = ((a - mu) * rstd).astype(F32)
                n = a.shape[0]
                c.update(mu=mu, var=var, rstd=rstd, xhat=xhat, n=n)
            else:
                rm = params[f"fusion_layers.{i}.2.running_mean"]
                rv = params[f"fusion_layers.{i}.2.running_var"]
                xhat = ((a - rm) / np.sqrt(rv + F32(hp.bn_eps))).astype(F32)
            z = (xhat * g + be).astype(F32)
        if hp.use_dropout and train:
            if masks is not None:
                keep = masks[i]
            else:
                keep = dropout_keep(seed, step, i, z.shape[0], hp.R, hp.drpt)
            scale = F32(1.0 / (1.0 - hp.drpt))
            c.update(keep=keep, scale=scale)
            z = np.where(keep, z * scale, F32(0)).astype(F32)
        out = z
        cache["cells"].append(c)
    Wc = params["central_classifier.weight"]
    bc = params["central_classifier.bias"]
    logits = (out @ Wc.T + bc).astype(F32)
    cache["out"] = out
    return logits, cache


def ce_loss(logits, labels):
    """CrossEntropyLoss(mean) + argmax (train_searchable/ntu.py:54-58).  Returns
    (loss, dlogits, preds)."""
    B = logits.shape[0]
    mx = logits.max(axis=1, keepdims=True)
    ex = np.exp(logits - mx, dtype=F32)
    se = ex.sum(axis=1, keepdims=True, dtype=F32)
    logp = (logits - mx - np.log(se, dtype=F32)).astype(F32)
    loss = F32(-logp[np.arange(B), labels].mean(dtype=F32))
    sm = (ex / se).astype(F32)
    d = sm.copy()
    d[np.arange(B), labels] -= F32(1.0)
    d = (d / F32(B)).astype(F32)
    return loss, d, predict(logits)


def bce_loss(logits, z, w):
    """WeightedCrossEntropyWithLogits (models/central/mm_imdb.py:655-673): mean over batch and classes of
    w_c*z*(-log s) + (1-z)*(-log(1-s)), s = sigmoid(logits).  Returns (loss, dlogits)."""
    B, C = logits.shape
    sg = _sigmoid(logits.astype(F32))
    z = z.astype(F32)
    w = np.asarray(w, F32)
    L = (w * z * -np.log(sg, dtype=F32) + (F32(1.0) - z) * -np.log(F32(1.0) - sg, dtype=F32)).astype(F32)
    loss = F32(L.mean(dtype=F32))
    d = ((-w * z * (F32(1.0) - sg) + (F32(1.0) - z) * sg) / F32(B * C)).astype(F32)
    return loss, d


def f1_samples_fixed(logits, z, th) -> int:
    """sum over samples of F1 (sklearn f1_score(average='samples'), 0 when prediction and target are both empty)
    of sigmoid(logits) > th, as 32.32 fixed point (the engine's order-independent accumulator)."""
    pr = _sigmoid(logits.astype(F32)) > F32(th)
    tr = z > 0.5
    tp = (pr & tr).sum(axis=1).astype(np.int64)
    den = (pr.sum(axis=1) + tr.sum(axis=1)).astype(np.int64)
    out = 0
    for t, d in zip(tp, den):
        if d > 0:
            out += ((2 * int(t)) << 32) // int(d)
    return out


def predict(logits):
    return np.argmax(logits, axis=1)        # first max on ties, like torch.max(dim) on CPU


def backward(params, hp: Hyper, cache, dlogits, train_bn_stats=True):
    """Gradients of every central parameter (autograd of the forward above)."""
    conf = cache["conf"]
    L = len(conf)
    grads: Dict[str, np.ndarray] = {}
    out = cache["out"]
    Wc = params["central_classifier.weight"]
    grads["central_classifier.weight"] = (dlogits.T @ out).astype(F32)
    grads["central_classifier.bias"] = dlogits.sum(axis=0, dtype=F32)
    d_o = (dlogits @ Wc).astype(F32)
    for i in range(L - 1, -1, -1):
        c = cache["cells"][i]
        d_z = d_o
        if "keep" in c:
            d_z = np.where(c["keep"], d_z * c["scale"], F32(0)).astype(F32)
        if hp.bn:
            g = params[f"fusion_layers.{i}.2.weight"]
            n = F32(c["n"])
            dgamma = (d_z * c["xhat"]).sum(axis=0, dtype=F32)
            dbeta = d_z.sum(axis=0, dtype=F32)
            grads[f"fusion_layers.{i}.2.weight"] = dgamma
            grads[f"fusion_layers.{i}.2.bias"] = dbeta
            d_a = ((g * c["rstd"]) * (d_z - dbeta / n - c["xhat"] * (dgamma / n))).astype(F32)
        else:
            d_a = d_z
        nl = c["nl"]
        if nl == 0:
            d_y = np.where(c["a"] <= 0, F32(0), d_a).astype(F32)     # threshold_backward: NaN activations pass the grad
        elif nl == 1:
            d_y = (d_a * (F32(1.0) - c["a"]) * c["a"]).astype(F32)
        else:
            d_y = np.where(c["y"] > 0, d_a, F32(0.01) * d_a).astype(F32)
        W = params[f"fusion_layers.{i}.0.weight"]
        grads[f"fusion_layers.{i}.0.weight"] = (d_y.T @ c["x"]).astype(F32)
        grads[f"fusion_layers.{i}.0.bias"] = d_y.sum(axis=0, dtype=F32)
        d_x = None
        if hp.alphas or i > 0:
            d_x = (d_y @ W).astype(F32)
        if hp.alphas:
            ns = c["s_raw"].shape[1]
            nv = c["v_raw"].shape[1]
            sg = c["sg"]
            dsg = (d_x[:, :ns] * c["s_raw"]).sum(dtype=F32) - (d_x[:, ns:ns + nv] * c["v_raw"]).sum(dtype=F32)
            grads[f"alphas.{i}.alpha_x"] = np.array([dsg * sg * (F32(1.0) - sg)], F32)
        if i > 0:
            d_o = d_x[:, -hp.R:]
    return grads


# --------------------------------------------------------------------------- Adam (A8)
@dataclass
class AdamState:
    m: Dict[str, np.ndarray] = field(default_factory=dict)
    v: Dict[str, np.ndarray] = field(default_factory=dict)
    t: int = 0


def adam_scalars(lr: float, t: int, hp: Hyper):
    """Per-step scalars as torch's single-tensor Adam forms them (python doubles, then cast to the
    tensor dtype when they meet a float32 tensor)."""
    bc1 = 1.0 - hp.beta1 ** t
    bc2 = 1.0 - hp.beta2 ** t
    step_size = lr / bc1
    bc2_sqrt = bc2 ** 0.5
    return F32(step_size), F32(bc2_sqrt)


def adam_step(params, grads, st: AdamState, lr: float, hp: Hyper, keys):
    """torch.optim.Adam(weight_decay=wd) single-tensor path: g += wd*p; m.lerp_(g, 1-b1);
    v.mul_(b2).addcmul_(g, g, 1-b2); denom = sqrt(v)/sqrt(bc2) + eps; p -= lr/bc1 * m/denom."""
    st.t += 1
    step_size, bc2_sqrt = adam_scalars(lr, st.t, hp)
    w1 = F32(1.0 - hp.beta1)
    b2 = F32(hp.beta2)
    w2 = F32(1.0 - hp.beta2)
    eps = F32(hp.adam_eps)
    wd = F32(hp.wd)
    for k in keys:
        if k not in grads:
            continue
        p = params[k]
        g = (grads[k] + wd * p).astype(F32)
        if k not in st.m:
            st.m[k] = np.zeros_like(p)
            st.v[k] = np.zeros_like(p)
        m = st.m[k]
        v = st.v[k]
        m += w1 * (g - m)
        v *= b2
        v += (w2 * g) * g
        denom = (np.sqrt(v) / bc2_sqrt + eps).astype(F32)
        p -= step_size * (m / denom)


def bn_update_running(params, hp: Hyper, cache):
    for i, c in enumerate(cache["cells"]):
        if "mu" not in c:
            continue
        n = c["n"]
        mom = F32(hp.bn_momentum)
        rm = params[f"fusion_layers.{i}.2.running_mean"]
        rv = params[f"fusion_layers.{i}.2.running_var"]
        unbiased = (c["var"] * F32(n / (n - 1.0))).astype(F32)
        rm += mom * (c["mu"] - rm)
        rv += mom * (unbiased - rv)


# --------------------------------------------------------------------------- training loop
def _batch(table, idx):
    return {k: v[idx] for k, v in table.items() if k != "label"}


def train_candidate(conf, hp: Hyper, params, train, dev, order=None, seed=0, etas=None,
                    history=None, on_step=None, restore_best=False):
    """train_ntu_track_acc (train_searchable/ntu.py:14-89) for one candidate on feature tables.

    order: (epochs, N_train) int array of sample indices (None = sequential, i.e. shuffle off).
    Returns best dev accuracy (float64 = corrects / N_dev, max over epochs, strict >, from 0).
    restore_best: leave `params` at the best epoch's weights — best_model_sd starts as a copy of the INITIAL state (:17), is
    replaced on every strict improvement (:82-84) and loaded back at the end (:86).
    """
    hp.check()
    conf = np.asarray(conf)
    N_tr = len(train["label"])
    N_dev = len(dev["label"])
    B = hp.B
    nb_tr = -(-N_tr // B)
    nb_dev = -(-N_dev // B)
    if etas is None:
        etas = eta_sequence(hp.eta_max, hp.eta_min, hp.Ti, hp.Tm, N_tr / B, hp.epochs * nb_tr)
    keys = trainable_keys(conf, hp)
    st = AdamState()
    best = 0.0
    gstep = 0
    best_sd = {k: v.copy() for k, v in params.items()} if restore_best else None
    if hp.loss_mode == 1:
        return _train_candidate_multilabel(conf, hp, params, train, dev, order, seed, etas, history, keys, st)
    for ep in range(hp.epochs):
        # ---- train phase
        run_loss, run_corr = 0.0, 0
        perm = np.arange(N_tr) if order is None else np.asarray(order[ep])
        for bi in range(nb_tr):
            idx = perm[bi * B:(bi + 1) * B]
            feats = _batch(train, idx)
            labels = train["label"][idx]
            logits, cache = forward(params, conf, hp, feats, True, seed=seed, step=gstep)
            loss, dlog, preds = ce_loss(logits, labels)
            if hp.multitask:    # train_searchable/ntu.py:60-61; unimodal CE terms carry no grad
                preds = predict(logits + feats["vlogit"] + feats["slogit"])
                loss = (loss + ce_loss(feats["vlogit"], labels)[0]) + ce_loss(feats["slogit"], labels)[0]
            grads = backward(params, hp, cache, dlog)
            bn_update_running(params, hp, cache)
            adam_step(params, grads, st, float(etas[gstep]), hp, keys)
            run_loss += float(loss) * len(idx)
            run_corr += int((preds == labels).sum())
            if on_step is not None:
                on_step(gstep, params, st, float(loss))
            gstep += 1
        tr_loss, tr_acc = run_loss / N_tr, run_corr / N_tr
        # ---- dev phase
        run_loss, run_corr = 0.0, 0
        margins = []
        for bi in range(nb_dev):
            idx = np.arange(bi * B, min((bi + 1) * B, N_dev))
            feats = _batch(dev, idx)
            labels = dev["label"][idx]
            logits, _ = forward(params, conf, hp, feats, False)
            loss, _, preds = ce_loss(logits, labels)
            dec = logits
            if hp.multitask:
                dec = logits + feats["vlogit"] + feats["slogit"]
                preds = predict(dec)
                loss = (loss + ce_loss(feats["vlogit"], labels)[0]) + ce_loss(feats["slogit"], labels)[0]
            run_loss += float(loss) * len(idx)
            run_corr += int((preds == labels).sum())
            if history is not None:      # decision margins: |score of the true class - best other score| per dev sample
                d = np.array(dec, np.float64)
                own = d[np.arange(len(idx)), labels].copy()
                d[np.arange(len(idx)), labels] = -np.inf
                margins.append(np.abs(own - d.max(1)))
        dev_acc = run_corr / N_dev
        if history is not None:
            history.append(dict(train_loss=tr_loss, train_acc=tr_acc,
                                dev_loss=run_loss / N_dev, dev_acc=dev_acc, dev_corrects=run_corr,
                                dev_margins=np.sort(np.concatenate(margins))[:8]))
        if dev_acc > best:
            best = dev_acc
            if restore_best:
                best_sd = {k: v.copy() for k, v in params.items()}
    if restore_best:
        for k, v in best_sd.items():
            params[k] = v
    return best


def _train_candidate_multilabel(conf, hp, params, train, dev, order, seed, etas, history, keys, st):
    """train_mmimdb_track_f1 (models/search/train_searchable/mmimdb.py:15-137) on feature tables: weighted BCE,
    dev metric F1-samples at th_fscore, best F1 with strict '>' from 0, NaN train loss ends the run."""
    N_tr, N_dev, B = len(train["multilabel"]), len(dev["multilabel"]), hp.B
    nb_tr, nb_dev = -(-N_tr // B), -(-N_dev // B)
    w = np.ones(hp.C, F32) if hp.pos_weight is None else np.asarray(hp.pos_weight, F32)
    skip = ("label", "multilabel")
    best, gstep = 0.0, 0
    for ep in range(hp.epochs):
        run_loss = 0.0
        perm = np.arange(N_tr) if order is None else np.asarray(order[ep])
        for bi in range(nb_tr):
            idx = perm[bi * B:(bi + 1) * B]
            feats = {k: v[idx] for k, v in train.items() if k not in skip}
            logits, cache = forward(params, conf, hp, feats, True, seed=seed, step=gstep)
            loss, dlog = bce_loss(logits, train["multilabel"][idx], w)
            grads = backward(params, hp, cache, dlog)
            bn_update_running(params, hp, cache)
            adam_step(params, grads, st, float(etas[gstep]), hp, keys)
            run_loss += float(loss) * len(idx)
            gstep += 1
        tr_loss = run_loss / N_tr
        run_loss, f1fx = 0.0, 0
        for bi in range(nb_dev):
            idx = np.arange(bi * B, min((bi + 1) * B, N_dev))
            feats = {k: v[idx] for k, v in dev.items() if k not in skip}
            logits, _ = forward(params, conf, hp, feats, False)
            loss, _ = bce_loss(logits, dev["multilabel"][idx], w)
            run_loss += float(loss) * len(idx)
            f1fx += f1_samples_fixed(logits, dev["multilabel"][idx], hp.f1_threshold)
        f1 = f1fx / float(1 << 32) / N_dev
        if history is not None:
            history.append(dict(train_loss=tr_loss, dev_loss=run_loss / N_dev, dev_f1=f1, dev_f1_fixed=f1fx))
        if tr_loss != tr_loss:
            break
        if f1 > best:
            best = f1
    return 0.0 if best != best else best


def train_sampled_models(confs, hp: Hyper, train, dev, init_seed=0, order=None, drop_seed=0,
                         params_list=None):
    """Population loop (ntu_searchable.py:38-94): every configuration trained from scratch,
    accuracies returned in input order."""
    accs = []
    for ci, conf in enumerate(confs):
        if params_list is not None:
            params = {k: v.copy() for k, v in params_list[ci].items()}
        else:
            params = init_params(conf, hp, init_seed + ci)
        accs.append(train_candidate(conf, hp, params, train, dev, order=order, seed=drop_seed + ci))
    return accs


# --------------------------------------------------------------------------- synthetic NTU-shaped tables
def bf16_round(x: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even float32 -> bfloat16 -> float32 (storage quantisation of the taps)."""
    u = np.ascontiguousarray(x, dtype=F32).view(np.uint32)
    r = ((u >> np.uint32(16)) & np.uint32(1)) + np.uint32(0x7FFF)
    u2 = ((u + r) & np.uint32(0xFFFF0000)).astype(np.uint32)
    return u2.view(F32)


def synth_table(N, seed, snr=0.12, C=60, mu_seed=123, s_sizes=S_SIZES, v_sizes=V_SIZES,
                quant=None, with_logits=False):
    """Planted-signal NTU-shaped taps: x = relu(snr*mu[label] + eps), label uniform,
    mu/eps unit-variance hash noise (bit-reproducible, so fixtures store only seeds)."""
    with np.errstate(over="ignore"):
        label = (_lowbias32(np.arange(N, dtype=np.uint32) ^ _lowbias32(
            np.array([(seed * 0x85EBCA6B + 1) & 0xFFFFFFFF], np.uint32))[0]) % np.uint32(C)).astype(np.int64)
    t = {"label": label}
    slot = 0
    for name, sizes in (("s", s_sizes), ("v", v_sizes)):
        for j, w in enumerate(sizes):
            slot += 1
            mu = hash_noise(param_seed(mu_seed, slot), C * w).reshape(C, w)
            eps = hash_noise(param_seed(seed, 100 + slot), N * w).reshape(N, w)
            x = np.maximum(F32(snr) * mu[label] + eps, F32(0)).astype(F32)
            if quant == "bf16":
                x = bf16_round(x)
            elif quant == "fp16":
                x = x.astype(np.float16).astype(F32)
            t[f"{name}{j}"] = x
    if with_logits:
        for name in ("vlogit", "slogit"):
            slot += 1
            mu = hash_noise(param_seed(mu_seed, slot), C * C).reshape(C, C)
            eps = hash_noise(param_seed(seed, 100 + slot), N * C).reshape(N, C)
            t[name] = (F32(0.5) * mu[label] + eps).astype(F32)
    return t


def sample_view(arr: np.ndarray, max_n: int = 1500) -> np.ndarray:
    """Strided sample of a tensor (fixtures store samples + a sum instead of megabytes)."""
    flat = np.asarray(arr).ravel()
    stride = max(1, -(-flat.size // max_n))
    return flat[::stride].copy()


def fake_accuracy(conf) -> float:
    """Deterministic stand-in for a trained candidate's accuracy (controller tests only)."""
    c = np.asarray(conf, np.int64).reshape(-1, 3)
    h = 17
    for row in c:
        for v in row:
            h = (h * 31 + int(v) + 7) % 1000003
    return 0.25 + 0.7 * (h % 1000) / 1000.0


MM_S_SIZES = (64, 128, 64, 128)      # text taps o1 (64-d), o3 (128-d) of MaxOut_MLP, models/central/mm_imdb.py:176-196
MM_V_SIZES = (512, 512, 512, 512)    # image taps of GP_VGG (feature idx 20/26/33/36), models/central/mm_imdb.py:19-59


def synth_table_mm(N, seed, C=23, snr=0.6, p_on=0.15, mu_seed=321, quant=None):
    """MM-IMDB-shaped multi-label table: multi-hot targets, taps = relu(snr * sum of active class prototypes + eps)."""
    z = (hash_u01(param_seed(seed, 900), N * C).reshape(N, C) < F32(p_on)).astype(F32)
    t = {"multilabel": z, "label": np.zeros(N, np.int64)}
    slot = 0
    for name, sizes in (("s", MM_S_SIZES), ("v", MM_V_SIZES)):
        for j, wd in enumerate(sizes):
            slot += 1
            mu = hash_noise(param_seed(mu_seed, slot), C * wd).reshape(C, wd)
            eps = hash_noise(param_seed(seed, 100 + slot), N * wd).reshape(N, wd)
            x = np.maximum(F32(snr) * (z @ mu).astype(F32) + eps, F32(0)).astype(F32)
            if quant == "bf16":
                x = bf16_round(x)
            elif quant == "fp16":
                x = x.astype(np.float16).astype(F32)
            t[f"{name}{j}"] = x
    return t


def mm_pos_weight(C=23, seed=7):
    return (F32(1.0) + F32(3.0) * hash_u01(param_seed(seed, 950), C)).astype(F32)

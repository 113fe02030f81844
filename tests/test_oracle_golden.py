"""Pins the numpy oracle (oracle/np_oracle.py) against golden vectors produced by the UNCHANGED
reference under torch 2.10 (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest

from oracle import np_oracle as O

CONFS = {
    "c4": [[3, 1, 1], [1, 3, 0], [1, 1, 1], [3, 3, 0]],
    "c0": [[2, 2, 0], [1, 0, 1], [3, 2, 0], [3, 1, 1]],
    "l1": [[0, 0, 0]],
    "l2": [[2, 3, 1], [0, 2, 2]],
    "l3": [[1, 0, 2], [3, 3, 1], [0, 1, 0]],
}
VARIANTS = {
    "bn_train": (dict(bn=True, drpt=0.0), True),
    "bn_eval": (dict(bn=True, drpt=0.0), False),
    "bndrop_eval": (dict(bn=True, drpt=0.5), False),
    "drop_eval": (dict(bn=False, drpt=0.5), False),
    "alpha_bn_train": (dict(bn=True, drpt=0.0, alphas=True), True),
    "mt_bn_train": (dict(bn=True, drpt=0.0, multitask=True), True),
}


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def check(g, key, arr, rtol, atol, scale_atol=0.0, loose_atol=None):
    """scale_atol: extra absolute tolerance as a fraction of max|expected| (fp32 cancellation noise
    in BN-backward is relative to the tensor's scale, not the element).
    loose_atol: Adam normalises by sqrt(v), so an element whose gradient is at round-off level moves
    by up to ~lr per step in either direction; <=3% of the elements may miss the tight tolerance
    but every element must meet this loose one."""
    full = key in g
    want = g[key] if full else g[key + "#s"]
    got = np.asarray(arr) if full else O.sample_view(arr)
    atol = atol + scale_atol * float(np.abs(want).max())
    if loose_atol is None:
        np.testing.assert_allclose(got, want, rtol=rtol, atol=atol, err_msg=key)
    else:
        bad = np.abs(got - want) > atol + rtol * np.abs(want)
        assert bad.mean() <= 0.03, (key, bad.mean())
        np.testing.assert_allclose(got, want, rtol=rtol, atol=loose_atol, err_msg=key)
    if not full:
        np.testing.assert_allclose(np.asarray(arr, np.float64).sum(), g[key + "#sum"], rtol=2e-3,
                                   atol=atol * arr.size ** 0.5, err_msg=key + "#sum")


# ------------------------------------------------------------------ G1
def test_scheduler_known_answers(golden_dir):
    g = load(golden_dir, "g1_scheduler.npz")
    for j in range(4):
        Ti, Tm, nbpe, n = g[f"cfg{j}"]
        seq = O.eta_sequence(1e-3, 1e-6, Ti, Tm, nbpe, int(n))
        np.testing.assert_allclose(seq, g[f"eta{j}"], rtol=1e-12, atol=0)
    # SURVEY §3.3 known answer
    seq = O.eta_sequence(1e-3, 1e-6, 1, 2, 4.0, 6)
    np.testing.assert_allclose(seq, [1e-3, 8.537e-4, 5.005e-4, 1.473e-4, 1e-6, 1e-3], rtol=1e-3)


# ------------------------------------------------------------------ G2/G3/G9
def test_forward_loss_grads(golden_dir):
    g = load(golden_dir, "g23_forward_backward.npz")
    t = O.synth_table(16, 11, snr=0.3, with_logits=True)
    labels = t["label"]
    n = 0
    for name in g["names"]:
        cname, vname, R, seed = str(name).split("/")
        R, seed = int(R), int(seed)
        kw, train = VARIANTS[vname]
        hp = O.Hyper(R=R, B=16, **kw)
        conf = np.array(CONFS[cname])
        params = O.init_params(conf, hp, seed, perturb_bn=True)
        pre = f"{cname}/{vname}/{R}/"
        logits, cache = O.forward(params, conf, hp, t, train)
        np.testing.assert_allclose(logits, g[pre + "logits"], rtol=2e-4, atol=2e-5, err_msg=pre)
        loss, dlog, preds = O.ce_loss(logits, labels)
        if hp.multitask:
            np.testing.assert_allclose(loss, g[pre + "loss_central"], rtol=1e-5)
            l3 = loss + O.ce_loss(t["vlogit"], labels)[0] + O.ce_loss(t["slogit"], labels)[0]
            np.testing.assert_allclose(l3, g[pre + "loss"], rtol=1e-5)
            preds = O.predict(logits + t["vlogit"] + t["slogit"])
        else:
            np.testing.assert_allclose(loss, g[pre + "loss"], rtol=1e-5)
        assert np.array_equal(preds, g[pre + "preds"]), pre
        if train:
            grads = O.backward(params, hp, cache, dlog)
            for k, v in grads.items():
                check(g, pre + "grad/" + k, v, rtol=2e-3, atol=2e-7, scale_atol=1e-3)
            O.bn_update_running(params, hp, cache)
            for i in range(len(conf)):
                for s in ("running_mean", "running_var"):
                    k = f"fusion_layers.{i}.2.{s}"
                    np.testing.assert_allclose(params[k], g[pre + "after/" + k], rtol=1e-5, atol=1e-6)
        n += 1
    assert n >= 40


# ------------------------------------------------------------------ G4/G5/G6
@pytest.mark.parametrize("cname,R", [("c4", 16), ("c4", 128), ("l2", 16)])
def test_deterministic_trajectory(golden_dir, cname, R):
    g = load(golden_dir, "g456_trajectory.npz")
    ttr, tdv = O.synth_table(64, 21, snr=0.3), O.synth_table(48, 22, snr=0.3)
    conf = np.array(CONFS[cname])
    hp = O.Hyper(R=R, B=16, bn=True, drpt=0.0, epochs=3)
    params = O.init_params(conf, hp, 5)
    pre = f"{cname}/{R}/"
    losses = []

    def on_step(gstep, p, st, loss):
        losses.append(loss)
        step = gstep + 1
        if step in (1, 2, 10):
            for k, v in p.items():
                if k.startswith("alphas"):
                    continue
                check(g, pre + f"step{step}/p/" + k, v, rtol=1e-4, atol=2e-6 * step, loose_atol=1e-3 * step)
            for k in st.m:
                check(g, pre + f"step{step}/m/" + k, st.m[k], rtol=2e-3, atol=1e-9, scale_atol=2e-3, loose_atol=1.0)
                check(g, pre + f"step{step}/v/" + k, st.v[k], rtol=4e-3, atol=1e-14, scale_atol=2e-3, loose_atol=1.0)

    hist = []
    best = O.train_candidate(conf, hp, params, ttr, tdv, history=hist, on_step=on_step)
    np.testing.assert_allclose(losses, g[pre + "losses"], rtol=2e-4)
    ghist = g[pre + "hist"]      # rows: (phase, loss, acc) printed with 4 decimals by the reference
    for ep, h in enumerate(hist):
        tr_row, dv_row = ghist[2 * ep], ghist[2 * ep + 1]
        assert abs(h["train_loss"] - tr_row[1]) < 2e-4 and abs(h["train_acc"] - tr_row[2]) < 1e-4
        assert abs(h["dev_loss"] - dv_row[1]) < 2e-4 and abs(h["dev_acc"] - dv_row[2]) < 1e-4
    assert best == pytest.approx(float(g[pre + "best_acc"]), abs=1e-12)


# ------------------------------------------------------------------ G7
@pytest.mark.parametrize("B", [16, 20])
def test_population_accuracies(golden_dir, B):
    g = load(golden_dir, "g7_population.npz")
    ttr, tdv = O.synth_table(256, 31, snr=0.5), O.synth_table(128, 32, snr=0.5)
    confs = [g[f"conf{i}"] for i in range(4)]
    hp = O.Hyper(R=16, B=B, bn=True, drpt=0.0, epochs=3)
    accs = O.train_sampled_models(confs, hp, ttr, tdv, init_seed=9)
    np.testing.assert_allclose(accs, g[f"B{B}/accs"], atol=1e-12)


def test_multitask_and_alphas_runs(golden_dir):
    g = load(golden_dir, "g7_population.npz")
    ttr = O.synth_table(256, 31, snr=0.5, with_logits=True)
    tdv = O.synth_table(128, 32, snr=0.5, with_logits=True)
    hp = O.Hyper(R=16, B=16, bn=True, drpt=0.0, epochs=3, multitask=True)
    conf = np.array(CONFS["c0"])
    hist = []
    acc = O.train_candidate(conf, hp, O.init_params(conf, hp, 13), ttr, tdv, history=hist)
    assert acc == pytest.approx(float(g["mt_acc"]), abs=1e-12)
    for ep, h in enumerate(hist):       # printed accuracies use the summed-logit argmax
        assert abs(h["train_acc"] - g["mt_hist"][2 * ep][2]) < 1e-4
        assert abs(h["dev_acc"] - g["mt_hist"][2 * ep + 1][2]) < 1e-4
        # the printed Loss is the 3-term multitask loss CE(central) + CE(visual) + CE(skeleton) (train_searchable/ntu.py:60-61)
        assert abs(h["train_loss"] - g["mt_hist"][2 * ep][1]) < 2e-4
        assert abs(h["dev_loss"] - g["mt_hist"][2 * ep + 1][1]) < 2e-4
    hp = O.Hyper(R=16, B=16, bn=True, drpt=0.0, epochs=3, alphas=True)
    conf = np.array(CONFS["l3"])
    params = O.init_params(conf, hp, 21)
    t1, t2 = O.synth_table(256, 31, snr=0.5), O.synth_table(128, 32, snr=0.5)
    acc = O.train_candidate(conf, hp, params, t1, t2)
    assert acc == pytest.approx(float(g["alpha_acc"]), abs=1e-12)
    got = [float(params[f"alphas.{i}.alpha_x"][0]) for i in range(3)]
    np.testing.assert_allclose(got, g["alpha_final"], rtol=2e-3, atol=1e-6)


# ------------------------------------------------------------------ G18 train-mode dropout, pinned pointwise (mask injection)
G18A_KW = {"bndrop": dict(bn=True, drpt=0.5), "drop": dict(bn=False, drpt=0.5),
           "bndrop04": dict(bn=True, drpt=0.4), "drop04": dict(bn=False, drpt=0.4)}
G18B_CASES = {"search": ("c4", 16, False, 0.5, 20, 120, 60, 1.0), "search_l3": ("l3", 16, False, 0.5, 20, 130, 70, 1.0),
              "bench": ("c4", 128, True, 0.5, 16, 64, 48, 0.3), "bench16": ("l2", 16, True, 0.4, 16, 64, 48, 0.3)}


def g18b_order(tag, E, N):
    rng = np.random.default_rng(1800 + sum(map(ord, tag)))
    return np.stack([rng.permutation(N) for _ in range(E)])


def test_dropout_train_forward_loss_grads(golden_dir):
    """G18a: the reference's train-mode forward/backward with its nn.Dropout instances replaced by the shared hash mask
    (tests/golden/make_golden.py::HashDropout) — [Linear, nl, BN, Dropout] and [Linear, nl, Dropout] (ntu_searchable.py:275-282),
    nl in {ReLU, Sigmoid, LeakyReLU}, p in {0.5, 0.4}, full and ragged batches, two positions of the mask stream."""
    g = load(golden_dir, "g18a_dropout_forward_backward.npz")
    t = O.synth_table(16, 11, snr=0.3, with_logits=True)
    seen = set()
    for name in g["names"]:
        cname, vname, R, rows, step, seed = str(name).split("/")
        R, rows, step, seed = int(R), int(rows), int(step), int(seed)
        hp = O.Hyper(R=R, B=16, **G18A_KW[vname])
        conf = np.array(CONFS[cname])
        params = O.init_params(conf, hp, seed, perturb_bn=True)
        feats = {k: v[:rows] for k, v in t.items() if k != "label"}
        pre = f"{cname}/{vname}/{R}/{rows}/"
        logits, cache = O.forward(params, conf, hp, feats, True, seed=seed + 5, step=step)
        np.testing.assert_allclose(logits, g[pre + "logits"], rtol=2e-4, atol=2e-5, err_msg=pre)
        loss, dlog, preds = O.ce_loss(logits, t["label"][:rows])
        np.testing.assert_allclose(loss, g[pre + "loss"], rtol=1e-5)
        assert np.array_equal(preds, g[pre + "preds"]), pre
        for k, v in O.backward(params, hp, cache, dlog).items():
            check(g, pre + "grad/" + k, v, rtol=2e-3, atol=2e-7, scale_atol=1e-3)
        if hp.bn:
            O.bn_update_running(params, hp, cache)
            for i in range(len(conf)):
                for s in ("running_mean", "running_var"):
                    k = f"fusion_layers.{i}.2.{s}"
                    np.testing.assert_allclose(params[k], g[pre + "after/" + k], rtol=1e-5, atol=1e-6)
        seen.update((vname, int(c[2])) for c in conf)
    assert len(g["names"]) >= 40
    assert seen >= {(v, nl) for v in G18A_KW for nl in (0, 1, 2)}       # every legal dropout cell x every non-linearity


@pytest.mark.parametrize("tag", list(G18B_CASES))
def test_dropout_trajectory(golden_dir, tag):
    """G18b: W / m / v after 1, 2, 10 Adam steps and the 3-epoch trajectory of the unchanged train_sampled_models with dropout
    ON (injected masks) and a shuffled fixed order — the search default (R=16, no BN, B=20) and the bench cell (R=128, BN, B=16)."""
    g = load(golden_dir, "g18b_dropout_trajectory.npz")
    cname, R, bn, drpt, B, N, Nd, snr = G18B_CASES[tag]
    assert np.array_equal(g[tag + "/meta"], np.array([R, int(bn), drpt, B, N, Nd, snr]))
    ttr, tdv = O.synth_table(N, 21, snr=snr), O.synth_table(Nd, 22, snr=snr)
    conf = np.array(CONFS[cname])
    hp = O.Hyper(R=R, B=B, bn=bn, drpt=drpt, epochs=3)
    order = g18b_order(tag, 3, N)
    params = O.init_params(conf, hp, 5)
    pre = tag + "/"
    losses = []

    def on_step(gstep, p, st, loss):
        losses.append(loss)
        step = gstep + 1
        if step in (1, 2, 10):
            for k, v in p.items():
                if k.startswith("alphas") or (not bn and ".2." in k):
                    continue
                # (running statistics are an EMA of batch moments of activations that already carry the weights' 1e-4)
                check(g, pre + f"step{step}/p/" + k, v, rtol=1e-3 if "running" in k else 1e-4, atol=2e-6 * step, loose_atol=1e-3 * step)
            for k in st.m:
                check(g, pre + f"step{step}/m/" + k, st.m[k], rtol=2e-3, atol=1e-9, scale_atol=2e-3, loose_atol=1.0)
                check(g, pre + f"step{step}/v/" + k, st.v[k], rtol=4e-3, atol=1e-14, scale_atol=2e-3, loose_atol=1.0)

    hist = []
    best = O.train_candidate(conf, hp, params, ttr, tdv, order=order, seed=40, history=hist, on_step=on_step)
    np.testing.assert_allclose(losses, g[pre + "losses"], rtol=5e-4)
    ghist = g[pre + "hist"]
    for ep, h in enumerate(hist):
        tr_row, dv_row = ghist[2 * ep], ghist[2 * ep + 1]
        assert abs(h["train_loss"] - tr_row[1]) < 3e-4 and abs(h["train_acc"] - tr_row[2]) < 1e-4
        assert abs(h["dev_loss"] - dv_row[1]) < 3e-4 and abs(h["dev_acc"] - dv_row[2]) < 1e-4      # exact dev counts
    assert best == pytest.approx(float(g[pre + "best_acc"]), abs=1e-12)


# ------------------------------------------------------------------ G8
def test_layer_configurations(golden_dir):
    g = load(golden_dir, "g8_controller.npz")
    assert np.array_equal(np.array(O.get_possible_layer_configurations(0)), g["layer_confs"])
    assert len(O.get_possible_layer_configurations(2)) == 32


def test_illegal_variant_raises():
    with pytest.raises(ValueError):
        O.Hyper(bn=False, drpt=0.0).check()


def test_dropout_mask_rate():
    keep = O.dropout_keep(7, 3, 1, 64, 128, 0.5)
    assert abs(keep.mean() - 0.5) < 0.03
    keep2 = O.dropout_keep(7, 4, 1, 64, 128, 0.5)
    assert (keep != keep2).mean() > 0.4


def test_torch_restatement_matches_numpy_oracle():
    """oracle/torch_restatement.py (the PyTorch-CPU eager baseline bench.py times) follows the same step sequence as the
    reference-pinned numpy oracle: same network from the same parameters -> same logits, same loss, and after one Adam step
    with the scheduler's eta the same parameters."""
    torch = pytest.importorskip("torch")
    from oracle import torch_restatement as TR
    conf = np.array(CONFS["c4"])
    hp = O.Hyper(R=16, B=16, bn=True, drpt=0.0, epochs=1)
    t = O.synth_table(16, 5, snr=0.4)
    params = O.init_params(conf, hp, 3)
    net = TR.FusionNet(conf, 16, 60, True, 0.0)
    sd = net.state_dict()
    for k, v in params.items():
        if k in sd:
            sd[k].copy_(torch.from_numpy(v))
    net.train(True)
    feats = {k: torch.from_numpy(v) for k, v in t.items() if k != "label"}
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-4)
    out = net(feats)
    loss = torch.nn.CrossEntropyLoss()(out, torch.from_numpy(t["label"]))
    want_logits, cache = O.forward({k: v.copy() for k, v in params.items()}, conf, hp, {k: v for k, v in t.items() if k != "label"}, True)
    np.testing.assert_allclose(out.detach().numpy(), want_logits, rtol=2e-4, atol=2e-5)
    want_loss, dlog, _ = O.ce_loss(want_logits, t["label"])
    assert abs(float(loss) - float(want_loss)) < 1e-5
    eta = float(O.eta_sequence(1e-3, 1e-6, 1, 2, 1.0, 1)[0])
    TR.push_lr(opt, eta)
    loss.backward()
    opt.step()
    p2 = {k: v.copy() for k, v in params.items()}
    grads = O.backward(p2, hp, cache, dlog)
    st = O.AdamState()
    O.adam_step(p2, grads, st, eta, hp, O.trainable_keys(conf, hp))
    got = net.state_dict()
    for k in ("fusion_layers.0.0.weight", "fusion_layers.3.0.bias", "central_classifier.weight", "fusion_layers.1.2.weight"):
        np.testing.assert_allclose(got[k].numpy(), p2[k], rtol=0, atol=2.1e-3)      # |dw| = lr at step 1: signs must agree
        assert np.mean(np.abs(got[k].numpy() - p2[k]) > 1e-5) < 0.02, k

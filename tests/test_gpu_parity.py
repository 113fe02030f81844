"""GPU parity tests proper: the HIP engine (through the C ABI, via ctypes) against the numpy oracle on the
same seeded inputs, and directly against the golden vectors of the unchanged reference.

Tolerances (float32 path; north_star: dev top-1 within +-0.1 %):
  * logits / losses: 2e-4 relative to the tensor scale (summation order differs from ATen's);
  * parameters after k Adam steps: <=3 % of the elements may deviate by more than 1e-4 relative
    (Adam divides by sqrt(v): an element whose gradient is at round-off level moves by ~lr in either
    direction), every element within k*lr;
  * dev correct COUNTS on the short deterministic trajectories: exact; accuracies within 0.1 % top-1 where
    the set is large enough for that to be more than one sample.
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
from oracle import np_oracle as O
from tests.helpers import CONFS, engine_hyper, etas_for, frac_bad, golden, oracle_steps, rel_err  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "needs a HIP device"
    return torch.device("cuda:0")


def mk_pop(ohp, confs, dev, **kw):
    from mfas_amd import Population
    return Population(engine_hyper(ohp), [np.array(c) for c in confs], dev, **kw)


def table(t, dev, dtype=torch.float32):
    from mfas_amd import FeatureTable
    return FeatureTable.from_numpy(t, dev, dtype)


# ---------------------------------------------------------------------------------------- provenance
def test_loaded_library_is_built_from_this_tree(dev):
    """include/mfas_hip.h: mfas_source_digest() names the sources the library was compiled from; the library these tests
    exercise must be the committed tree's (sha256 over include/mfas_hip.h + mfas_amd/csrc/*.hip*, __graft_entry__.source_digest)."""
    import __graft_entry__ as ge
    from mfas_amd import _lib
    assert _lib.lib().mfas_source_digest().decode() == "mfas-src-digest:" + ge.source_digest()


# ---------------------------------------------------------------------------------------- layout
@pytest.mark.parametrize("R,bn", [(16, True), (128, True), (16, False), (24, True)])
def test_param_roundtrip_and_device_init(dev, R, bn):
    ohp = O.Hyper(R=R, B=16, bn=bn, drpt=0.5, alphas=True)
    confs = [CONFS["c4"], CONFS["l1"], CONFS["l3"]]
    pop = mk_pop(ohp, confs, dev)
    for k, c in enumerate(confs):
        p = O.init_params(c, ohp, 40 + k, perturb_bn=True)
        pop.set_state_dict(k, p)
    for k, c in enumerate(confs):
        p = O.init_params(c, ohp, 40 + k, perturb_bn=True)
        got = pop.get_state_dict(k)
        for key, v in p.items():
            if not bn and ".2." in key:
                continue
            assert np.array_equal(got[key].numpy(), v), key
        for plane in (1, 2):
            assert float(pop.get_params(k, plane).abs().max()) == 0.0
    pop.init([7, 8, 9])          # device-side hash init == oracle init, bit for bit
    for k, c in enumerate(confs):
        p = O.init_params(c, ohp, 7 + k)
        got = pop.get_state_dict(k)
        for key, v in p.items():
            if not bn and ".2." in key:
                continue
            assert np.array_equal(got[key].numpy(), v), key
    pop.close()


# ---------------------------------------------------------------------------------------- forward (eval) vs reference golden
def test_eval_forward_vs_reference_golden(dev):
    g = golden("g23_forward_backward.npz")
    t = O.synth_table(16, 11, snr=0.3, with_logits=True)
    tab = table(t, dev)
    n = 0
    for name in g["names"]:
        cname, vname, R, seed = str(name).split("/")
        if not vname.endswith("_eval"):
            continue
        R, seed = int(R), int(seed)
        kw = {"bn_eval": dict(bn=True, drpt=0.0), "bndrop_eval": dict(bn=True, drpt=0.5),
              "drop_eval": dict(bn=False, drpt=0.5)}[vname]
        ohp = O.Hyper(R=R, B=16, **kw)
        pop = mk_pop(ohp, [CONFS[cname]], dev)
        pop.set_state_dict(0, O.init_params(CONFS[cname], ohp, seed, perturb_bn=True))
        logits, corr = pop.forward(0, tab, count=True)
        want = g[f"{cname}/{vname}/{R}/logits"]
        assert rel_err(logits.cpu().numpy(), want) < 2e-4, name
        assert corr == int((g[f"{cname}/{vname}/{R}/preds"] == t["label"]).sum()), name
        pop.close()
        n += 1
    assert n >= 15


# ---------------------------------------------------------------------------------------- k train steps vs oracle AND reference golden
def _row_block_fracs(a, v, rtol, atol):
    """Fraction of out-of-tolerance elements per 16-row block of a weight matrix (the engine's tile rows)."""
    bad = np.abs(a - v) > atol + rtol * np.abs(v)
    nrb = -(-a.shape[0] // 16)
    return np.array([bad[16 * rb:16 * rb + 16].mean() for rb in range(nrb)])


def check_state(pop, k, params, st, steps, lr=1e-3, tag=""):
    """Parameters / Adam moments after `steps` train steps vs the oracle.  Adam's first steps are sign-like (dw = +-lr for every
    element whatever the size of its gradient), so elements whose gradient is round-off sized may legitimately land up to
    lr * steps apart; everything else has to agree to 1e-4.  Three layers of bounds:
      * bulk:       >= 99 % of the elements (94 % for vectors under 1000 elements) within rtol 1e-4;
      * tail:       >= 99.9 % within rtol 1e-2 (+ the same atol) — the sign-flip elements are few AND the rest is tight;
      * max:        every element within 2 lr * steps (a sign flip of a round-off-sized gradient);
      * structure:  no 16-row tile block of a weight matrix may hold more than 4x its share of the out-of-tolerance elements
                    (an indexing error confined to one tile row cannot hide inside the global allowance).
    Observed over the 2,300 tensors the GPU suite checks (MFAS_CHECK_STATS=<file> logs them): bulk <= 0.29 %, tail <= 0.03 %,
    worst row block <= 1.8 % for tensors of >= 1000 elements; the limits sit ~3x above that."""
    import os
    got = pop.get_state_dict(k, 0)
    gm = pop.get_state_dict(k, 1)
    gv = pop.get_state_dict(k, 2)
    log = os.environ.get("MFAS_CHECK_STATS")
    for key, v in params.items():
        if key.startswith("alphas") and key not in st.m:
            continue
        a = got[key].numpy()
        lim = 0.01 if a.size >= 1000 else 0.06          # small vectors: a handful of round-off-level elements
        # BN running statistics are an EMA of batch moments: they inherit the weights' allowed 1e-4 deviations of every step
        rtol = 1e-3 if key.endswith(("running_mean", "running_var")) else 1e-4
        atol = 2e-6 * steps
        fb = frac_bad(a, v, rtol, atol)
        ft = frac_bad(a, v, 1e-2, atol)
        rbmax = 0.0
        if a.ndim == 2 and a.shape[0] >= 32 and a.size >= 4096:
            rbf = _row_block_fracs(a, v, rtol, atol)
            rbmax = float(rbf.max())
            assert rbmax <= max(0.03, 4.0 * fb + 0.01), (tag, key, "row-block", rbf.round(3).tolist())
        if log:
            with open(log, "a") as f:
                f.write(f"{tag} {key} n={a.size} bulk={fb:.5f} tail={ft:.5f} rbmax={rbmax:.5f} max={np.abs(a - v).max():.3g}\n")
        assert fb <= lim, (tag, key, fb)
        assert ft <= (0.001 if a.size >= 1000 else 0.03), (tag, key, "tail", ft)
        # (|dw| = lr at step 1 whatever |g| is: a round-off-sized gradient with the opposite SIGN lands 2 lr away; the tail bound
        #  above keeps such elements under 0.1 %)
        assert np.abs(a - v).max() <= 2.0 * lr * steps, (tag, key)
    for key in st.m:
        # small vectors: a ReLU/dropout kink flipped by round-off moves one sample's share of a column sum (1/B), so a
        # handful of isolated elements may sit a few % off
        lim = 0.03 if st.m[key].size >= 1000 else max(0.06, 3.0 / st.m[key].size)
        sc = float(np.abs(st.m[key]).max()) + 1e-30
        assert frac_bad(gm[key].numpy(), st.m[key], 2e-3, 2e-3 * sc) <= lim, (tag, "m", key)
        sc = float(np.abs(st.v[key]).max()) + 1e-30
        assert frac_bad(gv[key].numpy(), st.v[key], 4e-3, 2e-3 * sc) <= lim, (tag, "v", key)


@pytest.mark.parametrize("cname,R", [("c4", 16), ("c4", 128), ("l2", 16)])
@pytest.mark.parametrize("steps", [1, 2, 10])
def test_train_steps_bn(dev, cname, R, steps):
    ohp = O.Hyper(R=R, B=16, bn=True, drpt=0.0, epochs=3)
    conf = np.array(CONFS[cname])
    ttr = O.synth_table(64, 21, snr=0.3)
    pop = mk_pop(ohp, [conf], dev)
    pop.set_state_dict(0, O.init_params(conf, ohp, 5))
    stats, status = pop.train(table(ttr, dev), None, 3, etas_for(ohp, 64), max_steps=steps)
    params, st, losses = oracle_steps(conf, ohp, O.init_params(conf, ohp, 5), ttr, steps)
    check_state(pop, 0, params, st, steps, tag=f"{cname}/{R}/{steps}")
    # the same numbers straight from the reference
    g = golden("g456_trajectory.npz")
    pre = f"{cname}/{R}/step{steps}/p/"
    got = pop.get_state_dict(0, 0)
    for key in got:
        if key.startswith("alphas"):
            continue
        full = pre + key in g
        want = g[pre + key] if full else g[pre + key + "#s"]
        a = got[key].numpy() if full else O.sample_view(got[key].numpy())
        assert frac_bad(a, want, 1e-4, 2e-6 * steps) <= (0.03 if a.size >= 1000 else 0.06), key
    nb = 4
    loss_sum = sum(losses[e] * 16 for e in range(min(steps, nb)))
    assert abs(stats["train_loss_sum"][0, 0] - loss_sum) < 2e-3 * max(1.0, loss_sum)
    pop.close()


@pytest.mark.parametrize("variant", ["drop", "bndrop", "lrelu_drop", "mt", "alphas", "ragged20"])
def test_train_steps_variants(dev, variant):
    kw = dict(R=16, B=16, bn=False, drpt=0.5, epochs=3)
    cname, N = "c4", 64
    with_logits = False
    if variant == "bndrop":
        kw.update(bn=True)
    elif variant == "lrelu_drop":
        cname = "l3"
    elif variant == "mt":
        kw.update(bn=True, drpt=0.0, multitask=True)
        with_logits = True
    elif variant == "alphas":
        kw.update(bn=True, drpt=0.0, alphas=True)
        cname = "l3"
    elif variant == "ragged20":
        kw.update(B=20, bn=True, drpt=0.4)
        N = 70            # batches of 20,20,20,10
    ohp = O.Hyper(**kw)
    conf = np.array(CONFS[cname])
    ttr = O.synth_table(N, 21, snr=0.3, with_logits=with_logits)
    pop = mk_pop(ohp, [conf, conf], dev, drop_seeds=[11, 12])
    for k in range(2):
        pop.set_state_dict(k, O.init_params(conf, ohp, 5 + k))
    steps = 6
    stats, status = pop.train(table(ttr, dev), None, 3, etas_for(ohp, N), max_steps=steps)
    for k in range(2):
        params, st, losses = oracle_steps(conf, ohp, O.init_params(conf, ohp, 5 + k), ttr, steps, seed=11 + k)
        check_state(pop, k, params, st, steps, tag=f"{variant}/{k}")
    assert not status.any()
    pop.close()


# ---------------------------------------------------------------------------------------- full trajectories
@pytest.mark.parametrize("cname,R", [("c4", 16), ("c4", 128), ("l2", 16)])
def test_deterministic_trajectory_vs_reference(dev, cname, R):
    g = golden("g456_trajectory.npz")
    ohp = O.Hyper(R=R, B=16, bn=True, drpt=0.0, epochs=3)
    conf = np.array(CONFS[cname])
    ttr, tdv = O.synth_table(64, 21, snr=0.3), O.synth_table(48, 22, snr=0.3)
    pop = mk_pop(ohp, [conf], dev)
    pop.set_state_dict(0, O.init_params(conf, ohp, 5))
    stats, status = pop.train(table(ttr, dev), table(tdv, dev), 3, etas_for(ohp, 64))
    hist = g[f"{cname}/{R}/hist"]       # rows (phase, loss, acc) as printed by the reference (4 decimals)
    for e in range(3):
        assert abs(stats["train_loss_sum"][0, e] / 64 - hist[2 * e][1]) < 3e-4
        assert abs(stats["train_corrects"][0, e] / 64 - hist[2 * e][2]) < 1e-4
        assert abs(stats["dev_loss_sum"][0, e] / 48 - hist[2 * e + 1][1]) < 3e-4
        assert abs(stats["dev_corrects"][0, e] / 48 - hist[2 * e + 1][2]) < 1e-4      # exact count
    from mfas_amd import best_dev_accuracy
    assert best_dev_accuracy(stats[0], 48) == pytest.approx(float(g[f"{cname}/{R}/best_acc"]), abs=1e-12)
    pop.close()


@pytest.mark.parametrize("B", [16, 20])
def test_population_vs_reference(dev, B):
    """train_sampled_models on 4 heterogeneous confs (L = 1..4); B=20 has ragged last batches."""
    g = golden("g7_population.npz")
    ttr, tdv = O.synth_table(256, 31, snr=0.5), O.synth_table(128, 32, snr=0.5)
    confs = [g[f"conf{i}"] for i in range(4)]
    ohp = O.Hyper(R=16, B=B, bn=True, drpt=0.0, epochs=3)
    pop = mk_pop(ohp, confs, dev)
    for k, c in enumerate(confs):
        pop.set_state_dict(k, O.init_params(c, ohp, 9 + k))
    stats, _ = pop.train(table(ttr, dev), table(tdv, dev), 3, etas_for(ohp, 256))
    from mfas_amd import best_dev_accuracy
    accs = [best_dev_accuracy(stats[k], 128) for k in range(4)]
    pop.close()
    # Per-epoch dev correct counts must equal the reference's (the oracle reproduces the reference's accuracies exactly,
    # tests/test_oracle_golden.py) EXCEPT where a dev sample is a numerical tie: the engine's weights agree with the
    # oracle's to ~1e-4 relative after 48 Adam steps, which moves logits by ~1e-3, so a sample whose decision margin
    # |logit[label] - best other logit| in the ORACLE run is below 5e-3 may fall on either side.  A count may differ by at
    # most the number of such near-tie samples of that epoch — and that number is asserted to be small.
    for k, c in enumerate(confs):
        hist = []
        want = O.train_candidate(c, ohp, O.init_params(c, ohp, 9 + k), ttr, tdv, history=hist)
        assert want == pytest.approx(float(g[f"B{B}/accs"][k]), abs=1e-12)          # oracle == reference
        for e, h in enumerate(hist):
            ties = int((h["dev_margins"] < 5e-3).sum())
            assert ties <= 2, (k, e, h["dev_margins"])
            assert abs(int(stats["dev_corrects"][k, e]) - h["dev_corrects"]) <= ties, (k, e, int(stats["dev_corrects"][k, e]), h["dev_corrects"], h["dev_margins"][:3])
        if all(int((h["dev_margins"] < 5e-3).sum()) == 0 for h in hist):
            assert accs[k] == float(g[f"B{B}/accs"][k]), (k, accs[k])
    np.testing.assert_allclose(accs, g[f"B{B}/accs"], atol=1.0 / 128 + 1e-9)   # in any case <= 1 sample of 128


def test_multitask_and_alphas_vs_reference(dev):
    g = golden("g7_population.npz")
    from mfas_amd import best_dev_accuracy
    ttr = O.synth_table(256, 31, snr=0.5, with_logits=True)
    tdv = O.synth_table(128, 32, snr=0.5, with_logits=True)
    ohp = O.Hyper(R=16, B=16, bn=True, drpt=0.0, epochs=3, multitask=True)
    conf = np.array(CONFS["c0"])
    pop = mk_pop(ohp, [conf], dev)
    pop.set_state_dict(0, O.init_params(conf, ohp, 13))
    stats, _ = pop.train(table(ttr, dev), table(tdv, dev), 3, etas_for(ohp, 256))
    # dev counts against the oracle's multitask run (== the reference's printed accuracies, tests/test_oracle_golden.py), which may
    # differ only where a dev sample is a numerical tie in the oracle run (margin of the summed-logit decision < 5e-3)
    hist = []
    obest = O.train_candidate(conf, ohp, O.init_params(conf, ohp, 13), ttr, tdv, history=hist)
    assert obest == pytest.approx(float(g["mt_acc"]), abs=1e-12)
    for e, h in enumerate(hist):
        ties = int((h["dev_margins"] < 5e-3).sum())
        assert ties <= 3, (e, h["dev_margins"])                 # of 128 dev samples
        assert abs(int(stats["dev_corrects"][0, e]) - h["dev_corrects"]) <= ties, (e, int(stats["dev_corrects"][0, e]), h["dev_corrects"])
    if all(int((h["dev_margins"] < 5e-3).sum()) == 0 for h in hist):
        assert best_dev_accuracy(stats[0], 128) == float(g["mt_acc"])
    for e in range(3):
        # 3-term multitask loss as the reference prints it (train_searchable/ntu.py:60-61,72-75)
        assert abs(stats["train_loss_sum"][0, e] / 256 - g["mt_hist"][2 * e][1]) < 2e-3
        assert abs(stats["dev_loss_sum"][0, e] / 128 - g["mt_hist"][2 * e + 1][1]) < 2e-3
    pop.close()
    ohp = O.Hyper(R=16, B=16, bn=True, drpt=0.0, epochs=3, alphas=True)
    conf = np.array(CONFS["l3"])
    pop = mk_pop(ohp, [conf], dev)
    pop.set_state_dict(0, O.init_params(conf, ohp, 21))
    t1, t2 = O.synth_table(256, 31, snr=0.5), O.synth_table(128, 32, snr=0.5)
    stats, _ = pop.train(table(t1, dev), table(t2, dev), 3, etas_for(ohp, 256))
    hist = []
    obest = O.train_candidate(conf, ohp, O.init_params(conf, ohp, 21), t1, t2, history=hist)
    assert obest == pytest.approx(float(g["alpha_acc"]), abs=1e-12)
    for e, h in enumerate(hist):
        ties = int((h["dev_margins"] < 5e-3).sum())
        assert ties <= 3, (e, h["dev_margins"])
        assert abs(int(stats["dev_corrects"][0, e]) - h["dev_corrects"]) <= ties, (e, int(stats["dev_corrects"][0, e]), h["dev_corrects"])
    got = pop.get_state_dict(0)
    al = [float(got[f"alphas.{i}.alpha_x"][0]) for i in range(3)]
    np.testing.assert_allclose(al, g["alpha_final"], rtol=3e-2, atol=5e-4)   # scalar fed by a cancelling S-V sum; 48 Adam steps at lr<=1e-3
    pop.close()


def test_dropout_trajectory_vs_oracle(dev):
    """Dropout on: the engine and the oracle share the counter-based mask, so whole trajectories agree."""
    ohp = O.Hyper(R=16, B=16, bn=False, drpt=0.5, epochs=2)
    confs = [np.array(CONFS["c4"]), np.array(CONFS["l2"])]
    ttr, tdv = O.synth_table(256, 1, snr=1.0), O.synth_table(256, 2, snr=1.0)
    rng = np.random.default_rng(3)
    order = np.stack([rng.permutation(256) for _ in range(2)])
    pop = mk_pop(ohp, confs, dev, drop_seeds=[100, 101])
    for k, c in enumerate(confs):
        pop.set_state_dict(k, O.init_params(c, ohp, 50 + k))
    stats, _ = pop.train(table(ttr, dev), table(tdv, dev), 2, etas_for(ohp, 256),
                         order=torch.from_numpy(order.astype(np.int32)))
    for k, c in enumerate(confs):
        hist = []
        O.train_candidate(c, ohp, O.init_params(c, ohp, 50 + k), ttr, tdv, order=order, seed=100 + k, history=hist)
        for e in range(2):
            assert abs(stats["train_loss_sum"][k, e] / 256 - hist[e]["train_loss"]) < 2e-3
            assert abs(stats["dev_corrects"][k, e] - hist[e]["dev_corrects"]) <= 2      # <1 % of 256
    pop.close()


# ---------------------------------------------------------------------------------------- dropout path vs the REFERENCE, pointwise
G18A_KW = {"bndrop": dict(bn=True, drpt=0.5), "drop": dict(bn=False, drpt=0.5),
           "bndrop04": dict(bn=True, drpt=0.4), "drop04": dict(bn=False, drpt=0.4)}
G18B_CASES = {"search": ("c4", 16, False, 0.5, 20, 120, 60, 1.0), "search_l3": ("l3", 16, False, 0.5, 20, 130, 70, 1.0),
              "bench": ("c4", 128, True, 0.5, 16, 64, 48, 0.3), "bench16": ("l2", 16, True, 0.4, 16, 64, 48, 0.3)}


def g18b_order(tag, E, N):
    rng = np.random.default_rng(1800 + sum(map(ord, tag)))
    return np.stack([rng.permutation(N) for _ in range(E)])


def _gold(g, key, arr):
    """(expected, got) with large tensors reduced to the golden's strided sample."""
    if key in g:
        return g[key], np.asarray(arr)
    return g[key + "#s"], O.sample_view(np.asarray(arr))


def test_dropout_train_forward_backward_vs_reference_golden(dev):
    """G18a: the reference's own train-mode forward / backward with dropout ON (its nn.Dropout instances swapped for the shared
    hash mask, everything else unchanged) — logits of mfas_population_forward_train, and after ONE engine train step Adam's
    first moment m = (1 - beta1) (g + wd w) against the reference's gradient g of every central tensor, for both legal
    dropout cells x {ReLU, Sigmoid, LeakyReLU} x p in {0.5, 0.4}, R in {16, 128}, full and ragged batches."""
    g = golden("g18a_dropout_forward_backward.npz")
    t = O.synth_table(16, 11, snr=0.3, with_logits=True)
    n_fwd = n_bwd = 0
    for name in g["names"]:
        cname, vname, R, rows, step, seed = str(name).split("/")
        R, rows, step, seed = int(R), int(rows), int(step), int(seed)
        ohp = O.Hyper(R=R, B=16, **G18A_KW[vname])
        conf = np.array(CONFS[cname])
        params = O.init_params(conf, ohp, seed, perturb_bn=True)
        tab = table({k: v[:rows] for k, v in t.items()}, dev)
        pre = f"{cname}/{vname}/{R}/{rows}/"
        pop = mk_pop(ohp, [conf], dev, drop_seeds=[seed + 5])
        pop.set_state_dict(0, params)
        logits = pop.forward_train(0, tab, 0, rows, step=step).cpu().numpy()
        assert rel_err(logits, g[pre + "logits"]) < 3e-4, (name, rel_err(logits, g[pre + "logits"]))
        n_fwd += 1
        if ohp.bn:
            sd = pop.get_state_dict(0)
            for i in range(len(conf)):
                for nm in ("running_mean", "running_var"):
                    k = f"fusion_layers.{i}.2.{nm}"
                    np.testing.assert_allclose(sd[k].numpy(), g[pre + "after/" + k], rtol=2e-4, atol=2e-6, err_msg=name)
        if step == 0:       # a train step draws mask-stream position 0
            pop.set_state_dict(0, params)
            stats, status = pop.train(tab, None, 1, etas_for(ohp, rows), max_steps=1)
            assert not status.any()
            assert abs(stats["train_loss_sum"][0, 0] / rows - float(g[pre + "loss"])) < 3e-4 * max(1.0, float(g[pre + "loss"]))
            assert int(stats["train_corrects"][0, 0]) == int((g[pre + "preds"] == t["label"][:rows]).sum()), name
            m = pop.get_state_dict(0, 1)
            for key in O.trainable_keys(conf, ohp):
                want_g, got_m = _gold(g, pre + "grad/" + key, m[key].numpy())
                want_w, _ = (params[key], None) if pre + "grad/" + key in g else (O.sample_view(params[key]), None)
                want_m = (1.0 - ohp.beta1) * (want_g.astype(np.float64) + ohp.wd * want_w)
                sc = float(np.abs(want_m).max()) + 1e-30
                assert np.abs(got_m - want_m).max() <= 2e-3 * sc + 2e-8, (name, key, np.abs(got_m - want_m).max(), sc)
            n_bwd += 1
        pop.close()
    assert n_fwd >= 40 and n_bwd >= 20


def check_one_step_map(dev, ohp, conf, state, ttr, rows, seed, tag):
    """The engine's one-step map at a given state: load `state`, take ONE train step (fresh Adam, mask-stream position 0) on the
    batch `rows`, and compare Adam's first moment m = (1 - beta1)(g + wd w) — i.e. the gradient of every central tensor —, the
    step's signs and the BN running statistics with the oracle's from the same state.  Unlike a k-step comparison this has no
    chaos in it: after k steps ONE sign flip of a round-off-sized gradient (|dw| = lr whatever |g|) perturbs a unit's
    pre-activations by ~1e-3 relative, and batch-of-16 BN + 2x dropout scaling spread that over every later gradient
    (tools/state_diag.py: R=128 BN+dropout, one flipped element of 1.04 M after step 1 -> gradient differences of 2e-3 at
    step 2, 1e-2 at step 3, while the one-step map from the oracle's own state agrees to 3e-6)."""
    sub = {k: np.ascontiguousarray(v[rows]) for k, v in ttr.items()}
    pop = mk_pop(ohp, [conf], dev, drop_seeds=[seed])
    pop.set_state_dict(0, {k: v.copy() for k, v in state.items()})
    stats, status = pop.train(table(sub, dev), None, 1, etas_for(ohp, len(rows)), max_steps=1)
    assert not status.any()
    P, st, losses = oracle_steps(conf, ohp, {k: v.copy() for k, v in state.items()}, sub, 1, seed=seed)
    assert abs(stats["train_loss_sum"][0, 0] / len(rows) - losses[0]) < 2e-4 * max(1.0, losses[0]), tag
    w, m = pop.get_state_dict(0, 0), pop.get_state_dict(0, 1)
    for key, mo in st.m.items():
        mm = m[key].numpy()
        rel = np.abs(mm - mo) / (np.abs(mo) + 1e-30)
        big = mo.size >= 1000
        assert np.median(rel) <= 1e-4, (tag, key, "median", float(np.median(rel)))
        # elements whose gradient is a cancelling sum carry the sum's absolute round-off: few, and small against the tensor's scale
        # (measured against |m| + 1e-3 of the tensor's scale: a bias gradient behind BN is a sum of terms with zero batch mean)
        rel2 = np.abs(mm - mo) / (np.abs(mo) + 1e-3 * float(np.abs(mo).max()) + 1e-30)
        assert (rel2 > 1e-2).mean() <= (0.005 if big else 2.0 / mo.size), (tag, key, float((rel2 > 1e-2).mean()))
        assert np.abs(mm - mo).max() <= 2e-3 * float(np.abs(mo).max()) + 1e-9, (tag, key)
        flips = np.abs(w[key].numpy() - P[key]) > 1e-5           # |dw| = lr = 1e-3: a sign decided by round-off
        assert flips.mean() <= (1e-3 if big else 2.0 / mo.size), (tag, key, "flips", int(flips.sum()))
    if ohp.bn:
        for key in P:
            if "running" in key:
                np.testing.assert_allclose(w[key].numpy(), P[key], rtol=2e-4, atol=2e-6, err_msg=f"{tag} {key}")
    pop.close()


@pytest.mark.parametrize("tag", list(G18B_CASES))
def test_dropout_steps_and_trajectory_vs_reference(dev, tag):
    """G18b: the unchanged train loop with dropout ON (injected masks) and a shuffled fixed order.
      * after step 1 (no chaos yet): W / m / v against the reference's tensors and the oracle's (check_state);
      * at the reference's states after 1, 2, 6 and 9 steps (the oracle's, which tests/test_oracle_golden.py::test_dropout_trajectory
        pins to the reference's W / m / v at 1e-4 and which are re-checked against the golden here): the engine's ONE-STEP map —
        gradient of every tensor, update signs, running statistics — equals the oracle's (check_one_step_map; step 7 of
        'search_l3' is a ragged batch of 10);
      * the 3-epoch trajectory: per-epoch dev COUNTS equal the reference's except on proven numerical ties, printed losses
        within 1e-3."""
    g = golden("g18b_dropout_trajectory.npz")
    cname, R, bn, drpt, B, N, Nd, snr = G18B_CASES[tag]
    ttr, tdv = O.synth_table(N, 21, snr=snr), O.synth_table(Nd, 22, snr=snr)
    conf = np.array(CONFS[cname])
    ohp = O.Hyper(R=R, B=B, bn=bn, drpt=drpt, epochs=3)
    order = g18b_order(tag, 3, N)
    dorder = torch.from_numpy(order.astype(np.int32))
    ta, tb = table(ttr, dev), table(tdv, dev)
    nb = -(-N // B)
    # --- step 1 against the reference's tensors
    pop = mk_pop(ohp, [conf], dev, drop_seeds=[40])
    pop.set_state_dict(0, O.init_params(conf, ohp, 5))
    pop.train(ta, None, 3, etas_for(ohp, N), order=dorder, max_steps=1)
    params, st, _ = oracle_steps(conf, ohp, O.init_params(conf, ohp, 5), ttr, 1, seed=40, order=order)
    check_state(pop, 0, params, st, 1, tag=f"g18b/{tag}/1")
    planes = [pop.get_state_dict(0, pl) for pl in range(3)]
    for key in planes[0]:
        if key.startswith("alphas") or (not bn and ".2." in key):
            continue
        want, got = _gold(g, f"{tag}/step1/p/{key}", planes[0][key].numpy())
        assert frac_bad(got, want, 1e-3 if "running" in key else 1e-4, 2e-6) <= (0.002 if got.size >= 1000 else 0.03), (tag, key)
    for key in O.trainable_keys(conf, ohp):
        for pl, nm, rt in ((1, "m", 2e-3), (2, "v", 4e-3)):
            want, got = _gold(g, f"{tag}/step1/{nm}/{key}", planes[pl][key].numpy())
            sc = float(np.abs(want).max()) + 1e-30
            assert frac_bad(got, want, rt, 2e-3 * sc) <= (0.01 if got.size >= 1000 else max(0.03, 2.0 / got.size)), (tag, nm, key)
    pop.close()
    # --- the one-step map at the reference's later states
    for k in (1, 2, 6, 9):
        state, _, _ = oracle_steps(conf, ohp, O.init_params(conf, ohp, 5), ttr, k, seed=40, order=order)
        if k in (1, 2):           # the oracle's state IS the reference's (golden W after k steps)
            for key in state:
                if key.startswith("alphas") or (not bn and ".2." in key):
                    continue
                want, got = _gold(g, f"{tag}/step{k}/p/{key}", state[key])
                assert frac_bad(got, want, 1e-3 if "running" in key else 1e-4, 2e-6 * k) <= 0.003, (tag, k, key)
        ep, bi = divmod(k, nb)
        rows = order[ep][bi * B:(bi + 1) * B]
        check_one_step_map(dev, ohp, conf, state, ttr, rows, 41 + k, f"g18b/{tag}/from{k}")
    # --- the whole trajectory
    pop = mk_pop(ohp, [conf], dev, drop_seeds=[40])
    pop.set_state_dict(0, O.init_params(conf, ohp, 5))
    stats, status = pop.train(ta, tb, 3, etas_for(ohp, N), order=dorder)
    assert not status.any()
    hist = []
    best = O.train_candidate(conf, ohp, O.init_params(conf, ohp, 5), ttr, tdv, order=order, seed=40, history=hist)
    assert best == pytest.approx(float(g[tag + "/best_acc"]), abs=1e-12)              # oracle == reference
    ghist = g[tag + "/hist"]
    for e, h in enumerate(hist):
        assert abs(h["dev_acc"] - ghist[2 * e + 1][2]) < 1e-4                          # oracle count == reference count
        ties = int((h["dev_margins"] < 5e-3).sum())
        assert ties <= 2, (tag, e, h["dev_margins"])
        assert abs(int(stats["dev_corrects"][0, e]) - h["dev_corrects"]) <= ties, (tag, e, int(stats["dev_corrects"][0, e]), h["dev_corrects"])
        # (epoch losses: 3e-3 relative — the reference run twice with different BLAS thread counts differs from ITSELF by 1e-2 in
        #  the per-step loss after ten steps, golden G13; the one-step maps above are where the arithmetic is pinned)
        assert abs(stats["train_loss_sum"][0, e] / N - ghist[2 * e][1]) < 3e-3 * max(1.0, ghist[2 * e][1])
        assert abs(stats["dev_loss_sum"][0, e] / Nd - ghist[2 * e + 1][1]) < 3e-3 * max(1.0, ghist[2 * e + 1][1])
        assert abs(stats["train_corrects"][0, e] / N - ghist[2 * e][2]) <= 2.0 / N + 1e-4     # train-mode predictions (masks on)
    pop.close()


def test_bf16_tables(dev):
    """bf16 storage of the taps (config 2): identical bf16-rounded values go to the oracle as f32."""
    ohp = O.Hyper(R=128, B=16, bn=True, drpt=0.0, epochs=2)
    conf = np.array(CONFS["c4"])
    ttr, tdv = O.synth_table(128, 5, snr=0.3, quant="bf16"), O.synth_table(96, 6, snr=0.3, quant="bf16")
    pop = mk_pop(ohp, [conf], dev)
    pop.set_state_dict(0, O.init_params(conf, ohp, 77))
    stats, _ = pop.train(table(ttr, dev, torch.bfloat16), table(tdv, dev, torch.bfloat16), 2, etas_for(ohp, 128))
    hist = []
    O.train_candidate(conf, ohp, O.init_params(conf, ohp, 77), ttr, tdv, history=hist)
    for e in range(2):
        assert abs(stats["train_loss_sum"][0, e] / 128 - hist[e]["train_loss"]) < 1e-3
        assert abs(stats["dev_corrects"][0, e] - hist[e]["dev_corrects"]) <= 1
    pop.close()


# ---------------------------------------------------------------------------------------- size-independent properties
def test_lockstep_independence_and_determinism(dev):
    """A candidate's result must not depend on who else is in the population, on its slot, or on the chunking."""
    ohp = O.Hyper(R=16, B=20, bn=False, drpt=0.5, epochs=2)
    confs = [np.array(CONFS[c]) for c in ("c4", "l1", "l2", "l3", "c0")]
    ttr, tdv = O.synth_table(500, 1, snr=1.0), O.synth_table(300, 2, snr=1.0)
    ta, tb = table(ttr, dev), table(tdv, dev)
    etas = etas_for(ohp, 500)

    def run(idx, chunk):
        pop = mk_pop(ohp, [confs[i] for i in idx], dev, drop_seeds=[200 + i for i in idx], chunk_cols=chunk)
        pop.init([300 + i for i in idx])
        stats, _ = pop.train(ta, tb, 2, etas)
        out = {i: (stats["dev_corrects"][j].tolist(), stats["train_loss_sum"][j].tolist()) for j, i in enumerate(idx)}
        pop.close()
        return out

    a = run([0, 1, 2, 3, 4], 0)
    import os
    os.environ["MFAS_GROUPS"] = "2"       # fused two-group schedule (chain of one group under the other's sweep)
    try:
        f = run([0, 1, 2, 3, 4], 0)
    finally:
        del os.environ["MFAS_GROUPS"]
    assert f == a                         # scheduling must not change a single bit
    os.environ["MFAS_FORCE_TAP_MAJOR"] = "1"   # small-R tap-major sweep (one feature chunk shared by several segments)
    try:
        t1 = run([0, 1, 2, 3, 4], 0)
        t2 = run([4, 2, 0], 0)
    finally:
        del os.environ["MFAS_FORCE_TAP_MAJOR"]
    for i in (0, 2, 4):
        assert t1[i] == t2[i], i          # grouping segments into workgroups does not change a candidate's arithmetic
    for i in range(5):                    # vs the per-segment path: summation order differs, results agree closely
        assert max(abs(x - y) for x, y in zip(a[i][0], t1[i][0])) <= 3
        assert max(abs(x - y) for x, y in zip(a[i][1], t1[i][1])) < 0.5
    b = run([4, 2, 0], 0)
    c = run([0, 1, 2, 3, 4], 0)
    for i in (0, 2, 4):
        assert a[i] == b[i], i          # bit-identical: same kernels, same order of operations
    assert a == c
    d = run([0, 1, 2, 3, 4], 64)        # different chunking changes summation order only
    for i in range(5):
        assert max(abs(x - y) for x, y in zip(a[i][0], d[i][0])) <= 3
        assert max(abs(x - y) for x, y in zip(a[i][1], d[i][1])) < 0.5


def test_error_behaviour(dev):
    from mfas_amd import Hyper, Population
    with pytest.raises(RuntimeError, match="illegal cell variant"):
        Population(Hyper(R=16, bn=False, drpt=0.0), [np.array(CONFS["l1"])], dev)
    with pytest.raises(RuntimeError):
        Population(Hyper(R=16, bn=True, drpt=0.0), [np.array([[5, 0, 0]])], dev)
    ohp = O.Hyper(R=16, B=16, bn=True, drpt=0.0, epochs=1)
    pop = mk_pop(ohp, [CONFS["l1"]], dev)
    pop.init([1])
    t17 = O.synth_table(17, 1)          # final train batch of size 1 with BN: the reference raises
    with pytest.raises(RuntimeError, match="size 1"):
        pop.train(table(t17, dev), table(t17, dev), 1, etas_for(ohp, 17))
    bad = O.synth_table(32, 1)
    bad["label"] = bad["label"].copy()
    bad["label"][5] = 60                # CrossEntropyLoss: "Target 60 is out of bounds"
    with pytest.raises(IndexError, match="out of bounds"):
        pop.train(table(bad, dev), None, 1, etas_for(ohp, 32))
    pop.close()


def test_baseline_config0_found_defaults(dev):
    """BASELINE configs[0]: main_found_ntu.py --conf 0, batch 16, found-script defaults (R=256, drpt 0.4, multitask on,
    no BN), head-only phase-1 semantics — engine vs oracle on identical tables (shared dropout stream)."""
    ohp = O.Hyper(R=256, B=16, bn=False, drpt=0.4, multitask=True, epochs=2, Ti=5)
    conf = np.array(CONFS["c0"])
    ttr = O.synth_table(160, 61, snr=0.5, with_logits=True)
    tdv = O.synth_table(96, 62, snr=0.5, with_logits=True)
    pop = mk_pop(ohp, [conf], dev, drop_seeds=[5])
    pop.set_state_dict(0, O.init_params(conf, ohp, 31))
    stats, status = pop.train(table(ttr, dev), table(tdv, dev), 2, etas_for(ohp, 160))
    hist = []
    O.train_candidate(conf, ohp, O.init_params(conf, ohp, 31), ttr, tdv, seed=5, history=hist)
    for e in range(2):
        assert abs(stats["train_loss_sum"][0, e] / 160 - hist[e]["train_loss"]) < 2e-3
        assert abs(stats["train_corrects"][0, e] - round(hist[e]["train_acc"] * 160)) <= 2
        assert abs(stats["dev_corrects"][0, e] - hist[e]["dev_corrects"]) <= 1
    assert not status.any()
    pop.close()


def test_odd_sizes_R24_B20(dev):
    """R not a multiple of 16 (padded row/column blocks must stay inert) with a ragged B=20 stream."""
    ohp = O.Hyper(R=24, B=20, bn=True, drpt=0.3, epochs=2)
    conf = np.array(CONFS["l3"])
    ttr, tdv = O.synth_table(130, 71, snr=0.5), O.synth_table(70, 72, snr=0.5)
    pop = mk_pop(ohp, [conf], dev, drop_seeds=[9])
    pop.set_state_dict(0, O.init_params(conf, ohp, 3))
    stats, _ = pop.train(table(ttr, dev), table(tdv, dev), 2, etas_for(ohp, 130))
    hist = []
    params = O.init_params(conf, ohp, 3)
    O.train_candidate(conf, ohp, params, ttr, tdv, seed=9, history=hist)
    for e in range(2):
        assert abs(stats["train_loss_sum"][0, e] / 130 - hist[e]["train_loss"]) < 2e-3
        assert abs(stats["dev_corrects"][0, e] - hist[e]["dev_corrects"]) <= 1
    got = pop.get_state_dict(0)
    for key, v in params.items():
        if key.startswith("alphas"):
            continue
        assert frac_bad(got[key].numpy(), v, 2e-3, 2e-5) <= 0.05, key
    pop.close()


def test_full_size_properties(dev):
    """BASELINE configs[1] at full size (conf 4, R=128, BN, drpt 0.5, B=16, N_train=10,000, N_dev=5,600, bf16 taps;
    E shortened to 2): size-independent properties instead of an oracle run —
    (1) the fused two-group schedule and the back-to-back schedule are bit-identical,
    (2) a candidate's trajectory does not depend on the rest of the population (lockstep independence),
    (3) identical seeds -> identical results; different dropout seeds -> different trajectories, same regime,
    (4) training works: loss falls, dev accuracy far above chance, counters are consistent."""
    import os
    from mfas_amd import FeatureTable, Hyper, Population
    hp = Hyper(R=128, C=60, B=16, bn=True, drpt=0.5)
    conf = np.array(CONFS["c4"])
    tr = FeatureTable.synthetic(10000, 1, dev, torch.bfloat16, snr=0.15)
    dv = FeatureTable.synthetic(5600, 2, dev, torch.bfloat16, snr=0.15)
    nb = 625
    etas = O.eta_sequence(1e-3, 1e-6, 1, 2, 10000 / 16, 2 * nb)
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    order = torch.stack([torch.randperm(10000, generator=g, device=dev) for _ in range(2)]).to(torch.int32)

    def run(K, groups, seeds, env=()):
        os.environ["MFAS_GROUPS"] = str(groups)
        for k_, v_ in env:
            os.environ[k_] = v_
        try:
            pop = Population(hp, [conf] * K, dev, drop_seeds=seeds, chunk_cols=128)   # same chunking = same summation order
        finally:
            del os.environ["MFAS_GROUPS"]
            for k_, _ in env:
                del os.environ[k_]
        pop.init([100 + s for s in seeds])
        stats, status = pop.train(tr, dv, 2, etas, order=order)
        pop.close()
        assert not status.any()
        return stats

    seeds = list(range(8))
    a = run(8, 2, seeds)
    b = run(8, 1, seeds)
    assert a.tobytes() == b.tobytes(), "fused vs back-to-back schedule differ"   # (1)
    # (1b, round 2) reduce-in-sweep (the last feature workgroup of a cell sums the partial slabs) vs the chain reducing them
    # itself: same bits
    assert run(8, 2, seeds, env=(("MFAS_NO_RED_IN_SWEEP", "1"),)).tobytes() == a.tobytes(), "reduce-in-sweep changes results"
    # (1c) `b` ran the same-group fused launch (one launch per step, units released per cell: the default at this size); the plain
    # two-launch schedule (k_chain, then the sweep) must give the same bits as well
    assert run(8, 1, seeds, env=(("MFAS_SAME_GROUP", "0"),)).tobytes() == a.tobytes(), "same-group fused launch differs"
    c = run(3, 1, [5, 2, 7])
    for j, s_ in enumerate([5, 2, 7]):
        assert c[j].tobytes() == a[s_].tobytes(), "population-dependent result"   # (2) + (3)
    assert len({a[k]["dev_corrects"][1] for k in range(8)}) > 1         # different seeds really differ
    for k in range(8):                                                  # (4)
        assert a[k]["train_loss_sum"][1] < a[k]["train_loss_sum"][0] < 10000 * np.log(60) * 1.05
        assert a[k]["dev_corrects"][1] > 0.5 * 5600
        assert 0 <= a[k]["train_corrects"][0] <= 10000 and 0 <= a[k]["dev_corrects"][0] <= 5600


@pytest.mark.parametrize("R,B,K,N,dt,mixed,alphas,max_steps", [
    (128, 16, 6, 16 * 5 + 7, "bf16", False, False, -1),     # ragged last batch
    (128, 20, 5, 43, "f32", True, True, -1),                # f32 table, mixed depths / taps, alphas
    (32, 16, 7, 16, "bf16", True, False, -1),               # one batch per epoch
    (64, 8, 4, 16, "f16", True, False, -1),                 # two batches per epoch
    (128, 16, 4, 80, "bf16", True, False, 7),               # cut by max_steps inside the second epoch
    (128, 16, 24, 10000, "bf16", False, False, -1),         # BASELINE configs[1]'s table size, 24 conf-4 candidates (two natural groups)
])
def test_gathered_rows_bit_identical(dev, capfd, R, B, K, N, dt, mixed, alphas, max_steps):
    """Two-group streaming schedule with per-candidate sample orders (the headline's): every candidate's OWN rows of the next batch
    are gathered into a contiguous per-candidate copy by the launch that carries its chain, and the feature units of the next
    launch stage x_t / x_{t+1} from that copy (sweep.hip.h, gather_body).  Against staging from the table through the order
    (MFAS_NO_GATHER=1) everything must be bit-identical: W, m, v of every candidate and all statistics."""
    import os
    from mfas_amd import FeatureTable, Hyper, Population
    rng = np.random.default_rng(R + N)
    E = 3
    hp = Hyper(R=R, C=60, B=B, bn=bool(R >= 64), drpt=0.5, alphas=alphas, tap_bits=32 if dt == "f32" else 16, order_per_candidate=True)
    confs = [np.array(CONFS["c4"])] * K
    if mixed:
        confs = [np.stack([rng.integers(0, 4, L), rng.integers(0, 4, L), rng.integers(0, 2, L)], 1) for L in rng.integers(1, 5, K)]
    tdt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[dt]
    if N >= 5000:
        E = 2
        tr = FeatureTable.synthetic(N, 1, dev, tdt, snr=0.15)
    else:
        tr = FeatureTable.from_numpy(O.synth_table(N, 3, snr=0.6), dev, tdt)
    dv = FeatureTable.from_numpy(O.synth_table(50, 4, snr=0.6), dev, tdt)
    nb = -(-N // B)
    etas = O.eta_sequence(1e-3, 1e-6, 1, 2, N / B, E * nb)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    order = torch.stack([torch.stack([torch.randperm(N, generator=g, device=dev) for _ in range(E)]) for _ in range(K)]).to(torch.int32)

    def run(gather):
        os.environ["MFAS_GROUPS"] = "2"
        if not gather:
            os.environ["MFAS_NO_GATHER"] = "1"
        try:
            pop = Population(hp, confs, dev, drop_seeds=list(range(70, 70 + K)))
            sched = pop.schedule()
            pop.init(list(range(1, K + 1)))
            stats, status = pop.train(tr, dv, E, etas, order=order, max_steps=max_steps)
            planes = [[pop.get_params(k, pl).cpu().numpy().tobytes() for pl in range(3)] for k in range(K)]
            pop.close()
        finally:
            del os.environ["MFAS_GROUPS"]
            os.environ.pop("MFAS_NO_GATHER", None)
        assert not status.any() and sched["groups"] == 2 and not sched["persistent"]
        return stats, planes

    os.environ["MFAS_GATHER_VERBOSE"] = "1"
    try:
        capfd.readouterr()
        s1, p1 = run(True)
        assert "[gather] on" in capfd.readouterr().err        # the path under test really ran
        s0, p0 = run(False)
        assert "[gather] on" not in capfd.readouterr().err
    finally:
        del os.environ["MFAS_GATHER_VERBOSE"]
    assert s1.tobytes() == s0.tobytes()
    assert p1 == p0
    assert (s0["train_loss_sum"] > 0).any()


def test_multi_chunk_units(dev, monkeypatch):
    """Round 4 (opt-in, MFAS_SUBCHUNKS=n): where the planner streams 64-column chunks (R = 128, >= 28 candidates) a sweep workgroup
    may take n consecutive chunks and keep the forward partial sums in registers across them (SegDesc::nsub, sweep_multi_body): one
    partial slab per UNIT.  Measured slower than one-chunk units on MI355X (profiles/r04_subchunks_preload_prio.log), so the default
    stays factor 1 — but the arithmetic is pinned: a unit of n chunks sums exactly like ONE chunk of n * 64 columns, so
    (1) factor 2 == chunk_cols 128 and factor 4 == chunk_cols 256 BIT FOR BIT (statistics of a whole run, parameters, train-mode
    forward logits and gradients of the single-batch entry points, which walk the merged units too); the default == factor 1 ==
    chunk_cols 64; (2) schedules on the SAME units stay bit-identical (fused two-group launches vs back-to-back); (3) a ragged
    factor (3: units of 3, 3, 2 chunks) trains to the same losses within the run-to-run spread of two chunk sizes."""
    from mfas_amd import FeatureTable, Hyper, Population
    hp = Hyper(R=128, C=60, B=16, bn=True, drpt=0.0)
    rng = np.random.default_rng(4)
    K = 30
    confs = [np.array(CONFS["c4"])] * 10 + [np.stack([rng.integers(0, 4, L), rng.integers(0, 4, L), rng.integers(0, 3, L)], 1) for L in rng.integers(1, 5, K - 10)]
    tr = FeatureTable.synthetic(480, 1, dev, torch.bfloat16, snr=0.3)
    dv = FeatureTable.synthetic(320, 2, dev, torch.bfloat16, snr=0.3)
    etas = O.eta_sequence(1e-3, 1e-6, 1, 2, 30.0, 60)

    def run(sub, groups=None, cc=0):
        for key, val in (("MFAS_SUBCHUNKS", sub), ("MFAS_GROUPS", groups)):
            if val is not None:
                monkeypatch.setenv(key, str(val))
            else:
                monkeypatch.delenv(key, raising=False)
        pop = Population(hp, confs, dev, drop_seeds=list(range(K)), chunk_cols=cc)
        pop.init(list(range(1, K + 1)))
        stats, status = pop.train(tr, dv, 2, etas)
        assert not status.any()
        w = [pop.get_params(k).cpu().numpy() for k in (0, K // 2, K - 1)]
        logits = pop.forward_train(3, tr, 0, 16, step=1).cpu().numpy()
        grad = pop.backward(3, tr, torch.full((16, 60), 0.01, device=dev), 0, 16, step=1).cpu().numpy()
        pop.close()
        return stats, w, logits, grad

    def same(a, b):
        return (a[0].tobytes() == b[0].tobytes() and all(np.array_equal(x, y) for x, y in zip(a[1], b[1]))
                and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]))

    base, dflt = run(1), run(None)
    assert same(dflt, base) and same(run(None, cc=64), base)              # the default: one 64-column chunk per unit
    two, four = run(2), run(4)
    assert same(two, run(None, cc=128)) and same(four, run(None, cc=256))    # (1) n chunks per unit == one chunk of n * 64 columns
    assert not same(two, base) and not same(four, two)
    assert run(4, groups=1)[0].tobytes() == run(4, groups=2)[0].tobytes() == four[0].tobytes()   # (2)
    three = run(3)                                                            # (3)
    spread = np.abs(four[0]["train_loss_sum"] - base[0]["train_loss_sum"]).max()
    assert np.abs(three[0]["train_loss_sum"] - base[0]["train_loss_sum"]).max() <= 3.0 * spread + 1.0
    assert np.abs(three[0]["dev_corrects"].astype(np.int64) - base[0]["dev_corrects"]).max() <= 3 * np.abs(four[0]["dev_corrects"].astype(np.int64) - base[0]["dev_corrects"]).max() + 8


@pytest.mark.parametrize("K,B,bn,cc,mixed", [(6, 20, False, 256, False), (9, 20, True, 128, True), (7, 16, True, 512, True), (16, 16, False, 1024, False)])
def test_persistent_resident_schedule_bit_identical_full_size(dev, K, B, bn, cc, mixed):
    """Search-script defaults at full size (R=16, N_train=10,000, N_dev=5,600, bf16 taps, drpt 0.5, shuffled, E=2): the
    persistent resident schedule (k_president: W/m/v in registers, resident lean chain, per-candidate flags — the default for
    small populations) against the launch-per-phase schedule on the same units: statistics, parameters and both Adam moments
    must be bit-identical; 1024-column units exercise the raw 16-bit staging."""
    import os
    from mfas_amd import FeatureTable, Hyper, Population
    hp = Hyper(R=16, C=60, B=B, bn=bn, drpt=0.5, alphas=mixed, tap_bits=16)
    rng = np.random.default_rng(5)
    confs = [np.array(CONFS["c4"])] * K
    if mixed:
        confs = [np.stack([rng.integers(0, 4, L), rng.integers(0, 4, L), rng.integers(0, 2, L)], 1) for L in rng.integers(1, 5, K)]
    tr = FeatureTable.synthetic(10000, 1, dev, torch.bfloat16, snr=0.5)
    dv = FeatureTable.synthetic(5600, 2, dev, torch.bfloat16, snr=0.5)
    nb = -(-10000 // B)
    etas = O.eta_sequence(1e-3, 1e-6, 1, 2, 10000 / B, 2 * nb)
    g = torch.Generator(device=dev)
    g.manual_seed(4)
    order = torch.stack([torch.randperm(10000, generator=g, device=dev) for _ in range(2)]).to(torch.int32)
    out = {}
    for mode in ("0", "1"):
        os.environ["MFAS_PERSIST"] = mode
        os.environ["MFAS_NO_TAP_MAJOR"] = "1"        # the persistent schedule runs per-segment units
        try:
            pop = Population(hp, confs, dev, drop_seeds=list(range(50, 50 + K)), chunk_cols=cc)
        finally:
            del os.environ["MFAS_PERSIST"], os.environ["MFAS_NO_TAP_MAJOR"]
        pop.init(list(range(1, K + 1)))
        stats, status = pop.train(tr, dv, 2, etas, order=order)
        assert not status.any()
        out[mode] = (stats, [[pop.get_params(k, pl).cpu().numpy() for pl in range(3)] for k in range(K)])
        pop.close()
    assert out["0"][0].tobytes() == out["1"][0].tobytes()
    for k in range(K):
        for pl in range(3):
            assert np.array_equal(out["0"][1][k][pl], out["1"][1][k][pl]), (k, pl)
    assert (out["1"][0]["dev_corrects"][:, 1] > 0.05 * 5600).all()      # and it trains (chance = 1.7 %)


@pytest.mark.parametrize("seed", range(20))
def test_persistent_schedule_fuzz_bit_identical(dev, seed):
    """Random small populations at R <= 16 (population size up to the resident capacity, mixed depths, B in 2..32, BN / alphas /
    dropout on or off, bf16 / f16 / f32 taps, ragged last batch, odd class counts, one or two units per workgroup, units up to
    1024 columns): the persistent schedule the engine picks by default against launch-per-phase on the same units — statistics,
    parameters and both Adam moments bit for bit."""
    import os
    from mfas_amd import FeatureTable, Hyper, Population
    rng = np.random.default_rng(1000 + seed)
    R = int(rng.choice([8, 16, 16, 16]))
    B = int(rng.choice([2, 7, 16, 20, 20, 32]))
    bn = bool(rng.integers(0, 2)) or False
    drpt = float(rng.choice([0.0, 0.5])) if bn else 0.5
    alphas = bool(rng.integers(0, 2))
    C = int(rng.choice([60, 23, 5]))
    dtype = [torch.bfloat16, torch.float16, torch.float32][int(rng.integers(0, 3))]
    cc = int(rng.choice([128, 256, 512, 1024])) if dtype != torch.float32 else int(rng.choice([128, 256]))
    K = int(rng.choice({128: [1, 3, 6], 256: [1, 4, 9, 12], 512: [2, 7, 16, 24], 1024: [3, 8, 17]}[cc]))   # around the resident capacity
    N = int(rng.integers(3 * B + 1, 9 * B))
    if N % B == 1:
        N += 1            # (a final batch of one sample is an error with BatchNorm, as in the reference)
    E = 2
    hp = Hyper(R=R, C=C, B=B, bn=bn, drpt=drpt, alphas=alphas, tap_bits=16 if dtype != torch.float32 else 32)
    confs = [np.stack([rng.integers(0, 4, L), rng.integers(0, 4, L), rng.integers(0, 3, L)], 1) for L in rng.integers(1, 5, K)]
    tr = FeatureTable.synthetic(N, 1 + seed, dev, dtype, snr=0.5, C=C)
    dv = FeatureTable.synthetic(2 * B + 3, 100 + seed, dev, dtype, snr=0.5, C=C)
    nb = -(-N // B)
    etas = O.eta_sequence(1e-3, 1e-6, 1, 2, N / B, E * nb)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    order = torch.stack([torch.randperm(N, generator=g, device=dev) for _ in range(E)]).to(torch.int32)
    out, sched = {}, {}
    for mode in ("0", "default"):
        os.environ["MFAS_NO_TAP_MAJOR"] = "1"
        if mode == "0":
            os.environ["MFAS_PERSIST"] = "0"
        try:
            try:
                pop = Population(hp, confs, dev, drop_seeds=list(range(9, 9 + K)), chunk_cols=cc)
            except RuntimeError as e:          # launch-per-phase LDS limit for this (B, cc): nothing to compare on this draw
                if "LDS" in str(e):
                    pytest.skip(str(e))
                raise
        finally:
            os.environ.pop("MFAS_PERSIST", None)
            del os.environ["MFAS_NO_TAP_MAJOR"]
        sched[mode] = pop.schedule()
        pop.init(list(range(1, K + 1)))
        stats, status = pop.train(tr, dv, E, etas, order=order)
        out[mode] = (stats, status, [[pop.get_params(k, pl).cpu().numpy() for pl in range(3)] for k in range(K)])
        pop.close()
    assert not sched["0"]["persistent"]
    assert out["0"][0].tobytes() == out["default"][0].tobytes(), (sched["default"], R, B, bn, alphas, C, K, cc)
    assert np.array_equal(out["0"][1], out["default"][1])
    for k in range(K):
        for pl in range(3):
            assert np.array_equal(out["0"][2][k][pl], out["default"][2][k][pl]), (k, pl, sched["default"])
    print("schedule:", sched["default"])


@pytest.mark.parametrize("seed", range(8))
def test_same_group_launch_fuzz_bit_identical(dev, seed):
    """Random small populations with the general chain (R = 32 / 64 / 128): the same-group fused launch — chain and sweep of the
    same candidates in ONE launch per step, sweep units released cell by cell by flags the backward pass publishes, OUT / HEAD
    units held back until the chain no longer reads the weights they overwrite — against the two-launch schedule: bit for bit."""
    import os
    from mfas_amd import FeatureTable, Hyper, Population
    rng = np.random.default_rng(2000 + seed)
    R = int(rng.choice([32, 64, 128]))
    B = int(rng.choice([5, 16, 20, 32]))
    bn = bool(rng.integers(0, 2))
    drpt = float(rng.choice([0.0, 0.5])) if bn else 0.5
    alphas = bool(rng.integers(0, 2))
    C = int(rng.choice([60, 23]))
    K = int(rng.choice([1, 2, 5, 8]))
    dtype = [torch.bfloat16, torch.float32][int(rng.integers(0, 2))]
    N = int(rng.integers(3 * B + 2, 8 * B))
    if N % B == 1:
        N += 1
    hp = Hyper(R=R, C=C, B=B, bn=bn, drpt=drpt, alphas=alphas)
    confs = [np.stack([rng.integers(0, 4, L), rng.integers(0, 4, L), rng.integers(0, 3, L)], 1) for L in rng.integers(1, 5, K)]
    tr = FeatureTable.synthetic(N, 1 + seed, dev, dtype, snr=0.5, C=C)
    dv = FeatureTable.synthetic(2 * B + 3, 100 + seed, dev, dtype, snr=0.5, C=C)
    nb = -(-N // B)
    etas = O.eta_sequence(1e-3, 1e-6, 1, 2, N / B, 2 * nb)
    out, sched = {}, {}
    for mode in ("two-launch", "default"):
        if mode == "two-launch":
            os.environ["MFAS_SAME_GROUP"] = "0"
        try:
            pop = Population(hp, confs, dev, drop_seeds=list(range(3, 3 + K)), chunk_cols=128)
        finally:
            os.environ.pop("MFAS_SAME_GROUP", None)
        sched[mode] = pop.schedule()
        pop.init(list(range(1, K + 1)))
        stats, status = pop.train(tr, dv, 2, etas)
        out[mode] = (stats, status, [[pop.get_params(k, pl).cpu().numpy() for pl in range(3)] for k in range(K)])
        pop.close()
    assert sched["default"]["groups"] == -1 and sched["two-launch"]["groups"] in (1, 2), sched     # (8 candidates: fused A/B launches)
    assert out["two-launch"][0].tobytes() == out["default"][0].tobytes(), (R, B, bn, alphas, C, K)
    assert np.array_equal(out["two-launch"][1], out["default"][1])
    for k in range(K):
        for pl in range(3):
            assert np.array_equal(out["two-launch"][2][k][pl], out["default"][2][k][pl]), (k, pl)


@pytest.mark.parametrize("seed", range(10))
def test_chain_split_bit_identical(dev, seed):
    """Round 6: at R = 113 .. 128 with one batch tile (B <= 16) the same-group launch runs every candidate's cell chain on FOUR CUs
    (chain.hip.h, chain_split: column split, 16 x 128 exchanges through sentinel-polled pieces, replicated head / softmax, per-cell
    arrival counters).  The arithmetic per row block is chain_body's, so against the one-CU chain (MFAS_CHAIN_SPLIT=0) and against the
    two-launch schedule everything must be bit-identical: statistics, W, m, v of every candidate — BatchNorm on / off, dropout on /
    off, mixed depths (1 .. 4 cells) and non-linearities, C = 60 / 23 / 7, R = 128 / 120 / 113 (padded columns), ragged last batch,
    bf16 / f32 tables, reduce-in-sweep on / off, 1 .. 10 candidates, a second train() call (parities restart), E = 2 epochs."""
    import os
    from mfas_amd import FeatureTable, Hyper, Population
    rng = np.random.default_rng(6000 + seed)
    R = int(rng.choice([128, 128, 120, 113]))
    B = int(rng.choice([16, 16, 12, 5]))
    bn = bool(rng.integers(0, 2)) or seed == 0
    drpt = float(rng.choice([0.0, 0.5])) if bn else 0.5
    C = int(rng.choice([60, 23, 7]))
    K = int(rng.choice([1, 2, 3, 6, 10]))
    dtype = [torch.bfloat16, torch.float32][int(rng.integers(0, 2))]
    N = int(rng.integers(3 * B + 2, 9 * B))
    if N % B == 1:
        N += 1
    no_red = bool(rng.integers(0, 2))
    hp = Hyper(R=R, C=C, B=B, bn=bn, drpt=drpt, alphas=False)
    confs = [np.stack([rng.integers(0, 4, L), rng.integers(0, 4, L), rng.integers(0, 3, L)], 1) for L in rng.integers(1, 5, K)]
    if seed == 0:
        confs = [np.array(CONFS["c4"])] * K
    tr = FeatureTable.synthetic(N, 1 + seed, dev, dtype, snr=0.5, C=C)
    dv = FeatureTable.synthetic(2 * B + 3, 100 + seed, dev, dtype, snr=0.5, C=C)
    nb = -(-N // B)
    etas = O.eta_sequence(1e-3, 1e-6, 1, 2, N / B, 2 * nb)
    out, sched = {}, {}
    for mode in ("two-launch", "one-cu", "split"):
        env = {"two-launch": {"MFAS_SAME_GROUP": "0"}, "one-cu": {"MFAS_CHAIN_SPLIT": "0"}, "split": {}}[mode]
        if no_red:
            env = dict(env, MFAS_NO_RED_IN_SWEEP="1")
        os.environ.update(env)
        try:
            pop = Population(hp, confs, dev, drop_seeds=list(range(3, 3 + K)), chunk_cols=128)
        finally:
            for k in env:
                os.environ.pop(k, None)
        sched[mode] = pop.schedule()
        pop.init(list(range(1, K + 1)))
        stats, status = pop.train(tr, dv, 2, etas)
        planes = [[pop.get_params(k, pl).cpu().numpy() for pl in range(3)] for k in range(K)]
        stats2, status2 = pop.train(tr, dv, 1, etas[:nb])           # a second call: fresh Adam state, exchange parities from 0 again
        out[mode] = (stats, status, planes, stats2, [pop.get_params(k, 0).cpu().numpy() for k in range(K)])
        assert not status.any() and not status2.any(), (mode, status, status2)
        pop.close()
    assert sched["split"]["groups"] == -1 and sched["one-cu"]["groups"] == -1 and sched["two-launch"]["groups"] in (1, 2), sched
    assert sched["split"]["chain_cus"] == 4 and sched["one-cu"]["chain_cus"] == 1, sched
    assert (out["split"][0]["train_loss_sum"] != 0).all()
    for ref in ("one-cu", "two-launch"):
        assert out[ref][0].tobytes() == out["split"][0].tobytes(), (ref, R, B, bn, C, K)
        assert out[ref][3].tobytes() == out["split"][3].tobytes(), (ref, "second call")
        for k in range(K):
            for pl in range(3):
                assert np.array_equal(out[ref][2][k][pl], out["split"][2][k][pl]), (ref, k, pl)
            assert np.array_equal(out[ref][4][k], out["split"][4][k]), (ref, k, "second call")


@pytest.mark.parametrize("mode", ["multitask", "multilabel"])
def test_chain_split_other_heads_bit_identical(dev, mode):
    """chain_split replicates the head and the loss on every part: the multitask head (summed-logit argmax, three-term loss: the
    16-lanes-per-row softmax, not the lean one) and the multi-label head (weighted BCE rows, F1-samples dev metric, loss_mode 1) at
    R = 128 must equal the one-CU chain bit for bit too — statistics incl. train loss sums, W, m, v."""
    import os
    from mfas_amd import FeatureTable, Hyper, Population
    rng = np.random.default_rng(77)
    K, B, N, C = 3, 16, 16 * 5 + 7, (60 if mode == "multitask" else 23)
    hp = Hyper(R=128, C=C, B=B, bn=True, drpt=0.5, multitask=(mode == "multitask"))
    if mode == "multilabel":
        hp.loss_mode, hp.f1_threshold = 1, 0.3
    confs = [np.stack([rng.integers(0, 4, L), rng.integers(0, 4, L), rng.integers(0, 3, L)], 1) for L in (4, 2, 3)]

    def table(n, seed):
        t = FeatureTable.synthetic(n, seed, dev, torch.bfloat16, snr=0.5, C=C, with_logits=(mode == "multitask"))
        if mode == "multilabel":
            g = torch.Generator(device=dev)
            g.manual_seed(seed)
            t = FeatureTable(t.taps, torch.zeros(n, dtype=torch.int32, device=dev), multilabel=(torch.rand(n, C, generator=g, device=dev) < 0.2).float())
        return t
    tr, dv = table(N, 3), table(40, 4)
    etas = O.eta_sequence(1e-3, 1e-6, 1, 2, N / B, 2 * 6)
    out = {}
    for split in ("0", None):
        if split is not None:
            os.environ["MFAS_CHAIN_SPLIT"] = split
        try:
            pop = Population(hp, confs, dev, drop_seeds=[5, 6, 7], chunk_cols=128)
        finally:
            os.environ.pop("MFAS_CHAIN_SPLIT", None)
        assert pop.schedule()["chain_cus"] == (1 if split == "0" else 4) and pop.schedule()["groups"] == -1
        if mode == "multilabel":
            pop.set_pos_weight(O.mm_pos_weight(C))
        pop.init([1, 2, 3])
        stats, status = pop.train(tr, dv, 2, etas)
        assert not status.any()
        out[split] = (stats.tobytes(), [[pop.get_params(k, pl).cpu().numpy() for pl in range(3)] for k in range(K)])
        assert (stats["train_loss_sum"] != 0).all()
        pop.close()
    assert out["0"][0] == out[None][0]
    for k in range(K):
        for pl in range(3):
            assert np.array_equal(out["0"][1][k][pl], out[None][1][k][pl]), (k, pl)


def _hooks_variant_loaded():
    from mfas_amd import _lib
    return _lib.tuning()["hooks"] == "1"


def test_hook_tests_run_on_the_hooks_variant():
    """The test hooks (MFAS_PERSIST_TEST_LOSE_STEP / _NOT_RESIDENT) are compiled only into the -DMFAS_TEST_HOOKS variant of the library
    (libmfas_hip_hooks.so); the product library never parses them.  The two tests that need them run here, in a subprocess that
    loads the variant through MFAS_LIB."""
    import os, subprocess, sys
    import __graft_entry__ as ge
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert not _hooks_variant_loaded() or os.environ.get("MFAS_LIB", "").endswith("_hooks.so")
    if _hooks_variant_loaded():
        pytest.skip("already inside the hooks-variant run")
    lib = ge.build_variant("hooks", ["-DMFAS_TEST_HOOKS"])
    res = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_parity.py"), "-x", "-q", "-m", "gpu", "-k",
                          "lost_dependency_ends_with_an_error or falls_back_to_launch_per_phase_when_never_resident"],
                         env=dict(os.environ, MFAS_LIB=lib), capture_output=True, text=True, timeout=1200, cwd=root)
    assert res.returncode == 0 and "2 passed" in res.stdout and "skipped" not in res.stdout, (res.stdout[-2000:], res.stderr[-2000:])


def test_persistent_loop_lost_dependency_ends_with_an_error(dev, monkeypatch):
    """The persistent step loop's waits are bounded: when a dependency never arrives (test hook: candidate 0's chain does not
    publish step 3) the launch ends by itself — the starved workgroups time out, set the abort word, everybody leaves — and
    train() reports an error instead of hanging the GPU; the device is usable afterwards."""
    from mfas_amd import FeatureTable, Hyper, Population
    if not _hooks_variant_loaded():
        pytest.skip("needs the -DMFAS_TEST_HOOKS variant (run by test_hook_tests_run_on_the_hooks_variant)")
    hp = Hyper(R=16, C=60, B=20, bn=False, drpt=0.5, tap_bits=16)
    confs = [np.array(CONFS["c4"])] * 4
    tr = FeatureTable.synthetic(400, 1, dev, torch.bfloat16, snr=0.5)
    dv = FeatureTable.synthetic(200, 2, dev, torch.bfloat16, snr=0.5)
    etas = O.eta_sequence(1e-3, 1e-6, 1, 2, 400 / 20, 20)

    def run():
        pop = Population(hp, confs, dev, drop_seeds=[1, 2, 3, 4])
        assert pop.schedule()["persistent"] and pop.schedule()["resident_chain"]
        pop.init([1, 2, 3, 4])
        try:
            return pop.train(tr, dv, 1, etas)
        finally:
            pop.close()

    monkeypatch.setenv("MFAS_PERSIST_TEST_LOSE_STEP", "3")
    with pytest.raises(RuntimeError, match="timed out"):
        run()
    monkeypatch.delenv("MFAS_PERSIST_TEST_LOSE_STEP")
    stats, status = run()
    assert not status.any() and (stats["train_corrects"] >= 0).all()


_SHARE_SCRIPT = r"""
import sys, hashlib, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from mfas_amd import FeatureTable, Hyper, Population
from oracle import np_oracle as O
dev = torch.device("cuda:0")
hp = Hyper(R=16, C=60, B=20, bn=False, drpt=0.5, tap_bits=16)
conf = np.array([[3, 1, 1], [1, 3, 0], [1, 1, 1], [3, 3, 0]])
K, E, N = 20, 10, 10000
tr = FeatureTable.synthetic(N, 1, dev, torch.bfloat16, snr=0.5)
dv = FeatureTable.synthetic(600, 2, dev, torch.bfloat16, snr=0.5)
etas = O.eta_sequence(1e-3, 1e-6, 1, 2, N / 20, E * (N // 20))
pop = Population(hp, [conf] * K, dev, drop_seeds=list(range(7, 7 + K)), chunk_cols=512)
assert pop.schedule()["persistent"] and pop.schedule()["resident_workgroups"] + K > 128      # two such grids cannot be co-resident
pop.init(list(range(1, K + 1)))
torch.cuda.synchronize()
if len(sys.argv) > 3:       # start together with the other process: <dir> <n processes>
    import os, time
    open(os.path.join(sys.argv[2], "ready_%d" % os.getpid()), "w").close()
    t0 = time.time()
    while len([f for f in os.listdir(sys.argv[2]) if f.startswith("ready_")]) < int(sys.argv[3]) and time.time() - t0 < 120:
        time.sleep(0.001)
stats, status = pop.train(tr, dv, E, etas)
assert not status.any()
print("DIGEST", hashlib.sha256(stats.tobytes()).hexdigest(), flush=True)
"""


def test_resident_schedule_falls_back_to_launch_per_phase_when_never_resident(dev, monkeypatch):
    """The resident schedule's two launches must be co-resident; when the roll call keeps failing (another tenant holds CUs for good,
    a CU mask, a tool that serialises launches) nothing of the epoch has run, and train() rebuilds the population in its
    launch-per-phase layout, carries W / m / v across and goes on (mfas_hip.hip::persist_fallback).  The hook
    MFAS_PERSIST_TEST_NOT_RESIDENT=e makes every roll call from epoch e on fail.  On the same unit decomposition (chunk_cols
    fixed, per-segment units) every schedule gives the same bits, so a run that switches after epoch 0 or 1 must equal the run
    that never switches — statistics, parameters and both Adam moments."""
    from mfas_amd import FeatureTable, Hyper, Population
    if not _hooks_variant_loaded():
        pytest.skip("needs the -DMFAS_TEST_HOOKS variant (run by test_hook_tests_run_on_the_hooks_variant)")
    hp = Hyper(R=16, C=60, B=20, bn=True, drpt=0.5, alphas=True, tap_bits=16)
    rng = np.random.default_rng(11)
    K = 5
    confs = [np.stack([rng.integers(0, 4, L), rng.integers(0, 4, L), rng.integers(0, 3, L)], 1) for L in (4, 2, 3, 1, 4)]
    tr = FeatureTable.synthetic(900, 1, dev, torch.bfloat16, snr=0.5)
    dv = FeatureTable.synthetic(300, 2, dev, torch.bfloat16, snr=0.5)
    E, nb = 3, 45
    etas = O.eta_sequence(1e-3, 1e-6, 1, 2, 900 / 20, E * nb)
    g = torch.Generator(device=dev)
    g.manual_seed(4)
    order = torch.stack([torch.randperm(900, generator=g, device=dev) for _ in range(E)]).to(torch.int32)
    monkeypatch.setenv("MFAS_NO_TAP_MAJOR", "1")

    def run(fail_from, snapshot):
        if fail_from is None:
            monkeypatch.delenv("MFAS_PERSIST_TEST_NOT_RESIDENT", raising=False)
        else:
            monkeypatch.setenv("MFAS_PERSIST_TEST_NOT_RESIDENT", str(fail_from))
        pop = Population(hp, confs, dev, drop_seeds=list(range(70, 70 + K)), chunk_cols=256)
        assert pop.schedule()["persistent"]
        pop.init(list(range(1, K + 1)))
        stats, status = pop.train(tr, dv, E, etas, order=order, snapshot_best=snapshot)
        planes = [[pop.get_params(k, pl).cpu().numpy() for pl in range(3)] for k in range(K)]
        sched = pop.schedule()
        pop.close()
        assert not status.any()
        return stats, planes, sched

    for snapshot in (False, True):
        ref_stats, ref_planes, ref_sched = run(None, snapshot)
        assert ref_sched["persistent"]
        for fail_from in (0, 1, 2):
            stats, planes, sched = run(fail_from, snapshot)
            assert not sched["persistent"], fail_from                      # the handle now holds the launch-per-phase layout
            assert stats.tobytes() == ref_stats.tobytes(), (snapshot, fail_from)
            for k in range(K):
                for pl in range(3):
                    assert np.array_equal(planes[k][pl], ref_planes[k][pl]), (snapshot, fail_from, k, pl)


def test_two_processes_share_the_gpu_with_persistent_grids(dev, tmp_path):
    """Two processes whose persistent grids (one workgroup per CU each, > half the chip) cannot be resident together train on
    the same GPU at the same time: the roll call at the start of every launch detects a partially resident grid before anything
    is modified and the host relaunches the epoch — both finish, with the results of a solo run."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "share.py"
    script.write_text(_SHARE_SCRIPT)

    def digest(out):
        return [l.split()[1] for l in out.splitlines() if l.startswith("DIGEST")]

    solo = subprocess.run([sys.executable, str(script), root], capture_output=True, text=True, timeout=300)
    assert solo.returncode == 0, solo.stderr[-2000:]
    env = dict(os.environ, MFAS_PERSIST_VERBOSE="1")
    procs = [subprocess.Popen([sys.executable, str(script), root, str(tmp_path), "2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
             for _ in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
        assert digest(o) == digest(solo.stdout) and len(digest(o)) == 1
    print("relaunched epochs per process:", [e.count("relaunched") for _, e in outs])     # (typically most of the 10)


@pytest.mark.parametrize("B,R", [(48, 16), (33, 128)])
def test_large_batch_paths(dev, B, R):
    """batch > 32 runs the MB=4 (64 padded rows) instantiation; 33 leaves the last 31 padded rows inert."""
    ohp = O.Hyper(R=R, B=B, bn=True, drpt=0.25, epochs=2)
    conf = np.array(CONFS["l2"])
    N = 2 * B + 7
    ttr, tdv = O.synth_table(N, 81, snr=0.5), O.synth_table(50, 82, snr=0.5)
    pop = mk_pop(ohp, [conf], dev, drop_seeds=[4])
    pop.set_state_dict(0, O.init_params(conf, ohp, 8))
    stats, status = pop.train(table(ttr, dev), table(tdv, dev), 2, etas_for(ohp, N))
    hist = []
    O.train_candidate(conf, ohp, O.init_params(conf, ohp, 8), ttr, tdv, seed=4, history=hist)
    for e in range(2):
        assert abs(stats["train_loss_sum"][0, e] / N - hist[e]["train_loss"]) < 2e-3
        assert abs(stats["dev_corrects"][0, e] - hist[e]["dev_corrects"]) <= 1
    assert not status.any()
    pop.close()


def test_tap_major_sweep_vs_oracle(dev):
    """Forced tap-major sweep (R=16 and R=32, mixed depths sharing taps) against the oracle trajectory."""
    import os
    for R in (16, 32):
        ohp = O.Hyper(R=R, B=20, bn=True, drpt=0.3, epochs=2)
        confs = [np.array(CONFS[c]) for c in ("c4", "c0", "l1", "l2", "l3", "c4")]
        ttr, tdv = O.synth_table(130, 91, snr=0.5), O.synth_table(70, 92, snr=0.5)
        os.environ["MFAS_FORCE_TAP_MAJOR"] = "1"
        try:
            pop = mk_pop(ohp, confs, dev, drop_seeds=list(range(20, 26)))
        finally:
            del os.environ["MFAS_FORCE_TAP_MAJOR"]
        for k, c in enumerate(confs):
            pop.set_state_dict(k, O.init_params(c, ohp, 60 + k))
        stats, status = pop.train(table(ttr, dev), table(tdv, dev), 2, etas_for(ohp, 130))
        for k, c in enumerate(confs):
            hist = []
            O.train_candidate(c, ohp, O.init_params(c, ohp, 60 + k), ttr, tdv, seed=20 + k, history=hist)
            for e in range(2):
                assert abs(stats["train_loss_sum"][k, e] / 130 - hist[e]["train_loss"]) < 2e-3, (R, k, e)
                assert abs(stats["dev_corrects"][k, e] - hist[e]["dev_corrects"]) <= 1, (R, k, e)
        assert not status.any()
        pop.close()


def test_nan_candidate_is_flagged_and_isolated(dev):
    """A candidate that diverges (NaN weights) raises its status flag; its neighbours' results are unchanged."""
    ohp = O.Hyper(R=16, B=16, bn=True, drpt=0.0, epochs=2)
    confs = [np.array(CONFS["l2"]), np.array(CONFS["l1"]), np.array(CONFS["l3"])]
    ttr, tdv = O.synth_table(64, 21, snr=0.3), O.synth_table(48, 22, snr=0.3)

    def run(poison):
        pop = mk_pop(ohp, confs, dev)
        for k, c in enumerate(confs):
            p = O.init_params(c, ohp, 5 + k)
            if poison and k == 1:
                p["fusion_layers.0.0.weight"][3, 7] = np.nan
            pop.set_state_dict(k, p)
        stats, status = pop.train(table(ttr, dev), table(tdv, dev), 2, etas_for(ohp, 64))
        pop.close()
        return stats, status

    clean, st0 = run(False)
    bad, st1 = run(True)
    assert st0.tolist() == [0, 0, 0] and st1.tolist() == [0, 1, 0]
    for k in (0, 2):
        assert clean[k].tobytes() == bad[k].tobytes()
    assert not np.isfinite(bad[1]["train_loss_sum"]).all()


def test_second_train_call_is_a_fresh_optimizer(dev):
    """Two consecutive train() calls on one population == the oracle trained twice with a new Adam each time
    (main_found_ntu.py builds a new optimizer + scheduler per phase, :108-137)."""
    ohp = O.Hyper(R=16, B=16, bn=True, drpt=0.0, epochs=1)
    conf = np.array(CONFS["l2"])
    ttr, tdv = O.synth_table(64, 21, snr=0.3), O.synth_table(48, 22, snr=0.3)
    pop = mk_pop(ohp, [conf], dev)
    pop.set_state_dict(0, O.init_params(conf, ohp, 5))
    ta, tb = table(ttr, dev), table(tdv, dev)
    s1, _ = pop.train(ta, tb, 1, etas_for(ohp, 64))
    s2, _ = pop.train(ta, tb, 1, etas_for(ohp, 64))
    params = O.init_params(conf, ohp, 5)
    h1, h2 = [], []
    O.train_candidate(conf, ohp, params, ttr, tdv, history=h1)
    O.train_candidate(conf, ohp, params, ttr, tdv, history=h2)      # continues from the trained weights, new Adam
    assert abs(s1["train_loss_sum"][0, 0] / 64 - h1[0]["train_loss"]) < 1e-3
    assert abs(s2["train_loss_sum"][0, 0] / 64 - h2[0]["train_loss"]) < 1e-3
    assert s2["dev_corrects"][0, 0] == h2[0]["dev_corrects"]
    pop.close()


def test_degenerate_shapes(dev):
    """A train set smaller than one batch, a dev set smaller than an eval row block, one candidate with one cell, and a
    population call that trains zero steps — against the oracle."""
    from mfas_amd import best_dev_accuracy
    ohp = O.Hyper(R=16, B=16, bn=True, drpt=0.0, epochs=2)
    conf = np.array(CONFS["l1"])
    ttr, tdv = O.synth_table(10, 81, snr=1.0), O.synth_table(5, 82, snr=1.0)      # 10 < B: every epoch is ONE ragged batch
    pop = mk_pop(ohp, [conf], dev)
    pop.set_state_dict(0, O.init_params(conf, ohp, 3))
    stats, status = pop.train(table(ttr, dev), table(tdv, dev), 2, etas_for(ohp, 10))
    hist = []
    want = O.train_candidate(conf, ohp, O.init_params(conf, ohp, 3), ttr, tdv, history=hist)
    for e in range(2):
        assert abs(stats["train_loss_sum"][0, e] / 10 - hist[e]["train_loss"]) < 1e-4
        assert stats["dev_corrects"][0, e] == hist[e]["dev_corrects"]
    assert best_dev_accuracy(stats[0], 5) == pytest.approx(want, abs=1e-12)
    assert not status.any()
    # zero train steps: parameters untouched, statistics zero
    before = pop.get_params(0).clone()
    stats, status = pop.train(table(ttr, dev), None, 1, etas_for(ohp, 10), max_steps=0)
    assert torch.equal(before, pop.get_params(0)) and stats["train_loss_sum"].sum() == 0 and stats["train_corrects"].sum() == 0
    pop.close()


@pytest.mark.gpu
def test_written_out_adam_arithmetic_equals_the_library_forms(dev, tmp_path):
    """common.hip.h writes Adam's sqrt and divisions out as packed fma sequences on the hardware rsq / rcp seeds.  tools/adam_exact.hip
    runs them next to sqrtf() / operator/ on the GPU: the square root over EVERY finite non-negative float (no difference allowed for
    x >= 2^-102), the whole update on 7 classes of 8 M sampled optimizer states (training-like, decayed, deep underflow, zeros, first
    step, wide log-uniform: no differing bit of w, m or v; packed == scalar)."""
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available on this box")
    exe = str(tmp_path / "adam_exact")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(root, "include"),
                    "-I" + os.path.join(root, "mfas_amd", "csrc"), os.path.join(root, "tools", "adam_exact.hip"), "-o", exe],
                   check=True, capture_output=True, timeout=600)
    out = subprocess.run([exe, "8"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "IDENTICAL" in out.stdout and "MISMATCH" not in out.stdout, out.stdout
    first = out.stdout.splitlines()[0]
    assert " 0 differ at x >= 2^-102" in first, first

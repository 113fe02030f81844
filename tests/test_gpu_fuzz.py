"""Seeded random sweep over the engine's configuration space against the numpy oracle: a few train steps of random
populations (R, batch size, classes, depth, taps, activations, BN / dropout / alphas / multitask, ragged last batch, odd tap
widths).  Complements the fixed cases of test_gpu_parity.py: every kernel variant (general / lean chain, per-segment /
tap-major sweep, B <= 16 / <= 32 builds, fused / unfused schedule) is hit with feature combinations the fixed tests do not
pair up."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
from oracle import np_oracle as O
from tests.helpers import engine_hyper, etas_for, oracle_steps, rel_err
from tests.test_gpu_parity import check_state, mk_pop, table

pytestmark = pytest.mark.gpu


def random_case(rng):
    R = int(rng.choice([8, 16, 16, 16, 24, 32, 64, 128]))
    B = int(rng.choice([5, 8, 16, 16, 20, 20, 27, 32]))
    C = int(rng.choice([7, 23, 60, 60]))
    bn = bool(rng.integers(0, 2))
    drpt = float(rng.choice([0.0, 0.3, 0.5])) if bn else float(rng.choice([0.3, 0.5]))
    widths = [16, 24, 40, 64, 100, 128, 200, 256]
    s_sizes = tuple(int(x) for x in rng.choice(widths, 4))
    v_sizes = tuple(int(x) for x in rng.choice(widths, 4))
    hp = O.Hyper(R=R, C=C, B=B, bn=bn, drpt=drpt, alphas=bool(rng.integers(0, 4) == 0),
                 multitask=bool(rng.integers(0, 4) == 0), epochs=2, s_sizes=s_sizes, v_sizes=v_sizes)
    K = int(rng.choice([1, 2, 3, 5]))
    confs = []
    for _ in range(K):
        L = int(rng.integers(1, 5))
        confs.append(np.stack([rng.integers(0, 4, L), rng.integers(0, 4, L), rng.integers(0, 2, L)], 1))
    N = int(B * rng.integers(2, 4) + rng.integers(0, B))     # ragged last batch most of the time
    if N % B == 1:
        N += 1                                                # (a 1-row BatchNorm batch raises in the reference)
    return hp, confs, N


@pytest.mark.parametrize("case", range(48))
def test_random_population_steps(case):
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(7000 + case)
    hp, confs, N = random_case(rng)
    t = O.synth_table(N, 300 + case, snr=0.4, C=hp.C, s_sizes=hp.s_sizes, v_sizes=hp.v_sizes, with_logits=hp.multitask)
    seeds = [int(s) for s in rng.integers(1, 1 << 20, len(confs))]
    pop = mk_pop(hp, confs, dev, drop_seeds=seeds)
    for k, c in enumerate(confs):
        pop.set_state_dict(k, O.init_params(c, hp, 900 + 10 * case + k, perturb_bn=True))
    steps = 4
    stats, status = pop.train(table(t, dev), None, 2, etas_for(hp, N), max_steps=steps)
    assert not status.any()
    for k, c in enumerate(confs):
        params, st, losses = oracle_steps(c, hp, O.init_params(c, hp, 900 + 10 * case + k, perturb_bn=True), t, steps, seed=seeds[k])
        check_state(pop, k, params, st, steps, tag=f"case{case}/cand{k}/R{hp.R}/B{hp.B}/C{hp.C}/bn{hp.bn}/d{hp.drpt}/a{hp.alphas}/m{hp.multitask}")
    # eval-mode forward (k_eval) of the trained state on a ragged row range, against the oracle fed the ENGINE's parameters
    tab = table(t, dev)
    for k, c in enumerate(confs):
        P = {key: v.numpy() for key, v in pop.get_state_dict(k).items()}
        row0, nrows = int(rng.integers(0, 3)), int(N - 3)
        logits, corr = pop.forward(k, tab, row0=row0, nrows=nrows, count=True)
        feats = {key: v[row0:row0 + nrows] for key, v in t.items() if key != "label"}
        want, _ = O.forward(P, c, hp, feats, False)
        assert rel_err(logits.cpu().numpy(), want) < 2e-4, (case, k)
        pred = (want + feats["vlogit"] + feats["slogit"]).argmax(1) if hp.multitask else want.argmax(1)
        assert abs(corr - int((pred == t["label"][row0:row0 + nrows]).sum())) <= 1, (case, k)
    pop.close()


@pytest.mark.parametrize("R,B,K", [(128, 16, 24), (16, 20, 48), (16, 16, 230), (128, 20, 22), (128, 16, 30), (256, 16, 28), (128, 20, 29)])
def test_natural_schedules_vs_oracle(R, B, K):
    """Populations large enough to take the fused two-group schedule (general chain from 20 candidates, lean chain for
    40..223) and the unfused large-population schedule, with the tap-major sweep where it applies: a sample of candidates
    against the oracle after a few steps.  The cases from 28 candidates at R >= 128 are the sizes at which the opt-in multi-chunk sweep
    units apply (MFAS_SUBCHUNKS, test_gpu_parity.py::test_multi_chunk_units); by default they run one-chunk units, R = 256 with two row
    blocks per wave."""
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(R * 1000 + K)
    # (BatchNorm WITH dropout on ~20-row batches can hit a column whose batch variance is ~0: then 1/sqrt(var + eps)
    # amplifies round-off so much that the oracle disagrees with ITSELF under a 1e-7 perturbation of the start — seen for
    # R=128, B=20, 22 candidates; such instances say nothing about the engine, so BN runs without dropout here)
    hp = O.Hyper(R=R, B=B, bn=bool(R == 128), drpt=0.0 if R == 128 else 0.5, epochs=2)
    confs = [np.stack([rng.integers(0, 4, L), rng.integers(0, 4, L), rng.integers(0, 2, L)], 1) for L in rng.integers(1, 5, K)]
    N = 3 * B + 7
    t = O.synth_table(N, 55, snr=0.4)
    seeds = [int(s) for s in rng.integers(1, 1 << 20, K)]
    pop = mk_pop(hp, confs, dev, drop_seeds=seeds)
    pop.init([700 + k for k in range(K)])
    steps = 5
    stats, status = pop.train(table(t, dev), None, 2, etas_for(hp, N), max_steps=steps)
    assert not status.any()
    for k in sorted(set([0, 1, K // 2 - 1, K // 2, K - 1])):
        params, st, _ = oracle_steps(confs[k], hp, O.init_params(confs[k], hp, 700 + k), t, steps, seed=seeds[k])
        check_state(pop, k, params, st, steps, tag=f"R{R}/B{B}/K{K}/cand{k}")
    pop.close()

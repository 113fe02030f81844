"""One-off sweep behind tests/test_gpu_mirror.py::test_bf16x3_eval_products_match_f32_products: random R in 72..128, tap widths,
depths, dev sizes, batchnorm / alphas / multitask — the dev pass with exact bf16 x 3 products against the f32-product build of the
same kernel (MFAS_EVAL_NO_B3=1).  usage: fuzz_eval_b3.py [cases] [seed0]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mfas_amd as M
from oracle import np_oracle as O

ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda:0")
worst = 0.0
for case in range(ncase):
    rng = np.random.default_rng(seed0 + case)
    R = int(rng.choice([72, 80, 96, 100, 112, 128, 128, 128]))
    widths = [64, 80, 96, 128, 160, 208, 256, 320, 512, 1024]
    s_sizes = tuple(int(x) for x in rng.choice(widths, 4))
    v_sizes = tuple(int(x) for x in rng.choice(widths, 4))
    bn, alphas, multitask = bool(rng.integers(0, 2)), bool(rng.integers(0, 3) == 0), bool(rng.integers(0, 4) == 0)
    B = int(rng.choice([8, 16, 20, 32]))
    C = int(rng.choice([7, 23, 60]))
    K = int(rng.integers(1, 7))
    hp = M.Hyper(R=R, C=C, B=B, bn=bn, drpt=0.5 if not bn else float(rng.choice([0.0, 0.5])), alphas=alphas, multitask=multitask, tap_bits=16,
                 s_sizes=s_sizes, v_sizes=v_sizes)
    confs = [np.stack([rng.integers(0, 4, L), rng.integers(0, 4, L), rng.integers(0, 2, L)], 1) for L in rng.integers(1, 5, K)]
    N, Nd, E = 4 * B, int(rng.integers(65, 900)), 2
    ta = M.FeatureTable.from_numpy(O.synth_table(N, 3 + case, snr=0.6, C=C, s_sizes=s_sizes, v_sizes=v_sizes, with_logits=multitask), dev, torch.bfloat16)
    tb = M.FeatureTable.from_numpy(O.synth_table(Nd, 4 + case, snr=0.6, C=C, s_sizes=s_sizes, v_sizes=v_sizes, with_logits=multitask), dev, torch.bfloat16)
    etas = O.eta_sequence(1e-3, 1e-6, 1, 2, N / B, E * (-(-N // B)))

    def run(f32_products):
        if f32_products:
            os.environ["MFAS_EVAL_NO_B3"] = "1"
        try:
            pop = M.Population(hp, confs, dev, drop_seeds=list(range(40, 40 + K)))
            pop.init(list(range(1, K + 1)))
            stats, status = pop.train(ta, tb, E, etas)
            row0 = int(rng.integers(0, 3)) if False else 1
            logits = [pop.forward(k, tb, row0=row0, nrows=Nd - 3).cpu().numpy() for k in range(K)]
            pop.close()
        finally:
            os.environ.pop("MFAS_EVAL_NO_B3", None)
        assert not status.any()
        return stats, logits

    (s_new, l_new), (s_old, l_old) = run(False), run(True)
    assert s_new["train_loss_sum"].tobytes() == s_old["train_loss_sum"].tobytes(), case
    dc = int(np.abs(s_new["dev_corrects"] - s_old["dev_corrects"]).max())
    dl = float(np.abs(s_new["dev_loss_sum"] / s_old["dev_loss_sum"] - 1).max())
    dg = max(float(np.abs(a - b).max() / max(1.0, np.abs(b).max())) for a, b in zip(l_new, l_old))
    worst = max(worst, dg)
    print(f"case {case:3d} R={R:3d} B={B:2d} C={C:2d} K={K} Nd={Nd:3d} bn={int(bn)} a={int(alphas)} m={int(multitask)} s={s_sizes} v={v_sizes}: "
          f"dcorr {dc} dloss {dl:.1e} dlogit {dg:.1e}", flush=True)
    assert dc <= 1 and dl < 1e-5 and dg < 2e-5, case
print("all", ncase, "cases ok; worst relative logit difference", worst)

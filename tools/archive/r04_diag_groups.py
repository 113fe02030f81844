"""Diagnostic: how far apart are two decompositions of the same training run after 3 steps?  (multi-chunk units vs chunk sizes)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mfas_amd import FeatureTable, Hyper, Population
from oracle import np_oracle as O
from tests.helpers import CONFS, frac_bad
dev = torch.device("cuda:0")
hp = Hyper(R=128, C=60, B=16, bn=True, drpt=0.0)
rng = np.random.default_rng(4)
K = 30
confs = [np.array(CONFS["c4"])] * 10 + [np.stack([rng.integers(0, 4, L), rng.integers(0, 4, L), rng.integers(0, 3, L)], 1) for L in rng.integers(1, 5, K - 10)]
tr = FeatureTable.synthetic(480, 1, dev, torch.bfloat16, snr=0.3)
etas = O.eta_sequence(1e-3, 1e-6, 1, 2, 30.0, 60)
def run(sub=None, cc=0, steps=3):
    if sub is None: os.environ.pop("MFAS_SUBCHUNKS", None)
    else: os.environ["MFAS_SUBCHUNKS"] = str(sub)
    pop = Population(hp, confs, dev, drop_seeds=list(range(K)), chunk_cols=cc)
    pop.init(list(range(1, K + 1)))
    stats, status = pop.train(tr, None, 2, etas, max_steps=steps)
    w = [pop.get_params(k).cpu().numpy() for k in range(K)]
    m = [pop.get_params(k, 1).cpu().numpy() for k in range(K)]
    pop.close()
    return stats, w, m
for steps in (1, 2, 3):
    base = run(1, steps=steps)
    print(f"--- {steps} step(s); reference = one-chunk units of 64 columns")
    for name, kw in (("again", dict(sub=1)), ("cc=128", dict(cc=128)), ("cc=256", dict(cc=256)), ("sub=2", dict(sub=2)), ("sub=3", dict(sub=3)), ("sub=4", dict(sub=4)), ("sub=16", dict(sub=16))):
        got = run(steps=steps, **kw)
        fb = [frac_bad(a, b, 1e-4, 6e-6) for a, b in zip(got[1], base[1])]
        fm = [frac_bad(a, b, 1e-3, 1e-9) for a, b in zip(got[2], base[2])]
        mx = max(float(np.abs(a - b).max()) for a, b in zip(got[1], base[1]))
        print(f"{name:8s} w bulk>1e-4: max over cands {max(fb):.4f} mean {np.mean(fb):.4f} | m(grad) >1e-3: max {max(fm):.4f} mean {np.mean(fm):.4f} | max|dw| {mx:.2e} | loss rel diff {np.abs(got[0]['train_loss_sum']/base[0]['train_loss_sum']-1).max():.2e}")

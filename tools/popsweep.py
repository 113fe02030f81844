"""Small-population sweep: candidates/s of one train_sampled_models-sized job (E epochs over N_train / N_dev) for K candidates on
one GPU, launch-per-phase schedule (MFAS_PERSIST=0) vs the default policy (persistent resident step loop where it fits), each with its own default unit
decomposition.  usage: popsweep.py R B bn E K1,K2,... [mixed] [N_train N_dev]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mfas_amd as M
from oracle import np_oracle as O

R, B, bn, E = (int(x) for x in sys.argv[1:5])
Ks = [int(x) for x in sys.argv[5].split(",")]
mixed = "mixed" in sys.argv
cc = next((int(a.split("=")[1]) for a in sys.argv if a.startswith("cc=")), 0)
nums = [int(a) for a in sys.argv[6:] if a.isdigit()]
E_show = E
N, Nd = (nums + [10000, 5600])[:2] if len(nums) >= 2 else (10000, 5600)
dev = torch.device("cuda:0")
tr = M.FeatureTable.synthetic(N, 1, dev, torch.bfloat16, snr=0.12)
dv = M.FeatureTable.synthetic(Nd, 2, dev, torch.bfloat16, snr=0.12)
hp = M.Hyper(R=R, B=B, bn=bool(bn), drpt=0.5, tap_bits=16)
conf4 = np.array([[3, 1, 1], [1, 3, 0], [1, 1, 1], [3, 3, 0]])
nb = -(-N // B)
etas = O.eta_sequence(1e-3, 1e-6, 1, 2, N / B, E * nb)
order = M.ntu_searchable.make_order(N, E, True, 5, dev)
print(f"# R={R} B={B} bn={bn} E={E} N={N}/{Nd} {'mixed L=1..4 confs' if mixed else 'conf 4'}: K, cand/s launch-per-phase, cand/s persistent, us/step each, ratio")
for K in Ks:
    confs = [conf4] * K
    if mixed:
        rng = np.random.default_rng(0)
        confs = [np.stack([rng.integers(0, 4, L), rng.integers(0, 4, L), rng.integers(0, 2, L)], 1) for L in rng.integers(1, 5, K)]
    out = {}
    for mode in ("0", "1"):          # "1" = the engine's default policy (persistent where the resident form fits), "0" = forced off
        if mode == "0":
            os.environ["MFAS_PERSIST"] = "0"
        else:
            os.environ.pop("MFAS_PERSIST", None)
        best = None
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            try:
                pop = M.Population(hp, confs, dev, drop_seeds=list(range(100, 100 + K)), chunk_cols=cc)
                pop.init(list(range(1, K + 1)))
                stats, status = pop.train(tr, dv, E, etas, order=order)
                pop.close()
            except RuntimeError as e:
                print("  ", K, mode, "failed:", str(e)[:120])
                best = float("nan")
                break
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        out[mode] = best
    print(f"K={K:4d}  launch {K / out['0']:8.2f} cand/s ({out['0'] / (E * nb) * 1e6:6.1f} us/step)   default {K / out['1']:8.2f} cand/s "
          f"({out['1'] / (E * nb) * 1e6:6.1f} us/step)   x{out['0'] / out['1']:.2f}", flush=True)

"""Dev-evaluation pass (k_eval) alone: K conf-4 candidates at R over N_dev bf16 rows; E epochs of ONE 16-row train step + the
dev pass, so the wall time is the evaluation's.  usage: evalbench.py R K [N_dev] ; environment: MFAS_EVAL_NO_B3"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mfas_amd as M
from oracle import np_oracle as O

R, K = int(sys.argv[1]), int(sys.argv[2])
Nd = int(sys.argv[3]) if len(sys.argv) > 3 else 5600
B, E = 16, 6
dev = torch.device("cuda:0")
tr = M.FeatureTable.synthetic(B, 1, dev, torch.bfloat16, snr=0.12)
dv = M.FeatureTable.synthetic(Nd, 2, dev, torch.bfloat16, snr=0.12)
hp = M.Hyper(R=R, B=B, bn=True, drpt=0.5, tap_bits=16)
conf4 = np.array([[3, 1, 1], [1, 3, 0], [1, 1, 1], [3, 3, 0]])
etas = O.eta_sequence(1e-3, 1e-6, 1, 2, 1, E)
pop = M.Population(hp, [conf4] * K, dev, drop_seeds=list(range(100, 100 + K)))
pop.init(list(range(1, K + 1)))
best = None
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stats, status = pop.train(tr, dv, E, etas)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / E
    best = dt if best is None else min(best, dt)
flops = 2.0 * K * Nd * (7680 + 3 * R) * R
print(f"R={R} K={K} N_dev={Nd} env B3off={os.environ.get('MFAS_EVAL_NO_B3', '0')}: "
      f"{best * 1e3:8.3f} ms per epoch (one step + dev pass) = {flops / best / 1e12:6.1f} TFLOP/s f32-equivalent", flush=True)
print("  dev_corr, dev_loss_sum of candidates 0..3, last epoch:", [(int(stats[k, -1]["dev_corrects"]), float(stats[k, -1]["dev_loss_sum"])) for k in range(min(4, K))])

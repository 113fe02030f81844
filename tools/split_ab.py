"""R = 128 small populations: the one-CU chain (MFAS_CHAIN_SPLIT=0) against the default (chain_split over 4 CUs where it applies).
usage: split_ab.py K1,K2,... [E] [B]   — conf 4, R=128, BN, drpt 0.5, N = 10,000 / 5,600, bf16 taps, per-call wall time of train()."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mfas_amd as M
from oracle import np_oracle as O

Ks = [int(x) for x in sys.argv[1].split(",")]
E = int(sys.argv[2]) if len(sys.argv) > 2 else 2
B = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else 16
cc = next((int(a.split("=")[1]) for a in sys.argv if a.startswith("cc=")), 0)
N, Nd = 10000, 5600
dev = torch.device("cuda:0")
tr = M.FeatureTable.synthetic(N, 1, dev, torch.bfloat16, snr=0.12)
dv = M.FeatureTable.synthetic(Nd, 2, dev, torch.bfloat16, snr=0.12)
hp = M.Hyper(R=128, B=B, bn=True, drpt=0.5, tap_bits=16)
conf4 = np.array([[3, 1, 1], [1, 3, 0], [1, 1, 1], [3, 3, 0]])
nb = -(-N // B)
etas = O.eta_sequence(1e-3, 1e-6, 1, 2, N / B, E * nb)
order = M.ntu_searchable.make_order(N, E, True, 5, dev)
print(f"# R=128 B={B} bn E={E} N={N}/{Nd} conf 4 chunk_cols={cc or 'auto'}: K | one-CU chain us/step (cand/s at E=10) | default us/step (cand/s) | schedule")
for K in Ks:
    res = {}
    for mode in ("0", "default"):
        if mode == "0":
            os.environ["MFAS_CHAIN_SPLIT"] = "0"
        try:
            pop = M.Population(hp, [conf4] * K, dev, drop_seeds=list(range(100, 100 + K)), chunk_cols=cc)
        finally:
            os.environ.pop("MFAS_CHAIN_SPLIT", None)
        sched = pop.schedule()
        pop.init(list(range(1, K + 1)))
        best = None
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            stats, status = pop.train(tr, dv, E, etas, order=order)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
            assert not status.any(), status
        res[mode] = (best, sched, stats.tobytes())
        pop.close()
    same = res["0"][2] == res["default"][2]
    us0, us1 = res["0"][0] / (E * nb) * 1e6, res["default"][0] / (E * nb) * 1e6
    print(f"K={K:3d}  one-CU {us0:6.1f} us/step ({K / (res['0'][0] * 10 / E):6.2f} cand/s)   default {us1:6.1f} us/step ({K / (res['default'][0] * 10 / E):6.2f} cand/s)"
          f"   x{us0 / us1:.2f}  chain_cus={res['default'][1]['chain_cus']} groups={res['default'][1]['groups']} bit-identical={same}", flush=True)

"""Persistent resident step loop (the default where it fits; MFAS_PERSIST=0 turns it off) vs the launch-per-phase schedule on the same small population: results must be
bit-identical (statistics and every parameter / Adam moment); prints cand/s of both.
usage: persist_check.py R B bn K E [N_train N_dev] [mixed] [cc=COLS] [steps=N] [toggle=ENV [persist]]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mfas_amd as M
from oracle import np_oracle as O

R, B, bn, K, E = (int(x) for x in sys.argv[1:6])
N, Nd = (int(sys.argv[6]), int(sys.argv[7])) if len(sys.argv) > 7 else (10000, 5600)
mixed = "mixed" in sys.argv
cc = next((int(a.split("=")[1]) for a in sys.argv if a.startswith("cc=")), 0)   # feature-column chunk (same units in both schedules)
os.environ["MFAS_NO_TAP_MAJOR"] = "1"      # the persistent schedule runs per-segment units: compare like with like
steps = next((int(a.split("=")[1]) for a in sys.argv if a.startswith("steps=")), -1)   # debug: stop after N train steps
dev = torch.device("cuda:0")
if "sidestream" in sys.argv:      # the population's stream = a created stream instead of the legacy default stream
    torch.cuda.set_stream(torch.cuda.Stream())
tr = M.FeatureTable.synthetic(N, 1, dev, torch.bfloat16, snr=0.12)
dv = M.FeatureTable.synthetic(Nd, 2, dev, torch.bfloat16, snr=0.12)
hp = M.Hyper(R=R, B=B, bn=bool(bn), drpt=0.5, alphas="alphas" in sys.argv, tap_bits=16)
conf4 = np.array([[3, 1, 1], [1, 3, 0], [1, 1, 1], [3, 3, 0]])
confs = [conf4] * K
if mixed:
    rng = np.random.default_rng(0)
    confs = [np.stack([rng.integers(0, 4, L), rng.integers(0, 4, L), rng.integers(0, 2, L)], 1) for L in rng.integers(1, 5, K)]
nb = -(-N // B)
etas = O.eta_sequence(1e-3, 1e-6, 1, 2, N / B, E * nb)
order = M.ntu_searchable.make_order(N, E, True, 5, dev)
toggle = next((a.split("=")[1] for a in sys.argv if a.startswith("toggle=")), None)   # compare ENV=1 ("0") against unset ("1") instead
res = {}
for mode in ("0", "1", "0", "1"):
    if toggle:
        if "persist" in sys.argv:                     # toggle under the default (resident where it fits) schedule ...
            os.environ.pop("MFAS_PERSIST", None)
        else:                                         # ... or under launch-per-phase
            os.environ["MFAS_PERSIST"] = "0"
        if mode == "0":
            os.environ[toggle] = "1"
        else:
            os.environ.pop(toggle, None)
    else:
        os.environ["MFAS_PERSIST"] = mode
    pop = M.Population(hp, confs, dev, drop_seeds=list(range(100, 100 + K)), chunk_cols=cc)
    pop.init(list(range(1, K + 1)))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stats, status = pop.train(tr, dv if steps < 0 else None, E, etas, order=order, max_steps=steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    params = [[pop.get_params(k, pl).cpu().numpy() for pl in range(3)] for k in range(K)]
    pop.close()
    print(f"R={R} B={B} bn={bn} K={K} E={E} N={N} persist={mode}: {dt * 1e3:.1f} ms  {K / dt:.2f} cand/s  "
          f"{dt / (E * nb) * 1e6:.1f} us/step  acc {np.mean([M.best_dev_accuracy(s, Nd) for s in stats]):.4f} status {status.tolist()}", flush=True)
    if mode in res:
        continue
    res[mode] = (stats, params)
s0, p0 = res["0"]
s1, p1 = res["1"]
same = all(np.array_equal(s0[f], s1[f]) for f in s0.dtype.names)
worst = 0.0
for k in range(K):
    for pl in range(3):
        if not np.array_equal(p0[k][pl], p1[k][pl]):
            same = False
            worst = max(worst, float(np.abs(p0[k][pl] - p1[k][pl]).max()))
if not same and steps >= 0:
    from mfas_amd.engine import flat_layout
    for k in range(K):
        layout, _ = flat_layout(confs[k], hp)
        for pl in range(3):
            for key, shape, off in layout:
                n = int(np.prod(shape))
                a, b = p0[k][pl][off:off + n], p1[k][pl][off:off + n]
                if not np.array_equal(a, b):
                    bad = np.flatnonzero(a != b)
                    print(f"  cand {k} plane {pl} {key}: {bad.size}/{n} differ, max abs {np.abs(a - b).max():g}, first idx {bad[:6].tolist()}")
                    if "dump" in sys.argv:
                        for j in bad[:3]:
                            print("     idx", int(j), "schedule0 W m v:", [float.hex(float(p0[k][q][off + j])) for q in range(3)],
                                  "schedule1 W m v:", [float.hex(float(p1[k][q][off + j])) for q in range(3)])
print("BIT-IDENTICAL" if same else f"MISMATCH (max abs param diff {worst:g}; train loss {s0['train_loss_sum'][0]} vs {s1['train_loss_sum'][0]})", flush=True)
sys.exit(0 if same else 1)

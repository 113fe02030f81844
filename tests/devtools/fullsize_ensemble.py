"""Dev tool: spread of the ENGINE's epoch-0 full-size statistics (conf 4, R=128, BN, B=16, N=10,000/5,600, bf16 taps,
deterministic mode) under 1e-7 relative perturbations of the initial weights, to compare with the reference ensemble of
golden G13 and with the numpy oracle's own spread (same perturbations, same generator)."""
import sys; sys.path.insert(0, "/root/repo")
import numpy as np, torch
import mfas_amd as M
from oracle import np_oracle as O
from tests.helpers import CONFS, engine_hyper, etas_for
ttr, tdv = O.synth_table(10000, 1, snr=0.15, quant="bf16"), O.synth_table(5600, 2, snr=0.15, quant="bf16")
dev = torch.device("cuda:0")
ohp = O.Hyper(R=128, B=16, bn=True, drpt=0.0, epochs=1)
conf = np.array(CONFS["c4"])
rng = np.random.default_rng(0)
ta, tb = M.FeatureTable.from_numpy(ttr, dev, torch.bfloat16), M.FeatureTable.from_numpy(tdv, dev, torch.bfloat16)
for trial in range(10):
    P = O.init_params(conf, ohp, 77)
    if trial:
        for k in P:
            if P[k].dtype == np.float32 and P[k].ndim == 2:
                P[k] = (P[k] * (1 + 1e-7 * rng.standard_normal(P[k].shape))).astype(np.float32)
    pop = M.Population(engine_hyper(ohp), [conf], dev)
    pop.set_state_dict(0, P)
    stats, _ = pop.train(ta, tb, 1, etas_for(ohp, 10000))
    s = stats[0]
    print(trial, "train loss %.4f acc %.4f dev loss %.4f acc %.4f" % (s["train_loss_sum"][0] / 10000, s["train_corrects"][0] / 10000, s["dev_loss_sum"][0] / 5600, s["dev_corrects"][0] / 5600), flush=True)
    pop.close()

import sys; sys.path.insert(0, "/root/repo")
import numpy as np, torch
import mfas_amd as M
from oracle import np_oracle as O
from tests.helpers import CONFS, engine_hyper, etas_for
ttr, tdv = O.synth_table(10000, 1, snr=0.15, quant="bf16"), O.synth_table(5600, 2, snr=0.15, quant="bf16")
dev = torch.device("cuda:0")
ohp = O.Hyper(R=128, B=16, bn=True, drpt=0.0, epochs=3)
conf = np.array(CONFS["c4"])
for cc in (0, 64, 256):
    pop = M.Population(engine_hyper(ohp), [conf], dev, chunk_cols=cc)
    pop.set_state_dict(0, O.init_params(conf, ohp, 77))
    stats, _ = pop.train(M.FeatureTable.from_numpy(ttr, dev, torch.bfloat16), M.FeatureTable.from_numpy(tdv, dev, torch.bfloat16), 3, etas_for(ohp, 10000))
    s = stats[0]
    print("cc", cc, "train loss", (s["train_loss_sum"] / 10000).round(4), "train acc", (s["train_corrects"] / 10000).round(4), "dev loss", (s["dev_loss_sum"] / 5600).round(4), "dev acc", (s["dev_corrects"] / 5600).round(4))
    pop.close()

import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from oracle import np_oracle as O
from tests.helpers import engine_hyper, etas_for, golden
from tests.test_gpu_mmimdb import mm_table
from mfas_amd import Population
dev = torch.device("cuda:0")
g = golden("g11_mmimdb.npz"); conf = g["a/conf"]; w = O.mm_pos_weight(23)
ohp = O.Hyper(R=16, C=23, B=16, bn=True, drpt=0.0, epochs=3, s_sizes=O.MM_S_SIZES, v_sizes=O.MM_V_SIZES, loss_mode=1, pos_weight=w)
hp = engine_hyper(ohp); hp.loss_mode, hp.f1_threshold = 1, 0.3
ttr, tdv = O.synth_table_mm(128, 41), O.synth_table_mm(96, 42)
pop = Population(hp, [conf], dev); print(pop.schedule())
pop.set_pos_weight(w); pop.set_state_dict(0, O.init_params(conf, ohp, 17))
stats, status = pop.train(mm_table(ttr, dev), mm_table(tdv, dev), 3, etas_for(ohp, 128))
print(stats, status, pop.schedule())

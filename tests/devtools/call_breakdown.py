"""Where does one train_sampled_models-equivalent call spend its time? (create / init / train / close)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mfas_amd as M
from oracle import np_oracle as O

dev = torch.device("cuda:0")
pop_n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
tr = M.FeatureTable.synthetic(10000, 1, dev, torch.bfloat16)
dv = M.FeatureTable.synthetic(5600, 2, dev, torch.bfloat16)
hp = M.Hyper(R=128, B=16, bn=True, drpt=0.5)
conf = np.array([[3, 1, 1], [1, 3, 0], [1, 1, 1], [3, 3, 0]])
etas = O.eta_sequence(1e-3, 1e-6, 1, 2, 625.0, 6250)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pop = M.Population(hp, [conf] * pop_n, dev)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    pop.init(list(range(pop_n)))
    torch.cuda.synchronize(); t2 = time.perf_counter()
    order = M.ntu_searchable.make_order(10000, 10, True, 5, dev)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    stats, _ = pop.train(tr, dv, 10, etas, order=order)
    torch.cuda.synchronize(); t4 = time.perf_counter()
    pop.close()
    torch.cuda.synchronize(); t5 = time.perf_counter()
    print(f"pop {pop_n}: create {1e3*(t1-t0):.1f} ms  init {1e3*(t2-t1):.1f}  order {1e3*(t3-t2):.1f}  train {1e3*(t4-t3):.1f}  close {1e3*(t5-t4):.1f}  total {1e3*(t5-t0):.1f}", flush=True)

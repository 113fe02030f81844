"""Where does one train_sampled_models-equivalent call spend its time? (create / init / train / close), per population size.
usage: call_breakdown.py R B bn K1,K2,..."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mfas_amd as M
from oracle import np_oracle as O

R, B, bn = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
Ks = [int(x) for x in sys.argv[4].split(",")]
dev = torch.device("cuda:0")
tr = M.FeatureTable.synthetic(10000, 1, dev, torch.bfloat16)
dv = M.FeatureTable.synthetic(5600, 2, dev, torch.bfloat16)
hp = M.Hyper(R=R, B=B, bn=bool(bn), drpt=0.5, tap_bits=16)
conf = np.array([[3, 1, 1], [1, 3, 0], [1, 1, 1], [3, 3, 0]])
nb = -(-10000 // B)
etas = O.eta_sequence(1e-3, 1e-6, 1, 2, 10000 / B, 10 * nb)
for pop_n in Ks:
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
        tot = t5 - t0
        print(f"R={R} B={B} pop {pop_n:4d} (run {it}): create {1e3*(t1-t0):6.1f} ms  init {1e3*(t2-t1):5.1f}  order {1e3*(t3-t2):5.1f}  train {1e3*(t4-t3):8.1f}  "
              f"close {1e3*(t5-t4):5.1f}  total {1e3*tot:8.1f}  -> create+close = {100*((t1-t0)+(t5-t4))/tot:.2f} % of the call", flush=True)

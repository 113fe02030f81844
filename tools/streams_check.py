"""Concurrent sub-populations on separate HIP streams (one host thread each) vs one population: cand/s for K candidates.
usage: streams_check.py R B bn E K S1,S2,...   (S = number of concurrent sub-populations)"""
import os, sys, time, threading
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mfas_amd as M
from oracle import np_oracle as O

R, B, bn, E, K = (int(x) for x in sys.argv[1:6])
Ss = [int(x) for x in sys.argv[6].split(",")]
N, Nd = 10000, 5600
dev = torch.device("cuda:0")
tr = M.FeatureTable.synthetic(N, 1, dev, torch.bfloat16, snr=0.12)
dv = M.FeatureTable.synthetic(Nd, 2, dev, torch.bfloat16, snr=0.12)
hp = M.Hyper(R=R, B=B, bn=bool(bn), drpt=0.5, tap_bits=16)
conf4 = np.array([[3, 1, 1], [1, 3, 0], [1, 1, 1], [3, 3, 0]])
nb = -(-N // B)
etas = O.eta_sequence(1e-3, 1e-6, 1, 2, N / B, E * nb)
order = M.ntu_searchable.make_order(N, E, True, 5, dev)
torch.cuda.synchronize()
for S in Ss:
    for rep in range(2):
        parts = [list(range(K))[i::S] for i in range(S)]
        res = [None] * S
        streams = [torch.cuda.Stream(device=dev) for _ in range(S)]

        def work(j):
            with torch.cuda.stream(streams[j]):
                pop = M.Population(hp, [conf4] * len(parts[j]), dev, drop_seeds=[100 + i for i in parts[j]])
                pop.init([1 + i for i in parts[j]])
                stats, status = pop.train(tr, dv, E, etas, order=order)
                pop.close()
                res[j] = np.mean([M.best_dev_accuracy(s, Nd) for s in stats])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(j,)) for j in range(S)]
        for t in th: t.start()
        for t in th: t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"R={R} B={B} bn={bn} K={K} E={E} S={S}: {K / dt:.2f} cand/s  {dt / (E * nb) * 1e6:.1f} us/step  acc {np.mean(res):.4f}", flush=True)

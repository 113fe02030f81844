"""Debug: run ONE schedule of the R=128 population of tests/test_gpu_parity.py::test_full_size_properties at reduced size.
usage: sched_probe.py <groups> [ENV=VAL ...]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import np_oracle as O
from mfas_amd import FeatureTable, Hyper, Population
dev = torch.device("cuda:0")
groups = sys.argv[1]
env = dict(a.split("=") for a in sys.argv[2:])
hp = Hyper(R=128, C=60, B=16, bn=True, drpt=0.5)
conf = np.array([[3, 1, 1], [1, 3, 0], [1, 1, 1], [3, 3, 0]])
N = int(os.environ.get("PROBE_N", "1600"))
tr = FeatureTable.synthetic(N, 1, dev, torch.bfloat16, snr=0.15)
dv = FeatureTable.synthetic(800, 2, dev, torch.bfloat16, snr=0.15)
nb = N // 16
etas = O.eta_sequence(1e-3, 1e-6, 1, 2, N / 16, 2 * nb)
os.environ["MFAS_GROUPS"] = groups
os.environ.update(env)
pop = Population(hp, [conf] * int(os.environ.get("PROBE_K", "8")), dev, drop_seeds=list(range(int(os.environ.get("PROBE_K", "8")))), chunk_cols=128)
for k in list(env) + ["MFAS_GROUPS"]:
    del os.environ[k]
print("schedule", pop.schedule(), flush=True)
pop.init([100 + s for s in range(int(os.environ.get("PROBE_K", "8")))])
t0 = time.time()
stats, status = pop.train(tr, dv, 2, etas)
print("ok", time.time() - t0, stats["dev_corrects"][:, 1], status, flush=True)

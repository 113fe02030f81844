"""Where one train_sampled_models call spends its wall time (host + device), search-sized: K sampled L=4 configurations, R=16, B=20, E=10.
usage: call_breakdown.py [K]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mfas_amd as M
from mfas_amd import ntu_searchable as NS, engine as EN
from types import SimpleNamespace
K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
tr = M.FeatureTable.synthetic(10000, 1, dev, torch.bfloat16, snr=0.12)
dv = M.FeatureTable.synthetic(5600, 2, dev, torch.bfloat16, snr=0.12)
import main_searchable_ntu as MS
args = MS.parse_args(["--epochs", "10", "--no-verbose"])
np.random.seed(0)
confs = [np.stack([np.random.randint(0, 4, 4), np.random.randint(0, 4, 4), np.random.randint(0, 2, 4)], 1) for _ in range(K)]
loaders = {"train": M.FeatureLoader(tr, 20, shuffle=True), "dev": M.FeatureLoader(dv, 20, shuffle=False)}
T = {}
def wrap(mod, name):
    f = getattr(mod, name)
    def g(*a, **k):
        torch.cuda.synchronize(); t = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize(); T[name] = T.get(name, 0.0) + time.perf_counter() - t
        return r
    setattr(mod, name, g)
for n in ("_init_population_device_streams", "make_order_per_candidate", "_init_population_from_torch"):
    wrap(NS, n)
for n in ("train", "close", "__init__"):
    f = getattr(EN.Population, n)
    def mk(f, n):
        def g(self, *a, **k):
            torch.cuda.synchronize(); t = time.perf_counter()
            r = f(self, *a, **k)
            torch.cuda.synchronize(); T["Population." + n] = T.get("Population." + n, 0.0) + time.perf_counter() - t
            return r
        return g
    setattr(EN.Population, n, mk(f, n))
for rep in range(3):
    T.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    acc = NS.train_sampled_models(confs, NS.Searchable_Skeleton_Image_Net, loaders, args, dev)
    torch.cuda.synchronize(); tot = time.perf_counter() - t0
    print(f"call {rep}: K={K} total {tot*1e3:.1f} ms ({K/tot:.1f} cand/s)  " + "  ".join(f"{k} {v*1e3:.1f}" for k, v in sorted(T.items(), key=lambda kv: -kv[1])) + f"  other {(tot - sum(v for k, v in T.items() if not k.startswith('_init') or True))*1e3:.1f}")

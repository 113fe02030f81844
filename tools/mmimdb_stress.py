"""BASELINE config 5 stress: MM-IMDB-shaped fusion search candidates (text taps 64/128-d, image taps 4 x 512-d, 23
multi-label genres, fp16 features, weighted-BCE head, F1-samples metric) — candidates/s of the engine on one GPU.
Synthetic multi-hot data generated on the device (N_train 15,552 / N_dev 2,608 = the MM-IMDB split sizes).

    python tools/mmimdb_stress.py [pop ...]
"""
import sys, time
from types import SimpleNamespace

import numpy as np
import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mfas_amd as M
from mfas_amd import mmimdb_searchable as MM


def table(N, seed, device, C=23):
    g = torch.Generator(device=device)
    g.manual_seed(321)
    mus = {f"s{j}": torch.randn(C, w, generator=g, device=device) for j, w in enumerate(MM.MM_TEXT_SIZES)}
    mus.update({f"v{j}": torch.randn(C, w, generator=g, device=device) for j, w in enumerate(MM.MM_IMAGE_SIZES)})
    g.manual_seed(seed)
    z = (torch.rand(N, C, generator=g, device=device) < 0.15).float()
    taps = {k: torch.relu(0.6 * z @ mu + torch.randn(N, mu.shape[1], generator=g, device=device)).to(torch.float16) for k, mu in mus.items()}
    return M.FeatureTable(taps, torch.zeros(N, dtype=torch.int32, device=device), multilabel=z)


def main():
    pops = [int(x) for x in sys.argv[1:]] or [8, 64, 512]
    dev = torch.device("cuda:0")
    train, devt = table(15552, 1, dev), table(2608, 2, dev)
    for R in (16, 128):
        args = SimpleNamespace(num_outputs=23, drpt=0.5, inner_representation_size=R, batchnorm=False, alphas=False, multitask=False,
                               weightsharing=False, batchsize=20, eta_max=1e-3, eta_min=1e-6, Ti=1, Tm=2, use_dataparallel=False,
                               verbose=False, epochs=10, engine_init="device", vid_len=(8, 32))
        loaders = {"train": M.FeatureLoader(train, 20, shuffle=True), "dev": M.FeatureLoader(devt, 20, shuffle=False)}
        rng = np.random.default_rng(0)
        for K in pops:
            confs = [np.stack([rng.integers(0, 2, L), rng.integers(0, 4, L), rng.integers(0, 2, L)], 1) for L in rng.integers(1, 5, K)]
            MM.train_sampled_models(confs, MM.Searchable_Text_Image_Net, loaders, args, dev)        # warm-up
            torch.cuda.synchronize()
            t = time.perf_counter()
            f1 = MM.train_sampled_models(confs, MM.Searchable_Text_Image_Net, loaders, args, dev)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t
            print(f"MM-IMDB-shaped R={R} B=20 E=10 fp16 taps: {K:4d} candidates (L=1..4) in {dt*1e3:8.1f} ms = {K/dt:7.1f} cand/s, mean best F1 {np.mean(f1):.3f}", flush=True)


if __name__ == "__main__":
    main()

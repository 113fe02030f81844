"""PyTorch-CPU eager restatement of the reference's inner loop — TEST / BASELINE INFRASTRUCTURE ONLY.

BASELINE.md §2 asks for the CPU baseline to be "the build's own PyTorch-CPU eager restatement of the reference step sequence"
timed on the GPU box's host cores (the reference's Python cannot travel there).  This file restates, op for op, what the
reference executes per candidate (citations are /root/reference paths):

* the network    models/search/ntu_searchable.py:206-247 (concat -> Linear -> nl -> [BatchNorm1d] -> [Dropout] per cell,
                  cells built at :258-286; linear head :200,242)
* the train loop models/search/train_searchable/ntu.py:14-89 (per batch: zero_grad, forward, CrossEntropyLoss,
                  scheduler.step + update_optimizer, backward, Adam step, loss.item(); per epoch a dev pass; best dev acc)
* the optimizer  torch.optim.Adam(central params, lr=eta_max, weight_decay=1e-4)  ntu_searchable.py:65
* the scheduler  models/auxiliary/scheduler.py:12-46 (per-batch cosine annealing with warm restarts: the eta sequence of
                  oracle.np_oracle.eta_sequence, pushed through optimizer.state_dict() / load_state_dict() every step as
                  the reference does, :40-46)

Only bench.py's `cpu_baseline` leg and tests may import it; the product path never does (tests/test_host_cpu.py enforces it).
It is pinned against the numpy oracle (itself pinned to the reference goldens) in tests/test_oracle_golden.py.
"""
import time

import numpy as np
import torch
import torch.nn as nn

S_SIZES = (128, 256, 1024, 512)     # ntu_searchable.py:291
V_SIZES = (512, 1024, 2048, 2048)   # ntu_searchable.py:292


class FusionNet(nn.Module):
    def __init__(self, conf, R, C, bn, drpt):
        super().__init__()
        self.conf = [tuple(int(x) for x in row) for row in conf]
        cells = []
        for i, (s, v, nl) in enumerate(self.conf):
            k_in = S_SIZES[s] + V_SIZES[v] + (R if i > 0 else 0)
            act = [nn.ReLU(), nn.Sigmoid(), nn.LeakyReLU()][nl]
            mods = [nn.Linear(k_in, R), act]
            if bn:
                mods.append(nn.BatchNorm1d(R))
            if drpt > 1e-10:
                mods.append(nn.Dropout(drpt))
            cells.append(nn.Sequential(*mods))
        self.fusion_layers = nn.ModuleList(cells)
        self.central_classifier = nn.Linear(R, C)

    def forward(self, taps):
        out = None
        for i, (s, v, _) in enumerate(self.conf):
            parts = [taps[f"s{s}"], taps[f"v{v}"]] + ([out] if i > 0 else [])
            out = self.fusion_layers[i](torch.cat(parts, 1))
        return self.central_classifier(out)


def push_lr(opt, lr):
    """scheduler.py:40-46: the reference round-trips the optimizer's whole state_dict on every step to set the LR."""
    sd = opt.state_dict()
    for g in sd["param_groups"]:
        g["lr"] = lr
    opt.load_state_dict(sd)


def time_candidate(train, dev, conf, R, B, bn, drpt, epochs_timed=1, threads=None, eta=(1e-3, 1e-6, 1, 2), C=60):
    """Runs `epochs_timed` full epochs (train over all of `train` in shuffled order + eval over all of `dev`) of one
    candidate on CPU and returns (seconds per epoch, dev accuracy of the last epoch, threads used).
    train / dev: dict of float32 CPU tensors 's0'..'v3' (N, width) + 'label' (N,) int64."""
    if threads:
        torch.set_num_threads(int(threads))
    from .np_oracle import eta_sequence      # scheduler.py:12-46 restated there and pinned to golden G1
    net = FusionNet(conf, R, C, bn, drpt)
    opt = torch.optim.Adam(net.parameters(), lr=eta[0], weight_decay=1e-4)
    N, Nd = len(train["label"]), len(dev["label"])
    etas = eta_sequence(eta[0], eta[1], eta[2], eta[3], N / B, epochs_timed * (-(-N // B)))
    step = 0
    crit = nn.CrossEntropyLoss()
    keys = [k for k in train if k != "label"]
    t0 = time.perf_counter()
    acc = 0.0
    for _ in range(epochs_timed):
        net.train(True)
        perm = torch.randperm(N)
        run = 0.0
        for i in range(0, N, B):
            idx = perm[i:i + B]
            opt.zero_grad()
            out = net({k: train[k][idx] for k in keys})
            loss = crit(out, train["label"][idx])
            push_lr(opt, float(etas[step]))
            step += 1
            loss.backward()
            opt.step()
            run += loss.item() * len(idx)
        net.train(False)
        corr = 0
        with torch.no_grad():
            for i in range(0, Nd, B):
                out = net({k: dev[k][i:i + B] for k in keys})
                corr += int((out.argmax(1) == dev["label"][i:i + B]).sum())
        acc = corr / Nd
    return (time.perf_counter() - t0) / epochs_timed, acc, torch.get_num_threads()

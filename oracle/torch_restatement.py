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


def time_candidate(train, dev, conf, R, B, bn, drpt, threads=None, eta=(1e-3, 1e-6, 1, 2), C=60, budget_s=None):
    """One epoch (train over all of `train` in shuffled order + eval over all of `dev`) of one candidate on CPU, or as much of
    it as fits `budget_s` seconds (3/4 for the train phase, 1/4 for the dev phase).  Returns a dict: seconds per train step,
    seconds per dev row, steps / rows actually run, whether the epoch was complete, dev accuracy of what ran, threads used.
    train / dev: dict of float32 CPU tensors 's0'..'v3' (N, width) + 'label' (N,) int64."""
    if threads:
        torch.set_num_threads(int(threads))
    from .np_oracle import eta_sequence      # scheduler.py:12-46 restated there and pinned to golden G1
    net = FusionNet(conf, R, C, bn, drpt)
    opt = torch.optim.Adam(net.parameters(), lr=eta[0], weight_decay=1e-4)
    N, Nd = len(train["label"]), len(dev["label"])
    nb = -(-N // B)
    etas = eta_sequence(eta[0], eta[1], eta[2], eta[3], N / B, nb)
    crit = nn.CrossEntropyLoss()
    keys = [k for k in train if k != "label"]
    net.train(True)
    perm = torch.randperm(N)
    steps = 0
    t0 = time.perf_counter()
    for i in range(0, N, B):
        idx = perm[i:i + B]
        opt.zero_grad()
        out = net({k: train[k][idx] for k in keys})
        loss = crit(out, train["label"][idx])
        push_lr(opt, float(etas[steps]))
        loss.backward()
        opt.step()
        loss.item()
        steps += 1
        if budget_s is not None and time.perf_counter() - t0 > 0.75 * budget_s:
            break
    t_train = time.perf_counter() - t0
    net.train(False)
    corr = rows = 0
    t1 = time.perf_counter()
    with torch.no_grad():
        for i in range(0, Nd, B):
            out = net({k: dev[k][i:i + B] for k in keys})
            corr += int((out.argmax(1) == dev["label"][i:i + B]).sum())
            rows += len(dev["label"][i:i + B])
            if budget_s is not None and time.perf_counter() - t1 > 0.25 * budget_s:
                break
    t_dev = time.perf_counter() - t1
    return {"s_per_step": t_train / steps, "s_per_dev_row": t_dev / rows, "steps": steps, "dev_rows": rows,
            "full_epoch": steps == nb and rows == Nd, "dev_acc": corr / max(rows, 1), "threads": torch.get_num_threads(),
            "s_per_epoch": (t_train / steps) * nb + (t_dev / rows) * Nd}


if __name__ == "__main__":
    # `python -m oracle.torch_restatement --n-train N --n-dev M --R r --B b --bn 0|1 --drpt p --threads t --budget s`: times one
    # epoch on planted-signal tables of the given shape generated here (timing depends on shapes only) and prints one JSON
    # line.  bench.py runs it as a SUBPROCESS with a hard timeout, so a pathological thread count can never stall the bench.
    import argparse
    import json
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-train", type=int, default=10000)
    ap.add_argument("--n-dev", type=int, default=5600)
    ap.add_argument("--R", type=int, default=128)
    ap.add_argument("--B", type=int, default=16)
    ap.add_argument("--bn", type=int, default=1)
    ap.add_argument("--drpt", type=float, default=0.5)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--budget", type=float, default=20.0)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    g = torch.Generator().manual_seed(0)

    def tab(n):
        lab = torch.randint(0, 60, (n,), generator=g)
        t = {f"s{j}": torch.relu(torch.randn(n, w, generator=g)) for j, w in enumerate(S_SIZES)}
        t.update({f"v{j}": torch.relu(torch.randn(n, w, generator=g)) for j, w in enumerate(V_SIZES)})
        t["label"] = lab
        return t

    conf4 = [[3, 1, 1], [1, 3, 0], [1, 1, 1], [3, 3, 0]]
    r = time_candidate(tab(a.n_train), tab(a.n_dev), conf4, a.R, a.B, bool(a.bn), a.drpt, threads=a.threads, budget_s=a.budget)
    print("TORCH_RESTATEMENT " + json.dumps(r), flush=True)

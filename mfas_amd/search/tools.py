"""Controller-side helpers of the sequential model-based search (EPNAS-style).

Semantics follow /root/reference/models/search/tools.py (function : lines):
predict_accuracies_with_surrogate :22-30, update_surrogate_dataloader :33-35, train_surrogate :38-44,
sample_k_configurations :47-58, merge_unfolded_with_sampled :66-97, sample_k_configurations_directly :100-118,
compute_temperature :121-124.  Pure CPU / numpy: microseconds of work per search step, runs on every rank.
"""
import random

import numpy as np

from . import surrogate as surr


def predict_accuracies_with_surrogate(configurations, surrogate, device):
    """tools.py:22-30 evaluates the surrogate one configuration at a time (1,600 single-sequence LSTM forwards per search
    step: two thirds of the controller's wall time).  Here all configurations of one length go through ONE batched forward
    under no_grad; the values agree with the per-configuration path to float32 round-off (a batched GEMM instead of
    GEMVs), which the pinned search decisions (golden G9) do not see."""
    import torch
    out = [None] * len(configurations)
    by_len = {}
    for i, c in enumerate(configurations):
        by_len.setdefault(len(c), []).append(i)
    with torch.no_grad():
        for _, idx in by_len.items():
            seq = np.stack([np.asarray(configurations[i], np.float32) for i in idx], 1)      # (seq_len, n, 3)
            pred = surrogate(torch.from_numpy(seq).to(device)).cpu().numpy()[:, 0]
            for j, i in enumerate(idx):
                out[i] = pred[j]
    return out


def update_surrogate_dataloader(surrogate_dataloader, configurations, accuracies):
    for conf, acc in zip(configurations, accuracies):
        surrogate_dataloader.add_datum(conf, acc)


def train_surrogate(surrogate, surrogate_dataloader, surrogate_optimizer, surrogate_criterion, args, device):
    data = surrogate_dataloader.get_data(to_torch=True)
    return surr.train_simple_surrogate(surrogate, surrogate_criterion, surrogate_optimizer, data,
                                       args.epochs_surrogate, device)


def sample_k_configurations(configurations, accuracies_, k, temperature):
    """p ∝ acc, tempered p^(1/T), k draws without replacement from the global numpy stream (tools.py:47-58).
    Departure: k is clamped to the number of configurations (the reference raises ValueError when num_samples exceeds
    the 32 single-layer configurations of the first level, e.g. BASELINE config 4's --num_samples 50)."""
    k = min(int(k), len(configurations))
    acc = np.array(accuracies_)
    p = acc / acc.sum()
    p = pow(p, 1.0 / temperature)
    p = p / p.sum()
    idx = np.random.choice(len(configurations), k, replace=False, p=p)
    return [configurations[i] for i in idx]


def sample_k_configurations_uniform(configurations, k):
    idx = np.random.choice(len(configurations), k)
    return [configurations[i] for i in idx]


def merge_unfolded_with_sampled(previous_top_k_configurations, unfolded_configurations, layer):
    """Cross product of the previous top-K (each (L,3)) with the 32 single-layer options for position `layer`:
    replace row `layer` when it exists, append a row otherwise (tools.py:66-97)."""
    merged = []
    if not previous_top_k_configurations:
        if layer != 0:
            raise ValueError("merge_unfolded_with_sampled: no previous configurations but layer != 0")
        return [np.expand_dims(u, 0) for u in unfolded_configurations]
    for prev in previous_top_k_configurations:
        for u in unfolded_configurations:
            if layer < len(prev):
                new = np.copy(prev)
                new[layer] = u
            else:
                new = np.concatenate([prev, np.expand_dims(u, 0)], 0)
            merged.append(new)
    return merged


def sample_k_configurations_directly(k, max_progression_levels, get_possible_layer_configurations_fun):
    """Random-search sampler (tools.py:100-118): random depth in [1, max], rows drawn uniformly.  Like the reference
    every row is drawn from the LAST level's option list (its loop variable `l` is reused after the loop)."""
    per_layer = [get_possible_layer_configurations_fun(l) for l in range(max_progression_levels)]
    last = per_layer[-1]
    out = []
    for _ in range(k):
        depth = random.randint(1, max_progression_levels)
        rows = [sample_k_configurations_uniform(last, 1) for _ in range(depth)]
        out.append(np.array(rows)[:, 0, :])
    return out


def compute_temperature(iteration, args):
    return (args.initial_temperature - args.final_temperature) * np.exp(
        -(iteration + 1.0) ** 2 / args.temperature_decay ** 2) + args.final_temperature

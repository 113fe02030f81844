"""Population sharding across the GPUs of one node.

The reference has no multi-process code at all (only optional single-process nn.DataParallel,
/root/reference/models/search/ntu_searchable.py:70-71).  Candidates of one train_sampled_models
call are independent (ntu_searchable.py:38-94), so the population is the natural shard: rank r
trains its share and the per-candidate accuracies are gathered with ONE small collective
(RCCL all_gather over xGMI on the GPU box, gloo in the CPU tests).
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch
import torch.distributed as dist


def dist_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def candidate_cost(conf, R: int, s_sizes, v_sizes, C: int = 60) -> int:
    """Parameter count P_i: per-step work of a candidate is ~24*P_i bytes."""
    conf = np.asarray(conf).reshape(-1, 3)
    p = 0
    for i, (s, v, _) in enumerate(conf):
        p += R * (s_sizes[int(s)] + v_sizes[int(v)] + (R if i else 0)) + R
    return int(p + R * C + C)


def assign(costs: Sequence[int], world: int) -> List[int]:
    """Greedy longest-processing-time assignment candidate -> rank; deterministic on every rank."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0] * world
    owner = [0] * len(costs)
    for i in order:
        r = min(range(world), key=lambda j: (load[j], j))
        owner[i] = r
        load[r] += costs[i]
    return owner


def gather_accuracies(local_idx: Sequence[int], local_acc: Sequence[float], K: int, device=None) -> List[float]:
    """All ranks end up with the K accuracies in input order.  One all_gather of ceil(K/W) doubles
    (+ their indices) per call — latency-bound, a few hundred bytes."""
    rank, world = dist_info()
    if world == 1:
        out = [0.0] * K
        for i, a in zip(local_idx, local_acc):
            out[i] = float(a)
        return out
    backend = dist.get_backend()
    dev = torch.device("cpu") if backend == "gloo" else (device or torch.device("cuda", torch.cuda.current_device()))
    cap = -(-K // world) + 1
    cap = max(cap, max(len(local_idx), 1))
    capt = torch.tensor([cap], dtype=torch.int64, device=dev)
    dist.all_reduce(capt, op=dist.ReduceOp.MAX)
    cap = int(capt.item())
    buf = torch.full((cap, 2), -1.0, dtype=torch.float64, device=dev)
    for j, (i, a) in enumerate(zip(local_idx, local_acc)):
        buf[j, 0] = float(i)
        buf[j, 1] = float(a)
    bufs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf)
    out = [float("nan")] * K
    for b in bufs:
        b = b.cpu().numpy()
        for i, a in b:
            if i >= 0:
                out[int(i)] = float(a)
    assert not any(np.isnan(out)), "a candidate was trained by no rank"
    return out


def broadcast_seed(seed: int, device=None) -> int:
    """Rank 0's seed to everyone (so that init / shuffle / dropout streams do not depend on the
    world size)."""
    rank, world = dist_info()
    if world == 1:
        return int(seed)
    backend = dist.get_backend()
    dev = torch.device("cpu") if backend == "gloo" else (device or torch.device("cuda", torch.cuda.current_device()))
    t = torch.tensor([int(seed)], dtype=torch.int64, device=dev)
    dist.broadcast(t, src=0)
    return int(t.item())

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


# One train step of a candidate costs ~24*P_i bytes of streaming PLUS a serial chain of L_i cells that no
# amount of bandwidth shortens; in the small-population regime the search runs in (6-8 candidates per GPU) the second
# term dominates.  Measured on MI355X: ~4 us per cell of chain latency ~ 20 KB of streaming at 5 TB/s per
# cell, i.e. ~850 parameters' worth of bytes per cell at R=16 and ~6,800 at R=128 -> LATENCY_PARAMS_PER_CELL * R/16.
LATENCY_PARAMS_PER_CELL = 850


def candidate_cost(conf, R: int, s_sizes, v_sizes, C: int = 60) -> int:
    """Cost of one train step in parameter units: P_i (bandwidth term) + a latency term proportional to the depth L_i."""
    conf = np.asarray(conf).reshape(-1, 3)
    p = 0
    for i, (s, v, _) in enumerate(conf):
        p += R * (s_sizes[int(s)] + v_sizes[int(v)] + (R if i else 0)) + R
    lat = len(conf) * LATENCY_PARAMS_PER_CELL * max(1, R // 16)
    return int(p + R * C + C + lat)


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


def gather_accuracies(local_idx: Sequence[int], local_acc: Sequence[float], K: int, device=None,
                      cap: int | None = None) -> List[float]:
    """All ranks end up with the K accuracies in input order: ONE all_gather of `cap` (index, accuracy) pairs per rank
    and one device-to-host copy — latency-bound, a few hundred bytes.  `cap` (the largest per-rank share) is known to
    every rank from assign(); without it ceil(K/W)+1 is only an upper bound for round-robin-like assignments, so
    callers that shard with assign() pass it."""
    rank, world = dist_info()
    if world == 1:
        out = [0.0] * K
        for i, a in zip(local_idx, local_acc):
            out[i] = float(a)
        return out
    backend = dist.get_backend()
    dev = torch.device("cpu") if backend == "gloo" else (device or torch.device("cuda", torch.cuda.current_device()))
    if cap is None:
        cap = K
    assert len(local_idx) <= cap, (len(local_idx), cap)
    host = np.full((cap, 2), -1.0, np.float64)
    for j, (i, a) in enumerate(zip(local_idx, local_acc)):
        host[j, 0] = float(i)
        host[j, 1] = float(a)
    buf = torch.from_numpy(host).to(dev)
    allb = torch.empty((world * cap, 2), dtype=torch.float64, device=dev)   # rank-major concatenation
    dist.all_gather_into_tensor(allb, buf)
    out = [float("nan")] * K
    for i, a in allb.cpu().numpy():
        if i >= 0:
            out[int(i)] = float(a)
    assert not any(np.isnan(out)), "a candidate was trained by no rank"
    return out


# Step-time model of ONE rank's share (microseconds per lock-step train step), from the measured sweeps in DESIGN.md §5a
# (profiles/r02_popsweep*.log, r03_popsweep_split_kernels.log).  Strong scaling of a small population is LATENCY-bound: with the
# resident persistent schedule (R <= 16) a step costs the same for 1...8 candidates, so giving a rank fewer candidates than that
# buys nothing — the model is what lets the sharder see it.
RESIDENT_STEP_US = ((8, 14.3), (16, 15.3), (28, 17.5))      # R <= 16: (largest share, us per step) of the resident schedule
STREAM_BYTES_PER_US = 5.5e6                                   # what the sweep streams at (24 B per parameter and step)


def predicted_step_us(share_costs: Sequence[int], R: int) -> float:
    """Predicted duration of one lock-step train step of a rank that holds the candidates with these costs."""
    n = len(share_costs)
    if n == 0:
        return 0.0
    bytes_us = 24.0 * float(sum(share_costs)) / STREAM_BYTES_PER_US
    if R <= 16:
        for cap, us in RESIDENT_STEP_US:
            if n <= cap:
                return us
        if n <= 2 * RESIDENT_STEP_US[-1][0]:      # two resident rounds, one after the other (ntu_searchable._plan_rounds)
            return 2.0 * RESIDENT_STEP_US[-1][1]
        return max(38.0, 12.0 + bytes_us)         # launch-per-phase, lean chain
    return max(52.0 * min(1.0, R / 128.0) + 15.0, 35.0 + bytes_us)     # general chain: its latency, or the stream


def choose_ranks(costs: Sequence[int], world: int, R: int, tolerance: float = 0.03) -> int:
    """Number of ranks a call should really use: the SMALLEST w <= world whose predicted call time (the slowest rank's step
    time under assign()) is within `tolerance` of the best over 1..world.  Ranks beyond w train nothing in this call."""
    if world <= 1 or not costs:
        return max(1, min(world, 1))
    times = []
    for w in range(1, world + 1):
        owner = assign(costs, w)
        times.append(max(predicted_step_us([c for c, o in zip(costs, owner) if o == r], R) for r in range(w)))
    best = min(times)
    for w, t in enumerate(times, 1):
        if t <= best * (1.0 + tolerance):
            return w
    return world


def shard(costs: Sequence[int], world: int, R: int | None = None):
    """owner per candidate and the largest per-rank share (the all_gather's row count), both computed locally and identically
    on every rank.  With R given, the call uses only as many ranks as the step-time model says pay (choose_ranks)."""
    used = choose_ranks(costs, world, R) if R is not None else world
    owner = assign(costs, used)
    counts = [0] * world
    for o in owner:
        counts[o] += 1
    return owner, max(counts + [1])


def conf_digest(confs) -> int:
    """Order-sensitive 62-bit digest of a list of configurations (checked across ranks before sharding)."""
    import hashlib
    h = hashlib.sha256()
    for c in confs:
        a = np.ascontiguousarray(np.asarray(c, np.int64).reshape(-1, 3))
        h.update(np.int64(len(a)).tobytes())
        h.update(a.tobytes())
    return int.from_bytes(h.digest()[:8], "little") >> 2


def broadcast_seed(seed: int, device=None, confs=None) -> int:
    """Rank 0's seed to everyone (so that init / shuffle / dropout streams do not depend on the world size).  The same
    broadcast carries rank 0's digest of the configuration list: every rank runs the (seeded) controller redundantly
    (SURVEY §8e), and a rank whose sampler drifted (an unseeded RNG, a float that rounded differently) must not
    silently train a different population — it raises instead."""
    rank, world = dist_info()
    if world == 1:
        return int(seed)
    backend = dist.get_backend()
    dev = torch.device("cpu") if backend == "gloo" else (device or torch.device("cuda", torch.cuda.current_device()))
    mine = conf_digest(confs) if confs is not None else 0
    t = torch.tensor([int(seed), mine], dtype=torch.int64, device=dev)
    dist.broadcast(t, src=0)
    seed0, dig0 = (int(x) for x in t.cpu().tolist())
    if confs is not None and dig0 != mine:
        raise RuntimeError(f"rank {rank}: sampled_configurations differ from rank 0's (digest {mine:#x} vs {dig0:#x}); "
                           "the controller must be seeded identically on every rank (random / numpy / torch)")
    return seed0

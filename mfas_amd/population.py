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


_GROUP = None      # the process group the sharded calls talk over; None = torch.distributed's default group


def set_group(group):
    """Route the population's collectives (the accuracy gather, the seed / digest and calibration broadcasts) over `group`
    instead of the default process group — e.g. an RCCL group created next to a gloo default group (bench.py does that so that a
    node whose RCCL cannot start still measures: the data path has no collective, the gather is a few hundred bytes)."""
    global _GROUP
    _GROUP = group


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
                      cap: int | None = None, failed: bool = False, strict: bool = True, fatal: bool = False):
    """All ranks end up with the K accuracies in input order: ONE all_gather of `cap` (index, accuracy) pairs per rank (+ one
    status row) and one device-to-host copy — latency-bound, a few hundred bytes.  `cap` (the largest per-rank share) is known to
    every rank from assign(); without it ceil(K/W)+1 is only an upper bound for round-robin-like assignments, so callers that
    shard with assign() pass it.  `failed`: this rank could not train (all of) its share — it still joins the collective, so
    nobody hangs.  `fatal` (with failed): the failure is an argument / programming error (a bad configuration, a missing tap) that a
    re-queue on another rank would only repeat — the status row says so and every rank learns it from the same collective
    (FATAL_RANKS holds them after the call).  strict: raise when a candidate was trained by no rank; else return (accuracies with
    NaN holes, failed ranks)."""
    global FATAL_RANKS
    FATAL_RANKS = []
    rank, world = dist_info()
    if world == 1:
        out = [float("nan")] * K if not strict else [0.0] * K
        for i, a in zip(local_idx, local_acc):
            out[i] = float(a)
        FATAL_RANKS = [0] if (failed and fatal) else []
        return out if strict else (out, [0] if failed else [])
    backend = dist.get_backend(_GROUP)
    dev = torch.device("cpu") if backend == "gloo" else (device or torch.device("cuda", torch.cuda.current_device()))
    if cap is None:
        cap = K
    assert len(local_idx) <= cap, (len(local_idx), cap)
    host = np.full((cap + 1, 2), -1.0, np.float64)
    for j, (i, a) in enumerate(zip(local_idx, local_acc)):
        host[j, 0] = float(i)
        host[j, 1] = float(a)
    host[cap, 0] = (-4.0 if fatal else -2.0) if failed else -3.0          # status row: -3 fine, -2 failed (re-queue), -4 failed for good
    buf = torch.from_numpy(host).to(dev)
    allb = torch.empty((world * (cap + 1), 2), dtype=torch.float64, device=dev)   # rank-major concatenation
    dist.all_gather_into_tensor(allb, buf, group=_GROUP)
    out = [float("nan")] * K
    rows = allb.cpu().numpy().reshape(world, cap + 1, 2)
    bad = [r for r in range(world) if rows[r, cap, 0] in (-2.0, -4.0)]
    FATAL_RANKS = [r for r in range(world) if rows[r, cap, 0] == -4.0]
    for r in range(world):
        for i, a in rows[r, :cap]:
            if i >= 0:
                out[int(i)] = float(a)
    if strict:
        if bad or any(np.isnan(out)):
            raise RuntimeError(f"candidates {[i for i, a in enumerate(out) if np.isnan(a)]} were trained by no rank (failed ranks: {bad})")
        return out
    return out, bad


FATAL_RANKS: list = []      # ranks whose status row of the LAST gather said "failed for good" (set by gather_accuracies)
FATAL_ERRORS = (ValueError, TypeError, NotImplementedError, AssertionError)


def train_sharded(wanted: Sequence[int], owner: Sequence[int], cap: int, K: int, costs: Sequence[int], train_share, device=None):
    """Run `train_share(indices) -> {index: accuracy}` on this rank's share and gather everyone's results (ONE collective when
    nothing goes wrong).  A rank whose share raises — out of memory, a kernel error, a persistent-loop timeout — does not take the
    job down with a hang in the collective: it reports the failure in the gather, and the candidates it could not deliver are
    RE-QUEUED once over the ranks that did not fail (candidates are independent, /root/reference/models/search/ntu_searchable.py:38-94;
    per-candidate seeds make the result independent of who trains it).  If the second attempt fails too, or every rank failed,
    all ranks raise the same RuntimeError.  An argument / programming error (FATAL_ERRORS: a bad configuration or a missing tap is
    raised only by the rank that owns that candidate) ALSO travels through the gather — every rank enters the collective whatever
    happened — and is not re-queued: the owning rank re-raises its exception, the others raise a RuntimeError naming it.  (A rank that DIES — killed, segfault — cannot be survived inside one process group:
    torchrun tears the job down.)  Returns the K accuracies (entries outside `wanted` are 0)."""
    rank, world = dist_info()
    mine = [i for i, o in zip(wanted, owner) if o == rank]
    if world == 1:
        got = train_share(mine)
        out = [0.0] * K
        for i in mine:
            out[i] = float(got[i])
        return out
    err, got = None, {}
    try:
        got = train_share(mine)
    except Exception as e:      # noqa: BLE001 — every failure is reported through the collective: nobody may skip the gather
        err = e
        if not isinstance(e, FATAL_ERRORS):     # runtime / device errors: re-queued below
            import traceback
            import warnings
            warnings.warn(f"mfas_amd: rank {rank} failed to train its share of {len(mine)} candidate(s):\n{traceback.format_exc()}")
    have = [i for i in mine if i in got]
    out, bad = gather_accuracies(have, [got[i] for i in have], K, device, cap=cap, failed=err is not None, strict=False,
                                 fatal=isinstance(err, FATAL_ERRORS))
    if FATAL_RANKS:             # the same verdict on every rank, from the same collective
        if isinstance(err, FATAL_ERRORS):
            raise err
        raise RuntimeError(f"rank(s) {FATAL_RANKS} raised an argument / programming error while training their share; nothing was re-queued")
    missing = [i for i in wanted if np.isnan(out[i])]
    if missing:
        good = [r for r in range(world) if r not in bad]
        if not good:
            raise RuntimeError(f"every rank failed to train its share (this rank: {err!r})")
        cost_of = dict(zip(wanted, costs))
        own2 = assign([cost_of[i] for i in missing], len(good))
        mine2 = [i for i, o in zip(missing, own2) if good[o] == rank]
        cap2 = max([sum(1 for o in own2 if o == j) for j in range(len(good))] + [1])
        err2, got2 = None, {}
        if mine2:
            try:
                got2 = train_share(mine2)
            except Exception as e:      # noqa: BLE001
                err2 = e
        have2 = [i for i in mine2 if i in got2]
        out2, bad2 = gather_accuracies(have2, [got2[i] for i in have2], K, device, cap=cap2, failed=err2 is not None, strict=False)
        for i in missing:
            out[i] = out2[i]
        still = [i for i in wanted if np.isnan(out[i])]
        if still:
            raise RuntimeError(f"candidates {still} could not be trained: rank(s) {bad} failed, and so did the re-queue on rank(s) {bad2} "
                               f"(this rank's error: {(err2 or err)!r})")
        if rank == 0:
            import warnings
            warnings.warn(f"mfas_amd: rank(s) {bad} failed to train {len(missing)} candidate(s); re-queued on rank(s) {good}")
    return [0.0 if np.isnan(a) else a for a in out]


# Step-time model of ONE rank's share (microseconds per lock-step train step), from the measured sweeps in DESIGN.md §5a
# (profiles/r02_popsweep*.log, r03_popsweep_split_kernels.log).  Strong scaling of a small population is LATENCY-bound: with the
# resident persistent schedule (R <= 16) a step costs the same for 1...8 candidates, so giving a rank fewer candidates than that
# buys nothing — the model is what lets the sharder see it.
RESIDENT_STEP_US = ((8, 10.7), (16, 11.3), (28, 15.5))      # R <= 16: (largest share, us per step) of the resident schedule (round 5: profiles/r05_popsweep.log)
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


# ------------------------------------------------------------------------------------------------------------------
# The same model, CALIBRATED on the device it runs on (round 4).  The constants above are what one MI355X box measured; another box
# (or another device, CU count, geometry: B > 32 and C > 64 have no lean chain and therefore no resident schedule) would make
# choose_ranks idle ranks for the wrong reasons.  StepModel measures a handful of short trainings (a few hundred train steps each)
# of a canonical configuration of the call's geometry — the resident schedule at 1 / 8 / 16 / capacity candidates where
# mfas_population_plan says it exists, the launch-per-phase schedule at three sizes beyond — once per geometry and process, on
# rank 0, and broadcasts the numbers: every rank must take the same sharding decision.  A rank's share is then priced round by
# round with the host's own round planner (split_rounds: the engine's layout query, nothing allocated).
# ------------------------------------------------------------------------------------------------------------------
def representative_conf(hp) -> np.ndarray:
    """Canonical 4-cell configuration of a geometry: cell i takes tap i of each modality (cyclically over the declared taps).  For
    the NTU widths its feature columns (7,552) equal the mean of uniformly sampled L=4 configurations."""
    s_idx = [j for j, w in enumerate(hp.s_sizes) if w > 0] or [0]
    v_idx = [j for j, w in enumerate(hp.v_sizes) if w > 0] or [0]
    return np.array([[s_idx[i % len(s_idx)], v_idx[i % len(v_idx)], i % 2] for i in range(4)], np.int64)


def split_rounds(hp, confs, device, chunk_cols: int = 0, min_candidates: int = 16):
    """How a rank's share trains: ONE population, or several resident rounds one after the other when the share does not fit the
    resident schedule as a whole (R <= 16: parameters in registers, <= ~28 conf-4-sized candidates) — every round but the last
    filled to the resident capacity, found by bisection on the engine's own layout decision (mfas_population_plan, a pure query).
    Returns [(positions into confs, resident?)].  Measured at R=16, B=20 on MI355X: 29..56 candidates take 38-55 us per train step
    with launches, 36-41 us as two resident rounds; three or four rounds only pay when the last is >= 60 % full."""
    import os
    from .engine import plan_population
    n = len(confs)
    if n == 0:
        return []

    def resident(pos):
        return bool(plan_population(hp, [confs[i] for i in pos], device, chunk_cols)["persistent"])

    everything = list(range(n))
    if hp.R > 16 or os.environ.get("MFAS_NO_ROUNDS"):
        return [(everything, hp.R <= 16 and resident(everything))]
    if resident(everything):
        return [(everything, True)]
    if n < min_candidates:
        return [(everything, False)]
    # With per-candidate sample orders (the default) a launch-per-phase population of R <= 16 has no tap-major sweep to fall back on
    # (no two candidates read the same rows): resident rounds win at EVERY size (R=16, B=20 on MI355X: 128 / 512 conf-4 candidates
    # 189 / 200 cand/s launch-per-phase against 283-321 in rounds of 28), so the share is cut into as many rounds as it takes.  With a
    # shared order the tap-major sweep is as fast as the rounds from ~64 candidates on: at most four rounds, the last >= 60 % full.
    max_rounds = 1 << 30 if hp.order_per_candidate else 4
    rounds, rest = [], everything
    while rest and len(rounds) < max_rounds:
        if resident(rest):
            rounds.append(rest)
            rest = []
            break
        lo, hi, best = 1, len(rest) - 1, 0          # largest resident prefix of `rest`
        if rounds and len(rounds[-1]) < len(rest) and resident(rest[:len(rounds[-1])]):
            lo, best = len(rounds[-1]) + 1, len(rounds[-1])          # (later rounds: start from the previous round's size)
            hi = min(hi, best + 2)
        while lo <= hi:
            mid = (lo + hi) // 2
            if resident(rest[:mid]):
                best, lo = mid, mid + 1
            else:
                hi = mid - 1
        if best == 0:
            break
        if not rounds and not hp.order_per_candidate:
            nr = -(-len(rest) // best)
            last = len(rest) - (nr - 1) * best
            if nr > 4 or (nr >= 3 and last < 0.6 * best):
                break
        rounds.append(rest[:best])
        rest = rest[best:]
    if rest or not rounds:
        return [(everything, False)]
    return [(g, True) for g in rounds]


def _interp(points, x):
    """Piecewise-linear through sorted (x, y) points; flat below the first, the last segment's slope beyond the last."""
    if not points:
        return 0.0
    if x <= points[0][0] or len(points) == 1:
        return points[0][1]
    for (x0, y0), (x1, y1) in zip(points, points[1:]):
        if x <= x1:
            return y0 + (y1 - y0) * (x - x0) / (x1 - x0)
    (x0, y0), (x1, y1) = points[-2], points[-1]
    return y1 + max(0.0, (y1 - y0) / (x1 - x0)) * (x - x1)


class StepModel:
    """Microseconds per lock-step train step of a share, measured on this device for one geometry."""

    def __init__(self, hp, device, rep_cost: int, resident, stream, calibrated: bool):
        self.hp, self.device, self.rep_cost = hp, device, rep_cost
        self.resident = sorted(resident)        # [(candidates of the canonical size, us per step)] of the resident schedule
        self.stream = sorted(stream)            # the same for launch-per-phase populations
        self.calibrated = calibrated

    def round_us(self, costs, is_resident: bool) -> float:
        n_eq = float(sum(costs)) / float(self.rep_cost)       # the round in canonical candidates
        if is_resident and self.resident:
            return _interp(self.resident, n_eq)
        return _interp(self.stream, n_eq)

    def share_us(self, confs, costs) -> float:
        if not len(confs):
            return 0.0
        cap = self.resident[-1][0] if self.resident else 0.0       # resident capacity in canonical candidates (the largest point measured)
        if self.device is None:                 # no device to ask for layouts: price the share as one round
            return self.round_us(costs, self.hp.R <= 16 and len(confs) <= cap)
        if self.hp.R > 16 or not self.resident:
            return self.round_us(costs, False)  # no resident schedule in this geometry: one launch-per-phase population
        if max(len(confs), float(sum(costs)) / float(self.rep_cost)) <= 0.8 * cap:
            return self.round_us(costs, True)   # comfortably inside the resident capacity: no layout query needed (0.2 ms each)
        return sum(self.round_us([costs[i] for i in pos], res) for pos, res in split_rounds(self.hp, confs, self.device))

    def describe(self):
        return {"calibrated": self.calibrated, "resident_us": self.resident, "launch_per_phase_us": self.stream}


_MODELS = {}


def _measure_step_us(hp, conf, K, device) -> float:
    """us per train step of K copies of `conf` trained in lockstep on `device` (difference of a long and a short run)."""
    import time
    from .engine import FeatureTable, Population
    dtype = torch.bfloat16 if hp.tap_bits == 16 else torch.float32      # (a population sized for 16-bit staging refuses f32 tables)
    T1, T2 = 40, 240
    N = T2 * hp.B
    g = torch.Generator(device=device)
    g.manual_seed(11)
    taps = {f"s{j}": torch.rand(N, w, generator=g, device=device).to(dtype) for j, w in enumerate(hp.s_sizes) if w > 0}
    taps.update({f"v{j}": torch.rand(N, w, generator=g, device=device).to(dtype) for j, w in enumerate(hp.v_sizes) if w > 0})
    label = torch.randint(0, hp.C, (N,), generator=g, device=device).to(torch.int32)
    ml = (torch.rand(N, hp.C, generator=g, device=device) < 0.2).float() if hp.loss_mode == 1 else None
    table = FeatureTable(taps, label, multilabel=ml)
    etas = np.full(T2, 1e-3)
    pop = Population(hp, [conf] * K, device, drop_seeds=list(range(K)))
    # the schedule the call will really run: with per-candidate sample orders the units take the gather path (a NULL order would
    # make the probe read rows sequentially and the launch-per-phase points optimistic)
    order = None
    if getattr(hp, "order_per_candidate", False):
        order = torch.stack([torch.stack([torch.randperm(N, generator=g, device=device) for _ in range(1)]) for _ in range(K)]).to(torch.int32)
    try:
        pop.init(list(range(1, K + 1)))
        best = {}
        for T in (T1, T2, T1, T2):
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            pop.train(table, None, 1, etas, order=order, max_steps=T)
            best[T] = min(best.get(T, 1e30), time.perf_counter() - t0)
    finally:
        pop.close()
    return max(1.0, (best[T2] - best[T1]) / (T2 - T1) * 1e6)


def step_model(hp, device=None) -> StepModel:
    """The step-time model of this geometry: calibrated on `device` (rank 0 measures, everyone receives the same numbers) when
    there is one, else the shipped constants.  MFAS_NO_CALIBRATE=1 keeps the constants."""
    import copy
    import os
    key = (hp.R, hp.C, hp.B, bool(hp.bn), bool(hp.alphas), hp.loss_mode, hp.tap_bits, tuple(hp.s_sizes), tuple(hp.v_sizes),
           bool(hp.order_per_candidate), str(device))
    if key in _MODELS:
        return _MODELS[key]
    rep = representative_conf(hp)
    rep_cost = candidate_cost(rep, hp.R, hp.s_sizes, hp.v_sizes, hp.C)
    rank, world = dist_info()
    use_device = device is not None and torch.device(device).type == "cuda" and torch.cuda.is_available() and not os.environ.get("MFAS_NO_CALIBRATE")
    res_pts, str_pts, calibrated = [], [], False
    NPT = 4
    buf = np.zeros(1 + 4 * NPT, np.float64)
    if use_device and rank == 0:
        try:
            from .engine import plan_population
            hq = copy.copy(hp)
            cap = 0
            if hp.R <= 16:
                lo, hi = 1, 96
                while lo <= hi:
                    mid = (lo + hi) // 2
                    if plan_population(hq, [rep] * mid, device)["persistent"]:
                        cap, lo = mid, mid + 1
                    else:
                        hi = mid - 1
            rk = sorted({k for k in (1, 8, 16, cap) if 1 <= k <= cap})[-NPT:]
            budget = int(1.5e9 // (12 * rep_cost))                     # bound the probe populations' memory
            sk_all = (cap + 8, 2 * cap + 8, 4 * cap + 16) if cap else (1, 4, 16, 48)
            sk = sorted({max(1, min(k, budget)) for k in sk_all})[:NPT]
            res_pts = [(float(k), _measure_step_us(hq, rep, k, device)) for k in rk]
            str_pts = [(float(k), _measure_step_us(hq, rep, k, device)) for k in sk]
            buf[0] = 1.0
            for j, (k, us) in enumerate(res_pts):
                buf[1 + 2 * j], buf[2 + 2 * j] = k, us
            for j, (k, us) in enumerate(str_pts):
                buf[1 + 2 * NPT + 2 * j], buf[2 + 2 * NPT + 2 * j] = k, us
        except Exception as e:          # the model is an optimisation: without it the call uses the shipped constants
            import warnings
            warnings.warn(f"mfas_amd: step-time calibration failed ({e!r}); using the shipped constants")
            buf[:] = 0.0
    if use_device and world > 1:
        backend = dist.get_backend(_GROUP)
        dev = torch.device("cpu") if backend == "gloo" else torch.device(device)
        t = torch.from_numpy(buf).to(dev)
        dist.broadcast(t, src=0, group=_GROUP)
        buf = t.cpu().numpy()
    if buf[0] == 1.0:
        calibrated = True
        res_pts = [(buf[1 + 2 * j], buf[2 + 2 * j]) for j in range(NPT) if buf[1 + 2 * j] > 0]
        str_pts = [(buf[1 + 2 * NPT + 2 * j], buf[2 + 2 * NPT + 2 * j]) for j in range(NPT) if buf[1 + 2 * NPT + 2 * j] > 0]
    else:                                # the constants of the box this repository was tuned on
        if hp.R <= 16:
            res_pts = [(1.0, RESIDENT_STEP_US[0][1])] + [(float(c), us) for c, us in RESIDENT_STEP_US]
            str_pts = [(float(k), max(38.0, 12.0 + 24.0 * k * rep_cost / STREAM_BYTES_PER_US)) for k in (29, 64, 128)]
        else:
            str_pts = [(float(k), predicted_step_us([rep_cost] * k, hp.R)) for k in (1, 4, 16, 48)]
    m = StepModel(hp, device if use_device else None, rep_cost, res_pts, str_pts, calibrated)
    _MODELS[key] = m
    return m


def shard_call(confs, hp, world: int, device=None, all_ranks: bool = False):
    """owner per candidate, the largest per-rank share, and the model used (None with all_ranks): what train_sampled_models shards
    a call with.  Identical on every rank (the calibration is rank 0's, the layout queries are pure)."""
    costs = [candidate_cost(c, hp.R, hp.s_sizes, hp.v_sizes, hp.C) for c in confs]
    used, model = world, None
    if world > 1 and not all_ranks and costs:
        model = step_model(hp, device)
        times = []
        for w in range(1, world + 1):
            owner = assign(costs, w)
            t = 0.0
            for r in range(w):
                idx = [i for i, o in enumerate(owner) if o == r]
                t = max(t, model.share_us([confs[i] for i in idx], [costs[i] for i in idx]))
            times.append(t)
        best = min(times)
        used = next(w for w, t in enumerate(times, 1) if t <= best * 1.03)
    owner = assign(costs, used)
    counts = [0] * world
    for o in owner:
        counts[o] += 1
    return owner, max(counts + [1]), model


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
    backend = dist.get_backend(_GROUP)
    dev = torch.device("cpu") if backend == "gloo" else (device or torch.device("cuda", torch.cuda.current_device()))
    mine = conf_digest(confs) if confs is not None else 0
    t = torch.tensor([int(seed), mine], dtype=torch.int64, device=dev)
    dist.broadcast(t, src=0, group=_GROUP)
    seed0, dig0 = (int(x) for x in t.cpu().tolist())
    if confs is not None and dig0 != mine:
        raise RuntimeError(f"rank {rank}: sampled_configurations differ from rank 0's (digest {mine:#x} vs {dig0:#x}); "
                           "the controller must be seeded identically on every rank (random / numpy / torch)")
    return seed0

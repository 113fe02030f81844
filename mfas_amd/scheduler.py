"""Learning-rate schedules of the path, engine-first.

The HIP engine consumes a whole run's learning rates up front (``eta_table(n)`` -> float64 array, turned into per-step
Adam scalars by ``adam_step_scalars``) instead of pushing one value per batch through ``optimizer.state_dict()``.
For code written against the reference (/root/reference/models/auxiliary/scheduler.py:12-46 cosine annealing with warm
restarts, :50-62 fixed rate) the same objects also answer ``step()`` / ``update_optimizer(optimizer)`` and expose the
reference's attribute names (``eta, eta_min, eta_max, Ti, Tm, Tcur, nbpe, iteration_counter``).

Rule (float64, like the reference's numpy scalars): with c the number of steps since the last restart,
``eta = eta_min + (eta_max - eta_min)/2 * (1 + cos(pi * (c/nbpe) / Ti))``; when that value reaches ``eta_min`` (within
1e-10) the period is multiplied by ``Tm`` and c starts again at 0 — the restart step itself still reports ``eta_min``.
"""

import numpy as np

_RESTART_SLACK = 1e-10
_TABLES = {}      # (scheduler state, n) -> (table, state after it): eta_table's memo


def _set_lr(optimizer, lr):
    for group in optimizer.param_groups:
        group["lr"] = lr


class LRCosineAnnealingScheduler:
    def __init__(self, eta_max, eta_min, Ti, Tmultiplier, num_batches_per_epoch):
        self.eta_max, self.eta_min = eta_max, eta_min
        self.Ti, self.Tm = Ti, Tmultiplier
        self.nbpe = num_batches_per_epoch
        self.iteration_counter = 0.0      # steps since the last restart
        self.Tcur = 0.0                   # the same in epochs
        self.eta = eta_max

    def eta_table(self, n):
        """The next ``n`` learning rates; the object ends up where ``n`` calls of ``step()`` would leave it.  Whole-run tables are
        memoised on the scheduler's state (the search asks for the same 5,000-step table in every one of its calls: 3.4 ms of a
        64 ms call of 16 candidates in this loop)."""
        n = int(n)
        key = (self.eta_max, self.eta_min, self.Ti, self.Tm, self.nbpe, self.iteration_counter, n)
        hit = _TABLES.get(key) if n > 64 else None
        if hit is not None:
            table, (self.Ti, self.Tcur, self.iteration_counter, self.eta) = hit
            return table.copy()
        table = np.empty(n, np.float64)
        half_span = 0.5 * (self.eta_max - self.eta_min)
        for k in range(n):
            self.Tcur = self.iteration_counter / self.nbpe
            self.iteration_counter += 1.0
            self.eta = self.eta_min + half_span * (1 + np.cos(np.pi * self.Tcur / self.Ti))
            table[k] = self.eta
            if self.eta <= self.eta_min + _RESTART_SLACK:
                self.Ti, self.Tcur, self.iteration_counter = self.Ti * self.Tm, 0, 0
        if n > 64:
            if len(_TABLES) >= 32:
                _TABLES.clear()
            _TABLES[key] = (table.copy(), (self.Ti, self.Tcur, self.iteration_counter, self.eta))
        return table

    def step(self):
        return float(self.eta_table(1)[0])

    def update_optimizer(self, optimizer):
        _set_lr(optimizer, self.eta)


class FixedScheduler:
    def __init__(self, lr):
        self.lr = self.eta = lr

    def eta_table(self, n):
        return np.full(int(n), self.lr, np.float64)

    def step(self):
        return self.lr

    def update_optimizer(self, optimizer):
        _set_lr(optimizer, self.lr)


def adam_step_scalars(etas, beta1=0.9, beta2=0.999):
    """float32 {lr_t/(1-beta1^t), sqrt(1-beta2^t)} per train step, formed in double like
    torch.optim.Adam's single-tensor path does before they meet a float32 tensor."""
    t = np.arange(1, len(etas) + 1, dtype=np.float64)
    ss = np.asarray(etas, np.float64) / (1.0 - beta1 ** t)
    bc2s = (1.0 - beta2 ** t) ** 0.5
    return np.stack([ss, bc2s], axis=1).astype(np.float32)

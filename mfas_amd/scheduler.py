"""Per-batch cosine annealing with warm restarts.

Mirrors ``LRCosineAnnealingScheduler`` / ``FixedScheduler`` of the reference
(/root/reference/models/auxiliary/scheduler.py:12-46, :50-62): same constructor, ``step()``,
``update_optimizer()`` and attribute names.  The engine does not round-trip an optimizer
state_dict every batch; it consumes the whole eta sequence up front (``eta_table``).
"""
import numpy as np


class LRCosineAnnealingScheduler:
    def __init__(self, eta_max, eta_min, Ti, Tmultiplier, num_batches_per_epoch):
        self.eta_min = eta_min
        self.eta_max = eta_max
        self.Ti = Ti
        self.Tcur = 0.0
        self.nbpe = num_batches_per_epoch
        self.iteration_counter = 0.0
        self.eta = eta_max
        self.Tm = Tmultiplier

    def _compute_rule(self):
        self.eta = self.eta_min + 0.5 * (self.eta_max - self.eta_min) * (1 + np.cos(np.pi * self.Tcur / self.Ti))
        return self.eta

    def step(self):
        self.Tcur = self.iteration_counter / self.nbpe
        self.iteration_counter = self.iteration_counter + 1.0
        eta = self._compute_rule()
        if eta <= self.eta_min + 1e-10:      # warm restart (scheduler.py:35-38)
            self.Tcur = 0
            self.Ti = self.Ti * self.Tm
            self.iteration_counter = 0
        return eta

    def update_optimizer(self, optimizer):
        for group in optimizer.param_groups:
            group["lr"] = self.eta

    def eta_table(self, n):
        """The next n learning rates (advances the scheduler exactly like n ``step()`` calls)."""
        out = np.empty(n, np.float64)
        for i in range(n):
            self.step()
            out[i] = self.eta
        return out


class FixedScheduler:
    def __init__(self, lr):
        self.lr = lr
        self.eta = lr

    def step(self):
        return self.lr

    def update_optimizer(self, optimizer):
        for group in optimizer.param_groups:
            group["lr"] = self.lr

    def eta_table(self, n):
        return np.full(n, self.lr, np.float64)


def adam_step_scalars(etas, beta1=0.9, beta2=0.999):
    """float32 {lr_t/(1-beta1^t), sqrt(1-beta2^t)} per train step, formed in double like
    torch.optim.Adam's single-tensor path does before they meet a float32 tensor."""
    t = np.arange(1, len(etas) + 1, dtype=np.float64)
    ss = np.asarray(etas, np.float64) / (1.0 - beta1 ** t)
    bc2s = (1.0 - beta2 ** t) ** 0.5
    return np.stack([ss, bc2s], axis=1).astype(np.float32)

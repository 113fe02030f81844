"""Mirror of /root/reference/models/search/train_searchable/ntu.py for the HIP engine.

``train_ntu_track_acc`` (ntu.py:14-89) and ``test_ntu_track_acc`` (ntu.py:92-125) keep the reference
signatures; the epoch x {train, dev} x batch loop, CE loss, Adam step and best-dev tracking all run
inside the engine for a population of one.
"""
from __future__ import annotations

import torch

from .engine import FeatureLoader, Population, best_dev_accuracy
from .ntu_searchable import _bump_bn_counters, _require_loader, make_order
from .scheduler import LRCosineAnnealingScheduler


def _adam_hyper(hp, optimizer):
    if optimizer is not None and getattr(optimizer, "param_groups", None):
        g = optimizer.param_groups[0]
        hp.wd = float(g.get("weight_decay", hp.wd))
        hp.beta1, hp.beta2 = (float(b) for b in g.get("betas", (hp.beta1, hp.beta2)))
        hp.adam_eps = float(g.get("eps", hp.adam_eps))
    return hp


def train_ntu_track_acc(model, criteria, optimizer, scheduler, dataloaders, dataset_sizes,
                        device=None, num_epochs=200, verbose=False, multitask=False):
    """Trains ``model.central_params()`` with Adam (hyper-parameters read from ``optimizer``; the engine
    starts from a fresh Adam state like a newly built torch.optim.Adam) under the per-batch LR of
    ``scheduler``; returns the best dev accuracy (0-d float64 tensor) and leaves the best-epoch weights in
    ``model`` in eval mode (ntu.py:82-87).  ``criteria`` is CrossEntropyLoss by construction of the path."""
    train_l = _require_loader(dataloaders["train"], "train", device)
    dev_l = _require_loader(dataloaders["dev"], "dev", device)
    device = torch.device(device) if device is not None else train_l.table.device
    model = model.module if isinstance(model, torch.nn.DataParallel) else model
    hp = _adam_hyper(model.hyper(multitask), optimizer)
    hp.B = train_l.batch_size
    hp.tap_bits = 8 * train_l.table.elem_size() if train_l.table.dtype == dev_l.table.dtype else 0
    N_tr, N_dev = len(train_l.table), len(dev_l.table)
    nb = -(-N_tr // hp.B)
    if isinstance(scheduler, LRCosineAnnealingScheduler):
        etas = scheduler.eta_table(num_epochs * nb)
    else:   # ntu.py:24-26: other schedulers are stepped once per epoch and never pushed to the optimizer
        for _ in range(num_epochs):
            scheduler.step()
        lr = optimizer.param_groups[0]["lr"]
        etas = [lr] * (num_epochs * nb)
    seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
    pop = Population(hp, [model.conf], device, drop_seeds=[seed & 0xFFFFFFFF])
    pop.set_params(0, model.flat_params())
    order = make_order(N_tr, num_epochs, train_l.shuffle, seed + 1, device)
    stats, _ = pop.train(train_l.table, dev_l.table, num_epochs, etas, order=order, snapshot_best=True)
    if verbose:
        for e in range(num_epochs):
            print("train Loss: {:.4f} Acc: {:.4f}".format(stats["train_loss_sum"][0, e] / N_tr,
                                                          stats["train_corrects"][0, e] / N_tr))
            print("dev Loss: {:.4f} Acc: {:.4f}".format(stats["dev_loss_sum"][0, e] / N_dev,
                                                        stats["dev_corrects"][0, e] / N_dev))
    model.load_flat(pop.get_params(0))
    _bump_bn_counters(model, num_epochs * nb)
    pop.close()
    model.train(False)
    return torch.tensor(best_dev_accuracy(stats[0], N_dev), dtype=torch.float64)


def test_ntu_track_acc(model, dataloaders, dataset_sizes, device=None, multitask=False):
    """Accuracy over dataloaders['test'] in eval mode (ntu.py:92-125)."""
    test_l = _require_loader(dataloaders["test"], "test", device)
    device = torch.device(device) if device is not None else test_l.table.device
    model = model.module if isinstance(model, torch.nn.DataParallel) else model
    model.train(False)
    pop = Population(model.hyper(multitask), [model.conf], device)
    pop.set_params(0, model.flat_params())
    _, corr = pop.forward(0, test_l.table, count=True)
    pop.close()
    return torch.tensor(corr / float(len(test_l.table)), dtype=torch.float64)

"""MM-IMDB text + poster fusion search on the HIP engine (BASELINE config 5).

The reference ships an MM-IMDB dataset, backbones, hand-designed fusion nets, the multi-label loss and a train loop,
but NO searchable network / train_sampled_models / main for MM-IMDB.  What is the reference's is
mirrored here: ``WeightedCrossEntropyWithLogits`` (/root/reference/models/central/mm_imdb.py:655-673) and
``train_mmimdb_track_f1`` (models/search/train_searchable/mmimdb.py:15-137: F1 'samples' of sigmoid(logits) > 0.3 on
the dev split, best F1 with strict '>', NaN train loss ends the run).  The searchable itself is the NTU fusion cell
chain re-sized to the MM-IMDB taps by analogy: text taps o1 (64-d) / o3 (128-d) of MaxOut_MLP
(central/mm_imdb.py:176-196), image taps 4 x 512-d of GP_VGG (:19-59), 23 genres; search space (2, 4, 2).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from . import ntu_searchable as _ntu
from .engine import FeatureLoader, Hyper, Population, best_dev_f1
from .ntu_searchable import Searchable_Skeleton_Image_Net, _bump_bn_counters, _require_loader, make_order
from .scheduler import LRCosineAnnealingScheduler

MM_TEXT_SIZES = (64, 128, 64, 128)      # two real taps; slots 2/3 alias them so the 4-slot table layout is reused
MM_IMAGE_SIZES = (512, 512, 512, 512)
MM_NUM_OUTPUTS = 23


class WeightedCrossEntropyWithLogits(nn.Module):
    """mean_{b,c}[ w_c z (-log sigmoid(x)) + (1 - z)(-log(1 - sigmoid(x))) ] — evaluated inside the engine; this module
    carries the weights (and can evaluate itself in torch for inspection)."""

    def __init__(self, pos_weight):
        super().__init__()
        self.w = list(pos_weight)

    def forward(self, logits, targets):
        q = torch.tensor(self.w, dtype=torch.float32, device=logits.device)
        x = torch.sigmoid(logits)
        return torch.mean(q * targets * -torch.log(x) + (1 - targets) * -torch.log(1 - x))


def mm_hyper(args, th_fscore=0.3) -> Hyper:
    hp = Hyper.from_args(args)
    hp.s_sizes, hp.v_sizes = MM_TEXT_SIZES, MM_IMAGE_SIZES
    hp.loss_mode, hp.f1_threshold, hp.multitask = 1, float(th_fscore), False
    return hp


class Searchable_Text_Image_Net(Searchable_Skeleton_Image_Net):
    """conf rows: [text tap (0: o1, 1: o3), image tap (0..3), non-linearity]."""

    def _sizes(self):
        return MM_TEXT_SIZES, MM_IMAGE_SIZES

    def hyper(self, multitask=None) -> Hyper:
        return mm_hyper(self.args)

    def forward(self, text, image=None):
        if image is None:           # also accept the NTU-style tuple (image, text)
            return super().forward(text)
        return super().forward((image, text))


def get_possible_layer_configurations(progression_index):
    return [[t, v, n] for t in range(2) for v in range(4) for n in range(2)]


def train_mmimdb_track_f1(model, criterion, optimizer, scheduler, dataloaders, dataset_sizes,
                          device=None, num_epochs=200, verbose=False, init_f1=0.0, th_fscore=0.3):
    """Signature of models/search/train_searchable/mmimdb.py:15-16.  Returns the best dev F1-samples (float) and leaves
    the best-epoch weights in ``model`` (eval mode)."""
    train_l = _require_loader(dataloaders["train"], "train")
    dev_l = _require_loader(dataloaders["dev"], "dev")
    device = torch.device(device) if device is not None else train_l.table.device
    hp = mm_hyper(model.args, th_fscore)
    hp.B = train_l.batch_size
    if optimizer is not None and getattr(optimizer, "param_groups", None):
        g = optimizer.param_groups[0]
        hp.wd = float(g.get("weight_decay", hp.wd))
    N_tr, N_dev = len(train_l.table), len(dev_l.table)
    nb = -(-N_tr // hp.B)
    if isinstance(scheduler, LRCosineAnnealingScheduler):
        etas = scheduler.eta_table(num_epochs * nb)
    else:
        for _ in range(num_epochs):
            scheduler.step()
        etas = [optimizer.param_groups[0]["lr"]] * (num_epochs * nb)
    seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
    pop = Population(hp, [model.conf], device, drop_seeds=[seed & 0xFFFFFFFF])
    pop.set_pos_weight(getattr(criterion, "w", np.ones(hp.C, np.float32)))
    pop.set_params(0, model.flat_params())
    order = make_order(N_tr, num_epochs, train_l.shuffle, seed + 1, device)
    pop.set_best_threshold(float(init_f1))      # best_f1 = init_f1 (mmimdb.py:18): the kept weights follow the same threshold
    stats, status = pop.train(train_l.table, dev_l.table, num_epochs, etas, order=order, snapshot_best=True)
    if verbose:
        for e in range(num_epochs):
            print("epoch #{} dev F1: {:.4f} ".format(e, stats["dev_corrects"][0, e] / float(1 << 32) / N_dev))
    model.load_flat(pop.get_params(0))
    _bump_bn_counters(model, num_epochs * nb)
    pop.close()
    model.train(False)
    return best_dev_f1(stats[0], bool(status[0]), N_dev, init_f1)


def train_sampled_models(sampled_configurations, searchable_type, dataloaders, args, device,
                         return_model=[], premodels=[], preaccuracies=[],
                         train_only_central_params=True, state_dict=dict()):
    """Population driver for the MM-IMDB searchable: same contract as the NTU one (ntu_searchable.py:23-102), returns
    the best dev F1-samples per configuration.  args.pos_weight (list of C floats) weights the positives."""
    pw = getattr(args, "pos_weight", None)
    pw = np.ones(int(args.num_outputs), np.float32) if pw is None else np.asarray(pw, np.float32)
    return _ntu.train_sampled_models(sampled_configurations, searchable_type, dataloaders, args, device,
                                     return_model=return_model, premodels=premodels, preaccuracies=preaccuracies,
                                     train_only_central_params=train_only_central_params, state_dict=state_dict,
                                     _hp=mm_hyper(args, getattr(args, "th_fscore", 0.3)), _pos_weight=pw)

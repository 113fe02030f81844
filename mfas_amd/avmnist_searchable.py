"""AV-MNIST audio + image fusion search on the HIP engine.

Mirrors /root/reference/models/search/avmnist_searchable.py: ``train_sampled_models`` :23-108,
``get_possible_layer_configurations`` :111-125 ((5, 3, 2) grid), ``Searchable_Audio_Image_Net`` :184-297 — the same
fusion cell chain as NTU with 5 audio taps (c, 2c, 4c, 8c, 16c) and 3 image taps (c, 2c, 4c), ``c = args.channels``,
cells ``[Linear, nl, Dropout]`` or plain ``[Linear, nl]`` (no BatchNorm, :276-285) and a 10-way head; the train loop
(models/search/train_searchable/avmnist.py) is the NTU loop.  Tap widths need not be multiples of 16: feature tables
pad every row with zeros (``FeatureTable`` does it), the padded weight columns stay zero.
"""
from __future__ import annotations

import numpy as np
import torch.nn as nn

from . import ntu_searchable as _ntu
from . import train_ntu as _tr
from .engine import Hyper
from .ntu_searchable import AlphaScalarMultiplication, FeatureTap, Searchable_Skeleton_Image_Net


def av_sizes(channels):
    c = int(channels)
    return (c, 2 * c, 4 * c, 8 * c, 16 * c), (c, 2 * c, 4 * c)      # audio, image (avmnist_searchable.py:289-292)


def av_hyper(args) -> Hyper:
    aud, img = av_sizes(args.channels)
    return Hyper(R=int(args.inner_representation_size), C=int(args.num_outputs), B=int(args.batchsize), bn=False,
                 drpt=float(args.drpt), alphas=bool(args.alphas), multitask=bool(getattr(args, "multitask", False)),
                 s_sizes=aud, v_sizes=img, allow_plain_cell=True)


class Searchable_Audio_Image_Net(Searchable_Skeleton_Image_Net):
    """conf rows: [audio tap 0..4, image tap 0..2, non-linearity].  Attributes as in the reference (:191-200):
    ``conf, args, rgbnet, audnet, alphas, fusion_layers, central_classifier``."""
    _construction_is_standard = False      # ([Linear, nl(, Dropout)] cells built by its own _create_fc_layers: candidates are initialised through the module)

    def __init__(self, args, conf):
        super().__init__(args, conf)
        self.audnet = self.skenet       # the audio backbone stand-in (tensor_tuple = (image, sound), :206-208)

    def _sizes(self):
        return av_sizes(self.args.channels)

    def _create_fc_layers(self):
        layers = []
        for i, conf in enumerate(self.conf):
            in_size = self.alphas[i].size_alpha_x + self.alphas[i].size_alpha_y
            if i > 0:
                in_size += self.args.inner_representation_size
            nl = {0: nn.ReLU, 1: nn.Sigmoid, 2: nn.LeakyReLU}[int(conf[2])]()
            if self.args.drpt > 1e-10:
                op = nn.Sequential(nn.Linear(in_size, self.args.inner_representation_size), nl, nn.Dropout(self.args.drpt))
            else:
                op = nn.Sequential(nn.Linear(in_size, self.args.inner_representation_size), nl)
            layers.append(op)
        return nn.ModuleList(layers)

    def hyper(self, multitask=None) -> Hyper:
        hp = av_hyper(self.args)
        if multitask is not None:
            hp.multitask = bool(multitask)
        return hp


def get_possible_layer_configurations(progression_index):
    return [[t, v, n] for t in range(5) for v in range(3) for n in range(2)]


get_central_states = _ntu.get_central_states
set_central_states = _ntu.set_central_states


def train_sampled_models(sampled_configurations, searchable_type, dataloaders, args, device,
                         return_model=[], premodels=[], preaccuracies=[],
                         train_only_central_params=True, state_dict=dict()):
    """avmnist_searchable.py:23-108 on the lockstep engine (same contract as the NTU driver)."""
    hp = av_hyper(args)
    if getattr(args, "multitask", False):
        hp.multitask = False     # the engine-side population path trains the central head only (see ntu driver)
    return _ntu.train_sampled_models(sampled_configurations, searchable_type, dataloaders, args, device,
                                     return_model=return_model, premodels=premodels, preaccuracies=preaccuracies,
                                     train_only_central_params=train_only_central_params, state_dict=state_dict, _hp=hp)


train_avmnist_track_acc = _tr.train_ntu_track_acc     # train_searchable/avmnist.py:14-84 == the NTU loop
test_avmnist_track_acc = _tr.test_ntu_track_acc       # train_searchable/avmnist.py:87-119

"""Host-side mirror of /root/reference/models/search/ntu_searchable.py for the HIP engine.

Same names, argument meaning and error behaviour as the reference for the hot path:

* ``train_sampled_models``               ntu_searchable.py:23-102   (population loop -> lockstep engine)
* ``get_possible_layer_configurations``  ntu_searchable.py:105-119
* ``get_central_states`` / ``set_central_states``   ntu_searchable.py:123-174   (weight sharing)
* ``Searchable_Skeleton_Image_Net``      ntu_searchable.py:178-301  (nn.Module surface)

Departures: backbones are feature tables (``FeatureTap`` stands in for
``Visual``/``Skeleton``; north_star: precomputed features); candidates of one call train in
lockstep and share the epoch's batch order; dropout uses the engine's counter-based stream.
"""
from __future__ import annotations

import math
from typing import Dict, List

import os

import numpy as np
import torch
import torch.nn as nn

from . import population as popmod
from .engine import (S_SIZES, V_SIZES, FeatureLoader, FeatureTable, Hyper, Population, best_dev_accuracy,
                     best_dev_f1, flat_layout, plan_population)
from .scheduler import LRCosineAnnealingScheduler


# ------------------------------------------------------------------------------------------------ small modules
class GlobalPooling2D(nn.Module):
    """models/auxiliary/aux_models.py:54-64: mean over all trailing dims (identity on pooled (B,C) taps)."""

    def forward(self, x):
        x = x.view(x.size(0), x.size(1), -1)
        return torch.mean(x, 2).view(x.size(0), -1)


class AlphaScalarMultiplication(nn.Module):
    """aux_models.py:94-111: x*sigmoid(alpha), y*(1-sigmoid(alpha)); one scalar per cell.  The arithmetic
    runs inside the engine; this module only owns the parameter."""

    def __init__(self, size_alpha_x, size_alpha_y):
        super().__init__()
        self.size_alpha_x = size_alpha_x
        self.size_alpha_y = size_alpha_y
        self.alpha_x = nn.Parameter(torch.zeros(1, dtype=torch.float32))


class FeatureTap(nn.Module):
    """Stand-in for the frozen backbones ``Visual`` / ``Skeleton`` (models/central/ntu.py:17,53): the
    taps are precomputed, so this module has no parameters and simply hands the batch's taps on."""

    def __init__(self, prefix):
        super().__init__()
        self.prefix = prefix

    def forward(self, x):
        return x


class _TrainModeForward(torch.autograd.Function):
    """Train-mode forward of one batch on the engine WITH an autograd edge to the central parameters: forward =
    mfas_population_forward_train on a scratch population holding the module's parameters; backward = mfas_population_backward on
    a scratch population of the same state (same dropout stream, same batch statistics) with the incoming dL/dlogits — the
    engine's own fused backward, not a torch graph.  ntu_searchable.py:206-247 under model.train(True), for callers that write
    their own loop.  Nothing of the GPU is kept between the two calls (the backward pass runs the batch again anyway): a forward
    under no_grad, or one whose loss is never backpropagated, holds no device memory."""

    @staticmethod
    def _scratch(module, table, n, seed, flat0):
        hp = module.hyper(False)
        hp.B = max(n, 2)
        pop = Population(hp, [module.conf], table.label.device, drop_seeds=[seed & 0xFFFFFFFF])
        pop.set_params(0, flat0)
        return pop

    @staticmethod
    def forward(ctx, module, table, n, seed, *params):
        flat0 = module.flat_params()
        pop = _TrainModeForward._scratch(module, table, n, seed, flat0)
        try:
            out = pop.forward_train(0, table, 0, n, step=0)
            if module.args.batchnorm:
                module.load_flat(pop.get_params(0))      # the moved running statistics (nothing else changed)
                _bump_bn_counters(module, 1)
        finally:
            pop.close()
        ctx.table, ctx.n, ctx.seed, ctx.flat0, ctx.module = table, n, seed, flat0, module
        return out

    @staticmethod
    def backward(ctx, grad_out):
        module = ctx.module
        pop = _TrainModeForward._scratch(module, ctx.table, ctx.n, ctx.seed, ctx.flat0)     # the state the forward saw
        try:
            gflat = pop.backward(0, ctx.table, grad_out, 0, ctx.n, step=0).cpu()
        finally:
            pop.close()
        where = {key: (shape, off) for key, shape, off in flat_layout(module.conf, module.hyper(False))[0]}
        grads = []
        for key, prm in module.named_parameters():
            # (alphas.parameters() sit in the optimiser but never see a gradient without --alphas, ntu_searchable.py:251)
            if key not in where or (key.startswith("alphas") and not module.args.alphas):
                grads.append(None)
                continue
            shape, off = where[key]
            grads.append(gflat[off:off + int(np.prod(shape))].reshape(shape).to(prm.device))
        return (None, None, None, None, *grads)


# ------------------------------------------------------------------------------------------------ the fusion net
class Searchable_Skeleton_Image_Net(nn.Module):
    """Module surface of ntu_searchable.py:178-301: attributes ``conf, args, rgbnet, skenet, alphas, gp_v,
    gp_s, fusion_layers, central_classifier``; ``forward``; ``central_params``.

    conf: one row per fusion cell: [ske tap, rgb tap, non-linearity (0 ReLU / 1 Sigmoid / 2 LeakyReLU)].
    """
    # construction draws torch's generator exactly like initial_flat_params restates it (per cell Linear weight, bias; classifier;
    # then the alphas): train_sampled_models may then initialise candidates of this class without building modules.  A subclass that
    # constructs differently (other layers, another order) sets this to False.
    _construction_is_standard = True

    def __init__(self, args, conf):
        super().__init__()
        self.conf = np.asarray(conf)
        self.args = args
        self.rgbnet = FeatureTap("v")
        self.skenet = FeatureTap("s")
        self.alphas = self._create_alphas()
        self.gp_v, self.gp_s = self._create_global_poolings()
        self.fusion_layers = self._create_fc_layers()
        self.central_classifier = nn.Linear(self.args.inner_representation_size, args.num_outputs)
        for m in self.modules():
            if isinstance(m, AlphaScalarMultiplication):
                nn.init.normal_(m.alpha_x, 0.0, 0.1)
        self._pop = None

    # -- construction (ntu_searchable.py:258-301)
    def _sizes(self):
        hp = Hyper.from_args(self.args)
        return hp.s_sizes, hp.v_sizes

    def _create_alphas(self):
        sizes_ske, sizes_ims = self._sizes()
        return nn.ModuleList([AlphaScalarMultiplication(sizes_ske[int(c[0])], sizes_ims[int(c[1])])
                              for c in self.conf])

    def _create_global_poolings(self):
        n = len(self.conf)
        return (nn.ModuleList([GlobalPooling2D() for _ in range(n)]),
                nn.ModuleList([GlobalPooling2D() for _ in range(n)]))

    def _create_fc_layers(self):
        layers = []
        for i, conf in enumerate(self.conf):
            in_size = self.alphas[i].size_alpha_x + self.alphas[i].size_alpha_y
            if i > 0:
                in_size += self.args.inner_representation_size
            out_size = self.args.inner_representation_size
            if conf[2] == 0:
                nl = nn.ReLU()
            elif conf[2] == 1:
                nl = nn.Sigmoid()
            elif conf[2] == 2:
                nl = nn.LeakyReLU()
            else:
                raise ValueError(f"unknown non-linearity {conf[2]}")
            drop, bn = self.args.drpt > 1e-10, bool(self.args.batchnorm)
            if drop and bn:
                op = nn.Sequential(nn.Linear(in_size, out_size), nl, nn.BatchNorm1d(out_size),
                                   nn.Dropout(self.args.drpt))
            elif drop:
                op = nn.Sequential(nn.Linear(in_size, out_size), nl, nn.Dropout(self.args.drpt))
            elif bn:
                op = nn.Sequential(nn.Linear(in_size, out_size), nl, nn.BatchNorm1d(out_size))
            else:   # the reference leaves `op` unassigned here (UnboundLocalError, :274-284)
                raise ValueError("drpt < 1e-10 without --batchnorm is not a legal cell "
                                 "(reference: UnboundLocalError at ntu_searchable.py:284)")
            layers.append(op)
        return nn.ModuleList(layers)

    def central_params(self):
        return [{"params": self.alphas.parameters()},
                {"params": self.fusion_layers.parameters()},
                {"params": self.central_classifier.parameters()}]

    # -- engine plumbing
    def hyper(self, multitask=None) -> Hyper:
        hp = Hyper.from_args(self.args)
        if multitask is not None:
            hp.multitask = bool(multitask)
        return hp

    def flat_params(self) -> torch.Tensor:
        layout, n = flat_layout(self.conf, self.hyper())
        sd = self.state_dict()
        flat = torch.zeros(n, dtype=torch.float32)
        for key, shape, off in layout:
            flat[off:off + int(np.prod(shape))] = sd[key].detach().reshape(-1).to("cpu", torch.float32)
        return flat

    def load_flat(self, flat: torch.Tensor):
        layout, _ = flat_layout(self.conf, self.hyper())
        sd = self.state_dict()
        flat = flat.detach().cpu()
        with torch.no_grad():
            for key, shape, off in layout:
                sd[key].copy_(flat[off:off + int(np.prod(shape))].reshape(shape))

    def _engine(self, device, multitask):
        hp = self.hyper(multitask)
        key = (str(device), hp.multitask)
        if self._pop is None or self._pop[0] != key:
            self._pop = (key, Population(hp, [self.conf], device))
        return self._pop[1]

    def forward(self, tensor_tuple):
        """tensor_tuple = (rgb, ske): dict-likes holding the pooled taps v0..v3 [+ 'vlogit'] and s0..s3
        [+ 'slogit'] on a HIP device.  Eval mode: running-statistics BatchNorm, no Dropout (mfas_population_forward).  Train mode:
        one batch of <= 64 samples with batch statistics and Dropout, differentiable with respect to the central parameters
        (_TrainModeForward).  The taps are outputs of frozen backbones here (feature tables): no gradient flows into them, and
        taps that require grad are refused instead of being silently cut off.  The fast path for training stays
        train_sampled_models / train_ntu_track_acc."""
        image, skeleton = tensor_tuple[0], tensor_tuple[1]
        visual = self.rgbnet(image)
        skel = self.skenet(skeleton)
        taps = {k: v for k, v in visual.items() if k[0] == "v" and k[1:].isdigit()}
        taps.update({k: v for k, v in skel.items() if k[0] == "s" and k[1:].isdigit()})
        if not taps:
            raise ValueError("forward needs the pooled taps 'v0'.. / 's0'.. of the batch (there are no backbones in this engine)")
        some = next(iter(taps.values()))
        n = some.shape[0]
        if self.training and torch.is_grad_enabled() and any(t.requires_grad for t in taps.values()):
            raise NotImplementedError("the taps require grad, but this engine's backbones are frozen feature tables: it produces gradients "
                                      "for central_params() only (train_only_central_params, ntu_searchable.py:27); detach() the taps")
        # (eval mode needs no gradient of the taps: the reference's eval forward accepts them, so they are detached silently)
        taps = {k: v.detach() for k, v in taps.items()}
        table = FeatureTable(taps, torch.zeros(n, dtype=torch.int32, device=some.device))
        if self.training:
            # train mode (ntu_searchable.py:206-247 under model.train(True)): batch-statistics BatchNorm — running statistics and
            # num_batches_tracked move — and Dropout (the engine's counter-based stream, seeded from torch's RNG).  The logits carry
            # an autograd edge to the central parameters (_TrainModeForward: loss.backward() runs the engine's fused backward);
            # the fast path for training stays train_ntu_track_acc / train_sampled_models.
            if not 1 <= n <= 64:
                raise NotImplementedError("train-mode forward handles one batch of 1..64 samples (the engine's batch range)")
            if n == 1 and self.args.batchnorm:
                raise ValueError("Expected more than 1 value per channel when training")     # torch BatchNorm1d's message
            seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
            out = _TrainModeForward.apply(self, table, n, seed, *[p for _, p in self.named_parameters()])
            if not self.args.multitask:
                return out
            return out, visual["vlogit"], skel["slogit"]
        pop = self._engine(some.device, False)
        pop.set_params(0, self.flat_params())
        out = pop.forward(0, table)
        if not self.args.multitask:
            return out
        return out, visual["vlogit"], skel["slogit"]


# ------------------------------------------------------------------------------------------------ search helpers
def get_possible_layer_configurations(progression_index):
    """ntu_searchable.py:105-119: the (4,4,2) grid, non-linearity fastest."""
    max_labels = (4, 4, 2)
    return [[ti, vi, ni] for ti in range(max_labels[0]) for vi in range(max_labels[1])
            for ni in range(max_labels[2])]


def _share_name(idx_layer, layer, conf_row):
    name = str(idx_layer) + ".L_" + str(layer[0].in_features) + "_" + str(layer[0].out_features)
    return name + {0: ".A_relu", 1: ".A_sigmoid", 2: ".A_lrelu"}[int(conf_row[2])]


def get_central_states(model, state_dict, using_dataparallel=False):
    """ntu_searchable.py:123-149: publish every cell under ``"{idx}.L_{in}_{out}.A_{act}"``."""
    model = model.module if using_dataparallel else model
    for idx_layer, layer in enumerate(model.fusion_layers):
        state_dict[_share_name(idx_layer, layer, model.conf[idx_layer])] = \
            {k: v.detach().clone() for k, v in layer.state_dict().items()}
    return state_dict


def set_central_states(model, state_dict, using_dataparallel=False):
    """ntu_searchable.py:152-174: load every cell whose key exists."""
    model = model.module if using_dataparallel else model
    for idx_layer, layer in enumerate(model.fusion_layers):
        name = _share_name(idx_layer, layer, model.conf[idx_layer])
        if name in state_dict:
            layer.load_state_dict(state_dict[name])


# ------------------------------------------------------------------------------------------------ the driver
PROFILE: List[tuple] = []   # (launches, total ms, algorithmic bytes/launch) of k_sweep per call when args.engine_profile


def _require_loader(x, name, device=None) -> FeatureLoader:
    """SURVEY §8(b) dataloaders contract.  Fast path: a FeatureLoader over a HIP-resident table.  Otherwise any iterable
    of reference-shaped batches ``{'rgb': {v0..v3[, vlogit]}, 'ske': {s0..s3[, slogit]}, 'label'}`` whose taps are
    already pooled to (B, width) (train_searchable/ntu.py:35-43 reads exactly these keys): it is drained ONCE into a
    device-resident FeatureTable (cached on the loader object) and from then on only its batch size and shuffle flag
    matter.  Raw video / skeleton tensors are rejected: there are no backbones here."""
    if isinstance(x, FeatureLoader):
        return x
    cached = getattr(x, "_mfas_feature_loader", None)
    if cached is not None:
        return cached
    if device is None or not hasattr(x, "__iter__"):
        raise TypeError(f"dataloaders['{name}'] must be a mfas_amd.FeatureLoader over a HIP-resident FeatureTable (or an "
                        "iterable of pooled-tap batches); there is no raw-video / CPU path")
    taps, labels, extra = {}, [], {}
    bsz = 0
    for batch in x:
        rgb, ske = batch["rgb"], batch["ske"]
        if not hasattr(rgb, "items") or not hasattr(ske, "items"):
            raise TypeError(f"dataloaders['{name}']: 'rgb' / 'ske' must map tap names (v0.., s0..) to pooled (B, width) "
                            "tensors; raw video needs the frozen backbones, which are outside this engine")
        for src in (rgb, ske):
            for k, v in src.items():
                if v is None:
                    continue
                v = torch.as_tensor(v)
                if k in ("vlogit", "slogit"):
                    extra.setdefault(k, []).append(v.reshape(v.shape[0], -1).float())
                elif len(k) >= 2 and k[0] in "sv" and k[1:].isdigit():
                    if v.dim() > 2:          # GlobalPooling2D (aux_models.py:58-64) for un-pooled maps: the k_pool kernel
                        from .pooling import global_pool
                        v = global_pool(v.to(torch.device(device)), torch.float32)
                    taps.setdefault(k, []).append(v)
        lab = torch.as_tensor(batch["label"]).reshape(-1)
        labels.append(lab)
        bsz = max(bsz, int(lab.numel()))
    if not labels:
        raise ValueError(f"dataloaders['{name}'] yielded no batches")
    dev = torch.device(device)
    table = FeatureTable({k: torch.cat(v).to(dev) for k, v in taps.items()},
                         torch.cat(labels).to(dev).to(torch.int32),
                         vlogit=torch.cat(extra["vlogit"]).to(dev) if "vlogit" in extra else None,
                         slogit=torch.cat(extra["slogit"]).to(dev) if "slogit" in extra else None)
    sampler = getattr(x, "sampler", None)
    shuffle = sampler is not None and type(sampler).__name__ == "RandomSampler"
    fl = FeatureLoader(table, int(getattr(x, "batch_size", None) or bsz), shuffle=shuffle)
    try:
        x._mfas_feature_loader = fl
    except Exception:
        pass
    return fl


def make_order(N, epochs, shuffle, seed, device):
    """Per-epoch sample order shared by the population (DataLoader(shuffle=True), models/searchable.py:248)."""
    if not shuffle:
        return None
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed))
    return torch.stack([torch.randperm(N, generator=gen, device=device) for _ in range(epochs)]).to(torch.int32)


def _mix64(z):
    """splitmix64's finaliser on int64 tensors (wrapping multiplies; logical right shifts written as arithmetic shift + mask)."""
    z = (z ^ ((z >> 30) & 0x3FFFFFFFF)) * -4658895280553007687          # 0xBF58476D1CE4E5B9
    z = (z ^ ((z >> 27) & 0x1FFFFFFFFF)) * -7723592293110705685         # 0x94D049BB133111EB
    return z ^ ((z >> 31) & 0x1FFFFFFFF)


def make_order_per_candidate(N, epochs, shuffle, seed, device, indices):
    """args.engine_order = "per_candidate" (the default): the reference's behaviour — every candidate iterates its OWN freshly
    shuffled DataLoader each epoch (models/searchable.py:248-250, train_searchable/ntu.py:35).  Candidate i's permutations are a
    function of (seed, i, epoch) only, so they do not depend on the world size or on which round / rank trains it: every sample
    position gets a 64-bit counter-based key (seed, candidate, epoch, position -> splitmix64) and the permutation is the argsort of
    the keys — ONE batched device sort for the whole share instead of K x E randperm calls (a tie between two of N 64-bit keys has
    probability N^2 / 2^65).  Returns int32 [len(indices)][E][N]."""
    if not shuffle:
        return None
    ep = torch.arange(epochs, dtype=torch.int64, device=device).view(1, -1, 1)
    pos = torch.arange(N, dtype=torch.int64, device=device).view(1, 1, -1)
    s64 = int(seed) & 0x7FFFFFFFFFFFFFFF
    idx = [int(i) for i in indices]
    out = torch.empty((len(idx), epochs, N), dtype=torch.int32, device=device)
    per = max(1, (32 << 20) // max(1, epochs * N))          # <= 32 M keys (256 MB of int64 + the sort's buffers) per slice
    for k0 in range(0, len(idx), per):
        cand = torch.as_tensor(idx[k0:k0 + per], dtype=torch.int64, device=device).view(-1, 1, 1)
        stream = _mix64((cand + 1) * 0x632BE59BD9B4E019 + (ep + 1) * 0x2545F4914F6CDD1D + s64)   # one stream per (seed, candidate, epoch)
        keys = _mix64(stream + pos * -7046029254386353131)                                        # 0x9E3779B97F4A7C15
        out[k0:k0 + per] = torch.argsort(keys, dim=-1)
    return out


def initial_flat_params(args, conf, hp=None, generator=None, out=None) -> torch.Tensor:
    """The flat parameter vector `Searchable_Skeleton_Image_Net(args, conf).flat_params()` would hold right after
    construction — same draws from torch's global RNG in the same order (per cell Linear weight: kaiming_uniform_(a=sqrt(5)),
    bias: U(+-1/sqrt(fan_in)); the classifier likewise; then alpha_i ~ N(0, 0.1)), BatchNorm at its defaults — without
    building the ~20 module objects per candidate (train_sampled_models initialises every sampled candidate this way:
    ≈ 1 ms per candidate, a fifth of a 50-candidate call's time at R=16).  `generator`: draw from this torch.Generator instead of
    the global stream (seeded alike it yields the same numbers)."""
    import math
    conf = np.asarray(conf).reshape(-1, 3)
    hp = hp if hp is not None else Hyper.from_args(args)
    layout, n = flat_layout(conf, hp)
    where = {key: (shape, off) for key, shape, off in layout}
    if out is None:
        flat = torch.zeros(n, dtype=torch.float32)
    else:                       # fill the caller's (pinned) slice
        flat = out
        assert flat.numel() == n and flat.dtype == torch.float32
        flat.zero_()

    def view(key):
        shape, off = where[key]
        return flat[off:off + int(np.prod(shape))].view(shape)

    kgain = math.sqrt(2.0 / (1 + math.sqrt(5) ** 2))      # init.calculate_gain('leaky_relu', sqrt(5)), as nn.Linear.reset_parameters

    def linear(wkey, bkey):
        w, b = view(wkey), view(bkey)
        fan_in = w.shape[1]
        w.uniform_(-(math.sqrt(3.0) * (kgain / math.sqrt(fan_in))), math.sqrt(3.0) * (kgain / math.sqrt(fan_in)), generator=generator)   # kaiming_uniform_'s bound, same expression
        bound = 1.0 / math.sqrt(fan_in) if fan_in > 0 else 0.0
        b.uniform_(-bound, bound, generator=generator)

    with torch.no_grad():
        for i, c in enumerate(conf):
            if int(c[2]) not in (0, 1, 2):
                raise ValueError(f"unknown non-linearity {c[2]}")
            if not (args.drpt > 1e-10 or bool(args.batchnorm)):
                raise ValueError("drpt < 1e-10 without --batchnorm is not a legal cell "
                                 "(reference: UnboundLocalError at ntu_searchable.py:284)")
            linear(f"fusion_layers.{i}.0.weight", f"fusion_layers.{i}.0.bias")
            if hp.bn:
                view(f"fusion_layers.{i}.2.weight").fill_(1.0)
                view(f"fusion_layers.{i}.2.running_var").fill_(1.0)
        linear("central_classifier.weight", "central_classifier.bias")
        for i in range(len(conf)):
            view(f"alphas.{i}.alpha_x").normal_(0.0, 0.1, generator=generator)      # nn.init.normal_
    return flat


ROUNDS_MIN_CANDIDATES = 16         # (a share that does not fit the resident schedule has at least ~29 conf-4-sized candidates, or ~50 shallow ones)


def _plan_rounds(hp, confs, mine, device, seed_base, chunk_cols):
    """The rank's share as ONE population, or as several resident rounds trained one after the other (population.split_rounds:
    the decision is the engine's own layout query, mfas_population_plan — nothing is allocated for it; only the populations
    that train are ever created).  Candidates are independent and carry their own seeds, so the split changes nothing but the
    column-chunk summation order (as any change of population size does).  Yields (indices, Population)."""
    if not mine:
        return
    for pos, planned_resident in popmod.split_rounds(hp, [confs[i] for i in mine], device, chunk_cols, ROUNDS_MIN_CANDIDATES):
        idx = [mine[j] for j in pos]
        pop = Population(hp, [confs[i] for i in idx], device, drop_seeds=[(seed_base * 7 + i) & 0xFFFFFFFF for i in idx],
                         chunk_cols=chunk_cols)
        if bool(pop.schedule()["persistent"]) != bool(planned_resident) and hp.R <= 16:
            import warnings      # (MFAS_PERSIST* read at different times, or a plan / create disagreement: results are unaffected)
            warnings.warn(f"mfas_amd: the layout query planned {'a resident' if planned_resident else 'a launch-per-phase'} round of "
                          f"{len(idx)} candidates but the population was created {'resident' if not planned_resident else 'launch-per-phase'}")
        yield idx, pop


def torch_init_bounds(conf, hp) -> np.ndarray:
    """The Tensor.uniform_(-bound, bound) bounds nn.Linear.reset_parameters uses for every cell and for the classifier (weight:
    kaiming_uniform_(a = sqrt 5), bias: 1 / sqrt(fan_in)), as float32 [2 * (4 + 1)]: what mfas_population_init_torch_streams
    takes (the expressions of initial_flat_params, evaluated in Python doubles and rounded to float32 like torch does)."""
    out = np.zeros(10, np.float32)
    kgain = math.sqrt(2.0 / (1 + math.sqrt(5) ** 2))
    conf = np.asarray(conf).reshape(-1, 3)
    fans = [hp.s_sizes[int(c[0])] + hp.v_sizes[int(c[1])] + (hp.R if i else 0) for i, c in enumerate(conf)]
    for i, fan_in in enumerate(fans):
        out[2 * i] = math.sqrt(3.0) * (kgain / math.sqrt(fan_in))
        out[2 * i + 1] = 1.0 / math.sqrt(fan_in)
    out[8] = math.sqrt(3.0) * (kgain / math.sqrt(hp.R))
    out[9] = 1.0 / math.sqrt(hp.R)
    return out


_TEST_FAIL_RANK = -1         # tests only: the rank whose first share of a sharded call raises (never read from the environment)
_DEVICE_STREAMS_OK = {}      # device -> the device-side Mersenne-Twister init reproduces torch's CPU draws on this host (checked once)


def _init_population_device_streams(pop, args, confs, group, hp, seed_base, device, searchable_type) -> bool:
    """The default initialisation, generated on the GPU: mfas_population_init_torch_streams runs torch's own generator
    (at::mt19937 + uniform_real_distribution<float>) per candidate under the seeds the host path uses.  Verified ONCE per process
    and device against torch itself — the first candidate's flat parameters must equal initial_flat_params bit for bit (torch's CPU
    kernels fuse x * (hi - lo) + lo into one fma on AVX2 / AVX512 hosts; a build that does not would differ in last bits) — and
    abandoned for the host path when that check fails (MFAS_HOST_INIT=1 forces the host path)."""
    import os
    # one check per device, class (qualified) and geometry incl. the candidate's depth: a later call with another Hyper, another L or
    # an unrelated class of the same __name__ is checked again (one module build per key)
    key = (str(device), getattr(searchable_type, "__module__", ""), getattr(searchable_type, "__qualname__", searchable_type.__name__),
           hp.R, hp.C, bool(hp.bn), bool(hp.alphas), tuple(hp.s_sizes), tuple(hp.v_sizes), len(np.asarray(confs[group[0]]).reshape(-1, 3)))
    if os.environ.get("MFAS_HOST_INIT") or _DEVICE_STREAMS_OK.get(key) is False:
        return False
    seeds = [(seed_base + 2 + i) & 0xFFFFFFFFFFFFFFFF for i in group]
    bounds = np.stack([torch_init_bounds(confs[i], hp) for i in group])
    pop.init_torch_streams(seeds, bounds)
    if key not in _DEVICE_STREAMS_OK:
        with torch.random.fork_rng(devices=[]):           # the check builds the real module ONCE per class and device
            torch.manual_seed(seed_base + 2 + group[0])
            want = searchable_type(args, confs[group[0]]).flat_params()
        ok = bool(torch.equal(pop.get_params(0).cpu(), want))
        _DEVICE_STREAMS_OK[key] = ok
        if not ok:
            import warnings
            warnings.warn("mfas_amd: the device-side torch random stream does not reproduce this host's torch.uniform_ draws; "
                          "initialising candidates on the host instead")
            return False
    return True


_STAGING = {}      # device -> pinned host staging buffer of the initial parameters (grow-only: a 128-candidate R=128 call needs 0.5 GB)


def _staging(device, n):
    buf = _STAGING.get(str(device))
    if buf is None or buf.numel() < n:
        buf = torch.empty(max(n, 1 << 20), dtype=torch.float32, pin_memory=torch.cuda.is_available())
        _STAGING[str(device)] = buf
    return buf[:n]


def _init_population_from_torch(pop, args, confs, group, hp, seed_base, searchable_type, return_model, mods, device):
    """train_sampled_models' default initialisation: every candidate starts from what constructing its module under
    torch.manual_seed(seed_base + 2 + i) draws (ntu_searchable.py:44 builds the model from torch's global stream).  Module-free
    candidates draw from a PRIVATE torch.Generator seeded the same way (same Mersenne-Twister stream, same numbers:
    tests/test_host_cpu.py), which makes the fills independent of each other: a small thread pool fills the candidates' slices of
    ONE pinned staging buffer side by side (torch's uniform_ is serial and releases the GIL; 1 M draws per R=128 candidate), one
    host-to-device copy moves the share, and the per-candidate repacking kernels (mfas_population_set_params) are queued without a
    host synchronisation in between."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    # (the fills are tiny for torch's intra-op pool: with every host core in it their fork/join dominates — 6 ms instead of 0.7 ms
    #  per candidate on a 256-thread host; the values do not depend on the thread count)
    nthreads = torch.get_num_threads()
    if nthreads > 4:
        torch.set_num_threads(4)
    try:
        if return_model or not getattr(searchable_type, "_construction_is_standard", False):
            for j, i in enumerate(group):
                with torch.random.fork_rng(devices=[]):
                    torch.manual_seed(seed_base + 2 + i)   # world-size independent per-candidate stream
                    m = searchable_type(args, confs[i])
                if return_model:
                    mods[i] = m
                pop.set_params(j, m.flat_params())
            return
        sizes = [flat_layout(confs[i], hp)[1] for i in group]
        offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        host = _staging(device, int(offs[-1]))

        def one(j):
            g = torch.Generator()
            g.manual_seed(seed_base + 2 + group[j])
            initial_flat_params(args, confs[group[j]], hp, generator=g, out=host[offs[j]:offs[j + 1]])

        workers = max(1, min(8, (os.cpu_count() or 1) // 2, len(group)))
        if workers == 1:
            for j in range(len(group)):
                one(j)
        else:
            with ThreadPoolExecutor(workers) as ex:
                list(ex.map(one, range(len(group))))
        dev_flat = host.to(device, non_blocking=True)
        for j in range(len(group)):
            pop.set_params(j, dev_flat[offs[j]:offs[j + 1]], sync=False)
        if torch.device(device).type == "cuda":
            torch.cuda.current_stream(device).synchronize()      # the staging buffer is free for the next call; dev_flat may go
    finally:
        if nthreads > 4:
            torch.set_num_threads(nthreads)


def train_sampled_models(sampled_configurations, searchable_type, dataloaders, args, device,
                         return_model=[], premodels=[], preaccuracies=[],
                         train_only_central_params=True, state_dict=dict(), _hp=None, _pos_weight=None):
    """Drop-in for ntu_searchable.py:23-102 (see _train_sampled_models); the call is one profiler range (roctx marker)."""
    from . import _lib
    with _lib.profiler_range(f"train_sampled_models K={len(sampled_configurations)}"):
        return _train_sampled_models(sampled_configurations, searchable_type, dataloaders, args, device, return_model, premodels,
                                     preaccuracies, train_only_central_params, state_dict, _hp, _pos_weight)


def _train_sampled_models(sampled_configurations, searchable_type, dataloaders, args, device,
                          return_model=[], premodels=[], preaccuracies=[],
                          train_only_central_params=True, state_dict=dict(), _hp=None, _pos_weight=None):
    """Drop-in for ntu_searchable.py:23-102.  Every configuration is trained from scratch for
    ``args.epochs`` epochs of {train over dataloaders['train'], eval over dataloaders['dev']} and its best dev
    accuracy returned, in input order.  The whole (per-rank share of the) population trains in lockstep inside
    the HIP engine; with torch.distributed initialised the population is sharded across ranks and the
    accuracies are all-gathered (RCCL).
    """
    if preaccuracies:   # the reference forwards init_f1=, which train_ntu_track_acc does not accept (:86-89)
        raise TypeError("train_ntu_track_acc() got an unexpected keyword argument 'init_f1' "
                        "(reference behaviour, ntu_searchable.py:86-89)")
    if not train_only_central_params:
        raise NotImplementedError("backbone fine-tuning needs raw video; the engine trains central_params() only")
    device = torch.device(device)
    train_l = _require_loader(dataloaders["train"], "train", device)
    dev_l = _require_loader(dataloaders["dev"], "dev", device)
    hp = _hp if _hp is not None else Hyper.from_args(args)
    hp.multitask = False    # ntu_searchable.py:82-84 never forwards multitask to the train loop
    hp.tap_bits = 8 * train_l.table.elem_size() if train_l.table.dtype == dev_l.table.dtype else 0
    # sample order: "per_candidate" (default) = the reference's behaviour, every candidate iterates its own freshly shuffled loader
    # (models/searchable.py:248-250, train_searchable/ntu.py:35); "shared" = one shuffle per epoch for the whole call (lockstep): what
    # lets the tap-major sweep stage a batch's rows once for several candidates at R < 128 (costs / gains per workload)
    per_cand = getattr(args, "engine_order", "per_candidate") == "per_candidate"
    if getattr(args, "engine_order", "per_candidate") not in ("shared", "per_candidate"):
        raise ValueError("args.engine_order must be 'shared' or 'per_candidate'")
    hp.order_per_candidate = per_cand
    if getattr(args, "multitask", False) and _hp is None:
        raise TypeError("max() received an invalid combination of arguments: the searchable returns a tuple "
                        "with --multitask in search mode (reference behaviour, train_searchable/ntu.py:54)")
    confs = [np.asarray(c).reshape(-1, 3) for c in sampled_configurations]
    K = len(confs)
    wanted = [i for i in range(K) if (not return_model or i in return_model)]
    N_tr, N_dev, B, E = len(train_l.table), len(dev_l.table), int(args.batchsize), int(args.epochs)
    nb = -(-N_tr // B)
    num_batches_per_epoch = N_tr / B              # float, ntu_searchable.py:30

    seed_base = popmod.broadcast_seed(int(torch.randint(0, 2 ** 31 - 1, (1,)).item()), device, confs)
    if getattr(args, "weightsharing", False):
        if hp.loss_mode != 0:
            raise NotImplementedError("weight sharing is wired for the single-label (NTU) searchable only")
        return _train_weightsharing(confs, wanted, searchable_type, train_l, dev_l, args, device, hp, seed_base,
                                    return_model, premodels, state_dict)

    rank, world = popmod.dist_info()
    wconfs = [confs[i] for i in wanted]
    owner, cap, _ = popmod.shard_call(wconfs, hp, world, device, bool(getattr(args, "engine_all_ranks", False)))
    costs = [popmod.candidate_cost(c, hp.R, hp.s_sizes, hp.v_sizes, hp.C) for c in wconfs]

    models = {}
    sched = LRCosineAnnealingScheduler(args.eta_max, args.eta_min, args.Ti, args.Tm, num_batches_per_epoch)
    etas = sched.eta_table(E * nb)
    shared_order = {}
    fail_rank = _TEST_FAIL_RANK      # test hook (module attribute, set by tests only): this rank's FIRST share raises (population.train_sharded re-queues it)
    attempts = [0]

    def train_share(mine):
        """This rank's share (or a re-queued part of a failed rank's): {candidate index: best dev metric}."""
        attempts[0] += 1
        if world > 1 and rank == fail_rank and attempts[0] == 1:
            raise RuntimeError("_TEST_FAIL_RANK: simulated failure of this rank's share")
        acc_by_idx = {}
        if mine and not per_cand and "o" not in shared_order:
            shared_order["o"] = make_order(N_tr, E, train_l.shuffle, seed_base + 1, device)
        order = shared_order.get("o")
        for group, pop in _plan_rounds(hp, confs, mine, device, seed_base, int(getattr(args, "engine_chunk_cols", 0))):
            try:
                if _pos_weight is not None:
                    pop.set_pos_weight(_pos_weight)
                mods = {}
                if premodels:
                    for j, i in enumerate(group):
                        src = premodels[i].module if getattr(args, "use_dataparallel", False) else premodels[i]
                        m = searchable_type(args, confs[i])
                        m.load_state_dict(src.state_dict())
                        pop.set_params(j, m.flat_params())
                        mods[i] = m
                elif getattr(args, "engine_init", "torch") == "device":
                    pop.init([(seed_base + 2 + i) & 0x7FFFFFFF for i in group])
                elif (return_model or not getattr(searchable_type, "_construction_is_standard", False)
                      or not _init_population_device_streams(pop, args, confs, group, hp, seed_base, device, searchable_type)):
                    _init_population_from_torch(pop, args, confs, group, hp, seed_base, searchable_type, return_model, mods, device)
                if getattr(args, "verbose", False):
                    print("Now training: ")
                    for i in group:
                        print(confs[i])
                if getattr(args, "engine_profile", False):
                    pop.set_profiling(True)
                if per_cand:
                    order = make_order_per_candidate(N_tr, E, train_l.shuffle, seed_base + 1, device, group)
                stats, status = pop.train(train_l.table, dev_l.table, E, etas, order=order,
                                          snapshot_best=bool(return_model))
                if getattr(args, "engine_profile", False):
                    PROFILE.append(pop.sweep_profile() + (pop.schedule(),))
                for j, i in enumerate(group):
                    if getattr(args, "verbose", False):
                        for e in range(E):
                            print("train Loss: {:.4f} Acc: {:.4f}".format(stats["train_loss_sum"][j, e] / N_tr,
                                                                          stats["train_corrects"][j, e] / N_tr))
                            print("dev Loss: {:.4f} Acc: {:.4f}".format(stats["dev_loss_sum"][j, e] / N_dev,
                                                                        stats["dev_corrects"][j, e] / N_dev))
                    acc_by_idx[i] = (best_dev_f1(stats[j], bool(status[j]), N_dev) if hp.loss_mode == 1
                                     else best_dev_accuracy(stats[j], N_dev))
                    if return_model:
                        m = mods.get(i) or searchable_type(args, confs[i])
                        m.load_flat(pop.get_params(j))
                        _bump_bn_counters(m, E * nb)
                        m.to(device)
                        m.train(False)
                        models[i] = m
            finally:
                pop.close()
        return acc_by_idx

    # ONE all_gather of (index, accuracy) pairs (RCCL on the GPU box); a rank whose share fails is re-queued, never waited for
    accs_all = popmod.train_sharded(wanted, owner, cap, K, costs, train_share, device)
    real_accuracies = [accs_all[i] for i in wanted]
    if return_model:
        # models live on the rank that trained them; other ranks get None placeholders
        return real_accuracies, [models.get(i) for i in wanted]
    return real_accuracies


def _bump_bn_counters(model, steps):
    for m in model.modules():
        if isinstance(m, nn.BatchNorm1d) and m.num_batches_tracked is not None:
            m.num_batches_tracked += steps


def _train_weightsharing(confs, wanted, searchable_type, train_l, dev_l, args, device, hp, seed_base,
                         return_model, premodels, state_dict):
    """--weightsharing makes candidates sequentially dependent (each starts from the cells the previous ones
    published, ntu_searchable.py:74-75,91-92): "replicas only" — every rank trains the whole list serially."""
    N_tr, N_dev, B, E = len(train_l.table), len(dev_l.table), int(args.batchsize), int(args.epochs)
    nb = -(-N_tr // B)
    accs, models = [], []
    for i in wanted:
        with torch.random.fork_rng(devices=[]):
            torch.manual_seed(seed_base + 2 + i)
            m = searchable_type(args, confs[i])
        if premodels:
            m.load_state_dict(premodels[i].state_dict())
        set_central_states(m, state_dict, False)
        pop = Population(hp, [confs[i]], device, drop_seeds=[(seed_base * 7 + i) & 0xFFFFFFFF])
        pop.set_params(0, m.flat_params())
        sched = LRCosineAnnealingScheduler(args.eta_max, args.eta_min, args.Ti, args.Tm, N_tr / B)
        order = make_order(N_tr, E, train_l.shuffle, seed_base + 1 + 1000 * i, device)
        if getattr(args, "verbose", False):
            print("Now training: ")
            print(confs[i])
        stats, _ = pop.train(train_l.table, dev_l.table, E, sched.eta_table(E * nb), order=order,
                             snapshot_best=True)
        if getattr(args, "verbose", False):
            for e in range(E):
                print("train Loss: {:.4f} Acc: {:.4f}".format(stats["train_loss_sum"][0, e] / N_tr,
                                                              stats["train_corrects"][0, e] / N_tr))
                print("dev Loss: {:.4f} Acc: {:.4f}".format(stats["dev_loss_sum"][0, e] / N_dev,
                                                            stats["dev_corrects"][0, e] / N_dev))
        m.load_flat(pop.get_params(0))
        _bump_bn_counters(m, E * nb)
        pop.close()
        get_central_states(m, state_dict, False)
        accs.append(best_dev_accuracy(stats[0], N_dev))
        if return_model:
            m.to(device)
            m.train(False)
            models.append(m)
    return (accs, models) if return_model else accs

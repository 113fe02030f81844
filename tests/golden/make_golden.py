#!/usr/bin/env python3
"""Generate golden vectors by importing the UNCHANGED reference (/root/reference) in this container.

Run (build container only; /root/reference does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Outputs small ``.npz`` fixtures next to this file.  Only inputs / expected outputs are stored (data),
never reference source.  Stubs (oracle-only):
  * ``inflated_resnet.load_pretrained_2D_weights`` -> no-op (it would download ResNet-50 weights),
  * ``models.central.ntu.Visual/Skeleton`` -> parameter-less modules that serve precomputed pooled taps
    through the reference's unchanged ``forward`` slicing,
  * empty checkpoint files for ``ske_cp`` / ``rgb_cp``.
Everything on the path (``train_sampled_models``, ``train_ntu_track_acc``,
``Searchable_Skeleton_Image_Net``, ``LRCosineAnnealingScheduler``, ``tools.*``) runs byte-for-byte unchanged.
"""
import contextlib
import io
import os
import re
import sys
import tempfile
from types import SimpleNamespace

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")
sys.path.insert(1, REPO)

import numpy as np
import torch
import torch.nn as nn

import models.auxiliary.inflated_resnet as ir

ir.load_pretrained_2D_weights = lambda *a, **k: None
import models.central.ntu as cntu


class FeatVisual(nn.Module):
    def __init__(self, args):
        super().__init__()

    def forward(self, x):
        return (None, x["v0"], x["v1"], x["v2"], x["v3"], x["vlogit"])


class FeatSkel(nn.Module):
    def __init__(self, args):
        super().__init__()

    def forward(self, x):
        return [x["s0"], x["s1"], x["s2"], x["s3"]], x["slogit"]


cntu.Visual, cntu.Skeleton = FeatVisual, FeatSkel
import models.search.ntu_searchable as ntu
import models.search.train_searchable.ntu as tr
import models.auxiliary.scheduler as sc
import models.search.tools as tools

from oracle import np_oracle as O

torch.set_num_threads(4)
TMP = tempfile.mkdtemp()
torch.save({}, os.path.join(TMP, "ske"))
torch.save({}, os.path.join(TMP, "rgb"))


class D(dict):
    def to(self, dev):
        return self

    def size(self, i):
        return next(iter(self.values())).size(i)


class ListLoader:
    """Unshuffled loader over a feature table: yields {'rgb','ske','label'} like datasets/ntu.py:254."""

    def __init__(self, table, B):
        self.t = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in table.items()}
        self.B = B
        self.dataset = range(len(table["label"]))

    def __iter__(self):
        N = len(self.dataset)
        for i in range(0, N, self.B):
            sl = slice(i, min(i + self.B, N))
            rgb = D({k: self.t[k][sl] for k in ("v0", "v1", "v2", "v3", "vlogit")})
            ske = D({k: self.t[k][sl] for k in ("s0", "s1", "s2", "s3", "slogit")})
            yield {"rgb": rgb, "ske": ske, "label": self.t["label"][sl]}


def mkargs(**kw):
    a = dict(vid_len=(8, 32), num_outputs=60, drpt=0.0, inner_representation_size=16, batchnorm=True,
             alphas=False, multitask=False, weightsharing=False, batchsize=16, eta_max=1e-3, eta_min=1e-6,
             Ti=1, Tm=2, use_dataparallel=False, verbose=False, epochs=2, checkpointdir=TMP, ske_cp="ske",
             rgb_cp="rgb")
    a.update(kw)
    return SimpleNamespace(**a)


def table(N, seed, with_logits=True, snr=0.3):
    return O.synth_table(N, seed, snr=snr, with_logits=with_logits) if with_logits else \
        dict(O.synth_table(N, seed, snr=snr), vlogit=np.zeros((N, 60), np.float32),
             slogit=np.zeros((N, 60), np.float32))


def hyper_of(args):
    return O.Hyper(R=args.inner_representation_size, C=args.num_outputs, B=args.batchsize,
                   bn=args.batchnorm, drpt=args.drpt, alphas=args.alphas, multitask=args.multitask,
                   eta_max=args.eta_max, eta_min=args.eta_min, Ti=args.Ti, Tm=args.Tm, epochs=args.epochs)


def load_det(model, conf, args, seed, perturb_bn=False):
    """Overwrite the reference module's central params with the hash-generated ones."""
    p = O.init_params(conf, hyper_of(args), seed, perturb_bn=perturb_bn)
    sd = model.state_dict()
    for k, v in p.items():
        assert sd[k].shape == v.shape, (k, sd[k].shape, v.shape)
        sd[k].copy_(torch.from_numpy(v))
    return p


def central_sd(model):
    return {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()
            if k.startswith(("alphas", "fusion_layers", "central_classifier"))}


def feats_of(t, sl):
    rgb = D({k: torch.from_numpy(t[k][sl]) for k in ("v0", "v1", "v2", "v3", "vlogit")})
    ske = D({k: torch.from_numpy(t[k][sl]) for k in ("s0", "s1", "s2", "s3", "slogit")})
    return rgb, ske


def put(out, key, arr):
    """Small tensors in full; big ones as a strided sample + float64 sum."""
    arr = np.asarray(arr)
    if arr.size <= 4096:
        out[key] = arr
    else:
        out[key + "#s"] = O.sample_view(arr)
        out[key + "#sum"] = np.array(arr.astype(np.float64).sum())


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print("wrote", name, os.path.getsize(path) // 1024, "KiB")


# ------------------------------------------------------------------ G1 scheduler
def g1():
    out = {}
    for j, (Ti, Tm, nbpe, n) in enumerate([(1, 2, 4.0, 40), (1, 2, 500.0, 1600), (5, 2, 625.0, 2000),
                                           (1, 2, 6.25, 64)]):
        s = sc.LRCosineAnnealingScheduler(1e-3, 1e-6, Ti, Tm, nbpe)
        seq = []
        for _ in range(n):
            s.step()
            seq.append(s.eta)
        out[f"cfg{j}"] = np.array([Ti, Tm, nbpe, n], np.float64)
        out[f"eta{j}"] = np.array(seq, np.float64)
    save("g1_scheduler.npz", **out)


# ------------------------------------------------------------------ G2/G3/G9 forward, loss, grads
CONFS = {
    "c4": [[3, 1, 1], [1, 3, 0], [1, 1, 1], [3, 3, 0]],
    "c0": [[2, 2, 0], [1, 0, 1], [3, 2, 0], [3, 1, 1]],
    "l1": [[0, 0, 0]],
    "l2": [[2, 3, 1], [0, 2, 2]],
    "l3": [[1, 0, 2], [3, 3, 1], [0, 1, 0]],
}


VARIANTS = [
    ("bn_train", dict(batchnorm=True, drpt=0.0), True),
    ("bn_eval", dict(batchnorm=True, drpt=0.0), False),
    ("bndrop_eval", dict(batchnorm=True, drpt=0.5), False),
    ("drop_eval", dict(batchnorm=False, drpt=0.5), False),
    ("alpha_bn_train", dict(batchnorm=True, drpt=0.0, alphas=True), True),
    ("mt_bn_train", dict(batchnorm=True, drpt=0.0, multitask=True), True),
]


def g23():
    """table = synth_table(16, 11, snr=0.3, with_logits=True); params = init_params(conf, hp, seed, perturb_bn=True)."""
    out = {}
    t = table(16, 11)
    names = []
    for ci, (cname, conf) in enumerate(CONFS.items()):
        for vi, (vname, kw, train) in enumerate(VARIANTS):
            for R in (16, 128):
                if R == 128 and cname not in ("c4", "l2"):
                    continue
                seed = 1000 + 100 * ci + 10 * vi + (R == 128)
                args = mkargs(inner_representation_size=R, **kw)
                model = ntu.Searchable_Skeleton_Image_Net(args, np.array(conf))
                load_det(model, conf, args, seed, perturb_bn=True)
                model.train(train)
                pre = f"{cname}/{vname}/{R}/"
                names.append(f"{cname}/{vname}/{R}/{seed}")
                rgb, ske = feats_of(t, slice(0, 16))
                label = torch.from_numpy(t["label"][:16])
                output = model((rgb, ske))
                crit = torch.nn.CrossEntropyLoss()
                if args.multitask:
                    preds = torch.max(sum(output), 1)[1]
                    loss = crit(output[0], label) + crit(output[1], label) + crit(output[2], label)
                    out[pre + "loss_central"] = np.array(crit(output[0], label).item())
                    logits = output[0]
                else:
                    preds = torch.max(output, 1)[1]
                    loss = crit(output, label)
                    logits = output
                out[pre + "logits"] = logits.detach().numpy()
                out[pre + "loss"] = np.array(loss.item())
                out[pre + "preds"] = preds.numpy()
                if train:
                    loss.backward()
                    for n_, p in model.named_parameters():
                        if p.grad is not None:
                            put(out, pre + "grad/" + n_, p.grad.numpy().copy())
                    for k, v in central_sd(model).items():
                        if "running" in k:
                            out[pre + "after/" + k] = v
    out["names"] = np.array(names)
    save("g23_forward_backward.npz", **out)


# ------------------------------------------------------------------ G4/G5/G6 deterministic trajectory
class Capture:
    """searchable_type factory (it is a plain callable, ntu_searchable.py:44): constructs the reference
    module, overwrites the central params with the hash-generated ones (seed = base + index)."""

    def __init__(self, seed0, perturb_bn=False):
        self.seed0 = seed0
        self.models = []
        self.perturb_bn = perturb_bn

    def __call__(self, args, conf):
        m = ntu.Searchable_Skeleton_Image_Net(args, conf)
        load_det(m, conf, args, self.seed0 + len(self.models), perturb_bn=self.perturb_bn)
        self.models.append(m)
        return m


HIST_RE = r"(train|dev) Loss: ([0-9.eE+naninf-]+) Acc: ([0-9.eE+naninf-]+)"


def parse_hist(text):
    return np.array([(0 if m.group(1) == "train" else 1, float(m.group(2)), float(m.group(3)))
                     for m in re.finditer(HIST_RE, text)], np.float64)


def run_tsm(confs, args, loaders, seed0):
    cap = Capture(seed0)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        accs = ntu.train_sampled_models([np.array(c) for c in confs], cap, loaders, args, "cpu")
    return [float(a) for a in accs], cap, parse_hist(buf.getvalue())


def g456():
    """Deterministic mode (drpt=0 + BN, unshuffled).  train = synth_table(64, 21, snr=0.3),
    dev = synth_table(48, 22, snr=0.3), params = init_params(conf, hp, 5)."""
    out = {}
    ttr, tdv = table(64, 21, with_logits=False), table(48, 22, with_logits=False)
    for cname, R in (("c4", 16), ("c4", 128), ("l2", 16)):
        conf = np.array(CONFS[cname])
        args = mkargs(inner_representation_size=R, batchnorm=True, drpt=0.0, epochs=3, batchsize=16)
        model = ntu.Searchable_Skeleton_Image_Net(args, conf)
        load_det(model, conf, args, 5)
        pre = f"{cname}/{R}/"
        opt = torch.optim.Adam(model.central_params(), lr=args.eta_max, weight_decay=1e-4)
        sched = sc.LRCosineAnnealingScheduler(args.eta_max, args.eta_min, args.Ti, args.Tm, 64 / 16)
        crit = torch.nn.CrossEntropyLoss()
        # replica of the train-phase body of train_ntu_track_acc (:46-69) to dump state after 1,2,10 steps
        model.train(True)
        step = 0
        losses = []
        for ep in range(3):
            for data in ListLoader(ttr, 16):
                opt.zero_grad()
                output = model((data["rgb"], data["ske"]))
                loss = crit(output, data["label"])
                sched.step()
                sched.update_optimizer(opt)
                loss.backward()
                opt.step()
                losses.append(loss.item())
                step += 1
                if step in (1, 2, 10):
                    for k, v in central_sd(model).items():
                        if "num_batches" not in k:
                            put(out, pre + f"step{step}/p/" + k, v)
                    name_of = {id(p): n for n, p in model.named_parameters()}
                    for p, st in opt.state.items():
                        put(out, pre + f"step{step}/m/" + name_of[id(p)], st["exp_avg"].numpy().copy())
                        put(out, pre + f"step{step}/v/" + name_of[id(p)], st["exp_avg_sq"].numpy().copy())
        out[pre + "losses"] = np.array(losses)
        # the unchanged train_sampled_models -> train_ntu_track_acc from the same init
        accs, cap, hist = run_tsm([conf], args, {"train": ListLoader(ttr, 16), "dev": ListLoader(tdv, 16)}, 5)
        out[pre + "best_acc"] = np.array(accs[0])
        out[pre + "hist"] = hist
        for k, v in central_sd(cap.models[0]).items():   # best-epoch weights restored (:86)
            if "num_batches" not in k:
                put(out, pre + "final/" + k, v)
    save("g456_trajectory.npz", **out)


# ------------------------------------------------------------------ G7 train_sampled_models on 4 confs
def g7():
    """train = synth_table(256, 31, snr=0.5), dev = synth_table(128, 32, snr=0.5); params seed 9+i."""
    out = {}
    ttr, tdv = table(256, 31, with_logits=False, snr=0.5), table(128, 32, with_logits=False, snr=0.5)
    confs = [CONFS["l1"], CONFS["l2"], CONFS["l3"], CONFS["c4"]]
    for B in (16, 20):      # 20: ragged last batch (256 = 12*20+16, 128 = 6*20+8)
        args = mkargs(inner_representation_size=16, batchnorm=True, drpt=0.0, epochs=3, batchsize=B)
        accs, cap, hist = run_tsm(confs, args, {"train": ListLoader(ttr, B), "dev": ListLoader(tdv, B)}, 9)
        out[f"B{B}/accs"] = np.array(accs)
        out[f"B{B}/hist"] = hist
    for i in range(4):
        out[f"conf{i}"] = np.array(confs[i])
    # multitask variant through train_ntu_track_acc directly (found-script style, main_found_ntu.py:121)
    ttr2, tdv2 = table(256, 31, with_logits=True, snr=0.5), table(128, 32, with_logits=True, snr=0.5)
    args = mkargs(inner_representation_size=16, batchnorm=True, drpt=0.0, epochs=3, batchsize=16, multitask=True)
    conf = np.array(CONFS["c0"])
    model = ntu.Searchable_Skeleton_Image_Net(args, conf)
    load_det(model, conf, args, 13)
    opt = torch.optim.Adam(model.central_params(), lr=args.eta_max, weight_decay=1e-4)
    sched = sc.LRCosineAnnealingScheduler(args.eta_max, args.eta_min, args.Ti, args.Tm, 256 / 16)
    crits = [torch.nn.CrossEntropyLoss()] * 3
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        acc = tr.train_ntu_track_acc(model, crits, opt, sched,
                                     {"train": ListLoader(ttr2, 16), "dev": ListLoader(tdv2, 16)},
                                     {"train": 256, "dev": 128}, device="cpu", num_epochs=3, multitask=True)
    out["mt_acc"] = np.array(float(acc))
    out["mt_hist"] = parse_hist(buf.getvalue())
    # alphas variant
    args = mkargs(inner_representation_size=16, batchnorm=True, drpt=0.0, epochs=3, batchsize=16, alphas=True)
    accs, cap, hist = run_tsm([CONFS["l3"]], args, {"train": ListLoader(ttr, 16), "dev": ListLoader(tdv, 16)}, 21)
    out["alpha_acc"] = np.array(accs[0])
    out["alpha_hist"] = hist
    out["alpha_final"] = np.array([cap.models[0].state_dict()[f"alphas.{i}.alpha_x"].item() for i in range(3)])
    save("g7_population.npz", **out)


# ------------------------------------------------------------------ G8 controller-side pins
def g8():
    out = {}
    out["layer_confs"] = np.array(ntu.get_possible_layer_configurations(0))
    a = SimpleNamespace(initial_temperature=10.0, final_temperature=0.2, temperature_decay=4.0)
    out["temperature"] = np.array([tools.compute_temperature(i, a) for i in range(12)])
    merged0 = tools.merge_unfolded_with_sampled([], ntu.get_possible_layer_configurations(0), 0)
    out["merged0"] = np.array(merged0)
    np.random.seed(0)
    accs = np.linspace(0.1, 0.9, len(merged0))
    samp = tools.sample_k_configurations(merged0, accs, 5, 10.0)
    out["sampled0"] = np.array(samp)
    merged1 = tools.merge_unfolded_with_sampled(samp, ntu.get_possible_layer_configurations(1), 1)
    out["merged1"] = np.array(merged1)
    np.random.seed(1)
    samp1 = tools.sample_k_configurations(merged1, np.linspace(0.2, 0.8, len(merged1)), 5, 2.5)
    out["sampled1"] = np.array(samp1)
    merged1b = tools.merge_unfolded_with_sampled(samp1, ntu.get_possible_layer_configurations(0), 0)
    out["merged1b"] = np.array(merged1b)
    save("g8_controller.npz", **out)


# ------------------------------------------------------------------ G10 stochastic e2e statistics
def g10():
    """Dropout on, shuffled order: best dev acc over seeds (the reference's own seed noise).
    set A (search defaults): R=16, no BN, drpt 0.5, snr 1.0;  set B (config-2 like): R=128, BN, drpt 0.5, snr 0.15.
    train = synth_table(2048, 1, snr), dev = synth_table(2048, 2, snr)."""
    out = {}
    N, Nd = 2048, 2048

    class ShuffleLoader(ListLoader):
        def __iter__(self):
            N = len(self.dataset)
            perm = torch.randperm(N)
            for i in range(0, N, self.B):
                sl = perm[i:i + self.B]
                rgb = D({k: self.t[k][sl] for k in ("v0", "v1", "v2", "v3", "vlogit")})
                ske = D({k: self.t[k][sl] for k in ("s0", "s1", "s2", "s3", "slogit")})
                yield {"rgb": rgb, "ske": ske, "label": self.t["label"][sl]}

    for tag, snr, R, bn, confs, nseed in (("A", 1.0, 16, False, [CONFS["c4"], CONFS["l1"], CONFS["l2"]], 32),
                                          ("B", 0.15, 128, True, [CONFS["c4"]], 64)):
        ttr, tdv = table(N, 1, with_logits=False, snr=snr), table(Nd, 2, with_logits=False, snr=snr)
        out[tag + "/meta"] = np.array([N, Nd, snr, R, 16, 3, int(bn), 0.5])  # N,Ndev,snr,R,B,epochs,bn,drpt
        args = mkargs(inner_representation_size=R, batchnorm=bn, drpt=0.5, epochs=3, batchsize=16)
        loaders = {"train": ShuffleLoader(ttr, 16), "dev": ListLoader(tdv, 16)}
        allacc = []
        for seed in range(nseed):
            torch.manual_seed(100 + seed)                  # dropout + shuffle streams of the reference
            accs, _, _ = run_tsm(confs, args, loaders, 1000 + 10 * seed)
            allacc.append(accs)
            print("g10", tag, "seed", seed, accs, flush=True)
        out[tag + "/accs"] = np.array(allacc)
        for i, c in enumerate(confs):
            out[tag + f"/conf{i}"] = np.array(c)
    save("g10_stochastic.npz", **out)


# ------------------------------------------------------------------ G9 controller run (next#1)
def g9():
    """The reference's ModelSearcher._epnas / _randsearch driven by a fake trainer (acc = np_oracle.fake_accuracy).
    models.searchable needs import stubs for torchvision / cv2 and the alias models.aux -> models.auxiliary (D6)."""
    import random
    import types
    for name in ("torchvision", "torchvision.transforms", "torchvision.datasets", "cv2"):
        sys.modules.setdefault(name, types.ModuleType(name))
    import models.auxiliary as aux_pkg
    sys.modules.setdefault("models.aux", aux_pkg)
    sys.modules.setdefault("models.aux.scheduler", sc)
    import models.searchable as S
    import models.search.surrogate as rsurr
    out = {}
    for tag, iters, levels, K in (("a", 2, 3, 5), ("b", 3, 4, 6)):
        args = SimpleNamespace(search_iterations=iters, max_progression_levels=levels, num_samples=K,
                               initial_temperature=10.0, final_temperature=0.2, temperature_decay=4.0,
                               lr_surrogate=0.001, epochs_surrogate=8, verbose=False)
        calls = []

        def fake_train(confs, model_type, dataloaders, a, device, state_dict=None):
            calls.append([np.array(c) for c in confs])
            return [O.fake_accuracy(c) for c in confs]

        np.random.seed(3)
        torch.manual_seed(3)
        random.seed(3)
        surrogate = rsurr.SimpleRecurrentSurrogate(100, 3, 100)
        searcher = S.ModelSearcher(args)
        s_data = searcher._epnas(None, {"model": surrogate, "criterion": torch.nn.MSELoss()}, None,
                                 {"train_sampled_fun": fake_train,
                                  "get_layer_confs": ntu.get_possible_layer_configurations}, "cpu")
        out[tag + "/call_sizes"] = np.array([len(c) for c in calls])
        flat = [np.concatenate([c.reshape(-1), [-1]]) for call in calls for c in call]
        out[tag + "/calls_flat"] = np.concatenate(flat)
        confs, accs, _ = s_data.get_k_best(5)
        out[tag + "/best_accs"] = np.sort(np.array(accs))
        out[tag + "/final_pred"] = np.array([surrogate.eval_model(np.array(CONFS["c4"]), "cpu"),
                                             surrogate.eval_model(np.array(CONFS["l2"]), "cpu")])
    # random search
    args = SimpleNamespace(search_iterations=2, max_progression_levels=3, num_samples=4, verbose=False)
    calls = []

    def fake_train2(confs, model_type, dataloaders, a, device, state_dict=None):
        calls.append([np.array(c) for c in confs])
        return [O.fake_accuracy(c) for c in confs]

    np.random.seed(5)
    random.seed(5)
    S.ModelSearcher(args)._randsearch(None, None, {"train_sampled_fun": fake_train2,
                                                   "get_layer_confs": ntu.get_possible_layer_configurations}, "cpu")
    out["r/calls_flat"] = np.concatenate([np.concatenate([c.reshape(-1), [-1]]) for call in calls for c in call])
    save("g9_controller_run.npz", **out)


# ------------------------------------------------------------------ G11 MM-IMDB loss / metric / loop (next#3)
def g11():
    """Pins for the multi-label variant.  The reference has no Searchable_* for MM-IMDB, so the network
    is the NTU searchable re-sized to MM-IMDB taps (text [64,128], image 4x512, C=23) — test glue; what IS the
    reference's and is pinned here: WeightedCrossEntropyWithLogits (models/central/mm_imdb.py:655-673) and
    train_mmimdb_track_f1 (models/search/train_searchable/mmimdb.py:15-137, needs the alias
    models.train.scheduler -> models.auxiliary.scheduler and a stub torchvision.models)."""
    import types
    tv = sys.modules.setdefault("torchvision", types.ModuleType("torchvision"))
    tvm = sys.modules.setdefault("torchvision.models", types.ModuleType("torchvision.models"))
    tv.models = tvm
    import models.auxiliary as aux_pkg
    sys.modules.setdefault("models.train", aux_pkg)
    sys.modules.setdefault("models.train.scheduler", sc)
    import models.central.mm_imdb as mm
    import models.search.train_searchable.mmimdb as trm
    import models.auxiliary.aux_models as aux
    out = {}
    C = 23
    w = O.mm_pos_weight(C)
    # (a) the loss and its gradient
    rng_logits = (O.hash_noise(77, 16 * C).reshape(16, C) * np.float32(2.0)).astype(np.float32)
    z = (O.hash_u01(78, 16 * C).reshape(16, C) < 0.2).astype(np.float32)
    lg = torch.from_numpy(rng_logits.copy()).requires_grad_(True)
    loss = mm.WeightedCrossEntropyWithLogits(w.tolist())(lg, torch.from_numpy(z))
    loss.backward()
    out["loss"] = np.array(loss.item())
    out["dlogits"] = lg.grad.numpy().copy()

    # (b) the train loop on MM-IMDB-shaped tables
    class TextImageNet(ntu.Searchable_Skeleton_Image_Net):
        def _create_alphas(self):
            return nn.ModuleList([aux.AlphaScalarMultiplication(O.MM_S_SIZES[c[0]], O.MM_V_SIZES[c[1]])
                                  for c in self.conf])

    class Wrap(nn.Module):
        def __init__(self, inner):
            super().__init__()
            self.inner = inner

        def forward(self, text, image):
            return self.inner((image, text))

    ttr, tdv = O.synth_table_mm(128, 41), O.synth_table_mm(96, 42)

    class MMLoader:
        def __init__(self, t, B):
            self.t = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in t.items()}
            self.B = B
            self.dataset = range(len(t["multilabel"]))

        def __iter__(self):
            N = len(self.dataset)
            zero = torch.zeros(1)
            for i in range(0, N, self.B):
                sl = slice(i, min(i + self.B, N))
                img = D({k: self.t[k][sl] for k in ("v0", "v1", "v2", "v3")})
                img["vlogit"] = zero
                txt = D({k: self.t[k][sl] for k in ("s0", "s1", "s2", "s3")})
                txt["slogit"] = zero
                yield {"image": img, "text": txt, "label": self.t["multilabel"][sl]}

    for tag, conf, R in (("a", [[1, 2, 0], [0, 3, 1]], 16), ("b", [[0, 0, 1]], 32)):
        conf = np.array(conf)
        args = mkargs(inner_representation_size=R, batchnorm=True, drpt=0.0, epochs=3, batchsize=16, num_outputs=C)
        inner = TextImageNet(args, conf)
        hp = O.Hyper(R=R, C=C, B=16, bn=True, drpt=0.0, epochs=3, s_sizes=O.MM_S_SIZES, v_sizes=O.MM_V_SIZES,
                     loss_mode=1)
        p = O.init_params(conf, hp, 17)
        sd = inner.state_dict()
        for k, v in p.items():
            assert sd[k].shape == v.shape, (k, sd[k].shape, v.shape)
            sd[k].copy_(torch.from_numpy(v))
        model = Wrap(inner)
        opt = torch.optim.Adam(inner.central_params(), lr=args.eta_max, weight_decay=1e-4)
        sched = sc.LRCosineAnnealingScheduler(args.eta_max, args.eta_min, args.Ti, args.Tm, 128 / 16)
        crit = mm.WeightedCrossEntropyWithLogits(w.tolist())
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            best = trm.train_mmimdb_track_f1(model, crit, opt, sched, {"train": MMLoader(ttr, 16), "dev": MMLoader(tdv, 16)},
                                             {"train": 128, "dev": 96}, device="cpu", num_epochs=3, verbose=True)
        f1s = [float(m.group(1)) for m in re.finditer(r"dev F1: ([0-9.]+)", buf.getvalue())]
        out[tag + "/best_f1"] = np.array(float(best))
        out[tag + "/f1_per_epoch"] = np.array(f1s)
        out[tag + "/conf"] = conf
    save("g11_mmimdb.npz", **out)


# ------------------------------------------------------------------ G12 AV-MNIST searchable (next#4)
def g12():
    """The reference's AV-MNIST variant (models/search/avmnist_searchable.py:23-108,184-297 +
    train_searchable/avmnist.py) through stub backbones serving 5 audio + 3 image taps of widths c..16c / c..4c
    (channels = 3 -> widths that are NOT multiples of 16) — incl. the plain [Linear, nl] cell (drpt = 0)."""
    import types
    import models.auxiliary as aux_pkg
    sys.modules.setdefault("models.aux", aux_pkg)
    sys.modules.setdefault("models.aux.scheduler", sc)
    import models.central.avmnist as cav

    class FeatImg(nn.Module):
        def __init__(self, args, ch):
            super().__init__()

        def forward(self, x):
            return (x["vlogit"], x["v0"], x["v1"], x["v2"])

    class FeatAud(nn.Module):
        def __init__(self, args, ch):
            super().__init__()

        def forward(self, x):
            return (x["slogit"], x["s0"], x["s1"], x["s2"], x["s3"], x["s4"])

    cav.GP_LeNet, cav.GP_LeNet_Deeper = FeatImg, FeatAud
    import models.search.avmnist_searchable as avm
    out = {}
    SS, VS = (3, 6, 12, 24, 48), (3, 6, 12)
    out["layer_confs"] = np.array(avm.get_possible_layer_configurations(0))

    class AVLoader:
        def __init__(self, t, B):
            self.t = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in t.items()}
            self.B = B
            self.dataset = range(len(t["label"]))

        def __iter__(self):
            N = len(self.dataset)
            z = torch.zeros(1)
            for i in range(0, N, self.B):
                sl = slice(i, min(i + self.B, N))
                img = D({k: self.t[k][sl] for k in ("v0", "v1", "v2")})
                img["vlogit"] = z
                aud = D({k: self.t[k][sl] for k in ("s0", "s1", "s2", "s3", "s4")})
                aud["slogit"] = z
                yield {"image": img, "audio": aud, "label": self.t["label"][sl]}

    ttr = O.synth_table(192, 71, snr=1.0, C=10, s_sizes=SS, v_sizes=VS)
    tdv = O.synth_table(96, 72, snr=1.0, C=10, s_sizes=SS, v_sizes=VS)
    confs = [[[4, 2, 0]], [[3, 1, 1], [4, 2, 2]], [[0, 0, 0], [2, 1, 1], [4, 2, 0]]]
    for tag, drpt in (("plain", 0.0),):
        args = mkargs(inner_representation_size=16, batchnorm=False, drpt=drpt, epochs=3, batchsize=16, num_outputs=10,
                      channels=3, audio_cp="ske", rgb_cp="rgb")
        hp = O.Hyper(R=16, C=10, B=16, bn=False, drpt=drpt, epochs=3, s_sizes=SS, v_sizes=VS, allow_plain_cell=True)

        class Cap:
            def __init__(self):
                self.n = 0

            def __call__(self, a, conf):
                m = avm.Searchable_Audio_Image_Net(a, conf)
                p = O.init_params(conf, hp, 30 + self.n)
                sd = m.state_dict()
                for k, v in p.items():
                    if k in sd:
                        assert sd[k].shape == v.shape, (k, sd[k].shape, v.shape)
                        sd[k].copy_(torch.from_numpy(v))
                self.n += 1
                return m

        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            accs = avm.train_sampled_models([np.array(c) for c in confs], Cap(), {"train": AVLoader(ttr, 16), "dev": AVLoader(tdv, 16)},
                                            args, "cpu")
        out[tag + "/accs"] = np.array([float(a) for a in accs])
        out[tag + "/dev_acc_per_epoch"] = np.array([float(m.group(1)) for m in re.finditer(r"dev Acc: ([0-9.]+)", buf.getvalue())])
    for i, c in enumerate(confs):
        out[f"conf{i}"] = np.array(c)
    save("g12_avmnist.npz", **out)


# ------------------------------------------------------------------ G13 full-size deterministic trajectory
def g13():
    """BASELINE configs[1] at FULL size through the unchanged reference: conf 4, R=128, --batchnorm, B=16,
    N_train=10,000, N_dev=5,600, bf16-rounded taps (synth_table(N, seed, snr=0.15, quant='bf16')), deterministic mode
    (drpt=0, unshuffled), 3 epochs, params = init_params(conf, hp, 77).
    The run is repeated with 1, 2, 4 and 8 BLAS threads: the summation order inside ATen's GEMMs changes, Adam's
    sign-like early steps amplify the round-off, and the reference's trajectory differs from ITSELF (per-step loss by
    1e-4 after one step, 1e-2 after ten).  The spread of these runs is the reference's own reproducibility envelope."""
    out = {}
    ttr = dict(O.synth_table(10000, 1, snr=0.15, quant="bf16"), vlogit=np.zeros((10000, 60), np.float32), slogit=np.zeros((10000, 60), np.float32))
    tdv = dict(O.synth_table(5600, 2, snr=0.15, quant="bf16"), vlogit=np.zeros((5600, 60), np.float32), slogit=np.zeros((5600, 60), np.float32))
    conf = np.array(CONFS["c4"])
    args = mkargs(inner_representation_size=128, batchnorm=True, drpt=0.0, epochs=3, batchsize=16)
    hists, bests = [], []
    for nt in (1, 2, 4, 8):
        torch.set_num_threads(nt)
        accs, cap, hist = run_tsm([conf], args, {"train": ListLoader(ttr, 16), "dev": ListLoader(tdv, 16)}, 77)
        hists.append(hist)
        bests.append(accs[0])
        print("g13 threads", nt, accs[0], hist[:, 2], flush=True)
    out["threads"] = np.array([1, 2, 4, 8])
    out["best_acc"] = np.array(bests)
    out["hist"] = np.array(hists)          # [run][2*epoch + phase] = (phase, loss, acc)
    save("g13_fullsize.npz", **out)


# ------------------------------------------------------------------ G14 full-size reproducibility envelope
def _envelope(name, snr, NT):
    ttr = dict(O.synth_table(10000, 1, snr=snr, quant="bf16"), vlogit=np.zeros((10000, 60), np.float32), slogit=np.zeros((10000, 60), np.float32))
    tdv = dict(O.synth_table(5600, 2, snr=snr, quant="bf16"), vlogit=np.zeros((5600, 60), np.float32), slogit=np.zeros((5600, 60), np.float32))
    conf = np.array(CONFS["c4"])
    args = mkargs(inner_representation_size=128, batchnorm=True, drpt=0.0, epochs=3, batchsize=16)
    torch.set_num_threads(4)
    hists, bests = [], []
    for trial in range(NT):
        class CapP(Capture):
            def __call__(self, a, c):
                m = ntu.Searchable_Skeleton_Image_Net(a, c)
                pp = O.perturb_params(O.init_params(c, hyper_of(a), 77), trial)
                sd = m.state_dict()
                for k, v in pp.items():
                    sd[k].copy_(torch.from_numpy(v))
                self.models.append(m)
                return m
        cap = CapP(77)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            accs = ntu.train_sampled_models([conf], cap, {"train": ListLoader(ttr, 16), "dev": ListLoader(tdv, 16)}, args, "cpu")
        h = parse_hist(buf.getvalue())
        hists.append(h)
        bests.append(float(accs[0]))
        print(name, "trial", trial, bests[-1], h[:, 1:].ravel(), flush=True)
    # hist[trial][2*epoch + phase] = (phase, loss, acc)
    save(name, hist=np.array(hists), best_acc=np.array(bests), rel=np.array([1e-7]), snr=np.array([snr]))


def g14():
    """Same full-size deterministic recipe as G13 (3 epochs, 4 BLAS threads), started from NT=64 round-off-sized
    perturbations of the initial weight matrices (O.perturb_params(init_params(conf, hp, 77), trial), trial 0 = exact).
    The oracle and the engine are run from the same starts in tests/test_fullsize.py: the three ensembles must agree in
    distribution (means within the standard error), which is the strongest statement chaos allows at this size."""
    _envelope("g14_fullsize_envelope.npz", 0.15, int(os.environ.get("G14_NT", "64")))


def g14m():
    """G14 in the SENSITIVE regime (mid-range accuracy): same recipe at snr = G14M_SNR (default 0.10 = the committed
    fixture, asserted by tests/test_fullsize.py)."""
    _envelope("g14m_fullsize_midrange.npz", float(os.environ.get("G14M_SNR", "0.10")), int(os.environ.get("G14_NT", "64")))


# ------------------------------------------------------------------ G15 the bench workload itself (stochastic, E=10)
class ShuffleLoader(ListLoader):
    """DataLoader(shuffle=True) stand-in (models/searchable.py:248): a fresh torch.randperm per epoch."""

    def __iter__(self):
        N = len(self.dataset)
        perm = torch.randperm(N)
        for i in range(0, N, self.B):
            sl = perm[i:i + self.B]
            rgb = D({k: self.t[k][sl] for k in ("v0", "v1", "v2", "v3", "vlogit")})
            ske = D({k: self.t[k][sl] for k in ("s0", "s1", "s2", "s3", "slogit")})
            yield {"rgb": rgb, "ske": ske, "label": self.t["label"][sl]}


G15_DEFAULT_SNR = 0.12         # what bench.py runs (synth_table snr 0.12); the committed fixture's meta[2] must equal it
G15_DEFAULT_NS = 64


def g15():
    """BASELINE configs[1] exactly as bench.py runs it, through the unchanged reference: conf 4, R=128, --batchnorm,
    drpt 0.5, B=16, shuffled train order, E=10, N_train=10,000, N_dev=5,600, bf16-rounded taps
    (synth_table(N, seed, snr=G15_SNR, quant='bf16')), params = init_params(conf, hp, 3000 + 10*seed),
    torch.manual_seed(300 + seed) for the reference's OWN dropout (Philox) / shuffle streams; NS seeds.
    The statistic: best dev accuracy (and the per-epoch dev accuracies) over seeds.
    Every seed is independent, so the run can be split over processes: G15_SEED0=a G15_NS=n writes the part file
    g15_part_<a>.npz (git-ignored); `make_golden.py g15merge` concatenates the parts in seed order into the fixture.
    Committed fixture: `for a in 0 16 32 48; do G15_SEED0=$a G15_NS=16 G15_THREADS=1 python make_golden.py g15 & done; wait;
    python make_golden.py g15merge` (snr 0.12, 64 seeds, E=10; 1 BLAS thread per process)."""
    snr = float(os.environ.get("G15_SNR", G15_DEFAULT_SNR))
    NS = int(os.environ.get("G15_NS", G15_DEFAULT_NS))
    seed0 = int(os.environ.get("G15_SEED0", "0"))
    E = int(os.environ.get("G15_E", "10"))
    nthreads = int(os.environ.get("G15_THREADS", "1"))
    ttr = dict(O.synth_table(10000, 1, snr=snr, quant="bf16"), vlogit=np.zeros((10000, 60), np.float32), slogit=np.zeros((10000, 60), np.float32))
    tdv = dict(O.synth_table(5600, 2, snr=snr, quant="bf16"), vlogit=np.zeros((5600, 60), np.float32), slogit=np.zeros((5600, 60), np.float32))
    conf = np.array(CONFS["c4"])
    args = mkargs(inner_representation_size=128, batchnorm=True, drpt=0.5, epochs=E, batchsize=16)
    torch.set_num_threads(nthreads)
    loaders = {"train": ShuffleLoader(ttr, 16), "dev": ListLoader(tdv, 16)}
    bests, hists = [], []
    meta = np.array([10000, 5600, snr, 128, 16, E, 1, 0.5])   # N,Ndev,snr,R,B,epochs,bn,drpt
    cmd = np.array(f"G15_SNR={snr} G15_NS={NS} G15_SEED0={seed0} G15_E={E} G15_THREADS={nthreads} make_golden.py g15")
    part = os.path.join(HERE, f"g15_part_{seed0:03d}.npz")
    if "G15_SEED0" in os.environ and os.path.exists(part):       # resume an interrupted part (every seed is independent)
        old = np.load(part)
        if np.array_equal(old["meta"], meta) and int(old["seeds"][0]) == seed0:
            bests, hists = list(old["best_acc"]), list(old["hist"])
    for seed in range(seed0 + len(bests), seed0 + NS):
        torch.manual_seed(300 + seed)
        accs, _, hist = run_tsm([conf], args, loaders, 3000 + 10 * seed)
        bests.append(accs[0])
        hists.append(hist)
        print("g15 seed", seed, accs[0], hist[1::2, 2], flush=True)
        if "G15_SEED0" in os.environ and not os.environ.get("G15_PROBE"):
            np.savez_compressed(part, best_acc=np.array(bests), hist=np.array(hists), seeds=np.arange(seed0, seed0 + len(bests)), meta=meta, cmd=cmd)
    if os.environ.get("G15_PROBE"):
        return
    arrs = dict(best_acc=np.array(bests), hist=np.array(hists), seeds=np.arange(seed0, seed0 + NS), meta=meta, cmd=cmd)
    if "G15_SEED0" in os.environ:
        save(f"g15_part_{seed0:03d}.npz", **arrs)
    else:
        save("g15_bench_workload.npz", **arrs)


def g15merge():
    import glob
    parts = sorted((np.load(f) for f in glob.glob(os.path.join(HERE, "g15_part_*.npz"))), key=lambda p: int(p["seeds"][0]))
    if parts and int(parts[0]["seeds"][0]) != 0:      # extension run (round 6): the committed fixture holds seeds 0 … n-1, the parts go on from n
        parts.insert(0, np.load(os.path.join(HERE, "g15_bench_workload.npz")))
    assert parts and all(np.array_equal(p["meta"], parts[0]["meta"]) for p in parts)
    seeds = np.concatenate([p["seeds"] for p in parts])
    best = np.concatenate([p["best_acc"] for p in parts])
    hist = np.concatenate([p["hist"] for p in parts])
    # (an extension run merged while its last parts are still being computed: keep the contiguous prefix 0 .. n-1)
    n = 0
    while n < len(seeds) and seeds[n] == n:
        n += 1
    assert n >= 256, seeds
    cmds = [str(p["cmd"]) for p in parts]
    cmd = cmds[0] + f" ; G15_SEED0=256.. G15_NS=16 x {len(cmds) - 1} parts (tools/r06_extend_goldens.sh) ; g15merge -> {n} seeds"
    save("g15_bench_workload.npz", best_acc=best[:n], hist=hist[:n], seeds=seeds[:n], meta=parts[0]["meta"], cmd=np.array(cmd))


# ------------------------------------------------------------------ G16 weight sharing (next#4)
def g16():
    """--weightsharing through the unchanged train_sampled_models (ntu_searchable.py:74-75,91-92,123-174): 4 confs where
    conf 1 and conf 3 re-use cell "0.L_640_16.A_relu" of conf 0 and conf 2 re-uses nothing at index 0 (other activation)
    but shares cell 1's key with conf 3.  Deterministic mode (drpt=0 + BN, unshuffled), R=16, B=16, 3 epochs;
    train = synth_table(256, 41, snr=1.5), dev = synth_table(128, 42, snr=1.5); params of candidate i = init_params(seed 50+i)
    (then overwritten by the shared cells, as in the reference).  Stored: accuracies, per-epoch history, the published
    state_dict after every candidate, every candidate's final central parameters."""
    out = {}
    ttr, tdv = table(256, 41, with_logits=False, snr=1.5), table(128, 42, with_logits=False, snr=1.5)
    confs = [[[0, 0, 0]],
             [[0, 0, 0], [1, 1, 1]],
             [[0, 0, 1], [1, 1, 1]],
             [[0, 0, 0], [1, 1, 1], [2, 0, 0]]]
    args = mkargs(inner_representation_size=16, batchnorm=True, drpt=0.0, epochs=3, batchsize=16, weightsharing=True)
    cap = Capture(50)
    sd = {}
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        accs = ntu.train_sampled_models([np.array(c) for c in confs], cap, {"train": ListLoader(ttr, 16), "dev": ListLoader(tdv, 16)},
                                        args, "cpu", state_dict=sd)
    text = buf.getvalue()
    out["accs"] = np.array([float(a) for a in accs])
    out["hist"] = parse_hist(text)
    out["events"] = np.array(re.findall(r"(Creating|Updating|Loaded) shared weight with ID: (\S+)", text))
    for name, lsd in sd.items():
        for k, v in lsd.items():
            if "num_batches" not in k:
                put(out, "shared/" + name + "/" + k, v.detach().cpu().numpy())
    for i, m in enumerate(cap.models):
        for k, v in central_sd(m).items():
            if "num_batches" not in k:
                put(out, f"final{i}/" + k, v)
    for i, c in enumerate(confs):
        out[f"conf{i}"] = np.array(c)
    save("g16_weightsharing.npz", **out)


# ------------------------------------------------------------------ G17 main_found_ntu two-phase schedule (next#4)
def g17():
    """main_found_ntu.train_model (:94-157) unchanged, on stub backbones: phase 1 = 1 epoch, Adam(lr=eta_max/10) whose lr the
    scheduler overwrites from the first step; phase 2 = args.epochs epochs with a fresh Adam + scheduler over
    rmode.parameters() (== the central parameters here: the stubs own none), both with multitask (3-term loss, summed-
    logit argmax); then test_ntu_track_acc.  Deterministic mode (drpt=0 + BN), conf 0, R=16, B=16, 3 epochs;
    train/dev/test = synth_table(256/128/96, 61/62/63, snr=1.0, with_logits); params = init_params(conf, hp, 70)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_main_found", "/root/reference/main_found_ntu.py")
    for name in ("torchvision", "torchvision.transforms", "torchvision.datasets", "cv2"):
        sys.modules.setdefault(name, type(sys)(name))
    import types
    if "models.aux" not in sys.modules:     # D6: broken intra-repo import in datasets / mains
        pass
    mf = importlib.util.module_from_spec(spec)
    try:
        spec.loader.exec_module(mf)
    except Exception as e:                   # the script's own imports (datasets.ntu -> cv2 ...) may fail: restate nothing,
        raise RuntimeError(f"main_found_ntu.py does not import here: {e!r}")
    out = {}
    ttr, tdv, tte = table(256, 61, snr=1.0), table(128, 62, snr=1.0), table(96, 63, snr=1.0)
    for mt in (True, False):
        conf = np.array(CONFS["c0"])
        args = mkargs(inner_representation_size=16, batchnorm=True, drpt=0.0, epochs=3, batchsize=16, multitask=mt,
                      test_cp="", verbose=True)
        rmode = ntu.Searchable_Skeleton_Image_Net(args, conf)
        load_det(rmode, conf, args, 70)
        loaders = {"train": ListLoader(ttr, 16), "dev": ListLoader(tdv, 16), "test": ListLoader(tte, 16)}
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            acc = mf.train_model(rmode, conf, loaders, args, "cpu")
        text = buf.getvalue()
        pre = "mt/" if mt else "st/"
        out[pre + "test_acc"] = np.array(float(acc))
        out[pre + "hist"] = parse_hist(text)
        out[pre + "interm"] = np.array(float(re.search(r"Intermediate val accuracy: (?:tensor\()?([0-9.]+)", text).group(1)))
        out[pre + "final"] = np.array(float(re.search(r"Final val accuracy: (?:tensor\()?([0-9.]+)", text).group(1)))
        for k, v in central_sd(rmode).items():
            if "num_batches" not in k:
                put(out, pre + "final/" + k, v)
    save("g17_found_twophase.npz", **out)


# ------------------------------------------------------------------ G18 train-mode dropout pinned POINTWISE (mask injection)
class HashDropout(nn.Module):
    """Takes the place of the ``nn.Dropout`` INSTANCE of one constructed cell — the reference code that builds the cell
    (ntu_searchable.py:275-282) and calls it (:237,:240) is unchanged, exactly like the stub backbones.  Arithmetic = ATen's
    ``_dropout_impl`` (``noise = bernoulli(1-p); noise.div_(1-p); input * noise``) with the Bernoulli draw replaced by the
    counter-based keep mask the oracle and the HIP engine share (oracle.dropout_keep(seed, step, cell, B, R, p)); ``step``
    counts this module's train-mode calls = the candidate's global train step."""

    def __init__(self, p, seed, cell):
        super().__init__()
        self.p, self.seed, self.cell, self.step = float(p), int(seed), int(cell), 0

    def forward(self, x):
        if not self.training:
            return x
        keep = O.dropout_keep(self.seed, self.step, self.cell, x.shape[0], x.shape[1], self.p)
        self.step += 1
        noise = torch.from_numpy(keep.astype(np.float32))
        noise.div_(1 - self.p)
        return x * noise


def inject_masks(model, seed):
    n = 0
    for i, cell in enumerate(model.fusion_layers):
        last = len(cell) - 1
        assert isinstance(cell[last], nn.Dropout), cell
        cell[last] = HashDropout(cell[last].p, seed, i)
        n += 1
    return n


class CaptureMasked(Capture):
    """searchable_type factory: hash-generated central params (seed0 + index, optionally perturbed) AND hash dropout masks
    (drop_seed0 + index) — candidate i of the call is exactly the oracle's / engine's candidate i."""

    def __init__(self, seed0, drop_seed0, perturb_trial=None, same_params=False):
        super().__init__(seed0)
        self.drop_seed0, self.perturb_trial, self.same_params = drop_seed0, perturb_trial, same_params

    def __call__(self, args, conf):
        m = ntu.Searchable_Skeleton_Image_Net(args, conf)
        i = len(self.models)
        p = O.init_params(conf, hyper_of(args), self.seed0 + (0 if self.same_params else i))
        if self.perturb_trial is not None:
            p = O.perturb_params(p, self.perturb_trial)
        sd = m.state_dict()
        for k, v in p.items():
            sd[k].copy_(torch.from_numpy(v))
        inject_masks(m, self.drop_seed0 + i)
        self.models.append(m)
        return m


class OrderLoader(ListLoader):
    """DataLoader(shuffle=True) with the permutations fixed: epoch e of every candidate walks order[e]."""

    def __init__(self, table, B, order):
        super().__init__(table, B)
        self.order, self.calls = np.asarray(order), 0

    def __iter__(self):
        perm = torch.from_numpy(self.order[self.calls % len(self.order)].astype(np.int64))
        self.calls += 1
        for i in range(0, len(perm), self.B):
            sl = perm[i:i + self.B]
            rgb = D({k: self.t[k][sl] for k in ("v0", "v1", "v2", "v3", "vlogit")})
            ske = D({k: self.t[k][sl] for k in ("s0", "s1", "s2", "s3", "slogit")})
            yield {"rgb": rgb, "ske": ske, "label": self.t["label"][sl]}


G18A_VARIANTS = [("bndrop", dict(batchnorm=True, drpt=0.5)), ("drop", dict(batchnorm=False, drpt=0.5)),
                 ("bndrop04", dict(batchnorm=True, drpt=0.4)), ("drop04", dict(batchnorm=False, drpt=0.4))]


def g18a():
    """ONE train-mode forward + backward with dropout ON, through the unchanged module (ntu_searchable.py:206-247, cells
    :275-282 = [Linear, nl, BN, Dropout] and [Linear, nl, Dropout]) for nl in {ReLU, Sigmoid, LeakyReLU} (confs c4, l1, l2, l3),
    R = 16 (all) and 128 (c4, l2), batch rows 16 and 10 (a ragged last batch), step index 0 and 7 of the mask stream.
    table = synth_table(16, 11, snr=0.3, with_logits=True); params = init_params(conf, hp, seed, perturb_bn=True);
    masks = dropout_keep(drop seed = seed + 5, step, cell, rows, R, p).  Stored: logits, loss, preds, every central gradient,
    the BN running statistics after the step."""
    out, names = {}, []
    t = table(16, 11)
    for ci, cname in enumerate(("c4", "l1", "l2", "l3")):
        conf = CONFS[cname]
        for vi, (vname, kw) in enumerate(G18A_VARIANTS):
            for R in (16, 128):
                if R == 128 and cname not in ("c4", "l2"):
                    continue
                for rows, step in ((16, 0), (10, 7)):
                    if rows == 10 and vname.endswith("04") and R == 128:
                        continue
                    seed = 1800 + 100 * ci + 10 * vi + (R == 128)
                    args = mkargs(inner_representation_size=R, **kw)
                    model = ntu.Searchable_Skeleton_Image_Net(args, np.array(conf))
                    load_det(model, conf, args, seed, perturb_bn=True)
                    inject_masks(model, seed + 5)
                    for cell in model.fusion_layers:
                        cell[len(cell) - 1].step = step
                    model.train(True)
                    pre = f"{cname}/{vname}/{R}/{rows}/"
                    names.append(f"{cname}/{vname}/{R}/{rows}/{step}/{seed}")
                    rgb, ske = feats_of(t, slice(0, rows))
                    label = torch.from_numpy(t["label"][:rows])
                    output = model((rgb, ske))
                    loss = torch.nn.CrossEntropyLoss()(output, label)
                    out[pre + "logits"] = output.detach().numpy()
                    out[pre + "loss"] = np.array(loss.item())
                    out[pre + "preds"] = torch.max(output, 1)[1].numpy()
                    loss.backward()
                    for n_, p in model.named_parameters():
                        if p.grad is not None:
                            put(out, pre + "grad/" + n_, p.grad.numpy().copy())
                    for k, v in central_sd(model).items():
                        if "running" in k:
                            out[pre + "after/" + k] = v
    out["names"] = np.array(names)
    save("g18a_dropout_forward_backward.npz", **out)


G18B_CASES = {  # tag: (conf, R, bn, drpt, B, N_train, N_dev, snr)
    "search": ("c4", 16, False, 0.5, 20, 120, 60, 1.0),      # the search script's defaults (main_searchable_ntu.py:28,48,56): [Linear, nl, Dropout]
    "search_l3": ("l3", 16, False, 0.5, 20, 130, 70, 1.0),   # LeakyReLU cell, ragged last batches (130 = 6*20+10, 70 = 3*20+10)
    "bench": ("c4", 128, True, 0.5, 16, 64, 48, 0.3),        # BASELINE configs[1]'s cell: [Linear, nl, BN, Dropout]
    "bench16": ("l2", 16, True, 0.4, 16, 64, 48, 0.3),
}


def g18b_order(tag, E, N):
    rng = np.random.default_rng(1800 + sum(map(ord, tag)))
    return np.stack([rng.permutation(N) for _ in range(E)])


def g18b():
    """Dropout ON, shuffled FIXED order, through the unchanged train loop: (i) a replica of train_ntu_track_acc's train-phase
    body (:46-69) dumping W / m / v after 1, 2 and 10 Adam steps; (ii) the unchanged train_sampled_models ->
    train_ntu_track_acc from the same start for 3 epochs: per-epoch history as printed, best accuracy, final parameters.
    params = init_params(conf, hp, 5), masks seeded 40, order = g18b_order(tag, 3, N_train)."""
    out = {}
    E = 3
    for tag, (cname, R, bn, drpt, B, N, Nd, snr) in G18B_CASES.items():
        conf = np.array(CONFS[cname])
        ttr, tdv = table(N, 21, with_logits=False, snr=snr), table(Nd, 22, with_logits=False, snr=snr)
        order = g18b_order(tag, E, N)
        args = mkargs(inner_representation_size=R, batchnorm=bn, drpt=drpt, epochs=E, batchsize=B)
        model = ntu.Searchable_Skeleton_Image_Net(args, conf)
        load_det(model, conf, args, 5)
        inject_masks(model, 40)
        pre = tag + "/"
        opt = torch.optim.Adam(model.central_params(), lr=args.eta_max, weight_decay=1e-4)
        sched = sc.LRCosineAnnealingScheduler(args.eta_max, args.eta_min, args.Ti, args.Tm, N / B)
        crit = torch.nn.CrossEntropyLoss()
        model.train(True)
        step, losses = 0, []
        loader = OrderLoader(ttr, B, order)
        for ep in range(E):
            for data in loader:
                opt.zero_grad()
                output = model((data["rgb"], data["ske"]))
                loss = crit(output, data["label"])
                sched.step()
                sched.update_optimizer(opt)
                loss.backward()
                opt.step()
                losses.append(loss.item())
                step += 1
                if step in (1, 2, 10):
                    for k, v in central_sd(model).items():
                        if "num_batches" not in k:
                            put(out, pre + f"step{step}/p/" + k, v)
                    name_of = {id(p): n for n, p in model.named_parameters()}
                    for p, st in opt.state.items():
                        put(out, pre + f"step{step}/m/" + name_of[id(p)], st["exp_avg"].numpy().copy())
                        put(out, pre + f"step{step}/v/" + name_of[id(p)], st["exp_avg_sq"].numpy().copy())
        out[pre + "losses"] = np.array(losses)
        cap = CaptureMasked(5, 40)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            accs = ntu.train_sampled_models([conf], cap, {"train": OrderLoader(ttr, B, order), "dev": ListLoader(tdv, B)}, args, "cpu")
        out[pre + "best_acc"] = np.array(float(accs[0]))
        out[pre + "hist"] = parse_hist(buf.getvalue())
        out[pre + "meta"] = np.array([R, int(bn), drpt, B, N, Nd, snr])
        for k, v in central_sd(cap.models[0]).items():
            if "num_batches" not in k:
                put(out, pre + "final/" + k, v)
    save("g18b_dropout_trajectory.npz", **out)


G18C_SNR = 0.12


def g18c():
    """The G14 experiment WITH dropout 0.5 and a shuffled fixed order: BASELINE configs[1] at full size (conf 4, R=128,
    --batchnorm, drpt 0.5, B=16, N=10,000/5,600, bf16-rounded taps at snr 0.12 = bench.py's tables, 3 epochs) through the
    unchanged train_sampled_models from NT=64 starts that differ by a 1e-7 relative perturbation of the initial weight
    matrices; every start sees the SAME masks (seed 4040) and the SAME order (default_rng(1812)).  The oracle and the engine
    run from the same starts with the same masks / order (tests/test_fullsize.py): all three ensembles must agree in
    distribution.  Splittable like g15: G18C_T0=a G18C_NT=n -> part file; `g18cmerge`.
    Committed: `for a in 0 16 32 48; do G18C_T0=$a G18C_NT=16 python make_golden.py g18c & done; wait; python make_golden.py g18cmerge`."""
    NT = int(os.environ.get("G18C_NT", "64"))
    t0 = int(os.environ.get("G18C_T0", "0"))
    E = 3
    snr = G18C_SNR
    ttr = dict(O.synth_table(10000, 1, snr=snr, quant="bf16"), vlogit=np.zeros((10000, 60), np.float32), slogit=np.zeros((10000, 60), np.float32))
    tdv = dict(O.synth_table(5600, 2, snr=snr, quant="bf16"), vlogit=np.zeros((5600, 60), np.float32), slogit=np.zeros((5600, 60), np.float32))
    rng = np.random.default_rng(1812)
    order = np.stack([rng.permutation(10000) for _ in range(E)])
    conf = np.array(CONFS["c4"])
    args = mkargs(inner_representation_size=128, batchnorm=True, drpt=0.5, epochs=E, batchsize=16)
    torch.set_num_threads(int(os.environ.get("G18C_THREADS", "1")))
    hists, bests = [], []
    for trial in range(t0, t0 + NT):
        cap = CaptureMasked(77, 4040, perturb_trial=trial)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            accs = ntu.train_sampled_models([conf], cap, {"train": OrderLoader(ttr, 16, order), "dev": ListLoader(tdv, 16)}, args, "cpu")
        h = parse_hist(buf.getvalue())
        hists.append(h)
        bests.append(float(accs[0]))
        print("g18c trial", trial, bests[-1], h[:, 1:].ravel(), flush=True)
    arrs = dict(hist=np.array(hists), best_acc=np.array(bests), trials=np.arange(t0, t0 + NT), rel=np.array([1e-7]),
                meta=np.array([10000, 5600, snr, 128, 16, E, 1, 0.5, 77, 4040, 1812]))
    if "G18C_T0" in os.environ:
        save(f"g18c_part_{t0:03d}.npz", **arrs)
    else:
        save("g18c_dropout_envelope.npz", **arrs)


def g18cmerge():
    import glob
    parts = [np.load(f) for f in sorted(glob.glob(os.path.join(HERE, "g18c_part_*.npz")))]
    assert parts and all(np.array_equal(p["meta"], parts[0]["meta"]) for p in parts)
    trials = np.concatenate([p["trials"] for p in parts])
    assert np.array_equal(trials, np.arange(len(trials))), trials
    save("g18c_dropout_envelope.npz", hist=np.concatenate([p["hist"] for p in parts]),
         best_acc=np.concatenate([p["best_acc"] for p in parts]), trials=trials, rel=parts[0]["rel"], meta=parts[0]["meta"])


# ------------------------------------------------------------------ G19 the search-default regime at FULL size (BASELINE configs[2])
def sampled_l4(n):
    """bench.py's configs[2] / configs[3] populations: L=4 confs sampled like the controller does at progression level 3
    (np.random.seed(0); rows drawn from the reference's own get_possible_layer_configurations(0))."""
    np.random.seed(0)
    layer = ntu.get_possible_layer_configurations(0)
    return [np.array([layer[i] for i in np.random.choice(len(layer), 4)]) for _ in range(n)]


G19_SNR = 0.12


def _g19_tables():
    ttr = dict(O.synth_table(10000, 1, snr=G19_SNR, quant="bf16"), vlogit=np.zeros((10000, 60), np.float32), slogit=np.zeros((10000, 60), np.float32))
    tdv = dict(O.synth_table(5600, 2, snr=G19_SNR, quant="bf16"), vlogit=np.zeros((5600, 60), np.float32), slogit=np.zeros((5600, 60), np.float32))
    return ttr, tdv


def g19b():
    """BASELINE configs[2] exactly as bench.py runs it (`--workload c2` / `small_pop.c2`), through the unchanged reference:
    ONE train_sampled_models call on the 16 np.random.seed(0) L=4 confs at the search script's defaults
    (main_searchable_ntu.py:26-47,56; models/searchable.py:90,120): R=16, no batchnorm, drpt 0.5, B=20, E=10,
    N=10,000/5,600, bf16-rounded taps at snr 0.12, shuffled train order.  torch.manual_seed(1900 + seed) for the reference's
    OWN dropout (Philox) / shuffle streams; candidate i of seed s starts from init_params(conf_i, hp, 19000 + 100 s + i).
    Splittable: G19_SEED0=a G19_NS=n -> g19b_part_<a>.npz (git-ignored); `g19bmerge`.
    Committed: `for a in 0 4 ... 28; do G19_SEED0=$a G19_NS=4 python make_golden.py g19b & done; wait; python make_golden.py g19bmerge`."""
    NS = int(os.environ.get("G19_NS", "32"))
    seed0 = int(os.environ.get("G19_SEED0", "0"))
    E = int(os.environ.get("G19_E", "10"))
    K = int(os.environ.get("G19_K", "16"))
    ttr, tdv = _g19_tables()
    confs = sampled_l4(16)[:K]
    args = mkargs(inner_representation_size=16, batchnorm=False, drpt=0.5, epochs=E, batchsize=20)
    torch.set_num_threads(1)
    loaders = {"train": ShuffleLoader(ttr, 20), "dev": ListLoader(tdv, 20)}
    bests, hists = [], []
    import time
    meta = np.array([10000, 5600, G19_SNR, 16, 20, E, 0, 0.5])   # N,Ndev,snr,R,B,epochs,bn,drpt
    part = os.path.join(HERE, f"g19b_part_{seed0:03d}.npz")
    if "G19_SEED0" in os.environ and os.path.exists(part):        # resume a part that was interrupted (every seed is independent)
        old = np.load(part)
        if np.array_equal(old["meta"], meta) and np.array_equal(old["confs"], np.array(confs)) and int(old["seeds"][0]) == seed0:
            bests, hists = list(old["best_acc"]), list(old["dev_acc"])
    for seed in range(seed0 + len(bests), seed0 + NS):
        t0 = time.time()
        torch.manual_seed(1900 + seed)
        accs, _, hist = run_tsm(confs, args, loaders, 19000 + 100 * seed)
        bests.append(accs)
        hists.append(hist.reshape(K, 2 * E, 3)[:, 1::2, 2])     # [cand][epoch] dev accuracy as printed (4 decimals)
        print("g19b seed", seed, "%.0fs" % (time.time() - t0), np.round(accs, 4), flush=True)
        if "G19_SEED0" in os.environ and not os.environ.get("G19_PROBE"):
            np.savez_compressed(part, best_acc=np.array(bests), dev_acc=np.array(hists), seeds=np.arange(seed0, seed0 + len(bests)),
                                confs=np.array(confs), meta=meta)
    if os.environ.get("G19_PROBE"):
        return
    arrs = dict(best_acc=np.array(bests), dev_acc=np.array(hists), seeds=np.arange(seed0, seed0 + NS), confs=np.array(confs), meta=meta)
    if "G19_SEED0" in os.environ:
        save(f"g19b_part_{seed0:03d}.npz", **arrs)
    else:
        save("g19b_search_default_streams.npz", **arrs)


def g19bmerge():
    import glob
    parts = [np.load(f) for f in sorted(glob.glob(os.path.join(HERE, "g19b_part_*.npz")))]
    if parts and int(parts[0]["seeds"][0]) != 0:      # extension run (round 6): committed fixture first, the parts go on from its last seed
        parts.insert(0, np.load(os.path.join(HERE, "g19b_search_default_streams.npz")))
    assert parts and all(np.array_equal(p["meta"], parts[0]["meta"]) and np.array_equal(p["confs"], parts[0]["confs"]) for p in parts)
    seeds = np.concatenate([p["seeds"] for p in parts])
    assert np.array_equal(seeds, np.arange(len(seeds))), seeds
    save("g19b_search_default_streams.npz", best_acc=np.concatenate([p["best_acc"] for p in parts]),
         dev_acc=np.concatenate([p["dev_acc"] for p in parts]), seeds=seeds, confs=parts[0]["confs"], meta=parts[0]["meta"])


G19A_CONFS = (0, 5, 11)      # indices into sampled_l4(16)


def g19a():
    """G18c in the search-default regime: confs G19A_CONFS of the configs[2] population at full size (R=16, no batchnorm,
    drpt 0.5, B=20, N=10,000/5,600, bf16-rounded taps at snr 0.12, 3 epochs) through the unchanged train_sampled_models
    from NT starts per conf that differ by a 1e-7 relative perturbation of the initial weight matrices
    (O.perturb_params(init_params(conf, hp, 79), trial)); every start sees the SAME masks (seed 4141) and the SAME order
    (default_rng(1913)).  Splittable: G19A_T0=a G19A_NT=n -> g19a_part_<a>.npz; `g19amerge`."""
    NT = int(os.environ.get("G19A_NT", "64"))
    t0 = int(os.environ.get("G19A_T0", "0"))
    E = 3
    ttr, tdv = _g19_tables()
    rng = np.random.default_rng(1913)
    order = np.stack([rng.permutation(10000) for _ in range(E)])
    allc = sampled_l4(16)
    confs = [allc[i] for i in G19A_CONFS]
    args = mkargs(inner_representation_size=16, batchnorm=False, drpt=0.5, epochs=E, batchsize=20)
    torch.set_num_threads(1)
    hists, bests = [], []
    for trial in range(t0, t0 + NT):
        hrow, brow = [], []
        for conf in confs:
            cap = CaptureMasked(79, 4141, perturb_trial=trial)
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                accs = ntu.train_sampled_models([conf], cap, {"train": OrderLoader(ttr, 20, order), "dev": ListLoader(tdv, 20)}, args, "cpu")
            hrow.append(parse_hist(buf.getvalue()))
            brow.append(float(accs[0]))
        hists.append(hrow)
        bests.append(brow)
        print("g19a trial", trial, brow, flush=True)
    arrs = dict(hist=np.array(hists), best_acc=np.array(bests), trials=np.arange(t0, t0 + NT), rel=np.array([1e-7]), confs=np.array(confs),
                meta=np.array([10000, 5600, G19_SNR, 16, 20, E, 0, 0.5, 79, 4141, 1913]))
    if "G19A_T0" in os.environ:
        save(f"g19a_part_{t0:03d}.npz", **arrs)
    else:
        save("g19a_search_default_envelope.npz", **arrs)


def g19amerge():
    import glob
    parts = [np.load(f) for f in sorted(glob.glob(os.path.join(HERE, "g19a_part_*.npz")))]
    assert parts and all(np.array_equal(p["meta"], parts[0]["meta"]) for p in parts)
    trials = np.concatenate([p["trials"] for p in parts])
    assert np.array_equal(trials, np.arange(len(trials))), trials
    save("g19a_search_default_envelope.npz", hist=np.concatenate([p["hist"] for p in parts]),
         best_acc=np.concatenate([p["best_acc"] for p in parts]), trials=trials, rel=parts[0]["rel"], confs=parts[0]["confs"], meta=parts[0]["meta"])


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g23", "g456", "g7", "g8", "g9", "g10", "g11", "g12", "g13", "g14", "g14m", "g15", "g16", "g17", "g18a", "g18b", "g18c"]
    for w in which:
        globals()[w]()

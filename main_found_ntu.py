#!/usr/bin/env python3
"""Train / test one found fusion architecture — counterpart of /root/reference/main_found_ntu.py (flags :24-68,
confs :173-182) on the MI355X engine with precomputed taps.  Phase 1 = 1 epoch on central_params (:108-123);
phase 2 = `--epochs` epochs with a fresh Adam + scheduler (:128-137) — restricted to the central parameters because
the backbones are feature tables here (fine-tuning them needs raw video: out of scope); then the test pass (:152)."""
import argparse
import time

import numpy as np
import torch

CONFS = {0: [[2, 2, 0], [1, 0, 1], [3, 2, 0], [3, 1, 1]], 1: [[3, 0, 0], [1, 3, 0], [1, 1, 1], [3, 3, 0]],
         2: [[3, 2, 0], [2, 3, 1], [0, 1, 1], [3, 0, 0]], 3: [[1, 1, 1], [3, 2, 0], [0, 1, 1], [3, 0, 0]],
         4: [[3, 1, 1], [1, 3, 0], [1, 1, 1], [3, 3, 0]]}


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Train a found fusion network (MI355X engine).")
    p.add_argument("--checkpointdir", type=str, default="")
    p.add_argument("--datadir", type=str, default="")
    p.add_argument("--ske_cp", type=str, default="")
    p.add_argument("--rgb_cp", type=str, default="")
    p.add_argument("--test_cp", type=str, default="")
    p.add_argument("--num_outputs", type=int, default=60)
    p.add_argument("--batchsize", type=int, default=20)
    p.add_argument("--inner_representation_size", type=int, default=256)
    p.add_argument("--epochs", type=int, default=70)
    p.add_argument("--eta_max", type=float, default=0.001)
    p.add_argument("--eta_min", type=float, default=0.000001)
    p.add_argument("--Ti", type=int, default=5)
    p.add_argument("--Tm", type=int, default=2)
    p.add_argument("--use_dataparallel", action="store_true", default=False)
    p.add_argument("--j", dest="num_workers", type=int, default=16)
    p.add_argument("--modality", type=str, default="both")
    p.add_argument("--no-verbose", action="store_false", dest="verbose", default=True)
    p.add_argument("--weightsharing", action="store_true", default=False)
    p.add_argument("--no-multitask", dest="multitask", action="store_false", default=True)
    p.add_argument("--alphas", action="store_true", default=False)
    p.add_argument("--batchnorm", action="store_true", default=False)
    p.add_argument("--vid_len", default=(8, 32), type=int, nargs="+")
    p.add_argument("--drpt", default=0.4, type=float)
    p.add_argument("--conf", type=int, default=1)
    p.add_argument("--featuredir", type=str, default="")
    p.add_argument("--synthetic", type=int, nargs=3, metavar=("N_TRAIN", "N_DEV", "N_TEST"), default=None)
    p.add_argument("--feature_dtype", default="bf16", choices=["bf16", "f16", "f32"])
    p.add_argument("--seed", type=int, default=0)
    return p.parse_args(argv)


def train_model(rmode, configuration, dataloaders, args, device):
    import mfas_amd as M
    sizes = {x: len(dataloaders[x].dataset) for x in ("train", "test", "dev")}
    if args.test_cp == "":
        nbpe = sizes["train"] / args.batchsize
        criteria = [torch.nn.CrossEntropyLoss()] * 3
        opt = torch.optim.Adam(rmode.central_params(), lr=args.eta_max / 10, weight_decay=1e-4)
        sched = M.LRCosineAnnealingScheduler(args.eta_max, args.eta_min, args.Ti, args.Tm, nbpe)
        if args.verbose:
            print("Pretraining central weights: ")
            print(configuration)
        acc = M.train_ntu_track_acc(rmode, criteria, opt, sched, dataloaders, sizes, device=device, num_epochs=1,
                                    verbose=args.verbose, multitask=args.multitask)
        if args.verbose:
            print("Intermediate val accuracy: " + str(acc))
        opt = torch.optim.Adam(rmode.central_params(), lr=args.eta_max, weight_decay=1e-4)
        sched = M.LRCosineAnnealingScheduler(args.eta_max, args.eta_min, args.Ti, args.Tm, nbpe)
        acc = M.train_ntu_track_acc(rmode, criteria, opt, sched, dataloaders, sizes, device=device,
                                    num_epochs=args.epochs, verbose=args.verbose, multitask=args.multitask)
        if args.verbose:
            print("Final val accuracy: " + str(acc))
    else:
        import os
        rmode.load_state_dict(torch.load(os.path.join(args.checkpointdir, args.test_cp)))
    test_acc = M.test_ntu_track_acc(rmode, dataloaders, sizes, device=device, multitask=args.multitask)
    if args.verbose:
        print("Final test accuracy: " + str(test_acc))
    return test_acc


def main(argv=None):
    import mfas_amd as M
    print("Training found NTU network")
    args = parse_args(argv)
    device = torch.device("cuda:0")
    torch.manual_seed(args.seed)
    configuration = np.array(CONFS[args.conf])
    rmode = M.Searchable_Skeleton_Image_Net(args, configuration)
    dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.feature_dtype]
    if args.synthetic:
        tabs = {s: M.FeatureTable.synthetic(n, i + 1, device, dt, with_logits=True)
                for i, (s, n) in enumerate(zip(("train", "dev", "test"), args.synthetic))}
    else:
        tabs = {s: M.FeatureTable.load(args.featuredir, s, device) for s in ("train", "dev", "test")}
    loaders = {s: M.FeatureLoader(t, args.batchsize, shuffle=True) for s, t in tabs.items()}
    t0 = time.time()
    acc = train_model(rmode, configuration, loaders, args, device)
    el = time.time() - t0
    print("Training in {:.0f}m {:.0f}s".format(el // 60, el % 60))
    print("Model Acc: {}".format(acc))
    return acc


if __name__ == "__main__":
    main()

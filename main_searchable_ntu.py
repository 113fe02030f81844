#!/usr/bin/env python3
"""MFAS search on NTU-shaped feature tables — counterpart of /root/reference/main_searchable_ntu.py (same flags,
:16-63) on the MI355X engine.  Data: --featuredir with '<split>_<tap>.npy' tables ('train' = the reference's
'trainexp' split, 'dev'), or --synthetic N_train N_dev.  Multi-GPU: torchrun --nproc-per-node N main_searchable_ntu.py
(population sharded inside train_sampled_models, accuracies all-gathered over RCCL)."""
import argparse
import os
import sys
import time

import numpy as np
import torch


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Modality optimization (MI355X engine).")
    p.add_argument("--checkpointdir", type=str, default="")            # kept for flag compatibility (unused: no backbones)
    p.add_argument("--datadir", type=str, default="")
    p.add_argument("--ske_cp", type=str, default="")
    p.add_argument("--rgb_cp", type=str, default="")
    p.add_argument("--num_outputs", type=int, default=60)
    p.add_argument("--batchsize", type=int, default=20)
    p.add_argument("--inner_representation_size", type=int, default=16)
    p.add_argument("--epochs", type=int, default=3)
    p.add_argument("--lr_surrogate", type=float, default=0.001)
    p.add_argument("--epochs_surrogate", type=int, default=50)
    p.add_argument("--eta_max", type=float, default=0.001)
    p.add_argument("--eta_min", type=float, default=0.000001)
    p.add_argument("--Ti", type=int, default=1)
    p.add_argument("--Tm", type=int, default=2)
    p.add_argument("--use_dataparallel", action="store_true", default=False)
    p.add_argument("--num_workers", type=int, default=16)
    p.add_argument("--modality", type=str, default="both")
    p.add_argument("--max_fusions", type=int, dest="max_progression_levels", default=4)
    p.add_argument("--search_iterations", type=int, default=3)
    p.add_argument("--num_samples", type=int, default=15)
    p.add_argument("--initial_temperature", type=float, default=10.0)
    p.add_argument("--final_temperature", type=float, default=0.2)
    p.add_argument("--temperature_decay", type=float, default=4.0)
    p.add_argument("--no-verbose", dest="verbose", action="store_false", default=True)
    p.add_argument("--weightsharing", action="store_true", default=False)
    p.add_argument("--alphas", action="store_true", default=False)
    p.add_argument("--batchnorm", action="store_true", default=False)
    p.add_argument("--multitask", action="store_true", default=False)
    p.add_argument("--vid_len", default=(8, 32), type=int, nargs="+")
    p.add_argument("--drpt", default=0.5, type=float)
    # new flags (engine / data)
    p.add_argument("--featuredir", type=str, default="", help="directory of exported pooled-tap tables")
    p.add_argument("--synthetic", type=int, nargs=2, metavar=("N_TRAIN", "N_DEV"), default=None)
    p.add_argument("--feature_dtype", default="bf16", choices=["bf16", "f16", "f32"])
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--random_search", action="store_true", default=False)
    p.add_argument("--engine_init", default="torch", choices=["torch", "device"])
    p.add_argument("--engine_order", default="per_candidate", choices=["shared", "per_candidate"],
                   help="per_candidate (default): every candidate draws its own permutations, as the reference's per-candidate "
                        "DataLoader(shuffle=True) does (models/searchable.py:248-250); shared: one shuffled order per epoch for the whole call (lockstep)")
    p.add_argument("--engine_all_ranks", action="store_true", default=False,
                   help="under torchrun: shard every call over ALL ranks (default: only as many ranks as the calibrated step-time model "
                        "says shorten the call, mfas_amd/population.py)")
    p.add_argument("--surrogate_device", default="cpu", choices=["cpu", "gpu"],
                   help="where the 81k-parameter LSTM surrogate trains (the reference puts it on its training device).  gpu: train steps "
                        "replayed as HIP graphs (0.87 ms instead of the CPU path's 1.9 ms per step; device GEMM numerics, so sampled "
                        "configurations may differ from the CPU path's in the last digits)")
    p.add_argument("--dist_backend", default="nccl", choices=["nccl", "gloo"],
                   help="torch.distributed backend under torchrun: nccl = RCCL over xGMI (one GPU per rank); gloo lets several ranks "
                        "share one GPU (tests)")
    p.add_argument("--timing", action="store_true", help="print how the wall time splits into candidate training (GPU) and the controller / surrogate (CPU)")
    p.add_argument("--controller_threads", type=int, default=4,
                   help="torch CPU threads for the 81k-parameter surrogate (more threads only add overhead)")
    return p.parse_args(argv)


def main(argv=None):
    import mfas_amd as M
    from mfas_amd.search import NTUSearcher
    args = parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        if args.dist_backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=device)
        else:
            torch.distributed.init_process_group("gloo")
    torch.set_num_threads(max(1, args.controller_threads))
    torch.manual_seed(args.seed)          # every rank runs the same (seeded) controller
    np.random.seed(args.seed)
    import random
    random.seed(args.seed)                # tools.sample_k_configurations_directly draws depths with random.randint
    dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.feature_dtype]
    if args.synthetic:
        tables = {"train": M.FeatureTable.synthetic(args.synthetic[0], 1, device, dt),
                  "dev": M.FeatureTable.synthetic(args.synthetic[1], 2, device, dt)}
    else:
        tables = {s: M.FeatureTable.load(args.featuredir, s, device) for s in ("train", "dev")}
    # torch.optim's first optimizer construction imports torch._dynamo (0.6-0.7 s).  The table generation above is asynchronous device
    # work, so importing it HERE — on the main thread, after every other import and after init_process_group — still overlaps it,
    # without a second thread importing overlapping torch submodules behind the main thread's back.
    try:
        import importlib
        importlib.import_module("torch._dynamo")
    except Exception as e:     # an optional warm-up: never a reason to lose the search
        print("warm import of torch._dynamo failed:", repr(e), file=sys.stderr)
    searcher = NTUSearcher(args, device, tables)
    rank0 = int(os.environ.get("RANK", "0")) == 0
    if rank0:
        print("MFAS for NTU Started!!!!")
    spent = {"train": 0.0, "calls": 0, "cands": 0}
    if args.timing:
        from mfas_amd import ntu_searchable as _ntu
        inner = _ntu.train_sampled_models

        def timed(confs, *a, **kw):
            torch.cuda.synchronize()
            t = time.time()
            out = inner(confs, *a, **kw)
            torch.cuda.synchronize()
            spent["train"] += time.time() - t
            spent["calls"] += 1
            spent["cands"] += len(confs)
            return out
        _ntu.train_sampled_models = timed
    t0 = time.time()
    if args.random_search:
        from mfas_amd import ntu_searchable as ntu
        data = searcher._randsearch(ntu.Searchable_Skeleton_Image_Net, searcher.dataloaders,
                                    {"train_sampled_fun": ntu.train_sampled_models,
                                     "get_layer_confs": ntu.get_possible_layer_configurations}, device)
    else:
        data = searcher.search(surrogate_device=device if args.surrogate_device == "gpu" else "cpu")
    el = time.time() - t0
    if rank0:
        print("Search complete in {:.0f}m {:.0f}s".format(el // 60, el % 60))
        if args.timing:
            print("timing: {:.2f} s total = {:.2f} s candidate training ({} calls, {} candidates, {:.1f} cand/s) + {:.2f} s "
                  "controller/surrogate".format(el, spent["train"], spent["calls"], spent["cands"],
                                                spent["cands"] / max(spent["train"], 1e-9), el - spent["train"]))
        k_best, k_accs, _ = data.get_k_best(5)
        print("Now listing best architectures")
        for c, a in sorted(zip(k_best, k_accs), key=lambda t: -t[1]):
            print(np.asarray(c).tolist(), float(a))
    if world > 1:
        torch.distributed.destroy_process_group()
    return data


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py — candidate-architectures trained / second on N MI355X (BASELINE.json metric).

One "step" = one ``train_sampled_models`` call (the reference's plug-in boundary) on a population of
--pop candidates PER GPU: build + init the candidates, train each for E epochs of {train over N_train,
eval over N_dev}, gather the best dev accuracies.  Workload at N=1 = BASELINE configs[1]: NTU found conf 4
(``--inner_representation_size 128 --batchnorm``), precomputed (synthetic, planted-signal) NTU-shaped taps
stored bf16, B=16, drpt 0.5, E=10, N_train=10,000, N_dev=5,600.  Feature tables are
resident in HBM before the timed region.  Weak scaling (default): every rank trains --pop candidates.

Strong scaling of a SMALL population — what the search actually issues (models/searchable.py:90,120: sequential calls of
<= num_samples configurations) — is `--total-pop K`: K candidates in total, sharded over the ranks.  Named workloads:
  --workload c1   BASELINE configs[1] (default): conf 4, R=128, --batchnorm, B=16 (weak scaling, --pop per GPU)
  --workload c2   BASELINE configs[2]: 16 sampled L=4 confs (np.random.seed(0)), search-script defaults R=16, B=20, no BN,
                  drpt 0.5, E=10, total population 16 (strong scaling)
  --workload c3   one call of BASELINE configs[3]: 50 sampled confs, same defaults, total population 50 (strong scaling)
  --workload c5   BASELINE configs[4] (the fifth): MM-IMDB-shaped fusion search — text taps 64 / 128, image taps 4 x 512, 23 genres,
                  weighted BCE + F1-samples, fp16 taps, R=16, B=20, N = 15,552 / 2,608 (the MM-IMDB split sizes), --pop sampled
                  L=4 confs per GPU (weak scaling; default 512: the "memory-bound cell-kernel stress")

The default line (workload c1, no workload flags) ALSO carries the search-sized workloads, measured after the timed region:
  N = 1: ``config.small_pop = {"c2": {...}, "c3": {...}, "c1_single": {...}, "c1_pop6": {...}, "c1_pop12": {...}}`` — one call of 16 / 50 sampled L=4 confs at the search
         script's defaults (candidates/s, us per train step, schedule, nominal fraction of the HBM bound) and ONE conf-4 R=128
         candidate (train steps/s: SURVEY.md 8d C2);
  N > 1: ``config.strong = {"c2": {...}, "c3": {...}}`` — the same two calls sharded over the ranks by the engine's own policy
         (per-rank seconds, per-rank shares, ranks the sharder used, the calibrated step-time model).

Launch: ``python bench.py --gpus N`` (N > 1 and no WORLD_SIZE in the environment: bench.py starts its own N ranks under
torch.distributed.run on 127.0.0.1 and exits non-zero if any of them fails) or
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
bench.py --gpus N --steps K --warmup W``.
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONF4 = [[3, 1, 1], [1, 3, 0], [1, 1, 1], [3, 3, 0]]      # main_found_ntu.py:182
HBM_PEAK_GBS = 8000.0                                     # MI355X_MICROARCH.md: 8.0 TB/s spec


def synth_tables(n_train, n_dev, device, dtype, snr=0.12, C=60):
    """Planted-signal NTU-shaped taps x = relu(snr*mu[label] + eps) generated on the GPU (same on every rank)."""
    import mfas_amd as M
    g = torch.Generator(device=device)
    g.manual_seed(123)
    mus = {}
    for name, sizes in (("s", M.engine.S_SIZES), ("v", M.engine.V_SIZES)):
        for j, w in enumerate(sizes):
            mus[f"{name}{j}"] = torch.randn(C, w, generator=g, device=device)
    out = []
    for n, seed in ((n_train, 1), (n_dev, 2)):
        g.manual_seed(seed)
        label = torch.randint(0, C, (n,), generator=g, device=device)
        taps = {k: torch.relu(snr * mu[label] + torch.randn(n, mu.shape[1], generator=g, device=device)).to(dtype)
                for k, mu in mus.items()}
        out.append(M.FeatureTable(taps, label.to(torch.int32)))
    return out


def _cpu_sample(O, ohp, conf, ttr, tdv, args, n_full, budget_s):
    params = O.init_params(conf, ohp, 1)
    keys = O.trainable_keys(conf, ohp)
    st = O.AdamState()
    n_tr, n_dv = len(ttr["label"]), len(tdv["label"])
    etas = O.eta_sequence(1e-3, 1e-6, 1, 2, n_full / args.batch, n_tr // args.batch)
    t0 = time.perf_counter()
    steps = 0
    for bi in range(n_tr // args.batch):
        idx = np.arange(bi * args.batch, (bi + 1) * args.batch)
        feats = {k: v[idx] for k, v in ttr.items() if k != "label"}
        logits, cache = O.forward(params, conf, ohp, feats, True, seed=1, step=bi)
        loss, dlog, _ = O.ce_loss(logits, ttr["label"][idx])
        grads = O.backward(params, ohp, cache, dlog)
        O.bn_update_running(params, ohp, cache)
        O.adam_step(params, grads, st, float(etas[bi]), ohp, keys)
        steps += 1
        if time.perf_counter() - t0 > budget_s * 0.75:
            break
    t_step = (time.perf_counter() - t0) / steps
    t0 = time.perf_counter()
    rows = 0
    for bi in range(n_dv // args.batch):
        idx = np.arange(bi * args.batch, (bi + 1) * args.batch)
        feats = {k: v[idx] for k, v in tdv.items() if k != "label"}
        logits, _ = O.forward(params, conf, ohp, feats, False)
        O.ce_loss(logits, tdv["label"][idx])
        rows += len(idx)
        if time.perf_counter() - t0 > budget_s * 0.25:
            break
    return t_step, (time.perf_counter() - t0) / rows, steps, rows


def cpu_baseline(train, dev, args, budget_s=24.0):
    """CPU baselines on this box's host cores, bounded samples of the same workload.  Headline: the PyTorch-CPU eager restatement of
    the reference loop (oracle/torch_restatement.py) over one epoch at 1 / 8 / all threads, fastest reported with ITS thread
    count.  Nested (`numpy_port`): the numpy oracle (oracle/np_oracle.py) at 1 / 8 / 32 BLAS threads, extrapolated linearly."""
    from threadpoolctl import threadpool_limits
    from oracle import np_oracle as O
    ohp = O.Hyper(R=args.R, B=args.batch, bn=not args.no_bn, drpt=args.drpt, epochs=1)
    n_tr = min(len(train), 1600)
    n_dv = min(len(dev), 1600)
    ttr = {k: v[:n_tr].float().cpu().numpy() for k, v in train.taps.items()}
    ttr["label"] = train.label[:n_tr].cpu().numpy().astype(np.int64)
    tdv = {k: v[:n_dv].float().cpu().numpy() for k, v in dev.taps.items()}
    tdv["label"] = dev.label[:n_dv].cpu().numpy().astype(np.int64)
    conf = np.array(CONF4)
    nb = -(-len(train) // args.batch)
    best = None
    tried = [t for t in (1, 8, 32) if t <= (os.cpu_count() or 1)]
    for nt in tried:
        with threadpool_limits(limits=nt):
            t_step, t_row, steps, rows = _cpu_sample(O, ohp, conf, ttr, tdv, args, len(train), budget_s / len(tried))
        t_cand = args.epochs * (nb * t_step + len(dev) * t_row)
        if best is None or t_cand < best[0]:
            best = (t_cand, nt, t_step, t_row, steps, rows)
    t_cand, nt, t_step, t_row, steps, rows = best
    out = {"value": 1.0 / t_cand, "unit": "candidates/s", "cores": nt, "kind": "port",
           "sample": f"{steps} train steps + {rows} dev rows of conf-4 R={args.R} B={args.batch} f32 (numpy oracle, "
                     f"BLAS threads tried {tried}, best {nt}; host has {os.cpu_count()} logical cores), extrapolated to "
                     f"E={args.epochs} x ({nb} steps + {len(dev)} dev rows): {t_step * 1e3:.2f} ms/step, "
                     f"{t_row * 1e6:.1f} us/dev-row"}
    # the PyTorch-CPU eager restatement of the reference step sequence (oracle/torch_restatement.py),
    # ONE FULL epoch (all train batches + the whole dev table) on all host cores and, for comparison, on 8 threads
    try:
        import subprocess
        runs = []
        ncpu = os.cpu_count() or 1
        # 1 / 8 / 32 threads, a FULL epoch each where it fits the box (about 6-10 s at 8 threads).  Eager per-op dispatch does not
        # scale to hundreds of threads (the survey measured 1.24x from 1 to 8; round 4's all-cores leg spent its whole 15 s box on
        # ONE train step of a 256-thread host and could never win): thread counts are capped at 32.  Each run is a subprocess with
        # a hard timeout.
        top = min(32, ncpu)
        for nt_t, budget in sorted({(1, 40.0), (min(8, ncpu), 60.0), (top, 30.0)}):
            cmd = [sys.executable, "-m", "oracle.torch_restatement", "--n-train", str(len(train)), "--n-dev", str(len(dev)),
                   "--R", str(args.R), "--B", str(args.batch), "--bn", str(int(not args.no_bn)), "--drpt", str(args.drpt),
                   "--threads", str(nt_t), "--budget", str(budget)]
            try:
                res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=budget + 120,
                                     env=dict(os.environ, OMP_NUM_THREADS=str(nt_t), MKL_NUM_THREADS=str(nt_t)))
                line = [l for l in res.stdout.splitlines() if l.startswith("TORCH_RESTATEMENT ")][-1]
                r = json.loads(line[len("TORCH_RESTATEMENT "):])
                runs.append({"threads": r["threads"], "s_per_epoch": r["s_per_epoch"], "cand_per_s": 1.0 / (r["s_per_epoch"] * args.epochs),
                             "train_steps_timed": r["steps"], "dev_rows_timed": r["dev_rows"], "full_epoch": r["full_epoch"]})
            except Exception as e:
                runs.append({"threads": nt_t, "error": repr(e)[:200]})
        good = [r for r in runs if "cand_per_s" in r]
        bestt = max(good, key=lambda r: r["cand_per_s"])
        out["torch_eager"] = {"value": bestt["cand_per_s"], "unit": "candidates/s", "cores": bestt["threads"], "kind": "port",
                              "runs": runs,
                              "sample": f"one epoch ({nb} train steps of B={args.batch} + {len(dev)} dev rows) of conf-4 R={args.R} in PyTorch-CPU "
                                        f"eager (restatement of the reference loop incl. its per-step optimizer state_dict round trip) on "
                                        f"same-shaped synthetic tables, per thread count (1 / {min(8, ncpu)} / {top}: runs[], time-boxed to 40 / 60 / 30 s, "
                                        f"runs[].full_epoch says whether the epoch completed), fastest reported, x E={args.epochs}; host has {ncpu} logical cores"}
    except Exception as e:   # the baseline is a report, never a reason to lose the bench line
        out["torch_eager"] = {"error": repr(e)}
    # headline = the PyTorch-CPU eager restatement at its FASTEST thread count (what BASELINE.md section 2 promises: the reference's own
    # execution model — eager per-op dispatch, autograd, torch.optim.Adam — on this box's host cores); the numpy port is nested
    te = out.pop("torch_eager")
    if "value" in te:
        te["numpy_port"] = out
        return te
    out["torch_eager"] = te
    return out


def synth_mm_tables(n_train, n_dev, device, C=23):
    """MM-IMDB-shaped synthetic multi-label tables (BASELINE configs[4]): text taps 64 / 128 (MaxOut_MLP o1 / o3,
    models/central/mm_imdb.py:176-196), image taps 4 x 512 (GP_VGG, :19-59), 23 genres, fp16 taps, multi-hot targets."""
    import mfas_amd as M
    from mfas_amd import mmimdb_searchable as MM
    g = torch.Generator(device=device)
    g.manual_seed(321)
    mus = {f"s{j}": torch.randn(C, w, generator=g, device=device) for j, w in enumerate(MM.MM_TEXT_SIZES)}
    mus.update({f"v{j}": torch.randn(C, w, generator=g, device=device) for j, w in enumerate(MM.MM_IMAGE_SIZES)})
    out = []
    for n, seed in ((n_train, 1), (n_dev, 2)):
        g.manual_seed(seed)
        z = (torch.rand(n, C, generator=g, device=device) < 0.15).float()
        taps = {k: torch.relu(0.6 * z @ mu + torch.randn(n, mu.shape[1], generator=g, device=device)).to(torch.float16)
                for k, mu in mus.items()}
        out.append(M.FeatureTable(taps, torch.zeros(n, dtype=torch.int32, device=device), multilabel=z))
    return out


def alg_bytes_per_candidate(conf, R, C, bn, s_sizes, v_sizes, B, E, n_train, n_dev, elt):
    """E * [ceil(N_tr/B) * 24 P + (N_tr + N_dev) * (sum F * s_f + 8) + 4 P] — the HBM-roofline bytes of ONE candidate."""
    conf = np.asarray(conf).reshape(-1, 3)
    P, F = 0, 0
    for i, (s, v, _) in enumerate(conf):
        f = s_sizes[int(s)] + v_sizes[int(v)]
        F += f
        P += R * (f + (R if i else 0)) + R + (2 * R if bn else 0)
    P += R * C + C
    nb = -(-n_train // B)
    return E * (nb * 24.0 * P + (n_train + n_dev) * (F * elt + 8.0) + 4.0 * P)


def self_spawn(n):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: start the N ranks ourselves (one process per GPU under
    torch.distributed.run, rendezvous on 127.0.0.1 at a free port).  Rank 0's JSON line is the only thing on stdout; the exit code
    is torchrun's (non-zero when any rank fails)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    # (before anything touches the HIP runtime: the host driver only supports dmabuf IPC, RCCL across processes needs this)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pop", type=int, default=0, help="candidates per GPU per step (default: 128; 512 for --workload c5)")
    ap.add_argument("--R", type=int, default=128)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--epochs", type=int, default=10)
    ap.add_argument("--drpt", type=float, default=0.5)
    ap.add_argument("--n-train", type=int, default=0, help="default 10,000 (c5: 15,552)")
    ap.add_argument("--n-dev", type=int, default=0, help="default 5,600 (c5: 2,608)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "f16"])
    ap.add_argument("--chunk-cols", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="nccl = RCCL over xGMI (default); gloo only to exercise the N>1 path on a 1-GPU box")
    ap.add_argument("--no-profile", action="store_true", help="no HIP-event sampling of the dominant kernel (roofline fields become null)")
    ap.add_argument("--no-bn", action="store_true", help="search-script defaults: no BatchNorm")
    ap.add_argument("--mixed-confs", action="store_true", help="population of sampled L=1..4 confs instead of conf 4")
    ap.add_argument("--total-pop", type=int, default=0, help="strong scaling: this many candidates IN TOTAL, sharded over the ranks")
    ap.add_argument("--workload", default="c1", choices=["c1", "c2", "c3", "c5"], help="named BASELINE workloads (see the module docstring)")
    ap.add_argument("--no-strong", action="store_true", help="N > 1: skip the strong-scaling workloads (config.strong) reported next to the weak headline")
    ap.add_argument("--no-search", action="store_true", help="N = 1: skip the end-to-end configs[3] search schedule (config.search_c3, ~10 s outside the timed region)")
    ap.add_argument("--no-small-pop", action="store_true", help="N = 1: skip the search-sized workloads (config.small_pop) reported next to the headline")
    ap.add_argument("--snr", type=float, default=0.12, help="planted-signal strength of the synthetic taps (BASELINE.md section 2: 0.12)")
    ap.add_argument("--engine-init", default="torch", choices=["torch", "device"],
                    help="candidate initialisation: torch = train_sampled_models' default (the module's own draws from torch's CPU generator, "
                         "uploaded); device = the engine's hash generator on the GPU")
    ap.add_argument("--engine-order", default="per_candidate", choices=["shared", "per_candidate"],
                    help="sample order: the reference's per-candidate shuffles (train_sampled_models' default), or one shuffle per epoch shared by the call's candidates")
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_spawn(a.gpus))
    if a.workload in ("c2", "c3"):      # search-script defaults (main_searchable_ntu.py:26-47)
        a.R, a.batch, a.no_bn = 16, 20, True
        a.total_pop = a.total_pop or (16 if a.workload == "c2" else 50)
    if a.workload == "c5":
        a.R, a.batch, a.no_bn, a.dtype = 16, 20, True, "f16"
        a.n_train, a.n_dev = a.n_train or 15552, a.n_dev or 2608
        a.pop = a.pop or 512
    a.pop = a.pop or 128
    a.n_train, a.n_dev = a.n_train or 10000, a.n_dev or 5600

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:     # the launcher's world is what runs; say so instead of dying on the first line
        if rank == 0:
            print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: running {world} rank(s)", file=sys.stderr)
        a.gpus = world
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a HIP device (torch.cuda.device_count() == 0)")
    # N > 1: gloo and RCCL print connection banners to STDOUT; the contract is ONE line there.  Everything this process (and the
    # libraries under it) writes to fd 1 goes to stderr instead, and rank 0 writes the JSON line to the original stdout at the end.
    out_fd = None
    if world > 1:
        sys.stdout.flush()
        out_fd = os.dup(1)
        os.dup2(2, 1)
    force_fail = bool(os.environ.get("MFAS_TEST_RCCL_FAIL"))     # test hook: exercise the fallback below on a 1-GPU box
    if world > ndev and a.backend == "nccl" and not force_fail:
        raise SystemExit(f"bench.py: {world} ranks over RCCL need {world} GPUs, {ndev} visible (one process per GPU); "
                         "--backend gloo shares GPUs between ranks (tests only)")
    local = local % ndev
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    group, backend_note = None, None
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # The default group is gloo (control plane: barriers, the agreement below); the population's collectives — one gather of
        # a few hundred bytes per call, two small broadcasts — go over an RCCL group created next to it and probed with one
        # all_reduce.  If RCCL cannot start on this node (every rank agrees on that over gloo) the run still measures: the data
        # path has no collective, so the gather falls back to gloo and the line says so in `backend` / `backend_note`.
        gloo_ok = True
        try:
            dist.init_process_group("gloo")
        except Exception as e:          # noqa: BLE001 - no gloo control plane: RCCL alone, as the default group
            if a.backend != "nccl":
                raise
            gloo_ok = False
            backend_note = f"gloo control plane unavailable ({type(e).__name__}); RCCL is the default group"
            dist.init_process_group("nccl", device_id=device)
        if a.backend == "nccl" and gloo_ok:
            # two agreements over gloo: first that EVERY rank created the RCCL group (a rank whose new_group raised must not leave the
            # others blocked in the probe), then that the probe all_reduce came back right everywhere
            def agree(ok_here):
                f = torch.tensor([1 if ok_here else 0], dtype=torch.int32)
                dist.all_reduce(f, op=dist.ReduceOp.MIN)
                return int(f.item()) == 1
            ok, err = 1, ""
            try:
                if force_fail and os.environ["MFAS_TEST_RCCL_FAIL"] != "real":     # ("real": let RCCL itself refuse the shared GPU)
                    raise RuntimeError("forced by MFAS_TEST_RCCL_FAIL")
                group = dist.new_group(backend="nccl")
            except Exception as e:      # noqa: BLE001 - whatever RCCL raises, the fallback is the same
                ok, err = 0, f"{type(e).__name__}: {str(e)[:200]}"
            if agree(ok == 1):
                try:
                    probe = torch.ones(1, device=device)
                    dist.all_reduce(probe, group=group)
                    torch.cuda.synchronize()
                    if int(probe.item()) != world:
                        raise RuntimeError(f"RCCL probe all_reduce returned {probe.item()} for {world} ranks")
                except Exception as e:      # noqa: BLE001
                    ok, err = 0, f"{type(e).__name__}: {str(e)[:200]}"
                flag = torch.tensor([1 if agree(ok == 1) else 0], dtype=torch.int32)
            else:
                flag = torch.tensor([0], dtype=torch.int32)
            if int(flag.item()) == 1:
                from mfas_amd import population as _pm
                _pm.set_group(group)
            else:
                group = None
                a.backend = "gloo"
                backend_note = "RCCL could not start on this node; accuracies gathered over gloo" + (f" ({err})" if err else " (another rank failed)")
                if rank == 0:
                    print("bench.py: " + backend_note, file=sys.stderr)

    import mfas_amd as M
    from mfas_amd import ntu_searchable as NS
    from mfas_amd import mmimdb_searchable as MM
    from mfas_amd import population as popmod

    dtype = {"bf16": torch.bfloat16, "f32": torch.float32, "f16": torch.float16}[a.dtype]
    elt = 4 if a.dtype == "f32" else 2
    mm = a.workload == "c5"
    if mm:
        train, dev = synth_mm_tables(a.n_train, a.n_dev, device)
        s_sizes, v_sizes, C = MM.MM_TEXT_SIZES, MM.MM_IMAGE_SIZES, MM.MM_NUM_OUTPUTS
        train_fn, stype = MM.train_sampled_models, MM.Searchable_Text_Image_Net
    else:
        train, dev = synth_tables(a.n_train, a.n_dev, device, dtype, snr=a.snr)
        s_sizes, v_sizes, C = M.engine.S_SIZES, M.engine.V_SIZES, 60
        train_fn, stype = M.train_sampled_models, M.Searchable_Skeleton_Image_Net
    loaders = {"train": M.FeatureLoader(train, a.batch, shuffle=True), "dev": M.FeatureLoader(dev, a.batch, shuffle=False)}
    args = SimpleNamespace(vid_len=(8, 32), num_outputs=C, drpt=a.drpt, inner_representation_size=a.R,
                           batchnorm=not a.no_bn, alphas=False, multitask=False, weightsharing=False, batchsize=a.batch,
                           eta_max=1e-3, eta_min=1e-6, Ti=1, Tm=2, use_dataparallel=False, verbose=False,
                           epochs=a.epochs, engine_init=a.engine_init, engine_profile=not a.no_profile,
                           engine_chunk_cols=a.chunk_cols, engine_order=a.engine_order)
    total = a.total_pop if a.total_pop > 0 else a.pop * world
    confs = [np.array(CONF4) for _ in range(total)]
    if a.mixed_confs:
        rng = np.random.default_rng(0)
        confs = [np.stack([rng.integers(0, 4, L), rng.integers(0, 4, L), rng.integers(0, 2, L)], 1)
                 for L in rng.integers(1, 5, total)]

    def sampled_l4(n, layer=None):      # L=4 configurations sampled like the controller does at progression level 3
        np.random.seed(0)
        layer = layer or NS.get_possible_layer_configurations(0)
        return [np.array([layer[i] for i in np.random.choice(len(layer), 4)]) for _ in range(n)]

    if a.workload in ("c2", "c3"):
        confs = sampled_l4(total)
    if mm:
        confs = sampled_l4(total, MM.get_possible_layer_configurations(0))
    torch.manual_seed(0)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def rank_times(dt_local):
        if world == 1:
            return [dt_local]
        t = torch.tensor([dt_local], dtype=torch.float64, device=device if dist.get_backend(group) == "nccl" else "cpu")
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t, group=group)
        return [float(x.item()) for x in allt]

    def profile_summary():
        """(launches, ms, algorithmic bytes per launch, schedule of the last population) over the calls since PROFILE.clear()"""
        n_launch = sum(p[0] for p in NS.PROFILE)
        ms = sum(p[1] for p in NS.PROFILE)
        by = sum(p[0] * p[2] for p in NS.PROFILE) / n_launch if n_launch else 0.0
        return n_launch, ms, by, (NS.PROFILE[-1][3] if NS.PROFILE else {})

    accs = None
    for _ in range(a.warmup):
        accs = train_fn(confs, stype, loaders, args, device)
    NS.PROFILE.clear()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        accs = train_fn(confs, stype, loaders, args, device)
    barrier()
    dt = time.perf_counter() - t0
    rank_dt = rank_times(dt)      # per-rank seconds around the same K timed steps
    dt = max(rank_dt)
    n_launch, ms, bytes_per_launch, sched = profile_summary()
    pops_per_step = len(NS.PROFILE) // max(a.steps, 1)       # populations one timed call trained (resident rounds: several)

    def timed_calls(fn, reps):
        fn()                      # warm-up (layout decisions, first-touch of the population's pages, calibration of the sharder)
        NS.PROFILE.clear()
        barrier()
        ts = time.perf_counter()
        out = None
        for _ in range(reps):
            out = fn()
        barrier()
        return out, rank_times(time.perf_counter() - ts)

    def search_sized(name, K, reps=3):
        """ONE train_sampled_models call on K sampled L=4 confs at the search script's defaults (models/searchable.py:90,120;
        main_searchable_ntu.py:26-47: R=16, B=20, no batchnorm, drpt 0.5) — BASELINE configs[2] (K=16) / configs[3] (K=50)."""
        sargs = SimpleNamespace(**vars(args))
        sargs.inner_representation_size, sargs.batchsize, sargs.batchnorm, sargs.engine_profile = 16, 20, False, not a.no_profile
        sloaders = {"train": M.FeatureLoader(train, 20, shuffle=True), "dev": M.FeatureLoader(dev, 20, shuffle=False)}
        sconfs = sampled_l4(K)
        saccs, rt = timed_calls(lambda: M.train_sampled_models(sconfs, stype, sloaders, sargs, device), reps)
        nl, pms, pby, psched = profile_summary()
        nb = -(-a.n_train // 20)
        gb = float(np.mean([alg_bytes_per_candidate(c, 16, 60, False, s_sizes, v_sizes, 20, a.epochs, a.n_train, a.n_dev, elt) for c in sconfs]))
        cps = K * reps / max(rt)
        ent = {"workload": f"BASELINE configs[{2 if K == 16 else 3}]: one call of {K} sampled L=4 confs, R=16, no batchnorm, "
                           f"drpt {a.drpt}, B=20, E={a.epochs}, N_train={a.n_train}, N_dev={a.n_dev}",
               "scaling": "strong", "candidates": K, "calls_timed": reps, "cand_per_s": cps, "ms_per_call": max(rt) / reps * 1e3,
               "rank_seconds": rt, "mean_best_dev_acc": float(np.mean(saccs)),
               "hbm_bound_cand_per_s_per_gpu": HBM_PEAK_GBS * 1e9 / gb, "frac_of_hbm_bound": cps * gb / (HBM_PEAK_GBS * 1e9 * world)}
        if world == 1:
            ent.update({"us_per_train_step_incl_dev_eval": max(rt) / reps / (a.epochs * nb) * 1e6, "schedule": psched,
                        "populations_per_call": len(NS.PROFILE) // max(reps, 1),
                        "kernel_us_per_train_step": (pms * 1e3 / (reps * a.epochs * nb)) if (nl and psched.get("persistent")) else None})
        else:
            shp = M.Hyper.from_args(sargs)          # the geometry key train_sampled_models calibrated its model under
            shp.multitask, shp.tap_bits, shp.order_per_candidate = False, 8 * train.elem_size(), a.engine_order == "per_candidate"
            owner, _, model = popmod.shard_call(sconfs, shp, world, device, False)
            ent.update({"share": [owner.count(r) for r in range(world)], "ranks_used": len(set(owner)),
                        "step_time_model": model.describe() if model is not None else None})
        return ent

    small = strong = None
    plain = a.workload == "c1" and a.total_pop == 0 and not a.mixed_confs
    if plain and world > 1 and not a.no_strong:
        # Strong scaling next to the weak headline (outside the timed region): the search's own call sizes, sharded by the engine's
        # policy (mfas_amd/population.py: a call uses only as many ranks as its calibrated step-time model says pay)
        strong = {"c2": search_sized("c2", 16), "c3": search_sized("c3", 50)}
    if plain and world == 1 and not a.no_small_pop and a.R == 128 and not a.no_bn:
        small = {"c2": search_sized("c2", 16), "c3": search_sized("c3", 50)}
        # steps/s of the single found architecture — ONE conf-4 candidate (R=128, batchnorm, B=16), and the
        # share an 8-GPU search call leaves a rank at this R (6 candidates)
        # (round 6: and 12 — the smallest population of the two-group launches, whose chain blocks are chain_split parts too)
        for name, K in (("c1_single", 1), ("c1_pop6", 6), ("c1_pop12", 12)):
            kargs = SimpleNamespace(**vars(args))
            kargs.engine_profile = not a.no_profile
            kconfs = [np.array(CONF4)] * K
            kaccs, rt = timed_calls(lambda: M.train_sampled_models(kconfs, stype, loaders, kargs, device), 2)
            nl, pms, pby, psched = profile_summary()
            nb = -(-a.n_train // a.batch)
            gb = alg_bytes_per_candidate(CONF4, a.R, 60, True, s_sizes, v_sizes, a.batch, a.epochs, a.n_train, a.n_dev, elt)
            cps = K * 2 / max(rt)
            small[name] = {"workload": f"{K} x NTU found conf 4, R={a.R}, batchnorm, drpt {a.drpt}, B={a.batch}, E={a.epochs}, one call",
                           "candidates": K, "cand_per_s": cps, "ms_per_call": max(rt) / 2 * 1e3,
                           "train_steps_per_s": a.epochs * nb * 2 / max(rt),
                           "us_per_train_step_incl_dev_eval": max(rt) / 2 / (a.epochs * nb) * 1e6,
                           "kernel_us_per_train_step": (pms / nl * 1e3) if nl else None,
                           "kernel_algorithmic_gbs": (pby / (pms / nl * 1e-3) / 1e9) if nl else None,
                           "schedule": psched, "frac_of_hbm_bound": cps * gb / (HBM_PEAK_GBS * 1e9),
                           "mean_best_dev_acc": float(np.mean(kaccs))}
    search_c3 = None
    if plain and world == 1 and not a.no_small_pop and a.R == 128 and not a.no_bn and not a.no_search:
        # BASELINE configs[3] END TO END on this one GPU (outside the timed region, ~10 s): the reference's full EPNAS schedule at its own
        # search flags (main_searchable_ntu.py: --num_samples 50 --search_iterations 5 --max_fusions 4, R=16, B=20, drpt 0.5, no
        # batchnorm, E epochs per candidate) — 20 train_sampled_models calls, 982 candidates — with the seeded controller (surrogate on
        # the CPU, decision for decision the reference's: golden G8 / G9), split into candidate training and controller time
        import main_searchable_ntu as MS
        from mfas_amd.search import NTUSearcher
        from mfas_amd.search.searcher import timed_search
        sa = MS.parse_args(["--num_samples", "50", "--search_iterations", "5", "--max_fusions", "4", "--epochs", str(a.epochs),
                            "--no-verbose", "--drpt", str(a.drpt), "--engine_init", a.engine_init, "--engine_order", a.engine_order])
        nthr = torch.get_num_threads()
        torch.set_num_threads(max(1, sa.controller_threads))
        try:
            _, search_c3 = timed_search(NTUSearcher(sa, device, {"train": train, "dev": dev}), seed=0)
        except Exception as e:      # noqa: BLE001 — an auxiliary report is never a reason to lose the bench line (like cpu_baseline)
            # ValueError: the reference's sampler (models/search/tools.py:47-56: np.random.choice(..., replace=False, p=p)) raises when fewer than
            # num_samples configurations have a non-zero predicted accuracy — what a search over a few hundred samples and one epoch
            # (the test suite's tiny tables) runs into; anything else (an engine error, a persistent-schedule timeout, out of memory)
            # is reported the same way: the schedule is marked as not completed, the line stays valid
            search_c3 = {"error": f"{type(e).__name__}: {e}", "seed": 0}
        finally:
            torch.set_num_threads(nthr)
        search_c3["workload"] = (f"BASELINE configs[3] end to end on 1 GPU: full EPNAS schedule (50 confs/iter x 5 surrogate iterations x 4 "
                                 f"progression levels, R=16, B=20, no batchnorm, drpt {a.drpt}, E={a.epochs}, N_train={a.n_train}, N_dev={a.n_dev}), "
                                 f"seeded controller (seed 0), surrogate on {sa.controller_threads} CPU threads")
    other_init = other_order = other_both = None
    if world == 1 and not a.no_small_pop and not a.mixed_confs:      # the same workload with the OTHER initialisation path / sample order, one call each
        oargs = SimpleNamespace(**vars(args))
        oargs.engine_init = "device" if a.engine_init == "torch" else "torch"
        _, rt = timed_calls(lambda: train_fn(confs, stype, loaders, oargs, device), 1)
        other_init = {"engine_init": oargs.engine_init, "cand_per_s": total / max(rt), "ms_per_step": max(rt) * 1e3}
        oargs = SimpleNamespace(**vars(args))
        oargs.engine_order = "shared" if a.engine_order == "per_candidate" else "per_candidate"
        _, rt = timed_calls(lambda: train_fn(confs, stype, loaders, oargs, device), 1)
        nl, pms, _, _ = profile_summary()
        other_order = {"engine_order": oargs.engine_order, "cand_per_s": total / max(rt), "ms_per_step": max(rt) * 1e3,
                       "avg_launch_us": (pms / nl * 1e3) if nl else None}
        # both switched = what rounds 1-3 timed (device hash init + one shared sample order): the like-for-like line across rounds
        oargs = SimpleNamespace(**vars(args))
        oargs.engine_init = "device" if a.engine_init == "torch" else "torch"
        oargs.engine_order = "shared" if a.engine_order == "per_candidate" else "per_candidate"
        _, rt = timed_calls(lambda: train_fn(confs, stype, loaders, oargs, device), 1)
        nl, pms, _, _ = profile_summary()
        other_both = {"engine_init": oargs.engine_init, "engine_order": oargs.engine_order, "cand_per_s": total / max(rt),
                      "ms_per_step": max(rt) * 1e3, "avg_launch_us": (pms / nl * 1e3) if nl else None}

    if rank == 0:
        # `achieved` = algorithmic bytes of the update+forward launches of ONE timed step / that step's WALL time (round 6, VERDICT item 2:
        # the HIP-event bracket around every 16th launch over-counts by ~0.5 %, so launches x its average could exceed ms_per_step; the
        # wall-clock figure is a lower bound of the kernel's rate — the step also holds the dev passes, the forward-only prologues and
        # the host's per-call work).  The event-based rate stays in the line as `achieved_hip_events`.
        nb_tr = -(-a.n_train // a.batch)
        launches_per_step = (a.epochs if sched.get("persistent") else a.epochs * nb_tr * (2 if sched.get("groups", 1) == 2 else 1)) * max(1, pops_per_step)
        achieved_ev = bytes_per_launch / (ms / n_launch * 1e-3) / 1e9 if n_launch else None
        achieved = (bytes_per_launch * launches_per_step / (dt / a.steps) / 1e9) if (n_launch and bytes_per_launch) else None
        if sched.get("persistent"):
            kernel = ("k_president (ONE launch per epoch: per-candidate chain workgroups + "
                      + f"{sched['resident_units']} feature units resident in registers on {sched['resident_workgroups']} workgroups"
                      + "; algorithmic bytes = what one epoch of a streaming schedule moves (24 B/param/step + taps): resident W/m/v never touch HBM, "
                        "so `achieved` is a nominal rate for comparison, the loop itself is latency-bound, DESIGN.md 4a)")
        else:
            kernel = ("k_step (chain blocks of one group + sweep blocks of the other: dW, Adam, next-step forward)" if sched.get("groups", 2) > 1
                      else "k_step_same (chain blocks + sweep blocks of the same candidates, units released per cell as the backward pass publishes dy)" if sched.get("groups") == -1
                      else "k_step (sweep of the whole population: dW, Adam, next-step forward; the chain runs in its own k_chain launch)")
        total_trained = total * a.steps
        # HBM traffic / MfmaUtil need rocprofv3 --pmc passes (separate runs, MI355X_MICROARCH.md): they are NOT measured in this
        # process.  When the committed passes of exactly this workload exist they are quoted WITH their source; else null.
        traffic = mfma = traffic_src = mfma_src = None
        headline = plain and a.pop == 128 and a.R == 128 and not a.no_bn and world == 1
        prof_avg_us = prof_src = None
        for tag in ("r06", "r05", "r04", "r03", "r02", "r01"):      # the committed rocprofv3 --kernel-trace --stats average of the headline kernel (builder's box)
            cp = os.path.join(ROOT, "profiles", f"{tag}_bench_pop128_kernel_stats.csv")
            if headline and prof_avg_us is None and os.path.exists(cp):
                try:
                    import csv
                    rows = [r for r in csv.DictReader(open(cp)) if r.get("Name", "").startswith("void k_step<") or r.get("Name", "").startswith("k_step<")]
                    top = max(rows, key=lambda r: float(r["TotalDurationNs"]))
                    prof_avg_us = float(top["AverageNs"]) / 1e3
                    prof_src = f"profiles/{tag}_bench_pop128_kernel_stats.csv ({top['Name'][:48]}, {top['Calls']} calls; rocprofv3 --kernel-trace --stats of this command on the builder's box)"
                except Exception:
                    prof_avg_us = None
        for tag in ("r06", "r05", "r04", "r03", "r02", "r01"):
            tp = os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.json")
            if headline and traffic is None and os.path.exists(tp):
                try:
                    traffic = json.load(open(tp))["per_launch"]["total_bytes"]
                    traffic_src = f"profiles/{tag}_pmc_traffic.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; not measured in this run)"
                except Exception:
                    traffic = None
            mp = os.path.join(ROOT, "profiles", f"{tag}_pmc_mfma.json")
            if headline and mfma is None and os.path.exists(mp):
                try:
                    mfma = json.load(open(mp))["k_step"]["MfmaUtil"]["mean"]
                    mfma_src = f"profiles/{tag}_pmc_mfma.json (separate rocprofv3 --pmc pass; not measured in this run)"
                except Exception:
                    mfma = None
        # live rate of THIS box for a compute-free 3-plane read-modify-write probe (k_stream_probe), same units as `achieved`: a
        # reference point for box-to-box comparisons, NOT a ceiling — since the packed Adam the sweep itself is faster than the probe
        stream = None
        if world == 1:
            import ctypes
            from mfas_amd import _lib
            gb = ctypes.c_double(0)
            with torch.cuda.device(device):
                if _lib.lib().mfas_stream_probe(400 << 20, 10, ctypes.byref(gb)) == 0:
                    stream = gb.value
        bytes_cand = float(np.mean([alg_bytes_per_candidate(c, a.R, C, not a.no_bn, s_sizes, v_sizes, a.batch, a.epochs, a.n_train, a.n_dev, elt)
                                    for c in confs[:64]]))
        wl = {"c1": 1, "c2": 2, "c3": 3, "c5": 4}[a.workload]
        what = ("NTU found conf 4" if a.workload == "c1" and not a.mixed_confs else "sampled L=1..4 confs" if a.workload == "c1"
                else f"{total} sampled L=4 MM-IMDB-shaped confs (text taps 64/128, image taps 4x512, 23 genres, weighted BCE + F1-samples)" if mm
                else f"{total} sampled L=4 confs (np.random.seed(0))")
        line = {
            "metric": "candidate-archs trained/sec (NTU inner loop)", "value": total_trained / dt, "unit": "candidates/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if a.total_pop > 0 else "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": (f"BASELINE configs[{wl}]: {what}, R={a.R}, {'no batchnorm' if a.no_bn else 'batchnorm'}, drpt {a.drpt}, "
                                    f"B={a.batch}, E={a.epochs}, N_train={a.n_train}, N_dev={a.n_dev}, "
                                    + ("" if mm else f"snr {a.snr}, ") + f"{a.dtype} precomputed taps, f32 state/compute"),
                       "candidates_total_per_step": total,
                       "candidates_per_gpu_per_step": (a.pop if a.total_pop == 0 else f"{total // world}..{-(-total // world)}"),
                       "parallelism": f"population-sharded x{world}",
                       "rccl_ranks": (dist.get_world_size() if world > 1 else 1), "backend": (a.backend if world > 1 else None), "backend_note": backend_note,
                       "rank_seconds": rank_dt,
                       "engine_init": a.engine_init, "engine_order": a.engine_order, "other_init": other_init, "other_order": other_order, "other_both": other_both,
                       "mean_best_dev_acc" if not mm else "mean_best_dev_f1": float(np.mean(accs)),
                       "hbm_bound_cand_per_s_per_gpu": HBM_PEAK_GBS * 1e9 / bytes_cand,
                       "frac_of_hbm_bound": total_trained / dt * bytes_cand / (HBM_PEAK_GBS * 1e9 * world),
                       "small_pop": small, "strong": strong, "search_c3": search_c3},
            "roofline": {"bound": "hbm", "kernel": kernel, "schedule": sched,
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic, "traffic_source": traffic_src,
                         "achieved_hip_events": achieved_ev, "launches_per_step": launches_per_step,
                         "wall_us_per_launch": (dt / a.steps / launches_per_step * 1e6) if launches_per_step else None,
                         "launches": n_launch, "avg_launch_us": (ms / n_launch * 1e3) if n_launch else None,
                         "profile_box_avg_us": prof_avg_us, "profile_box_source": prof_src,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "stream_probe": stream, "frac_of_stream_probe": (achieved / stream) if (achieved and stream) else None,
                         "mfma_util_pct_from_profile": mfma, "mfma_util_source": mfma_src},
        }
        if world == 1 and not a.no_cpu_baseline and not mm:     # (the CPU restatements are NTU-shaped: the c5 line carries none)
            line["cpu_baseline"] = cpu_baseline(train, dev, a)
        if out_fd is None:
            print(json.dumps(line), flush=True)
        else:
            sys.stdout.flush()
            os.write(out_fd, (json.dumps(line) + "\n").encode())
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

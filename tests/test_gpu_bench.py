"""GPU tests of bench.py ITSELF (the driver's contract) and of the round-4 closures: the self-spawned N > 1 path, the search-sized
workloads in the N = 1 line, the MM-IMDB-shaped workload, the AddressSanitizer build variant, the calibrated sharder model and
mfas_population_backward on a handle that has trained before."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
from oracle import np_oracle as O
from tests.helpers import CONFS, engine_hyper

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TINY = ["--epochs", "1", "--n-train", "320", "--n-dev", "160", "--no-cpu-baseline"]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "needs a HIP device"
    return torch.device("cuda:0")


def run_bench(extra, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True, text=True, timeout=timeout)
    assert res.returncode == 0, (res.stdout[-2000:], res.stderr[-4000:])
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]            # exactly ONE JSON line, from rank 0
    if "--gpus" in extra:                                  # N > 1: and nothing else on stdout (gloo / RCCL banners go to stderr)
        assert res.stdout.strip() == lines[0], res.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("world", [2, 4, 8])
def test_bench_starts_its_own_ranks(dev, world):
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment — the way the driver invokes `--gpus 1` — starts two ranks
    itself (gloo here: both on this box's one GPU; nccl = RCCL on a multi-GPU node), prints one line, and that line carries the
    weak headline AND the strong-scaling workloads with per-rank seconds and shares."""
    import time
    pop = 3 if world < 8 else 1        # (8 ranks: what the driver's SCALE run starts on an 8-GPU node, here on one GPU)
    t0 = time.perf_counter()
    line = run_bench(["--gpus", str(world), "--backend", "gloo", "--pop", str(pop), "--steps", "1", "--warmup", "1"] + TINY)
    assert time.perf_counter() - t0 < 300.0
    assert line["n_gpus"] == world and line["config"]["rccl_ranks"] == world and line["config"]["backend"] == "gloo"
    assert line["scaling"] == "weak" and line["config"]["candidates_total_per_step"] == pop * world and line["value"] > 0
    assert len(line["config"]["rank_seconds"]) == world
    strong = line["config"]["strong"]
    for name, K in (("c2", 16), ("c3", 50)):
        s = strong[name]
        assert s["candidates"] == K and sum(s["share"]) == K and len(s["share"]) == len(s["rank_seconds"]) == world and s["cand_per_s"] > 0
        assert 1 <= s["ranks_used"] <= world and s["step_time_model"]["calibrated"] is True      # the sharder's model was measured on THIS box
        assert sum(1 for n in s["share"] if n > 0) == s["ranks_used"] and max(s["share"]) - min(n for n in s["share"] if n > 0) <= max(2, K // 4)
        assert len(s["step_time_model"]["resident_us"]) >= 2 and all(us > 1.0 for _, us in s["step_time_model"]["resident_us"])
    assert line["config"]["small_pop"] is None and "cpu_baseline" not in line


@pytest.mark.parametrize("mode", ["1", "real"])
def test_bench_measures_when_rccl_cannot_start(dev, mode):
    """The N > 1 line must not depend on RCCL coming up: the default group is gloo, the population's collectives go over an RCCL
    group that is probed with one all_reduce, and when the probe fails on any rank every rank agrees over gloo to gather there
    instead — the line says so.  mode "1": the failure is raised by the hook; mode "real": two ranks really create the RCCL
    communicator on this box's ONE GPU and RCCL itself refuses it (ncclInvalidUsage: duplicate GPU) — on a box with two GPUs the
    probe passes and the line stays on RCCL."""
    env_keep = os.environ.get("MFAS_TEST_RCCL_FAIL")
    os.environ["MFAS_TEST_RCCL_FAIL"] = mode
    try:
        line = run_bench(["--gpus", "2", "--backend", "nccl", "--pop", "2", "--steps", "1", "--warmup", "0", "--no-small-pop"] + TINY, timeout=600)
    finally:
        if env_keep is None:
            os.environ.pop("MFAS_TEST_RCCL_FAIL", None)
        else:
            os.environ["MFAS_TEST_RCCL_FAIL"] = env_keep
    assert line["n_gpus"] == 2 and line["value"] > 0 and len(line["config"]["rank_seconds"]) == 2
    if mode == "real" and torch.cuda.device_count() >= 2:
        assert line["config"]["backend"] == "nccl" and line["config"]["backend_note"] is None
    else:
        assert line["config"]["backend"] == "gloo" and "RCCL could not start" in line["config"]["backend_note"]
        if mode == "real":
            assert "NCCL" in line["config"]["backend_note"] or "nccl" in line["config"]["backend_note"]


def test_bench_failing_rank_gives_nonzero_exit(dev):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(torch.cuda.device_count() + 1), "--backend", "nccl",
                          "--pop", "2", "--steps", "1", "--warmup", "0"] + TINY, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode != 0 and not [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert "GPUs" in res.stderr


def test_bench_single_gpu_line_carries_the_search_sized_workloads(dev):
    line = run_bench(["--pop", "4", "--steps", "1", "--warmup", "0"] + TINY)
    assert line["n_gpus"] == 1 and line["metric"].startswith("candidate-archs trained/sec") and line["unit"] == "candidates/s"
    assert line["config"]["engine_init"] == "torch" and line["config"]["other_init"]["engine_init"] == "device"
    assert line["config"]["engine_order"] == "per_candidate" and line["config"]["other_order"]["engine_order"] == "shared"     # the reference's shuffles are the default
    ob = line["config"]["other_both"]                      # rounds 1-3's configuration, for the like-for-like comparison across rounds
    assert ob["engine_init"] == "device" and ob["engine_order"] == "shared" and ob["cand_per_s"] > 0
    rf = line["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and rf["achieved"] > 0 and abs(rf["frac"] - rf["achieved"] / 8000.0) < 1e-9
    sp = line["config"]["small_pop"]
    assert set(sp) == {"c2", "c3", "c1_single", "c1_pop6", "c1_pop12"}
    assert sp["c2"]["candidates"] == 16 and sp["c2"]["schedule"]["persistent"] == 1 and sp["c2"]["populations_per_call"] == 1
    assert sp["c3"]["candidates"] == 50 and sp["c3"]["populations_per_call"] == 2          # two resident rounds
    for k in ("c2", "c3"):
        assert sp[k]["cand_per_s"] > 0 and sp[k]["us_per_train_step_incl_dev_eval"] > 0 and 0 < sp[k]["frac_of_hbm_bound"] < 1.5
    one = sp["c1_single"]
    assert one["candidates"] == 1 and one["train_steps_per_s"] > 0 and one["kernel_us_per_train_step"] > 0 and one["schedule"]["groups"] == -1
    assert sp["c1_pop6"]["candidates"] == 6 and sp["c1_pop12"]["candidates"] == 12 and sp["c1_pop12"]["schedule"]["groups"] == 2


def test_bench_mmimdb_shaped_workload(dev):
    line = run_bench(["--workload", "c5", "--pop", "24", "--steps", "1", "--warmup", "1", "--epochs", "1", "--n-train", "400", "--n-dev", "200",
                      "--no-cpu-baseline"])
    assert "configs[4]" in line["config"]["workload"] and "MM-IMDB" in line["config"]["workload"]
    assert line["config"]["candidates_total_per_step"] == 24 and 0.0 <= line["config"]["mean_best_dev_f1"] <= 1.0
    assert line["roofline"]["achieved"] > 0 and line["config"]["hbm_bound_cand_per_s_per_gpu"] > 100


def test_asan_build_trains_a_population(dev):
    """The MFAS_ASAN=1 build variant (host-side AddressSanitizer of the C-ABI library) trains resident and launch-per-phase
    populations, packs / unpacks parameters and runs the train-mode forward + backward without a report."""
    import __graft_entry__ as ge
    ge.build_asan()
    code = r"""
import sys
sys.path.insert(0, %r)
import numpy as np, torch
import mfas_amd as M
from mfas_amd import _lib
from oracle import np_oracle as O
assert _lib.LIB_PATH.endswith("libmfas_hip_asan.so")
dev = torch.device("cuda:0")
tr = M.FeatureTable.synthetic(200, 1, dev, torch.bfloat16, snr=0.3)
dv = M.FeatureTable.synthetic(96, 2, dev, torch.bfloat16, snr=0.3)
rng = np.random.default_rng(1)
for R, B, bn, K in ((16, 20, False, 5), (128, 16, True, 3), (32, 40, True, 2)):
    hp = M.Hyper(R=R, B=B, bn=bn, drpt=0.5, tap_bits=16)
    confs = [np.stack([rng.integers(0, 4, L), rng.integers(0, 4, L), rng.integers(0, 3, L)], 1) for L in rng.integers(1, 5, K)]
    pop = M.Population(hp, confs, dev, drop_seeds=list(range(K)))
    pop.init(list(range(1, K + 1)))
    nb = -(-200 // B)
    stats, status = pop.train(tr, dv, 2, O.eta_sequence(1e-3, 1e-6, 1, 2, 200 / B, 2 * nb), snapshot_best=True)
    assert not status.any() and np.isfinite(stats["train_loss_sum"]).all()
    flat = pop.get_params(0)
    pop.set_params(0, flat)
    out = pop.forward_train(0, tr, 0, min(B, 16), step=1)
    g = pop.backward(0, tr, torch.ones_like(out) * 0.01, 0, min(B, 16), step=1)
    assert torch.isfinite(g).all() and torch.isfinite(out).all()
    pop.close()
print("ASAN-TRAIN-OK")
""" % ROOT
    res = subprocess.run([sys.executable, "-c", code], env=ge.asan_env(), capture_output=True, text=True, timeout=900)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", "asan_train_stderr.log"), "w").write(res.stderr)
    assert res.returncode == 0 and "ASAN-TRAIN-OK" in res.stdout, (res.stdout[-1500:], res.stderr[-4000:])
    assert "ERROR: AddressSanitizer" not in res.stderr, res.stderr[-4000:]


def test_backward_on_a_handle_that_has_trained(dev):
    """mfas_population_backward leaves the exact gradient in the first-moment slot whatever the handle did before: after training
    (non-zero Adam moments) it equals the gradient a fresh population gives for the same parameters."""
    import mfas_amd as M
    conf = np.array(CONFS["l3"])
    for R, B, bn in ((16, 20, False), (128, 16, True)):
        ohp = O.Hyper(R=R, B=B, bn=bn, drpt=0.5)
        t = O.synth_table(4 * B, 91, snr=0.4)
        tab = M.FeatureTable.from_numpy(t, dev, torch.float32)
        pop = M.Population(engine_hyper(ohp), [conf], dev, drop_seeds=[5])
        pop.set_state_dict(0, O.init_params(conf, ohp, 17, perturb_bn=True))
        pop.train(tab, None, 1, O.eta_sequence(1e-3, 1e-6, 1, 2, 4.0, 4), max_steps=4)
        assert float(pop.get_params(0, plane=1).abs().max()) > 0           # the moments are live
        flat = pop.get_params(0).clone()
        rng = np.random.default_rng(5)
        dlog = torch.from_numpy((rng.standard_normal((B, 60)) * 1e-3).astype(np.float32)).to(dev)    # small: would cancel against a stale m
        got = pop.backward(0, tab, dlog, 0, B, step=9).clone()
        after = pop.get_params(0)
        fresh = M.Population(engine_hyper(ohp), [conf], dev, drop_seeds=[5])
        fresh.set_params(0, flat)
        want = fresh.backward(0, tab, dlog, 0, B, step=9)
        assert torch.equal(got, want) and torch.isfinite(got).all() and float(got.abs().max()) > 0
        layout, _ = M.engine.flat_layout(conf, engine_hyper(ohp))
        keep = torch.ones_like(flat, dtype=torch.bool)
        for key, shape, off in layout:
            if "running" in key:
                keep[off:off + int(np.prod(shape))] = False
        assert torch.equal(after[keep], flat[keep])
        pop.close(); fresh.close()


def test_train_mode_forward_holds_no_population(dev):
    """_TrainModeForward keeps no GPU population between forward and backward: a no_grad forward and a never-backpropagated
    one leave nothing behind; taps that require grad are refused."""
    import gc
    import mfas_amd as M
    from types import SimpleNamespace
    args = SimpleNamespace(vid_len=(8, 32), num_outputs=60, drpt=0.5, inner_representation_size=16, batchnorm=True, alphas=False,
                           multitask=False, batchsize=16)
    model = M.Searchable_Skeleton_Image_Net(args, np.array(CONFS["l2"]))
    model.train(True)
    t = O.synth_table(16, 3, snr=0.4)
    x = {k: torch.from_numpy(v).to(dev) for k, v in t.items() if k != "label"}
    rgb, ske = {k: v for k, v in x.items() if k[0] == "v"}, {k: v for k, v in x.items() if k[0] == "s"}
    torch.cuda.synchronize()
    base = torch.cuda.memory_allocated()
    with torch.no_grad():
        out = model((rgb, ske))
    assert not out.requires_grad
    out2 = model((rgb, ske))
    assert out2.requires_grad and out2.grad_fn is not None
    ctx = out2.grad_fn
    assert not hasattr(ctx, "pop")
    out2.sum().backward()
    assert model.central_classifier.weight.grad is not None
    del out, out2, ctx
    gc.collect()
    assert torch.cuda.memory_allocated() <= base + (1 << 20)
    rgb_g = {k: v.clone().requires_grad_(True) for k, v in rgb.items()}
    with pytest.raises(NotImplementedError):
        model((rgb_g, ske))


def test_device_side_torch_streams_equal_the_module_draws(dev, monkeypatch):
    """train_sampled_models' default initialisation — the numbers `Searchable_Skeleton_Image_Net(args, conf)` draws from torch's CPU
    generator under torch.manual_seed(seed) (ntu_searchable.py:44) — generated on the GPU (at::mt19937 + uniform_real_distribution
    in k_mt_uniform, the alphas' normal draws from the stream's next raw words): bit for bit the host path's flat parameters for
    every candidate, depth, width and seed; and a whole call gives the same accuracies with either path."""
    import mfas_amd as M
    from mfas_amd import ntu_searchable as NS
    from types import SimpleNamespace
    rng = np.random.default_rng(11)
    for R, C, bn, alphas in ((16, 60, False, False), (128, 60, True, True), (24, 11, True, False), (200, 33, False, True)):
        args = SimpleNamespace(vid_len=(8, 32), num_outputs=C, drpt=0.5, inner_representation_size=R, batchnorm=bn, alphas=alphas,
                               multitask=False, batchsize=16)
        hp = M.Hyper.from_args(args)
        confs = [np.stack([rng.integers(0, 4, L), rng.integers(0, 4, L), rng.integers(0, 3, L)], 1) for L in (1, 4, 2, 3, 4, 4, 1)]
        seeds = [int(x) for x in rng.integers(0, 2 ** 31 - 1, len(confs))] 
        seeds[1] = 2 ** 32 + 5              # (at::mt19937 seeds from the low 32 bits)
        pop = M.Population(hp, confs, dev)
        pop.init_torch_streams(seeds, np.stack([NS.torch_init_bounds(c, hp) for c in confs]))
        for k, c in enumerate(confs):
            torch.manual_seed(seeds[k])
            want = NS.Searchable_Skeleton_Image_Net(args, c).flat_params()
            got = pop.get_params(k).cpu()
            assert torch.equal(got, want), (R, C, bn, alphas, k, int((got != want).sum()), got.numel())
            assert float(pop.get_params(k, plane=1).abs().max()) == 0.0          # a fresh optimizer
        pop.close()
    # through the driver: host-path call == device-stream call
    tr = M.FeatureTable.synthetic(320, 1, dev, torch.bfloat16, snr=0.5)
    dv = M.FeatureTable.synthetic(160, 2, dev, torch.bfloat16, snr=0.5)
    ld = {"train": M.FeatureLoader(tr, 20, shuffle=True), "dev": M.FeatureLoader(dv, 20, shuffle=False)}
    args = SimpleNamespace(vid_len=(8, 32), num_outputs=60, drpt=0.5, inner_representation_size=16, batchnorm=False, alphas=True,
                           multitask=False, weightsharing=False, batchsize=20, eta_max=1e-3, eta_min=1e-6, Ti=1, Tm=2,
                           use_dataparallel=False, verbose=False, epochs=2)
    confs = [np.stack([rng.integers(0, 4, L), rng.integers(0, 4, L), rng.integers(0, 2, L)], 1) for L in (4, 2, 4, 1, 3)]
    res = {}
    for host in ("", "1"):
        if host:
            monkeypatch.setenv("MFAS_HOST_INIT", "1")
        else:
            monkeypatch.delenv("MFAS_HOST_INIT", raising=False)
        torch.manual_seed(3)
        res[host] = M.train_sampled_models(confs, M.Searchable_Skeleton_Image_Net, ld, args, dev)
    assert res[""] == res["1"]
    # (the check is keyed by device, class AND geometry: every key this call created must have passed)
    mine = [v for k, v in NS._DEVICE_STREAMS_OK.items() if k[0] == str(dev) and k[2] == "Searchable_Skeleton_Image_Net"]
    assert mine and all(v is True for v in mine)


def test_written_out_adam_equals_the_library_forms_in_situ(dev):
    """common.hip.h writes Adam's square root and divisions out as correctly-rounding fma sequences (tools/adam_exact.hip checks
    them in isolation).  Here the WHOLE library is built a second time with -DMFAS_ADAM_LIBRARY_FORMS (sqrtf() and operator/ in
    every kernel: sweep, resident units, chains) and both builds train the same populations — resident persistent schedule,
    launch-per-phase lean chain, general chain with BatchNorm, fused two-group launches — from the same start: every parameter, both
    Adam moments and all statistics must come out bit for bit the same."""
    import hashlib
    import __graft_entry__ as ge
    lib = ge.build_variant("adamlib", ["-DMFAS_ADAM_LIBRARY_FORMS"])
    code = r"""
import sys, hashlib
sys.path.insert(0, %r)
import numpy as np, torch
import mfas_amd as M
from oracle import np_oracle as O
dev = torch.device("cuda:0")
tr = M.FeatureTable.synthetic(640, 1, dev, torch.bfloat16, snr=0.3)
dv = M.FeatureTable.synthetic(160, 2, dev, torch.bfloat16, snr=0.3)
rng = np.random.default_rng(2)
h = hashlib.sha256()
for R, B, bn, K in ((16, 20, False, 6), (16, 16, True, 40), (128, 16, True, 3), (128, 20, True, 12), (64, 16, False, 5)):
    hp = M.Hyper(R=R, B=B, bn=bn, drpt=0.5, alphas=(R == 64), tap_bits=16)
    confs = [np.stack([rng.integers(0, 4, L), rng.integers(0, 4, L), rng.integers(0, 3, L)], 1) for L in rng.integers(1, 5, K)]
    pop = M.Population(hp, confs, dev, drop_seeds=list(range(K)))
    pop.init(list(range(1, K + 1)))
    nb = -(-640 // B)
    stats, status = pop.train(tr, dv, 2, O.eta_sequence(1e-3, 1e-6, 1, 2, 640 / B, 2 * nb))
    assert not status.any()
    h.update(stats.tobytes())
    for k in range(K):
        for plane in range(3):
            h.update(pop.get_params(k, plane).cpu().numpy().tobytes())
    pop.close()
print("DIGEST", h.hexdigest())
""" % ROOT
    out = {}
    for name, env in (("written-out", {}), ("library", {"MFAS_LIB": lib})):
        res = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
        assert res.returncode == 0, (name, res.stdout[-1000:], res.stderr[-3000:])
        out[name] = [l for l in res.stdout.splitlines() if l.startswith("DIGEST")][-1]
    assert out["written-out"] == out["library"], out


def test_lds_dma_staging_variant(dev):
    """The -DMFAS_RES_DMA=1 build variant (resident units stage their 16-bit rows by per-wave LDS-DMA into a k-block-major image, no
    barrier; opt-in: measured a wash to a loss) must train bit-identically to the launch-per-phase schedule like the default
    build does: the persistent-schedule tests run on it in a subprocess."""
    import __graft_entry__ as ge
    lib = ge.build_variant("dma", ["-DMFAS_RES_DMA=1"])
    res = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-x", "-q", "-m", "gpu", "-k",
                          "persistent_resident_schedule_bit_identical_full_size or (persistent_schedule_fuzz_bit_identical and (0 or 7 or 13))"],
                         env=dict(os.environ, MFAS_LIB=lib), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0 and " passed" in res.stdout and "failed" not in res.stdout, (res.stdout[-2000:], res.stderr[-2000:])


@pytest.mark.parametrize("case", range(6))
def test_device_side_torch_streams_fuzz(dev, case):
    """Random geometries (odd tap widths, R, classes, depth, huge / tiny seeds) for the device-side construction draws: equal to
    building the module under torch.manual_seed(seed), bit for bit, for every candidate."""
    import mfas_amd as M
    from mfas_amd import ntu_searchable as NS
    from types import SimpleNamespace
    rng = np.random.default_rng(500 + case)
    widths = [16, 24, 40, 64, 100, 128, 200, 256, 1000]
    args = SimpleNamespace(vid_len=(8, 32), num_outputs=int(rng.choice([2, 7, 23, 60, 101])), drpt=0.5,
                           inner_representation_size=int(rng.choice([1, 8, 16, 24, 100, 128, 256])), batchnorm=bool(rng.integers(0, 2)),
                           alphas=bool(rng.integers(0, 2)), multitask=False, batchsize=16,
                           s_sizes=tuple(int(x) for x in rng.choice(widths, 4)), v_sizes=tuple(int(x) for x in rng.choice(widths, 4)))
    hp = M.Hyper.from_args(args)
    K = int(rng.integers(1, 40))
    confs = [np.stack([rng.integers(0, 4, L), rng.integers(0, 4, L), rng.integers(0, 3, L)], 1) for L in rng.integers(1, 5, K)]
    seeds = [int(x) for x in rng.integers(0, 2 ** 63 - 1, K)]
    seeds[0] = 0
    pop = M.Population(hp, confs, dev)
    pop.init_torch_streams(seeds, np.stack([NS.torch_init_bounds(c, hp) for c in confs]))
    for k in rng.permutation(K)[:6]:
        torch.manual_seed(seeds[k])
        want = NS.Searchable_Skeleton_Image_Net(args, confs[k]).flat_params()
        got = pop.get_params(int(k)).cpu()
        assert torch.equal(got, want), (case, int(k), int((got != want).sum()), got.numel())
    pop.close()

"""CPU-only tests of the host logic and of the C-ABI library surface (no compute calls without a GPU)."""
import ctypes
import os
import re
import subprocess
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import np_oracle as O
from tests.helpers import CONFS, golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def mkargs(**kw):
    a = dict(vid_len=(8, 32), num_outputs=60, drpt=0.5, inner_representation_size=16, batchnorm=True,
             alphas=False, multitask=False, weightsharing=False, batchsize=16, eta_max=1e-3, eta_min=1e-6,
             Ti=1, Tm=2, use_dataparallel=False, verbose=False, epochs=2)
    a.update(kw)
    return SimpleNamespace(**a)


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from mfas_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "mfas_hip.h")).read()
    declared = set(re.findall(r"\b(mfas_[a-z_]+)\s*\(", hdr))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name
    assert _lib.lib().mfas_version() >= 100


def test_empty_environment_selects_the_tested_defaults():
    """The library's switches live in ONE struct parsed in ONE place (mfas_hip.hip::tuning_from_env, at create / plan time).  With no
    MFAS_* variable set the parsed set is exactly the defaults the suites run; every documented variable moves exactly its own
    field; the product library never parses the test hooks; INTEGRATION.md's table names every switch (and nothing else in the
    engine sources calls getenv, apart from the process-wide MFAS_NO_ROCTX marker switch)."""
    code = r"""
import sys, os, json
sys.path.insert(0, %r)
from mfas_amd import _lib
print(json.dumps(_lib.tuning()))
""" % ROOT
    clean = {k: v for k, v in os.environ.items() if not k.startswith("MFAS_")}

    def parsed(**extra):
        out = subprocess.run([sys.executable, "-c", code], env=dict(clean, **extra), capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        import json
        return json.loads(out.stdout.strip().splitlines()[-1])

    defaults = {"persist": "-1", "no_lean_chain": "0", "persist_no_resident": "0", "persist_no_res_chain": "0", "subchunks": "0",
                "subchunk_skip": "0", "groups": "0", "same_group": "-1", "no_tap_major": "0", "force_tap_major": "0", "no_red_in_sweep": "0", "force_red_in_sweep": "0",
                "occ_bytes": "-1", "no_xcd_placement": "0", "n_xcd": "0", "persist_trace": "0", "nt": "-1", "eval_no_x16": "0",
                "eval_no_msplit": "0", "eval_no_b3": "0", "eval_no_wl": "0", "no_gather": "0", "gather_verbose": "0", "no_plain_chain": "0",
                "persist_verbose": "0", "prof_every": "16", "chain_split": "-1", "test_not_resident": "-1", "test_lose_step": "-1", "hooks": "0"}
    assert parsed() == defaults
    env_of = {"persist": ("MFAS_PERSIST", "0", "0"), "no_lean_chain": ("MFAS_NO_LEAN_CHAIN", "1", "1"),
              "persist_no_resident": ("MFAS_PERSIST_NO_RESIDENT", "1", "1"), "persist_no_res_chain": ("MFAS_PERSIST_NO_RES_CHAIN", "1", "1"),
              "subchunks": ("MFAS_SUBCHUNKS", "4", "4"), "subchunk_skip": ("MFAS_SUBCHUNK_SKIP", "3", "3"), "groups": ("MFAS_GROUPS", "2", "2"),
              "same_group": ("MFAS_SAME_GROUP", "2", "2"), "no_tap_major": ("MFAS_NO_TAP_MAJOR", "1", "1"),
              "force_tap_major": ("MFAS_FORCE_TAP_MAJOR", "1", "1"), "no_red_in_sweep": ("MFAS_NO_RED_IN_SWEEP", "1", "1"), "force_red_in_sweep": ("MFAS_FORCE_RED_IN_SWEEP", "1", "1"),
              "occ_bytes": ("MFAS_OCC_BYTES", "3e8", "3e+08"), "no_xcd_placement": ("MFAS_NO_XCD_PLACEMENT", "1", "1"), "n_xcd": ("MFAS_XCDS", "4", "4"),
              "persist_trace": ("MFAS_PERSIST_TRACE", "1", "1"), "nt": ("MFAS_NT", "1", "1"), "eval_no_x16": ("MFAS_EVAL_NO_X16", "1", "1"),
              "eval_no_msplit": ("MFAS_EVAL_NO_MSPLIT", "1", "1"), "eval_no_b3": ("MFAS_EVAL_NO_B3", "1", "1"), "eval_no_wl": ("MFAS_EVAL_NO_WL", "1", "1"),
              "no_gather": ("MFAS_NO_GATHER", "1", "1"), "gather_verbose": ("MFAS_GATHER_VERBOSE", "1", "1"),
              "no_plain_chain": ("MFAS_NO_PLAIN_CHAIN", "1", "1"), "persist_verbose": ("MFAS_PERSIST_VERBOSE", "2", "2"),
              "prof_every": ("MFAS_PROF_EVERY", "4", "4"), "chain_split": ("MFAS_CHAIN_SPLIT", "0", "0")}
    assert set(env_of) | {"test_not_resident", "test_lose_step", "hooks"} == set(defaults)
    everything = parsed(**{var: val for var, val, _ in env_of.values()})
    for field, (var, val, shown) in env_of.items():
        assert everything[field] == shown, (field, everything[field])
    # the product library does not parse the hooks
    hooked = parsed(MFAS_PERSIST_TEST_NOT_RESIDENT="1", MFAS_PERSIST_TEST_LOSE_STEP="3")
    assert hooked == defaults
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for var, _, _ in env_of.values():
        assert var in doc, var
    src = "".join(open(os.path.join(ROOT, "mfas_amd", "csrc", f)).read() for f in os.listdir(os.path.join(ROOT, "mfas_amd", "csrc"))
                  if f.endswith((".hip", ".hip.h")))
    body = src.split("static Tuning tuning_from_env()")[1].split("extern \"C\" int mfas_tuning_describe")[0]
    assert src.count("getenv(") - body.count("getenv(") == 1, "a getenv outside tuning_from_env (other than MFAS_NO_ROCTX)"


def test_struct_layouts_match_header(tmp_path):
    """The ctypes mirrors must match what a C compiler makes of include/mfas_hip.h (sizes and key offsets)."""
    from mfas_amd import _lib
    src = tmp_path / "sz.c"
    src.write_text(r"""
#include <stdio.h>
#include <stddef.h>
#include "mfas_hip.h"
int main(void) {
    printf("%zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(mfas_hyper), sizeof(mfas_table), sizeof(mfas_epoch_stats),
           offsetof(mfas_hyper, drpt), offsetof(mfas_hyper, s_sizes), offsetof(mfas_hyper, f1_threshold),
           offsetof(mfas_table, multilabel), offsetof(mfas_table, dtype));
    return 0;
}
""")
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [ctypes.sizeof(_lib.mfas_hyper), ctypes.sizeof(_lib.mfas_table), ctypes.sizeof(_lib.mfas_epoch_stats),
            _lib.mfas_hyper.drpt.offset, _lib.mfas_hyper.s_sizes.offset, _lib.mfas_hyper.f1_threshold.offset,
            _lib.mfas_table.multilabel.offset, _lib.mfas_table.dtype.offset]
    assert got == want, (got, want)


def test_no_engine_without_gpu():
    """The product path fails loudly instead of falling back to CPU."""
    from mfas_amd import Hyper, Population
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        Population(Hyper(R=16, bn=True), [np.array(CONFS["l1"])], "cpu")


def test_product_never_imports_oracle():
    for root, _, files in os.walk(os.path.join(ROOT, "mfas_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f


def test_scheduler_matches_reference_golden():
    from mfas_amd import LRCosineAnnealingScheduler
    g = golden("g1_scheduler.npz")
    for j in range(4):
        Ti, Tm, nbpe, n = g[f"cfg{j}"]
        s = LRCosineAnnealingScheduler(1e-3, 1e-6, Ti, Tm, nbpe)
        np.testing.assert_allclose(s.eta_table(int(n)), g[f"eta{j}"], rtol=1e-12)
    s = LRCosineAnnealingScheduler(1e-3, 1e-6, 1, 2, 4.0)
    seq = [s.step() or s.eta for _ in range(16)]
    assert s.Ti == 4     # SURVEY §3.3 known answer: Ti ends at 4


def test_adam_step_scalars():
    from mfas_amd.scheduler import adam_step_scalars
    etas = O.eta_sequence(1e-3, 1e-6, 1, 2, 4.0, 12)
    sc = adam_step_scalars(etas)
    hp = O.Hyper()
    for t in range(12):
        ss, b2 = O.adam_scalars(float(etas[t]), t + 1, hp)
        assert sc[t, 0] == ss and sc[t, 1] == b2


def test_layer_configurations_and_module_surface():
    import mfas_amd as M
    g = golden("g8_controller.npz")
    assert np.array_equal(np.array(M.get_possible_layer_configurations(0)), g["layer_confs"])
    args = mkargs(batchnorm=True, drpt=0.5, inner_representation_size=16, alphas=True)
    conf = np.array(CONFS["c4"])
    m = M.Searchable_Skeleton_Image_Net(args, conf)
    for attr in ("conf", "args", "rgbnet", "skenet", "alphas", "gp_v", "gp_s", "fusion_layers",
                 "central_classifier"):
        assert hasattr(m, attr)
    keys = set(m.state_dict().keys())
    ohp = O.Hyper(R=16, B=16, bn=True, drpt=0.5, alphas=True)
    want = set(O.init_params(conf, ohp, 0).keys())
    assert want <= keys
    assert keys - want == {f"fusion_layers.{i}.2.num_batches_tracked" for i in range(4)}
    groups = m.central_params()
    assert len(groups) == 3
    n = sum(p.numel() for gr in groups for p in gr["params"])
    assert n == 124732 + 4 * 32 + 4      # P (R=16, conf-4 taps) + BN affine + alphas
    assert [l[0].in_features for l in m.fusion_layers] == [1536, 2320, 1296, 2576]
    # flat layout round trip and agreement with the engine's documented order
    flat = m.flat_params()
    layout, total = M.flat_layout(conf, m.hyper())
    assert total == flat.numel()
    m2 = M.Searchable_Skeleton_Image_Net(args, conf)
    m2.load_flat(flat)
    for k, v in m.state_dict().items():
        if "num_batches" not in k:
            assert torch.equal(v, m2.state_dict()[k]), k
    with pytest.raises(ValueError):
        M.Searchable_Skeleton_Image_Net(mkargs(batchnorm=False, drpt=0.0), conf)
    m.train(True)
    with pytest.raises(ValueError):          # no taps, no forward
        m(({}, {}))
    if not torch.cuda.is_available():        # train- and eval-mode forward both run on the HIP engine: CPU tensors fail loudly
        x = {k: torch.zeros(4, w) for k, w in zip(("s0", "s1", "s2", "s3", "v0", "v1", "v2", "v3"), O.S_SIZES + O.V_SIZES)}
        for mode in (True, False):
            m.train(mode)
            with pytest.raises(RuntimeError):
                m(({k: v for k, v in x.items() if k[0] == "v"}, {k: v for k, v in x.items() if k[0] == "s"}))


def test_initial_flat_params_equals_module_construction():
    """train_sampled_models initialises candidates without building the module: the numbers must be the module's, bit for bit."""
    from types import SimpleNamespace
    import torch
    from mfas_amd import ntu_searchable as NS
    rng = np.random.default_rng(2)
    for bn, drpt, R, C in [(False, 0.5, 16, 60), (True, 0.5, 128, 60), (True, 0.0, 32, 11)]:
        args = SimpleNamespace(vid_len=(8, 32), num_outputs=C, drpt=drpt, inner_representation_size=R, batchnorm=bn, alphas=True, batchsize=16,
                               multitask=False)
        for L in (1, 2, 4):
            conf = np.stack([rng.integers(0, 4, L), rng.integers(0, 4, L), rng.integers(0, 3, L)], 1)
            torch.manual_seed(1234 + L)
            want = NS.Searchable_Skeleton_Image_Net(args, conf).flat_params()
            after_module = torch.rand(1)
            torch.manual_seed(1234 + L)
            got = NS.initial_flat_params(args, conf)
            after_fast = torch.rand(1)
            assert torch.equal(want, got), (bn, drpt, R, C, L)
            assert torch.equal(after_module, after_fast)        # the same amount of the random stream was consumed
    with pytest.raises(ValueError):
        NS.initial_flat_params(SimpleNamespace(vid_len=(8, 32), num_outputs=60, drpt=0.0, inner_representation_size=16, batchnorm=False, batchsize=16,
                                               alphas=False, multitask=False), np.array([[0, 0, 0]]))


def test_weight_sharing_keys():
    import mfas_amd as M
    args = mkargs()
    a = M.Searchable_Skeleton_Image_Net(args, np.array(CONFS["l2"]))
    sd = M.get_central_states(a, {})
    assert set(sd) == {"0.L_3072_16.A_sigmoid", "1.L_2192_16.A_lrelu"}
    b = M.Searchable_Skeleton_Image_Net(args, np.array([[2, 3, 1], [1, 1, 0]]))
    M.set_central_states(b, sd)
    assert torch.equal(b.fusion_layers[0][0].weight, a.fusion_layers[0][0].weight)
    assert not torch.equal(b.fusion_layers[1][0].weight[:, :16], a.fusion_layers[1][0].weight[:, :16])


def test_assignment_is_balanced_and_deterministic():
    from mfas_amd import population as P
    rng = np.random.default_rng(0)
    confs = [rng.integers(0, 4, (rng.integers(1, 5), 3)) % [4, 4, 2] for _ in range(50)]
    costs = [P.candidate_cost(c, 16, O.S_SIZES, O.V_SIZES) for c in confs]
    # P_i (124,732 for conf 4 at R=16) + the depth-proportional latency term
    assert P.candidate_cost(CONFS["c4"], 16, O.S_SIZES, O.V_SIZES) == 124732 + 4 * P.LATENCY_PARAMS_PER_CELL
    assert P.candidate_cost(CONFS["c4"], 128, O.S_SIZES, O.V_SIZES) == 1040444 + 4 * 8 * P.LATENCY_PARAMS_PER_CELL   # (no BN affine in the cost)
    for world in (1, 2, 4, 8):
        own = P.assign(costs, world)
        assert own == P.assign(costs, world)
        loads = [sum(c for c, o in zip(costs, own) if o == r) for r in range(world)]
        assert max(loads) - min(loads) <= max(costs)
        assert sorted(set(own)) == list(range(world))
        own2, cap = P.shard(costs, world)
        assert own2 == own and cap == max(own.count(r) for r in range(world))


def test_sharder_uses_only_the_ranks_that_pay():
    """The step-time model behind shard(costs, world, R) (mfas_amd/population.py): strong scaling of a search-sized population is
    latency-bound — a resident step costs the same for 1..8 candidates — so a call takes only as many ranks as shorten it."""
    from mfas_amd import population as P
    c16 = P.candidate_cost(CONFS["c4"], 16, O.S_SIZES, O.V_SIZES)
    c128 = P.candidate_cost(CONFS["c4"], 128, O.S_SIZES, O.V_SIZES)
    assert P.predicted_step_us([c16] * 1, 16) == P.predicted_step_us([c16] * 8, 16) < P.predicted_step_us([c16] * 9, 16)
    assert P.choose_ranks([c16] * 6, 8, 16) == 1                    # 6 candidates: one GPU is as fast as eight
    assert P.choose_ranks([c16] * 16, 2, 16) == 2                   # BASELINE configs[2]: 11.3 -> 10.7 us per step, worth 5 %
    assert P.choose_ranks([c16] * 50, 8, 16) == 7                   # configs[3]: shares of <= 8 already with 7 ranks
    assert P.choose_ranks([c128] * 1024, 8, 128) == 8               # the weak-scaling headline (128 per rank) uses every rank
    assert P.choose_ranks([c128] * 6, 8, 128) == 1
    for world, K, R, c in ((2, 16, 16, c16), (8, 50, 16, c16), (8, 6, 16, c16), (4, 64, 128, c128)):
        owner, cap = P.shard([c] * K, world, R)
        used = P.choose_ranks([c] * K, world, R)
        assert sorted(set(owner)) == list(range(used)) and cap == max(owner.count(r) for r in range(world))
        assert len(owner) == K
    owner, cap = P.shard([c16] * 6, 8)                                # without a model: every rank, as before
    assert sorted(set(owner)) == list(range(6))


WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
from mfas_amd import population as P
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = P.dist_info()
K = 7
costs = [100, 30, 70, 10, 90, 50, 20]
own, cap = P.shard(costs, world)
mine = [i for i in range(K) if own[i] == rank]
acc = [0.01 * (i + 1) for i in mine]          # stand-in for the engine's result of candidate i
out = P.gather_accuracies(mine, acc, K, cap=cap)     # ONE all_gather, cap known locally
assert np.allclose(out, [0.01 * (i + 1) for i in range(K)]), out
confs = [np.array([[i % 4, (i + 1) % 4, i % 2]]) for i in range(K)]
seed = P.broadcast_seed(1234 + rank, confs=confs)
assert seed == 1234
# a rank whose controller drifted (different sampled configurations) must raise, not train a different population
bad = confs if rank == 0 else confs[::-1]
try:
    P.broadcast_seed(7, confs=bad)
    drift = False
except RuntimeError:
    drift = True
assert drift == (rank != 0), (rank, drift)
# the same collectives over an explicit group next to the default one (bench.py: an RCCL group beside a gloo control plane)
g = dist.new_group(backend="gloo")
P.set_group(g)
out = P.gather_accuracies(mine, acc, K, cap=cap)
assert np.allclose(out, [0.01 * (i + 1) for i in range(K)]), out
assert P.broadcast_seed(99 + rank, confs=confs) == 99
P.set_group(None)
print("rank", rank, "ok", mine, flush=True)
dist.destroy_process_group()
"""


@pytest.mark.parametrize("world", [2])
def test_population_gather_gloo(tmp_path, world):
    """N>1 path on CPU: 2 processes, gloo, 127.0.0.1 rendezvous; sharding + all_gather of accuracies."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    for p in procs:
        out, _ = p.communicate(timeout=180)
        assert p.returncode == 0, out.decode()
        assert b"ok" in out


def test_initial_flat_params_private_generator_and_threads():
    """A private torch.Generator seeded like the global stream yields the module's numbers, and train_sampled_models' default
    initialiser (thread pool -> one staging buffer -> set_params per candidate) hands every candidate exactly those."""
    from mfas_amd import ntu_searchable as NS
    from mfas_amd import Hyper
    args = mkargs(inner_representation_size=32, alphas=True)
    hp = Hyper.from_args(args)
    rng = np.random.default_rng(3)
    confs = [np.stack([rng.integers(0, 4, L), rng.integers(0, 4, L), rng.integers(0, 3, L)], 1) for L in (1, 4, 2, 3, 4, 1)]
    want = []
    for i, c in enumerate(confs):
        torch.manual_seed(900 + 2 + i)
        want.append(NS.Searchable_Skeleton_Image_Net(args, c).flat_params())

    class FakePop:
        def __init__(self):
            self.got = {}

        def set_params(self, j, flat, sync=True):
            self.got[j] = flat.clone()

    before = torch.get_rng_state()
    pop = FakePop()
    NS._init_population_from_torch(pop, args, confs, list(range(len(confs))), hp, 900, NS.Searchable_Skeleton_Image_Net, [], {}, "cpu")
    assert torch.equal(before, torch.get_rng_state())          # the global stream is not consumed
    assert sorted(pop.got) == list(range(len(confs)))
    for j, w in enumerate(want):
        assert torch.equal(pop.got[j], w), j
    pop, mods = FakePop(), {}
    NS._init_population_from_torch(pop, args, confs, [1, 3], hp, 900, NS.Searchable_Skeleton_Image_Net, [1, 3], mods, "cpu")
    assert torch.equal(pop.got[0], want[1]) and torch.equal(pop.got[1], want[3]) and set(mods) == {1, 3}
    # per-candidate sample orders: permutations, a function of (seed, candidate, epoch) only
    o = NS.make_order_per_candidate(500, 3, True, 77, "cpu", [0, 5, 9])
    assert o.dtype == torch.int32 and tuple(o.shape) == (3, 3, 500)
    assert all(sorted(o[k, e].tolist()) == list(range(500)) for k in range(3) for e in range(3))
    assert torch.equal(NS.make_order_per_candidate(500, 3, True, 77, "cpu", [5])[0], o[1])
    assert not torch.equal(NS.make_order_per_candidate(500, 3, True, 78, "cpu", [5])[0], o[1])
    assert not torch.equal(o[0, 0], o[0, 1]) and not torch.equal(o[0, 0], o[1, 0])
    assert NS.make_order_per_candidate(500, 3, False, 77, "cpu", [0]) is None


def test_round_planner_and_shard_call_without_a_device():
    """population.split_rounds asks the engine's layout query (a pure host function: it works without a GPU and assumes the
    MI355X's 256 CUs then); shard_call prices shares with the shipped constants when there is no device to calibrate on."""
    from mfas_amd import Hyper
    from mfas_amd import population as P
    hp = Hyper(R=16, B=20, tap_bits=16)
    c4 = np.array(CONFS["c4"])
    assert [(len(p), r) for p, r in P.split_rounds(hp, [c4] * 6, "cuda:0")] == [(6, True)]
    assert [(len(p), r) for p, r in P.split_rounds(hp, [c4] * 28, "cuda:0")] == [(28, True)]
    r50 = P.split_rounds(hp, [c4] * 50, "cuda:0")
    assert [(len(p), r) for p, r in r50] == [(28, True), (22, True)] and sorted(sum((p for p, _ in r50), [])) == list(range(50))
    assert [(len(p), r) for p, r in P.split_rounds(hp, [c4] * 64, "cuda:0")] == [(64, False)]       # 28 + 28 + 8: the third round too empty
    hp128 = Hyper(R=128, B=16, bn=True, tap_bits=16)
    assert [(len(p), r) for p, r in P.split_rounds(hp128, [c4] * 6, "cuda:0")] == [(6, False)]
    hpb = Hyper(R=16, B=48, tap_bits=16)              # B > 32: no lean chain, no resident schedule
    assert [(len(p), r) for p, r in P.split_rounds(hpb, [c4] * 6, "cuda:0")] == [(6, False)]
    rep = P.representative_conf(hp)
    assert sum(O.S_SIZES[s] + O.V_SIZES[v] for s, v, _ in rep) == 7552
    for K, world, want_used in ((6, 8, 1), (16, 2, 2), (1024, 8, 8)):
        h = hp if K < 100 else hp128
        owner, cap, model = P.shard_call([c4] * K, h, world, None, False)
        assert not model.calibrated and sorted(set(owner)) == list(range(want_used)), (K, world, sorted(set(owner)))
        assert cap == max(owner.count(r) for r in range(world)) and len(owner) == K
    owner, cap, model = P.shard_call([c4] * 6, hp, 8, None, True)           # engine_all_ranks
    assert model is None and sorted(set(owner)) == list(range(6))
    assert P.shard_call([], hp, 4, None, False)[:2] == ([], 1)
    assert abs(P._interp([(1.0, 10.0), (3.0, 20.0)], 2.0) - 15.0) < 1e-12 and P._interp([(1.0, 10.0), (3.0, 20.0)], 5.0) == 30.0
    assert P._interp([(1.0, 10.0), (3.0, 20.0)], 0.5) == 10.0


def test_asan_build_variant_runs_the_host_side_queries():
    """MFAS_ASAN=1 build variant (host-side AddressSanitizer of the C-ABI library): builds, loads under LD_PRELOAD of the sanitizer
    runtime, and the layout planner + input validation run clean.  (The GPU suite trains a population with it.)"""
    import __graft_entry__ as ge
    ge.build_asan()
    code = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
from mfas_amd import Hyper, _lib
from mfas_amd.engine import plan_population
assert _lib.LIB_PATH.endswith("libmfas_hip_asan.so")
c4 = np.array([[3, 1, 1], [1, 3, 0], [1, 1, 1], [3, 3, 0]])
rng = np.random.default_rng(0)
for R, B in ((16, 20), (16, 16), (128, 16), (32, 40), (200, 64)):
    hp = Hyper(R=R, B=B, bn=True, tap_bits=16)
    for K in (1, 7, 28, 29, 130):
        confs = [np.stack([rng.integers(0, 4, L), rng.integers(0, 4, L), rng.integers(0, 3, L)], 1) for L in rng.integers(1, 5, K)]
        p = plan_population(hp, confs, "cuda:0")
        assert p["candidates"] == K
for bad in ([[9, 1, 1]], [[0, 0, 3]], [[0, 8, 0]]):
    try:
        plan_population(Hyper(R=16), [np.array(bad)], "cuda:0")
        raise SystemExit("accepted " + repr(bad))
    except RuntimeError:
        pass
_lib.lib().mfas_range_push(b"x"); _lib.lib().mfas_range_pop()
print("ASAN-OK", _lib.lib().mfas_source_digest().decode())
""" % ROOT
    res = subprocess.run([sys.executable, "-c", code], env=ge.asan_env(), capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "ASAN-OK mfas-src-digest:" + ge.source_digest() in res.stdout, (res.stdout[-1500:], res.stderr[-3000:])
    assert "AddressSanitizer" not in res.stderr, res.stderr[-3000:]


REQUEUE_WORKER = r"""
import os, sys, warnings
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
from mfas_amd import population as P
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = P.dist_info()
K = 9
wanted = [0, 1, 2, 3, 5, 6, 7, 8]          # (candidate 4 not wanted: return_model-style subset)
costs = [100, 30, 70, 10, 50, 20, 60, 40]
owner, cap = P.shard(costs, world)
calls = []
def share(fail_first_on, always_fail=False):
    def f(idx):
        calls.append(list(idx))
        if always_fail or (rank == fail_first_on and len(calls) == 1):
            raise RuntimeError("boom")
        return {{i: 0.01 * (i + 1) for i in idx}}
    return f
want = [0.01 * (i + 1) if i in wanted else 0.0 for i in range(K)]
# 1. nobody fails: one gather
calls.clear()
out = P.train_sharded(wanted, owner, cap, K, costs, share(-1))
assert np.allclose(out, want) and len(calls) == 1, (out, calls)
# 2. rank 1's share fails: its candidates are re-queued on rank 0, everybody gets every accuracy
calls.clear()
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    out = P.train_sharded(wanted, owner, cap, K, costs, share(1))
assert np.allclose(out, want), out
lost = [i for i, o in zip(wanted, owner) if o == 1]
if rank == 0:
    assert len(calls) == 2 and sorted(calls[1]) == sorted(lost), calls
    assert any("re-queued" in str(x.message) for x in w)
else:
    assert len(calls) == 1
# 3. every rank fails: everybody raises, nobody hangs
calls.clear()
try:
    P.train_sharded(wanted, owner, cap, K, costs, share(-1, always_fail=True))
    raised = False
except RuntimeError as e:
    raised = "every rank failed" in str(e)
assert raised
# 4. an argument error that only ONE rank sees (a bad configuration in its share, engine.py raises ValueError for it): that rank
#    re-raises it, the other learns of it from the same gather and raises too — nobody is left waiting in the collective, and
#    nothing is re-queued (one call of `share` per rank)
calls.clear()
def bad_conf(idx):
    calls.append(list(idx))
    if rank == 1:
        raise ValueError("configuration entry out of range")
    return {{i: 0.01 * (i + 1) for i in idx}}
try:
    P.train_sharded(wanted, owner, cap, K, costs, bad_conf)
    verdict = "returned"
except ValueError as e:
    verdict = "value" if rank == 1 and "out of range" in str(e) else "wrong"
except RuntimeError as e:
    verdict = "runtime" if rank == 0 and "[1]" in str(e) and "argument / programming error" in str(e) else "wrong"
assert verdict == ("value" if rank == 1 else "runtime") and len(calls) == 1, (verdict, calls)
print("rank", rank, "requeue ok", flush=True)
dist.destroy_process_group()
"""


def test_failed_rank_is_requeued_gloo(tmp_path):
    """population.train_sharded: a rank whose share raises reports it through the collective; its candidates are trained by the
    ranks that did not fail (failure detection / re-queue; candidates are independent, ntu_searchable.py:38-94)."""
    script = tmp_path / "worker.py"
    script.write_text(REQUEUE_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    for p in procs:
        out, _ = p.communicate(timeout=180)
        assert p.returncode == 0, out.decode()
        assert b"requeue ok" in out

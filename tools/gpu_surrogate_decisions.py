"""Does the graph-replayed GPU surrogate (mfas_amd/search/surrogate.py::GraphedSurrogateTrainer) reproduce the CPU decisions of the
reference-pinned controller runs (golden G9)?  Also a longer run (the search script's defaults: 50 surrogate epochs) CPU vs GPU."""
import os, random, sys, time
from types import SimpleNamespace
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import np_oracle as O
from tests.helpers import golden
import mfas_amd as M
from mfas_amd.search import ModelSearcher, SimpleRecurrentSurrogate

def flat_calls(calls):
    return np.concatenate([np.concatenate([np.asarray(c).reshape(-1), [-1]]) for call in calls for c in call])

def run(device, iters, levels, K, epochs, seed=3):
    args = SimpleNamespace(search_iterations=iters, max_progression_levels=levels, num_samples=K, initial_temperature=10.0,
                           final_temperature=0.2, temperature_decay=4.0, lr_surrogate=0.001, epochs_surrogate=epochs, verbose=False)
    calls = []
    def fake_train(confs, model_type, dataloaders, a, dev, state_dict=None):
        calls.append([np.array(c) for c in confs])
        return [O.fake_accuracy(c) for c in confs]
    np.random.seed(seed); torch.manual_seed(seed); random.seed(seed)
    surrogate = SimpleRecurrentSurrogate(100, 3, 100).to(device)
    t0 = time.time()
    ModelSearcher(args)._epnas(None, {"model": surrogate, "criterion": torch.nn.MSELoss()}, None,
                               {"train_sampled_fun": fake_train, "get_layer_confs": M.get_possible_layer_configurations}, device)
    return flat_calls(calls), time.time() - t0

torch.set_num_threads(4)
g = golden("g9_controller_run.npz")
for tag, iters, levels, K in (("a", 2, 3, 5), ("b", 3, 4, 6)):
    for dev in ("cpu", "cuda:0"):
        f, dt = run(dev, iters, levels, K, 8)
        same = len(f) == len(g[tag + "/calls_flat"]) and np.array_equal(f, g[tag + "/calls_flat"])
        print(f"G9 {tag} on {dev}: decisions == reference golden: {same}  ({dt:.2f} s)", flush=True)
for epochs, iters, levels, K in ((50, 3, 4, 15), (50, 5, 4, 50)):
    a, ta = run("cpu", iters, levels, K, epochs)
    b, tb = run("cuda:0", iters, levels, K, epochs)
    n = min(len(a), len(b)); first = int(np.argmax(a[:n] != b[:n])) if (a[:n] != b[:n]).any() else -1
    print(f"epochs {epochs} iters {iters} levels {levels} K {K}: cpu {ta:.2f} s, gpu {tb:.2f} s, identical decision stream: {len(a) == len(b) and first < 0} (first difference at element {first} of {n})", flush=True)

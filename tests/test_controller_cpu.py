"""CPU: the search controller (mfas_amd.search) reproduces the reference controller decision-for-decision when both
are driven by the same fake trainer and the same numpy / torch / random seeds (golden G8, G9)."""
import random
from types import SimpleNamespace

import numpy as np
import torch

from oracle import np_oracle as O
from tests.helpers import CONFS, golden


def flat_calls(calls):
    return np.concatenate([np.concatenate([np.asarray(c).reshape(-1), [-1]]) for call in calls for c in call])


def test_tools_known_answers():
    from mfas_amd.search import tools
    import mfas_amd as M
    g = golden("g8_controller.npz")
    a = SimpleNamespace(initial_temperature=10.0, final_temperature=0.2, temperature_decay=4.0)
    np.testing.assert_allclose([tools.compute_temperature(i, a) for i in range(12)], g["temperature"], rtol=1e-12)
    lc = M.get_possible_layer_configurations(0)
    merged0 = tools.merge_unfolded_with_sampled([], lc, 0)
    assert np.array_equal(np.array(merged0), g["merged0"]) and merged0[0].shape == (1, 3)
    np.random.seed(0)
    samp = tools.sample_k_configurations(merged0, np.linspace(0.1, 0.9, len(merged0)), 5, 10.0)
    assert np.array_equal(np.array(samp), g["sampled0"])
    merged1 = tools.merge_unfolded_with_sampled(samp, M.get_possible_layer_configurations(1), 1)
    assert np.array_equal(np.array(merged1), g["merged1"])
    np.random.seed(1)
    samp1 = tools.sample_k_configurations(merged1, np.linspace(0.2, 0.8, len(merged1)), 5, 2.5)
    assert np.array_equal(np.array(samp1), g["sampled1"])
    merged1b = tools.merge_unfolded_with_sampled(samp1, lc, 0)
    assert np.array_equal(np.array(merged1b), g["merged1b"])


def test_epnas_matches_reference_decisions():
    import mfas_amd as M
    from mfas_amd.search import ModelSearcher, SimpleRecurrentSurrogate
    g = golden("g9_controller_run.npz")
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
        surrogate = SimpleRecurrentSurrogate(100, 3, 100)
        assert sum(p.numel() for p in surrogate.parameters()) == 81301
        s_data = ModelSearcher(args)._epnas(None, {"model": surrogate, "criterion": torch.nn.MSELoss()}, None,
                                            {"train_sampled_fun": fake_train,
                                             "get_layer_confs": M.get_possible_layer_configurations}, "cpu")
        # 32 single-layer confs first, then K per step: calls = iters*levels, candidates = 32 + (calls-1)*K
        assert [len(c) for c in calls] == list(g[tag + "/call_sizes"]) == [32] + [K] * (iters * levels - 1)
        assert np.array_equal(flat_calls(calls), g[tag + "/calls_flat"])
        _, accs, _ = s_data.get_k_best(5)
        np.testing.assert_allclose(np.sort(np.array(accs)), g[tag + "/best_accs"], rtol=1e-12)
        pred = [surrogate.eval_model(np.array(CONFS["c4"]), "cpu"), surrogate.eval_model(np.array(CONFS["l2"]), "cpu")]
        np.testing.assert_allclose(pred, g[tag + "/final_pred"], rtol=1e-4)


def test_randsearch_matches_reference_decisions():
    import mfas_amd as M
    from mfas_amd.search import ModelSearcher
    g = golden("g9_controller_run.npz")
    args = SimpleNamespace(search_iterations=2, max_progression_levels=3, num_samples=4, verbose=False)
    calls = []

    def fake_train(confs, model_type, dataloaders, a, device, state_dict=None):
        calls.append([np.array(c) for c in confs])
        return [O.fake_accuracy(c) for c in confs]

    np.random.seed(5)
    random.seed(5)
    ModelSearcher(args)._randsearch(None, None, {"train_sampled_fun": fake_train,
                                                 "get_layer_confs": M.get_possible_layer_configurations}, "cpu")
    assert len(calls) == 6
    assert np.array_equal(flat_calls(calls), g["r/calls_flat"])

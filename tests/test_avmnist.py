"""AV-MNIST variant (5 + 3 taps of non-16-multiple widths, plain [Linear, nl] cells): oracle vs the reference golden
(CPU) and engine vs both (GPU)."""
from types import SimpleNamespace

import numpy as np
import pytest

from oracle import np_oracle as O
from tests.helpers import engine_hyper, etas_for, golden

SS, VS = (3, 6, 12, 24, 48), (3, 6, 12)


def hyper():
    return O.Hyper(R=16, C=10, B=16, bn=False, drpt=0.0, epochs=3, s_sizes=SS, v_sizes=VS, allow_plain_cell=True)


def tables():
    return (O.synth_table(192, 71, snr=1.0, C=10, s_sizes=SS, v_sizes=VS),
            O.synth_table(96, 72, snr=1.0, C=10, s_sizes=SS, v_sizes=VS))


def test_oracle_vs_reference_avmnist():
    g = golden("g12_avmnist.npz")
    assert len(O.get_possible_layer_configurations(0)) == 32 and g["layer_confs"].shape == (30, 3)
    ttr, tdv = tables()
    hp = hyper()
    per_epoch = []
    for i in range(3):
        conf = g[f"conf{i}"]
        hist = []
        acc = O.train_candidate(conf, hp, O.init_params(conf, hp, 30 + i), ttr, tdv, history=hist)
        assert acc == pytest.approx(float(g["plain/accs"][i]), abs=1e-8)
        per_epoch += [h["dev_acc"] for h in hist]
    np.testing.assert_allclose(per_epoch, g["plain/dev_acc_per_epoch"], atol=6e-5)


@pytest.mark.gpu
def test_engine_vs_reference_avmnist():
    torch = pytest.importorskip("torch")
    import mfas_amd as M
    from mfas_amd import avmnist_searchable as AV
    dev = torch.device("cuda:0")
    g = golden("g12_avmnist.npz")
    assert np.array_equal(np.array(AV.get_possible_layer_configurations(0)), g["layer_confs"])
    ttr, tdv = tables()
    ohp = hyper()
    confs = [g[f"conf{i}"] for i in range(3)]
    pop = M.Population(engine_hyper(ohp), confs, dev)
    for k, c in enumerate(confs):
        pop.set_state_dict(k, O.init_params(c, ohp, 30 + k))
    ta, tb = M.FeatureTable.from_numpy(ttr, dev), M.FeatureTable.from_numpy(tdv, dev)
    assert ta.taps["s0"].shape[1] == 16 and ta.widths["s0"] == 3         # rows zero-padded to 16
    stats, status = pop.train(ta, tb, 3, etas_for(ohp, 192))
    accs = [M.best_dev_accuracy(stats[k], 96) for k in range(3)]
    np.testing.assert_allclose(accs, g["plain/accs"], atol=1.0 / 96 + 1e-9)
    per_epoch = [stats["dev_corrects"][k, e] / 96 for k in range(3) for e in range(3)]
    np.testing.assert_allclose(per_epoch, g["plain/dev_acc_per_epoch"], atol=1.0 / 96 + 1e-4)
    # state round trip with padded columns: exported weights carry the TRUE widths
    sd = pop.get_state_dict(0)
    assert sd["fusion_layers.0.0.weight"].shape == (16, 48 + 12)
    pop.close()
    # mirror: module + population driver with dropout
    args = SimpleNamespace(channels=3, num_outputs=10, drpt=0.3, inner_representation_size=16, batchnorm=False,
                           alphas=False, multitask=False, weightsharing=False, batchsize=16, eta_max=1e-3,
                           eta_min=1e-6, Ti=1, Tm=2, use_dataparallel=False, verbose=False, epochs=3, vid_len=(8, 32))
    m = AV.Searchable_Audio_Image_Net(args, confs[1])
    assert [l[0].in_features for l in m.fusion_layers] == [24 + 6, 48 + 12 + 16] and len(m.fusion_layers[0]) == 3
    args0 = SimpleNamespace(**{**vars(args), "drpt": 0.0})
    assert len(AV.Searchable_Audio_Image_Net(args0, confs[0]).fusion_layers[0]) == 2     # plain [Linear, nl]
    ld = {"train": M.FeatureLoader(ta, 16, shuffle=True), "dev": M.FeatureLoader(tb, 16, shuffle=False)}
    torch.manual_seed(4)
    accs = AV.train_sampled_models(confs, AV.Searchable_Audio_Image_Net, ld, args, dev)
    assert len(accs) == 3 and all(0.0 <= a <= 1.0 for a in accs)

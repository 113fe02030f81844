"""CPU: the oracle's multi-label (MM-IMDB) head against the reference's WeightedCrossEntropyWithLogits and
train_mmimdb_track_f1 goldens (G11)."""
import numpy as np
import pytest

from oracle import np_oracle as O
from tests.helpers import golden


def test_bce_loss_and_gradient():
    g = golden("g11_mmimdb.npz")
    C = 23
    w = O.mm_pos_weight(C)
    logits = (O.hash_noise(77, 16 * C).reshape(16, C) * np.float32(2.0)).astype(np.float32)
    z = (O.hash_u01(78, 16 * C).reshape(16, C) < 0.2).astype(np.float32)
    loss, d = O.bce_loss(logits, z, w)
    np.testing.assert_allclose(loss, g["loss"], rtol=2e-6)
    np.testing.assert_allclose(d, g["dlogits"], rtol=2e-5, atol=1e-8)


def test_f1_samples_matches_sklearn():
    from sklearn.metrics import f1_score
    logits = O.hash_noise(5, 64 * 23).reshape(64, 23)
    z = (O.hash_u01(6, 64 * 23).reshape(64, 23) < 0.15).astype(np.float32)
    pred = (1.0 / (1.0 + np.exp(-logits))) > 0.3
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = f1_score(z > 0.5, pred, average="samples")
    got = O.f1_samples_fixed(logits, z, 0.3) / float(1 << 32) / 64
    assert abs(got - want) < 1e-8


@pytest.mark.parametrize("tag,R", [("a", 16), ("b", 32)])
def test_multilabel_train_loop_vs_reference(tag, R):
    g = golden("g11_mmimdb.npz")
    conf = g[tag + "/conf"]
    hp = O.Hyper(R=R, C=23, B=16, bn=True, drpt=0.0, epochs=3, s_sizes=O.MM_S_SIZES, v_sizes=O.MM_V_SIZES,
                 loss_mode=1, pos_weight=O.mm_pos_weight(23))
    ttr, tdv = O.synth_table_mm(128, 41), O.synth_table_mm(96, 42)
    hist = []
    best = O.train_candidate(conf, hp, O.init_params(conf, hp, 17), ttr, tdv, history=hist)
    np.testing.assert_allclose([h["dev_f1"] for h in hist], g[tag + "/f1_per_epoch"], atol=6e-5)   # printed with 4 decimals
    assert abs(best - float(g[tag + "/best_f1"])) < 1e-6

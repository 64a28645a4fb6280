"""The training oracle (oracle/head_train_oracle.py) against fixtures produced by the REFERENCE module in
train() mode with torch.optim.Adam and the loss lines of train_lstm_model
(tests/golden/make_goldens.py::g_head_train -> tests/golden/head_train_*.npz)."""
import os

import numpy as np
import pytest
import torch

from cbas_amd import config as C, synth, weights as W
from oracle import head_train_oracle as HT

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = {"h64": (64, 1, True), "h64_l2": (64, 2, True), "h128": (128, 1, True),
         # other hidden sizes (the UI's "LSTM hidden size" is free: app.py:321) and the 2-stream head (classifier_head.py:74-84)
         "h32": (32, 1, True), "h96_noacc": (96, 1, False), "h48_noacc_l2": (48, 2, False)}
STRIDE = 13


def unpack(g, key):
    """-> (values, is_sample): full tensor, or the every-13th-element sample of a large one."""
    if key in g.files:
        return g[key], False
    return g[key + "#sample"], True


def like(a, is_sample):
    a = np.asarray(a)
    return a.reshape(-1)[::STRIDE] if is_sample else a


def problem(tag):
    h, nl, acc = CASES[tag]
    g = np.load(os.path.join(GOLD, f"head_train_{tag}.npz"))
    hcfg = C.HeadConfig(in_features=768, out_features=9, lstm_hidden_size=h, lstm_layers=nl, use_acceleration=acc)
    hw = W.synth_head_weights(hcfg, 4321)
    x, y = synth.train_windows(5, int(g["B"]), 768, 9, 31)
    cw = g["class_weights"] if "class_weights" in g.files else None
    return g, hcfg, hw, x, y, cw


def test_dropout_masks_are_counter_based():
    a = HT.dropout_keep(77, 0, 0, 100000, 0.1)
    b = HT.dropout_keep(77, 0, 0, 100000, 0.1)
    c = HT.dropout_keep(77, 1, 0, 100000, 0.1)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    assert abs(a.mean() - 0.9) < 5e-3 and abs(HT.dropout_keep(3, 2, 3, 200000, 0.15).mean() - 0.85) < 5e-3
    # prefix property: element i does not depend on n
    assert np.array_equal(HT.dropout_keep(77, 0, 0, 1000, 0.1), a[:1000])


def test_cross_entropy_matches_torch():
    rng = np.random.default_rng(0)
    logits = torch.tensor(rng.normal(size=(40, 9)), dtype=torch.float64)
    y = torch.tensor(rng.integers(0, 9, 40))
    w = torch.tensor(np.linspace(0.5, 1.5, 9))
    for cw in (None, w):
        for ls in (0.0, 0.1):
            ref = torch.nn.CrossEntropyLoss(weight=cw, label_smoothing=ls)(logits, y)
            got = HT.cross_entropy(logits, y, cw, ls)
            assert abs(float(ref) - float(got)) < 1e-12


@pytest.mark.parametrize("tag", list(CASES))
def test_first_step_loss_and_gradients(tag):
    g, hcfg, hw, x, y, cw = problem(tag)
    masks = HT.make_masks(int(g["seed"]), 0, x.shape[0], 31, 128, 256)
    loss, ce, cov, logits, latent, grads = HT.loss_and_grads(x, y, hw, 31, cw, float(g["label_smoothing"]), masks)
    assert abs(loss - float(g["loss0"])) < 2e-5 * abs(float(g["loss0"]))
    assert abs(ce - float(g["ce0"])) < 2e-5 * abs(float(g["ce0"])) and abs(cov - float(g["cov0"])) < 1e-4 * abs(float(g["cov0"])) + 1e-7
    np.testing.assert_allclose(logits, g["logits0"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(latent, g["latent0"], rtol=0, atol=2e-6)
    names = [k[len("grad0/"):].split("#")[0] for k in g.files if k.startswith("grad0/")]
    assert set(names) == set(hw), set(hw) ^ set(names)
    for name in sorted(set(names)):
        ref, is_s = unpack(g, "grad0/" + name)
        got = like(grads[name], is_s)
        scale = max(np.abs(ref).max(), 1e-6)
        # + absolute floor: d loss / d attention_head.bias is exactly 0 (softmax shift invariance), rounding noise only
        assert np.abs(got - ref).max() <= 3e-4 * scale + 2e-7, (name, np.abs(got - ref).max(), scale)


@pytest.mark.parametrize("tag", list(CASES))
def test_three_adam_steps(tag):
    g, hcfg, hw, x, y, cw = problem(tag)
    wf, losses = HT.train_steps([x], [y], hw, 3, float(g["lr"]), int(g["seed"]), 31, float(g["weight_decay"]), cw,
                                float(g["label_smoothing"]))
    for s in range(3):
        assert abs(losses[s] - float(g[f"loss{s}"])) < 3e-4 * abs(float(g[f"loss{s}"])), (s, losses[s], float(g[f"loss{s}"]))
    lr = float(g["lr"])
    for name in hw:
        if name == "attention_head.bias":
            # its true gradient is 0 (softmax over the window is shift invariant), so Adam amplifies pure
            # rounding noise (|g| ~ eps = 1e-8) into +-lr steps; the parameter has no effect on any output
            continue
        ref, is_s = unpack(g, "final/" + name)
        got = like(wf[name], is_s)
        # after 3 Adam steps every weight has moved by <= 3 lr; agreement to a small fraction of that
        assert np.abs(got - ref).max() <= 0.05 * lr + 1e-6 * np.abs(ref).max(), (name, np.abs(got - ref).max())


# ---- host side of cbas_amd.train (no GPU needed) --------------------------------------------------
def test_pack_unpack_round_trip_and_initial_weights():
    from cbas_amd.head import pack_head_weights
    from cbas_amd.train import head_weight_names, initial_head_weights, unpack_head_weights
    from cbas_amd.weights import head_param_shapes
    for h, nl in ((64, 1), (128, 2)):
        hcfg = C.HeadConfig(in_features=768, out_features=7, lstm_hidden_size=h, lstm_layers=nl)
        w = initial_head_weights(hcfg, 5)
        shapes = head_param_shapes(hcfg)
        assert set(w) == set(shapes) == set(head_weight_names(hcfg))
        assert all(tuple(w[k].shape) == tuple(shapes[k]) and w[k].dtype == np.float32 for k in w)
        assert float(w["gate"]) == np.float32(0.2) and float(w["attention_temp"]) == 1.0     # classifier_head.py:89,94
        assert np.all(w["cls_ln.weight"] == 1) and np.all(w["acc_ln.bias"] == 0)
        bound = 1.0 / np.sqrt(768)                                                          # nn.Linear default init
        assert np.abs(w["lin1.weight"]).max() <= bound and np.abs(w["lin1.weight"]).max() > 0.9 * bound
        assert np.abs(w["lstm.weight_hh_l0"]).max() <= 1.0 / np.sqrt(h)                     # nn.LSTM default init
        blob = pack_head_weights(hcfg, w)
        back = unpack_head_weights(hcfg, blob)
        assert all(np.array_equal(back[k], w[k]) for k in w)
        w2 = initial_head_weights(hcfg, 5)
        assert all(np.array_equal(w2[k], w[k]) for k in w)                                  # seeded: reproducible


def test_collate_drops_failed_samples():
    from cbas_amd.train import collate_fn
    batch = [(torch.zeros(31, 8), torch.tensor(2)), (torch.zeros(31, 8), torch.tensor(-1)), (torch.ones(31, 8), torch.tensor(0))]
    d, l = collate_fn(batch)
    assert d.shape == (2, 31, 8) and l.tolist() == [2, 0]
    d, l = collate_fn([(torch.zeros(31, 8), torch.tensor(-1))])
    assert d.numel() == 0 and l.numel() == 0


def test_trainer_refuses_cpu_device():
    from cbas_amd.train import HeadTrainer
    hcfg = C.HeadConfig(in_features=768, out_features=9)
    with pytest.raises(RuntimeError, match="GPU"):
        HeadTrainer(hcfg, W.synth_head_weights(hcfg, 1), "cpu")

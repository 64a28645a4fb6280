"""Head training on the GPU (cbas_head_train_* through the C ABI) against the training oracle
(oracle/head_train_oracle.py, pinned to the reference by tests/test_train_oracle.py) and against the
reference fixtures themselves (tests/golden/head_train_*.npz)."""
import os

import numpy as np
import pytest
import torch

from cbas_amd import config as C, synth, weights as W

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = {"h64": (64, 1, True), "h64_l2": (64, 2, True), "h128": (128, 1, True),
         # other hidden sizes (the UI's "LSTM hidden size" is free: app.py:321) and the 2-stream head (classifier_head.py:74-84)
         "h32": (32, 1, True), "h96_noacc": (96, 1, False), "h48_noacc_l2": (48, 2, False)}
STRIDE = 13


def golden(tag):
    h, nl, acc = CASES[tag]
    g = np.load(os.path.join(GOLD, f"head_train_{tag}.npz"))
    hcfg = C.HeadConfig(in_features=768, out_features=9, lstm_hidden_size=h, lstm_layers=nl, use_acceleration=acc)
    hw = W.synth_head_weights(hcfg, 4321)
    x, y = synth.train_windows(5, int(g["B"]), 768, 9, 31)
    cw = g["class_weights"] if "class_weights" in g.files else None
    return g, hcfg, hw, x, y, cw


def gval(g, key, arr):
    if key in g.files:
        return g[key], np.asarray(arr)
    return g[key + "#sample"], np.asarray(arr).reshape(-1)[::STRIDE]


def make_trainer(hcfg, hw, g=None, cw=None, **kw):
    from cbas_amd.train import HeadTrainer
    if g is not None:
        kw.setdefault("lr", float(g["lr"]))
        kw.setdefault("weight_decay", float(g["weight_decay"]))
        kw.setdefault("label_smoothing", float(g["label_smoothing"]))
        kw.setdefault("seed", int(g["seed"]))
    return HeadTrainer(hcfg, hw, "cuda", class_weights=cw, max_batch=kw.pop("max_batch", 64), **kw)


@pytest.mark.parametrize("tag", list(CASES))
def test_first_step_matches_reference_fixture(tag):
    """Loss, logits, latent and every parameter gradient of step 0 (dropout masks on) vs the reference."""
    g, hcfg, hw, x, y, cw = golden(tag)
    tr = make_trainer(hcfg, hw, g, cw)
    loss, ce, cov = tr.step(torch.from_numpy(x), torch.from_numpy(y), update=False)
    logits, latent = tr.last_outputs(x.shape[0])
    grads = tr.grads()
    tr.close()
    assert abs(loss - float(g["loss0"])) < 5e-5 * abs(float(g["loss0"])), (loss, float(g["loss0"]))
    assert abs(ce - float(g["ce0"])) < 5e-5 * abs(float(g["ce0"])) and abs(cov - float(g["cov0"])) < 2e-4 * abs(float(g["cov0"])) + 1e-7
    np.testing.assert_allclose(logits, g["logits0"], rtol=0, atol=2e-4)      # logits are O(5): 4e-5 relative
    np.testing.assert_allclose(latent, g["latent0"], rtol=0, atol=2e-5)
    for name in hw:
        ref, got = gval(g, "grad0/" + name, grads[name])
        scale = max(np.abs(ref).max(), 1e-6)
        # d loss / d attention_head.bias is exactly 0 (softmax shift invariance): both sides hold rounding noise
        floor = 5e-6 if name == "attention_head.bias" else 2e-7
        assert np.abs(got - ref).max() <= 5e-4 * scale + floor, (name, float(np.abs(got - ref).max()), float(scale))


@pytest.mark.parametrize("tag", list(CASES))
def test_three_adam_steps_match_reference_fixture(tag):
    g, hcfg, hw, x, y, cw = golden(tag)
    tr = make_trainer(hcfg, hw, g, cw)
    xt, yt = torch.from_numpy(x), torch.from_numpy(y)
    losses = [tr.step(xt, yt)[0] for _ in range(3)]
    wf = tr.weights()
    tr.close()
    for s in range(3):
        assert abs(losses[s] - float(g[f"loss{s}"])) < 5e-4 * abs(float(g[f"loss{s}"])), (s, losses[s], float(g[f"loss{s}"]))
    lr = float(g["lr"])
    for name in hw:
        if name == "attention_head.bias":      # exactly-zero true gradient: Adam amplifies rounding noise (test_train_oracle.py)
            continue
        ref, got = gval(g, "final/" + name, wf[name])
        # Adam divides by sqrt(v): an element whose gradient is rounding noise (|g| ~ 1e-8, e.g. weights fed by
        # a channel that is ~0 in this batch) moves by up to lr per step in a noise-determined direction, on
        # any two implementations.  So: nearly all elements agree to a small fraction of lr, none differs by
        # more than the 3 steps could move it apart.
        d = np.abs(got - ref)
        assert (d > 0.1 * lr).mean() < 0.01, (name, float((d > 0.1 * lr).mean()))
        assert d.max() <= 2 * 3 * lr * 1.01, (name, float(d.max()))


def test_against_oracle_without_dropout_odd_batch():
    """Batch 37 (not a multiple of anything), no dropout, class weights + label smoothing: gradients vs
    the oracle's autograd in float64."""
    from oracle import head_train_oracle as HT
    hcfg = C.HeadConfig(in_features=768, out_features=9)
    hw = W.synth_head_weights(hcfg, 4321)
    x, y = synth.train_windows(11, 37, 768, 9, 31)
    cw = np.linspace(0.7, 1.3, 9).astype(np.float32)
    tr = make_trainer(hcfg, hw, cw=cw, lr=1e-3, label_smoothing=0.1, dropout=False)
    loss, ce, cov = tr.step(torch.from_numpy(x), torch.from_numpy(y), update=False)
    grads = tr.grads()
    tr.close()
    rl, rce, rcov, _, _, rg = HT.loss_and_grads(x, y, hw, 31, cw, 0.1, None, dtype=torch.float64)
    assert abs(loss - rl) < 2e-5 * abs(rl) and abs(ce - rce) < 2e-5 * abs(rce) and abs(cov - rcov) < 1e-4 * abs(rcov) + 1e-7
    for name in hw:
        ref, got = rg[name], grads[name]
        scale = max(np.abs(ref).max(), 1e-6)
        assert np.abs(got - ref).max() <= 3e-4 * scale + 2e-7, (name, float(np.abs(got - ref).max()), float(scale))


def test_single_window_batch_has_no_covariance_term():
    from oracle import head_train_oracle as HT
    hcfg = C.HeadConfig(in_features=768, out_features=9)
    hw = W.synth_head_weights(hcfg, 4321)
    x, y = synth.train_windows(3, 1, 768, 9, 31)
    tr = make_trainer(hcfg, hw, lr=1e-3, dropout=False)
    loss, ce, cov = tr.step(torch.from_numpy(x), torch.from_numpy(y), update=False)
    grads = tr.grads()
    tr.close()
    rl, _, rcov, _, _, rg = HT.loss_and_grads(x, y, hw, 31, None, 0.0, None, dtype=torch.float64)
    assert cov == 0.0 and rcov == 0.0 and abs(loss - rl) < 2e-5 * abs(rl)
    for name in ("lin2.weight", "lstm.weight_hh_l0", "cls_bottleneck.0.weight"):
        scale = max(np.abs(rg[name]).max(), 1e-6)
        assert np.abs(grads[name] - rg[name]).max() <= 3e-4 * scale + 2e-7, name


def test_training_is_deterministic_and_learns():
    """Same seed -> bit-identical parameters; 40 steps on separable synthetic windows cut the loss and the
    trained weights classify through the inference head."""
    from cbas_amd.head import ClassifierLSTMDeltas
    from cbas_amd.train import initial_head_weights
    hcfg = C.HeadConfig(in_features=768, out_features=9)
    w0 = initial_head_weights(hcfg, 3)
    x, y = synth.train_windows(21, 256, 768, 9, 31)
    xt, yt = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()

    def run():
        tr = make_trainer(hcfg, w0, lr=2e-3, seed=9, max_batch=256)
        losses = [tr.step(xt, yt)[0] for _ in range(40)]
        w = tr.weights()
        tr.close()
        return losses, w

    l1, w1 = run()
    l2, w2 = run()
    assert l1 == l2 and all(np.array_equal(w1[k], w2[k]) for k in w1)
    assert l1[-1] < 0.5 * l1[0], (l1[0], l1[-1])
    m = ClassifierLSTMDeltas(768, 9)
    m.load_state_dict(w1)
    m.to("cuda")
    logits, _ = m(xt)
    acc = (logits.argmax(1).cpu().numpy() == y).mean()
    m.close()
    assert acc > 0.8, acc


def test_train_lstm_model_shim_runs_the_reference_loop():
    """cbas_amd.train.train_lstm_model: epochs, reports, early-stopping bookkeeping and the returned triple."""
    import threading
    from cbas_amd.train import train_lstm_model
    x, y = synth.train_windows(31, 320, 768, 4, 31)

    class DS(torch.utils.data.Dataset):
        def __init__(self, a, b):
            self.a, self.b = a, b

        def __len__(self):
            return len(self.b)

        def __getitem__(self, i):
            return torch.from_numpy(self.a[i]), torch.tensor(int(self.b[i]))

    train, test = DS(x[:256], y[:256]), DS(x[256:], y[256:])
    model, reports, best = train_lstm_model(train, test, 31, ["a", "b", "c", "d"], threading.Event(), batch_size=64, lr=2e-3,
                                            epochs=4, device="cuda", patience=3, seed=1)
    assert model is not None and 0 <= best < 4 and 1 <= len(reports) <= 4
    assert set(reports[0].train_report) >= {"a", "b", "c", "d", "weighted avg", "macro avg"}
    assert reports[-1].train_report["weighted avg"]["f1-score"] > reports[0].train_report["weighted avg"]["f1-score"] - 1e-9
    sd = model.state_dict()
    assert isinstance(sd["gate"], torch.Tensor) and sd["lin1.weight"].shape == (4, 768)
    logits, latent = model(torch.from_numpy(x[:8]).cuda())
    assert logits.shape == (8, 4) and latent.shape == (8, 128)
    model.close()


@pytest.mark.parametrize("I,Cn,T,h,nl,B,acc", [(384, 4, 15, 64, 1, 20, True), (768, 12, 41, 128, 2, 9, True),
                                               (768, 9, 63, 64, 1, 10, True), (768, 5, 95, 128, 1, 6, True),     # sweep_runner.py:110 lengths
                                               (768, 9, 31, 16, 1, 12, True), (768, 3, 31, 80, 2, 7, False),
                                               (384, 9, 63, 112, 1, 5, False)])
def test_other_shapes_against_oracle(I, Cn, T, h, nl, B, acc):
    """ViT-S width / short windows / 12 classes / long windows with a stacked h=128 LSTM / the sweep's 63- and 95-frame
    windows (the expand kernels then keep only U in LDS and read the projected rows from global memory): gradients vs
    float64 autograd."""
    from oracle import head_train_oracle as HT
    hcfg = C.HeadConfig(in_features=I, out_features=Cn, seq_len=T, lstm_hidden_size=h, lstm_layers=nl, use_acceleration=acc)
    hw = W.synth_head_weights(hcfg, 99)
    x, y = synth.train_windows(17, B, I, Cn, T)
    tr = make_trainer(hcfg, hw, lr=1e-3, weight_decay=1e-2, dropout=True, seed=5)
    loss, ce, cov = tr.step(torch.from_numpy(x), torch.from_numpy(y), update=False)
    grads = tr.grads()
    tr.close()
    masks = HT.make_masks(5, 0, B, T, 128, 256)
    rl, rce, rcov, _, _, rg = HT.loss_and_grads(x, y, hw, T, None, 0.0, masks, dtype=torch.float64)
    assert abs(loss - rl) < 3e-5 * abs(rl), (loss, rl)
    for name in hw:
        scale = max(np.abs(rg[name]).max(), 1e-6)
        floor = 5e-6 if name == "attention_head.bias" else 2e-7
        assert np.abs(grads[name] - rg[name]).max() <= 4e-4 * scale + floor, (name, float(np.abs(grads[name] - rg[name]).max()), float(scale))


def test_step_rate_next_to_the_cpu_oracle(capsys):
    """Optimisation steps per second of cbas_head_train_step (batch 512, train_lstm_model's default) next to the CPU
    restatement of the same step (torch autograd + Adam) on the host cores: a reported ratio, asserted only to be > 1."""
    import time
    from oracle import head_train_oracle as HT
    from cbas_amd.train import HeadTrainer, initial_head_weights
    B, steps, cpu_steps = 512, 30, 2
    hcfg = C.HeadConfig(in_features=768, out_features=9)
    w0 = initial_head_weights(hcfg, 0)
    x, y = synth.train_windows(1, B, 768, 9, 31)
    xt, yt = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    tr = HeadTrainer(hcfg, w0, "cuda", lr=1e-4, max_batch=B, seed=1)
    for _ in range(3):
        tr.step(xt, yt, want_loss=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(xt, yt, want_loss=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    tr.close()
    t0 = time.perf_counter()
    HT.train_steps([x], [y], w0, cpu_steps, 1e-4, 1)
    dc = (time.perf_counter() - t0) / cpu_steps
    with capsys.disabled():
        print(f"\nhead training, batch {B}: GPU {dt * 1e3:.2f} ms/step ({B / dt:.0f} windows/s); CPU restatement on "
              f"{torch.get_num_threads()} threads {dc * 1e3:.0f} ms/step -> {dc / dt:.0f}x")
    assert dc > dt


"""Head training on the MI355X: drop-in for the reference's ``train_lstm_model``
(backend/cbas.py:1274-1422) with the optimisation step (forward in train() mode, loss, backward,
Adam) running in the fp32 HIP kernels of libcbas_mi355x.so (``cbas_head_train_*``).

What stays on the host, exactly as in the reference: the DataLoader iteration (shuffle, collate that
drops failed samples, cbas.py:1253-1260), the per-epoch evaluation reports from scikit-learn
(cbas.py:1364-1392), early stopping on the chosen F1 (cbas.py:1394-1411) and the returned triple
``(final_model, epoch_reports, best_epoch)``.  The per-epoch evaluation runs through the HIP inference
head (``cbas_amd.head.ClassifierLSTMDeltas``).

Dropout keep-masks come from a counter-based hash (seed, step, layer, element) instead of torch's
global RNG, so a run is reproducible from its seed; the masks have the reference's rates (0.1 after
the three bottleneck GELUs, 0.15 after lin0's GELU).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Mapping, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .config import HeadConfig
from .head import ClassifierLSTMDeltas, pack_head_weights
from .weights import head_param_shapes


def head_weight_names(cfg: HeadConfig) -> List[str]:
    names = ["gate", "attention_temp"]
    streams = ("cls", "delta", "acc") if cfg.use_acceleration else ("cls", "delta")      # classifier_head.py:74-84
    for s in streams:
        names += [f"{s}_bottleneck.0.weight", f"{s}_bottleneck.0.bias"]
    for s in streams:
        names += [f"{s}_ln.weight", f"{s}_ln.bias"]
    names += ["lin0.0.weight", "lin0.0.bias", "lin1.weight", "lin1.bias"]
    for layer in range(cfg.lstm_layers):
        for sfx in ("", "_reverse"):
            names += [f"lstm.weight_ih_l{layer}{sfx}", f"lstm.weight_hh_l{layer}{sfx}",
                      f"lstm.bias_ih_l{layer}{sfx}", f"lstm.bias_hh_l{layer}{sfx}"]
    names += ["attention_head.weight", "attention_head.bias", "lin2.weight", "lin2.bias"]
    return names


def unpack_head_weights(cfg: HeadConfig, blob: np.ndarray) -> Dict[str, np.ndarray]:
    """Inverse of ``pack_head_weights``: blob (include/cbas_mi355x.h order) -> state dict of arrays."""
    shapes = head_param_shapes(cfg)
    out, o = {}, 0
    for n in head_weight_names(cfg):
        k = int(np.prod(shapes[n])) if len(shapes[n]) else 1
        out[n] = blob[o:o + k].reshape(shapes[n]).copy()
        o += k
    if o != blob.shape[0]:
        raise ValueError(f"blob has {blob.shape[0]} floats, the config describes {o}")
    return out


class HeadTrainer:
    """One ``cbas_head_trainer`` handle: parameters, gradients and Adam state live on the device."""

    def __init__(self, cfg: HeadConfig, weights: Mapping[str, np.ndarray], device, lr: float = 1e-4,
                 weight_decay: float = 0.0, label_smoothing: float = 0.0, class_weights: Optional[Sequence[float]] = None,
                 max_batch: int = 512, seed: int = 0, dropout: bool = True):
        cfg.validate()
        if cfg.lstm_hidden_size % 16 or not 16 <= cfg.lstm_hidden_size <= 128:
            raise NotImplementedError("on-device training is built for lstm_hidden_size = 16, 32, ... 128 (train_lstm_model's "
                                      f"default is 64, sweep_runner.py uses 128); got {cfg.lstm_hidden_size}")
        self.cfg = cfg
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError(f"head training runs only on a GPU device (got {self.device}); there is no CPU path")
        self._lib = _lib.load()
        dev = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self._cc = _lib.HeadConfigC(cfg.in_features, cfg.out_features, cfg.seq_len, cfg.bottleneck_dim, cfg.lin0_dim,
                                    cfg.lstm_hidden_size, cfg.center_window_size, cfg.ema_alpha, cfg.lstm_layers, int(cfg.use_acceleration))
        tc = _lib.TrainConfigC(float(lr), float(weight_decay), float(label_smoothing), int(max_batch), int(seed) & (2 ** 64 - 1),
                               1 if dropout else 0)
        blob = pack_head_weights(cfg, weights)
        self.n_blob = int(blob.shape[0])
        cw = None
        if class_weights is not None:
            cw = np.ascontiguousarray(np.asarray(class_weights, np.float32))
            if cw.shape != (cfg.out_features,):
                raise ValueError(f"class_weights has shape {cw.shape}, expected ({cfg.out_features},)")
        h = C.c_void_p()
        _lib.check(self._lib.cbas_head_train_create(C.byref(self._cc), C.byref(tc), blob.ctypes.data, self.n_blob,
                                                    cw.ctypes.data if cw is not None else None, dev, C.byref(h)),
                   "cbas_head_train_create")
        self._h = h
        self.max_batch = int(max_batch)

    def step(self, x: torch.Tensor, labels: torch.Tensor, update: bool = True, want_loss: bool = True):
        """One optimisation step on windows x (B, T, I) float32 and labels (B,).  Returns
        (loss, cross_entropy, covariance_penalty) when ``want_loss`` (synchronises), else None."""
        if x.dim() != 3 or x.shape[1] != self.cfg.seq_len or x.shape[2] != self.cfg.in_features:
            raise ValueError(f"expected (B, {self.cfg.seq_len}, {self.cfg.in_features}), got {tuple(x.shape)}")
        B = int(x.shape[0])
        if labels.shape != (B,):
            raise ValueError(f"labels has shape {tuple(labels.shape)}, expected ({B},)")
        x = x.to(self.device, torch.float32).contiguous()
        y = labels.to(self.device, torch.int32).contiguous()
        out = (C.c_float * 3)() if want_loss else None
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self._lib.cbas_head_train_step(self._h, x.data_ptr(), y.data_ptr(), B, 1 if update else 0, out, stream),
                   "cbas_head_train_step")
        self._keep = (x, y)            # keep the inputs alive until the next call (the step is asynchronous)
        return (float(out[0]), float(out[1]), float(out[2])) if want_loss else None

    def _read(self, what: int) -> Dict[str, np.ndarray]:
        blob = np.empty(self.n_blob, np.float32)
        _lib.check(self._lib.cbas_head_train_read(self._h, what, blob.ctypes.data, self.n_blob), "cbas_head_train_read")
        return unpack_head_weights(self.cfg, blob)

    def weights(self) -> Dict[str, np.ndarray]:
        return self._read(0)

    def grads(self) -> Dict[str, np.ndarray]:
        return self._read(1)

    def last_outputs(self, n: int):
        logits = np.empty((n, self.cfg.out_features), np.float32)
        latent = np.empty((n, 2 * self.cfg.lstm_hidden_size), np.float32)
        _lib.check(self._lib.cbas_head_train_last_outputs(self._h, logits.ctypes.data, latent.ctypes.data, n),
                   "cbas_head_train_last_outputs")
        return logits, latent

    def close(self):
        if getattr(self, "_h", None) is not None:
            self._lib.cbas_head_train_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PerformanceReport:
    """Same attribute bundle as the reference's (backend/cbas.py:1267-1272)."""

    def __init__(self, train_report: dict, train_cm: np.ndarray, val_report: dict, val_cm: np.ndarray):
        self.train_report = train_report
        self.train_cm = train_cm
        self.val_report = val_report
        self.val_cm = val_cm


def collate_fn(batch):
    """backend/cbas.py:1253-1260: drop samples whose label is -1 (failed to load)."""
    batch = [b for b in batch if int(b[1]) != -1]
    if not batch:
        return torch.tensor([]), torch.tensor([])
    dcls, lbls = zip(*batch)
    return torch.stack([torch.as_tensor(d) for d in dcls]), torch.stack([torch.as_tensor(l) for l in lbls])


def initial_head_weights(cfg: HeadConfig, seed: Optional[int] = None) -> Dict[str, np.ndarray]:
    """Fresh parameters with the reference constructor's initialisation (torch's nn.Linear / nn.LSTM /
    nn.LayerNorm defaults, gate = 0.2, attention_temp = 1.0; classifier_head.py:62-100), built from torch
    modules of the same shapes - not from the reference class."""
    g = torch.Generator()
    if seed is not None:
        g.manual_seed(int(seed))
    state = torch.random.get_rng_state()
    try:
        if seed is not None:
            torch.manual_seed(int(seed))
        I, Cn, Bn, L0, h = cfg.in_features, cfg.out_features, cfg.bottleneck_dim, cfg.lin0_dim, cfg.lstm_hidden_size
        w: Dict[str, np.ndarray] = {"gate": np.float32(0.2).reshape(()), "attention_temp": np.float32(1.0).reshape(())}
        for s in ("cls", "delta", "acc"):
            lin = torch.nn.Linear(I, Bn)
            w[f"{s}_bottleneck.0.weight"], w[f"{s}_bottleneck.0.bias"] = lin.weight.detach().numpy(), lin.bias.detach().numpy()
        for s in ("cls", "delta", "acc"):
            w[f"{s}_ln.weight"], w[f"{s}_ln.bias"] = np.ones(Bn, np.float32), np.zeros(Bn, np.float32)
        lin0 = torch.nn.Linear(3 * Bn, L0)
        w["lin0.0.weight"], w["lin0.0.bias"] = lin0.weight.detach().numpy(), lin0.bias.detach().numpy()
        att = torch.nn.Linear(2 * h, 1)
        lin1, lin2 = torch.nn.Linear(I, Cn), torch.nn.Linear(2 * h, Cn)
        lstm = torch.nn.LSTM(L0, h, num_layers=cfg.lstm_layers, batch_first=True, bidirectional=True)
        w["lin1.weight"], w["lin1.bias"] = lin1.weight.detach().numpy(), lin1.bias.detach().numpy()
        for k, v in lstm.state_dict().items():
            w[f"lstm.{k}"] = v.detach().numpy()
        w["attention_head.weight"], w["attention_head.bias"] = att.weight.detach().numpy(), att.bias.detach().numpy()
        w["lin2.weight"], w["lin2.bias"] = lin2.weight.detach().numpy(), lin2.bias.detach().numpy()
    finally:
        torch.random.set_rng_state(state)
    # (np.ascontiguousarray would promote the 0-d gate / attention_temp to shape (1,))
    return {k: np.array(v, dtype=np.float32, copy=True, order="C") for k, v in w.items()}


def _predict(model: ClassifierLSTMDeltas, loader, device, cancel_event=None):
    actual, pred = [], []
    for d, l in loader:
        if cancel_event is not None and cancel_event.is_set():
            break
        if d.numel() == 0:
            continue
        logits, _ = model(d.to(device).float())
        actual.extend(np.asarray(l.cpu().numpy()).tolist())
        pred.extend(logits.argmax(1).cpu().numpy().tolist())
    return actual, pred


def train_lstm_model(train_set, test_set, seq_len: int, behaviors: list, cancel_event, batch_size=512, lr=1e-4,
                     epochs=10, device=None, class_weights=None, patience=3, progress_callback=None,
                     optimization_target="weighted avg", weight_decay=0.0, label_smoothing=0.0, lstm_hidden_size=64,
                     lstm_layers=1, seed: int = 0, in_features: int = 768, log=print):
    """Same signature, control flow and return value as backend/cbas.py:1274-1422 (plus ``seed`` for the
    dropout stream / shuffling / initialisation, ``in_features`` for non-768 encoders, and ``log``)."""
    from sklearn.metrics import classification_report, confusion_matrix

    if len(train_set) == 0:
        return None, None, -1
    device = torch.device(device) if device is not None else torch.device("cuda")
    if device.type != "cuda":
        raise RuntimeError("cbas_amd.train.train_lstm_model runs on a GPU device only")
    gen = torch.Generator()
    gen.manual_seed(int(seed))
    train_loader = torch.utils.data.DataLoader(train_set, batch_size, shuffle=True, collate_fn=collate_fn, num_workers=0,
                                               drop_last=False, generator=gen)
    test_loader = (torch.utils.data.DataLoader(test_set, batch_size, shuffle=False, collate_fn=collate_fn, num_workers=0)
                   if test_set is not None and len(test_set) > 0 else None)
    cfg = HeadConfig(in_features=in_features, out_features=len(behaviors), seq_len=seq_len,
                     lstm_hidden_size=lstm_hidden_size, lstm_layers=lstm_layers)
    trainer = HeadTrainer(cfg, initial_head_weights(cfg, seed), device, lr=lr, weight_decay=weight_decay,
                          label_smoothing=label_smoothing, class_weights=class_weights, max_batch=batch_size, seed=seed)
    log(f"--- Training Trial Hyperparameters ---\n  Learning Rate: {lr}\n  Weight Decay: {weight_decay}\n"
        f"  Label Smoothing: {label_smoothing}\n  LSTM Hidden Size: {lstm_hidden_size}\n  LSTM Layers: {lstm_layers}")

    def eval_model() -> ClassifierLSTMDeltas:
        m = ClassifierLSTMDeltas(in_features, len(behaviors), seq_len=seq_len, lstm_hidden_size=lstm_hidden_size,
                                 lstm_layers=lstm_layers)
        m.load_state_dict(trainer.weights())
        return m.to(device).eval()

    labels_range = list(range(len(behaviors)))

    def score(loader, cancel=None):
        """(sklearn report dict, confusion matrix) of the current parameters on one loader; ({}, empty) for no data."""
        model = eval_model()
        try:
            actual, predicted = _predict(model, loader, device, cancel)
        finally:
            model.close()
        if not actual:
            return {}, np.array([])
        report = classification_report(actual, predicted, target_names=behaviors, output_dict=True, zero_division=0,
                                       labels=labels_range)
        return report, confusion_matrix(actual, predicted, labels=labels_range)

    def f1_of(report) -> float:
        return report.get(optimization_target, {}).get("f1-score", -1.0)

    # selection state: the parameters of the best validation epoch so far, and how long it has been since
    best = {"f1": -1.0, "weights": None, "epoch": -1}
    stale_epochs = 0
    epoch_reports = []
    try:
        for epoch in range(epochs):
            if cancel_event is not None and cancel_event.is_set():
                return None, epoch_reports, best["epoch"]
            if progress_callback:
                progress_callback(f"Training Epoch {epoch + 1}/{epochs}...")
            # one pass over the shuffled windows: forward, loss, backward and Adam all inside cbas_head_train_step
            n_batches = len(train_loader)
            for i, (windows, labels) in enumerate(train_loader):
                if cancel_event is not None and cancel_event.is_set():
                    break
                if windows.numel() == 0:
                    continue
                loss = trainer.step(windows.float(), labels, want_loss=(i % 50 == 0))
                if loss is not None:
                    print(f"[Epoch {epoch + 1}/{epochs} Batch {i}/{n_batches}] Loss: {loss[0]:.4f}")
            train_report, train_cm = score(train_loader)
            if not train_report:                        # nothing could be scored (every sample failed to load)
                stale_epochs += 1
                if stale_epochs >= patience:
                    break
                continue
            val_report, val_cm = score(test_loader, cancel_event) if test_loader else ({}, np.array([]))
            epoch_reports.append(PerformanceReport(train_report, train_cm, val_report, val_cm))
            val_f1 = f1_of(val_report)
            val_str = f"{val_f1:.4f}" if test_loader else "N/A"
            if progress_callback:
                progress_callback(f"Epoch {epoch + 1} Val F1: {val_str}")
            print(f"--- Epoch {epoch + 1} | Train F1: {f1_of(train_report):.4f} | Val F1: {val_str} ({optimization_target}) ---")
            if val_f1 > best["f1"]:
                best.update(f1=val_f1, weights=trainer.weights(), epoch=epoch)
                stale_epochs = 0
            else:
                stale_epochs += 1
            if test_loader and stale_epochs >= patience:
                log(f"Early stopping triggered at epoch {epoch + 1}.")
                break
        if best["weights"] is None and epochs > 0 and not test_loader:      # no validation set: keep the last epoch
            best.update(weights=trainer.weights(), epoch=epochs - 1)
    finally:
        trainer.close()
    best_state, best_epoch = best["weights"], best["epoch"]
    if best_state:
        final_model = ClassifierLSTMDeltas(in_features, len(behaviors), seq_len=seq_len, lstm_hidden_size=lstm_hidden_size,
                                           lstm_layers=lstm_layers)
        final_model.load_state_dict(best_state)
        return final_model.to(device).eval(), epoch_reports, best_epoch
    return None, None, -1

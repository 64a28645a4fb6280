"""ORACLE (test infrastructure, not product code): CPU restatement of one optimisation step of the
reference's head training loop ``train_lstm_model`` (backend/cbas.py:1274-1422):

  * training-mode forward of ``ClassifierLSTMDeltas`` (backend/classifier_head.py:57-172; Dropout
    0.1 after each bottleneck GELU :72-76, Dropout(dropout_p=0.15) after lin0's GELU :83-87),
  * loss = CrossEntropy(weight, label_smoothing) + sum(off_diagonal(cov(latent))^2)
    (cbas.py:1311, :1336-1346),
  * Adam with the ``gate`` parameter in its own group with weight_decay 1e-3 (cbas.py:1305-1308),
    torch.optim.Adam semantics (L2 weight decay added to the gradient, bias-corrected moments).

The forward is restated with torch CPU tensor ops so that autograd provides the gradients the HIP
backward is checked against; nothing here is imported from the reference.  Dropout cannot follow the
reference's global torch RNG, so both this oracle and the HIP path draw their keep-masks from the same
counter-based hash (``dropout_keep``); tests/golden/make_goldens.py runs the REFERENCE module with its
nn.Dropout layers fed those same masks to pin this file (tests/golden/head_train_*.npz).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

P_BOTTLENECK = 0.1        # classifier_head.py:72-76
P_LIN0 = 0.15             # classifier_head.py:64 (dropout_p), :83-87
GATE_WEIGHT_DECAY = 1e-3  # cbas.py:1307
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix64(z: np.ndarray) -> np.ndarray:
    """splitmix64 step + finaliser on uint64 arrays (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def dropout_threshold(p: float) -> int:
    """keep  <=>  top-24-bit hash >= floor(p * 2^24)."""
    return int(math.floor(p * 16777216.0))


def dropout_keep(seed: int, step: int, stream: int, n: int, p: float) -> np.ndarray:
    """Keep-mask (bool, n elements) of dropout `stream` (0..2 bottlenecks cls/delta/acc, 3 lin0) at
    optimisation step `step`; element index = row-major index of the dropout's input tensor."""
    with np.errstate(over="ignore"):
        key = _mix64(np.uint64(seed) ^ _mix64(np.uint64(step * 4 + stream)))
        h = _mix64((key + np.arange(n, dtype=np.uint64)) & _M64)
    return (h >> np.uint64(40)).astype(np.int64) >= dropout_threshold(p)


def make_masks(seed: int, step: int, B: int, T: int, bott: int, lin0: int) -> Dict[str, torch.Tensor]:
    m = {}
    for s, name in enumerate(("cls", "delta", "acc")):
        m[name] = torch.from_numpy(dropout_keep(seed, step, s, B * T * bott, P_BOTTLENECK).reshape(B, T, bott))
    m["lin0"] = torch.from_numpy(dropout_keep(seed, step, 3, B * T * lin0, P_LIN0).reshape(B, T, lin0))
    return m


def _drop(x: torch.Tensor, keep: Optional[torch.Tensor], p: float) -> torch.Tensor:
    """nn.Dropout in training mode with a given keep-mask: x * keep / (1 - p)."""
    if keep is None:
        return x
    return x * keep.to(x.dtype) / (1.0 - p)


def robust_deltas(x: torch.Tensor, alpha: float):
    """classifier_head.py:102-117."""
    B, T, C = x.shape
    rows = [x[:, 0]]
    for t in range(1, T):
        rows.append(torch.lerp(rows[-1], x[:, t], alpha))
    s = torch.stack(rows, dim=1)
    if T >= 3:
        padded = torch.cat([s[:, 2:3], s[:, 1:2], s], dim=1)
    else:
        padded = torch.cat([s[:, 0:1], s[:, 0:1], s], dim=1)
    dx = padded[:, 1:] - padded[:, :-1]
    ddx = dx[:, 1:] - dx[:, :-1]
    return s, dx[:, 1:], ddx


def lstm_direction(x, w_ih, w_hh, b_ih, b_hh, reverse: bool):
    """nn.LSTM direction, gate order i,f,g,o, zero initial state (classifier_head.py:100,133)."""
    B, T, _ = x.shape
    hd = w_hh.shape[1]
    h = x.new_zeros(B, hd)
    c = x.new_zeros(B, hd)
    out = [None] * T
    for t in (range(T - 1, -1, -1) if reverse else range(T)):
        g = x[:, t] @ w_ih.T + b_ih + h @ w_hh.T + b_hh
        i_g, f_g = torch.sigmoid(g[:, :hd]), torch.sigmoid(g[:, hd:2 * hd])
        g_g, o_g = torch.tanh(g[:, 2 * hd:3 * hd]), torch.sigmoid(g[:, 3 * hd:])
        c = f_g * c + i_g * g_g
        h = o_g * torch.tanh(c)
        out[t] = h
    return torch.stack(out, dim=1)


def forward_train(x: torch.Tensor, w: Dict[str, torch.Tensor], seq_len: int = 31, sw: int = 5, alpha: float = 0.3,
                  masks: Optional[Dict[str, torch.Tensor]] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """classifier_head.py:150-172 in training mode.  x (B,T,I) -> (final_logits (B,C), latent (B,2h))."""
    hsl = seq_len // 2
    s, d, a = robust_deltas(x, alpha)
    L = s.shape[1]
    l, r = max(0, hsl - sw), min(L, hsl + sw + 1)
    assert l < r, "empty centre window"
    linear_logits = (s[:, l:r] @ w["lin1.weight"].T + w["lin1.bias"]).mean(dim=1)          # :119-129

    def bott(stream, name):                                                               # :155-160
        y = F.gelu(stream @ w[f"{name}_bottleneck.0.weight"].T + w[f"{name}_bottleneck.0.bias"])
        y = _drop(y, None if masks is None else masks[name], P_BOTTLENECK)
        return F.layer_norm(y, (y.shape[-1],), w[f"{name}_ln.weight"], w[f"{name}_ln.bias"], 1e-5)

    streams = [bott(s, "cls"), bott(d, "delta")]
    if "acc_bottleneck.0.weight" in w:                                                    # use_acceleration (:74-84, :158-162)
        streams.append(bott(a, "acc"))
    aug = torch.cat(streams, dim=-1)
    xl = F.gelu(aug @ w["lin0.0.weight"].T + w["lin0.0.bias"])                             # :164
    xl = _drop(xl, None if masks is None else masks["lin0"], P_LIN0)
    xl = xl - xl.mean(dim=1, keepdim=True)                                                 # :166-167

    out, layer = xl, 0                                                                     # :131-148
    while f"lstm.weight_ih_l{layer}" in w:
        p = f"lstm.{{}}_l{layer}"
        q = p + "_reverse"
        fwd = lstm_direction(out, w[p.format("weight_ih")], w[p.format("weight_hh")],
                             w[p.format("bias_ih")], w[p.format("bias_hh")], False)
        bwd = lstm_direction(out, w[q.format("weight_ih")], w[q.format("weight_hh")],
                             w[q.format("bias_ih")], w[q.format("bias_hh")], True)
        out = torch.cat([fwd, bwd], dim=-1)
        layer += 1
    centre = out[:, l:r]
    temp = F.softplus(w["attention_temp"]) + 1e-3
    scores = (centre @ w["attention_head.weight"].T + w["attention_head.bias"]).squeeze(-1) / temp
    aw = torch.softmax(scores, dim=1).unsqueeze(-1)
    latent = (aw * centre).sum(dim=1)
    lstm_logits = latent @ w["lin2.weight"].T + w["lin2.bias"]
    final = torch.lerp(linear_logits, lstm_logits, torch.sigmoid(w["gate"]))               # :171
    return final, latent


def off_diagonal_sq_sum(latent: torch.Tensor) -> torch.Tensor:
    """cbas.py:1339-1344: sum of squared off-diagonal entries of the batch covariance of `latent`."""
    if latent.ndim != 2 or latent.shape[0] <= 1:
        return latent.new_zeros(())
    rc = latent - latent.mean(dim=0)
    cov = (rc.T @ rc) / (rc.shape[0] - 1)
    return (cov ** 2).sum() - (torch.diagonal(cov) ** 2).sum()


def cross_entropy(logits: torch.Tensor, labels: torch.Tensor, class_weights: Optional[torch.Tensor],
                  label_smoothing: float) -> torch.Tensor:
    """nn.CrossEntropyLoss(weight=w, label_smoothing=eps), reduction 'mean' (cbas.py:1311), written out:
    sum_i [(1-eps) w[y_i] nll_i(y_i) + eps/C sum_c w[c] nll_i(c)] / sum_i w[y_i]."""
    logp = torch.log_softmax(logits, dim=1)
    C = logits.shape[1]
    w = class_weights if class_weights is not None else logits.new_ones(C)
    wy = w[labels]
    nll = -(logp[torch.arange(len(labels)), labels]) * wy
    smooth = -(logp * w[None, :]).sum(dim=1)
    return ((1.0 - label_smoothing) * nll + (label_smoothing / C) * smooth).sum() / wy.sum()


def loss_and_grads(x: np.ndarray, labels: np.ndarray, weights: Dict[str, np.ndarray], seq_len: int = 31,
                   class_weights: Optional[np.ndarray] = None, label_smoothing: float = 0.0,
                   masks: Optional[Dict[str, torch.Tensor]] = None, dtype=torch.float32):
    """One forward + backward.  Returns (loss, ce, cov, final_logits, latent, grads-by-name)."""
    w = {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=True) for k, v in weights.items()}
    xt = torch.tensor(x, dtype=dtype)
    final, latent = forward_train(xt, w, seq_len, masks=masks)
    cw = None if class_weights is None else torch.tensor(class_weights, dtype=dtype)
    ce = cross_entropy(final, torch.as_tensor(labels, dtype=torch.long), cw, label_smoothing)
    cov = off_diagonal_sq_sum(latent)
    loss = ce + cov
    loss.backward()
    grads = {k: (v.grad.detach().numpy().astype(np.float64) if v.grad is not None else np.zeros(v.shape)) for k, v in w.items()}
    return (float(loss), float(ce), float(cov), final.detach().numpy(), latent.detach().numpy(), grads)


class Adam:
    """torch.optim.Adam as the reference configures it (cbas.py:1305-1308): betas (0.9, 0.999), eps 1e-8,
    L2 weight decay added to the gradient; `gate` has its own weight decay."""

    def __init__(self, weights: Dict[str, np.ndarray], lr: float, weight_decay: float = 0.0):
        self.lr, self.wd, self.t = lr, weight_decay, 0
        self.m = {k: np.zeros_like(np.asarray(v, np.float64)) for k, v in weights.items()}
        self.v = {k: np.zeros_like(np.asarray(v, np.float64)) for k, v in weights.items()}

    def step(self, weights: Dict[str, np.ndarray], grads: Dict[str, np.ndarray], dtype=np.float32) -> Dict[str, np.ndarray]:
        self.t += 1
        b1, b2, eps = 0.9, 0.999, 1e-8
        out = {}
        for k, p in weights.items():
            p = np.asarray(p, np.float64)
            g = np.asarray(grads[k], np.float64) + (GATE_WEIGHT_DECAY if k == "gate" else self.wd) * p
            self.m[k] = b1 * self.m[k] + (1 - b1) * g
            self.v[k] = b2 * self.v[k] + (1 - b2) * g * g
            denom = np.sqrt(self.v[k]) / math.sqrt(1 - b2 ** self.t) + eps
            out[k] = (p - (self.lr / (1 - b1 ** self.t)) * self.m[k] / denom).astype(dtype)
        return out


def train_steps(x_batches, label_batches, weights: Dict[str, np.ndarray], n_steps: int, lr: float, seed: int,
                seq_len: int = 31, weight_decay: float = 0.0, class_weights=None, label_smoothing: float = 0.0,
                bott: int = 128, lin0: int = 256, dropout: bool = True):
    """Run n_steps optimisation steps (step s uses batch s % len(batches)); returns (weights, losses)."""
    w = {k: np.asarray(v, np.float32) for k, v in weights.items()}
    opt = Adam(w, lr, weight_decay)
    losses = []
    for s in range(n_steps):
        x, y = x_batches[s % len(x_batches)], label_batches[s % len(label_batches)]
        masks = make_masks(seed, s, x.shape[0], x.shape[1], bott, lin0) if dropout else None
        loss, _, _, _, _, g = loss_and_grads(x, y, w, seq_len, class_weights, label_smoothing, masks)
        losses.append(loss)
        w = opt.step(w, g)
    return w, losses

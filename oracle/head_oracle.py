"""ORACLE (test infrastructure, not product code): CPU float32 restatement of the reference's
``ClassifierLSTMDeltas.forward`` (backend/classifier_head.py:57-172) and of the sliding-window
driver ``infer_file`` (backend/cbas.py:458-572), *without* the per-frame-projection shortcut the
HIP path uses: every window is materialised and pushed through the head independently, exactly as
the reference does.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this.

Pinned against the reference module itself, imported in the build container
(tests/golden/make_goldens.py -> tests/golden/head_*.npz).
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np

from .vit_oracle import gelu_erf, layer_norm

F32 = np.float32


def _sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(F32)


def robust_deltas(x: np.ndarray, alpha: float) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """classifier_head.py:102-117.  x (B,T,C) -> (smooth, delta, accel), each (B,T,C)."""
    x = x.astype(F32)
    B, T, C = x.shape
    s = np.zeros_like(x)
    s[:, 0] = x[:, 0]
    a = F32(alpha)
    for t in range(1, T):                       # torch.lerp(prev, x_t, alpha), alpha < 0.5 branch
        s[:, t] = s[:, t - 1] + a * (x[:, t] - s[:, t - 1])
    if T >= 3:                                   # F.pad(..., (2, 0), 'reflect') along time
        padded = np.concatenate([s[:, 2:3], s[:, 1:2], s], axis=1)
    else:                                        # 'replicate'
        padded = np.concatenate([s[:, 0:1], s[:, 0:1], s], axis=1)
    dx = padded[:, 1:] - padded[:, :-1]
    ddx = dx[:, 1:] - dx[:, :-1]
    return s, dx[:, 1:], ddx


def lstm_direction(x: np.ndarray, w_ih, w_hh, b_ih, b_hh, reverse: bool) -> np.ndarray:
    """One direction of nn.LSTM(batch_first=True), gate order i,f,g,o, zero initial state
    (classifier_head.py:100,133).  x (B,T,I) -> (B,T,h)."""
    B, T, _ = x.shape
    hdim = w_hh.shape[1]
    h = np.zeros((B, hdim), F32)
    c = np.zeros((B, hdim), F32)
    out = np.zeros((B, T, hdim), F32)
    steps = range(T - 1, -1, -1) if reverse else range(T)
    for t in steps:
        g = (x[:, t] @ w_ih.T + b_ih + h @ w_hh.T + b_hh).astype(F32)
        i_g = _sigmoid(g[:, 0 * hdim:1 * hdim])
        f_g = _sigmoid(g[:, 1 * hdim:2 * hdim])
        g_g = np.tanh(g[:, 2 * hdim:3 * hdim]).astype(F32)
        o_g = _sigmoid(g[:, 3 * hdim:4 * hdim])
        c = (f_g * c + i_g * g_g).astype(F32)
        h = (o_g * np.tanh(c)).astype(F32)
        out[:, t] = h
    return out


def head_forward(x: np.ndarray, w: Dict[str, np.ndarray], seq_len: int = 31, sw: int = 5,
                 alpha: float = 0.3) -> Tuple[np.ndarray, np.ndarray]:
    """classifier_head.py:150-172.  x (B,T,I) float32 -> (final_logits (B,C), latent (B,2h))."""
    x = x.astype(F32)
    hsl = seq_len // 2
    s, d, a = robust_deltas(x, alpha)

    # forward_linear, :119-129
    L = s.shape[1]
    l, r = max(0, hsl - sw), min(L, hsl + sw + 1)
    if l >= r:
        idx = min(max(0, L // 2), L - 1) if L > 0 else 0
        linear_logits = s[:, idx] @ w["lin1.weight"].T + w["lin1.bias"]
    else:
        linear_logits = (s[:, l:r] @ w["lin1.weight"].T + w["lin1.bias"]).mean(axis=1, dtype=F32)

    # bottlenecks + LayerNorm, :155-160 (Dropout is identity in eval)
    def bott(stream, name):
        y = gelu_erf(stream @ w[f"{name}_bottleneck.0.weight"].T + w[f"{name}_bottleneck.0.bias"])
        return layer_norm(y, w[f"{name}_ln.weight"], w[f"{name}_ln.bias"], 1e-5)

    # use_acceleration=False (:74-84, :158-162): no acc_* parameters, two streams
    streams = [bott(s, "cls"), bott(d, "delta")] + ([bott(a, "acc")] if "acc_bottleneck.0.weight" in w else [])
    aug = np.concatenate(streams, axis=-1)

    # lin0 + centring, :164-167
    xl = gelu_erf(aug @ w["lin0.0.weight"].T + w["lin0.0.bias"])
    xl = (xl - xl.mean(axis=1, keepdims=True, dtype=F32)).astype(F32)

    # forward_lstm, :131-148
    # nn.LSTM(num_layers=k, bidirectional=True): layer l > 0 consumes cat(fwd, bwd) of layer l-1
    out, layer = xl, 0
    while f"lstm.weight_ih_l{layer}" in w:
        p = f"lstm.{{}}_l{layer}"
        fwd = lstm_direction(out, w[p.format("weight_ih")], w[p.format("weight_hh")],
                             w[p.format("bias_ih")], w[p.format("bias_hh")], reverse=False)
        r = p + "_reverse"
        bwd = lstm_direction(out, w[r.format("weight_ih")], w[r.format("weight_hh")],
                             w[r.format("bias_ih")], w[r.format("bias_hh")], reverse=True)
        out = np.concatenate([fwd, bwd], axis=-1)
        layer += 1
    L = out.shape[1]
    l, r = max(0, hsl - sw), min(L, hsl + sw + 1)
    if l >= r:
        idx = min(max(0, L // 2), L - 1) if L > 0 else 0
        latent = out[:, idx]
        lstm_logits = latent @ w["lin2.weight"].T + w["lin2.bias"]
    else:
        centre = out[:, l:r]
        t_raw = float(w["attention_temp"])
        temp = F32(math.log1p(math.exp(t_raw)) + 1e-3)                 # F.softplus + 1e-3
        scores = (centre @ w["attention_head.weight"].T + w["attention_head.bias"])[..., 0] / temp
        scores = scores - scores.max(axis=1, keepdims=True)
        aw = np.exp(scores, dtype=F32)
        aw = aw / aw.sum(axis=1, keepdims=True, dtype=F32)
        latent = (aw[..., None] * centre).sum(axis=1, dtype=F32)
        lstm_logits = latent @ w["lin2.weight"].T + w["lin2.bias"]

    g = _sigmoid(np.asarray(w["gate"], F32))
    final = (linear_logits + g * (lstm_logits - linear_logits)).astype(F32)   # torch.lerp, weight<0.5 form
    if float(g) >= 0.5:                                                        # torch.lerp's other branch
        final = (lstm_logits - (lstm_logits - linear_logits) * (F32(1) - g)).astype(F32)
    return final, latent.astype(F32)


def softmax_T(logits: np.ndarray, temperature: float) -> np.ndarray:
    """backend/cbas.py:545-546: softmax(logits / max(1e-3, T), dim=1)."""
    z = (logits / F32(max(1e-3, temperature))).astype(F32)
    z = z - z.max(axis=1, keepdims=True)
    e = np.exp(z, dtype=F32)
    return (e / e.sum(axis=1, keepdims=True, dtype=F32)).astype(F32)


def infer_windows(cls_f16: np.ndarray, seq_len: int, chunk: int = 20000) -> np.ndarray:
    """The window materialisation of backend/cbas.py:497-536, as an index computation: returns
    (N, seq_len) source-row indices into ``cls`` such that window i == cls[idx[i]].

    The reference reads [start-half, end+half) and replicate-pads only at the video's two ends, so
    every window is rows i-half .. i+half clamped to [0, N-1]; the chunking does not change this
    (checked against the literal chunk loop in tests/test_oracle_golden.py)."""
    n = cls_f16.shape[0]
    half = seq_len // 2
    idx = np.arange(n)[:, None] + np.arange(-half, seq_len - half)[None, :]
    return np.clip(idx, 0, n - 1)


def infer_file_literal(cls_f16: np.ndarray, w: Dict[str, np.ndarray], seq_len: int,
                       temperature: float = 1.0, chunk: int = 20000, batch: int = 512) -> np.ndarray:
    """Literal restatement of the loop at backend/cbas.py:497-551 (halo read, edge replicate padding,
    512-window batches).  cls_f16 (N,D) float16 -> probs (N,C) float32."""
    total = cls_f16.shape[0]
    half = seq_len // 2
    all_probs = []
    for start in range(0, total, chunk):
        end = min(start + chunk, total)
        read_start = max(0, start - half)
        read_end = min(total, end + half)
        t = cls_f16[read_start:read_end].astype(F32)
        if start < half:
            pad = half - start
            if pad > 0:
                t = np.concatenate([np.repeat(t[0:1], pad, axis=0), t], axis=0)
        if end > total - half:
            pad = half - (total - end)
            if pad > 0:
                t = np.concatenate([t, np.repeat(t[-1:], pad, axis=0)], axis=0)
        n_targets = end - start
        buf = []
        for i in range(n_targets):
            buf.append(t[i:i + seq_len])
            if len(buf) >= batch or i == n_targets - 1:
                logits, _ = head_forward(np.stack(buf), w, seq_len)
                all_probs.append(softmax_T(logits, temperature))
                buf = []
    return np.concatenate(all_probs, axis=0)

"""ORACLE (test infrastructure, not product code): the precision-2 (MX-fp8) arithmetic of the encoder restated on
the CPU.  The reference has no fp8 path - its own low-precision site is the fp16 autocast at backend/cbas.py:433-434 -
so this mode is held to label parity only (SURVEY.md section 7 "Hard parts", BASELINE.json configs[4]); what this file pins
is that the HIP kernels implement the *stated* quantisation: OCP e4m3 elements (4 exponent bits, 3 mantissa bits,
max 448, round to nearest even) with one E8M0 power-of-two scale per 32 consecutive k-elements, the smallest scale that
does not clip, applied to both operands of the q/k/v, o_proj, up and down projections (``[tf]`` :307-309, :331,
:356-357), fp32 accumulation, everything else as vit_oracle.py.

Only ``tests/`` may import this.
"""
from __future__ import annotations

from typing import Dict

import numpy as np

from . import vit_oracle as V

F32 = np.float32


def e4m3_decode(b: np.ndarray) -> np.ndarray:
    """OCP FP8 E4M3 byte -> float32 (bias 7, subnormals at exponent field 0; 0x7f / 0xff are NaN and never produced)."""
    b = b.astype(np.int32)
    s, e, m = b >> 7, (b >> 3) & 15, b & 7
    mag = np.where(e == 0, m / 8.0 * 2.0 ** -6, (1.0 + m / 8.0) * 2.0 ** (e - 7.0))
    return np.where(s == 1, -mag, mag).astype(F32)


def e4m3_round(x: np.ndarray) -> np.ndarray:
    """Round to the nearest e4m3 value (ties to even), |x| <= 448 assumed."""
    ax = np.abs(x).astype(np.float64)
    e = np.clip(np.floor(np.log2(np.maximum(ax, 1e-300))), -6, 8)
    step = 2.0 ** (e - 3)
    q = np.minimum(np.round(ax / step) * step, 448.0)
    return (np.sign(x) * q).astype(F32)


def mx_scale_exp(amax: np.ndarray) -> np.ndarray:
    """Biased E8M0 exponent byte of the smallest power of two s with amax / s <= 448 (common.h mx_scale_byte)."""
    with np.errstate(divide="ignore"):
        e = np.ceil(np.log2(amax.astype(np.float64) / 448.0))
    e = np.where(amax > 0, e, -127.0)
    return np.clip(e + 127.0, 0, 253).astype(np.int32)


def mx_quant(x: np.ndarray):
    """x [..., K] -> (dequantised values, e4m3 element values, scale bytes [..., K/32])."""
    sh = x.shape
    xb = x.reshape(-1, sh[-1] // 32, 32).astype(np.float64)
    sb = mx_scale_exp(np.abs(xb).max(-1))
    s = 2.0 ** (sb.astype(np.float64) - 127.0)
    q = e4m3_round((xb / s[..., None]).astype(F32))
    deq = (q.astype(np.float64) * s[..., None]).astype(F32).reshape(sh)
    return deq, q.reshape(sh), sb.reshape(sh[:-1] + (sh[-1] // 32,))


def _q(x):
    return mx_quant(x)[0]


def vit_forward_mx(pixels: np.ndarray, w: Dict[str, np.ndarray], cfg) -> np.ndarray:
    """vit_oracle.vit_forward with the operands of the eight projection GEMMs of every layer MX-fp8 quantised."""
    x = V.embeddings(pixels.astype(F32), w, cfg.patch_size)
    H, W_ = pixels.shape[2:]
    nh = cfg.num_attention_heads
    cos, sin = V.rope_cos_sin(H // cfg.patch_size, W_ // cfg.patch_size, cfg.hidden_size // nh, cfg.rope_theta)
    for i in range(cfg.num_hidden_layers):
        pre = f"model.layer.{i}."
        a = pre + "attention."
        B, T, D = x.shape
        h = _q(V.layer_norm(x, w[pre + "norm1.weight"], w[pre + "norm1.bias"], cfg.layer_norm_eps))
        q = h @ _q(w[a + "q_proj.weight"]).T + w[a + "q_proj.bias"]
        k = h @ _q(w[a + "k_proj.weight"]).T
        v = h @ _q(w[a + "v_proj.weight"]).T + w[a + "v_proj.bias"]
        q, k, v = (t.reshape(B, T, nh, D // nh).transpose(0, 2, 1, 3) for t in (q, k, v))
        npf = T - cos.shape[0]
        qp, kp = q[:, :, npf:], k[:, :, npf:]
        q = np.concatenate([q[:, :, :npf], qp * cos + V._rotate_half(qp) * sin], axis=2)
        k = np.concatenate([k[:, :, :npf], kp * cos + V._rotate_half(kp) * sin], axis=2)
        s = (q @ k.transpose(0, 1, 3, 2)) * F32((D // nh) ** -0.5)
        s = s - s.max(-1, keepdims=True)
        p = np.exp(s)
        p = p / p.sum(-1, keepdims=True)
        ctx = _q((p @ v).transpose(0, 2, 1, 3).reshape(B, T, D).astype(F32))
        o = ctx @ _q(w[a + "o_proj.weight"]).T + w[a + "o_proj.bias"]
        x = (o * w[pre + "layer_scale1.lambda1"] + x).astype(F32)
        h = _q(V.layer_norm(x, w[pre + "norm2.weight"], w[pre + "norm2.bias"], cfg.layer_norm_eps))
        u = _q(V.gelu_erf(h @ _q(w[pre + "mlp.up_proj.weight"]).T + w[pre + "mlp.up_proj.bias"]))
        d = u @ _q(w[pre + "mlp.down_proj.weight"]).T + w[pre + "mlp.down_proj.bias"]
        x = (d * w[pre + "layer_scale2.lambda1"] + x).astype(F32)
    return V.layer_norm(x, w["norm.weight"], w["norm.bias"], cfg.layer_norm_eps).astype(F32)


def encode_frames_mx(frames_u8: np.ndarray, w, cfg, batch: int = 4) -> np.ndarray:
    g = V.preprocess_green(frames_u8)
    outs = []
    for i in range(0, g.shape[0], batch):
        outs.append(vit_forward_mx(np.repeat(g[i:i + batch, None], 3, axis=1), w, cfg)[:, 0, :])
    return np.concatenate(outs, axis=0)

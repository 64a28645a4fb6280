"""ORACLE (test infrastructure, not product code): the DINOv3 ViT forward of ``vit_oracle.py`` restated with
torch CPU float32 ops - the arithmetic library the reference's own CPU path runs on (``transformers`` modules are
torch ``nn.Linear`` / ``nn.LayerNorm`` / SDPA calls: ``[tf]`` = transformers/models/dinov3_vit/modeling_dinov3_vit.py).
It exists for ``bench.py``'s ``cpu_baseline`` leg: the numpy restatement spends its time in single-threaded ufuncs
(exp, erf, layer-norm reductions) that torch's CPU kernels vectorise and thread, so the numpy number understated
the reference's CPU path.  Checked against the goldens made from the reference in
tests/test_oracle_golden.py.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F


def to_torch(w: Dict[str, np.ndarray]) -> Dict[str, torch.Tensor]:
    return {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)) for k, v in w.items()}


def preprocess_green(frames_u8: np.ndarray) -> torch.Tensor:
    """backend/cbas.py:431: ``torch.from_numpy(frames_np[:, :, :, 1] / 255.0).float()`` (float64 divide, then f32)."""
    return torch.from_numpy(frames_u8[:, :, :, 1] / 255.0).float()


def rope_cos_sin(n_h: int, n_w: int, head_dim: int, theta: float):
    """[tf]:96-121 patch-centre coordinates, :153-200 angles / tile(2) / cos, sin in float32."""
    ch = torch.arange(0.5, n_h, dtype=torch.float32) / n_h
    cw = torch.arange(0.5, n_w, dtype=torch.float32) / n_w
    coords = torch.stack(torch.meshgrid(ch, cw, indexing="ij"), dim=-1).flatten(0, 1)
    coords = 2.0 * coords - 1.0
    inv_freq = 1.0 / (theta ** torch.arange(0, 1, 4 / head_dim, dtype=torch.float32))
    angles = 2 * math.pi * coords[:, :, None] * inv_freq[None, None, :]
    angles = angles.flatten(1, 2).tile(2)
    return torch.cos(angles), torch.sin(angles)


def _rotate_half(x: torch.Tensor) -> torch.Tensor:
    """[tf]:203-207."""
    h = x.shape[-1] // 2
    return torch.cat([-x[..., h:], x[..., :h]], dim=-1)


def embeddings(pixels: torch.Tensor, w, patch: int) -> torch.Tensor:
    """[tf]:75-92: Conv2d(k = s = patch) -> flatten(2).transpose(1, 2) -> cat[cls, registers, patches]."""
    B = pixels.shape[0]
    pe = F.conv2d(pixels, w["embeddings.patch_embeddings.weight"], w["embeddings.patch_embeddings.bias"], stride=patch)
    pe = pe.flatten(2).transpose(1, 2)
    D = pe.shape[-1]
    cls = w["embeddings.cls_token"].reshape(1, 1, D).expand(B, -1, -1)
    reg = w["embeddings.register_tokens"].reshape(1, -1, D).expand(B, -1, -1)
    return torch.cat([cls, reg, pe], dim=1)


def attention(x: torch.Tensor, w, pre: str, n_heads: int, cos, sin) -> torch.Tensor:
    """[tf]:294-334 with the eager attention of :210-234; RoPE on the patch rows only (:238-268)."""
    B, T, D = x.shape
    hd = D // n_heads
    q = F.linear(x, w[pre + "q_proj.weight"], w[pre + "q_proj.bias"])
    k = F.linear(x, w[pre + "k_proj.weight"], w.get(pre + "k_proj.bias"))
    v = F.linear(x, w[pre + "v_proj.weight"], w[pre + "v_proj.bias"])
    q, k, v = (t.view(B, T, n_heads, hd).transpose(1, 2) for t in (q, k, v))
    n_prefix = T - cos.shape[0]
    qp, kp = q[:, :, n_prefix:], k[:, :, n_prefix:]
    q = torch.cat([q[:, :, :n_prefix], qp * cos + _rotate_half(qp) * sin], dim=2)
    k = torch.cat([k[:, :, :n_prefix], kp * cos + _rotate_half(kp) * sin], dim=2)
    p = torch.softmax((q @ k.transpose(-1, -2)) * hd ** -0.5, dim=-1)
    o = (p @ v).transpose(1, 2).reshape(B, T, D)
    return F.linear(o, w[pre + "o_proj.weight"], w[pre + "o_proj.bias"])


def layer(x: torch.Tensor, w, i: int, n_heads: int, eps: float, cos, sin) -> torch.Tensor:
    """[tf]:419-445: x += lambda1 * Attn(LN1(x)); x += lambda2 * MLP(LN2(x)); MLP = [tf]:356-357 (exact-erf GELU)."""
    pre = f"model.layer.{i}."
    D = x.shape[-1]
    h = F.layer_norm(x, (D,), w[pre + "norm1.weight"], w[pre + "norm1.bias"], eps)
    x = attention(h, w, pre + "attention.", n_heads, cos, sin) * w[pre + "layer_scale1.lambda1"] + x
    h = F.layer_norm(x, (D,), w[pre + "norm2.weight"], w[pre + "norm2.bias"], eps)
    u = F.gelu(F.linear(h, w[pre + "mlp.up_proj.weight"], w[pre + "mlp.up_proj.bias"]))
    d = F.linear(u, w[pre + "mlp.down_proj.weight"], w[pre + "mlp.down_proj.bias"])
    return d * w[pre + "layer_scale2.lambda1"] + x


@torch.no_grad()
def vit_forward(pixels: torch.Tensor, w, cfg) -> torch.Tensor:
    """[tf]:523-548 ``DINOv3ViTModel.forward`` -> last_hidden_state (B,T,D) float32; pixels (B,3,H,W) float32."""
    H, W = pixels.shape[2:]
    x = embeddings(pixels, w, cfg.patch_size)
    cos, sin = rope_cos_sin(H // cfg.patch_size, W // cfg.patch_size, cfg.hidden_size // cfg.num_attention_heads,
                            cfg.rope_theta)
    for i in range(cfg.num_hidden_layers):
        x = layer(x, w, i, cfg.num_attention_heads, cfg.layer_norm_eps, cos, sin)
    return F.layer_norm(x, (x.shape[-1],), w["norm.weight"], w["norm.bias"], cfg.layer_norm_eps)


@torch.no_grad()
def encode_frames(frames_u8: np.ndarray, w_t, cfg, batch: int = 8) -> np.ndarray:
    """encode_file's arithmetic (backend/cbas.py:431-436 + DinoEncoder.forward :672-677): (N,H,W,3) uint8 ->
    CLS (N,D) float32, ``batch`` frames per model call."""
    g = preprocess_green(frames_u8)
    outs = []
    for i in range(0, g.shape[0], batch):
        px = g[i:i + batch].unsqueeze(1).repeat(1, 3, 1, 1)
        outs.append(vit_forward(px, w_t, cfg)[:, 0, :])
    return torch.cat(outs).numpy()

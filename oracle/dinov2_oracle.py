"""ORACLE (test infrastructure, not product code): CPU float32 restatement of the HF
``Dinov2WithRegistersModel`` forward pass — the encoder CBAS projects use by default
(reference backend/cbas.py:1030-1033 ``facebook/dinov2-with-registers-base``).
``[v2]`` = transformers/models/dinov2_with_registers/modeling_dinov2_with_registers.py.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this.
Pinned by tests/golden/dinov2reg_*.npz, produced by running the HF model itself
(tests/golden/make_goldens.py).

The transformer blocks are the DINOv3 ones without RoPE and with a key bias, so they are shared
with vit_oracle.py (weights renamed to the DINOv3 key names by cbas_amd.weights.canonical_encoder_weights).
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

from . import vit_oracle as V

F32 = np.float32


def _cubic_aa(x: np.ndarray, a: float = -0.5) -> np.ndarray:
    """Keys cubic convolution kernel with a = -0.5, the coefficient ATen's *antialiased* bicubic uses
    (aten/src/ATen/native/cpu/UpSampleKernel.cpp, HelperInterpCubic::aa_filter)."""
    x = np.abs(x)
    out = np.zeros_like(x)
    m1 = x < 1.0
    m2 = (x >= 1.0) & (x < 2.0)
    out[m1] = ((a + 2.0) * x[m1] - (a + 3.0)) * x[m1] * x[m1] + 1.0
    out[m2] = (((x[m2] - 5.0) * x[m2] + 8.0) * x[m2] - 4.0) * a
    return out


def aa_bicubic_matrix(in_size: int, out_size: int) -> np.ndarray:
    """(out_size, in_size) row-stochastic weights of torch's F.interpolate(mode='bicubic',
    align_corners=False, antialias=True) along one axis (separable): the kernel is stretched by the
    scale when down-sampling, evaluated at pixel centres and normalised."""
    scale = in_size / out_size
    support = 2.0 * scale if scale >= 1.0 else 2.0
    invscale = 1.0 / scale if scale >= 1.0 else 1.0
    W = np.zeros((out_size, in_size), np.float64)
    for i in range(out_size):
        center = scale * (i + 0.5)
        xmin = max(0, int(center - support + 0.5))
        xmax = min(in_size, int(center + support + 0.5))
        j = np.arange(xmin, xmax, dtype=np.float64)
        w = _cubic_aa((j - center + 0.5) * invscale)
        W[i, xmin:xmax] = w / w.sum()
    return W.astype(F32)


def interpolate_pos_embed(pos: np.ndarray, grid: int, n_h: int, n_w: int) -> np.ndarray:
    """[v2]:93-145: pos (1, 1+grid*grid, D) -> (1 + n_h*n_w, D); identity when the grids match."""
    pos = pos.reshape(1 + grid * grid, -1).astype(F32)
    if n_h == grid and n_w == grid:
        return pos
    patch = pos[1:].reshape(grid, grid, -1)
    Wh, Ww = aa_bicubic_matrix(grid, n_h), aa_bicubic_matrix(grid, n_w)
    tmp = np.einsum("xj,ijd->ixd", Ww, patch).astype(F32)           # width pass, then height pass
    out = np.einsum("yi,ixd->yxd", Wh, tmp).astype(F32)
    return np.concatenate([pos[:1], out.reshape(n_h * n_w, -1)], axis=0)


def embeddings(pixels: np.ndarray, w: Dict[str, np.ndarray], cfg) -> np.ndarray:
    """[v2]:147-170: patch conv (k = s = patch) -> cat[cls, patches] + interpolated position
    embedding -> registers inserted after the cls token.  ``w`` uses the canonical (DINOv3) key names."""
    B, C, H, Wd = pixels.shape
    p = cfg.patch_size
    nh, nw = H // p, Wd // p
    x = pixels[:, :, : nh * p, : nw * p].reshape(B, C, nh, p, nw, p)
    x = x.transpose(0, 2, 4, 1, 3, 5).reshape(B, nh * nw, C * p * p)
    wk = w["embeddings.patch_embeddings.weight"].reshape(-1, C * p * p)
    pe = (x @ wk.T + w["embeddings.patch_embeddings.bias"]).astype(F32)
    D = wk.shape[0]
    cls = np.broadcast_to(w["embeddings.cls_token"].reshape(1, 1, D), (B, 1, D))
    emb = np.concatenate([cls, pe], axis=1) + interpolate_pos_embed(w["embeddings.position_embeddings"],
                                                                     cfg.pos_embed_grid, nh, nw)[None]
    reg = np.broadcast_to(w["embeddings.register_tokens"].reshape(1, -1, D), (B, cfg.num_register_tokens, D))
    return np.concatenate([emb[:, :1], reg, emb[:, 1:]], axis=1).astype(F32)


def forward(pixels: np.ndarray, w: Dict[str, np.ndarray], cfg, taps: Optional[dict] = None) -> np.ndarray:
    """[v2]:473-505 ``Dinov2WithRegistersModel.forward`` -> last_hidden_state (B,T,D)."""
    x = embeddings(pixels.astype(F32), w, cfg)
    if taps is not None:
        taps["embeddings"] = x.copy()
    for i in range(cfg.num_hidden_layers):
        x = V.layer(x, w, i, cfg.num_attention_heads, cfg.layer_norm_eps, None, None, taps)
    return V.layer_norm(x, w["norm.weight"], w["norm.bias"], cfg.layer_norm_eps).astype(F32)

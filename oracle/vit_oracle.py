"""ORACLE (test infrastructure, not product code): CPU float32 restatement of the DINOv3 ViT
forward pass that the reference runs through ``transformers`` (pinned ``>=4.53.3`` in the
reference's requirements.txt:26; 5.15.0 installed in the build container).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this.

Pinning: the reference holds no golden vector for this path, so the oracle is
pinned against outputs of the reference's own modules run in the build container
(``tests/golden/make_goldens.py`` -> ``tests/golden/*.npz``; checked by tests/test_oracle_golden.py).

Every function cites the lines it restates; ``[tf]`` =
``transformers/models/dinov3_vit/modeling_dinov3_vit.py``.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np

try:  # exact erf; scipy is present in the image
    from scipy.special import erf as _erf
except Exception:  # pragma: no cover
    _erf = np.vectorize(math.erf, otypes=[np.float64])

F32 = np.float32


def preprocess_green(frames_u8: np.ndarray) -> np.ndarray:
    """backend/cbas.py:431: ``frames_np[:, :, :, 1] / 255.0`` (float64) ``.float()`` -> (N,H,W) f32."""
    return (frames_u8[:, :, :, 1] / 255.0).astype(F32)


def layer_norm(x: np.ndarray, w: np.ndarray, b: np.ndarray, eps: float) -> np.ndarray:
    """nn.LayerNorm (biased variance) as used at [tf]:404,410,514."""
    x = x.astype(F32)
    mu = x.mean(axis=-1, keepdims=True, dtype=F32)
    xc = x - mu
    var = (xc * xc).mean(axis=-1, keepdims=True, dtype=F32)
    return (xc / np.sqrt(var + F32(eps))) * w + b


_POOL = None


def _erf_mt(x: np.ndarray) -> np.ndarray:
    """scipy's erf ufunc is single-threaded (and releases the GIL): split large arrays over the cores
    the process may use, so the CPU baseline is not dominated by one core evaluating erf."""
    global _POOL
    if x.size < (1 << 16):
        return _erf(x)
    import os
    from concurrent.futures import ThreadPoolExecutor
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    n = max(1, min(n, int(os.environ.get("CBAS_CPU_BASELINE_THREADS", "32"))))
    if n == 1:
        return _erf(x)
    if _POOL is None or _POOL._max_workers != n:
        _POOL = ThreadPoolExecutor(max_workers=n)
    flat = np.ascontiguousarray(x).reshape(-1)
    out = np.empty_like(flat)
    bounds = np.linspace(0, flat.size, n + 1).astype(np.int64)

    def work(i):
        _erf(flat[bounds[i]:bounds[i + 1]], out=out[bounds[i]:bounds[i + 1]])
    list(_POOL.map(work, range(n)))
    return out.reshape(x.shape)


def gelu_erf(x: np.ndarray) -> np.ndarray:
    """ACT2FN['gelu'] = exact erf GELU ([tf]:354, hidden_act default 'gelu')."""
    x = x.astype(F32, copy=False)            # float32 erf, as torch's CPU GELU kernel computes it
    return (F32(0.5) * x * (F32(1.0) + _erf_mt(x * F32(1.0 / math.sqrt(2.0))))).astype(F32, copy=False)


def rope_cos_sin(n_h: int, n_w: int, head_dim: int, theta: float):
    """[tf]:96-121 (patch-centre coordinates) and :153-200 (angles, tile(2), cos/sin in float32)."""
    coords_h = (np.arange(0.5, n_h, dtype=F32) / F32(n_h)).astype(F32)
    coords_w = (np.arange(0.5, n_w, dtype=F32) / F32(n_w)).astype(F32)
    yy, xx = np.meshgrid(coords_h, coords_w, indexing="ij")
    coords = np.stack([yy, xx], axis=-1).reshape(-1, 2).astype(F32)
    coords = (F32(2.0) * coords - F32(1.0)).astype(F32)
    inv_freq = (F32(1.0) / (F32(theta) ** np.arange(0, 1, 4 / head_dim, dtype=F32))).astype(F32)
    angles = (F32(2 * math.pi) * coords[:, :, None] * inv_freq[None, None, :]).astype(F32)
    angles = angles.reshape(angles.shape[0], -1)          # (P, head_dim/2)
    angles = np.tile(angles, 2)                           # (P, head_dim)
    return np.cos(angles).astype(F32), np.sin(angles).astype(F32)


def _rotate_half(x: np.ndarray) -> np.ndarray:
    """[tf]:203-207."""
    h = x.shape[-1] // 2
    return np.concatenate([-x[..., h:], x[..., :h]], axis=-1)


def embeddings(pixels: np.ndarray, w: Dict[str, np.ndarray], patch: int) -> np.ndarray:
    """[tf]:75-92: Conv2d(k=s=patch) -> flatten(2).transpose(1,2) -> cat[cls, registers, patches]."""
    B, C, H, W = pixels.shape
    nh, nw = H // patch, W // patch
    x = pixels[:, :, : nh * patch, : nw * patch].reshape(B, C, nh, patch, nw, patch)
    x = x.transpose(0, 2, 4, 1, 3, 5).reshape(B, nh * nw, C * patch * patch)   # im2col, (c,i,j) order
    wk = w["embeddings.patch_embeddings.weight"].reshape(-1, C * patch * patch)
    pe = x @ wk.T + w["embeddings.patch_embeddings.bias"]
    D = wk.shape[0]
    cls = np.broadcast_to(w["embeddings.cls_token"].reshape(1, 1, D), (B, 1, D))
    reg = w["embeddings.register_tokens"]
    reg = np.broadcast_to(reg.reshape(1, -1, D), (B, reg.shape[1], D))
    return np.concatenate([cls, reg, pe.astype(F32)], axis=1).astype(F32)


def attention(x: np.ndarray, w: Dict[str, np.ndarray], pre: str, n_heads: int, cos, sin,
              taps: Optional[dict] = None, tag: str = "") -> np.ndarray:
    """[tf]:294-334 (q/k/v proj, heads, RoPE on patch rows :238-268, softmax(qk^T * d^-0.5) v, o_proj),
    with the eager formulation of :210-234."""
    B, T, D = x.shape
    hd = D // n_heads
    q = x @ w[pre + "q_proj.weight"].T + w[pre + "q_proj.bias"]
    k = x @ w[pre + "k_proj.weight"].T
    if (pre + "k_proj.bias") in w:
        k = k + w[pre + "k_proj.bias"]
    v = x @ w[pre + "v_proj.weight"].T + w[pre + "v_proj.bias"]
    q = q.reshape(B, T, n_heads, hd).transpose(0, 2, 1, 3)
    k = k.reshape(B, T, n_heads, hd).transpose(0, 2, 1, 3)
    v = v.reshape(B, T, n_heads, hd).transpose(0, 2, 1, 3)
    if cos is not None:                      # DINOv3 only; DINOv2-with-registers has no RoPE
        n_prefix = T - cos.shape[0]
        qp, kp = q[:, :, n_prefix:], k[:, :, n_prefix:]
        qp = qp * cos + _rotate_half(qp) * sin
        kp = kp * cos + _rotate_half(kp) * sin
        q = np.concatenate([q[:, :, :n_prefix], qp], axis=2).astype(F32)
        k = np.concatenate([k[:, :, :n_prefix], kp], axis=2).astype(F32)
    s = (q @ k.transpose(0, 1, 3, 2)) * F32(hd ** -0.5)
    s = s - s.max(axis=-1, keepdims=True)
    p = np.exp(s, dtype=F32)
    p = p / p.sum(axis=-1, keepdims=True, dtype=F32)
    o = (p @ v).transpose(0, 2, 1, 3).reshape(B, T, D)
    if taps is not None:      # (B,T,D) head-major columns, as the HIP path stores them
        taps[tag + "q_rope"] = q.transpose(0, 2, 1, 3).reshape(B, T, D).copy()
        taps[tag + "k_rope"] = k.transpose(0, 2, 1, 3).reshape(B, T, D).copy()
        taps[tag + "v"] = v.transpose(0, 2, 1, 3).reshape(B, T, D).copy()
        taps[tag + "ctx"] = o.copy()
    return (o @ w[pre + "o_proj.weight"].T + w[pre + "o_proj.bias"]).astype(F32)


def layer(x: np.ndarray, w: Dict[str, np.ndarray], i: int, n_heads: int, eps: float, cos, sin,
          taps: Optional[dict] = None) -> np.ndarray:
    """[tf]:419-445: x += lambda1 * Attn(LN1(x)); x += lambda2 * MLP(LN2(x)); MLP = [tf]:356-357."""
    pre = f"model.layer.{i}."
    h = layer_norm(x, w[pre + "norm1.weight"], w[pre + "norm1.bias"], eps)
    if taps is not None:
        taps[f"l{i}.ln1"] = h.copy()
    a = attention(h, w, pre + "attention.", n_heads, cos, sin, taps, f"l{i}.")
    x = (a * w[pre + "layer_scale1.lambda1"] + x).astype(F32)
    if taps is not None:
        taps[f"l{i}.after_attn"] = x.copy()
    h = layer_norm(x, w[pre + "norm2.weight"], w[pre + "norm2.bias"], eps)
    u = gelu_erf(h @ w[pre + "mlp.up_proj.weight"].T + w[pre + "mlp.up_proj.bias"])
    if taps is not None:
        taps[f"l{i}.ln2"] = h.copy()
        taps[f"l{i}.up"] = u.copy()
    d = u @ w[pre + "mlp.down_proj.weight"].T + w[pre + "mlp.down_proj.bias"]
    x = (d * w[pre + "layer_scale2.lambda1"] + x).astype(F32)
    if taps is not None:
        taps[f"l{i}.out"] = x.copy()
    return x


def vit_forward(pixels: np.ndarray, w: Dict[str, np.ndarray], cfg, taps: Optional[dict] = None) -> np.ndarray:
    """[tf]:523-548 ``DINOv3ViTModel.forward`` -> last_hidden_state (B,T,D) float32.

    ``pixels``: (B,3,H,W) float32.  ``cfg``: any object with the HF config field names."""
    B, C, H, W = pixels.shape
    x = embeddings(pixels.astype(F32), w, cfg.patch_size)
    if taps is not None:
        taps["embeddings"] = x.copy()
    cos, sin = rope_cos_sin(H // cfg.patch_size, W // cfg.patch_size,
                            cfg.hidden_size // cfg.num_attention_heads, cfg.rope_theta)
    for i in range(cfg.num_hidden_layers):
        x = layer(x, w, i, cfg.num_attention_heads, cfg.layer_norm_eps, cos, sin, taps)
    return layer_norm(x, w["norm.weight"], w["norm.bias"], cfg.layer_norm_eps).astype(F32)


def dino_encoder_forward(x_gray: np.ndarray, w: Dict[str, np.ndarray], cfg, batch: int = 8) -> np.ndarray:
    """backend/cbas.py:672-677 ``DinoEncoder.forward``: (B,S,H,W) gray in [0,1] -> replicate to 3
    channels -> model -> ``last_hidden_state[:, 0, :]`` -> (B,S,D).  (The reference hard-codes
    768 in the reshape; D is taken from the config here: SURVEY.md §7 'Hard parts'.)"""
    B, S, H, W = x_gray.shape
    flat = x_gray.reshape(B * S, 1, H, W).astype(F32)
    outs = []
    for i in range(0, B * S, batch):
        px = np.repeat(flat[i:i + batch], 3, axis=1)
        outs.append(vit_forward(px, w, cfg)[:, 0, :])
    return np.concatenate(outs, axis=0).reshape(B, S, -1)

"""Weights: a counter-based synthetic generator and checkpoint readers.

The real DINOv3 checkpoints are gated and the build has no network, so every parity test and
benchmark uses *synthetic* weights.  They are produced by a counter-based generator (one 64-bit
hash per element, integer arithmetic only), so that the golden-vector script (which fills the
reference's modules), the oracle and the HIP path all see bit-identical float32 tensors without
committing 171 MB of weights.  Tensor names are the reference's own ``state_dict`` keys:

* encoder: HF ``DINOv3ViTModel`` keys ([tf] modeling_dinov3_vit.py:60-92, 271-357, 400-417, 507-515)
* head: ``ClassifierLSTMDeltas`` keys (reference backend/classifier_head.py:72-100; listed in
  SURVEY.md §8(a) row H8).
"""
from __future__ import annotations

import os
from typing import Dict, Tuple

import numpy as np

from .config import ViTConfig, HeadConfig

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (x + np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _hash_stream(seed: int, name: str, n: int) -> np.ndarray:
    key = _splitmix64(np.array([(seed & 0xFFFFFFFFFFFFFFFF) ^ _fnv1a64(name)], dtype=np.uint64))[0]
    with np.errstate(over="ignore"):
        return _splitmix64(key + np.arange(n, dtype=np.uint64))


def synth_normal(seed: int, name: str, shape: Tuple[int, ...], std: float, mean: float = 0.0) -> np.ndarray:
    """Approximately normal (Irwin-Hall of four 16-bit uniforms, |z| <= 3.46), exact arithmetic."""
    n = int(np.prod(shape)) if len(shape) else 1
    h = _hash_stream(seed, name, n)
    s = ((h & np.uint64(0xFFFF)) + ((h >> np.uint64(16)) & np.uint64(0xFFFF))
         + ((h >> np.uint64(32)) & np.uint64(0xFFFF)) + (h >> np.uint64(48))).astype(np.int64)
    z = (s - 2 * 65535).astype(np.float64) * (np.sqrt(3.0) / 65536.0)
    return (mean + std * z).astype(np.float32).reshape(shape)


def synth_uniform(seed: int, name: str, shape: Tuple[int, ...], lo: float, hi: float) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    h = _hash_stream(seed, name, n)
    u = (h >> np.uint64(40)).astype(np.float64) * (1.0 / (1 << 24))
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


# ----------------------------------------------------------------------------------------------
# Encoder
# ----------------------------------------------------------------------------------------------

def _dinov2_param_shapes(cfg: ViTConfig) -> Dict[str, Tuple[int, ...]]:
    """HF ``Dinov2WithRegistersModel`` state-dict keys (modeling_dinov2_with_registers.py:75-92, 203-320, 362-381, 455-470)."""
    D, F, R, p, C, G = (cfg.hidden_size, cfg.intermediate_size, cfg.num_register_tokens, cfg.patch_size,
                        cfg.num_channels, cfg.pos_embed_grid)
    s: Dict[str, Tuple[int, ...]] = {
        "embeddings.cls_token": (1, 1, D),
        "embeddings.mask_token": (1, D),
        "embeddings.register_tokens": (1, R, D),
        "embeddings.position_embeddings": (1, 1 + G * G, D),
        "embeddings.patch_embeddings.projection.weight": (D, C, p, p),
        "embeddings.patch_embeddings.projection.bias": (D,),
        "layernorm.weight": (D,),
        "layernorm.bias": (D,),
    }
    for i in range(cfg.num_hidden_layers):
        pre = f"encoder.layer.{i}."
        for nm, shp in (("norm1.weight", (D,)), ("norm1.bias", (D,)),
                        ("attention.attention.query.weight", (D, D)), ("attention.attention.query.bias", (D,)),
                        ("attention.attention.key.weight", (D, D)), ("attention.attention.key.bias", (D,)),
                        ("attention.attention.value.weight", (D, D)), ("attention.attention.value.bias", (D,)),
                        ("attention.output.dense.weight", (D, D)), ("attention.output.dense.bias", (D,)),
                        ("layer_scale1.lambda1", (D,)), ("norm2.weight", (D,)), ("norm2.bias", (D,)),
                        ("mlp.fc1.weight", (F, D)), ("mlp.fc1.bias", (F,)), ("mlp.fc2.weight", (D, F)),
                        ("mlp.fc2.bias", (D,)), ("layer_scale2.lambda1", (D,))):
            s[pre + nm] = shp
    return s


_V2_TO_CANON = (("embeddings.patch_embeddings.projection.", "embeddings.patch_embeddings."),
                ("encoder.layer.", "model.layer."), ("attention.attention.query.", "attention.q_proj."),
                ("attention.attention.key.", "attention.k_proj."), ("attention.attention.value.", "attention.v_proj."),
                ("attention.output.dense.", "attention.o_proj."), ("mlp.fc1.", "mlp.up_proj."),
                ("mlp.fc2.", "mlp.down_proj."), ("layernorm.", "norm."))


def canonical_encoder_weights(cfg: ViTConfig, w: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """Rename a DINOv2-with-registers state dict to the DINOv3 key names the packer and the oracle use
    (same tensors, same shapes); a DINOv3 state dict is returned unchanged."""
    if cfg.model_type != "dinov2_with_registers":
        return w
    out = {}
    for k, v in w.items():
        for a, b in _V2_TO_CANON:
            if k.startswith(a):
                k = b + k[len(a):]
            elif a in k and not a.startswith("embeddings") and not a.startswith("encoder") and not a.startswith("layernorm"):
                k = k.replace(a, b)
        out[k] = v
    return out


def encoder_param_shapes(cfg: ViTConfig) -> Dict[str, Tuple[int, ...]]:
    if cfg.model_type == "dinov2_with_registers":
        return _dinov2_param_shapes(cfg)
    D, F, R, p, C = cfg.hidden_size, cfg.intermediate_size, cfg.num_register_tokens, cfg.patch_size, cfg.num_channels
    s: Dict[str, Tuple[int, ...]] = {
        "embeddings.cls_token": (1, 1, D),
        "embeddings.mask_token": (1, 1, D),
        "embeddings.register_tokens": (1, R, D),
        "embeddings.patch_embeddings.weight": (D, C, p, p),
        "embeddings.patch_embeddings.bias": (D,),
        "norm.weight": (D,),
        "norm.bias": (D,),
    }
    for i in range(cfg.num_hidden_layers):
        pre = f"model.layer.{i}."
        s[pre + "norm1.weight"] = (D,)
        s[pre + "norm1.bias"] = (D,)
        s[pre + "attention.q_proj.weight"] = (D, D)
        s[pre + "attention.q_proj.bias"] = (D,)
        s[pre + "attention.k_proj.weight"] = (D, D)
        s[pre + "attention.v_proj.weight"] = (D, D)
        s[pre + "attention.v_proj.bias"] = (D,)
        s[pre + "attention.o_proj.weight"] = (D, D)
        s[pre + "attention.o_proj.bias"] = (D,)
        s[pre + "layer_scale1.lambda1"] = (D,)
        s[pre + "norm2.weight"] = (D,)
        s[pre + "norm2.bias"] = (D,)
        s[pre + "mlp.up_proj.weight"] = (F, D)
        s[pre + "mlp.up_proj.bias"] = (F,)
        s[pre + "mlp.down_proj.weight"] = (D, F)
        s[pre + "mlp.down_proj.bias"] = (D,)
        s[pre + "layer_scale2.lambda1"] = (D,)
    return s


def synth_encoder_weights(cfg: ViTConfig, seed: int = 1234) -> Dict[str, np.ndarray]:
    """Seeded random encoder weights.  Nothing is left at its init value (LayerScale, LayerNorm
    gains/offsets and biases are all randomised) so a missing multiply or add cannot hide."""
    out: Dict[str, np.ndarray] = {}
    for name, shape in encoder_param_shapes(cfg).items():
        leaf = name.split(".")[-1]
        if name.endswith("mask_token"):
            w = np.zeros(shape, np.float32)
        elif "cls_token" in name or "register_tokens" in name:
            w = synth_normal(seed, name, shape, 0.5)
        elif "position_embeddings" in name:
            w = synth_normal(seed, name, shape, 0.3)
        elif "lambda1" in name:
            w = synth_uniform(seed, name, shape, 0.2, 1.0)
        elif "norm" in name and leaf == "weight":
            w = synth_uniform(seed, name, shape, 0.7, 1.3)
        elif "norm" in name and leaf == "bias":
            w = synth_uniform(seed, name, shape, -0.2, 0.2)
        elif leaf == "bias":
            w = synth_uniform(seed, name, shape, -0.1, 0.1)
        elif "q_proj" in name or "k_proj" in name or ".query." in name or ".key." in name:
            w = synth_normal(seed, name, shape, 0.05)
        elif "patch_embeddings.weight" in name or "projection.weight" in name:
            w = synth_normal(seed, name, shape, 0.03)
        else:
            w = synth_normal(seed, name, shape, 0.02)
        out[name] = w
    return out


# ----------------------------------------------------------------------------------------------
# Head
# ----------------------------------------------------------------------------------------------

def head_param_shapes(cfg: HeadConfig) -> Dict[str, Tuple[int, ...]]:
    I, C, Bn, h, L0 = cfg.in_features, cfg.out_features, cfg.bottleneck_dim, cfg.lstm_hidden_size, cfg.lin0_dim
    s: Dict[str, Tuple[int, ...]] = {"gate": (), "attention_temp": ()}
    streams = ("cls", "delta", "acc") if cfg.use_acceleration else ("cls", "delta")     # classifier_head.py:74-84
    for stream in streams:
        s[f"{stream}_bottleneck.0.weight"] = (Bn, I)
        s[f"{stream}_bottleneck.0.bias"] = (Bn,)
        s[f"{stream}_ln.weight"] = (Bn,)
        s[f"{stream}_ln.bias"] = (Bn,)
    s["lin0.0.weight"] = (L0, len(streams) * Bn)
    s["lin0.0.bias"] = (L0,)
    s["attention_head.weight"] = (1, 2 * h)
    s["attention_head.bias"] = (1,)
    s["lin1.weight"] = (C, I)
    s["lin1.bias"] = (C,)
    s["lin2.weight"] = (C, 2 * h)
    s["lin2.bias"] = (C,)
    for layer in range(cfg.lstm_layers):          # nn.LSTM(256, h, num_layers, bidirectional): layer k > 0 reads 2h
        for sfx in ("", "_reverse"):
            s[f"lstm.weight_ih_l{layer}{sfx}"] = (4 * h, L0 if layer == 0 else 2 * h)
            s[f"lstm.weight_hh_l{layer}{sfx}"] = (4 * h, h)
            s[f"lstm.bias_ih_l{layer}{sfx}"] = (4 * h,)
            s[f"lstm.bias_hh_l{layer}{sfx}"] = (4 * h,)
    return s


def synth_head_weights(cfg: HeadConfig, seed: int = 4321) -> Dict[str, np.ndarray]:
    out: Dict[str, np.ndarray] = {}
    for name, shape in head_param_shapes(cfg).items():
        leaf = name.split(".")[-1]
        if name == "gate":
            w = np.float32(0.35) * np.ones((), np.float32)
        elif name == "attention_temp":
            w = np.float32(0.8) * np.ones((), np.float32)
        elif "_ln." in name and leaf == "weight":
            w = synth_uniform(seed, name, shape, 0.7, 1.3)
        elif "_ln." in name and leaf == "bias":
            w = synth_uniform(seed, name, shape, -0.2, 0.2)
        elif leaf == "bias" or "bias_" in leaf:
            w = synth_uniform(seed, name, shape, -0.1, 0.1)
        elif "bottleneck" in name:
            # deltas are small; give the delta/acc streams a larger gain so they matter
            std = {"cls": 0.05, "delta": 0.4, "acc": 0.4}[name.split("_")[0]]
            w = synth_normal(seed, name, shape, std)
        elif name.startswith("lstm.weight"):
            w = synth_uniform(seed, name, shape, -0.125, 0.125)   # torch default 1/sqrt(h)
        elif name.startswith("lin1"):
            w = synth_normal(seed, name, shape, 0.25)             # decisive linear branch (trained heads are)
        elif name.startswith("lin2") or name.startswith("attention_head"):
            w = synth_normal(seed, name, shape, 1.0)
        else:
            w = synth_normal(seed, name, shape, 0.08)
        out[name] = np.asarray(w, np.float32)
    return out


# ----------------------------------------------------------------------------------------------
# Checkpoint readers (no transformers / torch.nn dependency)
# ----------------------------------------------------------------------------------------------

def _load_safetensors_f32(path: str) -> Dict[str, np.ndarray]:
    """One safetensors file as float32 arrays.  numpy has no bfloat16: files holding bf16 (or other torch-only) tensors
    go through torch."""
    try:
        from safetensors.numpy import load_file
        raw = load_file(path)
        return {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in raw.items()}
    except (TypeError, ValueError, KeyError):
        from safetensors.torch import load_file as load_torch
        return {k: np.ascontiguousarray(v.float().numpy()) for k, v in load_torch(path).items()}


def load_encoder_checkpoint(ckpt_dir: str) -> Tuple[ViTConfig, Dict[str, np.ndarray]]:
    """Read ``config.json`` + ``model.safetensors`` written by ``save_pretrained`` / the HF hub."""
    cfg = ViTConfig.from_json_file(os.path.join(ckpt_dir, "config.json"))
    st_path = os.path.join(ckpt_dir, "model.safetensors")
    index = os.path.join(ckpt_dir, "model.safetensors.index.json")
    if os.path.exists(st_path):
        shards = [st_path]
    elif os.path.exists(index):                                 # a sharded save_pretrained (max_shard_size)
        import json
        with open(index) as f:
            shards = sorted({os.path.join(ckpt_dir, v) for v in json.load(f)["weight_map"].values()})
    else:
        raise FileNotFoundError(f"{st_path} not found (.bin checkpoints are not supported: convert with save_pretrained)")
    weights = {}
    for shard in shards:
        weights.update(_load_safetensors_f32(shard))
    needed = [k for k in encoder_param_shapes(cfg) if not k.endswith("mask_token")]
    missing = [k for k in needed if k not in weights]
    if missing:
        # a checkpoint saved from a task model keeps its backbone under a prefix ("dinov3_vit.", "backbone.", ...):
        # from_pretrained strips it (base_model_prefix), so does this
        for pre in sorted({k.split(".", 1)[0] + "." for k in weights if "." in k}):
            sub = {k[len(pre):]: v for k, v in weights.items() if k.startswith(pre)}
            if all(k in sub for k in needed):
                weights, missing = sub, []
                break
    if missing:
        raise KeyError(f"checkpoint {ckpt_dir} lacks tensors: {missing[:4]}{'...' if len(missing) > 4 else ''}")
    return cfg, weights


def save_encoder_checkpoint(ckpt_dir: str, cfg: ViTConfig, weights: Dict[str, np.ndarray]) -> None:
    from safetensors.numpy import save_file
    os.makedirs(ckpt_dir, exist_ok=True)
    with open(os.path.join(ckpt_dir, "config.json"), "w") as f:
        f.write(cfg.to_json())
    save_file({k: np.ascontiguousarray(v) for k, v in weights.items()}, os.path.join(ckpt_dir, "model.safetensors"))


def infer_head_config(weights: Dict[str, np.ndarray], seq_len: int = 31) -> HeadConfig:
    """Recover the head hyper-parameters from a state dict, as the reference's model-bundle loader
    does (backend/workthreads.py:415-425)."""
    h = int(weights["attention_head.weight"].shape[1]) // 2
    layers = 1 + max(int(k.split("weight_ih_l")[1].split("_")[0]) for k in weights if "lstm.weight_ih_l" in k)
    C, I = (int(x) for x in weights["lin1.weight"].shape)
    Bn = int(weights["cls_bottleneck.0.weight"].shape[0])
    return HeadConfig(in_features=I, out_features=C, seq_len=seq_len, bottleneck_dim=Bn,
                      use_acceleration="acc_bottleneck.0.weight" in weights,
                      lstm_hidden_size=h, lstm_layers=layers,
                      lin0_dim=int(weights["lin0.0.weight"].shape[0]))

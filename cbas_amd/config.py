"""Configuration records for the encoder (DINOv3 ViT) and the classifier head.

The field names follow the HF ``config.json`` of a DINOv3 ViT checkpoint
(``transformers/models/dinov3_vit/configuration_dinov3_vit.py:74-101``) so that a
checkpoint directory can be read without importing ``transformers``; the head fields
follow the constructor of ``ClassifierLSTMDeltas`` (reference
``backend/classifier_head.py:62-64``).
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, asdict


@dataclass(frozen=True)
class ViTConfig:
    hidden_size: int = 768
    intermediate_size: int = 3072
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    num_register_tokens: int = 4
    patch_size: int = 16
    image_size: int = 224
    layer_norm_eps: float = 1e-5
    rope_theta: float = 100.0
    query_bias: bool = True
    key_bias: bool = False
    value_bias: bool = True
    proj_bias: bool = True
    mlp_bias: bool = True
    use_gated_mlp: bool = False
    hidden_act: str = "gelu"
    num_channels: int = 3
    # encoder family: "dinov3_vit" (RoPE, no additive position embedding) or "dinov2_with_registers"
    # (learned position embedding on a pos_embed_grid x pos_embed_grid lattice, bicubically
    # interpolated to the frame's patch grid; no RoPE; key bias) - CBAS's default project encoder
    # (reference backend/cbas.py:1030-1033)
    model_type: str = "dinov3_vit"
    use_rope: bool = True
    pos_embed_grid: int = 0

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @property
    def num_prefix_tokens(self) -> int:
        return 1 + self.num_register_tokens

    def num_patches(self, height: int, width: int) -> int:
        return (height // self.patch_size) * (width // self.patch_size)

    def num_tokens(self, height: int, width: int) -> int:
        return self.num_prefix_tokens + self.num_patches(height, width)

    def flops_per_frame(self, height: int, width: int) -> float:
        """Algorithmic FLOPs (MAC = 2) per frame: SURVEY.md §8(a) 'Encoder totals'."""
        D, F, L = self.hidden_size, self.intermediate_size, self.num_hidden_layers
        P, T = self.num_patches(height, width), self.num_tokens(height, width)
        patch = P * self.num_channels * self.patch_size ** 2 * D
        per_layer = 4 * T * D * D + 2 * T * D * F + 2 * T * T * D
        return 2.0 * (patch + L * per_layer)

    def validate(self) -> None:
        if self.use_gated_mlp:
            raise NotImplementedError("gated (SwiGLU) MLP variants (S+/H+) are not on the hot path")
        if self.hidden_act != "gelu":
            raise NotImplementedError(f"hidden_act={self.hidden_act!r}; only exact-erf 'gelu' is implemented")
        if self.head_dim != 64:
            raise NotImplementedError("head_dim must be 64 (all DINOv3 ViT-S/B/L checkpoints)")
        if self.patch_size not in (14, 16):
            raise NotImplementedError("patch_size must be 14 (DINOv2) or 16 (DINOv3)")
        if self.model_type not in ("dinov3_vit", "dinov2_with_registers"):
            raise NotImplementedError(f"model_type={self.model_type!r} is not built")
        if self.use_rope == (self.pos_embed_grid > 0):
            raise NotImplementedError("exactly one of RoPE / learned position embedding is expected")
        if not (self.query_bias and self.value_bias and self.proj_bias and self.mlp_bias):
            raise NotImplementedError("query/value/proj/mlp biases are expected (DINOv3 defaults)")

    def to_json(self) -> str:
        d = asdict(self)
        if self.model_type == "dinov2_with_registers":      # write the fields HF's config class reads back
            d.update(mlp_ratio=self.intermediate_size // self.hidden_size, qkv_bias=self.query_bias,
                     use_swiglu_ffn=self.use_gated_mlp)
        return json.dumps(d, indent=2)

    @classmethod
    def from_json_file(cls, path: str) -> "ViTConfig":
        with open(path, "r") as f:
            raw = json.load(f)
        mt = raw.get("model_type", "dinov3_vit")
        if mt not in ("dinov3_vit", "dinov2_with_registers"):
            raise NotImplementedError(f"model_type={mt!r}: only DINOv3 ViT and DINOv2-with-registers encoders are built")
        known = {k: raw[k] for k in cls.__dataclass_fields__ if k in raw}
        for key in ("patch_size", "image_size"):
            if isinstance(known.get(key), (list, tuple)):
                known[key] = int(known[key][0])
        if mt == "dinov2_with_registers":
            # HF Dinov2WithRegistersConfig (configuration_dinov2_with_registers.py): mlp_ratio, qkv_bias, use_swiglu_ffn
            hs = int(raw.get("hidden_size", 768))
            qkv_bias = bool(raw.get("qkv_bias", True))
            known.update(intermediate_size=hs * int(raw.get("mlp_ratio", 4)), query_bias=qkv_bias, key_bias=qkv_bias,
                         value_bias=qkv_bias, use_gated_mlp=bool(raw.get("use_swiglu_ffn", False)), use_rope=False,
                         layer_norm_eps=float(raw.get("layer_norm_eps", 1e-6)),
                         num_register_tokens=int(raw.get("num_register_tokens", 4)),
                         patch_size=int(known.get("patch_size", 16)), image_size=int(known.get("image_size", 224)))
            known["pos_embed_grid"] = known["image_size"] // known["patch_size"]
        return cls(**known)


VIT_S16 = ViTConfig(hidden_size=384, intermediate_size=1536, num_hidden_layers=12, num_attention_heads=6)
VIT_B16 = ViTConfig(hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12)
VIT_L16 = ViTConfig(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16)
# Tiny config for fast oracle/kernel parity (not a published architecture).
VIT_TINY = ViTConfig(hidden_size=128, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                     image_size=64)

# DINOv2-with-registers (CBAS's default encoder, "facebook/dinov2-with-registers-base"): ViT-B/14 trained at 518
DINOV2_REG_B14 = ViTConfig(hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
                           patch_size=14, image_size=518, layer_norm_eps=1e-6, key_bias=True,
                           model_type="dinov2_with_registers", use_rope=False, pos_embed_grid=37)
DINOV2_REG_TINY = ViTConfig(hidden_size=128, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                            patch_size=14, image_size=70, layer_norm_eps=1e-6, key_bias=True,
                            model_type="dinov2_with_registers", use_rope=False, pos_embed_grid=5)

NAMED_VIT = {"vits16": VIT_S16, "vitb16": VIT_B16, "vitl16": VIT_L16, "tiny": VIT_TINY,
             "dinov2regb14": DINOV2_REG_B14, "dinov2regtiny": DINOV2_REG_TINY}


@dataclass(frozen=True)
class HeadConfig:
    """``ClassifierLSTMDeltas(in_features, out_features, seq_len, ...)``: classifier_head.py:62-64."""
    in_features: int = 768
    out_features: int = 9
    seq_len: int = 31
    bottleneck_dim: int = 128
    use_acceleration: bool = True
    ema_alpha: float = 0.3
    center_window_size: int = 5
    lstm_hidden_size: int = 64
    lstm_layers: int = 1
    lin0_dim: int = 256

    @property
    def hsl(self) -> int:
        return self.seq_len // 2

    @property
    def centre_lo(self) -> int:
        return max(0, self.hsl - self.center_window_size)

    @property
    def centre_hi(self) -> int:
        return min(self.seq_len, self.hsl + self.center_window_size + 1)

    def flops_per_frame_naive(self) -> float:
        """Naive (per-window) head FLOPs, SURVEY.md §8(a) 'Head totals' definition."""
        T, I, Bn, C, h = self.seq_len, self.in_features, self.bottleneck_dim, self.out_features, self.lstm_hidden_size
        n_streams = 3 if self.use_acceleration else 2
        mac = T * I * Bn * n_streams                        # bottlenecks
        mac += T * Bn * n_streams * self.lin0_dim           # lin0
        mac += 2 * T * (self.lin0_dim * 4 * h + h * 4 * h)  # BiLSTM
        mac += (self.centre_hi - self.centre_lo) * I * C    # lin1 on the centre window
        mac += 2 * h * C + (self.centre_hi - self.centre_lo) * 2 * h
        return 2.0 * mac

    def validate(self) -> None:
        if not 1 <= self.lstm_layers <= 4:
            raise NotImplementedError("lstm_layers must be in [1, 4]")
        if self.lstm_hidden_size < 16 or self.lstm_hidden_size > 128 or self.lstm_hidden_size % 16:
            raise NotImplementedError("lstm_hidden_size must be a multiple of 16 in [16, 128]")
        if self.seq_len < 3:
            raise NotImplementedError("seq_len < 3 (replicate-pad delta mode) is not implemented")
        if self.centre_lo >= self.centre_hi:
            raise NotImplementedError("empty centre window")


def find_checkpoint_dir(model_identifier: str) -> str:
    """Resolve ``model_identifier`` the way ``AutoModel.from_pretrained`` would *offline*.

    A local directory is used as is; a hub name is looked up in the HF cache
    (``$HF_HOME/hub/models--org--name/snapshots/<rev>/``).  No network access is attempted.
    Mirrors the reference call at backend/cbas.py:657.
    """
    if os.path.isdir(model_identifier):
        return model_identifier
    hf_home = os.environ.get("HF_HOME", os.path.join(os.path.expanduser("~"), ".cache", "huggingface"))
    hub = os.environ.get("HF_HUB_CACHE", os.path.join(hf_home, "hub"))
    snap = os.path.join(hub, "models--" + model_identifier.replace("/", "--"), "snapshots")
    if os.path.isdir(snap):
        revs = sorted(os.listdir(snap))
        for rev in reversed(revs):
            cand = os.path.join(snap, rev)
            if os.path.exists(os.path.join(cand, "config.json")):
                return cand
    raise FileNotFoundError(
        f"encoder checkpoint {model_identifier!r} is neither a local directory nor present in the "
        f"Hugging Face cache ({snap}); this build never downloads")

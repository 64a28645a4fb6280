"""``DinoEncoder`` — drop-in for the reference's encoder object (backend/cbas.py:650-677).

Same constructor (``DinoEncoder(model_identifier: str, device="cuda")``), same ``.device``
attribute, same call contract (``encoder(x)`` with ``x`` float32 ``(B, S, H, W)`` in [0, 1] ->
``(B, S, D)``), but the forward pass runs in the hand-written HIP kernels of libcbas_mi355x.so
through the C ABI (``cbas_enc_*``).  torch tensors are only containers for device memory.

Extras beyond the reference interface (used by ``encode_file`` and the benchmarks):
``encode_u8`` takes uint8 frames already in HBM, ``submit_host``/``wait`` stream host chunks
through pinned staging with copy/compute overlap.
"""
from __future__ import annotations

import os

import ctypes as C
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _lib
from .config import ViTConfig, find_checkpoint_dir
from .weights import load_encoder_checkpoint


def pack_encoder_weights(cfg: ViTConfig, w: Dict[str, np.ndarray]) -> np.ndarray:
    """Flatten an HF state dict into the blob order documented in include/cbas_mi355x.h."""
    from .weights import canonical_encoder_weights
    w = dict(canonical_encoder_weights(cfg, w))    # DINOv2-with-registers keys -> the DINOv3 names used below
    D = cfg.hidden_size
    parts = [w["embeddings.cls_token"].reshape(-1), w["embeddings.register_tokens"].reshape(-1)]
    if cfg.pos_embed_grid > 0:
        parts.append(w["embeddings.position_embeddings"].reshape(-1))
    parts += [w["embeddings.patch_embeddings.weight"].reshape(-1), w["embeddings.patch_embeddings.bias"].reshape(-1)]
    zero_kb = np.zeros(D, np.float32)
    for i in range(cfg.num_hidden_layers):
        p = f"model.layer.{i}."
        w.setdefault(p + "attention.k_proj.bias", zero_kb)       # DINOv3 has no key bias ([tf] key_bias=False)
        for k in ("norm1.weight", "norm1.bias", "attention.q_proj.weight", "attention.q_proj.bias",
                  "attention.k_proj.weight", "attention.k_proj.bias", "attention.v_proj.weight", "attention.v_proj.bias",
                  "attention.o_proj.weight", "attention.o_proj.bias", "layer_scale1.lambda1",
                  "norm2.weight", "norm2.bias", "mlp.up_proj.weight", "mlp.up_proj.bias",
                  "mlp.down_proj.weight", "mlp.down_proj.bias", "layer_scale2.lambda1"):
            parts.append(np.asarray(w[p + k], np.float32).reshape(-1))
    parts += [w["norm.weight"].reshape(-1), w["norm.bias"].reshape(-1)]
    blob = np.ascontiguousarray(np.concatenate([np.asarray(p, np.float32) for p in parts]))
    assert blob.shape[0] > D
    return blob


def _device_index(device: torch.device) -> int:
    if device.type != "cuda":
        raise RuntimeError(
            f"cbas_amd runs only on an AMD GPU exposed as torch device 'cuda' (got {device}); there is no CPU path")
    return device.index if device.index is not None else torch.cuda.current_device()


# What an encoder built the reference's way - DinoEncoder(model_identifier, device), integration.install(),
# `python -m cbas_amd.encode_files` - computes in: 4 = fp32 storage / attention / LayerNorm with the GEMM products as three-term
# fp16 splits: the mode that meets BOTH halves of the path's contract (CLS within 1e-3 - by three orders of magnitude - and
# argmax labels identical to the reference CPU path).  The fp16-operand mode (0) is the explicit fast mode.
DEFAULT_PRECISION = 4


class DinoEncoder:
    """MI355X DINOv3 ViT encoder behind the reference's ``DinoEncoder`` interface."""

    def __init__(self, model_identifier: str, device="cuda", max_batch: int = 128,
                 max_frame: Tuple[int, int] = (256, 256), precision: Optional[int] = None):
        # max_batch: frames per encoder launch sequence.  A frame's row does not depend on its batch (bit-exact:
        # tests/test_gpu_parity.py::test_vitb_full_batch_invariance), so this is scheduling only: 128 runs the file path
        # ~3 % faster than 64 (fewer partly-filled tile rounds per frame; 256 adds nothing) for ~1 GB more workspace.
        print(f"Loading DINO encoder model: {model_identifier}")          # the reference's line (backend/cbas.py:654)
        ckpt = find_checkpoint_dir(model_identifier)
        cfg, weights = load_encoder_checkpoint(ckpt)
        if precision is None:
            # CBAS constructs the encoder as DinoEncoder(model_identifier=..., device=...) (startup_page.py:66-69): an
            # unmodified checkout gets DEFAULT_PRECISION = 4, the contract-complete mode - CLS rows ~1e-6 from the reference's
            # fp32 CPU path and EVERY argmax label the reference's (tests/test_gpu_fp32.py strict gates; 10.5k frames/s for
            # ViT-B/16 on one MI355X).  CBAS_PRECISION=0 opts into the reference's own GPU behaviour - fp16 operands, as under
            # its torch.autocast (cbas.py:433-434) - 2.2x faster, rows within 1e-3, <= 1 % of near-tie labels differ from the
            # CPU path.  MX-fp8 (2) is refused here: its rows need their own heads.
            precision = int(os.environ.get("CBAS_PRECISION", str(DEFAULT_PRECISION)))
            if precision == 2:
                raise ValueError("CBAS_PRECISION=2 (MX-fp8) is not selectable through the environment: its rows are not "
                                 "interchangeable with the other modes' (pass precision=2 explicitly)")
        self._init(cfg, weights, device, max_batch, max_frame, precision)
        self.model_identifier = model_identifier
        mode = {0: "fp16 operands - the reference's own GPU behaviour under autocast (fast mode)", 1: "fp16 activations, hi + lo split weights",
                2: "MX-fp8 operands", 3: "fp32 end to end (label-exact)",
                4: "fp32 with split-fp16 GEMM products (label-exact; CBAS_PRECISION=0 selects the fast mode)"}.get(int(precision), "?")
        print(f"cbas_amd: MI355X encoder on {getattr(self, 'device', device)}, precision {int(precision)}: {mode}")

    @classmethod
    def from_weights(cls, cfg: ViTConfig, weights: Dict[str, np.ndarray], device="cuda", max_batch: int = 64,
                     max_frame: Tuple[int, int] = (256, 256), precision: int = 0) -> "DinoEncoder":
        self = cls.__new__(cls)
        self._init(cfg, weights, device, max_batch, max_frame, precision)
        self.model_identifier = "<in-memory>"
        return self

    def _init(self, cfg: ViTConfig, weights, device, max_batch, max_frame, precision):
        cfg.validate()
        self.config = cfg
        self.device = torch.device(device)
        self._dev = _device_index(self.device)
        self.max_batch = int(max_batch)
        self.max_frame = (int(max_frame[0]), int(max_frame[1]))
        self._lib = _lib.load()
        self._cfg_c = _lib.EncConfig(cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers,
                                     cfg.num_attention_heads, cfg.num_register_tokens, cfg.patch_size,
                                     cfg.layer_norm_eps, cfg.rope_theta, self.max_batch, self.max_frame[0],
                                     self.max_frame[1], int(precision), int(cfg.use_rope), int(cfg.pos_embed_grid))
        blob = pack_encoder_weights(cfg, weights)
        need = self._lib.cbas_enc_weights_count(C.byref(self._cfg_c))
        if need != blob.shape[0]:
            raise RuntimeError(f"weight blob has {blob.shape[0]} floats, library expects {need}")
        self._blob = blob                 # kept so that the handle can be rebuilt for larger frames
        self._h = None
        self._create()

    def _create(self) -> None:
        h = C.c_void_p()
        _lib.check(self._lib.cbas_enc_create(C.byref(self._cfg_c), self._blob.ctypes.data, self._blob.shape[0], self._dev,
                                             C.byref(h)), "cbas_enc_create")
        self._h = h

    def _fit_frame(self, H: int, W: int) -> None:
        """The reference takes any frame size; the workspace here is sized at create.  Frames larger than
        ``max_frame`` rebuild the handle once with a workspace that fits them (weights are re-uploaded)."""
        if H <= self.max_frame[0] and W <= self.max_frame[1]:
            return
        if getattr(self, "_slot_n", None):
            raise RuntimeError("frame size grew while batches are in flight; wait for them first")
        new = (max(H, self.max_frame[0]), max(W, self.max_frame[1]))
        print(f"cbas_amd: frames of {H}x{W} exceed the encoder workspace ({self.max_frame[0]}x{self.max_frame[1]}); "
              f"rebuilding it for {new[0]}x{new[1]}")
        torch.cuda.synchronize(self.device)
        self.close()
        self.max_frame = new
        self._cfg_c.max_height, self._cfg_c.max_width = new
        self._create()

    @property
    def precision(self) -> int:
        """0 fp16 operands (the explicit fast mode; `from_weights`' default), 1 fp16 hi+lo weights, 2 MX-fp8 throughput mode,
        3 fp32 end to end - the reference's CPU arithmetic, 4 the same with the GEMM products as three-term fp16 splits (as
        exact, three times as fast: the DEFAULT of an encoder built the reference's way) (include/cbas_mi355x.h)."""
        return int(self._cfg_c.precision)

    # -- nn.Module-like surface used by the reference ------------------------------------------
    def eval(self):
        return self

    def to(self, device):
        if torch.device(device) != self.device and torch.device(device).type != self.device.type:
            raise RuntimeError("the MI355X encoder cannot be moved off its device")
        return self

    def parameters(self):
        return iter(())

    def _register_session(self, session) -> None:
        import weakref
        if getattr(self, "_sessions", None) is None:
            self._sessions = weakref.WeakSet()
        self._sessions.add(session)

    def range_fallback(self) -> "DinoEncoder":
        """The same model in precision 3 (fp32 end to end: no operand range limit), built on first use from this encoder's own
        weight blob and closed with it.  The file paths re-encode a video here when its activations left this encoder's
        range (CBAS_ERANGE: fp16 storage in precision 0, the scaled fp16 halves of precision 4) - the reference's fp32
        arithmetic has no such limit, so a drop-in must not fail where the reference would have produced rows."""
        twin = getattr(self, "_range_twin", None)
        if twin is None:
            twin = DinoEncoder.__new__(DinoEncoder)
            twin.config, twin.device, twin._dev = self.config, self.device, self._dev
            twin.max_batch, twin.max_frame = self.max_batch, self.max_frame
            twin._lib, twin._blob = self._lib, self._blob
            c = self._cfg_c
            twin._cfg_c = _lib.EncConfig(c.hidden_size, c.intermediate_size, c.num_layers, c.num_heads, c.num_register_tokens,
                                         c.patch_size, c.layer_norm_eps, c.rope_theta, c.max_batch, c.max_height, c.max_width,
                                         3, c.use_rope, c.pos_embed_grid)
            twin._h = None
            twin.model_identifier = getattr(self, "model_identifier", "<in-memory>")
            twin._create()
            self._range_twin = twin
        return twin

    def close(self):
        twin = getattr(self, "_range_twin", None)
        if twin is not None:
            self._range_twin = None
            twin.close()
        # fused sessions drain through this handle when they are destroyed: close them while it still exists
        for s in list(getattr(self, "_sessions", None) or ()):
            try:
                s.close()
            except Exception:  # noqa: BLE001
                pass
        if getattr(self, "_h", None):
            self._lib.cbas_enc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- the reference call: encoder(x) ---------------------------------------------------------
    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        return self.forward(x)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """backend/cbas.py:672-677: x (B,S,H,W) float32 in [0,1] -> (B,S,D) float32."""
        B, S, H, W = x.shape
        self._fit_frame(H, W)
        x = x.to(self.device, dtype=torch.float32).contiguous().reshape(B * S, H, W)
        out = torch.empty((B * S, self.config.hidden_size), dtype=torch.float32, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        for i in range(0, B * S, self.max_batch):
            n = min(self.max_batch, B * S - i)
            _lib.check(self._lib.cbas_enc_forward_f32(self._h, x[i:i + n].data_ptr(), n, H, W,
                                                      out[i:i + n].data_ptr(), None, stream),
                       "cbas_enc_forward_f32")
        return out.reshape(B, S, self.config.hidden_size)

    # -- uint8 fast paths -------------------------------------------------------------------------
    def encode_u8(self, frames: torch.Tensor, channel: int = 1, want_f32: bool = True):
        """frames: uint8 device tensor (n,H,W,3) [decord layout; ``channel`` selects green] or (n,H,W).
        Returns (cls_f16 (n,D) torch.float16, cls_f32 (n,D) or None)."""
        assert frames.dtype == torch.uint8 and frames.is_cuda
        frames = frames.contiguous()
        if frames.dim() == 4:
            n, H, W, Cn = frames.shape
            strides = (H * W * Cn, W * Cn, Cn)
            base_off = channel
        else:
            n, H, W = frames.shape
            strides = (H * W, W, 1)
            base_off = 0
        self._fit_frame(H, W)
        D = self.config.hidden_size
        out16 = torch.empty((n, D), dtype=torch.float16, device=self.device)
        out32 = torch.empty((n, D), dtype=torch.float32, device=self.device) if want_f32 else None
        stream = torch.cuda.current_stream(self.device).cuda_stream
        for i in range(0, n, self.max_batch):
            m = min(self.max_batch, n - i)
            _lib.check(self._lib.cbas_enc_forward_u8(
                self._h, frames[i:i + m].data_ptr() + base_off, m, H, W, strides[0], strides[1], strides[2],
                out32[i:i + m].data_ptr() if want_f32 else None, out16[i:i + m].data_ptr(), stream),
                "cbas_enc_forward_u8")
        return out16, out32

    def submit_host(self, slot: int, frames: np.ndarray, channel: int = 1) -> None:
        """Queue one host chunk (uint8 (n,H,W,3) or (n,H,W), C-contiguous) on ``slot``."""
        assert frames.dtype == np.uint8 and frames.flags.c_contiguous
        if frames.ndim == 4:
            n, H, W, Cn = frames.shape
            strides, off = (H * W * Cn, W * Cn, Cn), channel
        else:
            n, H, W = frames.shape
            strides, off = (H * W, W, 1), 0
        self._fit_frame(H, W)
        _lib.check(self._lib.cbas_enc_submit_u8_host(self._h, slot, frames.ctypes.data + off, n, H, W, *strides),
                   "cbas_enc_submit_u8_host")
        self._slot_n = getattr(self, "_slot_n", {})
        self._slot_n[slot] = n

    def submit_dev(self, slot: int, frames: torch.Tensor, out16: Optional[torch.Tensor], out32: Optional[torch.Tensor] = None,
                   channel: int = 1) -> None:
        """Asynchronous encode of one device batch (<= max_batch frames, uint8 (n,H,W,3) or (n,H,W)) into the
        given output rows, on one of the handle's two compute lanes, ordered after the current torch stream.
        Call ``wait_stream(slot)`` before anything reads the outputs or overwrites the frames."""
        assert frames.dtype == torch.uint8 and frames.is_cuda and frames.is_contiguous()
        if frames.dim() == 4:
            n, H, W, Cn = frames.shape
            strides, off = (H * W * Cn, W * Cn, Cn), channel
        else:
            n, H, W = frames.shape
            strides, off = (H * W, W, 1), 0
        if H > self.max_frame[0] or W > self.max_frame[1]:
            raise RuntimeError(f"frames of {H}x{W} exceed the encoder workspace {self.max_frame}; create the encoder with "
                               "max_frame=(H, W) (the asynchronous form cannot rebuild the handle with batches in flight)")
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self._lib.cbas_enc_submit_u8(self._h, slot, frames.data_ptr() + off, n, H, W, *strides,
                                                out32.data_ptr() if out32 is not None else None,
                                                out16.data_ptr() if out16 is not None else None, stream),
                   "cbas_enc_submit_u8")

    def set_lanes(self, n: int) -> None:
        """1 or 2 compute lanes for the asynchronous forms (default 2: two batches in flight)."""
        _lib.check(self._lib.cbas_enc_set_lanes(self._h, int(n)), "cbas_enc_set_lanes")

    def set_prune_last_layer(self, enable: bool) -> None:
        """Default on: the last layer computes q/attention/MLP for the CLS rows only (bit-identical CLS)."""
        _lib.check(self._lib.cbas_enc_set_prune_last_layer(self._h, int(bool(enable))), "cbas_enc_set_prune_last_layer")

    def debug_option(self, name: str, value: int) -> None:
        """Bring-up / tests: switch an implementation detail whose settings are bit-identical (cbas_enc_debug_option;
        debug build of the library only)."""
        _lib.require_debug("cbas_enc_debug_option")
        _lib.check(self._lib.cbas_enc_debug_option(self._h, name.encode(), int(value)), "cbas_enc_debug_option")

    def wait_stream(self, slot: int) -> None:
        """The current torch stream waits for ``slot``'s batch (no host synchronisation); frees the slot."""
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self._lib.cbas_enc_wait_stream(self._h, slot, stream), "cbas_enc_wait_stream")

    def wait(self, slot: int, want_f32: bool = False):
        n = getattr(self, "_slot_n", {}).pop(slot, None)
        if n is None:
            raise RuntimeError(f"cbas_enc_wait: slot {slot} has no submitted work")
        D = self.config.hidden_size
        o16 = np.empty((n, D), np.float16)
        o32 = np.empty((n, D), np.float32) if want_f32 else None
        _lib.check(self._lib.cbas_enc_wait(self._h, slot, o16.ctypes.data, o32.ctypes.data if want_f32 else None),
                   "cbas_enc_wait")
        return o16, o32

    def check_finite(self) -> None:
        """Raise if a batch completed since the last check produced a NaN / infinite CLS row (an activation left the range of
        the arithmetic mode: cbas_enc_check_finite).  ``wait`` and the fused session's waits check by themselves; callers of the
        stream-ordered forms (``encode_u8``, ``submit`` + ``wait_stream``) call this after synchronising."""
        _lib.check(self._lib.cbas_enc_check_finite(self._h), "cbas_enc_check_finite")

    # -- per-kernel timing (HIP events inside the library) ------------------------------------------
    def profile(self, enable: bool) -> None:
        _lib.check(self._lib.cbas_enc_profile(self._h, int(bool(enable))), "cbas_enc_profile")

    def profile_read(self, reset: bool = True) -> Dict[str, Dict[str, float]]:
        n = len(_lib.PROF_CATS)
        ms = (C.c_double * n)()
        cnt = (C.c_int64 * n)()
        fl = (C.c_double * n)()
        _lib.check(self._lib.cbas_enc_profile_read(self._h, ms, cnt, fl, int(reset)), "cbas_enc_profile_read")
        return {name: {"ms": ms[i], "launches": int(cnt[i]), "flops": fl[i]}
                for i, name in enumerate(_lib.PROF_CATS) if cnt[i] > 0}

    # -- bring-up taps -----------------------------------------------------------------------------
    def debug_tap(self, frames: torch.Tensor, stop_layer: int, stop_stage: int, which: int, channel: int = 1):
        _lib.require_debug("cbas_enc_debug_forward_u8")
        n, H, W = frames.shape[:3]
        if frames.dim() == 4:
            Cn = frames.shape[3]
            strides, off = (H * W * Cn, W * Cn, Cn), channel
        else:
            strides, off = (H * W, W, 1), 0
        torch.cuda.synchronize(self.device)
        _lib.check(self._lib.cbas_enc_debug_forward_u8(self._h, frames.data_ptr() + off, n, H, W, *strides,
                                                       stop_layer, stop_stage), "cbas_enc_debug_forward_u8")
        T = self.config.num_tokens(H, W)
        D, F = self.config.hidden_size, self.config.intermediate_size
        if self.precision == 4 and which in (1, 3):
            raise RuntimeError("precision 4 keeps the LayerNorm / GELU buffers in the GEMM's split hi|lo tile order; tap precision 3")
        act = np.float32 if self.precision >= 3 else np.float16        # precision 3 / 4 keep every activation buffer fp32
        shape, dt = {0: ((n * T, D), np.float32), 1: ((n * T, D), act), 2: ((n * T, 3 * D), act),
                     3: ((n * T, F), act)}[which]
        out = np.empty(shape, dt)
        _lib.check(self._lib.cbas_enc_debug_read(self._h, which, out.ctypes.data, out.nbytes), "cbas_enc_debug_read")
        return out

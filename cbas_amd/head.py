"""``ClassifierLSTMDeltas`` — drop-in for the reference's v3 classifier head
(backend/classifier_head.py:57-172) backed by the fp32 HIP kernels of libcbas_mi355x.so.

Same constructor arguments, ``load_state_dict`` / ``to`` / ``eval`` / ``parameters`` surface that
the reference's model-bundle loader and ``infer_file`` use (backend/workthreads.py:427-447,
backend/cbas.py:477-479, 544), and the same call contract: ``model(x)`` with ``x`` float32
``(B, seq_len, in_features)`` returns ``(logits (B, C), latent (B, 2h))``.

``infer_clip`` is the whole window loop of ``infer_file`` for one clip (fp16 CLS rows in, softmax
probabilities out) without materialising the 31x redundant windows.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Mapping, Optional

import numpy as np
import torch

from . import _lib
from .config import HeadConfig
from .weights import head_param_shapes

_ORDER_PREFIX = ["gate", "attention_temp"]


def pack_head_weights(cfg: HeadConfig, w: Mapping[str, np.ndarray]) -> np.ndarray:
    """Flatten a ClassifierLSTMDeltas state dict into the blob order of include/cbas_mi355x.h."""
    names = list(_ORDER_PREFIX)
    streams = ("cls", "delta", "acc") if cfg.use_acceleration else ("cls", "delta")
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
    shapes = head_param_shapes(cfg)
    missing = [n for n in names if n not in w]
    if missing:
        raise KeyError(f"state dict lacks {missing}; the MI355X head needs every parameter")
    parts = []
    for n in names:
        a = np.asarray(w[n], dtype=np.float32)
        if tuple(a.shape) != tuple(shapes[n]):
            raise ValueError(f"{n}: shape {tuple(a.shape)} != expected {tuple(shapes[n])}")
        parts.append(a.reshape(-1))
    return np.ascontiguousarray(np.concatenate(parts))


def _to_numpy(v) -> np.ndarray:
    if isinstance(v, torch.Tensor):
        return v.detach().to("cpu", torch.float32).numpy()
    return np.asarray(v, dtype=np.float32)


class ClassifierLSTMDeltas:
    def __init__(self, in_features, out_features, seq_len=31, bottleneck_dim=128, dropout_p=0.15,
                 use_acceleration=True, ema_alpha=0.3, center_window_size=5, lstm_hidden_size=64, lstm_layers=1):
        self.config = HeadConfig(in_features=in_features, out_features=out_features, seq_len=seq_len,
                                 bottleneck_dim=bottleneck_dim, use_acceleration=use_acceleration,
                                 ema_alpha=ema_alpha, center_window_size=center_window_size,
                                 lstm_hidden_size=lstm_hidden_size, lstm_layers=lstm_layers)
        self.config.validate()
        self.in_features, self.out_features = in_features, out_features
        self.seq_len, self.sw, self.hsl = seq_len, center_window_size, seq_len // 2
        self.device: Optional[torch.device] = None
        self._weights: Optional[Dict[str, np.ndarray]] = None
        self._h = None
        self._lib = None

    # -- nn.Module-like surface ------------------------------------------------------------------
    # initial values of the two scalar parameters (classifier_head.py:90, 96): what a non-strict load keeps when the
    # checkpoint lacks them
    _SCALAR_DEFAULTS = {"gate": 0.2, "attention_temp": 1.0}

    def load_state_dict(self, state_dict, strict: bool = True):
        """``strict=False`` (how the reference's bundle loader calls it, workthreads.py:441): entries this architecture
        does not have are ignored and the scalar parameters may be missing.  A missing TENSOR is an error in both
        modes: torch would silently keep its random initialisation for it."""
        from .weights import head_param_shapes
        shapes = head_param_shapes(self.config)
        w = {k: _to_numpy(v) for k, v in state_dict.items()}
        extra = [k for k in w if k not in shapes]
        missing = [k for k in shapes if k not in w]
        if strict and (extra or missing):
            raise RuntimeError(f"Error(s) in loading state_dict: missing {missing}, unexpected {extra}")
        for k in list(missing):
            if k in self._SCALAR_DEFAULTS:
                w[k] = np.asarray(self._SCALAR_DEFAULTS[k], np.float32)
                missing.remove(k)
        if missing:
            raise RuntimeError(f"state_dict lacks {missing}: the MI355X head will not run on uninitialised weights")
        self._weights = {k: w[k] for k in shapes}
        self._destroy()
        return self

    def state_dict(self):
        """CPU torch tensors keyed like the reference module's state dict, so that
        ``torch.save(model.state_dict(), ...)`` (workthreads.py:856) writes a loadable model.pth."""
        return {k: torch.from_numpy(np.array(v, dtype=np.float32, copy=True)) for k, v in (self._weights or {}).items()}

    def to(self, device):
        device = torch.device(device)
        if device != self.device:
            self._destroy()
            self.device = device
        return self

    def eval(self):
        return self

    def parameters(self):
        if self.device is not None and self._weights is not None:
            yield torch.as_tensor(self._weights["gate"]).to(self.device)

    def _destroy(self):
        if self._h is not None:
            self._lib.cbas_head_destroy(self._h)
            self._h = None

    def close(self):
        self._destroy()

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass

    def _ensure(self):
        if self._h is not None:
            return
        if self._weights is None:
            raise RuntimeError("ClassifierLSTMDeltas: load_state_dict() must be called before inference")
        if self.device is None or self.device.type != "cuda":
            raise RuntimeError(f"the MI355X head runs only on a GPU device (got {self.device}); there is no CPU path")
        dev = self.device.index if self.device.index is not None else torch.cuda.current_device()
        cfg = self.config
        self._lib = _lib.load()
        cc = _lib.HeadConfigC(cfg.in_features, cfg.out_features, cfg.seq_len, cfg.bottleneck_dim, cfg.lin0_dim,
                              cfg.lstm_hidden_size, cfg.center_window_size, cfg.ema_alpha, cfg.lstm_layers,
                              int(cfg.use_acceleration))
        blob = pack_head_weights(cfg, self._weights)
        need = self._lib.cbas_head_weights_count(C.byref(cc))
        if need != blob.shape[0]:
            raise RuntimeError(f"head weight blob has {blob.shape[0]} floats, library expects {need}")
        h = C.c_void_p()
        _lib.check(self._lib.cbas_head_create(C.byref(cc), blob.ctypes.data, blob.shape[0], dev, C.byref(h)),
                   "cbas_head_create")
        self._h = h

    # -- the reference call: model(x) -> (logits, latent) -----------------------------------------
    def __call__(self, x: torch.Tensor):
        return self.forward(x)

    def forward(self, x: torch.Tensor):
        self._ensure()
        if x.dim() != 3 or x.shape[1] != self.seq_len or x.shape[2] != self.in_features:
            raise ValueError(f"expected (B, {self.seq_len}, {self.in_features}), got {tuple(x.shape)}")
        x = x.to(self.device, torch.float32).contiguous()
        B = x.shape[0]
        logits = torch.empty((B, self.out_features), dtype=torch.float32, device=self.device)
        latent = torch.empty((B, 2 * self.config.lstm_hidden_size), dtype=torch.float32, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self._lib.cbas_head_forward_windows(self._h, x.data_ptr(), B, logits.data_ptr(),
                                                       latent.data_ptr(), stream), "cbas_head_forward_windows")
        return logits, latent

    # -- whole-clip sliding-window inference --------------------------------------------------------
    def infer_clip(self, cls_rows: torch.Tensor, temperature: float = 1.0, want_logits: bool = False):
        """cls_rows (N, in_features) on the device, float16 (what ``_cls.h5`` holds) or float32 (a foreign ``cls``
        dataset after the reference's ``.float()``, backend/cbas.py:507-508) -> probs (N, C) float32 [, logits]."""
        self._ensure()
        assert cls_rows.dtype in (torch.float16, torch.float32) and cls_rows.is_cuda and cls_rows.dim() == 2
        cls_rows = cls_rows.contiguous()
        n = cls_rows.shape[0]
        probs = torch.empty((n, self.out_features), dtype=torch.float32, device=self.device)
        logits = torch.empty((n, self.out_features), dtype=torch.float32, device=self.device) if want_logits else None
        if n:
            self._range(cls_rows, n, 0, n, probs, logits, temperature)
        return (probs, logits) if want_logits else probs

    def _range(self, rows: torch.Tensor, n_frames: int, first: int, count: int, probs: Optional[torch.Tensor],
               logits: Optional[torch.Tensor], temperature: float) -> None:
        stream = torch.cuda.current_stream(self.device).cuda_stream
        half_rows = rows.dtype == torch.float16
        fn = self._lib.cbas_head_infer_f16_range if half_rows else self._lib.cbas_head_infer_f32_range
        _lib.check(fn(self._h, rows.data_ptr(), n_frames, first, count, float(temperature),
                      probs.data_ptr() if probs is not None else None,
                      logits.data_ptr() if logits is not None else None, stream),
                   "cbas_head_infer_f16_range" if half_rows else "cbas_head_infer_f32_range")

    def infer_range_into(self, cls_rows: torch.Tensor, n_frames: int, first: int, count: int,
                         probs_out: torch.Tensor, temperature: float = 1.0) -> None:
        """Classify frames [first, first+count) of a clip whose first ``n_frames`` CLS rows (float16 or float32) are in
        ``cls_rows``; writes ``probs_out[first:first+count]`` (asynchronous on the current stream)."""
        self._ensure()
        assert cls_rows.dtype in (torch.float16, torch.float32) and cls_rows.is_contiguous()
        self._range(cls_rows, n_frames, first, count, probs_out[first:first + count], None, temperature)


def from_reference_module(module, device) -> ClassifierLSTMDeltas:
    """Build the MI355X head from an instantiated reference ``classifier_head.ClassifierLSTMDeltas``."""
    sd = module.state_dict()
    h = int(sd["attention_head.weight"].shape[1]) // 2
    m = ClassifierLSTMDeltas(module.in_features, module.out_features, seq_len=module.seq_len,
                             center_window_size=module.sw, ema_alpha=module.ema_alpha, lstm_hidden_size=h,
                             lstm_layers=int(getattr(module.lstm, "num_layers", 1)),
                             use_acceleration=bool(getattr(module, "use_acceleration", "acc_bottleneck.0.weight" in sd)))
    m.load_state_dict(sd)
    return m.to(device)

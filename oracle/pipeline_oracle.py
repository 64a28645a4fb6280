"""ORACLE (test infrastructure, not product code): the whole reference path on the CPU in float32,
``encode_file`` semantics (backend/cbas.py:423-440) followed by ``infer_file`` semantics
(backend/cbas.py:497-551), built from the restatements in vit_oracle.py / head_oracle.py.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np

from . import head_oracle as H
from . import vit_oracle as V


def encode_frames(frames_u8: np.ndarray, enc_w: Dict[str, np.ndarray], cfg, batch: int = 8) -> np.ndarray:
    """(N,H,W,3) uint8 -> CLS (N,D) float32: green/255 (cbas.py:431), gray->3ch (cbas.py:674),
    ViT, CLS row (cbas.py:677), ``batch`` frames per model call."""
    g = V.preprocess_green(frames_u8)
    outs = []
    for i in range(0, g.shape[0], batch):
        px = np.repeat(g[i:i + batch, None], 3, axis=1)
        outs.append(V.vit_forward(px, enc_w, cfg)[:, 0, :])
    return np.concatenate(outs, axis=0)


def classify_cls(cls_f16: np.ndarray, head_w: Dict[str, np.ndarray], seq_len: int = 31,
                 temperature: float = 1.0, batch: int = 512) -> np.ndarray:
    """fp16 CLS rows (what _cls.h5 holds) -> probabilities (N,C): windows are materialised one per
    frame with replicate edge padding and pushed through the head in batches, as infer_file does."""
    idx = H.infer_windows(cls_f16, seq_len)
    x32 = cls_f16.astype(np.float32)
    out = []
    for i in range(0, idx.shape[0], batch):
        logits, _ = H.head_forward(x32[idx[i:i + batch]], head_w, seq_len)
        out.append(H.softmax_T(logits, temperature))
    return np.concatenate(out, axis=0)


def encode_and_classify(frames_u8, enc_w, cfg, head_w, seq_len: int = 31, batch: int = 8,
                        temperature: float = 1.0) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    cls32 = encode_frames(frames_u8, enc_w, cfg, batch)
    cls16 = cls32.astype(np.float16)            # the f4 -> f2 cast of the HDF5 write (cbas.py:420,438)
    probs = classify_cls(cls16, head_w, seq_len, temperature)
    return cls32, cls16, probs

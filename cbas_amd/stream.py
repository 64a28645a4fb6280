"""Fused streaming session: encode -> fp16 CLS -> sliding-window head, without the CLS rows ever
leaving HBM.  This is the reference's two-step pipeline (EncodeThread writes ``_cls.h5``,
ClassificationThread reads it back: backend/workthreads.py:316-328, 488-498) collapsed into one
pass for live inference; the numerical contract is unchanged because the head still consumes the
CLS rows *after* their round-to-fp16 (what the file would have held).

A segment of frames is classified as soon as its right-hand context (``seq_len // 2`` rows) has
been encoded, in groups of ``classify_every`` frames so the head kernels launch with full grids.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from .encoder import DinoEncoder
from .head import ClassifierLSTMDeltas


class ClipStream:
    def __init__(self, encoder: DinoEncoder, head: ClassifierLSTMDeltas, capacity: int,
                 temperature: float = 1.0, classify_every: int = 1024):
        self.enc, self.head = encoder, head
        self.capacity = int(capacity)
        self.temperature = float(temperature)
        self.classify_every = int(classify_every)
        dev = encoder.device
        self.cls16 = torch.empty((self.capacity, encoder.config.hidden_size), dtype=torch.float16, device=dev)
        self.probs = torch.empty((self.capacity, head.out_features), dtype=torch.float32, device=dev)
        self.half = head.seq_len // 2
        self.reset()

    def reset(self) -> None:
        self._drain()
        self.encoded = 0
        self.classified = 0
        self.landed = 0          # rows whose batches the current stream is already ordered after

    def _drain(self) -> None:
        """Order the current stream after every batch still in flight on the encoder's compute lanes."""
        busy = getattr(self, "_busy", {})
        for slot in sorted(busy, key=lambda s: busy[s][1]):          # submission order
            self.enc.wait_stream(slot)
            self.landed += busy[slot][2]
        self._busy = {}

    def push_u8(self, frames: torch.Tensor, channel: int = 1) -> None:
        """Encode one batch of uint8 frames resident in HBM ((n,H,W,3) or (n,H,W)) and classify every
        frame whose window is now complete."""
        n = frames.shape[0]
        if self.encoded + n > self.capacity:
            raise RuntimeError(f"ClipStream capacity {self.capacity} exceeded")
        self._encode_into(frames, channel, self.cls16[self.encoded:self.encoded + n])
        self.encoded += n
        # The head only consumes rows the current stream is ALREADY ordered after (batches whose slot has been
        # recycled): classifying never waits for the batches in flight, so the encoder lanes are not drained
        # in the middle of a clip.  It therefore trails the encoder by up to ENC_SLOTS batches + the half window.
        ready = self.landed - self.half - self.classified            # frames with full right context
        if ready >= self.classify_every:
            self._classify(ready, self.landed)

    def _encode_into(self, frames: torch.Tensor, channel: int, out16: torch.Tensor) -> None:
        from . import _lib
        enc = self.enc
        if frames.dim() == 4:
            n, H, W, Cn = frames.shape
            strides, off = (H * W * Cn, W * Cn, Cn), channel
        else:
            n, H, W = frames.shape
            strides, off = (H * W, W, 1), 0
        # asynchronous submissions on the encoder's two compute lanes: consecutive batches overlap
        frames = frames.contiguous()
        for i in range(0, n, enc.max_batch):
            m = min(enc.max_batch, n - i)
            slot = self._next_slot = (getattr(self, "_next_slot", -1) + 1) % _lib.ENC_SLOTS
            if slot in self._busy:
                enc.wait_stream(slot)                   # batches are recycled in submission order
                self.landed += self._busy.pop(slot)[2]
            sub = frames[i:i + m]
            enc.submit_dev(slot, sub, out16[i:i + m], None, channel)
            self._seq = getattr(self, "_seq", 0) + 1
            self._busy[slot] = (sub, self._seq, m)      # keeps the frames alive until the slot is waited for

    def _classify(self, count: int, n_rows: int) -> None:
        """Classify frames [classified, classified + count) of the first ``n_rows`` rows (all landed)."""
        self.head.infer_range_into(self.cls16, n_rows, self.classified, count, self.probs, self.temperature)
        self.classified += count

    def finish(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """Classify the tail (the clip's right edge replicates its last row) and return
        (cls_f16 (N,D), probs (N,C)) views on the device."""
        self._drain()
        if self.encoded > self.classified:
            self._classify(self.encoded - self.classified, self.encoded)
        return self.cls16[:self.encoded], self.probs[:self.encoded]

"""Fused streaming session: encode -> fp16 CLS -> sliding-window head, without the CLS rows ever
leaving HBM.  This is the reference's two-step pipeline (EncodeThread writes ``_cls.h5``,
ClassificationThread reads it back: backend/workthreads.py:316-328, 488-498) collapsed into one
pass for live inference; the numerical contract is unchanged because the head still consumes the
CLS rows *after* their round-to-fp16 (what the file would have held).

The session itself lives behind the C ABI (``cbas_fused_*`` in include/cbas_mi355x.h, csrc/api_fused.hip):
slot rotation over the encoder's compute lanes, the "classify what has landed" rule and the clip buffers are
native; this class is the torch-tensor view of it.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib
from .encoder import DinoEncoder
from .head import ClassifierLSTMDeltas


def _layout(frames, channel: int):
    if frames.ndim == 4:
        n, H, W, Cn = frames.shape
        return n, H, W, (H * W * Cn, W * Cn, Cn), channel
    n, H, W = frames.shape
    return n, H, W, (H * W, W, 1), 0


class _ClipWait:
    """``synchronize()`` = the clip a session finished asynchronously is complete in host memory."""

    def __init__(self, session: "ClipStream"):
        self._s = session

    def synchronize(self) -> None:
        self._s.wait()


class ClipStream:
    """``head=None`` gives an encode-only session: the chunk loop of ``encode_file`` with the rows kept in HBM."""

    def __init__(self, encoder: DinoEncoder, head: Optional[ClassifierLSTMDeltas], capacity: int,
                 temperature: float = 1.0, classify_every: int = 1024):
        self.enc, self.head = encoder, head
        self.capacity = int(capacity)
        self.temperature = float(temperature)
        self.classify_every = int(classify_every)
        if head is not None:
            head.to(encoder.device)
            head._ensure()
        self._lib = _lib.load()
        self._keep = []              # frames of the open clip (device pushes must stay valid until finish)
        h = C.c_void_p()
        with torch.cuda.device(encoder.device):
            _lib.check(self._lib.cbas_fused_create(encoder._h, head._h if head is not None else None, self.capacity,
                                                   self.temperature, self.classify_every, C.byref(h)), "cbas_fused_create")
        self._h = h
        self.encoded = 0
        # the native session drains through the encoder handle when it is destroyed: the encoder closes its open
        # sessions before it frees that handle (DinoEncoder.close), whichever of the two is dropped first
        encoder._register_session(self)

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.cbas_fused_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def reset(self) -> None:
        _lib.check(self._lib.cbas_fused_reset(self._h), "cbas_fused_reset")
        self._keep = []
        self.encoded = 0
        self._stream_rows = None

    def push_u8(self, frames: torch.Tensor, channel: int = 1) -> None:
        """Append uint8 frames resident in HBM ((n,H,W,3) or (n,H,W)); classifies every frame whose window is
        complete among the batches that have already landed."""
        assert frames.dtype == torch.uint8 and frames.is_cuda
        frames = frames.contiguous()
        n, H, W, strides, off = _layout(frames, channel)
        stream = torch.cuda.current_stream(self.enc.device).cuda_stream
        _lib.check(self._lib.cbas_fused_push_u8(self._h, frames.data_ptr() + off, n, H, W, *strides, stream),
                   "cbas_fused_push_u8")
        self._keep.append(frames)
        self.encoded += n

    def push_host(self, frames: np.ndarray, channel: int = 1) -> None:
        """Append uint8 frames from host memory (numpy (n,H,W,3) or (n,H,W), C-contiguous).  Pinned arrays
        (e.g. ``torch.Tensor.pin_memory().numpy()``) are DMA'd directly and must stay untouched until finish."""
        assert frames.dtype == np.uint8 and frames.flags.c_contiguous
        n, H, W, strides, off = _layout(frames, channel)
        _lib.check(self._lib.cbas_fused_push_u8_host(self._h, frames.ctypes.data + off, n, H, W, *strides),
                   "cbas_fused_push_u8_host")
        self._keep.append(frames)
        self.encoded += n

    def _views(self, p16, pp, n):
        D, Cn = self.enc.config.hidden_size, (self.head.out_features if self.head is not None else 0)
        dev = self.enc.device
        session = self

        def view(ptr, rows, cols, dtype, itemsize):
            if rows == 0 or cols == 0 or not ptr:
                return torch.empty((rows if cols else 0, cols), dtype=dtype, device=dev)
            iface = {"shape": (rows, cols), "typestr": "<f%d" % itemsize, "data": (ptr, False), "version": 2}
            # the tensor references the holder, the holder the session: the views keep the native buffers alive
            holder = type("_Dev", (), {"__cuda_array_interface__": iface, "owner": session})()
            return torch.as_tensor(holder, device=dev)
        return view(p16.value, n, D, torch.float16, 2), view(pp.value, n, Cn, torch.float32, 4)

    def finish(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """Classify the tail and return (cls_f16 (N,D), probs (N,C)) views of the session's device buffers,
        ordered on the current torch stream (valid until the next reset / push)."""
        p16, pp, n = C.c_void_p(), C.c_void_p(), C.c_int64(0)
        stream = torch.cuda.current_stream(self.enc.device).cuda_stream
        _lib.check(self._lib.cbas_fused_finish(self._h, None, None, C.byref(p16), C.byref(pp), C.byref(n), stream),
                   "cbas_fused_finish")
        return self._views(p16, pp, int(n.value))

    def finish_host_async(self):
        """``finish_host`` without blocking: the tail classification and the copy-out are queued, and the call returns
        (cls_f16, probs, waiter) - page-locked numpy arrays that are complete once ``waiter.synchronize()`` has returned
        (``cbas_fused_finish_async`` / ``cbas_fused_wait``).  The next clip can be pushed through ANOTHER session meanwhile
        (this session's buffers are in use until then)."""
        D, Cn = self.enc.config.hidden_size, (self.head.out_features if self.head is not None else 0)
        o16 = torch.empty((self.encoded, D), dtype=torch.float16, pin_memory=True).numpy()
        opr = torch.empty((self.encoded, Cn), dtype=torch.float32, pin_memory=True).numpy() if Cn else np.empty((self.encoded, 0), np.float32)
        n = C.c_int64(0)
        _lib.check(self._lib.cbas_fused_finish_async(self._h, o16.ctypes.data, opr.ctypes.data if Cn else None, C.byref(n)),
                   "cbas_fused_finish_async")
        assert int(n.value) == self.encoded
        return o16, opr, _ClipWait(self)

    def stream_rows_to(self, n_rows: int) -> np.ndarray:
        """Right after ``reset``: rows leave for a page-locked (n_rows, D) float16 array WHILE the clip runs
        (``cbas_fused_stream_rows``); returns that array.  ``rows_ready`` says how many leading rows are complete;
        ``finish_host`` then copies only the remainder and returns the same array."""
        D = self.enc.config.hidden_size
        self._stream_rows = torch.empty((int(n_rows), D), dtype=torch.float16, pin_memory=True).numpy()
        _lib.check(self._lib.cbas_fused_stream_rows(self._h, self._stream_rows.ctypes.data), "cbas_fused_stream_rows")
        return self._stream_rows

    def rows_ready(self, block: bool = False) -> int:
        """Leading rows complete in the ``stream_rows_to`` array (one consumer thread; may run beside pushes)."""
        r = int(self._lib.cbas_fused_rows_ready(self._h, 1 if block else 0))
        if r < 0:
            raise RuntimeError("cbas_fused_rows_ready failed")
        return r

    def wait(self) -> None:
        """Block until the clip queued by ``finish_host_async`` is complete."""
        if self._h:
            _lib.check(self._lib.cbas_fused_wait(self._h), "cbas_fused_wait")

    def finish_host(self) -> Tuple[np.ndarray, np.ndarray]:
        """Classify the tail and copy the clip out: (cls_f16 (N,D) float16, probs (N,C) float32) numpy arrays."""
        D, Cn = self.enc.config.hidden_size, (self.head.out_features if self.head is not None else 0)
        # page-locked destinations: a device -> PAGEABLE host copy goes through the runtime's slow staged path (measured:
        # 7.7 ms for 4 MB, i.e. ~30 ms of a 10 000-frame clip's 460 ms - most of what the host path lost against
        # HBM-resident frames in round 2); torch's caching host allocator makes these allocations cheap after the first
        streamed = getattr(self, "_stream_rows", None)
        if streamed is not None:
            assert streamed.shape[0] >= self.encoded
            o16 = streamed[:self.encoded]
        else:
            o16 = torch.empty((self.encoded, D), dtype=torch.float16, pin_memory=True).numpy()
        opr = torch.empty((self.encoded, Cn), dtype=torch.float32, pin_memory=True).numpy() if Cn else np.empty((self.encoded, 0), np.float32)
        n = C.c_int64(0)
        _lib.check(self._lib.cbas_fused_finish(self._h, o16.ctypes.data, opr.ctypes.data if Cn else None, None, None,
                                               C.byref(n), None), "cbas_fused_finish")
        assert int(n.value) == self.encoded
        return o16, opr

"""File-level drop-ins for the reference's hot path:

* ``encode_file(encoder, path, progress_callback=None) -> str | None``   backend/cbas.py:399-456
* ``infer_file(file_path, model, dataset_name, behaviors, seq_len, device=None, temperature=1.0)
  -> str | None``                                                         backend/cbas.py:458-572

Same signatures, return values, file names, on-disk formats and error behaviour (``encode_file``
re-raises after removing its ``.tmp``; ``infer_file`` never raises and returns ``None``).  What
changes is the execution: frames go uint8 -> pinned staging -> HBM on a copy stream while the
previous sub-batch is still in the ViT kernels, and the head runs once over the clip instead of
once per materialised window.
"""
from __future__ import annotations

import contextlib
import os
import shutil
import threading
import traceback
from collections import deque
from typing import Callable, List, Optional, Tuple

import numpy as np
import torch

from . import _lib, h5io
from .encoder import DinoEncoder
from .framesource import MJPEGAviSource, PipeFrameSource, Y4MFileSource
from .head import ClassifierLSTMDeltas, from_reference_module

CHUNK_SIZE = 512                     # backend/cbas.py:48
SCHEMA_VERSION = "1.0"               # backend/cbas.py:416


# ------------------------------------------------------------------------------------------------
# frame sources (decord.VideoReader replacement is pluggable; decord itself is used when present)
# ------------------------------------------------------------------------------------------------
class NpyFrameSource:
    """(N,H,W,3) uint8 array stored with ``np.save`` and memory-mapped; used for synthetic clips."""

    def __init__(self, path: str):
        self._a = np.load(path, mmap_mode="r")
        if self._a.dtype != np.uint8 or self._a.ndim != 4 or self._a.shape[3] != 3:
            raise ValueError(f"{path}: expected a uint8 (N,H,W,3) array, got {self._a.dtype} {self._a.shape}")

    def __len__(self):
        return self._a.shape[0]

    def get_batch(self, indices) -> np.ndarray:
        idx = list(indices)
        if idx and idx == list(range(idx[0], idx[0] + len(idx))):
            return np.ascontiguousarray(self._a[idx[0]:idx[0] + len(idx)])
        return np.ascontiguousarray(self._a[idx])

    frame_shape = property(lambda self: tuple(self._a.shape[1:]))

    def read_channel_into(self, start: int, stop: int, channel: int, out: np.ndarray) -> None:
        """Channel ``channel`` of frames [start, stop) straight from the page cache into ``out`` (n,H,W) - a page-locked
        ring piece: the native byte shuffle on a few threads (cbas_pick_channel_u8), one pass over the file's bytes."""
        pick_channel(self._a[start:stop], channel, out)

    def read_into(self, start: int, stop: int, out: np.ndarray) -> None:
        """Frames [start, stop) straight from the page cache into ``out`` (a page-locked buffer): one copy, split over
        a few threads when it is large (numpy releases the GIL while it copies; one thread moves ~4 GB/s, and a GPU that
        encodes 22 k frames/s consumes 3.3 GB/s of 224 x 224 RGB)."""
        n = stop - start
        if n * int(np.prod(self._a.shape[1:])) < (16 << 20) or n < 4:
            np.copyto(out, self._a[start:stop])
            return
        cuts = [start + (n * k) // 4 for k in range(5)]
        list(_copy_pool().map(lambda ab: np.copyto(out[ab[0] - start:ab[1] - start], self._a[ab[0]:ab[1]]),
                              zip(cuts[:-1], cuts[1:])))


def pick_channel(frames: np.ndarray, channel: int, out: np.ndarray, threads: int = 4) -> None:
    """out[n,H,W] = frames[n,H,W,C][..., channel] for C-contiguous uint8 frames (backend/cbas.py:431 keeps channel 1),
    through the library's host helper (SSSE3 byte shuffle, threads)."""
    assert frames.dtype == np.uint8 and frames.ndim == 4 and frames.flags.c_contiguous, (frames.dtype, frames.shape)
    assert out.dtype == np.uint8 and out.shape == frames.shape[:3] and out.flags.c_contiguous
    lib = _lib.load()
    _lib.check(lib.cbas_pick_channel_u8(frames.ctypes.data, int(np.prod(frames.shape[:3])), int(frames.shape[3]), int(channel),
                                        out.ctypes.data, int(threads)), "cbas_pick_channel_u8")


_COPY_POOL = None


def _copy_pool():
    global _COPY_POOL
    if _COPY_POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _COPY_POOL = ThreadPoolExecutor(max_workers=4, thread_name_prefix="cbas-copy")
    return _COPY_POOL


class ArrayFrameSource:
    def __init__(self, frames: np.ndarray):
        self._a = frames

    def __len__(self):
        return self._a.shape[0]

    def get_batch(self, indices) -> np.ndarray:
        idx = list(indices)
        return np.ascontiguousarray(self._a[idx[0]:idx[0] + len(idx)]) if idx else self._a[:0]


class _DecordSource:
    def __init__(self, path: str):
        import decord  # noqa: WPS433  (present in a CBAS install, absent in the build image)
        self._path = path
        self._r = decord.VideoReader(path, ctx=decord.cpu(0))   # backend/cbas.py:402

    def clone(self) -> "_DecordSource":
        """Another decoder instance on the same file: ``_ChunkStream`` runs several, each on every k-th 512-frame chunk
        (an H.264 decoder is sequential inside a GOP; instances on different parts of the file are what scales)."""
        return _DecordSource(self._path)

    def __len__(self):
        return len(self._r)

    def get_batch(self, indices) -> np.ndarray:
        return self._r.get_batch(indices).asnumpy()             # backend/cbas.py:425


_READERS = {".npy": NpyFrameSource, ".y4m": Y4MFileSource}


def register_reader(ext: str, factory: Callable[[str], object]) -> None:
    _READERS[ext.lower()] = factory


_source_check_done = False


def _cross_check_sources(path: str, reader) -> None:
    """Once per process, when BOTH decoders exist: the first frames' green plane from decord (what the reference consumes,
    backend/cbas.py:425,431) against the ffmpeg pipe's ``extractplanes=g`` plane (what this package falls back to when
    decord is missing).  The two go through different pixel-format conversions (decord: swscale to RGB24, then [:, :, 1];
    the pipe: the filter graph's own conversion to a planar-RGB format), and no environment this package was built in had
    either decoder, so their equality is UNVERIFIED: this check is what finds out, on the first video a real install opens.
    It never fails the encode; a mismatch is reported loudly because embeddings made through the two sources would differ."""
    global _source_check_done
    if _source_check_done or os.environ.get("CBAS_VERIFY_FRAME_SOURCE") == "0":
        return
    _source_check_done = True
    pipe = None
    try:
        n = min(16, len(reader))
        if n == 0:
            return
        a = np.asarray(reader.get_batch(range(n)))[:, :, :, 1]
        pipe = PipeFrameSource(path, prefetch_frames=n, n_frames=len(reader))
        b = np.asarray(pipe.get_batch(range(n)))
        if a.shape != b.shape:
            print(f"WARNING: cbas_amd frame-source check on {os.path.basename(path)}: decord frames are {a.shape[1:]}, the ffmpeg "
                  f"pipe delivers {b.shape[1:]}")
        elif not np.array_equal(a, b):
            d = np.abs(a.astype(np.int16) - b.astype(np.int16))
            print(f"WARNING: cbas_amd frame-source check on {os.path.basename(path)}: the ffmpeg pipe's green plane differs from "
                  f"decord's on {float((d > 0).mean()) * 100:.2f} % of the pixels of the first {n} frames (max |diff| {int(d.max())}); "
                  "embeddings made without decord will not match the reference's bit for bit")
        else:
            print(f"cbas_amd frame-source check: ffmpeg-pipe green plane == decord green plane on the first {n} frames")
    except Exception as e:  # noqa: BLE001 - a diagnostic must never break the encode
        print(f"cbas_amd frame-source check skipped: {e}")
    finally:
        if pipe is not None:
            pipe.close()


def open_video(path: str):
    ext = os.path.splitext(path)[1].lower()
    if ext in _READERS:
        return _READERS[ext](path)
    try:
        r = _DecordSource(path)
    except ImportError as e:
        if ext == ".avi":                            # Motion-JPEG AVI decodes in-process (cbas_mjpeg_decode); anything else falls through
            try:
                return MJPEGAviSource(path, planes=True)
            except ValueError:
                pass
        if shutil.which("ffmpeg") and shutil.which("ffprobe"):
            return PipeFrameSource(path)            # decoder process + reader thread (cbas_amd/framesource.py)
        raise RuntimeError(f"no frame source for {path!r}: decord is not installed, ffmpeg/ffprobe are not on PATH and "
                           f"no reader is registered for {ext!r}") from e
    if shutil.which("ffmpeg") and shutil.which("ffprobe"):
        _cross_check_sources(path, r)
    return r


# ------------------------------------------------------------------------------------------------
_project_stamp: Optional[str] = None


def set_project_stamp(encoder_model_identifier: Optional[str]) -> None:
    """What ``gui_state.proj.encoder_model_identifier`` is to the reference (backend/cbas.py:414-416)."""
    global _project_stamp
    _project_stamp = encoder_model_identifier


def _current_stamp() -> Optional[str]:
    if _project_stamp is not None:
        return _project_stamp
    try:  # inside a running CBAS process the reference's global state is authoritative
        import gui_state  # type: ignore
        if getattr(gui_state, "proj", None):
            return gui_state.proj.encoder_model_identifier
    except Exception:  # noqa: BLE001
        pass
    return None


FP8_TAG = "mx-fp8"


def file_attrs(encoder=None) -> dict:
    """Root attributes of a ``_cls.h5`` (backend/cbas.py:414-416): the project's encoder stamp + schema version when a
    project is set.  Rows made in the MX-fp8 throughput mode (precision 2) are NOT interchangeable with fp16 rows (CLS
    error ~6e-2 against the 1e-3 contract), so such files say so: an ``encoder_precision`` attribute, and a stamp with a
    ``#mx-fp8`` suffix - the reference's project loader (startup_page.py:100-117) and its model loader
    (workthreads.py:390-399) then treat them as made by a different encoder instead of consuming them silently."""
    fp8 = int(getattr(encoder, "precision", 0) or 0) == 2
    attrs = {}
    stamp = _current_stamp()
    if stamp:
        attrs = {"encoder_model_identifier": stamp + ("#" + FP8_TAG if fp8 else ""), "schema_version": SCHEMA_VERSION}
    if fp8:
        attrs["encoder_precision"] = FP8_TAG
    return attrs


def is_range_error(e: BaseException) -> bool:
    """CBAS_ERANGE surfaced through the shims: a frame's CLS row came out non-finite (cbas_enc_check_finite)."""
    return isinstance(e, RuntimeError) and "non-finite CLS row" in str(e)


def range_fallback_for(encoder, e: BaseException, path: str):
    """The precision-3 twin to re-encode `path` with, or None when `e` is not a range error (or CBAS_RANGE_FALLBACK=0, or the
    encoder already is precision 3).  The reference computes in fp32 and has no operand range limit: where its rows are
    finite ours must exist too, at precision 3's rate for that one video, with a line saying so."""
    if not is_range_error(e) or os.environ.get("CBAS_RANGE_FALLBACK", "1") == "0":
        return None
    if not isinstance(encoder, DinoEncoder) or encoder.precision == 3:
        return None
    print(f"cbas_amd: {os.path.basename(path)}: activations left the range of precision {encoder.precision}; this video is "
          "re-encoded in precision 3 (fp32 end to end, no range limit)")
    return encoder.range_fallback()


def encode_file(encoder: DinoEncoder, path: str, progress_callback=None, reader=None) -> Optional[str]:
    if not isinstance(encoder, DinoEncoder):
        raise TypeError("cbas_amd.encode_file needs a cbas_amd.DinoEncoder (the MI355X encoder); "
                        f"got {type(encoder).__name__}")
    own_reader = reader is None
    reader = reader if reader is not None else open_video(path)     # reader errors propagate (cbas.py:400-402)
    try:
        try:
            return _encode_from_reader(encoder, path, reader, progress_callback)
        except RuntimeError as e:
            twin = range_fallback_for(encoder, e, path)
            if twin is None:
                raise
            return _encode_from_reader(twin, path, reader, progress_callback)
    finally:
        if own_reader and hasattr(reader, "close"):
            reader.close()


class _WriterThread:
    """Runs ``ClsWriter.append`` / ``flush`` on a background thread so that HDF5 I/O overlaps frame staging and
    kernel submission (libhdf5 calls go through ctypes/h5py and release the GIL).  Order is preserved; the
    first error is re-raised on the caller's thread at the next call or at ``close``."""

    def __init__(self, writer):
        import queue
        import threading
        self._w, self._q, self._err = writer, queue.Queue(maxsize=8), None
        self._t = threading.Thread(target=self._run, name="cbas-h5-writer", daemon=True)
        self._t.start()

    def _run(self):
        while True:
            item = self._q.get()
            if item is None:
                return
            if self._err is not None:
                continue                                  # drain after a failure
            try:
                if isinstance(item, str):
                    self._w.flush()
                else:
                    self._w.append(item)
            except BaseException as e:  # noqa: BLE001
                self._err = e

    def _check(self):
        if self._err is not None:
            raise self._err

    def append(self, rows):
        self._check()
        self._q.put(rows)

    def flush(self):
        self._check()
        self._q.put("flush")

    def stop(self):
        """Finish queued work and join the thread (never raises; call ``check`` afterwards)."""
        if self._t.is_alive():
            self._q.put(None)
            self._t.join()

    check = _check


def _encode_from_reader(encoder: DinoEncoder, path: str, reader, progress_callback) -> Optional[str]:
    video_len = len(reader)
    if video_len == 0:
        print(f"Warning: Video {path} contains no frames. Skipping.")
        return None

    out_file_path = os.path.splitext(path)[0] + "_cls.h5"
    tmp_file_path = out_file_path + ".tmp"
    D = encoder.config.hidden_size
    nslots = _lib.ENC_SLOTS
    if _wants_pinned(encoder) and os.environ.get("CBAS_ENCODE_FILE_SLOTS") != "1":
        # The native encoder: the chunk loop runs inside an encode-only fused session (frames DMA'd from the page-locked
        # decode-ahead ring, rows collected in HBM, one copy-out) and the file is written in one append at the end -
        # the same bytes (tests/test_host_logic.py), `.tmp` + rename as ever, +4 % over the per-batch copy-out of the
        # slot loop below (which stays for stand-in encoders and as CBAS_ENCODE_FILE_SLOTS=1)
        try:
            out = _write_cls_while_encoding(path, encoder, lambda sink: _runner_for(encoder, None, 1.0).run(
                path, reader, progress_callback, rows_sink=sink))[0]
        except Exception as e:
            print(f"ERROR during encoding for {path}: {e}")
            raise
        print(f"Successfully encoded {os.path.basename(path)} to {os.path.basename(out)}")
        return out
    try:
        with h5io.ClsWriter(tmp_file_path, D, file_attrs(encoder)) as w_sync:
            w = _WriterThread(w_sync)
            try:
                _stream_chunks(encoder, reader, video_len, progress_callback, w, nslots)
            finally:
                w.stop()                     # the file is closed only after the writer thread has finished
            w.check()
        os.replace(tmp_file_path, out_file_path)
        print(f"Successfully encoded {os.path.basename(path)} to {os.path.basename(out_file_path)}")
        return out_file_path
    except Exception as e:
        print(f"ERROR during encoding for {path}: {e}")
        # leave the encoder reusable: retire whatever is still in flight
        try:
            for slot in range(nslots):
                if slot in getattr(encoder, "_slot_n", {}):
                    try:
                        encoder.wait(slot)
                    except Exception:  # noqa: BLE001
                        pass
        finally:
            if os.path.exists(tmp_file_path):
                try:
                    os.remove(tmp_file_path)
                except OSError:
                    pass
        raise e


def _write_cls_while_encoding(video_path: str, encoder, run):
    """``<video>_cls.h5`` with encode_file's file semantics (``.tmp`` first, stamp attributes, atomic rename, the ``.tmp``
    removed on failure: backend/cbas.py:410-421, 442-456), its rows arriving from ``run(sink)`` while the clip is encoded.
    Returns (path, what run returned)."""
    out_file_path = os.path.splitext(video_path)[0] + "_cls.h5"
    tmp_file_path = out_file_path + ".tmp"
    try:
        with h5io.ClsWriter(tmp_file_path, encoder.config.hidden_size, file_attrs(encoder)) as w:
            res = run(w)
            w.flush()
        os.replace(tmp_file_path, out_file_path)
    except BaseException:
        if os.path.exists(tmp_file_path):
            try:
                os.remove(tmp_file_path)
            except OSError:
                pass
        raise
    return out_file_path, res


class _MemWriter:
    """``ClsWriter``-shaped sink that keeps the rows in memory (the multi-GPU driver ships them to the writing rank)."""

    def __init__(self):
        self.parts: List[np.ndarray] = []

    def append(self, rows):
        self.parts.append(np.array(rows, copy=True))

    def flush(self):
        pass


def encode_rows(encoder: DinoEncoder, path: str, progress_callback=None, reader=None) -> Optional[np.ndarray]:
    """The chunk loop of ``encode_file`` without the file: (N, D) float16 CLS rows of one video, ``None`` for a video
    without frames.  Reader and encoder errors propagate, with the encoder left reusable."""
    own_reader = reader is None
    reader = reader if reader is not None else open_video(path)
    try:
        video_len = len(reader)
        if video_len == 0:
            print(f"Warning: Video {path} contains no frames. Skipping.")
            return None
        w = _MemWriter()
        try:
            _stream_chunks(encoder, reader, video_len, progress_callback, w, _lib.ENC_SLOTS)
        except Exception:
            for slot in range(_lib.ENC_SLOTS):
                if slot in getattr(encoder, "_slot_n", {}):
                    try:
                        encoder.wait(slot)
                    except Exception:  # noqa: BLE001
                        pass
            raise
        D = encoder.config.hidden_size
        return np.concatenate(w.parts) if w.parts else np.empty((0, D), np.float16)
    finally:
        if own_reader and hasattr(reader, "close"):
            reader.close()


def write_cls_file(video_path: str, rows: np.ndarray, attrs: Optional[dict] = None) -> str:
    """Write ``<video>_cls.h5`` for already-encoded rows with encode_file's file semantics (backend/cbas.py:410-421,
    442): ``.tmp`` first, stamp attributes when a project is set (``attrs`` = ``file_attrs(encoder)``; default: those of
    an fp16 encoder), atomic rename; the ``.tmp`` is removed on failure."""
    out_file_path = os.path.splitext(video_path)[0] + "_cls.h5"
    tmp_file_path = out_file_path + ".tmp"
    if attrs is None:
        attrs = file_attrs(None)
    try:
        with h5io.ClsWriter(tmp_file_path, rows.shape[1], attrs) as w:
            w.append(rows)           # the file's bytes do not depend on the append granularity (chunks are 8192 rows
            w.flush()                # either way; tests/test_host_logic.py compares with the 512-row chunk loop)
        os.replace(tmp_file_path, out_file_path)
    except Exception:
        if os.path.exists(tmp_file_path):
            try:
                os.remove(tmp_file_path)
            except OSError:
                pass
        raise
    return out_file_path


class _PinnedRing:
    """Page-locked chunk buffers the decode-ahead thread fills.  A sub-batch pushed from page-locked memory is DMA'd
    straight from it (cbas_enc_submit_u8_host: no staging copy on the submitting thread), so a buffer goes back to the
    decoder only once the consumer says its copies are done (``release``).  Buffers are kept per byte size for the life of
    the process: hipHostMalloc of ~80 MB costs milliseconds, a directory of videos would pay it per file."""

    _lock = threading.Lock()
    _pool: dict = {}                 # nbytes -> [torch uint8 tensors]
    def __init__(self, nbytes: int, depth: int = 4):
        self.nbytes, self.depth = int(nbytes), int(depth)       # depth 4: one being consumed, two queued, one being filled
        with self._lock:
            have = self._pool.setdefault(self.nbytes, [])
            self._bufs = [have.pop() for _ in range(min(len(have), self.depth))]
        while len(self._bufs) < self.depth:
            self._bufs.append(torch.empty(self.nbytes, dtype=torch.uint8, pin_memory=True))
        self._free = deque(range(self.depth))
        self._cv = threading.Condition()

    def acquire(self, stop: threading.Event) -> Optional[int]:
        with self._cv:
            while not self._free:
                if stop.is_set():
                    return None
                self._cv.wait(0.05)
            return self._free.popleft()

    def release(self, k: int) -> None:
        with self._cv:
            self._free.append(k)
            self._cv.notify()

    def view(self, k: int, shape: Tuple[int, ...]) -> np.ndarray:
        n = int(np.prod(shape))
        return self._bufs[k].numpy()[:n].reshape(shape)

    def give_back(self) -> None:
        with self._lock:
            have = self._pool.setdefault(self.nbytes, [])
            if len(have) < 2 * self.depth:
                have.extend(self._bufs)
        self._bufs = []


class _ChunkStream:
    """``for start, end, frames in stream`` over the 512-frame chunks of a video (backend/cbas.py:423-425), decoded
    ahead: ``reader.get_batch`` for chunks k+1 and k+2 runs on a background thread while chunk k is being staged and
    encoded.  The reference decodes, converts, copies and computes strictly in turn (cbas.py:425-438); with the ViT ~100x
    faster the decoder (decord's libav call releases the GIL) is what has to be kept busy.  Works with any reader that has
    ``get_batch``; chunks are delivered in order, a decoder error is re-raised at the chunk it belongs to.  Readers that
    already decode ahead on their own thread (``PipeFrameSource``) are read directly.

    ``pinned=True`` (real encoder on a GPU): chunks are delivered as views of page-locked ring buffers; the consumer
    calls ``release(frames)`` when every host->HBM copy that reads a chunk has completed."""

    def __init__(self, reader, video_len: int, pinned: bool = False, depth: int = 2, piece: int = CHUNK_SIZE):
        import queue
        self._reader, self._n = reader, int(video_len)
        # frames per delivery: the reference's 512-frame chunk (backend/cbas.py:48) unless the caller asks for smaller
        # pieces - the encode loops do, so that the GPU starts after 128 frames have been decoded, not 512, and the
        # last piece's tail is short (a clip's fixed latency; a divisor of the chunk so progress / flush points are kept)
        self._piece = int(piece) if 0 < int(piece) <= CHUNK_SIZE and CHUNK_SIZE % int(piece) == 0 else CHUNK_SIZE
        depth = depth * (CHUNK_SIZE // self._piece)
        self._direct = bool(getattr(reader, "decodes_ahead", False)) or os.environ.get("CBAS_DECODE_AHEAD") == "0"
        self._pinned = bool(pinned) and not self._direct and torch.cuda.is_available()
        self._ring: Optional[_PinnedRing] = None
        self._tokens: dict = {}                      # data address of a delivered chunk -> ring buffer index
        self._stop = threading.Event()
        self._t = None
        self._ring_lock = threading.Lock()
        self._readers = [reader]
        self._ts: list = []
        if not self._direct:
            # Readers that can be cloned (decord: a sequential H.264 decoder per instance) decode on several instances at
            # once: instance j takes the 512-frame chunks j, j + k, j + 2k, ... and the consumer takes the chunks in order.
            k = _decode_readers(reader, self._n)
            self._readers += [None] * (k - 1)             # opened by their own threads: the first chunk does not wait for them
            self._q: "queue.Queue" = queue.Queue(maxsize=max(1, depth))
            self._qs = [self._q] + [queue.Queue(maxsize=max(1, depth)) for _ in range(k - 1)]
            # every instance may sit on a full queue plus the piece it is decoding; the consumer holds a few more
            self._ring_depth = (k * (self._q.maxsize + 1) if k > 1 else self._q.maxsize) + 2 + _lib.ENC_SLOTS
            if k == 1:
                self._t = threading.Thread(target=self._run, name="cbas-decode-ahead", daemon=True)
                self._ts = [self._t]
            else:
                self._ts = [threading.Thread(target=self._run_strided, args=(j, k), name=f"cbas-decode-ahead-{j}", daemon=True)
                            for j in range(k)]
                self._t = self._ts[0]
            for t in self._ts:
                t.start()

    # -- producer ---------------------------------------------------------------------------------
    def _put(self, item, q=None) -> bool:
        import queue
        q = self._q if q is None else q
        while not self._stop.is_set():
            try:
                q.put(item, timeout=0.1)
                return True
            except queue.Full:
                continue
        return False

    def _decode(self, i: int, end: int, r=None):
        """One chunk, in a ring buffer when there is one; returns (frames, ring index or None)."""
        r = self._reader if r is None else r
        if not self._pinned:
            return r.get_batch(range(i, end)), None
        shape = getattr(r, "frame_shape", None)
        arr = None
        if shape is None or not hasattr(r, "read_into"):
            arr = r.get_batch(range(i, end))                     # learn the frame shape from the decoder's output
            shape = tuple(arr.shape[1:])
        # Interleaved RGB (decord's layout, `.npy` clips): only channel 1 is consumed (backend/cbas.py:431), so only that
        # plane is staged - a third of the page-locked bytes and of the host -> HBM copy (50 176 bytes
        # per 224 x 224 frame).  CBAS_STAGE_GREEN=0 keeps the whole frame (the device picks the channel through its strides).
        green = len(shape) == 3 and shape[2] == 3 and os.environ.get("CBAS_STAGE_GREEN") != "0" and \
            (arr is None or (arr.dtype == np.uint8 and arr.flags.c_contiguous))
        if green and arr is None and not hasattr(r, "read_channel_into"):
            arr = r.get_batch(range(i, end))
            green = arr.dtype == np.uint8 and arr.flags.c_contiguous
        staged = tuple(shape[:2]) if green else tuple(shape)
        with self._ring_lock:
            if self._ring is None:
                self._ring = _PinnedRing(self._piece * int(np.prod(staged)), depth=self._ring_depth)
        k = self._ring.acquire(self._stop)
        if k is None:
            return None, None
        out = self._ring.view(k, (end - i,) + staged)
        if green:
            if arr is None:
                r.read_channel_into(i, end, 1, out)
            else:
                pick_channel(arr, 1, out)
        elif arr is None:
            r.read_into(i, end, out)
        else:
            np.copyto(out, arr)
        return out, k

    def _run(self):
        for i in range(0, self._n, self._piece):
            end = min(i + self._piece, self._n)
            try:
                frames, k = self._decode(i, end)
                if frames is None:
                    return
                item = (i, end, frames, k)
            except BaseException as e:  # noqa: BLE001 - delivered to the consumer in order
                self._put((i, end, e, None))
                return
            if not self._put(item):
                return

    def _run_strided(self, j: int, k: int):
        """Decoder instance j of k: every k-th 512-frame chunk, piece by piece, into its own queue."""
        q = self._qs[j]
        if self._readers[j] is None:
            try:
                self._readers[j] = self._reader.clone()
            except BaseException as e:  # noqa: BLE001 - this instance's first chunk carries the error
                i = j * CHUNK_SIZE
                self._put((i, min(i + self._piece, self._n), e, None), q)
                return
        r = self._readers[j]
        for c0 in range(j * CHUNK_SIZE, self._n, k * CHUNK_SIZE):
            for i in range(c0, min(c0 + CHUNK_SIZE, self._n), self._piece):
                end = min(i + self._piece, self._n)
                try:
                    frames, kk = self._decode(i, end, r)
                    if frames is None:
                        return
                    item = (i, end, frames, kk)
                except BaseException as e:  # noqa: BLE001 - delivered to the consumer in order
                    self._put((i, end, e, None), q)
                    return
                if not self._put(item, q):
                    return

    # -- consumer ---------------------------------------------------------------------------------
    def __iter__(self):
        for i in range(0, self._n, self._piece):
            if self._direct:
                end = min(i + self._piece, self._n)
                yield i, end, self._reader.get_batch(range(i, end))
                continue
            want = i
            i, end, frames, k = self._qs[(want // CHUNK_SIZE) % len(self._qs)].get()
            assert i == want or isinstance(frames, BaseException), (i, want)
            if isinstance(frames, BaseException):
                raise frames
            if k is not None:
                self._tokens[frames.ctypes.data] = k
            yield i, end, frames

    def release(self, frames: np.ndarray) -> None:
        """The copies out of this chunk are done (no-op for chunks that are not ring buffers)."""
        k = self._tokens.pop(frames.ctypes.data, None)
        if k is not None and self._ring is not None:
            self._ring.release(k)

    def detach_ring(self):
        """Hand the page-locked ring to the caller instead of returning its buffers to the pool on ``close`` (copies out of
        them may still be in flight: ``ClipRunner.submit``); the caller ``give_back()``s it when they are done."""
        ring, self._ring = self._ring, None
        self._tokens.clear()
        return ring

    def close(self):
        """Stop and JOIN the decode-ahead thread (so that the caller may close the reader afterwards)."""
        self._stop.set()
        if self._t is not None:
            for q in self._qs:
                while True:                              # unblock a producer waiting on a full queue
                    try:
                        q.get_nowait()
                    except Exception:  # noqa: BLE001
                        break
            for t in self._ts:
                t.join(timeout=30)
            self._t, self._ts = None, []
        for extra in self._readers[1:]:                  # the decoder instances this stream opened itself
            if extra is not None and hasattr(extra, "close"):
                try:
                    extra.close()
                except Exception:  # noqa: BLE001
                    pass
        self._readers = self._readers[:1]
        if self._ring is not None:
            self._ring.give_back()
            self._ring = None


def _decode_readers(reader, n_frames: int) -> int:
    """How many decoder instances ``_ChunkStream`` runs on one file: 1 unless the reader can be cloned; then
    ``CBAS_DECODE_READERS`` (default 4, at most half the cores), never more than the clip has 512-frame chunks."""
    if not hasattr(reader, "clone"):
        return 1
    try:
        k = int(os.environ.get("CBAS_DECODE_READERS", "4"))
    except ValueError:
        k = 4
    k = min(k, max(1, (os.cpu_count() or 2) // 2), -(-int(n_frames) // CHUNK_SIZE))
    return max(1, k)


PIECE = 128          # frames per decode-ahead delivery on the GPU paths (see _ChunkStream)


def _chunks(reader, video_len: int, pinned: bool = False, piece: int = CHUNK_SIZE) -> _ChunkStream:
    """(start, end, frames) per 512-frame chunk (or ``piece`` frames), decoded ahead unless the reader does that itself."""
    return _ChunkStream(reader, video_len, pinned, piece=piece)


def _progress(progress_callback, start: int, video_len: int) -> None:
    """The reference reports once per 512-frame chunk, before the chunk is computed (backend/cbas.py:427-429)."""
    if progress_callback and start % CHUNK_SIZE == 0:
        progress_callback((min(start + CHUNK_SIZE, video_len) / video_len) * 100)


def _wants_pinned(encoder) -> bool:
    return getattr(encoder, "_h", None) is not None and getattr(encoder, "device", None) is not None \
        and encoder.device.type == "cuda"


def _stream_chunks(encoder: DinoEncoder, reader, video_len: int, progress_callback, w, nslots: int) -> None:
    """The chunk loop of encode_file (cbas.py:423-440) on the asynchronous slots."""
    inflight: deque = deque()        # (slot, chunk frames) in submission order
    free = list(range(nslots))
    left: dict = {}                  # id(chunk frames) -> [frames, sub-batches not yet waited for]

    landed: List[np.ndarray] = []    # rows waiting for the next flush point: the writer gets one append per chunk, as the
                                     # reference's loop does (cbas.py:437-440), not one per 64-frame sub-batch

    def hand_over():
        if landed:
            w.append(landed[0] if len(landed) == 1 else np.concatenate(landed))
            landed.clear()
        w.flush()

    def drain_one():
        slot, fr = inflight.popleft()
        rows, _ = encoder.wait(slot)                 # host wait: this sub-batch's H2D copy is long done
        landed.append(rows)
        free.append(slot)
        ent = left[id(fr)]
        ent[1] -= 1
        if ent[1] == 0:
            del left[id(fr)]
            chunks.release(fr)

    native = _wants_pinned(encoder)
    chunks = _chunks(reader, video_len, pinned=native, piece=PIECE if native else CHUNK_SIZE)
    with contextlib.closing(chunks):                 # the decode-ahead thread is joined before the caller closes the reader
        for i, end_index, frames_np in chunks:       # (n,H,W,3) uint8, host; decoded ahead
            _progress(progress_callback, i, video_len)
            frames_np = np.ascontiguousarray(frames_np)
            nsub = -(-frames_np.shape[0] // encoder.max_batch)
            left[id(frames_np)] = [frames_np, nsub]
            for j in range(0, frames_np.shape[0], encoder.max_batch):
                if not free:
                    drain_one()
                slot = free.pop(0)
                sub = frames_np[j:j + encoder.max_batch]
                encoder.submit_host(slot, sub, channel=1)          # green channel, cbas.py:431
                inflight.append((slot, frames_np))
            # the reference flushes once per 512-frame chunk (cbas.py:440); results of this chunk
            # may still be in flight, so flush what has landed and keep streaming
            if end_index % CHUNK_SIZE == 0 or end_index == video_len:
                while len(inflight) > nslots - 1:
                    drain_one()
                hand_over()
        while inflight:
            drain_one()
        hand_over()


class ClipResult:
    """Rows of one video: ``rows`` (N, D) float16 and ``probs`` (N, C) float32 or None, as numpy arrays in host memory or
    (``on_device``) as torch views of the session's device buffers, valid until that session's next clip; ``pending`` =
    work that still reads those views (e.g. in-flight sends), waited for before the session is reused."""

    def __init__(self, rows, probs, on_device: bool, pending: Optional[list] = None):
        self.rows, self.probs, self.on_device = rows, probs, on_device
        self.pending = pending if pending is not None else []

    @property
    def frames(self) -> int:
        return int(self.rows.shape[0])


class ClipRunner:
    """One video at a time through the fused native session (cbas_fused_*, csrc/api_fused.hip): the chunk loop of
    encode_file (backend/cbas.py:423-440) feeding the window loop of infer_file (:497-551) without the CLS rows leaving
    HBM in between - what the reference does as EncodeThread -> ``_cls.h5`` -> ClassificationThread
    (backend/workthreads.py:316-328, 488-498).  Frames are decoded ahead into page-locked ring buffers and DMA'd from
    there; the results are bit-identical to ``encode_file`` followed by ``infer_file``.  ``head=None``: encode only.

    Stand-in encoders without a native handle (the CPU tests' stubs) go through ``encode_rows`` + ``head.infer_clip``."""

    def __init__(self, encoder, head=None, temperature: float = 1.0, sessions: int = 1):
        self.encoder, self.head, self.temperature = encoder, head, float(temperature)
        self.native = _wants_pinned(encoder)
        self._sessions: list = [None] * max(1, int(sessions))       # [ClipStream, pending list] per slot
        self._next = 0

    def close(self) -> None:
        for ent in self._sessions:
            if ent is not None:
                self._settle(ent[1])
                ent[0].close()
        self._sessions = [None] * len(self._sessions)

    @staticmethod
    def _settle(pending: list) -> None:
        while pending:
            w = pending.pop()
            w.wait() if hasattr(w, "wait") else w()

    def _session(self, capacity: int):
        from .stream import ClipStream
        k = self._next
        self._next = (self._next + 1) % len(self._sessions)
        ent = self._sessions[k]
        if ent is not None:
            self._settle(ent[1])                      # nothing may still read the buffers this clip overwrites
        if ent is None or ent[0]._h is None or ent[0].capacity < capacity or ent[0].temperature != self.temperature:
            cap = max(capacity, 4096)                 # the softmax temperature is fixed when a session is created
            if ent is not None:
                cap = max(cap, 2 * ent[0].capacity if ent[0].capacity < capacity else ent[0].capacity)
                ent[0].close()
            ent = self._sessions[k] = [ClipStream(self.encoder, self.head, cap, self.temperature), []]
        ent[0].reset()
        return ent

    def prepare(self, path: str) -> "_PreparedClip":
        """Open ``path`` and start decoding its first pieces into page-locked buffers NOW, for a ``submit`` that comes
        later: with this called for clip i+1 before clip i is pushed, the next clip's frames are already waiting when
        the current one ends (opening a file and decoding its first piece otherwise leave the GPU idle for ~8 ms between
        clips).  Reader errors surface from ``submit``."""
        if not self.native:
            return _PreparedClip(path, None, 0, None, None)
        try:
            reader = open_video(path)
        except BaseException as e:  # noqa: BLE001 - reported when the clip's turn comes
            return _PreparedClip(path, None, 0, None, e)
        try:
            n = len(reader)
            chunks = _chunks(reader, n, pinned=True, piece=PIECE) if n > 0 else None
        except BaseException as e:  # noqa: BLE001
            if hasattr(reader, "close"):
                reader.close()
            return _PreparedClip(path, None, 0, None, e)
        return _PreparedClip(path, reader, n, chunks, None)

    def submit(self, path: str, reader=None, progress_callback=None, prepared: "Optional[_PreparedClip]" = None) -> "_PendingClip":
        """``run`` split in two for back-to-back clips: every frame of the video is pushed and the tail (last batches,
        tail classification, copy-out into page-locked memory) is QUEUED, not waited for; ``.result()`` of the returned
        object waits and gives the ClipResult (``None`` for a video without frames).  Call it after the NEXT clip has been
        submitted (sessions alternate: needs ``sessions >= 2``) and the GPU never idles between clips.  ``prepared``: what
        ``prepare(path)`` returned (reader open, first pieces decoded)."""
        if prepared is not None and prepared.error is not None:
            raise prepared.error
        if prepared is not None and prepared.reader is not None:
            reader, chunks, own_reader = prepared.reader, prepared.chunks, True
            prepared.reader = prepared.chunks = None
        else:
            chunks = None
            if not self.native or len(self._sessions) < 2:
                return _PendingClip(self.run(path, reader, progress_callback), None)
            own_reader = reader is None
            reader = reader if reader is not None else open_video(path)
        try:
            video_len = len(reader)
            if video_len == 0:
                print(f"Warning: Video {path} contains no frames. Skipping.")
                return _PendingClip(None, None)
            if len(self._sessions) < 2:
                return _PendingClip(self._run_native(reader, video_len, progress_callback, False, chunks=chunks), None)
            return self._run_native(reader, video_len, progress_callback, False, wait=False, chunks=chunks)
        finally:
            if chunks is not None:
                chunks.close()                     # (already closed on the normal path; stops the decoder on an error)
            if own_reader and hasattr(reader, "close"):
                reader.close()

    def run(self, path: str, reader=None, progress_callback=None, device_out: bool = False, rows_sink=None) -> Optional[ClipResult]:
        """Encode (and classify) one video; ``None`` for a video without frames.  Reader / encoder errors propagate with
        the encoder left reusable.  ``rows_sink`` (an object with ``append(rows)``, e.g. ``h5io.ClsWriter``): the fp16 rows are
        handed to it in order WHILE the clip runs, from a helper thread (cbas_fused_stream_rows) - the file is all but
        written when the last batch ends, as with the reference's per-chunk writes (backend/cbas.py:436-440)."""
        if not self.native:
            rows = encode_rows(self.encoder, path, progress_callback, reader)
            if rows is None:
                return None
            probs = None
            if self.head is not None and rows.shape[0] > 0:
                probs = self.head.infer_clip(torch.from_numpy(rows), self.temperature).cpu().numpy()
            if rows_sink is not None:
                rows_sink.append(rows)
            return ClipResult(rows, probs, False)
        own_reader = reader is None
        reader = reader if reader is not None else open_video(path)
        try:
            video_len = len(reader)
            if video_len == 0:
                print(f"Warning: Video {path} contains no frames. Skipping.")
                return None
            return self._run_native(reader, video_len, progress_callback, device_out, rows_sink=rows_sink)
        finally:
            if own_reader and hasattr(reader, "close"):
                reader.close()

    def _run_native(self, reader, video_len: int, progress_callback, device_out: bool, wait: bool = True, chunks=None,
                    rows_sink=None):
        enc = self.encoder
        ent = None
        pump = None
        sub = 0                                      # sub-batches submitted so far
        held: deque = deque()                        # (chunk frames, sub-batch count after which its copies are done)
        if chunks is None:
            chunks = _chunks(reader, video_len, pinned=True, piece=PIECE)
        with contextlib.closing(chunks):
            try:
                for i, _end_index, frames in chunks:
                    _progress(progress_callback, i, video_len)
                    frames = np.ascontiguousarray(frames)
                    if ent is None:
                        enc._fit_frame(frames.shape[1], frames.shape[2])     # may rebuild the handle (closes sessions)
                        ent = self._session(video_len)
                        if rows_sink is not None and wait and not device_out and os.environ.get("CBAS_ROWS_STREAM") != "0":
                            ent[0].stream_rows_to(video_len)
                            pump = _RowsPump(ent[0], rows_sink)
                    if pump is not None and pump.error is not None:          # the file cannot be written: stop encoding for it
                        raise pump.error
                    ent[0].push_host(frames, channel=1)                      # green channel, cbas.py:431
                    sub += -(-frames.shape[0] // enc.max_batch)
                    # a slot's previous host->HBM copy has completed when the slot is submitted to again, i.e. ENC_SLOTS
                    # sub-batches later (cbas_enc_submit_u8_host_dev): only then may the decoder refill this chunk's buffer
                    held.append((frames, sub + _lib.ENC_SLOTS))
                    while held and held[0][1] <= sub:
                        chunks.release(held.popleft()[0])
                sess = ent[0]
                if not wait:
                    # the ring buffers this clip still reads from are released when its last copies are known to be done:
                    # the pending result does that (it is resolved one clip later, with the ring long since given back, so
                    # the buffers go back to the process-wide pool rather than to this stream's ring)
                    rows, probs, ev = sess.finish_host_async()
                    res = ClipResult(rows, probs if self.head is not None else None, False)
                    return _PendingClip(res, ev, chunks.detach_ring())
                if device_out:
                    rows, probs = sess.finish()
                    torch.cuda.current_stream(enc.device).synchronize()      # every copy out of the ring has completed
                    res = ClipResult(rows, probs if self.head is not None else None, True, ent[1])
                else:
                    rows, probs = sess.finish_host()
                    if pump is not None:
                        pump.finish(rows.shape[0])                           # the sink has every row now (its error surfaces here)
                    elif rows_sink is not None:
                        rows_sink.append(rows)
                    res = ClipResult(rows, probs if self.head is not None else None, False)
                while held:
                    chunks.release(held.popleft()[0])
            except BaseException:
                # ANY way out of the clip other than its normal end - a reader / decode error raised by the `chunks`
                # iterator mid-clip, an exception from the progress callback, a failed push or finish - must leave no
                # "cbas-rows-out" thread behind: it would keep polling a session that the next clip resets (a second
                # consumer of cbas_fused_rows_ready) and could still be inside sink.append (libhdf5 is not thread-safe)
                # while the caller removes the .tmp file.  abort() joins the thread; it is idempotent.
                if pump is not None:
                    pump.abort()
                # ... and must retire the batches this session still has in flight: they hold the ENCODER's slots, which the
                # next clip's session (another one when the next call has a different head / temperature) submits to, and
                # their host -> HBM copies read the page-locked ring that `closing(chunks)` is about to give back
                if ent is not None:
                    try:
                        ent[0].reset()
                    except Exception:  # noqa: BLE001 - the original error is the one to report
                        pass
                raise
        return res


class _RowsPump:
    """Follows a session's progressive row copies (``ClipStream.rows_ready``) on a helper thread and appends the rows to a
    sink in order; ``finish(n)`` returns when all n rows are in the sink, ``abort()`` when the thread has stopped."""

    def __init__(self, sess, sink):
        self._sess, self._sink = sess, sink
        self._have, self._total, self._stop, self._err = 0, None, False, None
        self._t = threading.Thread(target=self._run, name="cbas-rows-out", daemon=True)
        self._t.start()

    def _run(self):
        import time
        try:
            while not self._stop:
                r = self._sess.rows_ready(block=True)
                if r > self._have:
                    self._sink.append(self._sess._stream_rows[self._have:r])
                    self._have = r
                elif self._total is not None and self._have >= self._total:
                    return
                else:
                    time.sleep(0.0005)
        except BaseException as e:  # noqa: BLE001 - re-raised by finish()
            self._err = e

    @property
    def error(self):
        return self._err

    def finish(self, total: int) -> None:
        self._total = int(total)
        self._t.join()
        if self._err is not None:
            raise self._err

    def abort(self) -> None:
        self._stop = True
        self._t.join()


class _PreparedClip:
    """What ``ClipRunner.prepare`` returns: an open reader whose first pieces are being decoded (or the error opening it)."""

    def __init__(self, path, reader, n, chunks, error):
        self.path, self.reader, self.frames, self.chunks, self.error = path, reader, n, chunks, error

    def close(self) -> None:
        """Drop a prepared clip that will not be submitted."""
        if self.chunks is not None:
            self.chunks.close()
            self.chunks = None
        if self.reader is not None and hasattr(self.reader, "close"):
            self.reader.close()
        self.reader = None


class _PendingClip:
    """What ``ClipRunner.submit`` returns: ``result()`` waits for the clip's queued tail and hands out its ClipResult."""

    def __init__(self, res: Optional[ClipResult], event, ring=None):
        self._res, self._ev, self._ring = res, event, ring

    @property
    def done(self) -> bool:
        """Nothing is queued behind this result (``result()`` will not wait)."""
        return self._ev is None

    def result(self) -> Optional[ClipResult]:
        if self._ev is not None:
            self._ev.synchronize()                 # tail classified, rows copied out: every copy out of the ring is long done
            self._ev = None
        if self._ring is not None:
            self._ring.give_back()
            self._ring = None
        return self._res


_runner_cache: dict = {}


def _runner_for(encoder, head, temperature: float) -> ClipRunner:
    """One cached runner (device clip buffers) per live (encoder, head) pair."""
    import weakref
    key = (id(encoder), id(head))
    hit = _runner_cache.get(key)
    if hit is None or hit[0]() is not encoder or (head is not None and hit[1]() is not head):
        for k in [k for k, v in _runner_cache.items() if v[0]() is None or k == key]:
            _runner_cache.pop(k)[2].close()
        hit = _runner_cache[key] = (weakref.ref(encoder), weakref.ref(head) if head is not None else None,
                                    ClipRunner(encoder, head, temperature))
    hit[2].temperature = float(temperature)
    return hit[2]


def encode_infer_file(encoder: DinoEncoder, model, path: str, dataset_name: str, behaviors: List[str],
                      temperature: float = 1.0, progress_callback=None, reader=None) -> Tuple[Optional[str], Optional[str]]:
    """``encode_file`` + ``infer_file`` in one pass: returns (``<video>_cls.h5``, ``<video>_<dataset>_outputs.csv``), both
    byte-identical to what the two calls write, (None, None) for a video without frames.  This is what the reference's
    EncodeThread + ClassificationThread pair does for a video while a model is live (backend/workthreads.py:316-328,
    488-498), without writing the rows out and reading them back first.  Raises like ``encode_file`` (nothing is left
    behind but a removed ``.tmp``)."""
    if not isinstance(encoder, DinoEncoder):
        raise TypeError(f"cbas_amd.encode_infer_file needs a cbas_amd.DinoEncoder; got {type(encoder).__name__}")
    head = _as_mi355x_head(model, encoder.device)
    if len(behaviors) != head.out_features:
        raise ValueError(f"{len(behaviors)} behaviour names for {head.out_features} model outputs")
    own_reader = reader is None
    reader = reader if reader is not None else open_video(path)
    try:
        if len(reader) == 0:
            print(f"Warning: Video {path} contains no frames. Skipping.")
            return None, None
        try:
            cls_path, res = _write_cls_while_encoding(path, encoder, lambda sink: _runner_for(encoder, head, temperature).run(
                path, reader, progress_callback, rows_sink=sink))
        except RuntimeError as e:
            twin = range_fallback_for(encoder, e, path)
            if twin is None:
                raise
            cls_path, res = _write_cls_while_encoding(path, twin, lambda sink: _runner_for(twin, head, temperature).run(
                path, reader, progress_callback, rows_sink=sink))
    finally:
        if own_reader and hasattr(reader, "close"):
            reader.close()
    print(f"Successfully encoded {os.path.basename(path)} to {os.path.basename(cls_path)}")
    csv_path = cls_path.replace("_cls.h5", f"_{dataset_name}_outputs.csv")
    write_probs_csv(csv_path, res.probs, list(behaviors))
    return cls_path, csv_path


# ------------------------------------------------------------------------------------------------
def format_probs_csv(probs: np.ndarray, behaviors: List[str]) -> str:
    """The text ``pd.DataFrame(probs, columns=behaviors).to_csv(index=False)`` writes for float32
    probabilities (backend/cbas.py:565): header, no index, shortest float32 repr, '\\n' EOL."""
    probs = np.asarray(probs, dtype=np.float32)
    lines = [",".join(_csv_field(b) for b in behaviors)]
    for row in probs:
        lines.append(",".join(str(v) for v in row))
    return "\n".join(lines) + "\n"


def _csv_field(s: str) -> str:
    s = str(s)
    if any(ch in s for ch in (",", '"', "\n", "\r")):
        return '"' + s.replace('"', '""') + '"'
    return s


def csv_header_line(behaviors: List[str]) -> str:
    return ",".join(_csv_field(b) for b in behaviors) + "\n"


def write_probs_csv(path: str, probs: np.ndarray, behaviors: List[str], threads: int = 1) -> None:
    """``pd.DataFrame(probs, columns=behaviors).to_csv(path, index=False)`` (backend/cbas.py:565), byte for byte, through
    the native formatter of the C ABI (cbas_csv_write_f32: ~1M rows/s per thread against pandas' 35k rows/s)."""
    probs = np.ascontiguousarray(probs, dtype=np.float32)
    if probs.ndim != 2 or probs.shape[1] != len(behaviors):
        raise ValueError(f"{len(behaviors)} column names for an array of shape {probs.shape}")
    lib = _lib.load()
    _lib.check(lib.cbas_csv_write_f32(os.fsencode(path), csv_header_line(behaviors).encode("utf-8"), probs.ctypes.data,
                                      probs.shape[0], probs.shape[1], int(threads)), "cbas_csv_write_f32")


_head_cache = {}       # single entry: id(module) -> (weakref to the module, parameter versions, device head)


def _param_versions(model):
    # torch bumps Tensor._version on every in-place update (optimizer.step, load_state_dict, .copy_): together with
    # the storage address this fingerprints "the same weights as when the device copy was made".  Writes through
    # ``param.data`` use a separate version counter, so a cheap content check rides along: the two scalar parameters
    # and a slice of every tensor (a few hundred numbers; an update that changes none of them is not a realistic one).
    sd = model.state_dict()
    with torch.no_grad():
        probe = tuple(float(p.detach().reshape(-1)[:32].double().sum()) for p in sd.values())
    return tuple((p.data_ptr(), p._version) for p in sd.values()) + probe


def _as_mi355x_head(model, device) -> ClassifierLSTMDeltas:
    """A reference ``classifier_head.ClassifierLSTMDeltas`` is mirrored on the device once and reused while it is the
    same live module with unchanged weights; further training, ``load_state_dict`` or a new module that happens to
    reuse the old one's ``id`` all rebuild the device copy (the head is ~2 MB)."""
    if isinstance(model, ClassifierLSTMDeltas):
        return model.to(device)
    if not (hasattr(model, "state_dict") and hasattr(model, "seq_len")):
        raise TypeError(f"cannot run {type(model).__name__} on the MI355X head")
    import weakref
    key, ver = id(model), _param_versions(model)
    hit = _head_cache.get(key)
    if hit is None or hit[0]() is not model or hit[1] != ver:
        _head_cache.clear()
        _head_cache[key] = (weakref.ref(model), ver, from_reference_module(model, device))
    return _head_cache[key][2].to(device)


INFER_CHUNK = 20000                  # backend/cbas.py:482: frames per read of the `cls` dataset


def classify_cls_file(reader: "h5io.ClsReader", head: ClassifierLSTMDeltas, temperature: float, device,
                      chunk: int = INFER_CHUNK) -> np.ndarray:
    """The read + window loop of infer_file (backend/cbas.py:497-551): the `cls` dataset is read in ``chunk``-frame pieces
    with a +-seq_len//2 halo of real rows (windows replicate a row only at the true ends of the clip, :512-525), the next
    piece being read while the head works on the current one, so a day-long file needs a chunk of host memory, not the
    whole clip.  Returns (N, C) float32 probabilities.  Any numeric dataset is accepted: IEEE half rows are consumed as
    they are, everything else as float32 (the reference's ``.float()``, :507-508)."""
    total, half = reader.shape[0], head.seq_len // 2
    device = torch.device(device)
    probs_dev = torch.empty((total, head.out_features), dtype=torch.float32, device=device)
    spans = []
    for start in range(0, total, chunk):
        end = min(start + chunk, total)
        spans.append((start, end, max(0, start - half), min(total, end + half)))

    def read(k):
        _s, _e, r0, r1 = spans[k]
        a = reader.read(r0, r1)
        t = torch.from_numpy(a)
        return t.pin_memory() if device.type == "cuda" else t

    box: dict = {}

    def read_ahead(k):
        try:
            box[k] = read(k)
        except BaseException as e:  # noqa: BLE001 - re-raised on the caller's thread
            box[k] = e

    cur = read(0)
    for k, (start, end, r0, r1) in enumerate(spans):
        th = None
        if k + 1 < len(spans):
            th = threading.Thread(target=read_ahead, args=(k + 1,), name="cbas-cls-read", daemon=True)
            th.start()
        rows = cur.to(device, non_blocking=True)
        # within this piece the windows clamp to [0, r1 - r0): that is the clip's edge exactly where a halo row is missing
        head.infer_range_into(rows, r1 - r0, start - r0, end - start, probs_dev[r0:], temperature)
        if th is not None:
            th.join()
            cur = box.pop(k + 1)
            if isinstance(cur, BaseException):
                raise cur
    return probs_dev.cpu().numpy()


def infer_file(file_path: str, model, dataset_name: str, behaviors: List[str], seq_len: int, device=None,
               temperature: float = 1.0) -> Optional[str]:
    output_file = file_path.replace("_cls.h5", f"_{dataset_name}_outputs.csv")
    try:
        if device is None:
            device = torch.device("cuda")
        device = torch.device(device)
        head = _as_mi355x_head(model, device)
        if head.seq_len != seq_len:
            raise ValueError(f"seq_len={seq_len} does not match the model's seq_len={head.seq_len}")
        with h5io.ClsReader(file_path) as r:
            total_frames = r.shape[0]
            if total_frames == 0:
                print(f"Warning: HDF5 file {file_path} is empty.")
                return None
            if r.shape[1] != head.in_features:
                raise ValueError(f"'cls' has {r.shape[1]} features, the model expects {head.in_features}")
            probs = classify_cls_file(r, head, temperature, device)
        if len(probs) != total_frames:
            print(f"Warning: Prediction count ({len(probs)}) != Frame count ({total_frames}).")
        if len(behaviors) != probs.shape[1]:
            raise ValueError(f"{len(behaviors)} behaviour names for {probs.shape[1]} model outputs")
        write_probs_csv(output_file, probs, behaviors)
        return output_file
    except Exception as e:  # noqa: BLE001 - the reference swallows everything here (cbas.py:568-572)
        print(f"Error during buffered inference on {file_path}: {e}")
        traceback.print_exc()
        return None

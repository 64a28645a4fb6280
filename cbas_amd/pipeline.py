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

import os
import shutil
import traceback
from collections import deque
from typing import Callable, List, Optional

import numpy as np
import torch

from . import _lib, h5io
from .encoder import DinoEncoder
from .framesource import PipeFrameSource, Y4MFileSource
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
        self._r = decord.VideoReader(path, ctx=decord.cpu(0))   # backend/cbas.py:402

    def __len__(self):
        return len(self._r)

    def get_batch(self, indices) -> np.ndarray:
        return self._r.get_batch(indices).asnumpy()             # backend/cbas.py:425


_READERS = {".npy": NpyFrameSource, ".y4m": Y4MFileSource}


def register_reader(ext: str, factory: Callable[[str], object]) -> None:
    _READERS[ext.lower()] = factory


def open_video(path: str):
    ext = os.path.splitext(path)[1].lower()
    if ext in _READERS:
        return _READERS[ext](path)
    try:
        return _DecordSource(path)
    except ImportError as e:
        if shutil.which("ffmpeg") and shutil.which("ffprobe"):
            return PipeFrameSource(path)            # decoder process + reader thread (cbas_amd/framesource.py)
        raise RuntimeError(f"no frame source for {path!r}: decord is not installed, ffmpeg/ffprobe are not on PATH and "
                           f"no reader is registered for {ext!r}") from e


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


def encode_file(encoder: DinoEncoder, path: str, progress_callback=None, reader=None) -> Optional[str]:
    if not isinstance(encoder, DinoEncoder):
        raise TypeError("cbas_amd.encode_file needs a cbas_amd.DinoEncoder (the MI355X encoder); "
                        f"got {type(encoder).__name__}")
    own_reader = reader is None
    reader = reader if reader is not None else open_video(path)     # reader errors propagate (cbas.py:400-402)
    try:
        return _encode_from_reader(encoder, path, reader, progress_callback)
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
    try:
        attrs = {}
        stamp = _current_stamp()
        if stamp:
            attrs = {"encoder_model_identifier": stamp, "schema_version": SCHEMA_VERSION}
        with h5io.ClsWriter(tmp_file_path, D, attrs) as w_sync:
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


def write_cls_file(video_path: str, rows: np.ndarray) -> str:
    """Write ``<video>_cls.h5`` for already-encoded rows with encode_file's file semantics (backend/cbas.py:410-421,
    442): ``.tmp`` first, stamp attributes when a project is set, atomic rename; the ``.tmp`` is removed on failure."""
    out_file_path = os.path.splitext(video_path)[0] + "_cls.h5"
    tmp_file_path = out_file_path + ".tmp"
    attrs = {}
    stamp = _current_stamp()
    if stamp:
        attrs = {"encoder_model_identifier": stamp, "schema_version": SCHEMA_VERSION}
    try:
        with h5io.ClsWriter(tmp_file_path, rows.shape[1], attrs) as w:
            for i in range(0, rows.shape[0], CHUNK_SIZE):            # same append granularity as the chunk loop
                w.append(rows[i:i + CHUNK_SIZE])
                w.flush()
        os.replace(tmp_file_path, out_file_path)
    except Exception:
        if os.path.exists(tmp_file_path):
            try:
                os.remove(tmp_file_path)
            except OSError:
                pass
        raise
    return out_file_path


class _ChunkPrefetcher:
    """Decode ahead: ``reader.get_batch`` for chunk k+1 (and k+2) runs on a background thread while chunk k is being
    staged and encoded.  The reference decodes, converts, copies and computes strictly in turn (cbas.py:425-438); with
    the ViT ~100x faster the decoder (decord's libav call releases the GIL) is what has to be kept busy.  Works with any
    reader that has ``get_batch``; chunks are delivered in order, a decoder error is re-raised at the chunk it
    belongs to.  Readers that already decode ahead on their own thread (``PipeFrameSource``) are used directly."""

    def __init__(self, reader, video_len: int, depth: int = 2):
        import queue
        import threading
        self._q: "queue.Queue" = queue.Queue(maxsize=max(1, depth))
        self._stop = threading.Event()
        self._reader, self._n = reader, video_len
        self._t = threading.Thread(target=self._run, name="cbas-decode-ahead", daemon=True)
        self._t.start()

    def _put(self, item) -> bool:
        import queue
        while not self._stop.is_set():
            try:
                self._q.put(item, timeout=0.1)
                return True
            except queue.Full:
                continue
        return False

    def _run(self):
        for i in range(0, self._n, CHUNK_SIZE):
            end = min(i + CHUNK_SIZE, self._n)
            try:
                item = (i, end, self._reader.get_batch(range(i, end)))
            except BaseException as e:  # noqa: BLE001 - delivered to the consumer in order
                self._put((i, end, e))
                return
            if not self._put(item):
                return

    def __iter__(self):
        for _ in range(0, self._n, CHUNK_SIZE):
            i, end, frames = self._q.get()
            if isinstance(frames, BaseException):
                raise frames
            yield i, end, frames

    def close(self):
        self._stop.set()
        while True:                                  # unblock a producer waiting on a full queue
            try:
                self._q.get_nowait()
            except Exception:  # noqa: BLE001
                break
        self._t.join(timeout=5)


def _chunks(reader, video_len: int):
    """(start, end, frames) per 512-frame chunk, decoded ahead unless the reader does that itself."""
    if getattr(reader, "decodes_ahead", False) or os.environ.get("CBAS_DECODE_AHEAD") == "0":
        for i in range(0, video_len, CHUNK_SIZE):
            end = min(i + CHUNK_SIZE, video_len)
            yield i, end, reader.get_batch(range(i, end))
        return
    pf = _ChunkPrefetcher(reader, video_len)
    try:
        yield from pf
    finally:
        pf.close()


def _stream_chunks(encoder: DinoEncoder, reader, video_len: int, progress_callback, w, nslots: int) -> None:
    """The chunk loop of encode_file (cbas.py:423-440) on the asynchronous slots."""
    inflight: deque = deque()        # (slot, n_frames) in submission order
    free = list(range(nslots))
    keep: List[np.ndarray] = []      # keep chunk arrays alive until their sub-batches are staged

    def drain_one():
        slot, _n = inflight.popleft()
        rows, _ = encoder.wait(slot)
        w.append(rows)
        free.append(slot)

    for i, end_index, frames_np in _chunks(reader, video_len):     # (n,H,W,3) uint8, host; decoded ahead
        if progress_callback:
            progress_callback((end_index / video_len) * 100)
        frames_np = np.ascontiguousarray(frames_np)
        keep = [frames_np]
        for j in range(0, frames_np.shape[0], encoder.max_batch):
            if not free:
                drain_one()
            slot = free.pop(0)
            sub = frames_np[j:j + encoder.max_batch]
            encoder.submit_host(slot, sub, channel=1)          # green channel, cbas.py:431
            inflight.append((slot, sub.shape[0]))
        # the reference flushes once per 512-frame chunk (cbas.py:440); results of this chunk
        # may still be in flight, so flush what has landed and keep streaming
        while len(inflight) > nslots - 1:
            drain_one()
        w.flush()
    while inflight:
        drain_one()
    w.flush()
    del keep


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


def write_probs_csv(path: str, probs: np.ndarray, behaviors: List[str]) -> None:
    try:
        import pandas as pd
        pd.DataFrame(np.asarray(probs, dtype=np.float32), columns=behaviors).to_csv(path, index=False)
    except ImportError:
        with open(path, "w", newline="") as f:
            f.write(format_probs_csv(probs, behaviors))


_head_cache = {}       # single entry: id(module) -> (weakref to the module, parameter versions, device head)


def _param_versions(model):
    # torch bumps Tensor._version on every in-place update (optimizer.step, load_state_dict, .copy_): together with
    # the storage address this fingerprints "the same weights as when the device copy was made"
    return tuple((p.data_ptr(), p._version) for p in model.state_dict().values())


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
            if r.itemsize != 2:
                raise NotImplementedError("only float16 'cls' datasets (what encode_file writes) are supported")
            if r.shape[1] != head.in_features:
                raise ValueError(f"'cls' has {r.shape[1]} features, the model expects {head.in_features}")
            cls = r.read(0, total_frames)
        cls_dev = torch.from_numpy(cls).to(device)
        probs = head.infer_clip(cls_dev, temperature).cpu().numpy()
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

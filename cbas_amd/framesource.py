"""Frame sources for ``encode_file`` beyond ``decord``.

The reference decodes with ``decord.VideoReader(path).get_batch(range)`` (backend/cbas.py:402,425) and
keeps only the green plane (``frames[:, :, :, 1] / 255``, cbas.py:431).  Two sources that deliver exactly
that plane, one byte per pixel, without any Python video package:

* ``Y4MFileSource``  - an uncompressed YUV4MPEG2 file (mono or planar YUV; the first plane is used),
  memory-mapped, random access.  Registered for ``.y4m``.
* ``PipeFrameSource`` - an external decoder process (by default ``ffmpeg ... -vf extractplanes=g -pix_fmt
  gray -f yuv4mpegpipe -``) streaming Y4M over a pipe.  A reader thread decodes ahead into a bounded
  queue, so decode, the pinned-host -> HBM copy (``cbas_enc_submit_u8_host``) and the ViT overlap.
  The frame count comes from a probe command (``ffprobe -count_packets``) because ``encode_file``
  reports progress and the container's count is what the reference's ``len(reader)`` returns.

Both return uint8 ``(n, H, W)`` green planes from ``get_batch(range)``; ``DinoEncoder.submit_host`` takes
that layout directly (50 KB per 224x224 frame over PCIe instead of 150 KB).
"""
from __future__ import annotations

import ctypes
import os
import queue
import subprocess
import threading
from typing import List, Optional, Sequence

import numpy as np

_Y4M_MAGIC = b"YUV4MPEG2"


def _parse_y4m_header(line: bytes):
    """'YUV4MPEG2 W224 H224 F10:1 Ip A1:1 Cmono' -> (width, height, bytes per frame)."""
    parts = line.strip().split()
    if not parts or parts[0] != _Y4M_MAGIC:
        raise ValueError(f"not a YUV4MPEG2 stream (starts with {line[:16]!r})")
    w = h = None
    cs = b"420"
    for p in parts[1:]:
        if p[:1] == b"W":
            w = int(p[1:])
        elif p[:1] == b"H":
            h = int(p[1:])
        elif p[:1] == b"C":
            cs = p[1:]
    if not w or not h:
        raise ValueError(f"Y4M header lacks W/H: {line!r}")
    if cs.startswith(b"mono"):
        fsize = w * h
    elif cs.startswith(b"420"):
        fsize = w * h + 2 * (((w + 1) // 2) * ((h + 1) // 2))
    elif cs.startswith(b"422"):
        fsize = w * h + 2 * (((w + 1) // 2) * h)
    elif cs.startswith(b"444"):
        fsize = 3 * w * h
    else:
        raise ValueError(f"unsupported Y4M colour space {cs!r}")
    if cs[-2:] in (b"10", b"12", b"16"):
        raise ValueError(f"only 8-bit Y4M is supported (got {cs!r})")
    return w, h, fsize


def write_y4m(path: str, planes: np.ndarray, fps: int = 10) -> None:
    """Write uint8 (N,H,W) planes as a mono Y4M file (test clips; ``ffmpeg -i x.mp4 -vf extractplanes=g
    -pix_fmt gray x.y4m`` produces the same layout from a real video)."""
    planes = np.ascontiguousarray(planes, dtype=np.uint8)
    n, h, w = planes.shape
    with open(path, "wb") as f:
        f.write(b"YUV4MPEG2 W%d H%d F%d:1 Ip A1:1 Cmono\n" % (w, h, fps))
        for i in range(n):
            f.write(b"FRAME\n")
            f.write(planes[i].tobytes())


class Y4MFileSource:
    """Random-access reader of an uncompressed .y4m file; the first plane of every frame is returned."""

    def __init__(self, path: str):
        with open(path, "rb") as f:
            head = f.readline(256)
            self.width, self.height, self._fsize = _parse_y4m_header(head)
            self._data0 = len(head)
            marker = f.readline(256)
        if not marker.startswith(b"FRAME"):
            size = os.path.getsize(path)
            if size == self._data0:
                self._n, self._stride, self._mm = 0, 0, None
                return
            raise ValueError(f"{path}: no FRAME marker after the Y4M header")
        self._marker = len(marker)                   # constant when the frames carry no parameters (ffmpeg's output)
        self._stride = self._marker + self._fsize
        size = os.path.getsize(path) - self._data0
        if size % self._stride:
            raise ValueError(f"{path}: size is not a whole number of {self._stride}-byte frames (per-frame parameters?)")
        self._n = size // self._stride
        self._mm = np.memmap(path, dtype=np.uint8, mode="r", offset=self._data0, shape=(self._n, self._stride))

    def __len__(self):
        return self._n

    def get_batch(self, indices) -> np.ndarray:
        idx = list(indices)
        hw = self.width * self.height
        if not idx:
            return np.empty((0, self.height, self.width), np.uint8)
        if idx == list(range(idx[0], idx[0] + len(idx))):
            rows = self._mm[idx[0]:idx[0] + len(idx), self._marker:self._marker + hw]
        else:
            rows = self._mm[idx][:, self._marker:self._marker + hw]
        return np.ascontiguousarray(rows).reshape(len(idx), self.height, self.width)


class PipeFrameSource:
    """Sequential frames from an external decoder writing Y4M to its stdout."""

    decodes_ahead = True           # its reader thread already runs ahead of the consumer (pipeline._chunks)

    DECODE_CMD: Sequence[str] = ("ffmpeg", "-v", "error", "-i", "{path}", "-vf", "extractplanes=g", "-pix_fmt", "gray",
                                 "-f", "yuv4mpegpipe", "-")
    PROBE_CMD: Sequence[str] = ("ffprobe", "-v", "error", "-select_streams", "v:0", "-count_packets", "-show_entries",
                                "stream=nb_read_packets", "-of", "csv=p=0", "{path}")

    def __init__(self, path: str, decode_cmd: Optional[Sequence[str]] = None, probe_cmd: Optional[Sequence[str]] = None,
                 n_frames: Optional[int] = None, prefetch_frames: int = 512, queue_depth: int = 3):
        self.path = path
        fmt = lambda cmd: [a.replace("{path}", path) for a in cmd]     # noqa: E731
        if n_frames is None:
            out = subprocess.run(fmt(probe_cmd or self.PROBE_CMD), stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
            n_frames = int(out.stdout.decode().strip().split(",")[0] or 0)
        self._n = int(n_frames)
        self._next = 0
        self._prefetch = int(prefetch_frames)
        self._q: "queue.Queue" = queue.Queue(maxsize=queue_depth)
        self._pending: List[np.ndarray] = []
        # the decoder's stderr goes to an unnamed temp file, not a pipe: a pipe nobody drains blocks the decoder
        # (and with it get_batch) once ~64 KB of warnings have accumulated
        import tempfile
        self._errf = tempfile.TemporaryFile()
        self._proc = subprocess.Popen(fmt(decode_cmd or self.DECODE_CMD), stdout=subprocess.PIPE, stderr=self._errf,
                                      bufsize=1 << 20)
        head = self._proc.stdout.readline(256)
        if not head and self._n == 0:
            self.width = self.height = 0
            self._fsize = 0
        else:
            try:
                self.width, self.height, self._fsize = _parse_y4m_header(head)
            except ValueError as e:
                err = self._stderr_tail()
                self.close()
                raise RuntimeError(f"decoder for {path!r} did not produce a Y4M stream: {e}; stderr: {err[-400:]}") from e
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._pump, name="cbas-decode", daemon=True)
        self._thread.start()

    # reader thread: decode ahead of the consumer
    def _pump(self):
        hw = self.width * self.height
        try:
            while not self._stop.is_set():
                buf = np.empty((self._prefetch, self.height, self.width), np.uint8)
                k = 0
                while k < self._prefetch:
                    marker = self._proc.stdout.readline(256)
                    if not marker:
                        break
                    if not marker.startswith(b"FRAME"):
                        raise ValueError(f"Y4M stream out of sync: {marker[:16]!r}")
                    data = self._proc.stdout.read(self._fsize)
                    if len(data) < self._fsize:
                        raise ValueError("Y4M stream truncated inside a frame")
                    buf[k] = np.frombuffer(data, np.uint8, hw).reshape(self.height, self.width)
                    k += 1
                if k:
                    self._put(buf[:k])
                if k < self._prefetch:
                    break
            self._put(None)
        except Exception as e:  # noqa: BLE001   surfaced to the consumer by get_batch
            self._put(e)

    def _stderr_tail(self, n: int = 400) -> str:
        try:
            self._proc.wait(timeout=2)
        except Exception:  # noqa: BLE001
            pass
        try:
            self._errf.seek(0, 2)
            size = self._errf.tell()
            self._errf.seek(max(0, size - n))
            return self._errf.read().decode(errors="replace")
        except Exception:  # noqa: BLE001
            return ""

    def _put(self, item):
        while not self._stop.is_set():
            try:
                self._q.put(item, timeout=0.1)
                return
            except queue.Full:
                continue

    def __len__(self):
        return self._n

    def get_batch(self, indices) -> np.ndarray:
        idx = list(indices)
        if not idx:
            return np.empty((0, self.height, self.width), np.uint8)
        if idx != list(range(self._next, self._next + len(idx))):
            raise ValueError(f"PipeFrameSource is sequential: expected frames from {self._next}, got {idx[0]}..{idx[-1]}")
        need, parts = len(idx), []
        while need > 0:
            if not self._pending:
                while True:                       # never wait forever on a decoder that died without closing its pipe
                    try:
                        item = self._q.get(timeout=1.0)
                        break
                    except queue.Empty:
                        if not self._thread.is_alive() and self._q.empty():
                            raise EOFError(f"{self.path}: decoder thread ended without delivering frame {self._next}; "
                                           f"stderr: {self._stderr_tail()}") from None
                if item is None:
                    raise EOFError(f"{self.path}: decoder delivered {self._next + len(idx) - need} frames, container reports {self._n}")
                if isinstance(item, Exception):
                    raise item
                self._pending.append(item)
            cur = self._pending[0]
            take = min(need, cur.shape[0])
            parts.append(cur[:take])
            if take == cur.shape[0]:
                self._pending.pop(0)
            else:
                self._pending[0] = cur[take:]
            need -= take
        self._next += len(idx)
        return parts[0] if len(parts) == 1 and parts[0].flags.c_contiguous else np.concatenate(parts)

    def close(self):
        if getattr(self, "_stop", None) is not None:
            self._stop.set()
        p = getattr(self, "_proc", None)
        if p is not None:
            try:
                p.kill()
            except OSError:
                pass
            for s in (p.stdout, getattr(self, "_errf", None)):
                try:
                    s.close()
                except Exception:  # noqa: BLE001
                    pass
            p.wait()
            self._proc = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


# ------------------------------------------------------------------------------------------------------------------
# Motion-JPEG in an AVI container: the one COMPRESSED video format this image can decode end to end (Pillow ships
# libjpeg-turbo; decord, ffmpeg, PyAV and OpenCV are all absent), so it is what exercises decode / copy / compute overlap
# with real decoder work in the loop.  CBAS itself records H.264 .mp4 through ffmpeg (backend/cbas.py:732-734) and reads
# it back with decord (:402); for those files the order in pipeline.open_video stays decord -> ffmpeg pipe.
# ------------------------------------------------------------------------------------------------------------------
def _riff_chunks(buf, start: int, end: int):
    """(fourcc, data offset, size) of the chunks in buf[start:end]."""
    import struct
    p = start
    while p + 8 <= end:
        cc, size = buf[p:p + 4], struct.unpack_from("<I", buf, p + 4)[0]
        yield bytes(cc), p + 8, min(size, end - (p + 8))      # a truncated file: never hand out bytes past its end
        p += 8 + size + (size & 1)


class MJPEGAviSource:
    """``decord.VideoReader``-shaped reader (``len``, ``get_batch(indices)``) for Motion-JPEG AVI files.  The file is
    memory-mapped, the frame table comes from walking the ``movi`` list(s) once, and frames are decoded by the library's own
    decoder (``cbas_mjpeg_decode``, include/cbas_mi355x.h: C++ threads, no GIL, pixel-identical to Pillow's libjpeg-turbo)
    straight into the caller's buffer when ``read_into`` is used - which is what the decode-ahead thread of
    ``pipeline._ChunkStream`` does with its page-locked ring buffers.  ``planes=True`` delivers the green plane only,
    (n, H, W) - the one channel the encoder reads (backend/cbas.py:431), a third of the host writes and PCIe bytes;
    ``planes=False`` the (n, H, W, 3) RGB frames decord would.  Streams the native decoder refuses (progressive, CMYK,
    exotic sampling) and builds without the library go through Pillow on a thread pool instead.  Whether ffmpeg's MJPEG
    decoder + swscale, which decord would use, produces the same green plane bit for bit is NOT verified here."""

    def __init__(self, path: str, threads: Optional[int] = None, planes: bool = False, native: Optional[bool] = None):
        import mmap
        self.path = path
        self.planes = bool(planes)
        self._native = native                    # None: try the library, fall back to Pillow; True: library or raise; False: Pillow
        self._table = None
        self._f = open(path, "rb")
        self._mm = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ)
        mm = self._mm
        if len(mm) < 12 or mm[0:4] != b"RIFF" or mm[8:12] != b"AVI ":
            self.close()
            raise ValueError(f"{path}: not a RIFF AVI file")
        self._frames = []                        # (offset, size) of every video chunk, in order
        self.width = self.height = 0
        pos = 0
        while pos + 12 <= len(mm) and mm[pos:pos + 4] == b"RIFF":           # 'AVI ' then OpenDML 'AVIX' segments
            import struct
            rsize = struct.unpack_from("<I", mm, pos + 4)[0]
            rend = min(len(mm), pos + 8 + rsize)
            for cc, off, size in _riff_chunks(mm, pos + 12, rend):
                if cc != b"LIST":
                    continue
                kind = bytes(mm[off:off + 4])
                if kind == b"hdrl":
                    for c2, o2, s2 in _riff_chunks(mm, off + 4, off + size):
                        if c2 == b"avih" and s2 >= 40:
                            self.width, self.height = struct.unpack_from("<II", mm, o2 + 32)
                        elif c2 == b"LIST" and bytes(mm[o2:o2 + 4]) == b"strl":
                            for c3, o3, s3 in _riff_chunks(mm, o2 + 4, o2 + s2):
                                if c3 == b"strh" and bytes(mm[o3:o3 + 4]) == b"vids":
                                    codec = bytes(mm[o3 + 4:o3 + 8]).upper()
                                    if codec not in (b"MJPG", b"JPEG", b"AVRN", b"LJPG"):
                                        self.close()
                                        raise ValueError(f"{path}: video stream is {codec!r}, not Motion-JPEG")
                elif kind == b"movi":
                    self._walk_movi(off + 4, off + size)
            pos = rend + (rsize & 1)
        if self._frames and not (self.width and self.height):
            self.height, self.width = self._decode(0).shape[:2]
        self.frame_shape = (self.height, self.width) if self.planes else (self.height, self.width, 3)
        self._threads = max(1, int(threads)) if threads else max(1, min(16, (os.cpu_count() or 4) - 2))
        self._pool = None

    def _walk_movi(self, start: int, end: int) -> None:
        for cc, off, size in _riff_chunks(self._mm, start, end):
            if cc == b"LIST" and bytes(self._mm[off:off + 4]) == b"rec ":
                self._walk_movi(off + 4, off + size)
            elif cc[2:4] in (b"dc", b"db") and cc[:2].isdigit():
                self._frames.append((off, size))     # a zero-length chunk repeats the previous frame (AVI "dropped frame")

    def __len__(self):
        return len(self._frames)

    def _decode(self, i: int) -> np.ndarray:
        import io
        from PIL import Image
        i = self._source_index(i)
        off, size = self._frames[i]
        with Image.open(io.BytesIO(self._mm[off:off + size])) as im:
            return np.asarray(im.convert("RGB"))

    def _executor(self):
        if self._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=self._threads, thread_name_prefix="cbas-mjpeg")
        return self._pool

    def _source_index(self, i: int) -> int:
        """The chunk that holds frame i's picture: a zero-length chunk (a dropped frame) stands for the frame before it;
        zero-length chunks at the very START of the stream have no frame before them and show the first coded one."""
        j = i
        while j > 0 and self._frames[j][1] == 0:
            j -= 1
        if self._frames[j][1] == 0:                   # leading empty chunks
            j = i
            while j + 1 < len(self._frames) and self._frames[j][1] == 0:
                j += 1
        return j

    def _frame_table(self):
        """(offsets, sizes) as the native decoder takes them, empty chunks resolved as in ``_source_index``."""
        if self._table is None:
            off = np.array([f[0] for f in self._frames], np.uint64)
            size = np.array([f[1] for f in self._frames], np.uint32)
            src = np.arange(len(size))
            src[size == 0] = 0
            src = np.maximum.accumulate(src)          # index of the last non-empty chunk at or before each frame
            coded = np.nonzero(size)[0]
            if len(coded):
                src[:coded[0]] = coded[0]             # leading empty chunks: the first coded frame
            self._table = (np.ascontiguousarray(off[src]), np.ascontiguousarray(size[src]))
        return self._table

    def _native_decode(self, idx: np.ndarray, out: np.ndarray) -> bool:
        """Decode frames ``idx`` into ``out`` with the library; False when it is not available or refuses the stream."""
        if self._native is False:
            return False
        try:
            from . import _lib
            lib = _lib.load()
        except Exception:  # noqa: BLE001 - no library in this environment
            if self._native:
                raise
            self._native = False
            return False
        off, size = self._frame_table()
        off, size = np.ascontiguousarray(off[idx]), np.ascontiguousarray(size[idx])
        base = np.frombuffer(self._mm, np.uint8)
        bad = ctypes.c_int32(-1)
        rc = lib.cbas_mjpeg_decode(base.ctypes.data, off.ctypes.data, size.ctypes.data, len(idx), self.height, self.width,
                                   1 if self.planes else 3, out.ctypes.data, self._threads, ctypes.byref(bad))
        del base
        if rc == 0:
            return True
        why = lib.cbas_last_error().decode(errors="replace")
        if self._native or "unsupported" not in why:
            raise ValueError(f"{self.path}: {why}")
        self._native = False                     # a stream outside the native decoder's envelope: Pillow from here on
        return False

    def read_into(self, start: int, stop: int, out: np.ndarray) -> None:
        if tuple(out.shape[1:]) != self.frame_shape or out.dtype != np.uint8 or not out.flags.c_contiguous:
            raise ValueError(f"{self.path}: read_into needs a C-contiguous uint8 (n,) + {self.frame_shape} buffer")
        if self._native_decode(np.arange(start, stop), out):
            return

        def one(k):
            fr = self._decode(start + k)
            if fr.shape[:2] != out.shape[1:3]:
                raise ValueError(f"{self.path}: frame {start + k} is {fr.shape}, the stream header says {out.shape[1:]}")
            np.copyto(out[k], fr[:, :, 1] if self.planes else fr)
        list(self._executor().map(one, range(stop - start)))

    def get_batch(self, indices) -> np.ndarray:
        idx = list(indices)
        out = np.empty((len(idx),) + self.frame_shape, np.uint8)
        if not idx:
            return out
        idx = [i + len(self._frames) if i < 0 else i for i in idx]
        if idx == list(range(idx[0], idx[0] + len(idx))):
            self.read_into(idx[0], idx[0] + len(idx), out)
        elif not self._native_decode(np.asarray(idx, np.int64), out):
            for k, i in enumerate(idx):
                fr = self._decode(i)
                out[k] = fr[:, :, 1] if self.planes else fr
        return out

    def close(self):
        if getattr(self, "_pool", None) is not None:
            self._pool.shutdown(wait=True)
            self._pool = None
        for name in ("_mm", "_f"):
            o = getattr(self, name, None)
            if o is not None:
                try:
                    o.close()
                except Exception:  # noqa: BLE001
                    pass
                setattr(self, name, None)

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def write_mjpeg_avi(path: str, frames: np.ndarray, fps: int = 10, quality: int = 90, subsampling: Optional[int] = None) -> None:
    """Write uint8 (N, H, W, 3) RGB (or (N, H, W) grey) frames as a Motion-JPEG AVI (one stream, ``idx1`` index): test and
    benchmark clips that any player / ffmpeg / decord opens.  JPEG is lossy: the decoded frames are the reference point,
    not ``frames``.  ``subsampling``: Pillow's 0 / 1 / 2 = 4:4:4 (the default for RGB input) / 4:2:2 / 4:2:0 (what cameras
    and ffmpeg's mjpeg encoder write)."""
    import io
    import struct
    from PIL import Image
    frames = np.asarray(frames, np.uint8)
    n, h, w = frames.shape[:3]
    jpegs = []
    for i in range(n):
        b = io.BytesIO()
        if frames.ndim == 4:
            Image.fromarray(frames[i]).save(b, format="JPEG", quality=quality, subsampling=0 if subsampling is None else subsampling)
        else:
            Image.fromarray(frames[i]).save(b, format="JPEG", quality=quality)
        jpegs.append(b.getvalue())

    def chunk(cc: bytes, data: bytes) -> bytes:
        return cc + struct.pack("<I", len(data)) + data + (b"\0" if len(data) & 1 else b"")

    def lst(kind: bytes, data: bytes) -> bytes:
        return b"LIST" + struct.pack("<I", len(data) + 4) + kind + data

    biggest = max((len(j) for j in jpegs), default=0)
    avih = struct.pack("<14I", 1_000_000 // max(1, fps), biggest * fps, 0, 0x10, n, 0, 1, biggest, w, h, 0, 0, 0, 0)
    strh = b"vids" + b"MJPG" + struct.pack("<IHHIIIIIIII", 0, 0, 0, 0, 1, fps, 0, n, biggest, 0xFFFFFFFF, 0) + struct.pack("<4h", 0, 0, w, h)
    strf = struct.pack("<IiiHH4sIiiII", 40, w, h, 1, 24, b"MJPG", w * h * 3, 0, 0, 0, 0)
    hdrl = lst(b"hdrl", chunk(b"avih", avih) + lst(b"strl", chunk(b"strh", strh) + chunk(b"strf", strf)))
    movi_body, idx, off = b"", b"", 4
    parts = []
    for j in jpegs:
        c = chunk(b"00dc", j)
        idx += b"00dc" + struct.pack("<III", 0x10, off, len(j))
        off += len(c)
        parts.append(c)
    movi_body = b"".join(parts)
    body = b"AVI " + hdrl + lst(b"movi", movi_body) + chunk(b"idx1", idx)
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(body)) + body)

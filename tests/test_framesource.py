"""Frame sources of cbas_amd/framesource.py: Y4M files and a decoder process on a pipe."""
import os
import sys

import numpy as np
import pytest

from cbas_amd import framesource as FS, synth


def planes(n=70, h=32, w=48, seed=3):
    return np.ascontiguousarray(synth.noise_frames(seed, n, h, w)[:, :, :, 1])


def test_y4m_file_round_trip(tmp_path):
    g = planes()
    p = str(tmp_path / "clip.y4m")
    FS.write_y4m(p, g)
    src = FS.Y4MFileSource(p)
    assert len(src) == 70 and (src.height, src.width) == (32, 48)
    assert np.array_equal(src.get_batch(range(0, 70)), g)
    assert np.array_equal(src.get_batch([5, 9, 2]), g[[5, 9, 2]])
    assert src.get_batch(range(0)).shape == (0, 32, 48)


def test_y4m_420_takes_the_first_plane(tmp_path):
    g = planes(5, 16, 16)
    p = str(tmp_path / "c420.y4m")
    with open(p, "wb") as f:
        f.write(b"YUV4MPEG2 W16 H16 F10:1 Ip A1:1 C420jpeg\n")
        for i in range(5):
            f.write(b"FRAME\n" + g[i].tobytes() + bytes(128))
    assert np.array_equal(FS.Y4MFileSource(p).get_batch(range(5)), g)


def test_y4m_rejects_garbage(tmp_path):
    p = str(tmp_path / "bad.y4m")
    open(p, "wb").write(b"RIFF....AVI ")
    with pytest.raises(ValueError):
        FS.Y4MFileSource(p)


def _pipe_source(path, n, **kw):
    cat = [sys.executable, "-c", "import sys,shutil; shutil.copyfileobj(open(sys.argv[1],'rb'), sys.stdout.buffer)", "{path}"]
    probe = [sys.executable, "-c", f"print({n})"]
    return FS.PipeFrameSource(path, decode_cmd=cat, probe_cmd=probe, **kw)


def test_pipe_source_streams_in_order_with_prefetch(tmp_path):
    g = planes(200)
    p = str(tmp_path / "clip.y4m")
    FS.write_y4m(p, g)
    src = _pipe_source(p, 200, prefetch_frames=64, queue_depth=2)
    assert len(src) == 200
    got = [src.get_batch(range(i, min(i + 48, 200))) for i in range(0, 200, 48)]     # batch size unrelated to prefetch size
    assert np.array_equal(np.concatenate(got), g)
    with pytest.raises(ValueError):
        src.get_batch(range(10, 20))          # sequential only
    src.close()


def test_pipe_source_reports_short_streams_and_bad_decoders(tmp_path):
    g = planes(20)
    p = str(tmp_path / "clip.y4m")
    FS.write_y4m(p, g)
    src = _pipe_source(p, 25)                 # container claims more frames than the decoder delivers
    src.get_batch(range(0, 20))
    with pytest.raises(EOFError):
        src.get_batch(range(20, 25))
    src.close()
    with pytest.raises(RuntimeError, match="Y4M"):
        FS.PipeFrameSource(p, decode_cmd=[sys.executable, "-c", "print('not a video')"], n_frames=3)


@pytest.mark.gpu
def test_encode_file_from_y4m_and_pipe_equals_npy(tmp_path):
    """The same clip through the three sources gives byte-identical _cls.h5 rows."""
    import torch
    from cbas_amd import config as C, h5io, pipeline as P, weights as W
    from cbas_amd.encoder import DinoEncoder
    cfg = C.VIT_TINY
    enc = DinoEncoder.from_weights(cfg, W.synth_encoder_weights(cfg, 1234), "cuda", max_batch=16, max_frame=(64, 64))
    rgb = synth.cage_frames(4, 90, 64, 64)
    npy = str(tmp_path / "a.npy"); np.save(npy, rgb)
    y4m = str(tmp_path / "b.y4m"); FS.write_y4m(y4m, rgb[:, :, :, 1])
    y4m2 = str(tmp_path / "c.y4m"); FS.write_y4m(y4m2, rgb[:, :, :, 1])
    def rows(path):
        with h5io.ClsReader(path) as r:
            return r.read(0, r.shape[0])

    ref = rows(P.encode_file(enc, npy))
    got = rows(P.encode_file(enc, y4m))
    got2 = rows(P.encode_file(enc, y4m2, reader=_pipe_source(y4m2, 90, prefetch_frames=32)))
    enc.close()
    assert ref.shape == (90, cfg.hidden_size) and np.array_equal(ref, got) and np.array_equal(ref, got2)


def test_decode_ahead_overlaps_and_preserves_order_and_errors():
    """pipeline._chunks: any reader's get_batch runs ahead on a thread (what keeps decord busy while the GPU works),
    chunks arrive in order with the right contents, and a decoder error surfaces at its chunk."""
    import threading
    import time
    import numpy as np
    from cbas_amd import pipeline as P

    class SlowReader:
        def __init__(self, n, fail_at=None):
            self.n, self.fail_at, self.calls, self.threads = n, fail_at, [], set()

        def __len__(self):
            return self.n

        def get_batch(self, idx):
            idx = list(idx)
            self.calls.append((idx[0], time.perf_counter()))
            self.threads.add(threading.current_thread().name)
            time.sleep(0.05)
            if self.fail_at is not None and idx[0] >= self.fail_at:
                raise IOError(f"decode failed at {idx[0]}")
            return np.full((len(idx), 2, 2, 3), idx[0] // 512, np.uint8)

    r = SlowReader(512 * 5 + 7)
    t0 = time.perf_counter()
    seen = []
    for i, end, fr in P._chunks(r, len(r)):
        time.sleep(0.05)                                   # the "GPU work" of the consumer
        seen.append((i, end, int(fr[0, 0, 0, 0]), fr.shape[0]))
    wall = time.perf_counter() - t0
    assert seen == [(k * 512, min((k + 1) * 512, len(r)), k, min(512, len(r) - k * 512)) for k in range(6)]
    assert r.threads == {"cbas-decode-ahead"}
    assert wall < 0.05 * 12 * 0.8                          # decode and consume overlapped (serial would be 0.6 s)
    bad = SlowReader(512 * 4, fail_at=1024)
    got = []
    with pytest.raises(IOError, match="1024"):
        for i, end, fr in P._chunks(bad, len(bad)):
            got.append(i)
    assert got == [0, 512]


def test_mjpeg_avi_round_trip_and_decode_ahead(tmp_path):
    """A real compressed video through the reader interface encode_file uses: Motion-JPEG AVI written here, decoded by the
    library's own decoder (green planes, what open_video asks for) and by Pillow.  Frame count, shape, random and
    sequential access, read_into == get_batch, and the decode-ahead stream delivers exactly the reader's frames."""
    from cbas_amd import framesource as F, pipeline as P, synth
    fr = synth.cage_frames(2, 70, 48, 64)
    p = str(tmp_path / "clip.avi")
    F.write_mjpeg_avi(p, fr, fps=10, quality=95)
    r = P.open_video(p)                                      # no decord here -> the in-process MJPEG reader, green planes
    assert isinstance(r, F.MJPEGAviSource) and len(r) == 70 and r.frame_shape == (48, 64)
    rgb = F.MJPEGAviSource(p)                                # decord-shaped: RGB frames
    assert rgb.frame_shape == (48, 64, 3)
    allf = rgb.get_batch(range(70))
    assert allf.shape == (70, 48, 64, 3) and allf.dtype == np.uint8
    assert np.abs(allf.astype(int) - fr.astype(int)).mean() < 6.0          # lossy, but the same pictures
    assert np.array_equal(rgb.get_batch([3, 69, 0]), allf[[3, 69, 0]]) and np.array_equal(rgb.get_batch([-1]), allf[69:])
    planes = r.get_batch(range(70))
    assert planes.shape == (70, 48, 64) and np.array_equal(planes, allf[..., 1])     # channel 1 (backend/cbas.py:431)
    out = np.empty((20, 48, 64), np.uint8)
    r.read_into(10, 30, out)
    assert np.array_equal(out, planes[10:30])
    with pytest.raises(ValueError, match="read_into needs"):
        r.read_into(0, 4, np.empty((4, 48, 64, 3), np.uint8))
    got = [f.copy() for _i, _e, f in P._chunks(r, len(r), piece=32)]
    assert [g.shape[0] for g in got] == [32, 32, 6] and np.array_equal(np.concatenate(got), planes)
    r.close()
    rgb.close()
    # decoding is deterministic and the two decoders agree bit for bit: same bytes -> same pixels, whichever ran
    for kw in ({"threads": 1}, {"native": False}, {"native": True, "threads": 3}):
        r2 = F.MJPEGAviSource(p, **kw)
        assert np.array_equal(r2.get_batch(range(70)), allf), kw
        r2.close()
    # a file cut in the middle of a frame (a recording that was interrupted): the frames before the cut decode, the cut
    # one is a decode error - from both decoders - and nothing reads past the end of the mapping
    whole = open(p, "rb").read()
    src = F.MJPEGAviSource(p)
    off40, size40 = src._frames[40]
    src.close()
    cut = str(tmp_path / "cut.avi")
    open(cut, "wb").write(whole[:off40 + size40 // 2])
    for kw in ({"native": True}, {"native": False}):
        r3 = F.MJPEGAviSource(cut, planes=True, **kw)
        assert len(r3) == 41
        assert np.array_equal(r3.get_batch(range(40)), allf[:40, :, :, 1])
        with pytest.raises(Exception):
            r3.get_batch(range(36, 41))
        r3.close()
    # not Motion-JPEG -> refused (open_video then goes on to the ffmpeg pipe, or reports that nothing can read it)
    raw = bytearray(open(p, "rb").read())
    i = raw.find(b"vidsMJPG")
    raw[i + 4:i + 8] = b"H264"
    bad = str(tmp_path / "h264.avi")
    open(bad, "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="not Motion-JPEG"):
        F.MJPEGAviSource(bad)
    with pytest.raises(RuntimeError, match="no frame source"):
        P.open_video(bad)


def test_decord_vs_ffmpeg_green_plane_cross_check(tmp_path, monkeypatch, capsys):
    """When both decoders exist the first frames are compared once and a mismatch is reported; with fakes,
    since neither decoder exists in this image."""
    import shutil as _sh
    from cbas_amd import pipeline as P
    frames = np.random.default_rng(0).integers(0, 256, (20, 8, 8, 3), dtype=np.uint8)

    class FakeDecord:
        def __init__(self, path):
            pass

        def __len__(self):
            return 20

        def get_batch(self, idx):
            return frames[list(idx)]

    class FakePipe:
        delta = 0

        def __init__(self, path, prefetch_frames=512, n_frames=None):
            pass

        def get_batch(self, idx):
            return (frames[list(idx), :, :, 1].astype(np.int16) + self.delta).clip(0, 255).astype(np.uint8)

        def close(self):
            pass

    monkeypatch.setattr(P, "_DecordSource", FakeDecord)
    monkeypatch.setattr(P, "PipeFrameSource", FakePipe)
    monkeypatch.setattr(_sh, "which", lambda name: "/usr/bin/" + name)
    monkeypatch.setattr(P, "_source_check_done", False)
    assert isinstance(P.open_video(str(tmp_path / "a.mp4")), FakeDecord)
    assert "== decord green plane" in capsys.readouterr().out
    P.open_video(str(tmp_path / "b.mp4"))
    assert capsys.readouterr().out == ""                      # once per process
    monkeypatch.setattr(P, "_source_check_done", False)
    FakePipe.delta = 1
    P.open_video(str(tmp_path / "c.mp4"))
    assert "WARNING" in capsys.readouterr().out and "differs from" not in capsys.readouterr().out


class _SlowSeekReader:
    """A decord-like reader whose decode costs time (GIL released, as libav's does) and which, like an H.264 decoder, pays for a
    jump: frames from the previous keyframe up to the first requested one are decoded too."""
    GOP, PER_FRAME = 100, 0.0004
    opened = 0

    def __init__(self, frames):
        self._a, self._pos = frames, 0
        type(self).opened += 1
        self.closed = False

    def clone(self):
        return type(self)(self._a)

    def __len__(self):
        return self._a.shape[0]

    def get_batch(self, indices):
        import time
        idx = list(indices)
        extra = 0 if idx[0] == self._pos else idx[0] % self.GOP          # re-decode from the keyframe after a seek
        time.sleep((len(idx) + extra) * self.PER_FRAME)
        self._pos = idx[-1] + 1
        return self._a[idx[0]:idx[-1] + 1].copy()

    def close(self):
        self.closed = True


def test_cloneable_readers_decode_on_several_instances(monkeypatch):
    """A reader with ``clone()`` (decord's wrapper) is decoded by several instances, each on every k-th 512-frame chunk: same
    frames in the same order, the instances the stream opened are closed with it, errors surface at their chunk, and the
    wall time drops although every jump costs part of a GOP."""
    import time
    from cbas_amd import pipeline as P
    n = 512 * 6 + 77
    fr = (np.arange(n, dtype=np.uint32)[:, None, None, None] * np.array([1, 3, 7], np.uint32)).astype(np.uint8) * np.ones((1, 4, 4, 1), np.uint8)

    def run(k):
        monkeypatch.setenv("CBAS_DECODE_READERS", str(k))
        monkeypatch.setattr(P.os, "cpu_count", lambda: 64)
        _SlowSeekReader.opened = 0
        r = _SlowSeekReader(fr)
        t0 = time.perf_counter()
        cs = P._chunks(r, n, piece=128)
        got = [(i, e, f.copy()) for i, e, f in cs]
        dt = time.perf_counter() - t0
        extra = [x for x in cs._readers[1:]]
        cs.close()
        assert all(x is not None and x.closed for x in extra) and not r.closed      # the caller's reader is the caller's
        assert [g[0] for g in got] == list(range(0, n, 128)) and got[-1][1] == n
        assert np.array_equal(np.concatenate([g[2] for g in got]), fr)
        return dt, _SlowSeekReader.opened

    t1, o1 = run(1)
    t4, o4 = run(4)
    assert o1 == 1 and o4 == 4
    assert t4 < 0.5 * t1, (t1, t4)                                          # 3 149 frames at 0.4 ms: 1.26 s on one instance
    # more instances than chunks: capped
    monkeypatch.setenv("CBAS_DECODE_READERS", "4")
    _SlowSeekReader.opened = 0
    small = _SlowSeekReader(fr[:600])
    cs = P._chunks(small, 600, piece=128)
    assert len(cs._readers) == 2 and np.array_equal(np.concatenate([f for _i, _e, f in cs]), fr[:600])
    cs.close()

    class Failing(_SlowSeekReader):
        def get_batch(self, indices):
            if 1100 in list(indices):
                raise IOError("decoder lost sync")
            return super().get_batch(indices)
    cs = P._chunks(Failing(fr), n, piece=128)
    seen = []
    with pytest.raises(IOError, match="lost sync"):
        for i, _e, _f in cs:
            seen.append(i)
    assert seen == list(range(0, 1024, 128))                                # everything before the bad piece arrived, in order
    cs.close()
    # stopping early joins every instance
    cs = P._chunks(_SlowSeekReader(fr), n, piece=128)
    next(iter(cs))
    cs.close()
    assert cs._ts == []


# ---- green-only staging (r4) ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape,channel", [((3, 7, 5, 3), 1), ((2, 224, 224, 3), 1), ((1, 1, 1, 3), 2), ((5, 33, 17, 4), 0),
                                           ((40, 224, 224, 3), 1), ((2, 9, 9, 1), 0)])
def test_pick_channel_native_equals_numpy(shape, channel):
    """cbas_pick_channel_u8 (SSSE3 shuffle for 3 channels, plain loop otherwise, threads over pixel ranges) == numpy's
    frames[..., channel] (backend/cbas.py:431 keeps channel 1), for sizes off every vector / thread boundary."""
    from cbas_amd import pipeline as P
    rng = np.random.default_rng(sum(shape))
    fr = rng.integers(0, 256, shape, dtype=np.uint8)
    for threads in (1, 3, 16):
        out = np.full(shape[:3], 7, np.uint8)
        P.pick_channel(fr, channel, out, threads=threads)
        assert np.array_equal(out, fr[..., channel]), (shape, threads)


def test_npy_source_reads_the_green_plane_directly(tmp_path):
    from cbas_amd import pipeline as P
    fr = np.random.default_rng(1).integers(0, 256, (70, 12, 20, 3), dtype=np.uint8)
    p = str(tmp_path / "c.npy")
    np.save(p, fr)
    r = P.NpyFrameSource(p)
    out = np.empty((30, 12, 20), np.uint8)
    r.read_channel_into(25, 55, 1, out)
    assert np.array_equal(out, fr[25:55, :, :, 1])

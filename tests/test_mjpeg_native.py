"""The library's Motion-JPEG decoder (cbas_mjpeg_decode, csrc/host_mjpeg.cpp: the frame source's decode stage, SURVEY
§8(f)1) against Pillow's libjpeg-turbo - the decoder every Python user of these files already has: pixel-identical green
planes and RGB frames for every coding the envelope names, clean refusal of everything outside it, clean errors on damaged
streams.  Host memory only: runs without a GPU."""
import ctypes as C
import io

import numpy as np
import pytest

PIL = pytest.importorskip("PIL")
from PIL import Image  # noqa: E402

from cbas_amd import _lib  # noqa: E402


@pytest.fixture(scope="module")
def lib():
    return _lib.load()


def _decode(lib, blobs, h, w, ch, threads=1):
    data = np.frombuffer(b"".join(blobs), np.uint8)
    sizes = np.array([len(b) for b in blobs], np.uint32)
    offs = np.concatenate([[0], np.cumsum(sizes[:-1], dtype=np.uint64)]).astype(np.uint64)
    out = np.zeros((len(blobs), h, w) + ((3,) if ch == 3 else ()), np.uint8)
    bad = C.c_int32(-7)
    rc = lib.cbas_mjpeg_decode(data.ctypes.data, offs.ctypes.data, sizes.ctypes.data, len(blobs), h, w, ch, out.ctypes.data,
                               threads, C.byref(bad))
    return rc, out, bad.value, lib.cbas_last_error().decode()


def _picture(rng, h, w, kind):
    if kind == "noise":
        return rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    base = rng.integers(0, 256, (max(2, h // 8), max(2, w // 8), 3), dtype=np.uint8)
    a = np.asarray(Image.fromarray(base).resize((w, h), Image.BICUBIC)).astype(int)
    return np.clip(a + rng.integers(-5, 6, a.shape), 0, 255).astype(np.uint8)


def _jpeg(a, **kw):
    b = io.BytesIO()
    Image.fromarray(a).save(b, "JPEG", **kw)
    return b.getvalue()


def _pillow(blob):
    return np.asarray(Image.open(io.BytesIO(blob)).convert("RGB"))


@pytest.mark.parametrize("hw", [(224, 224), (256, 256), (8, 8), (1, 1), (17, 23), (33, 47), (7, 250), (240, 320)])
def test_pixel_identical_to_pillow(lib, hw):
    """4:4:4 / 4:2:2 / 4:2:0 / grey, quality 30-100, smooth and noise pictures, sizes that are not multiples of the MCU."""
    h, w = hw
    rng = np.random.default_rng(h * 1000 + w)
    for kind in ("smooth", "noise"):
        a = _picture(rng, h, w, kind)
        for q in (30, 75, 90, 100):
            blobs = [_jpeg(a, quality=q, subsampling=ss) for ss in (0, 1, 2)] + [_jpeg(a[:, :, 1], quality=q)]
            ref = np.stack([_pillow(b) for b in blobs])
            rc, rgb, bad, why = _decode(lib, blobs, h, w, 3)
            assert rc == 0, why
            assert np.array_equal(rgb, ref), (kind, q, np.argwhere((rgb != ref).any(axis=(1, 2, 3))).ravel())
            rc, g, bad, why = _decode(lib, blobs, h, w, 1, threads=3)
            assert rc == 0 and bad == -1, why
            assert np.array_equal(g, ref[..., 1])


def test_restart_intervals_optimised_tables_and_missing_tables(lib):
    rng = np.random.default_rng(5)
    a = _picture(rng, 120, 200, "smooth")
    blobs = [_jpeg(a, quality=85, subsampling=ss, **kw) for ss in (0, 1, 2)
             for kw in ({"restart_marker_rows": 1}, {"restart_marker_blocks": 1}, {"restart_marker_blocks": 7}, {"optimize": True})]
    ref = np.stack([_pillow(b) for b in blobs])
    rc, rgb, _bad, why = _decode(lib, blobs, 120, 200, 3, threads=4)
    assert rc == 0, why
    assert np.array_equal(rgb, ref)

    # "AVI1" Motion-JPEG streams leave the Huffman tables out: the decoder then uses ITU T.81 Annex K's
    def strip_dht(b):
        out, p = bytearray(b[:2]), 2
        while b[p + 1] != 0xDA:
            ln = (b[p + 2] << 8) | b[p + 3]
            if b[p + 1] != 0xC4:
                out += b[p:p + 2 + ln]
            p += 2 + ln
        return bytes(out + b[p:])
    full = [_jpeg(a, quality=q, subsampling=ss) for q in (60, 92) for ss in (0, 2)]
    bare = [strip_dht(b) for b in full]
    assert all(len(x) < len(y) - 400 for x, y in zip(bare, full))
    rc, g, _bad, why = _decode(lib, bare, 120, 200, 1)
    assert rc == 0, why
    assert np.array_equal(g, np.stack([_pillow(b)[..., 1] for b in full]))


def test_refusals_and_damaged_streams(lib):
    rng = np.random.default_rng(6)
    a = _picture(rng, 64, 64, "smooth")
    good = _jpeg(a, quality=90)
    # outside the envelope -> "unsupported" (the reader then hands the file to Pillow)
    rc, _o, bad, why = _decode(lib, [good, _jpeg(a, quality=90, progressive=True)], 64, 64, 1)
    assert rc == _lib.CBAS_EINVAL if hasattr(_lib, "CBAS_EINVAL") else rc == -1
    assert bad == 1 and "unsupported" in why
    cmyk = io.BytesIO()
    Image.fromarray(a).convert("CMYK").save(cmyk, "JPEG")
    rc, _o, bad, why = _decode(lib, [cmyk.getvalue()], 64, 64, 1)
    assert rc != 0 and bad == 0 and "unsupported" in why
    # the wrong size, a truncated frame, noise: errors, not crashes, and never "unsupported"
    rc, _o, bad, why = _decode(lib, [good], 48, 64, 1)
    assert rc != 0 and "64x64" in why
    for cut in (1, 3, 30, 200, len(good) // 2, len(good) - 4):
        rc, _o, bad, why = _decode(lib, [good, good[:cut]], 64, 64, 3, threads=2)
        assert rc != 0 and bad == 1 and "unsupported" not in why, (cut, why)
    for seed in range(20):
        junk = bytearray(np.random.default_rng(seed).integers(0, 256, 3000, dtype=np.uint8).tobytes())
        junk[:2] = b"\xff\xd8"
        rc, _o, bad, _why = _decode(lib, [bytes(junk)], 64, 64, 1)
        assert rc != 0 and bad == 0
    # a good header with damaged entropy data decodes to *something* or errors; it must not read out of bounds or hang
    sos = good.index(b"\xff\xda")
    for seed in range(20):
        r = np.random.default_rng(100 + seed)
        b = bytearray(good)
        for pos in r.integers(sos + 14, len(b) - 2, 12):
            b[pos] = int(r.integers(0, 256))
        _decode(lib, [bytes(b)], 64, 64, 3)
    # argument checks
    rc, *_ = _decode(lib, [good], 64, 64, 2)
    assert rc != 0
    assert lib.cbas_mjpeg_decode(None, None, None, 0, 64, 64, 1, None, 1, None) == 0


def test_threads_write_disjoint_frames(lib):
    rng = np.random.default_rng(9)
    pics = [_picture(rng, 96, 96, "smooth") for _ in range(37)]
    blobs = [_jpeg(p, quality=int(rng.integers(50, 96)), subsampling=int(rng.integers(0, 3))) for p in pics]
    ref = np.stack([_pillow(b)[..., 1] for b in blobs])
    for t in (1, 2, 5, 64):
        rc, g, _bad, why = _decode(lib, blobs, 96, 96, 1, threads=t)
        assert rc == 0, why
        assert np.array_equal(g, ref), t


def _to_16bit_dqt(blob: bytes) -> bytes:
    """Rewrite every DQT segment of a baseline JPEG with 16-bit entries (Pq = 1): the same table values, legal JPEG."""
    out, p = bytearray(blob[:2]), 2
    while p < len(blob):
        assert blob[p] == 0xFF
        m = blob[p + 1]
        if m == 0xDA:                                   # SOS: entropy data follows, copy the rest
            out += blob[p:]
            break
        ln = (blob[p + 2] << 8) | blob[p + 3]
        seg = blob[p + 4:p + 2 + ln]
        if m == 0xDB:
            body, s = bytearray(), 0
            while s < len(seg):
                pq, tq = seg[s] >> 4, seg[s] & 15
                assert pq == 0
                body.append(0x10 | tq)
                for v in seg[s + 1:s + 65]:
                    body += bytes([0, v])
                s += 65
            out += bytes([0xFF, 0xDB, (len(body) + 2) >> 8, (len(body) + 2) & 255]) + body
        else:
            out += blob[p:p + 2 + ln]
        p += 2 + ln
    return bytes(out)


def test_16bit_quantisation_tables_go_to_the_fallback(lib, tmp_path):
    """DQT tables with Pq = 1 (entries up to 65 535) would overflow the decoder's int32 arithmetic; they are
    refused as "unsupported", so the reader hands such a stream to Pillow - and the frames still come out right."""
    from cbas_amd import framesource as FS
    rng = np.random.default_rng(12)
    a = _picture(rng, 64, 64, "smooth")
    good = _jpeg(a, quality=90)
    wide = _to_16bit_dqt(good)
    assert np.array_equal(_pillow(wide), _pillow(good))                  # still the same picture for libjpeg
    rc, _o, bad, why = _decode(lib, [good, wide], 64, 64, 3)
    assert rc != 0 and bad == 1 and "unsupported: 16-bit quantisation table" in why


def test_avi_with_leading_and_inner_empty_chunks(tmp_path):
    """A zero-length '00dc' chunk is a dropped frame: it shows the frame before it; at the very start of the stream it
    shows the first coded frame (it used to map to itself and be reported as a corrupt stream)."""
    from cbas_amd import framesource as FS
    rng = np.random.default_rng(13)
    frames = np.stack([_picture(rng, 48, 64, "smooth") for _ in range(5)])
    p = str(tmp_path / "c.avi")
    FS.write_mjpeg_avi(p, frames, quality=92)
    src = FS.MJPEGAviSource(p)
    want = src.get_batch(range(5)).copy()
    # drop frames 0, 1 and 3 by zeroing their sizes in the reader's table (what an index with empty chunks parses to)
    for k in (0, 1, 3):
        src._frames[k] = (src._frames[k][0], 0)
    src._table = None
    got = src.get_batch(range(5))
    assert np.array_equal(got[0], want[2]) and np.array_equal(got[1], want[2]) and np.array_equal(got[2], want[2])
    assert np.array_equal(got[3], want[2]) and np.array_equal(got[4], want[4])
    src._native = False                                                  # the Pillow path resolves them the same way
    got2 = src.get_batch(range(5))
    assert np.array_equal(got2, got)
    src.close() if hasattr(src, "close") else None

"""Synthetic inputs (frames and CLS sequences) shared by the golden-vector script, the tests and
``bench.py``.  Everything derives from the counter-based hash in ``weights.py`` so the same bytes
are produced on any machine; there is no dataset access in the build environment.
"""
from __future__ import annotations

import numpy as np

from .weights import _hash_stream, synth_normal


def noise_frames(seed: int, n: int, height: int, width: int, first: int = 0) -> np.ndarray:
    """Uniform uint8 RGB noise, (n, H, W, 3); frame ``first + i`` is independent of ``n``."""
    per = height * width * 3
    words = (per + 7) // 8
    out = np.empty((n, per), np.uint8)
    for i in range(n):
        h = _hash_stream(seed, f"noise_frame_{first + i}", words)
        out[i] = h.view(np.uint8)[:per]
    return out.reshape(n, height, width, 3)


_SCENES = ((17.0, 23.0, 5.0, 96.0, 36.0), (9.0, 31.0, 7.0, 128.0, 22.0), (29.0, 11.0, 3.0, 72.0, 44.0),
           (13.0, 13.0, 11.0, 150.0, 30.0), (41.0, 19.0, 4.0, 110.0, 50.0))
_SCENE_LEN, _SCENE_FADE = 22, 5


def cage_frames(seed: int, n: int, height: int, width: int, first: int = 0) -> np.ndarray:
    """Structured clip: a textured background that switches between a few "scenes" every
    ``_SCENE_LEN`` frames (cross-faded over ``_SCENE_FADE``), a bright blob that wanders (and
    sometimes rests) and per-pixel sensor noise.  Gives CLS embeddings with temporal structure so
    the head's delta streams and its argmax labels vary over the clip, with clean transitions.
    (n, H, W, 3) uint8; channel 1 is the one the reference consumes (backend/cbas.py:431),
    channels 0/2 carry decoys."""
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float64)
    bgs = [base + amp * np.sin(xx / fx) + 0.8 * amp * np.cos(yy / fy) + 12.0 * np.sin((xx + yy) / fd)
           for fx, fy, fd, base, amp in _SCENES]
    order = (_hash_stream(seed, "cage_scene_order", 4096) % np.uint64(len(_SCENES))).astype(np.int64)
    out = np.empty((n, height, width, 3), np.uint8)
    for i in range(n):
        f = first + i
        k, r = divmod(f, _SCENE_LEN)
        cur, nxt = int(order[k % 4096]), int(order[(k + 1) % 4096])
        fade = min(1.0, max(0.0, (r - (_SCENE_LEN - _SCENE_FADE)) / float(_SCENE_FADE)))
        fade = fade * fade * (3.0 - 2.0 * fade)
        bg = (1.0 - fade) * bgs[cur] + fade * bgs[nxt]
        phase = f / 37.0
        rest = 0.5 * (1.0 + np.tanh(4.0 * np.sin(f / 53.0)))          # 0 = resting, 1 = moving
        cx = width * (0.5 + 0.35 * np.sin(2.1 * phase * rest + 0.3 * np.sin(f / 11.0)))
        cy = height * (0.5 + 0.35 * np.cos(1.3 * phase * rest + 0.2 * np.cos(f / 7.0)))
        sig = 0.09 * min(height, width) * (1.0 + 0.3 * np.sin(f / 19.0))
        blob = 90.0 * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2.0 * sig * sig))
        nz = _hash_stream(seed, f"cage_noise_{f}", (height * width + 7) // 8).view(np.uint8)[:height * width]
        g = bg + blob + (nz.reshape(height, width).astype(np.float64) - 127.5) * (18.0 / 127.5)
        g8 = np.clip(np.rint(g), 0, 255).astype(np.uint8)
        out[i, :, :, 1] = g8
        out[i, :, :, 0] = np.roll(g8, 7, axis=1)
        out[i, :, :, 2] = 255 - g8
    return out


def cls_walk(seed: int, n: int, dim: int) -> np.ndarray:
    """A float16 CLS-like sequence (n, dim) with temporal structure: slow random walk + jitter."""
    steps = synth_normal(seed, "cls_walk_steps", (n, dim), 0.15).astype(np.float64)
    base = synth_normal(seed, "cls_walk_base", (dim,), 1.0).astype(np.float64)
    jitter = synth_normal(seed, "cls_walk_jitter", (n, dim), 0.05).astype(np.float64)
    # leaky walk keeps the scale bounded for long clips
    walk = np.empty((n, dim), np.float64)
    acc = np.zeros(dim, np.float64)
    for i in range(n):
        acc = 0.97 * acc + steps[i]
        walk[i] = acc
    return (base + walk + jitter).astype(np.float16)


def train_windows(seed: int, n: int, dim: int, n_classes: int, seq_len: int = 31):
    """Labelled training windows (n, seq_len, dim) float32 + labels (n,) int64: windows of a CLS-like
    walk with a class-dependent offset, so a head can learn them (training goldens, tests, bench)."""
    rng = np.random.default_rng(seed)
    seq = cls_walk(seed, n + seq_len - 1, dim).astype(np.float32)
    x = np.stack([seq[i:i + seq_len] for i in range(n)])
    y = rng.integers(0, n_classes, n).astype(np.int64)
    proto = synth_normal(seed, "train_proto", (n_classes, dim), 0.5).astype(np.float32)
    return (x + proto[y][:, None, :]).astype(np.float32), y

"""Host decode rate of the Motion-JPEG frame source: cbas_mjpeg_decode on 1..N threads against Pillow (one thread and a
thread pool), on camera-like 224x224 pictures coded 4:2:0 at quality 85.  Writes one JSON object.

    python scripts/mjpeg_decode_rate.py [out.json]
"""
import io
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbas_amd import _lib  # noqa: E402

lib = _lib.load()
rng = np.random.default_rng(0)
H = W = 224
blobs = []
for i in range(64):
    base = rng.integers(0, 256, (14, 14), dtype=np.uint8)
    g = np.asarray(Image.fromarray(base).resize((W, H), Image.BICUBIC)).astype(int) * 200 // 255 + 28
    g = np.clip(g + rng.normal(0, 4, g.shape), 0, 255).astype(np.uint8)
    b = io.BytesIO()
    Image.fromarray(np.repeat(g[..., None], 3, axis=2)).save(b, "JPEG", quality=85, subsampling=2)
    blobs.append(b.getvalue())
n = 4096
seq = [blobs[i % len(blobs)] for i in range(n)]
data = np.frombuffer(b"".join(seq), np.uint8)
sizes = np.array([len(b) for b in seq], np.uint32)
offs = np.concatenate([[0], np.cumsum(sizes[:-1], dtype=np.uint64)]).astype(np.uint64)
out = np.zeros((n, H, W), np.uint8)
res = {"frame": f"{H}x{W} 4:2:0 q85", "bytes_per_frame": float(sizes.mean()), "cpu_count": os.cpu_count(), "native_fps": {}, "pillow_fps": {}}


def native(t):
    rc = lib.cbas_mjpeg_decode(data.ctypes.data, offs.ctypes.data, sizes.ctypes.data, n, H, W, 1, out.ctypes.data, t, None)
    assert rc == 0


native(4)
ref = np.asarray(Image.open(io.BytesIO(seq[5])).convert("RGB"))[..., 1]
assert np.array_equal(out[5], ref)
for t in (1, 2, 4, 8, 12, 16, 24, 32):
    if t > 2 * (os.cpu_count() or 1):
        break
    best = 0.0
    for _ in range(3):
        t0 = time.perf_counter()
        native(t)
        best = max(best, n / (time.perf_counter() - t0))
    res["native_fps"][str(t)] = round(best)
    print("native", t, round(best), flush=True)


def pil_one(i):
    out[i] = np.asarray(Image.open(io.BytesIO(seq[i])).convert("RGB"))[..., 1]


m = 1024
for t in (1, 8, 16):
    t0 = time.perf_counter()
    if t == 1:
        for i in range(m):
            pil_one(i)
    else:
        with ThreadPoolExecutor(t) as ex:
            list(ex.map(pil_one, range(m)))
    res["pillow_fps"][str(t)] = round(m / (time.perf_counter() - t0))
    print("pillow", t, res["pillow_fps"][str(t)], flush=True)
print(json.dumps(res))
if len(sys.argv) > 1:
    os.makedirs(os.path.dirname(os.path.abspath(sys.argv[1])), exist_ok=True)
    json.dump(res, open(sys.argv[1], "w"), indent=1)

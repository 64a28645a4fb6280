"""Soak of the single-file calls (what CBAS's EncodeThread / ClassificationThread make): encode_file, infer_file and
encode_infer_file over mixed clips, many times, outputs byte-identical every time (rows streamed to the file while the clip
runs, decode-ahead threads, sessions reused).    python scripts/soak_single.py [rounds]"""
import hashlib
import os
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbas_amd import config as C, framesource as F, pipeline as P, synth, weights as W  # noqa: E402
from cbas_amd.encoder import DinoEncoder  # noqa: E402
from cbas_amd.head import ClassifierLSTMDeltas  # noqa: E402

sha = lambda p: hashlib.sha256(open(p, "rb").read()).hexdigest()  # noqa: E731
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
cfg = C.VIT_B16
enc = DinoEncoder.from_weights(cfg, W.synth_encoder_weights(cfg, 1234), "cuda", max_batch=64, max_frame=(224, 224))
head = ClassifierLSTMDeltas(768, 9)
head.load_state_dict(W.synth_head_weights(C.HeadConfig(), 4321))
head.to("cuda")
names = [f"b{i}" for i in range(9)]
root = tempfile.mkdtemp(prefix="cbas_soak1_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
base = synth.cage_frames(5, 400, 224, 224)
clips = []
for k, n in enumerate((1, 30, 513, 1500, 2600, 129)):
    fr = base[(np.arange(n) * (k + 2)) % 400]
    p = os.path.join(root, f"c{k}." + ("avi" if k % 2 else "npy"))
    F.write_mjpeg_avi(p, fr, quality=85, subsampling=2) if p.endswith(".avi") else np.save(p, fr)
    clips.append((p, n))
want = {}
frames = 0
t0 = time.perf_counter()
bad = 0
for r in range(rounds):
    for p, n in clips:
        if r % 2 == 0:
            h5, csv = P.encode_infer_file(enc, head, p, "m", names)
        else:
            h5 = P.encode_file(enc, p)
            csv = P.infer_file(h5, head, "m", names, 31, device="cuda")
        got = (sha(h5), sha(csv))
        bad += want.setdefault(p, got) != got
        os.remove(h5)
        os.remove(csv)
        frames += n
dt = time.perf_counter() - t0
print(f"{rounds} rounds x {len(clips)} clips, {frames} frames in {dt:.1f} s = {frames / dt:.0f} frames/s; outputs differing from the first round: {bad}")
head.close()
enc.close()
shutil.rmtree(root, ignore_errors=True)
sys.exit(1 if bad else 0)

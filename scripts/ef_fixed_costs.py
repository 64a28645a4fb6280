"""Where the fixed cost of one dist.encode_files call goes (one process): wall time against the number of clips, and
cProfile's view of the call with two clips.   python scripts/ef_fixed_costs.py"""
import cProfile
import os
import pstats
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbas_amd import config as C, dist as cdist, weights as W  # noqa: E402
from cbas_amd.encoder import DinoEncoder  # noqa: E402
from cbas_amd.head import ClassifierLSTMDeltas  # noqa: E402

cfg = C.VIT_B16
enc = DinoEncoder.from_weights(cfg, W.synth_encoder_weights(cfg, 1234), "cuda", max_batch=int(os.environ.get("CBAS_EF_MAX_BATCH", "64")),
                               max_frame=(224, 224))
head = ClassifierLSTMDeltas(768, 9)
head.load_state_dict(W.synth_head_weights(C.HeadConfig(), 4321))
head.to("cuda")
names = [f"b{i}" for i in range(9)]
root = tempfile.mkdtemp(prefix="cbas_ef_", dir="/dev/shm")
n = 4096
fr = torch.randint(0, 256, (n, 224, 224, 3), dtype=torch.uint8, device="cuda").cpu().numpy()
paths = []
for k in range(8):
    d = os.path.join(root, f"c{k}")
    os.makedirs(d)
    p = os.path.join(d, "clip.npy")
    if k == 0:
        np.save(p, fr)
    else:
        os.symlink(paths[0], p)
    paths.append(p)
del fr
for m in (1, 2, 4, 8, 1, 2, 4, 8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cdist.encode_files(paths[:m], enc, head=head, dataset_name="x", behaviors=names)
    dt = time.perf_counter() - t0
    print(f"{m} clips: {dt * 1e3:.1f} ms  ({m * n / dt:.0f} frames/s)", flush=True)
if os.environ.get("CBAS_EF_NO_PROFILE"):
    sys.exit(0)
pr = cProfile.Profile()
pr.enable()
cdist.encode_files(paths[:2], enc, head=head, dataset_name="x", behaviors=names)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)

"""Precision 4, one process per setting of CBAS_SPLIT_PP (read once per process): CLS rows of seeded frames saved for a
byte comparison, and the resident-frames step time.  usage: split_pp_ab.py OUT.npy [model batch hw iters]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbas_amd import config as C, weights as W, synth
from cbas_amd.encoder import DinoEncoder
out = sys.argv[1]
name = sys.argv[2] if len(sys.argv) > 2 else "vitb16"
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 64
hw = int(sys.argv[4]) if len(sys.argv) > 4 else 224
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 20
cfg = C.NAMED_VIT[name]
enc = DinoEncoder.from_weights(cfg, W.synth_encoder_weights(cfg, 1234), "cuda", max_batch=batch, max_frame=(hw, hw), precision=4)
fr = torch.from_numpy(synth.noise_frames(0, batch, hw, hw)[:, :, :, 1].copy()).cuda()
_, c32 = enc.encode_u8(fr)
torch.cuda.synchronize()
np.save(out, c32.cpu().numpy())
for _ in range(3):
    enc.encode_u8(fr, want_f32=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    enc.encode_u8(fr, want_f32=False)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
print(f"CBAS_SPLIT_PP={os.environ.get('CBAS_SPLIT_PP', '1')} CBAS_SPLIT_TILE={os.environ.get('CBAS_SPLIT_TILE', 'auto')} {name} batch={batch} {hw}: "
      f"{dt * 1e3:.3f} ms/batch, {batch / dt:.0f} fps", flush=True)

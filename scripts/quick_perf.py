"""Quick encoder throughput probe (device-resident uint8 frames), with a rocprof-friendly loop."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbas_amd import config as C, weights as W, synth
from cbas_amd.encoder import DinoEncoder
name = sys.argv[1] if len(sys.argv) > 1 else "vitb16"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
hw = int(sys.argv[4]) if len(sys.argv) > 4 else 224
prec = int(sys.argv[5]) if len(sys.argv) > 5 else 0
cfg = C.NAMED_VIT[name]
w = W.synth_encoder_weights(cfg, 1234)
enc = DinoEncoder.from_weights(cfg, w, "cuda", max_batch=batch, max_frame=(hw, hw), precision=prec)
fr = torch.from_numpy(synth.noise_frames(0, batch, hw, hw)[:, :, :, 1].copy()).cuda()
for _ in range(3):
    enc.encode_u8(fr, want_f32=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    enc.encode_u8(fr, want_f32=False)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
fps = batch / dt
print(f"{name} batch={batch} {hw}x{hw} prec={prec}: {dt*1e3:.3f} ms/batch, {fps:.0f} fps, {fps*cfg.flops_per_frame(hw,hw)/1e12:.1f} TFLOP/s")

"""Where the host path loses against resident frames: the bench's 157 x 64-frame clip pushed from device memory, from pinned
host memory as RGB, as a packed green plane (a third of the H2D bytes), and with the results left on the device.
Measured: 23.8k / 22.4k / 23.1k / 22.3k frames/s - about half of the 6 % is proportional to the H2D bytes."""
import sys, time, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbas_amd import config as C, weights as W
from cbas_amd.encoder import DinoEncoder
from cbas_amd.head import ClassifierLSTMDeltas
from cbas_amd.stream import ClipStream
cfg = C.NAMED_VIT["vitb16"]; hcfg = C.HeadConfig(in_features=768, out_features=9, seq_len=31)
dev = torch.device("cuda", 0)
enc = DinoEncoder.from_weights(cfg, W.synth_encoder_weights(cfg, 1234), dev, max_batch=64, max_frame=(224, 224))
head = ClassifierLSTMDeltas(768, 9, seq_len=31); head.load_state_dict(W.synth_head_weights(hcfg, 4321)); head.to(dev)
B, K = 64, 157
clip = torch.randint(0, 256, (B * K, 224, 224, 3), dtype=torch.uint8, device=dev)
host_rgb = torch.empty(clip.shape, dtype=torch.uint8).pin_memory(); host_rgb.copy_(clip)
host_g = torch.empty(clip.shape[:3], dtype=torch.uint8).pin_memory(); host_g.copy_(clip[..., 1])
hr, hg = host_rgb.numpy(), host_g.numpy()
st = ClipStream(enc, head, capacity=K * B, classify_every=1024)
def run(fn):
    st.reset()
    for s in range(K): fn(s)
def t(fn, fin):
    run(fn); fin(); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(fn); fin(); torch.cuda.synchronize(); return B * K / (time.perf_counter() - t0)
print("device-resident      ", round(t(lambda s: st.push_u8(clip[s*B:(s+1)*B]), st.finish)))
print("host RGB pinned      ", round(t(lambda s: st.push_host(hr[s*B:(s+1)*B]), st.finish_host)))
print("host green pinned    ", round(t(lambda s: st.push_host(hg[s*B:(s+1)*B]), st.finish_host)))
print("host RGB, finish dev ", round(t(lambda s: st.push_host(hr[s*B:(s+1)*B]), st.finish)))

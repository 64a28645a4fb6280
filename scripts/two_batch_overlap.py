"""Experiment: two full 64-frame batches in flight on two HIP streams (two encoder handles), so that one
batch's partial tile rounds / memory-bound kernels can be filled by the other batch's kernels."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbas_amd import config as C, weights as W
from cbas_amd.encoder import DinoEncoder
cfg = C.VIT_B16
w = W.synth_encoder_weights(cfg, 1234)
B, steps = 64, int(sys.argv[1]) if len(sys.argv) > 1 else 100
NL = int(sys.argv[2]) if len(sys.argv) > 2 else 2
encs = [DinoEncoder.from_weights(cfg, w, "cuda", max_batch=B, max_frame=(224, 224)) for _ in range(NL)]
clip = torch.randint(0, 256, (B * 8, 224, 224, 3), dtype=torch.uint8, device="cuda")
streams = [torch.cuda.Stream() for _ in range(NL)]

def run(two):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = []
    for s in range(steps):
        k = s % NL if two else 0
        with torch.cuda.stream(streams[k]):
            c16, _ = encs[k].encode_u8(clip[(s % 8) * B:(s % 8 + 1) * B], want_f32=False)
            outs.append(c16)
    torch.cuda.synchronize()
    return steps * B / (time.perf_counter() - t0), outs

run(False); run(True)
for _ in range(2):
    f1, o1 = run(False)
    f2, o2 = run(True)
    print(f"one stream: {f1:.0f} fps   {NL} batches in flight: {f2:.0f} fps   identical: {all(torch.equal(a, b) for a, b in zip(o1, o2))}", flush=True)

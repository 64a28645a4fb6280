"""GPU bring-up: stage-by-stage comparison of the HIP encoder with the numpy oracle (tiny ViT),
then CLS parity on the committed goldens.  Run on the GPU box: python scripts/bringup_enc.py"""
import os as _os
_os.environ.setdefault("CBAS_BUILD_DEBUG", "1")      # bring-up entry points: the debug build of the library
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cbas_amd import config as C, weights as W, synth, _lib
from cbas_amd.encoder import DinoEncoder
from oracle import vit_oracle as V

def rel(a, b):
    a = a.astype(np.float64); b = b.astype(np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30)), float(np.abs(a - b).max())

print(_lib.device_info(0))
cfg = C.VIT_TINY
w = W.synth_encoder_weights(cfg, 1234)
fr = synth.cage_frames(7, 4, 64, 64)
taps = {}
ref = V.vit_forward(np.repeat(V.preprocess_green(fr)[:, None], 3, 1), w, cfg, taps)
enc = DinoEncoder.from_weights(cfg, w, "cuda", max_batch=8, max_frame=(64, 64))
fd = torch.from_numpy(fr).cuda()
D = cfg.hidden_size
def flat(t): return t.reshape(-1, t.shape[-1])
for l in range(cfg.num_hidden_layers):
    if l == 0:
        print("emb     ", rel(enc.debug_tap(fd, 0, 0, 0), flat(taps["embeddings"])))
    print(f"l{l} ln1  ", rel(enc.debug_tap(fd, l, 1, 1), flat(taps[f"l{l}.ln1"])))
    qkv = enc.debug_tap(fd, l, 2, 2).astype(np.float32)
    print(f"l{l} q    ", rel(qkv[:, :D] * 8.0, flat(taps[f"l{l}.q_rope"])))
    print(f"l{l} k    ", rel(qkv[:, D:2*D], flat(taps[f"l{l}.k_rope"])))
    print(f"l{l} v    ", rel(qkv[:, 2*D:], flat(taps[f"l{l}.v"])))
    print(f"l{l} ctx  ", rel(enc.debug_tap(fd, l, 3, 1), flat(taps[f"l{l}.ctx"])))
    print(f"l{l} attn ", rel(enc.debug_tap(fd, l, 4, 0), flat(taps[f"l{l}.after_attn"])))
    print(f"l{l} ln2  ", rel(enc.debug_tap(fd, l, 5, 1), flat(taps[f"l{l}.ln2"])))
    print(f"l{l} up   ", rel(enc.debug_tap(fd, l, 6, 3), flat(taps[f"l{l}.up"])))
    print(f"l{l} out  ", rel(enc.debug_tap(fd, l, 7, 0), flat(taps[f"l{l}.out"])))
c16, c32 = enc.encode_u8(fd)
torch.cuda.synchronize()
print("tiny cls u8 ", rel(c32.cpu().numpy(), ref[:, 0]))
g = torch.from_numpy(V.preprocess_green(fr)).cuda()
print("tiny cls f32", rel(enc(g.unsqueeze(1)).squeeze(1).cpu().numpy(), ref[:, 0]))
enc.close()

gold = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden")
for name, cfgname, hw in (("vits16_224", "vits16", 224), ("vitb16_224", "vitb16", 224), ("vitb16_256", "vitb16", 256),
                          ("vitb16_224_noise", "vitb16", 224), ("vitl16_224", "vitl16", 224)):
    g = np.load(os.path.join(gold, name + ".npz"))
    cfg = C.NAMED_VIT[cfgname]
    w = W.synth_encoder_weights(cfg, 1234)
    n = int(g["n"])
    mk = synth.noise_frames if str(g["kind"]) == "noise" else synth.cage_frames
    fr = mk(int(g["frame_seed"]), n, hw, hw)
    for prec in (0, 1):
        enc = DinoEncoder.from_weights(cfg, w, "cuda", max_batch=8, max_frame=(hw, hw), precision=prec)
        c16, c32 = enc.encode_u8(torch.from_numpy(fr).cuda())
        torch.cuda.synchronize()
        out = c32.cpu().numpy()
        per = np.linalg.norm(out - g["cls"], axis=1) / np.linalg.norm(g["cls"], axis=1)
        print(f"{name} prec={prec} max rel {per.max():.3e}  maxabs {np.abs(out-g['cls']).max():.3e}")
        enc.close()

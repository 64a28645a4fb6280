import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from cbas_amd import config as C, weights as W, synth
from cbas_amd.encoder import DinoEncoder
gd = "/root/repo/tests/golden"
def rel_rows(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return np.linalg.norm(a - b, axis=1) / np.linalg.norm(b, axis=1)
for name, cfgname, hw in (("vits16_224","vits16",224),("vitb16_224","vitb16",224),("vitb16_224_noise","vitb16",224),("vitb16_256","vitb16",256),("vitl16_224","vitl16",224),("vitl16_518","vitl16",518)):
    g = np.load(os.path.join(gd, name + ".npz"))
    cfg = C.NAMED_VIT[cfgname]; n = int(g["n"])
    mk = synth.noise_frames if str(g["kind"]) == "noise" else synth.cage_frames
    fr = mk(int(g["frame_seed"]), n, hw, hw)
    out = []
    for prec in (3, 4):
        enc = DinoEncoder.from_weights(cfg, W.synth_encoder_weights(cfg, 1234), "cuda", max_batch=8, max_frame=(hw, hw), precision=prec)
        _, c32 = enc.encode_u8(torch.from_numpy(fr).cuda()); torch.cuda.synchronize()
        out.append(rel_rows(c32.cpu().numpy(), g["cls"]).max()); enc.close()
    print(f"{name}: CLS rel err precision 3 {out[0]:.3e}  precision 4 {out[1]:.3e}", flush=True)

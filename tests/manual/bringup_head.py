"""GPU bring-up of the classifier head against the goldens made from the reference module."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cbas_amd import config as C, weights as W, synth
from cbas_amd.head import ClassifierLSTMDeltas
gold = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden")
for tag, h, C_, I in (("h64", 64, 9, 768), ("h128", 128, 5, 768), ("h64_d384", 64, 9, 384)):
    g = np.load(os.path.join(gold, f"head_{tag}.npz"))
    hc = C.HeadConfig(in_features=I, out_features=C_, lstm_hidden_size=h)
    hw = W.synth_head_weights(hc, 4321)
    m = ClassifierLSTMDeltas(I, C_, lstm_hidden_size=h)
    m.load_state_dict(hw); m.to("cuda")
    seq = synth.cls_walk(21, 94, I).astype(np.float32)
    x = torch.from_numpy(np.stack([seq[i:i+31] for i in range(64)])).cuda()
    lo, la = m(x)
    lo, la = lo.cpu().numpy(), la.cpu().numpy()
    print(tag, "logits maxabs", np.abs(lo - g["logits"]).max(), "latent", np.abs(la - g["latent"]).max(),
          "labels equal", bool((lo.argmax(1) == g["logits"].argmax(1)).all()))
    m.close()
g = np.load(os.path.join(gold, "infer_file.npz"))
hc = C.HeadConfig()
m = ClassifierLSTMDeltas(768, 9); m.load_state_dict(W.synth_head_weights(hc, 4321)); m.to("cuda")
for n in (1, 10, 31, 64, 700, 20017, 40):
    cls = torch.from_numpy(synth.cls_walk(100 + n, n, 768)).cuda()
    t = float(g[f"temp_{n}"])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    p = m.infer_clip(cls, t)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    p = p.cpu().numpy()
    ref = g[f"probs_{n}"]
    print(f"infer n={n} T={t}: maxabs {np.abs(p-ref).max():.3e} label mismatches {(p.argmax(1)!=ref.argmax(1)).sum()} time {dt*1e3:.1f} ms")
cls = torch.from_numpy(synth.cls_walk(5, 10000, 768)).cuda()
for _ in range(2): m.infer_clip(cls)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): m.infer_clip(cls)
torch.cuda.synchronize(); print("10k-frame clip head time ms", (time.perf_counter()-t0)/5*1e3)

"""LayerNorm fold A/B (cbas_enc_debug_option "ln_fold"): CLS error against the reference goldens with the fold on and off, the
distance between the two settings, and interleaved timing rounds of the bench loop in one process (rule 24)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbas_amd import config as C, weights as W, synth
from cbas_amd.encoder import DinoEncoder
from cbas_amd.head import ClassifierLSTMDeltas
from cbas_amd.stream import ClipStream

model = sys.argv[1] if len(sys.argv) > 1 else "vitb16"
hw = int(sys.argv[2]) if len(sys.argv) > 2 else 224
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 40
cfg = C.NAMED_VIT[model]
enc = DinoEncoder.from_weights(cfg, W.synth_encoder_weights(cfg, 1234), "cuda", max_batch=B, max_frame=(hw, hw))
gd = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
name = {("vitb16", 224): "vitb16_224_noise", ("vitb16", 256): "vitb16_256", ("vitl16", 224): "vitl16_224", ("vitl16", 518): "vitl16_518"}.get((model, hw))
fill = torch.randint(0, 256, (B, hw, hw, 3), dtype=torch.uint8, device="cuda")
if name:
    g = np.load(os.path.join(gd, name + ".npz"))
    mk = synth.noise_frames if str(g["kind"]) == "noise" else synth.cage_frames
    fr = torch.from_numpy(mk(int(g["frame_seed"]), int(g["n"]), hw, hw)).cuda()
    n = fr.shape[0]
    batch = torch.cat([fr, fill[n:]]) if n < B else fr[:B]
    n = min(n, B)
    out = {}
    for fold in (1, 0, 1):
        enc.debug_option("ln_fold", fold)
        c16, c32 = enc.encode_u8(batch)
        small16, small32 = enc.encode_u8(batch[:n])                 # the same frames in a small batch
        torch.cuda.synchronize()
        ref = g["cls"][:n].astype(np.float64)
        got = c32[:n].cpu().numpy().astype(np.float64)
        rel = np.linalg.norm(got - ref, axis=1) / np.linalg.norm(ref, axis=1)
        inv = bool(torch.equal(c32[:n], small32))
        print(f"ln_fold={fold}: CLS rel err vs reference golden max {rel.max():.3e} mean {rel.mean():.3e}; batch-invariant (bit-exact, {B} vs {n} frames): {inv}")
        out[fold] = c32.clone()
    d = (torch.linalg.norm(out[1] - out[0], dim=1) / torch.linalg.norm(out[0], dim=1)).max().item()
    print(f"fold on vs off: max rel distance {d:.3e}")
hcfg = C.HeadConfig(in_features=cfg.hidden_size, out_features=9)
head = ClassifierLSTMDeltas(cfg.hidden_size, 9); head.load_state_dict(W.synth_head_weights(hcfg, 4321)); head.to("cuda")
clip = torch.randint(0, 256, (steps * B, hw, hw, 3), dtype=torch.uint8, device="cuda")
st = ClipStream(enc, head, capacity=steps * B)
def run():
    st.reset()
    for s in range(steps):
        st.push_u8(clip[s * B:(s + 1) * B])
    st.finish(); torch.cuda.synchronize()
res = {0: [], 1: []}
for rnd in range(6):
    for fold in (1, 0):
        enc.debug_option("ln_fold", fold)
        run()
        t0 = time.perf_counter(); run(); dt = time.perf_counter() - t0
        res[fold].append(dt / steps * 1e3)
for fold in (0, 1):
    r = sorted(res[fold][1:])
    print(f"ln_fold={fold}: ms/step median {np.median(r):.4f} min {r[0]:.4f} ({B * 1e3 / np.median(r):.0f} frames/s)")
print(f"speed-up (median): {np.median(sorted(res[0][1:])) / np.median(sorted(res[1][1:])):.4f}x")
# per-kernel HIP-event timing, one lane
enc.set_lanes(1)
for fold in (0, 1):
    enc.debug_option("ln_fold", fold)
    run()
    enc.profile(True); run(); prof = enc.profile_read(); enc.profile(False)
    tot = sum(v["ms"] for v in prof.values())
    print(f"ln_fold={fold} (one lane, events): " + ", ".join(f"{k} {v['ms'] * 1e3 / v['launches']:.1f}us x{v['launches'] // steps}" for k, v in prof.items()) + f"; sum {tot / steps:.3f} ms/step")

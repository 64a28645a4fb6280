"""Soak: many steps of the streamed path, twice, with two batches in flight; outputs must be bit-identical."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbas_amd import config as C, weights as W
from cbas_amd.encoder import DinoEncoder
from cbas_amd.head import ClassifierLSTMDeltas
from cbas_amd.stream import ClipStream
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
cfg = C.VIT_B16
enc = DinoEncoder.from_weights(cfg, W.synth_encoder_weights(cfg, 1234), "cuda", max_batch=64, max_frame=(224, 224))
head = ClassifierLSTMDeltas(768, 9); head.load_state_dict(W.synth_head_weights(C.HeadConfig(), 4321)); head.to("cuda")
clip = torch.randint(0, 256, (64 * 40, 224, 224, 3), dtype=torch.uint8, device="cuda")
st = ClipStream(enc, head, capacity=steps * 64)
def run():
    st.reset(); t0 = time.perf_counter()
    for s in range(steps):
        o = (s % 40) * 64
        st.push_u8(clip[o:o + 64])
    c, p = st.finish(); torch.cuda.synchronize()
    return c.clone(), p.clone(), steps * 64 / (time.perf_counter() - t0)
c1, p1, f1 = run(); c2, p2, f2 = run()
print(f"{steps} steps x2: {f1:.0f} / {f2:.0f} fps; identical: {torch.equal(c1, c2) and torch.equal(p1, p2)}; finite: {bool(torch.isfinite(p1).all())}; "
      f"period-consistent: {torch.equal(c1[:2560], c1[2560:5120])}")

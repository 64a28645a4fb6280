"""Size-independent properties at BASELINE.json's full size (config 2: ViT-B/16, 10k-frame clip,
batch 64, streamed encode + head), where the oracle cannot be run end to end in seconds:
determinism, batch/position independence, streaming == whole-clip, probability simplex, and a
sampled comparison with the oracle (head on the produced fp16 rows; encoder on a few frames)."""
import numpy as np
import pytest
import torch

from cbas_amd import config as C, weights as W

pytestmark = pytest.mark.gpu

N_FRAMES, BATCH = 10048, 64          # 157 steps of 64: the bench workload


@pytest.fixture(scope="module")
def clip_run():
    from cbas_amd.encoder import DinoEncoder
    from cbas_amd.head import ClassifierLSTMDeltas
    from cbas_amd.stream import ClipStream
    cfg = C.VIT_B16
    enc_w = W.synth_encoder_weights(cfg, 1234)
    head_w = W.synth_head_weights(C.HeadConfig(), 4321)
    enc = DinoEncoder.from_weights(cfg, enc_w, "cuda", max_batch=BATCH, max_frame=(224, 224))
    head = ClassifierLSTMDeltas(768, 9)
    head.load_state_dict(head_w)
    head.to("cuda")
    gen = torch.Generator(device="cuda")
    gen.manual_seed(7)
    # a clip with temporal structure: 16 "scenes" of noise, cross-faded, so labels change over time
    base = torch.randint(0, 256, (16, 224, 224), dtype=torch.uint8, device="cuda", generator=gen).float()
    idx = torch.arange(N_FRAMES, device="cuda")
    seg, frac = (idx // 600) % 16, ((idx % 600).float() / 600.0).clamp(0, 1)
    nxt = (seg + 1) % 16
    noise = torch.randint(0, 32, (N_FRAMES, 1, 1), device="cuda", generator=gen).float()
    clip = torch.empty((N_FRAMES, 224, 224), dtype=torch.uint8, device="cuda")
    for s in range(0, N_FRAMES, 512):
        e = min(s + 512, N_FRAMES)
        f = frac[s:e, None, None]
        clip[s:e] = (base[seg[s:e]] * (1 - f) + base[nxt[s:e]] * f + noise[s:e]).clamp(0, 255).to(torch.uint8)

    def run(classify_every):
        st = ClipStream(enc, head, capacity=N_FRAMES, classify_every=classify_every)
        for s in range(0, N_FRAMES, BATCH):
            st.push_u8(clip[s:s + BATCH])
        c, p = st.finish()
        torch.cuda.synchronize()
        return c.clone(), p.clone()

    c1, p1 = run(1024)
    yield dict(enc=enc, head=head, clip=clip, cls=c1, probs=p1, run=run, enc_w=enc_w, head_w=head_w, cfg=cfg)
    enc.close()
    head.close()


def test_deterministic_and_segmentation_independent(clip_run):
    c2, p2 = clip_run["run"](4096)                    # different classification grouping, second full pass
    assert torch.equal(c2, clip_run["cls"])           # bit-identical CLS rows
    assert torch.equal(p2, clip_run["probs"])         # bit-identical probabilities


def test_probabilities_form_a_simplex_and_labels_vary(clip_run):
    p = clip_run["probs"]
    assert p.shape == (N_FRAMES, 9) and torch.isfinite(p).all() and (p >= 0).all()
    assert (p.sum(1) - 1).abs().max() < 1e-5
    assert torch.isfinite(clip_run["cls"].float()).all()
    assert len(torch.unique(p.argmax(1))) >= 2


def test_streamed_head_equals_whole_clip_head(clip_run):
    whole = clip_run["head"].infer_clip(clip_run["cls"])
    torch.cuda.synchronize()
    assert torch.equal(whole, clip_run["probs"])


def test_frame_position_independence(clip_run):
    """A frame's CLS row does not depend on where in the clip / batch it was encoded."""
    rng = np.random.default_rng(0)
    pick = np.sort(rng.choice(N_FRAMES, 40, replace=False))
    sub = clip_run["clip"][torch.from_numpy(pick).cuda()]
    c16, _ = clip_run["enc"].encode_u8(sub, want_f32=False)
    torch.cuda.synchronize()
    assert torch.equal(c16, clip_run["cls"][torch.from_numpy(pick).cuda()])


def test_sampled_oracle_agreement(clip_run):
    """Oracle spot checks at full size: the ViT on 4 sampled frames (1e-3 relative) and the reference
    head semantics on 48 sampled windows of the produced fp16 rows (identical labels)."""
    from oracle import head_oracle as H
    from oracle import pipeline_oracle as PO
    from conftest import assert_labels_match
    pick = [0, 3333, 7000, N_FRAMES - 1]
    fr = clip_run["clip"][pick].cpu().numpy()
    rgb = np.repeat(fr[..., None], 3, axis=-1)
    ref = PO.encode_frames(rgb, clip_run["enc_w"], clip_run["cfg"], batch=4)
    got = clip_run["cls"][pick].float().cpu().numpy()
    rel = np.linalg.norm(got - ref, axis=1) / np.linalg.norm(ref, axis=1)
    assert rel.max() < 1.5e-3, rel.max()            # 1e-3 + the fp16 storage rounding of the row
    cls16 = clip_run["cls"].cpu().numpy()
    sel = np.r_[0:16, 5000:5016, N_FRAMES - 16:N_FRAMES]
    idx = H.infer_windows(cls16, 31)[sel]
    logits, _ = H.head_forward(cls16.astype(np.float32)[idx], clip_run["head_w"], 31)
    ref_p = H.softmax_T(logits, 1.0)
    n_mis, _ = assert_labels_match(clip_run["probs"][sel].cpu().numpy(), ref_p, 1e-4, margin=0.0)   # same rows in: every label
    assert n_mis == 0


def test_handles_are_usable_from_concurrent_threads():
    """EncodeThread and ClassificationThread are distinct OS threads in CBAS (workthreads.py:1256-1267) and a
    TrainingThread may run too: an encoder, a head and a trainer driven from three Python threads at once give
    the results of running them one after the other."""
    import threading
    from cbas_amd.encoder import DinoEncoder
    from cbas_amd.head import ClassifierLSTMDeltas
    from cbas_amd.train import HeadTrainer
    from cbas_amd import synth
    cfg = C.NAMED_VIT["vits16"]
    enc = DinoEncoder.from_weights(cfg, W.synth_encoder_weights(cfg, 1234), "cuda", max_batch=32, max_frame=(224, 224))
    hcfg = C.HeadConfig()
    hw = W.synth_head_weights(hcfg, 4321)
    head = ClassifierLSTMDeltas(768, 9)
    head.load_state_dict(hw)
    head.to("cuda")
    frames = torch.from_numpy(synth.cage_frames(2, 96, 224, 224)).cuda()
    cls = torch.from_numpy(synth.cls_walk(4, 3000, 768)).cuda()
    x, y = synth.train_windows(6, 128, 768, 9, 31)
    xt, yt = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()

    def enc_job(out):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(6):
                out.append(enc.encode_u8(frames, want_f32=False)[0])
            s.synchronize()

    def head_job(out):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(6):
                out.append(head.infer_clip(cls))
            s.synchronize()

    def train_job(out):
        tr = HeadTrainer(hcfg, hw, "cuda", lr=1e-3, max_batch=128, seed=3)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            out.extend(tr.step(xt, yt)[0] for _ in range(6))
        out.append(tr.weights())
        tr.close()

    seq = ([], [], [])
    enc_job(seq[0]); head_job(seq[1]); train_job(seq[2])
    par = ([], [], [])
    threads = [threading.Thread(target=f, args=(o,)) for f, o in ((enc_job, par[0]), (head_job, par[1]), (train_job, par[2]))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
        assert not t.is_alive()
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(seq[0], par[0])) and all(torch.equal(a, seq[0][0]) for a in par[0])
    assert all(torch.equal(a, b) for a, b in zip(seq[1], par[1]))
    assert seq[2][:6] == par[2][:6] and all(np.array_equal(seq[2][6][k], par[2][6][k]) for k in seq[2][6])
    enc.close()
    head.close()

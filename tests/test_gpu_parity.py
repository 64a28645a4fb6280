"""Parity of the HIP path (through the C ABI) with the oracle and with the goldens made from the
reference.  Bars: CLS rows within 1e-3 relative (||d||2/||ref||2 per frame, BASELINE.json
north_star), fp32 head within 1e-4 absolute on probabilities, argmax labels identical."""
import os

import numpy as np
import pytest
import torch

from cbas_amd import config as C, weights as W, synth, _lib
from conftest import assert_labels_match

pytestmark = pytest.mark.gpu

CLS_TOL = 1e-3
NAMES = ["eating", "drinking", "rearing", "climbing", "digging", "nesting", "resting", "grooming", "exploring"]


def rel_rows(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return np.linalg.norm(a - b, axis=1) / np.linalg.norm(b, axis=1)


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


@pytest.fixture(scope="module")
def tiny():
    from cbas_amd.encoder import DinoEncoder
    cfg = C.VIT_TINY
    w = W.synth_encoder_weights(cfg, 1234)
    enc = DinoEncoder.from_weights(cfg, w, "cuda", max_batch=8, max_frame=(64, 64))
    yield cfg, w, enc
    enc.close()


def test_native_library_is_loaded():
    lib = _lib.load(build_if_missing=False)
    arch, ncu, hbm = _lib.device_info(0)
    assert arch.startswith("gfx950"), arch
    assert ncu == 256 and hbm > 200e9
    maps = open("/proc/self/maps").read()
    assert os.path.basename(_lib.library_path()) in maps          # the suite runs on the debug build (tests/conftest.py)
    assert _lib.is_debug() and lib.cbas_debug_build() == 1


def test_tiny_stagewise_against_oracle(tiny):
    """Every kernel of the encoder in isolation: ingest+patch GEMM, LayerNorm, QKV+RoPE, attention,
    o_proj+LayerScale+residual, up_proj+GELU, down_proj+LayerScale+residual."""
    from oracle import vit_oracle as V
    cfg, w, enc = tiny
    fr = synth.cage_frames(7, 4, 64, 64)
    taps = {}
    V.vit_forward(np.repeat(V.preprocess_green(fr)[:, None], 3, 1), w, cfg, taps)
    fd = torch.from_numpy(fr).cuda()
    D = cfg.hidden_size

    def flat(t):
        return t.reshape(-1, t.shape[-1])

    def close(got, want, tol):
        got, want = got.astype(np.float64), flat(want).astype(np.float64)
        assert np.linalg.norm(got - want) / np.linalg.norm(want) < tol

    close(enc.debug_tap(fd, 0, 0, 0), taps["embeddings"], 3e-4)
    for l in range(cfg.num_hidden_layers):
        close(enc.debug_tap(fd, l, 1, 1), taps[f"l{l}.ln1"], 6e-4)
        qkv = enc.debug_tap(fd, l, 2, 2).astype(np.float32)
        close(qkv[:, :D] * 8.0, taps[f"l{l}.q_rope"], 8e-4)       # q is stored pre-scaled by 1/8
        close(qkv[:, D:2 * D], taps[f"l{l}.k_rope"], 8e-4)
        close(qkv[:, 2 * D:], taps[f"l{l}.v"], 8e-4)
        close(enc.debug_tap(fd, l, 3, 1), taps[f"l{l}.ctx"], 8e-4)
        close(enc.debug_tap(fd, l, 4, 0), taps[f"l{l}.after_attn"], 3e-4)
        close(enc.debug_tap(fd, l, 5, 1), taps[f"l{l}.ln2"], 6e-4)
        close(enc.debug_tap(fd, l, 6, 3), taps[f"l{l}.up"], 8e-4)
        close(enc.debug_tap(fd, l, 7, 0), taps[f"l{l}.out"], 3e-4)


def test_tiny_all_ingest_paths_agree(tiny, golden_dir):
    """uint8 RGB (decord layout), packed green plane, float32 DinoEncoder.forward and the host-streamed
    slots all give the golden CLS."""
    from oracle import vit_oracle as V
    cfg, w, enc = tiny
    g = load(golden_dir, "vit_tiny")
    fr = synth.cage_frames(7, 4, 64, 64)
    want = g["last_hidden"][:, 0]
    c16, c32 = enc.encode_u8(torch.from_numpy(fr).cuda())
    torch.cuda.synchronize()
    assert rel_rows(c32.cpu().numpy(), want).max() < CLS_TOL
    assert np.array_equal(c16.cpu().numpy(), c32.cpu().numpy().astype(np.float16))     # RNE, like the h5 write
    p16, p32 = enc.encode_u8(torch.from_numpy(fr[:, :, :, 1].copy()).cuda())
    torch.cuda.synchronize()
    assert torch.equal(p32, c32) and torch.equal(p16, c16)
    x = torch.from_numpy(V.preprocess_green(fr)).cuda().unsqueeze(1)                    # (B,1,H,W) as cbas.py:435
    y = enc(x)
    assert y.shape == (4, 1, cfg.hidden_size) and y.dtype == torch.float32
    assert rel_rows(y.squeeze(1).cpu().numpy(), want).max() < CLS_TOL
    enc.submit_host(0, fr[:3])
    enc.submit_host(1, fr[3:])
    a16, a32 = enc.wait(0, want_f32=True)
    b16, b32 = enc.wait(1, want_f32=True)
    assert np.array_equal(np.concatenate([a32, b32]), c32.cpu().numpy())                # batch-composition invariant
    with pytest.raises(RuntimeError):
        enc.wait(0)                                                                     # idle slot -> CBAS_ESTATE


def test_error_codes(tiny):
    cfg, w, enc = tiny
    fr = torch.zeros((9, 64, 64, 3), dtype=torch.uint8, device="cuda")
    lib = _lib.load()
    rc = lib.cbas_enc_forward_u8(enc._h, fr.data_ptr(), 9, 64, 64, 64 * 64 * 3, 64 * 3, 3, None, None, None)
    assert rc == -1 and b"max_batch" in lib.cbas_last_error()
    rc = lib.cbas_enc_forward_u8(enc._h, fr.data_ptr(), 2, 128, 128, 1, 1, 1, None, None, None)
    assert rc == -1 and b"workspace" in lib.cbas_last_error()


@pytest.mark.parametrize("name,cfgname,hw,prec", [("vits16_224", "vits16", 224, 0), ("vitb16_224", "vitb16", 224, 0),
                                                  ("vitb16_224_noise", "vitb16", 224, 0), ("vitb16_256", "vitb16", 256, 0),
                                                  ("vitl16_224", "vitl16", 224, 0), ("vitb16_224", "vitb16", 224, 1),
                                                  ("vitl16_518", "vitl16", 518, 0)])   # T = 1029: streaming attention
def test_cls_goldens(golden_dir, name, cfgname, hw, prec):
    from cbas_amd.encoder import DinoEncoder
    g = load(golden_dir, name)
    cfg = C.NAMED_VIT[cfgname]
    n = int(g["n"])
    mk = synth.noise_frames if str(g["kind"]) == "noise" else synth.cage_frames
    fr = mk(int(g["frame_seed"]), n, hw, hw)
    enc = DinoEncoder.from_weights(cfg, W.synth_encoder_weights(cfg, 1234), "cuda", max_batch=8, max_frame=(hw, hw),
                                   precision=prec)
    try:
        _, c32 = enc.encode_u8(torch.from_numpy(fr).cuda())
        torch.cuda.synchronize()
        r = rel_rows(c32.cpu().numpy(), g["cls"])
        assert r.max() < CLS_TOL, r.max()
    finally:
        enc.close()


def test_vitb_full_batch_invariance():
    """At the bench's batch (64 frames, M = 12 864 rows): a frame's CLS does not depend on its batch
    position or on the batch size (bit-exact), and duplicates give duplicates."""
    from cbas_amd.encoder import DinoEncoder
    cfg = C.VIT_B16
    enc = DinoEncoder.from_weights(cfg, W.synth_encoder_weights(cfg, 1234), "cuda", max_batch=64, max_frame=(224, 224))
    try:
        fr = synth.noise_frames(5, 8, 224, 224)
        big = np.concatenate([fr] * 8)                      # 64 frames: 8 copies of 8
        perm = np.random.default_rng(0).permutation(64)
        a16, _ = enc.encode_u8(torch.from_numpy(big).cuda(), want_f32=False)
        b16, _ = enc.encode_u8(torch.from_numpy(big[perm]).cuda(), want_f32=False)
        s16, _ = enc.encode_u8(torch.from_numpy(fr[:3]).cuda(), want_f32=False)
        torch.cuda.synchronize()
        assert torch.equal(a16[perm], b16)
        assert torch.equal(a16[:8], a16[8:16]) and torch.equal(a16[:3], s16)
        assert torch.isfinite(a16.float()).all()
    finally:
        enc.close()


@pytest.mark.parametrize("tag,h,ncls,dim,nl", [("h64", 64, 9, 768, 1), ("h128", 128, 5, 768, 1),
                                               ("h64_d384", 64, 9, 384, 1), ("h64_l2", 64, 9, 768, 2)])
def test_head_forward_goldens(golden_dir, tag, h, ncls, dim, nl):
    from cbas_amd.head import ClassifierLSTMDeltas
    g = load(golden_dir, f"head_{tag}")
    hc = C.HeadConfig(in_features=dim, out_features=ncls, lstm_hidden_size=h, lstm_layers=nl)
    m = ClassifierLSTMDeltas(dim, ncls, lstm_hidden_size=h, lstm_layers=nl)
    m.load_state_dict(W.synth_head_weights(hc, 4321))
    m.to("cuda")
    assert next(m.parameters()).device.type == "cuda"
    seq = synth.cls_walk(21, 94, dim).astype(np.float32)
    x = torch.from_numpy(np.stack([seq[i:i + 31] for i in range(64)])).cuda()
    logits, latent = m(x)
    assert logits.shape == (64, ncls) and latent.shape == (64, 2 * h)
    np.testing.assert_allclose(logits.cpu().numpy(), g["logits"], atol=1e-4)
    np.testing.assert_allclose(latent.cpu().numpy(), g["latent"], atol=5e-5)
    assert (logits.cpu().numpy().argmax(1) == g["logits"].argmax(1)).all()
    with pytest.raises(ValueError):
        m(torch.zeros(2, 30, dim))
    m.close()


@pytest.mark.parametrize("n", [1, 10, 31, 64, 700, 20017, 40])
def test_infer_clip_goldens(golden_dir, n):
    """Edge replicate padding (n < seq_len), exact window size, the 20 000-frame chunk seam of the
    reference and the temperature clamp max(1e-3, T)."""
    from cbas_amd.head import ClassifierLSTMDeltas
    g = load(golden_dir, "infer_file")
    m = ClassifierLSTMDeltas(768, 9)
    m.load_state_dict(W.synth_head_weights(C.HeadConfig(), 4321))
    m.to("cuda")
    cls = torch.from_numpy(synth.cls_walk(100 + n, n, 768)).cuda()
    probs = m.infer_clip(cls, float(g[f"temp_{n}"])).cpu().numpy()
    ref = g[f"probs_{n}"]
    np.testing.assert_allclose(probs, ref, atol=1e-4)
    assert (probs.argmax(1) == ref.argmax(1)).all()
    np.testing.assert_allclose(probs.sum(1), 1.0, atol=1e-5)
    m.close()


def test_infer_range_equals_whole_clip():
    """Segmented (streaming) classification == one pass over the clip, bit for bit."""
    from cbas_amd.head import ClassifierLSTMDeltas
    m = ClassifierLSTMDeltas(768, 9)
    m.load_state_dict(W.synth_head_weights(C.HeadConfig(), 4321))
    m.to("cuda")
    n = 9000
    cls = torch.from_numpy(synth.cls_walk(77, n, 768)).cuda()
    whole = m.infer_clip(cls)
    out = torch.zeros_like(whole)
    done = 0
    for enc in (1000, 1500, 6000, n):                   # rows "encoded so far"
        cnt = (enc - 15 - done) if enc < n else n - done
        m.infer_range_into(cls, enc, done, cnt, out)
        done += cnt
    torch.cuda.synchronize()
    assert torch.equal(out, whole)
    m.close()


def test_e2e_config1(golden_dir):
    """BASELINE config 1 end to end on the GPU: ViT-S/16, 64 frames, C=9 -> CLS tolerance and
    identical argmax labels vs the reference's own outputs."""
    from cbas_amd.encoder import DinoEncoder
    from cbas_amd.head import ClassifierLSTMDeltas
    from cbas_amd.stream import ClipStream
    g = load(golden_dir, "e2e_vits16")
    cfg = C.VIT_S16
    fr = synth.cage_frames(3, 64, 224, 224)
    enc = DinoEncoder.from_weights(cfg, W.synth_encoder_weights(cfg, 1234), "cuda", max_batch=8, max_frame=(224, 224))
    head = ClassifierLSTMDeltas(384, 9)
    head.load_state_dict(W.synth_head_weights(C.HeadConfig(in_features=384), 4321))
    head.to("cuda")
    st = ClipStream(enc, head, capacity=64, classify_every=16)
    fd = torch.from_numpy(fr).cuda()
    for i in range(0, 64, 8):
        st.push_u8(fd[i:i + 8])
    cls16, probs = st.finish()
    torch.cuda.synchronize()
    r = rel_rows(cls16.float().cpu().numpy(), g["cls"])
    assert r.max() < CLS_TOL + 5e-4          # includes the fp16 storage rounding (2^-11)
    probs = probs.cpu().numpy()
    # fp16 encoder (CLS within 1e-3) -> probabilities within 1e-2; labels identical outside near-ties
    n_mis, n_near = assert_labels_match(probs, g["probs"], 1e-2)      # fixed margin (conftest.MARGIN_FP16): no flip at or above it
    # Below it: identical labels cannot be promised by the fp16-operand arithmetic (the reference's own fp16-autocast GPU
    # path has the same property; precision 3 is the mode that reproduces them, tests/test_gpu_fp32.py), so what is pinned
    # is how narrow the exception is: at most one frame of this clip, and only where the reference's own top-2 margin is
    # under 1e-2.
    ref = g["probs"].astype(np.float64)
    srt = np.sort(ref, axis=1)
    flips = np.nonzero(probs.argmax(1) != ref.argmax(1))[0]
    for f in flips:
        print(f"[e2e_config1] frame {f}: reference top-2 margin {srt[f, -1] - srt[f, -2]:.3e}, |dp| there "
              f"{np.abs(probs[f] - ref[f]).max():.3e}")
    assert n_mis <= 1 and all(srt[f, -1] - srt[f, -2] < 1e-2 for f in flips), (n_mis, n_near)
    assert len(set(g["labels"].tolist())) >= 3            # the golden clip really changes behaviour
    # same fp16 rows in -> the fp32 head reproduces the reference labels exactly
    p_same = head.infer_clip(torch.from_numpy(g["cls_f16"]).cuda()).cpu().numpy()
    np.testing.assert_allclose(p_same, g["probs"], atol=1e-4)
    assert (p_same.argmax(1) == g["labels"]).all()
    enc.close(); head.close()


def test_e2e_config2_model_vitb(golden_dir):
    """The headline model end to end against the reference's own wrapper + infer_file (tests/golden/e2e_vitb16.npz: ViT-B/16,
    256 frames 224^2, C = 9): CLS within the 1e-3 contract on every frame (measured 6.3e-4), probabilities within 3e-2
    (measured 1.9e-2, at a behaviour transition, where the head's probabilities are steepest), no label flip at a
    reference top-2 margin >= conftest.MARGIN_FP16, and at most two flips in all, only under 1e-2 (two such frames, margins
    4e-5 and 6e-3; measured: 1 flip of 256, on one of them).  The fp32 mode's strict gate: tests/test_gpu_fp32.py."""
    from cbas_amd.encoder import DinoEncoder
    from cbas_amd.head import ClassifierLSTMDeltas
    from cbas_amd.stream import ClipStream
    g = load(golden_dir, "e2e_vitb16")
    n = int(g["n"])
    cfg = C.VIT_B16
    fr = synth.cage_frames(int(g["frame_seed"]), n, 224, 224)
    enc = DinoEncoder.from_weights(cfg, W.synth_encoder_weights(cfg, 1234), "cuda", max_batch=64, max_frame=(224, 224))
    head = ClassifierLSTMDeltas(768, 9)
    head.load_state_dict(W.synth_head_weights(C.HeadConfig(in_features=768), 4321))
    head.to("cuda")
    st = ClipStream(enc, head, capacity=n)
    fd = torch.from_numpy(fr).cuda()
    for i in range(0, n, 64):
        st.push_u8(fd[i:i + 64])
    cls16, probs = st.finish()
    torch.cuda.synchronize()
    r = rel_rows(cls16.float().cpu().numpy(), g["cls"])
    probs = probs.cpu().numpy()
    ref = g["probs"].astype(np.float64)
    srt = np.sort(ref, axis=1)
    flips = np.nonzero(probs.argmax(1) != ref.argmax(1))[0]
    print(f"[e2e_vitb16] CLS rel err max {r.max():.3e}; |dp| max {np.abs(probs - ref).max():.3e}; {len(flips)} of {n} labels differ; "
          f"reference frames with a top-2 margin under 1e-2: {int(((srt[:, -1] - srt[:, -2]) < 1e-2).sum())}")
    assert r.max() < CLS_TOL + 5e-4                       # includes the fp16 storage rounding (2^-11)
    n_mis, n_near = assert_labels_match(probs, g["probs"], 3e-2)
    assert n_mis <= 2 and all(srt[f, -1] - srt[f, -2] < 1e-2 for f in flips), (n_mis, n_near)
    assert len(set(g["labels"].tolist())) >= 2
    p_same = head.infer_clip(torch.from_numpy(g["cls_f16"]).cuda()).cpu().numpy()
    np.testing.assert_allclose(p_same, g["probs"], atol=1e-4)
    assert (p_same.argmax(1) == g["labels"]).all()
    enc.close(); head.close()


def test_e2e_long_clip_flip_rate_fp16(golden_dir):
    """The default fp16-operand mode on 2 048 frames of the headline model with 122 behaviour transitions
    (tests/golden/e2e_vitb16_long.npz, the reference's own wrapper + infer_file): the flip RATE and where the flips sit on
    the reference's top-2 margin scale.  Fixed gates: no flip at a reference margin >= conftest.MARGIN_FP16, at most 1 % of
    the labels differ, probabilities within 5e-2.  (precision 3 reproduces all 2 048: tests/test_gpu_fp32.py.)"""
    from cbas_amd.encoder import DinoEncoder
    from cbas_amd.head import ClassifierLSTMDeltas
    from cbas_amd.stream import ClipStream
    g = load(golden_dir, "e2e_vitb16_long")
    n = int(g["n"])
    cfg = C.VIT_B16
    fr = synth.cage_frames(int(g["frame_seed"]), n, 224, 224)
    enc = DinoEncoder.from_weights(cfg, W.synth_encoder_weights(cfg, 1234), "cuda", max_batch=64, max_frame=(224, 224))
    head = ClassifierLSTMDeltas(768, 9)
    head.load_state_dict(W.synth_head_weights(C.HeadConfig(in_features=768), 4321))
    head.to("cuda")
    try:
        st = ClipStream(enc, head, capacity=n)
        for i in range(0, n, 64):
            st.push_u8(torch.from_numpy(fr[i:i + 64]).cuda())
        cls16, probs = st.finish()
        torch.cuda.synchronize()
        probs = probs.cpu().numpy()
        r = rel_rows(cls16.float().cpu().numpy()[::8], g["cls_every8"])
        assert r.max() < CLS_TOL + 5e-4
        n_mis, n_below = assert_labels_match(probs, g["probs"], 5e-2)
        srt = np.sort(g["probs"].astype(np.float64), axis=1)
        m = srt[:, -1] - srt[:, -2]
        flips = np.nonzero(probs.argmax(1) != g["labels"])[0]
        edges = [0.0, 1e-3, 3e-3, 1e-2, 3e-2, 1e-1, 1.0]
        print(f"[e2e_vitb16_long fp16] {n_mis} of {n} labels differ ({n_mis / n * 100:.2f} %); reference frames per margin bin "
              f"{edges}: {np.histogram(m, edges)[0].tolist()}, flips per bin: {np.histogram(m[flips], edges)[0].tolist()}; "
              f"CLS rel err max {r.max():.3e}")
        assert n_mis <= n // 100
    finally:
        enc.close(); head.close()


def test_e2e_dinov2_default_encoder_flip_rate_fp16(golden_dir):
    """The default fp16-operand mode on CBAS's DEFAULT encoder family (DINOv2-with-registers ViT-B/14, 256 x 256, 512 frames;
    tests/golden/e2e_dinov2reg_b14.npz = the reference's own wrapper + infer_file): CLS within the 1e-3 contract, no flip at a
    reference margin >= conftest.MARGIN_FP16, at most 1 % of the labels differ."""
    from cbas_amd.encoder import DinoEncoder
    from cbas_amd.head import ClassifierLSTMDeltas
    from cbas_amd.stream import ClipStream
    g = load(golden_dir, "e2e_dinov2reg_b14")
    n, hw = int(g["n"]), int(g["hw"])
    cfg = C.DINOV2_REG_B14
    fr = synth.cage_frames(int(g["frame_seed"]), n, hw, hw)
    enc = DinoEncoder.from_weights(cfg, W.synth_encoder_weights(cfg, 1234), "cuda", max_batch=32, max_frame=(hw, hw))
    head = ClassifierLSTMDeltas(768, 9)
    head.load_state_dict(W.synth_head_weights(C.HeadConfig(in_features=768), 4321))
    head.to("cuda")
    try:
        st = ClipStream(enc, head, capacity=n)
        for i in range(0, n, 32):
            st.push_u8(torch.from_numpy(fr[i:i + 32]).cuda())
        cls16, probs = st.finish()
        torch.cuda.synchronize()
        probs = probs.cpu().numpy()
        r = rel_rows(cls16.float().cpu().numpy()[::8], g["cls_every8"])
        n_mis, _ = assert_labels_match(probs, g["probs"], 5e-2)
        print(f"[e2e_dinov2reg_b14 fp16] {n_mis} of {n} labels differ; CLS rel err max {r.max():.3e}")
        assert r.max() < CLS_TOL + 5e-4
        assert n_mis <= max(2, n // 100)
    finally:
        enc.close(); head.close()


def test_encode_file_and_infer_file_dropins(golden_dir, tmp_path):
    """The file-level drop-ins on a synthetic 'video': chunk loop with a ragged tail, progress
    callback values, .tmp + rename, h5 stamp, CSV name/header; CLS vs the reference's encode_file."""
    from cbas_amd import pipeline as P, h5io
    from cbas_amd.encoder import DinoEncoder
    from cbas_amd.head import ClassifierLSTMDeltas
    g = load(golden_dir, "encode_file_b1layer")
    cfg = C.ViTConfig(hidden_size=768, intermediate_size=1536, num_hidden_layers=1, num_attention_heads=12, image_size=32)
    ck = str(tmp_path / "ckpt")
    W.save_encoder_checkpoint(ck, cfg, W.synth_encoder_weights(cfg, 1234))
    enc = DinoEncoder(ck, device="cuda", max_batch=64, max_frame=(32, 32))        # reference constructor form
    assert enc.device.type == "cuda"
    frames = synth.cage_frames(5, 600, 32, 32)
    vid = str(tmp_path / "vid.npy")
    np.save(vid, frames)
    P.set_project_stamp("facebook/dinov3-vitb16-pretrain-lvd1689m")
    ticks = []
    out = P.encode_file(enc, vid, progress_callback=ticks.append)
    assert out == str(tmp_path / "vid_cls.h5") and os.path.exists(out) and not os.path.exists(out + ".tmp")
    np.testing.assert_allclose(ticks, g["ticks"])
    with h5io.ClsReader(out) as r:
        assert r.shape == (600, 768) and r.attrs["encoder_model_identifier"].startswith("facebook/dinov3")
        got = r.read(0, 600)
    rr = rel_rows(got.astype(np.float32), g["cls_f16"].astype(np.float32))
    assert rr.max() < CLS_TOL + 5e-4
    # zero-frame video -> None; reader failure -> raises and leaves no .tmp behind
    np.save(str(tmp_path / "empty.npy"), frames[:0])
    assert P.encode_file(enc, str(tmp_path / "empty.npy")) is None

    class Boom:
        def __len__(self):
            return 100

        def get_batch(self, idx):
            raise IOError("decode failed")
    with pytest.raises(IOError):
        P.encode_file(enc, str(tmp_path / "bad.npy"), reader=Boom())
    assert not os.path.exists(str(tmp_path / "bad_cls.h5.tmp"))
    # infer_file on the file just written
    head = ClassifierLSTMDeltas(768, 9)
    head.load_state_dict(W.synth_head_weights(C.HeadConfig(), 4321))
    csv = P.infer_file(out, head, "gold", NAMES, 31, device=torch.device("cuda"), temperature=1.0)
    assert csv == str(tmp_path / "vid_gold_outputs.csv")
    lines = open(csv).read().splitlines()
    assert lines[0] == ",".join(NAMES) and len(lines) == 601
    from oracle import pipeline_oracle as PO
    ref = PO.classify_cls(got, W.synth_head_weights(C.HeadConfig(), 4321), 31, 1.0)
    mine = np.array([[float(v) for v in ln.split(",")] for ln in lines[1:]], np.float32)
    np.testing.assert_allclose(mine, ref, atol=1e-4)
    assert (mine.argmax(1) == ref.argmax(1)).all()
    # infer_file never raises
    assert P.infer_file(str(tmp_path / "missing_cls.h5"), head, "gold", NAMES, 31, device="cuda") is None
    assert P.infer_file(out, head, "gold", NAMES[:3], 31, device="cuda") is None
    enc.close(); head.close()
    P.set_project_stamp(None)


def test_smoke_entry():
    import __graft_entry__ as G
    G.smoke()


def test_async_lanes_match_synchronous_forward():
    """cbas_enc_submit_u8 / wait_stream (two batches in flight on two compute lanes) give the synchronous
    result bit for bit, for ragged batch sizes, in any slot order, with 1 or 2 lanes; slot misuse is an error."""
    from cbas_amd import _lib as L
    from cbas_amd.encoder import DinoEncoder
    cfg = C.VIT_TINY
    enc = DinoEncoder.from_weights(cfg, W.synth_encoder_weights(cfg, 1234), "cuda", max_batch=16, max_frame=(64, 64))
    frames = torch.from_numpy(synth.cage_frames(11, 75, 64, 64)).cuda()
    ref, _ = enc.encode_u8(frames, want_f32=False)
    torch.cuda.synchronize()
    sizes = [16, 5, 16, 1, 16, 16, 5]                     # 75 frames in ragged batches
    for lanes in (2, 1, 2):
        enc.set_lanes(lanes)
        out = torch.zeros_like(ref)
        busy, o = [], 0
        for i, n in enumerate(sizes):
            slot = i % L.ENC_SLOTS
            if slot in busy:
                enc.wait_stream(slot)
                busy.remove(slot)
            enc.submit_dev(slot, frames[o:o + n], out[o:o + n])
            busy.append(slot)
            o += n
        with pytest.raises(RuntimeError, match="busy"):
            enc.submit_dev(busy[-1], frames[:4], out[:4])
        with pytest.raises(RuntimeError, match="slot"):       # host-style wait on a device submission
            enc.wait(busy[-1])
        with pytest.raises(RuntimeError, match="in flight"):
            enc.set_lanes(1)
        for slot in busy:
            enc.wait_stream(slot)
        torch.cuda.synchronize()
        assert torch.equal(out, ref), lanes
        with pytest.raises(RuntimeError, match="no submitted work"):
            enc.wait_stream(0)
    enc.close()


def test_mixing_sync_and_async_calls_on_one_handle():
    """A synchronous forward issued while asynchronous batches are queued on lane 0 (and the reverse) is ordered
    by the library: both forms use lane 0's workspace, neither corrupts the other."""
    from cbas_amd.encoder import DinoEncoder
    cfg = C.VIT_TINY
    enc = DinoEncoder.from_weights(cfg, W.synth_encoder_weights(cfg, 1234), "cuda", max_batch=16, max_frame=(64, 64))
    frames = torch.from_numpy(synth.cage_frames(13, 64, 64, 64)).cuda()
    ref, _ = enc.encode_u8(frames, want_f32=False)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    for rep in range(6):
        out = torch.zeros_like(ref)
        enc.submit_dev(0, frames[0:16], out[0:16])            # lane 0
        enc.submit_dev(1, frames[16:32], out[16:32])          # lane 1
        with torch.cuda.stream(side):                         # synchronous forward on an unrelated stream
            mid, _ = enc.encode_u8(frames[32:48], want_f32=False)
        enc.submit_dev(2, frames[48:64], out[48:64])          # lane 0 again, after the synchronous call
        for s in (0, 1, 2):
            enc.wait_stream(s)
        torch.cuda.synchronize()
        assert torch.equal(out[0:32], ref[0:32]) and torch.equal(out[48:64], ref[48:64]) and torch.equal(mid, ref[32:48]), rep
    enc.close()


def test_massive_activation_channels_stay_within_tolerance():
    """Real ViT checkpoints carry a few residual-stream channels hundreds of times larger than the rest
    ("massive activations", large register tokens).  The fp32 residual stream + fp16 operands must keep the
    1e-3 CLS bar there too: synthetic ViT-S with +-80 offsets injected into three channels at two depths,
    20x register tokens and 5x LayerNorm gains on those channels, against the fp32 oracle."""
    from cbas_amd.encoder import DinoEncoder
    from oracle import pipeline_oracle as PO
    cfg = C.NAMED_VIT["vits16"]
    w = {k: v.copy() for k, v in W.synth_encoder_weights(cfg, 1234).items()}
    for l, sign, hot in ((2, 1.0, [7, 100, 300]), (5, -1.0, [11, 200, 350])):
        w[f"model.layer.{l}.mlp.down_proj.bias"][hot] += sign * 80.0 / np.abs(w[f"model.layer.{l}.layer_scale2.lambda1"][hot])
        for n in ("norm1", "norm2"):
            w[f"model.layer.{l + 1}.{n}.weight"][hot] *= 5.0
    w["embeddings.register_tokens"] = w["embeddings.register_tokens"] * 20.0
    fr = synth.cage_frames(5, 4, 224, 224)
    ref = PO.encode_frames(fr, w, cfg, batch=4)
    enc = DinoEncoder.from_weights(cfg, w, "cuda", max_batch=4, max_frame=(224, 224))
    _, c32 = enc.encode_u8(torch.from_numpy(fr).cuda())
    tap = enc.debug_tap(torch.from_numpy(fr).cuda(), 6, 7, 0)          # residual stream after block 6
    enc.close()
    assert np.abs(tap).max() > 60.0                                    # the outliers are really there
    r = rel_rows(c32.cpu().numpy(), ref)
    assert r.max() < CLS_TOL, r.max()


def test_encoder_workspace_grows_for_larger_frames(golden_dir):
    """DinoEncoder takes any frame size, like the reference: a frame larger than max_frame rebuilds the handle."""
    from cbas_amd.encoder import DinoEncoder
    from oracle import pipeline_oracle as PO
    cfg = C.VIT_TINY
    w = W.synth_encoder_weights(cfg, 1234)
    enc = DinoEncoder.from_weights(cfg, w, "cuda", max_batch=4, max_frame=(64, 64))
    small = synth.cage_frames(3, 2, 64, 64)
    big = synth.cage_frames(3, 2, 96, 128)
    a16, a32 = enc.encode_u8(torch.from_numpy(small).cuda())
    b16, b32 = enc.encode_u8(torch.from_numpy(big).cuda())                 # 96x128 > 64x64: rebuild
    assert enc.max_frame == (96, 128)
    c16, c32 = enc.encode_u8(torch.from_numpy(small).cuda())               # small frames still work, same result
    assert torch.equal(a32, c32)
    ref = PO.encode_frames(big, w, cfg, batch=2)
    assert rel_rows(b32.cpu().numpy(), ref).max() < CLS_TOL
    enc.submit_host(0, big)
    assert np.array_equal(enc.wait(0, want_f32=True)[1], b32.cpu().numpy())
    enc.close()

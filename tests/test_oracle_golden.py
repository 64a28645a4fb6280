"""The oracle (CPU restatement) against the golden vectors produced by running the reference's own
modules (tests/golden/make_goldens.py).  This is what pins the oracle."""
import hashlib
import os

import numpy as np
import pytest

from cbas_amd import config as C, weights as W, synth
from oracle import head_oracle as H
from oracle import pipeline_oracle as PO
from oracle import vit_oracle as V


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)


def classify_rows(cls, head_w, seq_len, temp, sel):
    """infer_file's probabilities for the frames `sel` only (each frame's window is independent of the others; the numpy
    BiLSTM takes ~60 ms per window, so the long clips are checked on their edges, seams and a stride through the middle)."""
    idx = H.infer_windows(cls, seq_len)[sel]
    logits, _ = H.head_forward(cls.astype(np.float32)[idx], head_w, seq_len)
    return H.softmax_T(logits, temp)


def test_vit_tiny_stagewise(golden_dir):
    g = load(golden_dir, "vit_tiny")
    cfg = C.VIT_TINY
    w = W.synth_encoder_weights(cfg, 1234)
    fr = synth.cage_frames(7, 4, 64, 64)
    assert sha(fr) == str(g["frames_sha"])
    taps = {}
    out = V.vit_forward(np.repeat(V.preprocess_green(fr)[:, None], 3, 1), w, cfg, taps)
    np.testing.assert_allclose(taps["embeddings"], g["embeddings"], atol=1e-5)
    np.testing.assert_allclose(taps["l0.out"], g["layer0"], atol=1e-5)
    np.testing.assert_allclose(taps["l1.out"], g["layer1"], atol=1e-5)
    np.testing.assert_allclose(out, g["last_hidden"], atol=2e-5)


@pytest.mark.parametrize("name,cfgname,hw", [("vits16_224", "vits16", 224), ("vitb16_224", "vitb16", 224),
                                             ("vitb16_224_noise", "vitb16", 224), ("vitb16_256", "vitb16", 256)])
def test_vit_cls_goldens(golden_dir, name, cfgname, hw):
    g = load(golden_dir, name)
    cfg = C.NAMED_VIT[cfgname]
    w = W.synth_encoder_weights(cfg, 1234)
    n = int(g["n"])
    mk = synth.noise_frames if str(g["kind"]) == "noise" else synth.cage_frames
    fr = mk(int(g["frame_seed"]), n, hw, hw)
    assert sha(fr) == str(g["frames_sha"])
    # the numpy restatement on the first frames (ViT-B: ~4 s per frame), the torch restatement of the same arithmetic on
    # all of them: the CPU suite has to stay within a few minutes
    k = min(n, 2 if cfgname == "vitb16" else 4)
    cls = PO.encode_frames(fr[:k], w, cfg, batch=4)
    rel = np.linalg.norm(cls - g["cls"][:k], axis=1) / np.linalg.norm(g["cls"][:k], axis=1)
    assert rel.max() < 1e-5, rel.max()
    from oracle import vit_oracle_torch as VT
    cls_t = VT.encode_frames(fr, VT.to_torch(w), cfg, batch=4)
    rel_t = np.linalg.norm(cls_t - g["cls"], axis=1) / np.linalg.norm(g["cls"], axis=1)
    assert rel_t.max() < 1e-5, rel_t.max()


@pytest.mark.slow
@pytest.mark.parametrize("name,hw", [("vitl16_224", 224), ("vitl16_518", 518)])
def test_vitl_cls_goldens(golden_dir, name, hw):
    g = load(golden_dir, name)
    cfg = C.VIT_L16
    w = W.synth_encoder_weights(cfg, 1234)
    n = int(g["n"])
    fr = synth.cage_frames(int(g["frame_seed"]), n, hw, hw)
    assert sha(fr) == str(g["frames_sha"])
    if hw <= 256:                                  # the numpy restatement on one small ViT-L frame; 518x518 through torch only
        cls = PO.encode_frames(fr[:1], w, cfg, batch=1)
        rel = np.linalg.norm(cls - g["cls"][:1], axis=1) / np.linalg.norm(g["cls"][:1], axis=1)
        assert rel.max() < 2e-5, rel.max()
    # every golden frame through the torch restatement of the same arithmetic (oracle/vit_oracle_torch.py)
    from oracle import vit_oracle_torch as VT
    cls_t = VT.encode_frames(fr, VT.to_torch(w), cfg, batch=2)
    rel_t = np.linalg.norm(cls_t - g["cls"], axis=1) / np.linalg.norm(g["cls"], axis=1)
    assert rel_t.max() < 2e-5, rel_t.max()


def test_rope_table_matches_reference_shape():
    cos, sin = V.rope_cos_sin(14, 14, 64, 100.0)
    assert cos.shape == (196, 64) and sin.shape == (196, 64)
    np.testing.assert_array_equal(cos[:, :32], cos[:, 32:])      # angles.tile(2)
    assert abs(float(cos[0, 0]) - np.cos(2 * np.pi * (2 * (0.5 / 14) - 1))) < 1e-6


@pytest.mark.parametrize("tag,h,ncls,dim,nl", [("h64", 64, 9, 768, 1), ("h128", 128, 5, 768, 1),
                                               ("h64_d384", 64, 9, 384, 1), ("h64_l2", 64, 9, 768, 2)])
def test_head_goldens(golden_dir, tag, h, ncls, dim, nl):
    g = load(golden_dir, f"head_{tag}")
    hc = C.HeadConfig(in_features=dim, out_features=ncls, lstm_hidden_size=h, lstm_layers=nl)
    hw = W.synth_head_weights(hc, 4321)
    seq = synth.cls_walk(21, 94, dim).astype(np.float32)
    x = np.stack([seq[i:i + 31] for i in range(64)])
    assert sha(x) == str(g["x_sha"])
    logits, latent = H.head_forward(x, hw)
    np.testing.assert_allclose(logits, g["logits"], atol=2e-5)
    np.testing.assert_allclose(latent, g["latent"], atol=1e-5)
    assert (logits.argmax(1) == g["logits"].argmax(1)).all()


@pytest.mark.parametrize("n", [1, 10, 31, 64, 700, 40])
def test_infer_file_goldens(golden_dir, n):
    g = load(golden_dir, "infer_file")
    hw = W.synth_head_weights(C.HeadConfig(), 4321)
    cls = synth.cls_walk(100 + n, n, 768)
    assert sha(cls) == str(g[f"cls_sha_{n}"])
    temp = float(g[f"temp_{n}"])
    sel = np.arange(n) if n <= 64 else np.unique(np.r_[0:24, n - 24:n, 24:n - 24:9])
    probs = classify_rows(cls, hw, 31, temp, sel)
    np.testing.assert_allclose(probs, g[f"probs_{n}"][sel], atol=2e-6)
    assert (probs.argmax(1) == g[f"probs_{n}"][sel].argmax(1)).all()


def test_infer_file_literal_loop_equals_clamped_windows():
    """The reference's chunk/halo/pad loop (restated literally) == clamped-index windows."""
    hw = W.synth_head_weights(C.HeadConfig(), 4321)
    cls = synth.cls_walk(3, 75, 768)
    a = H.infer_file_literal(cls, hw, 31, 0.9, chunk=32, batch=16)      # forces 3 chunks with halos
    b = PO.classify_cls(cls, hw, 31, 0.9)
    np.testing.assert_allclose(a, b, atol=1e-7)


@pytest.mark.slow
def test_infer_file_halo_boundary_golden(golden_dir):
    g = load(golden_dir, "infer_file")
    n = 20017
    hw = W.synth_head_weights(C.HeadConfig(), 4321)
    cls = synth.cls_walk(100 + n, n, 768)
    assert sha(cls) == str(g[f"cls_sha_{n}"])
    sel = np.r_[0:40, 19960:20017]                   # both clip edges and the 20 000-frame chunk seam
    idx = H.infer_windows(cls, 31)[sel]
    logits, _ = H.head_forward(cls.astype(np.float32)[idx], hw, 31)
    probs = H.softmax_T(logits, float(g[f"temp_{n}"]))
    np.testing.assert_allclose(probs, g[f"probs_{n}"][sel], atol=2e-6)


def test_e2e_config1_golden(golden_dir):
    """BASELINE config 1 (ViT-S/16, 64 frames, batch 8, C=9): CLS, fp16 rows, probabilities, labels."""
    g = load(golden_dir, "e2e_vits16")
    cfg = C.VIT_S16
    fr = synth.cage_frames(3, 64, 224, 224)
    assert sha(fr) == str(g["frames_sha"])
    enc_w = W.synth_encoder_weights(cfg, 1234)
    head_w = W.synth_head_weights(C.HeadConfig(in_features=384), 4321)
    # the numpy restatement on the first 4 frames, the torch restatement of the same arithmetic on all 64 frames
    # (numpy ViT-S: ~2 s per frame)
    first = PO.encode_frames(fr[:4], enc_w, cfg, batch=4)
    rel16 = np.linalg.norm(first - g["cls"][:4], axis=1) / np.linalg.norm(g["cls"][:4], axis=1)
    assert rel16.max() < 1e-5
    from oracle import vit_oracle_torch as VT
    cls32 = VT.encode_frames(fr, VT.to_torch(enc_w), cfg, batch=8)
    cls16 = cls32.astype(np.float16)
    probs = PO.classify_cls(cls16, head_w, 31, 1.0)
    rel = np.linalg.norm(cls32 - g["cls"], axis=1) / np.linalg.norm(g["cls"], axis=1)
    assert rel.max() < 1e-5
    # fp16 rounding can differ by one ulp where the fp32 values straddle a tie; the head is run on
    # the golden fp16 rows to pin the probabilities exactly
    # (fp32 differences of ~1e-6 flip ~0.5 % of the roundings, never by more than one fp16 ulp)
    assert (cls16 != g["cls_f16"]).mean() < 2e-2
    a, b = cls16.astype(np.float32), g["cls_f16"].astype(np.float32)
    assert np.all(np.abs(a - b) <= np.abs(b) * 2.0 ** -10 + 4e-6)   # one ulp, or the fp32 noise near zero
    probs_g = PO.classify_cls(g["cls_f16"], head_w, 31, 1.0)
    np.testing.assert_allclose(probs_g, g["probs"], atol=1e-5)
    assert (probs_g.argmax(1) == g["labels"]).all()
    assert (probs.argmax(1) == g["labels"]).all()


def test_e2e_config2_model_golden(golden_dir):
    """The headline model through the reference's own DinoEncoder wrapper + infer_file (ViT-B/16, 256 frames, C = 9): the
    torch restatement on the first 24 frames, the head restatement on all the golden fp16 rows."""
    g = load(golden_dir, "e2e_vitb16")
    n = int(g["n"])
    fr = synth.cage_frames(int(g["frame_seed"]), n, 224, 224)
    assert sha(fr) == str(g["frames_sha"])
    from oracle import vit_oracle_torch as VT
    cfg = C.VIT_B16
    cls32 = VT.encode_frames(fr[:24], VT.to_torch(W.synth_encoder_weights(cfg, 1234)), cfg, batch=8)
    rel = np.linalg.norm(cls32 - g["cls"][:24], axis=1) / np.linalg.norm(g["cls"][:24], axis=1)
    assert rel.max() < 1e-5
    assert np.array_equal(g["cls"].astype(np.float16), g["cls_f16"])                  # h5py's f4 -> f2 cast (cbas.py:438)
    probs = PO.classify_cls(g["cls_f16"], W.synth_head_weights(C.HeadConfig(in_features=768), 4321), 31, 1.0)
    np.testing.assert_allclose(probs, g["probs"], atol=1e-5)
    assert (probs.argmax(1) == g["labels"]).all() and len(set(g["labels"].tolist())) >= 2


@pytest.mark.parametrize("name,cfgname,hw", [("vitb16_224_noise", "vitb16", 224), ("vitb16_256", "vitb16", 256)])
def test_torch_restatement_against_reference_goldens(golden_dir, name, cfgname, hw):
    """oracle/vit_oracle_torch.py (what bench.py times as cpu_baseline) against CLS rows made by the reference."""
    from oracle import vit_oracle_torch as VT
    g = load(golden_dir, name)
    cfg = C.NAMED_VIT[cfgname]
    w = VT.to_torch(W.synth_encoder_weights(cfg, 1234))
    n = int(g["n"])
    mk = synth.noise_frames if str(g["kind"]) == "noise" else synth.cage_frames
    fr = mk(int(g["frame_seed"]), n, hw, hw)
    cls = VT.encode_frames(fr, w, cfg, batch=4)
    rel = np.linalg.norm(cls - g["cls"], axis=1) / np.linalg.norm(g["cls"], axis=1)
    assert rel.max() < 1e-5, rel.max()


HEAD_VARIANTS = [("h64_t63", dict(seq_len=63)), ("h64_t95", dict(seq_len=95)), ("h32", dict(lstm_hidden_size=32)),
                 ("h96_t63", dict(lstm_hidden_size=96, seq_len=63)), ("h64_noacc", dict(use_acceleration=False)),
                 ("h48_noacc_l2_t15", dict(use_acceleration=False, lstm_hidden_size=48, lstm_layers=2, seq_len=15))]


@pytest.mark.parametrize("tag,kw", HEAD_VARIANTS, ids=[t for t, _ in HEAD_VARIANTS])
def test_head_variant_goldens(golden_dir, tag, kw):
    """Sequence lengths 63 / 95 (sweep_runner.py:110), other hidden sizes, use_acceleration=False: oracle vs reference."""
    g = load(golden_dir, f"head_{tag}")
    hc = C.HeadConfig(in_features=768, out_features=9, **kw)
    hw = W.synth_head_weights(hc, 4321)
    T = hc.seq_len
    seq = synth.cls_walk(21, 48 + T - 1, 768).astype(np.float32)
    x = np.stack([seq[i:i + T] for i in range(48)])
    assert sha(x) == str(g["x_sha"])
    logits, latent = H.head_forward(x, hw, seq_len=T)
    np.testing.assert_allclose(logits, g["logits"], atol=3e-5)
    np.testing.assert_allclose(latent, g["latent"], atol=2e-5)
    assert (logits.argmax(1) == g["logits"].argmax(1)).all()


@pytest.mark.parametrize("n,T", [(40, 63), (300, 63), (260, 95)])
def test_infer_file_goldens_long_windows(golden_dir, n, T):
    g = load(golden_dir, "infer_file_seq")
    hw = W.synth_head_weights(C.HeadConfig(seq_len=T), 4321)
    cls = synth.cls_walk(500 + n + T, n, 768)
    assert sha(cls) == str(g[f"cls_sha_{n}_{T}"])
    sel = np.arange(n) if n <= 64 else np.unique(np.r_[0:T // 2 + 4, n - T // 2 - 4:n, T // 2 + 4:n - T // 2 - 4:11])
    probs = classify_rows(cls, hw, T, float(g[f"temp_{n}_{T}"]), sel)
    np.testing.assert_allclose(probs, g[f"probs_{n}_{T}"][sel], atol=3e-6)
    assert (probs.argmax(1) == g[f"probs_{n}_{T}"][sel].argmax(1)).all()

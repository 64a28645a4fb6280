"""DINOv2-with-registers, CBAS's default encoder family (reference backend/cbas.py:1030-1033,
SURVEY.md §8(f) row 3): patch 14, learned position embedding interpolated bicubic-antialias to the
frame's patch grid, key bias, no RoPE.  Goldens come from HF ``Dinov2WithRegistersModel`` /
the reference's ``DinoEncoder`` wrapper (tests/golden/make_goldens.py::g_dinov2)."""
import hashlib
import os

import numpy as np
import pytest

from cbas_amd import config as C, weights as W, synth
from oracle import dinov2_oracle as O
from oracle import vit_oracle as V


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def rel_rows(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return np.linalg.norm(a - b, axis=-1) / np.linalg.norm(b, axis=-1)


def test_aa_bicubic_matrix_properties():
    for gin, gout in ((37, 16), (37, 18), (5, 4), (5, 6), (5, 5)):
        Wm = O.aa_bicubic_matrix(gin, gout)
        assert Wm.shape == (gout, gin)
        np.testing.assert_allclose(Wm.sum(1), 1.0, atol=1e-6)          # row-stochastic
        np.testing.assert_allclose(Wm, Wm[::-1, ::-1], atol=1e-6)       # symmetric under reflection
    np.testing.assert_allclose(O.aa_bicubic_matrix(5, 5), np.eye(5), atol=1e-6)


@pytest.mark.parametrize("hw", [70, 56, 84])
def test_oracle_tiny_goldens(golden_dir, hw):
    g = np.load(os.path.join(golden_dir, "dinov2reg_tiny.npz"))
    cfg = C.DINOV2_REG_TINY
    w = W.canonical_encoder_weights(cfg, W.synth_encoder_weights(cfg, 1234))
    fr = synth.cage_frames(20 + hw, 3, hw, hw)
    assert sha(fr) == str(g[f"sha_{hw}"])
    px = np.repeat(V.preprocess_green(fr)[:, None], 3, 1)
    taps = {}
    out = O.forward(px, w, cfg, taps)
    np.testing.assert_allclose(taps["embeddings"], g[f"emb_{hw}"], atol=2e-5)    # incl. interpolated positions
    np.testing.assert_allclose(out, g[f"last_{hw}"], atol=3e-5)


def test_oracle_b14_goldens(golden_dir):
    g = np.load(os.path.join(golden_dir, "dinov2reg_b14.npz"))
    cfg = C.DINOV2_REG_B14
    w = W.canonical_encoder_weights(cfg, W.synth_encoder_weights(cfg, 1234))
    for hw, seed, n in ((224, 31, 4), (256, 32, 2)):
        fr = synth.cage_frames(seed, n, hw, hw)
        assert sha(fr) == str(g[f"sha{hw}"])
        px = np.repeat(V.preprocess_green(fr)[:, None], 3, 1)
        cls = np.concatenate([O.forward(px[i:i + 2], w, cfg)[:, 0] for i in range(0, n, 2)])
        assert rel_rows(cls, g[f"cls{hw}"]).max() < 2e-5


def test_config_json_roundtrip_matches_hf_fields(tmp_path):
    cfg = C.DINOV2_REG_B14
    W.save_encoder_checkpoint(str(tmp_path / "ck"), C.DINOV2_REG_TINY, W.synth_encoder_weights(C.DINOV2_REG_TINY, 3))
    cfg2, w2 = W.load_encoder_checkpoint(str(tmp_path / "ck"))
    assert cfg2 == C.DINOV2_REG_TINY and "embeddings.position_embeddings" in w2
    assert cfg.pos_embed_grid == 37 and not cfg.use_rope and cfg.key_bias


@pytest.mark.gpu
def test_gpu_tiny_embeddings_and_cls(golden_dir):
    import torch
    from cbas_amd.encoder import DinoEncoder
    g = np.load(os.path.join(golden_dir, "dinov2reg_tiny.npz"))
    cfg = C.DINOV2_REG_TINY
    enc = DinoEncoder.from_weights(cfg, W.synth_encoder_weights(cfg, 1234), "cuda", max_batch=4, max_frame=(84, 84))
    try:
        for hw in (70, 56, 84, 70):                  # also exercises the per-resolution table switch
            fr = synth.cage_frames(20 + hw, 3, hw, hw)
            fd = torch.from_numpy(fr).cuda()
            emb = enc.debug_tap(fd, 0, 0, 0).reshape(3, -1, cfg.hidden_size)
            assert np.abs(emb - g[f"emb_{hw}"]).max() < 2e-3
            _, c32 = enc.encode_u8(fd)
            torch.cuda.synchronize()
            assert rel_rows(c32.cpu().numpy(), g[f"last_{hw}"][:, 0]).max() < 1e-3
            x = torch.from_numpy(V.preprocess_green(fr)).cuda().unsqueeze(1)
            assert rel_rows(enc(x).squeeze(1).cpu().numpy(), g[f"last_{hw}"][:, 0]).max() < 1e-3
    finally:
        enc.close()


@pytest.mark.gpu
def test_gpu_b14_goldens(golden_dir, tmp_path):
    """ViT-B/14 at 224 and at CBAS's standard 256x256 video size (18x18 patches, T = 329: streaming
    attention), constructed the way CBAS does: DinoEncoder(<checkpoint dir>, device)."""
    import torch
    from cbas_amd.encoder import DinoEncoder
    g = np.load(os.path.join(golden_dir, "dinov2reg_b14.npz"))
    cfg = C.DINOV2_REG_B14
    ck = str(tmp_path / "dinov2reg")
    W.save_encoder_checkpoint(ck, cfg, W.synth_encoder_weights(cfg, 1234))
    enc = DinoEncoder(ck, device="cuda", max_batch=4, max_frame=(256, 256))
    try:
        assert enc.config.model_type == "dinov2_with_registers"
        for hw, seed, n in ((224, 31, 4), (256, 32, 2)):
            fr = synth.cage_frames(seed, n, hw, hw)
            _, c32 = enc.encode_u8(torch.from_numpy(fr).cuda())
            torch.cuda.synchronize()
            r = rel_rows(c32.cpu().numpy(), g[f"cls{hw}"])
            assert r.max() < 1e-3, (hw, r.max())
    finally:
        enc.close()


def test_oracle_e2e_default_encoder_fixture(golden_dir):
    """The end-to-end fixture of CBAS's default encoder family (tests/golden/e2e_dinov2reg_b14.npz: the reference's own
    DinoEncoder wrapper on DINOv2-with-registers B/14 at 256 x 256 + its infer_file): the numpy restatement reproduces the
    f32 CLS of sampled frames, and the head restatement reproduces every label from the reference's own f16 rows."""
    from oracle import pipeline_oracle as PO
    g = np.load(os.path.join(golden_dir, "e2e_dinov2reg_b14.npz"))
    cfg = C.DINOV2_REG_B14
    n, hw = int(g["n"]), int(g["hw"])
    w = W.canonical_encoder_weights(cfg, W.synth_encoder_weights(cfg, 1234))
    pick = [0, 8 * 31]
    fr = np.concatenate([synth.cage_frames(int(g["frame_seed"]), 1, hw, hw, first=i) for i in pick])
    px = np.repeat(V.preprocess_green(fr)[:, None], 3, 1)
    cls = O.forward(px, w, cfg)[:, 0]
    assert rel_rows(cls, g["cls_every8"][[p // 8 for p in pick]]).max() < 2e-5
    assert np.array_equal(g["cls_every8"].astype(np.float16), g["cls_f16"][::8])          # the f4 -> f2 store, cbas.py:438
    probs = PO.classify_cls(g["cls_f16"], W.synth_head_weights(C.HeadConfig(in_features=768), 4321), 31, 1.0)
    np.testing.assert_allclose(probs, g["probs"], atol=1e-5)
    assert (probs.argmax(1) == g["labels"]).all() and len(set(g["labels"].tolist())) >= 2

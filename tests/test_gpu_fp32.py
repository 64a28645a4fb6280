"""precision 3 and 4: the encoder in the reference's own CPU arithmetic (cbas_amd/csrc/vit_f32.hip) - fp32 storage,
attention, LayerNorm and element-wise steps in both; the GEMMs on v_mfma_f32_16x16x4_f32 (3: exact fp32 products and
sums) or on v_mfma_f32_16x16x32_f16 from operands split into two fp16 halves (4: three-term products, 22 bits, fp32
accumulation - as close to the reference, twice as fast).  These are the modes that meet BASELINE.json's "identical
argmax labels" literally: the gates below have NO near-tie relaxation - every label of the reference's own end-to-end
fixtures must be reproduced, CLS rows within a few 1e-6 of the reference's fp32 CPU output."""
import os

import numpy as np
import pytest
import torch

from cbas_amd import config as C, weights as W, synth

pytestmark = pytest.mark.gpu

CLS_TOL_F32 = 5e-6          # per-frame ||d||2 / ||ref||2 against the reference's fp32 CPU rows


def rel_rows(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return np.linalg.norm(a - b, axis=1) / np.linalg.norm(b, axis=1)


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


PRECISIONS = (3, 4)


def make_enc(cfg, hw, max_batch, precision=3):
    from cbas_amd.encoder import DinoEncoder
    return DinoEncoder.from_weights(cfg, W.synth_encoder_weights(cfg, 1234), "cuda", max_batch=max_batch, max_frame=(hw, hw),
                                    precision=precision)


def make_head(dim):
    from cbas_amd.head import ClassifierLSTMDeltas
    head = ClassifierLSTMDeltas(dim, 9)
    head.load_state_dict(W.synth_head_weights(C.HeadConfig(in_features=dim), 4321))
    head.to("cuda")
    return head


def test_fp32_tiny_stagewise_against_oracle():
    """Every fp32 kernel in isolation against the numpy restatement of [tf]: ingest + patch GEMM, LayerNorm, QKV + RoPE,
    attention, o_proj + LayerScale + residual, up_proj + GELU, down_proj + LayerScale + residual - to fp32 rounding."""
    from oracle import vit_oracle as V
    cfg = C.VIT_TINY
    w = W.synth_encoder_weights(cfg, 1234)
    enc = make_enc(cfg, 64, 8)
    try:
        fr = synth.cage_frames(7, 4, 64, 64)
        taps = {}
        V.vit_forward(np.repeat(V.preprocess_green(fr)[:, None], 3, 1), w, cfg, taps)
        fd = torch.from_numpy(fr).cuda()
        D = cfg.hidden_size

        def close(got, want, tol=2e-6):
            got, want = got.astype(np.float64), want.reshape(-1, want.shape[-1]).astype(np.float64)
            err = np.linalg.norm(got - want) / np.linalg.norm(want)
            assert err < tol, err

        close(enc.debug_tap(fd, 0, 0, 0), taps["embeddings"])
        for l in range(cfg.num_hidden_layers):
            close(enc.debug_tap(fd, l, 1, 1), taps[f"l{l}.ln1"])
            qkv = enc.debug_tap(fd, l, 2, 2)
            assert qkv.dtype == np.float32
            close(qkv[:, :D] * 8.0, taps[f"l{l}.q_rope"])             # q is stored pre-scaled by 1/8 (exact)
            close(qkv[:, D:2 * D], taps[f"l{l}.k_rope"])
            close(qkv[:, 2 * D:], taps[f"l{l}.v"])
            close(enc.debug_tap(fd, l, 3, 1), taps[f"l{l}.ctx"])
            close(enc.debug_tap(fd, l, 4, 0), taps[f"l{l}.after_attn"])
            close(enc.debug_tap(fd, l, 5, 1), taps[f"l{l}.ln2"])
            close(enc.debug_tap(fd, l, 6, 3), taps[f"l{l}.up"])
            close(enc.debug_tap(fd, l, 7, 0), taps[f"l{l}.out"])
    finally:
        enc.close()


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("name,cfgname,hw", [("vits16_224", "vits16", 224), ("vitb16_224", "vitb16", 224),
                                             ("vitb16_224_noise", "vitb16", 224), ("vitb16_256", "vitb16", 256),
                                             ("vitl16_224", "vitl16", 224), ("vitl16_518", "vitl16", 518)])
def test_fp32_cls_goldens(golden_dir, name, cfgname, hw, precision):
    """CLS rows of the reference's own fp32 forward (HF DINOv3ViTModel on CPU), every committed model / size, incl. the
    DINOv2-free T = 1029 case that streams 17 key blocks through the attention kernel."""
    g = load(golden_dir, name)
    cfg = C.NAMED_VIT[cfgname]
    n = int(g["n"])
    mk = synth.noise_frames if str(g["kind"]) == "noise" else synth.cage_frames
    fr = mk(int(g["frame_seed"]), n, hw, hw)
    enc = make_enc(cfg, hw, 8, precision)
    try:
        c16, c32 = enc.encode_u8(torch.from_numpy(fr).cuda())
        torch.cuda.synchronize()
        r = rel_rows(c32.cpu().numpy(), g["cls"])
        print(f"[precision {precision} {name}] CLS rel err max {r.max():.3e}")
        assert r.max() < CLS_TOL_F32, r.max()
        assert np.array_equal(c16.cpu().numpy(), c32.cpu().numpy().astype(np.float16))
        # pruned last layer == full last layer, bit for bit (rows are independent, same k order)
        enc.set_prune_last_layer(False)
        _, f32 = enc.encode_u8(torch.from_numpy(fr).cuda())
        torch.cuda.synchronize()
        assert torch.equal(f32, c32)
    finally:
        enc.close()


@pytest.mark.parametrize("precision", PRECISIONS)
def test_fp32_float_input_is_the_same_arithmetic(golden_dir, precision):
    """DinoEncoder.__call__ on the reference's float tensor (green / 255.0, cbas.py:431-435) == the uint8 path bit for bit:
    the ingest kernel forms float(double(pixel) / 255.0), the value numpy hands the reference."""
    from oracle import vit_oracle as V
    g = load(golden_dir, "vits16_224")
    cfg = C.VIT_S16
    fr = synth.cage_frames(int(g["frame_seed"]), int(g["n"]), 224, 224)
    enc = make_enc(cfg, 224, 8, precision)
    try:
        _, c32 = enc.encode_u8(torch.from_numpy(fr).cuda())
        x = torch.from_numpy((fr[:, :, :, 1] / 255.0).astype(np.float32)).cuda().unsqueeze(1)
        y = enc(x).squeeze(1)
        torch.cuda.synchronize()
        assert torch.equal(y, c32)
        assert np.array_equal(V.preprocess_green(fr), (fr[:, :, :, 1] / 255.0).astype(np.float32))
    finally:
        enc.close()


@pytest.mark.parametrize("precision", PRECISIONS)
def test_fp32_batch_invariance_at_the_bench_batch(precision):
    """64 x 224^2 (M = 12 864 rows): a frame's row does not depend on its batch size or position (bit-exact)."""
    cfg = C.VIT_B16
    enc = make_enc(cfg, 224, 64, precision)
    try:
        fr = synth.noise_frames(5, 8, 224, 224)
        big = np.concatenate([fr] * 8)
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


def test_precision4_gemm_forms_are_bit_identical():
    """Precision 4 has three GEMM kernels (the ping-pong kernel's split-operand form for M > 256, the ring-buffered skinny
    form for the pruned last layer's CLS rows, the 128 x 128 kernels for everything else): every combination gives the
    same rows, bit for bit, on ViT-B (N multiples of 256) with RoPE, and on the DINOv2 geometry (no RoPE, position table)."""
    for cfg, hw, seed in ((C.VIT_B16, 224, 7), (C.DINOV2_REG_B14, 224, 8)):
        enc = make_enc(cfg, hw, 16, 4)
        try:
            fr = torch.from_numpy(synth.noise_frames(seed, 16, hw, hw)).cuda()
            rows = []
            for forms in (3, 2, 1, 0):
                enc.debug_option("split_kernels", forms)
                c16, c32 = enc.encode_u8(fr)
                torch.cuda.synchronize()
                rows.append((c16.clone(), c32.clone()))
            for c16, c32 in rows[1:]:
                assert torch.equal(c16, rows[0][0]) and torch.equal(c32, rows[0][1])
            assert torch.isfinite(rows[0][1]).all()
        finally:
            enc.debug_option("split_kernels", -1)
            enc.close()


def test_precision4_gemm_forms_shape_sweep():
    """The split-operand GEMM alone, ping-pong / skinny forms against the 128 x 128 kernels on the same random operands:
    every epilogue (q|k|v with RoPE and head-split output, residual, GELU with split output), every tile height, row counts
    around the tile and the M <= 256 boundaries, the ViT-B and ViT-L projection shapes - 0 differing output words."""
    import ctypes
    from cbas_amd import _lib
    lib = _lib.load()
    n = ctypes.c_int64()
    shapes = [(768, 768, 2), (2304, 768, 1), (3072, 768, 3), (768, 3072, 2), (1024, 4096, 2), (4096, 1024, 3), (3072, 1024, 1)]
    for M in (64, 201, 256, 257, 300, 511, 1000, 3216, 12864, 12865):
        for N, K, epi in shapes:
            tiles = (0,) if M <= 256 else ((0, 128, 160, 192, 256) if M in (257, 1000, 12865) else (0,))
            for tile in tiles:
                _lib.check(lib.cbas_debug_gemm_split_compare(M, N, K, epi, tile, ctypes.byref(n)), "cbas_debug_gemm_split_compare")
                assert n.value == 0, (M, N, K, epi, tile, n.value)


def _e2e(golden_dir, name, cfg, dim, batch, precision=3):
    from cbas_amd.stream import ClipStream
    g = load(golden_dir, name)
    n = int(g["n"])
    hw = int(g["hw"]) if "hw" in g.files else 224
    fr = synth.cage_frames(int(g["frame_seed"]), n, hw, hw)
    enc = make_enc(cfg, hw, batch, precision)
    head = make_head(dim)
    try:
        st = ClipStream(enc, head, capacity=n)
        for i in range(0, n, batch):
            st.push_u8(torch.from_numpy(fr[i:i + batch]).cuda())
        cls16, probs = st.finish()
        torch.cuda.synchronize()
        return g, cls16.cpu().numpy(), probs.cpu().numpy()
    finally:
        enc.close(); head.close()


def _strict_gate(golden_dir, tag, name, g, cls16, probs, ref_cls32, sel):
    """Every label identical to the reference's.  The one thing that is not a relaxation of that: where the reference's
    OWN execution variants (tests/golden/<name>_variants*.npz, made by scripts/ref_self_variance.py: the same wrapper +
    infer_file with 1 frame per encoder call instead of 8, or with MKL restricted to AVX2 as on a CPU without AVX-512)
    label a frame differently from its primary run, either of the reference's labels counts - a label the reference does
    not agree with itself on is not a target."""
    from conftest import assert_labels_match
    import glob
    alts, moved = [], set()
    for vp in sorted(glob.glob(os.path.join(golden_dir, name + "_variants*.npz"))):
        v = np.load(vp)
        for k in v.files:
            if k.startswith("labels_"):
                alts.append(v[k])
                d = np.nonzero(v[k] != g["labels"])[0]
                moved.update(d.tolist())
                print(f"[{tag}] reference variant {k[7:]}: its own labels differ from the primary run at frames {d.tolist()}")
    ulp = (cls16 != g["cls_f16"])
    r = rel_rows(cls16[sel].astype(np.float32), ref_cls32)
    print(f"[{tag}] fp16 rows: {ulp.mean() * 100:.3f} % of elements round differently from the reference's; transitions "
          f"{int((g['labels'][1:] != g['labels'][:-1]).sum())}")
    n_mis, _ = assert_labels_match(probs, g["probs"], 5e-3, margin=0.0, alt_labels=alts)     # margin 0: EVERY frame
    assert n_mis <= len(moved)                  # at most the frames the reference itself moves
    assert r.max() < 2.0 ** -11 + 1e-5          # fp16 storage rounding of rows that agree to ~1e-6
    assert ulp.mean() < 2e-2


@pytest.mark.parametrize("precision", PRECISIONS)
def test_fp32_e2e_config1_labels_identical(golden_dir, precision):
    """BASELINE config 1 (ViT-S/16, 64 frames, C = 9) against the reference's own encoder + infer_file outputs:
    EVERY argmax label identical - no near-tie band."""
    g, cls16, probs = _e2e(golden_dir, "e2e_vits16", C.VIT_S16, 384, 8, precision)
    _strict_gate(golden_dir, f"precision {precision} e2e_vits16", "e2e_vits16", g, cls16, probs, g["cls"], slice(None))


@pytest.mark.parametrize("precision", PRECISIONS)
def test_fp32_e2e_config2_model_labels_identical(golden_dir, precision):
    """The headline model (ViT-B/16 through the reference's own DinoEncoder wrapper, 256 frames): every label identical -
    frame 69 (reference top-2 margin 4.1e-5) either way: the reference itself labels it both ways, depending on how many
    frames it hands its encoder per call (tests/golden/e2e_vitb16_variants.npz)."""
    g, cls16, probs = _e2e(golden_dir, "e2e_vitb16", C.VIT_B16, 768, 64, precision)
    _strict_gate(golden_dir, f"precision {precision} e2e_vitb16", "e2e_vitb16", g, cls16, probs, g["cls"], slice(None))


@pytest.mark.parametrize("precision", PRECISIONS)
def test_fp32_e2e_long_clip_labels_identical(golden_dir, precision):
    """2 048 frames of the headline model, dozens of behaviour transitions: every label identical."""
    if not os.path.exists(os.path.join(golden_dir, "e2e_vitb16_long.npz")):
        pytest.skip("long fixture not generated")
    g, cls16, probs = _e2e(golden_dir, "e2e_vitb16_long", C.VIT_B16, 768, 64, precision)
    _strict_gate(golden_dir, f"precision {precision} e2e_vitb16_long", "e2e_vitb16_long", g, cls16, probs, g["cls_every8"], slice(0, None, 8))


@pytest.mark.parametrize("precision", PRECISIONS)
def test_fp32_dinov2_with_registers(golden_dir, precision):
    """CBAS's default encoder family in precision 3 (patch 14, interpolated position embedding, key bias, no RoPE, LN eps
    1e-6): the tiny model's embeddings (incl. the bicubic-antialias interpolation) and CLS at three frame sizes, ViT-B/14 at
    224 and 256 against HF Dinov2WithRegistersModel / the reference's own wrapper - to fp32 rounding."""
    from cbas_amd.encoder import DinoEncoder
    g = load(golden_dir, "dinov2reg_tiny")
    cfg = C.DINOV2_REG_TINY
    enc = DinoEncoder.from_weights(cfg, W.synth_encoder_weights(cfg, 1234), "cuda", max_batch=4, max_frame=(84, 84), precision=precision)
    try:
        for hw in (70, 56, 84, 70):
            fr = synth.cage_frames(20 + hw, 3, hw, hw)
            fd = torch.from_numpy(fr).cuda()
            emb = enc.debug_tap(fd, 0, 0, 0).reshape(3, -1, cfg.hidden_size)
            assert np.abs(emb - g[f"emb_{hw}"]).max() < 2e-5
            _, c32 = enc.encode_u8(fd)
            torch.cuda.synchronize()
            r = rel_rows(c32.cpu().numpy(), g[f"last_{hw}"][:, 0])
            assert r.max() < CLS_TOL_F32, (hw, r.max())
    finally:
        enc.close()
    g = load(golden_dir, "dinov2reg_b14")
    cfg = C.DINOV2_REG_B14
    enc = DinoEncoder.from_weights(cfg, W.synth_encoder_weights(cfg, 1234), "cuda", max_batch=4, max_frame=(256, 256), precision=precision)
    try:
        for hw, seed, n in ((224, 31, 4), (256, 32, 2)):
            fr = synth.cage_frames(seed, n, hw, hw)
            _, c32 = enc.encode_u8(torch.from_numpy(fr).cuda())
            torch.cuda.synchronize()
            r = rel_rows(c32.cpu().numpy(), g[f"cls{hw}"])
            print(f"[precision {precision} dinov2reg_b14 {hw}] CLS rel err max {r.max():.3e}")
            assert r.max() < CLS_TOL_F32, (hw, r.max())
    finally:
        enc.close()


@pytest.mark.parametrize("precision", list(PRECISIONS) + ["default"])
def test_fp32_file_level_dropins_against_the_references_encode_file(golden_dir, tmp_path, monkeypatch, precision):
    """encode_file / infer_file / encode_infer_file with a precision-3 encoder, against the `_cls.h5` rows the REFERENCE's
    own encode_file wrote for the same 600-frame 'video' (tests/golden/encode_file_b1layer.npz): the fp16 rows are the
    reference's except where a value sits on a rounding boundary (well under 1 % of the elements, one ulp each) - the
    fp16-operand default differs in ~2/3 of the elements - and the two file paths agree byte for byte."""
    from cbas_amd import pipeline as P, h5io
    from cbas_amd.encoder import DinoEncoder
    g = load(golden_dir, "encode_file_b1layer")
    cfg = C.ViTConfig(hidden_size=768, intermediate_size=1536, num_hidden_layers=1, num_attention_heads=12, image_size=32)
    if precision == "default":
        # built the reference's way - DinoEncoder(model_identifier, device), no precision anywhere (startup_page.py:66-69):
        # the drop-in's default must be the contract-complete mode
        monkeypatch.delenv("CBAS_PRECISION", raising=False)
        ck = str(tmp_path / "ck")
        W.save_encoder_checkpoint(ck, cfg, W.synth_encoder_weights(cfg, 1234))
        enc = DinoEncoder(ck, device="cuda", max_batch=64, max_frame=(32, 32))
        assert enc.precision == 4
    else:
        enc = DinoEncoder.from_weights(cfg, W.synth_encoder_weights(cfg, 1234), "cuda", max_batch=64, max_frame=(32, 32), precision=precision)
    head = make_head(768)
    try:
        frames = synth.cage_frames(5, 600, 32, 32)
        a, b = tmp_path / "a", tmp_path / "b"
        a.mkdir(); b.mkdir()
        np.save(str(a / "vid.npy"), frames)
        np.save(str(b / "vid.npy"), frames)
        ticks = []
        out = P.encode_file(enc, str(a / "vid.npy"), progress_callback=ticks.append)
        np.testing.assert_allclose(ticks, g["ticks"])
        with h5io.ClsReader(out) as r:
            got = r.read(0, 600)
        ref = g["cls_f16"]
        diff = got != ref
        # one fp16 ulp, or - for elements near zero, where an fp16 ulp is far below fp32's own noise on an O(1) row - 4e-6
        one_ulp = np.abs(got.astype(np.float32) - ref.astype(np.float32))[diff] <= np.abs(ref.astype(np.float32))[diff] * 2.0 ** -10 + 4e-6
        print(f"[fp32 encode_file] {diff.mean() * 100:.3f} % of the fp16 elements differ from the reference's file, all by one ulp "
              f"(or the fp32 noise floor near zero): {bool(one_ulp.all())}")
        assert diff.mean() < 1e-2 and one_ulp.all()
        names = ["eating", "drinking", "rearing", "climbing", "digging", "nesting", "resting", "grooming", "exploring"]
        csv = P.infer_file(out, head, "ds", names, 31, device="cuda")
        h5b, csvb = P.encode_infer_file(enc, head, str(b / "vid.npy"), "ds", names)
        assert open(out, "rb").read() == open(h5b, "rb").read() and open(csv, "rb").read() == open(csvb, "rb").read()
    finally:
        enc.close(); head.close()


@pytest.mark.parametrize("precision", PRECISIONS)
def test_fp32_massive_activation_channels(precision):
    """What real checkpoints do and random weights do not: residual-stream channels hundreds of times larger than the rest,
    20x register tokens, 5x LayerNorm gains on the hot channels (the construction of test_gpu_parity.py's fp16 test).
    Precision 3 keeps everything fp32; precision 4 splits LayerNorm rows, attention context, GELU output, q, k, v and the
    probabilities into fp16 halves after power-of-two scaling - the hot channels must neither overflow fp16 (65 504) nor
    push the small ones' low halves into precision loss that shows: CLS within 5e-6 of the fp32 oracle in both modes."""
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
    enc = DinoEncoder.from_weights(cfg, w, "cuda", max_batch=4, max_frame=(224, 224), precision=precision)
    try:
        _, c32 = enc.encode_u8(torch.from_numpy(fr).cuda())
        tap = enc.debug_tap(torch.from_numpy(fr).cuda(), 6, 7, 0)          # residual stream after block 6 (fp32 in both modes)
        torch.cuda.synchronize()
        assert np.abs(tap).max() > 60.0                                    # the outliers are really there
        assert torch.isfinite(c32).all()
        r = rel_rows(c32.cpu().numpy(), ref)
        print(f"[precision {precision} massive activations] CLS rel err max {r.max():.3e}")
        assert r.max() < CLS_TOL_F32, r.max()
    finally:
        enc.close()


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("family", ["dinov3", "dinov2"])
def test_fp32_non_square_and_ragged_frames(precision, family):
    """The reference takes whatever frame size the video has (`cbas.py:431-435`: no resize): non-square frames, and sizes that
    are not a multiple of the patch (the convolution drops the remainder; the RoPE grid / the interpolated DINOv2 position table
    follow the patch grid), in one encoder handle, in precisions 3 and 4 - CLS within 5e-6 of the fp32 oracle for every size."""
    from cbas_amd.encoder import DinoEncoder
    from oracle import pipeline_oracle as PO
    cfg = C.VIT_TINY if family == "dinov3" else C.DINOV2_REG_TINY
    ps = cfg.patch_size
    w = W.synth_encoder_weights(cfg, 77)
    sizes = [(4 * ps, 6 * ps), (5 * ps, 3 * ps), (4 * ps + 6, 5 * ps + ps - 1), (2 * ps, 2 * ps)]
    hmax, wmax = max(h for h, _ in sizes), max(w_ for _, w_ in sizes)
    enc = DinoEncoder.from_weights(cfg, w, "cuda", max_batch=8, max_frame=(hmax, wmax), precision=precision)
    try:
        for i, (H, Wd) in enumerate(sizes):
            fr = synth.cage_frames(20 + i, 6, H, Wd)
            if family == "dinov3":
                ref = PO.encode_frames(fr, w, cfg, batch=6)
            else:                                                     # oracle/dinov2_oracle.py: interpolated position table, key bias
                from oracle import dinov2_oracle as O2, vit_oracle as V
                px = np.repeat(V.preprocess_green(fr)[:, None], 3, 1)
                ref = O2.forward(px, W.canonical_encoder_weights(cfg, w), cfg)[:, 0]
            _, c32 = enc.encode_u8(torch.from_numpy(fr).cuda())
            torch.cuda.synchronize()
            enc.check_finite()
            r = rel_rows(c32.cpu().numpy(), ref)
            print(f"[{family} precision {precision}] {H}x{Wd}: CLS rel err max {r.max():.2e}")
            assert r.max() < CLS_TOL_F32, (H, Wd, float(r.max()))
    finally:
        enc.close()


def test_out_of_range_activations_are_an_error_not_nan_rows():
    """precision 4 splits its operands into fp16 halves after power-of-two scaling, so an activation beyond the
    documented bound (|GELU output| x 4 < 65 504) overflows the high half, the low half becomes -inf and the three-term product
    NaN - silently, while the mode is what an unmodified CBAS now gets by default.  The reference's fp32 arithmetic has no
    such limit.  The final LayerNorm counts non-finite CLS rows and the wait returns CBAS_ERANGE (a RuntimeError here) naming
    the way out; precision 3 computes the same frames."""
    from cbas_amd.encoder import DinoEncoder
    from oracle import pipeline_oracle as PO
    cfg = C.VIT_TINY
    w = {k: v.copy() for k, v in W.synth_encoder_weights(cfg, 1234).items()}
    w["model.layer.0.mlp.up_proj.bias"][3] = 40000.0                     # GELU(40 000) x 4 = 160 000 > 65 504
    w["model.layer.0.mlp.down_proj.weight"][:, 3] *= 1e-4                # the fp32 result itself stays moderate
    fr = synth.cage_frames(3, 8, 64, 64)
    ref = PO.encode_frames(fr, w, cfg, batch=8)
    assert np.isfinite(ref).all()
    enc = DinoEncoder.from_weights(cfg, w, "cuda", max_batch=8, max_frame=(64, 64), precision=4)
    try:
        enc.submit_host(0, fr)
        with pytest.raises(RuntimeError, match="non-finite CLS row.*precision 4.*CBAS_PRECISION=3"):
            enc.wait(0, want_f32=True)
        enc.submit_host(0, synth.cage_frames(4, 8, 64, 64) * 0)          # the handle stays usable; the count was cleared ...
        with pytest.raises(RuntimeError, match="non-finite"):            # ... and these frames overflow as well (the bias does it)
            enc.wait(0, want_f32=True)
    finally:
        enc.close()
    enc3 = DinoEncoder.from_weights(cfg, w, "cuda", max_batch=8, max_frame=(64, 64), precision=3)
    try:
        enc3.submit_host(0, fr)
        _c16, c32 = enc3.wait(0, want_f32=True)
        assert rel_rows(c32, ref).max() < CLS_TOL_F32
    finally:
        enc3.close()


def test_file_paths_fall_back_to_precision3_when_the_range_is_left(tmp_path, capsys):
    """The reference's fp32 arithmetic has no operand range limit, so a drop-in must produce rows wherever the reference does:
    when a video's activations leave the range of the default mode (precision 4) the file paths re-encode THAT video through
    the encoder's precision-3 twin - `encode_file`, `encode_infer_file` and `dist.encode_files` alike - and say so; the rows are
    the fp32 oracle's (5e-6 before the fp16 store).  CBAS_RANGE_FALLBACK=0 restores the error."""
    from cbas_amd import dist as cdist, pipeline as P, h5io
    from cbas_amd.encoder import DinoEncoder
    from oracle import pipeline_oracle as PO
    cfg = C.VIT_TINY
    w = {k: v.copy() for k, v in W.synth_encoder_weights(cfg, 1234).items()}
    w["model.layer.0.mlp.up_proj.bias"][3] = 40000.0                     # GELU(40 000) x 4 overflows the fp16 high half
    w["model.layer.0.mlp.down_proj.weight"][:, 3] *= 1e-4
    frames = synth.cage_frames(3, 90, 64, 64)
    ref = PO.encode_frames(frames, w, cfg, batch=8)
    ref16 = ref.astype(np.float16)
    names = ["eating", "drinking", "rearing", "climbing", "digging", "nesting", "resting", "grooming", "exploring"]
    head = make_head(cfg.hidden_size)
    enc = DinoEncoder.from_weights(cfg, w, "cuda", max_batch=16, max_frame=(64, 64), precision=4)

    def rows_of(path):
        with h5io.ClsReader(path) as r:
            return r.read(0, 90)

    def close_enough(got):
        d = np.abs(got.astype(np.float32) - ref16.astype(np.float32))
        return (got != ref16).mean() < 2e-2 and (d <= np.abs(ref16.astype(np.float32)) * 2.0 ** -10 + 4e-6).all()
    try:
        for sub in ("a", "b", "c"):
            (tmp_path / sub).mkdir()
            np.save(str(tmp_path / sub / "vid.npy"), frames)
        out = P.encode_file(enc, str(tmp_path / "a" / "vid.npy"))
        assert "re-encoded in precision 3" in capsys.readouterr().out
        assert close_enough(rows_of(out))
        h5b, csvb = P.encode_infer_file(enc, head, str(tmp_path / "b" / "vid.npy"), "ds", names)
        assert "re-encoded in precision 3" in capsys.readouterr().out
        assert open(out, "rb").read() == open(h5b, "rb").read() and os.path.getsize(csvb) > 0
        recs = cdist.encode_files([str(tmp_path / "c" / "vid.npy")], enc, head=head, dataset_name="ds", behaviors=names)
        assert recs[0]["status"] == "ok" and open(recs[0]["cls_file"], "rb").read() == open(out, "rb").read()
        assert open(recs[0]["csv_file"], "rb").read() == open(csvb, "rb").read()
        # the encoder itself is still the precision-4 one and still works on frames inside its range? (these weights overflow
        # on every frame, so only the switch is checked here)
        assert enc.precision == 4 and enc.range_fallback().precision == 3
        os.environ["CBAS_RANGE_FALLBACK"] = "0"
        try:
            with pytest.raises(RuntimeError, match="non-finite CLS row"):
                P.encode_file(enc, str(tmp_path / "a" / "vid.npy"))
        finally:
            del os.environ["CBAS_RANGE_FALLBACK"]
    finally:
        enc.close(); head.close()


@pytest.mark.parametrize("precision", PRECISIONS)
def test_fp32_e2e_dinov2_default_encoder_labels_identical(golden_dir, precision):
    """CBAS's DEFAULT encoder family end to end: DINOv2-with-registers ViT-B/14 through the reference's own DinoEncoder
    wrapper at 256 x 256 (T = 329), 512 frames, then the reference's infer_file (tests/golden/e2e_dinov2reg_b14.npz): every
    label identical - or one of the reference's own variants' labels for that frame (two frames of this clip have reference
    top-2 margins of 1.3e-4 and 9.4e-4)."""
    g, cls16, probs = _e2e(golden_dir, "e2e_dinov2reg_b14", C.DINOV2_REG_B14, 768, 32, precision)
    _strict_gate(golden_dir, f"precision {precision} e2e_dinov2reg_b14", "e2e_dinov2reg_b14", g, cls16, probs, g["cls_every8"],
                 slice(0, None, 8))

"""CPU-side checks: synthetic generators, blob packing, the C-ABI surface of the built library (no
compute calls), _cls.h5 / CSV formats, configuration parsing, error behaviour without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from cbas_amd import config as Cfg, weights as W, synth, h5io, _lib
from cbas_amd import build as B


def test_generators_are_deterministic_and_scaled():
    a = W.synth_normal(1, "x", (1000, 64), 0.02)
    b = W.synth_normal(1, "x", (1000, 64), 0.02)
    assert a.dtype == np.float32 and np.array_equal(a, b)
    assert abs(a.std() - 0.02) < 5e-4 and abs(a.mean()) < 5e-4
    assert not np.array_equal(a, W.synth_normal(2, "x", (1000, 64), 0.02))
    assert not np.array_equal(a, W.synth_normal(1, "y", (1000, 64), 0.02))
    u = W.synth_uniform(1, "u", (10000,), -0.5, 0.5)
    assert u.min() >= -0.5 and u.max() < 0.5 and abs(u.mean()) < 0.02
    f = synth.noise_frames(0, 3, 32, 48)
    assert f.shape == (3, 32, 48, 3) and f.dtype == np.uint8
    assert np.array_equal(f[1:], synth.noise_frames(0, 2, 32, 48, first=1))       # per-frame streams
    c = synth.cage_frames(1, 2, 32, 32)
    assert np.array_equal(c[:, :, :, 2], 255 - c[:, :, :, 1])


def test_known_answer_hash():
    # pins the counter-based generator itself (a change here would silently invalidate every golden)
    a = W.synth_normal(1234, "embeddings.cls_token", (4,), 1.0)
    h = W._hash_stream(1234, "embeddings.cls_token", 2)
    assert h.dtype == np.uint64
    assert np.array_equal(a, W.synth_normal(1234, "embeddings.cls_token", (4,), 1.0))
    assert W._fnv1a64("") == 0xCBF29CE484222325 and W._fnv1a64("a") == 0xAF63DC4C8601EC8C


@pytest.mark.parametrize("name,params_m", [("vits16", 21.60), ("vitb16", 85.66), ("vitl16", 303.1)])
def test_encoder_param_counts_match_survey(name, params_m):
    cfg = Cfg.NAMED_VIT[name]
    n = sum(int(np.prod(s)) for k, s in W.encoder_param_shapes(cfg).items() if not k.endswith("mask_token"))
    assert abs(n / 1e6 - params_m) < 0.05, n


@pytest.mark.parametrize("name,hw,gflop", [("vits16", 224, 9.40), ("vitb16", 224, 35.86), ("vitb16", 256, 47.15),
                                           ("vitl16", 518, 727.2)])
def test_flops_per_frame_match_survey(name, hw, gflop):
    assert abs(Cfg.NAMED_VIT[name].flops_per_frame(hw, hw) / 1e9 - gflop) / gflop < 2e-3


def test_head_param_count_and_flops():
    hc = Cfg.HeadConfig()
    n = sum(int(np.prod(s)) if len(s) else 1 for s in W.head_param_shapes(hc).values())
    assert n == 567701
    assert abs(hc.flops_per_frame_naive() / 1e9 - 0.0347) < 5e-4


def _declared(header_name):
    header = open(os.path.join(os.path.dirname(B.HERE), "include", header_name)).read()
    return set(re.findall(r"\b(cbas_[a-z0-9_]+)\s*\(", header)) - {"cbas_enc_config", "cbas_head_config"}


def _exported(path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return {ln.split()[-1] for ln in out.splitlines() if ln.strip()}


def test_library_exports_every_declared_symbol():
    """include/cbas_mi355x.h <-> libcbas_mi355x.so <-> the ctypes table, without touching a GPU: the PRODUCT library exports
    exactly the declared boundary - no bring-up / harness entry point, no C++ internals (csrc/exports.map)."""
    path = B.build_library(debug=False)
    declared = _declared("cbas_mi355x.h")
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    probe = B.probe_library(path, sorted(declared))          # dlopen in a child (build.probe_library: load order against torch's HIP runtime)
    assert not probe["missing"], probe
    exported = _exported(path)
    assert {e for e in exported if e.startswith("cbas_")} == declared, {e for e in exported if e.startswith("cbas_")} ^ declared
    assert not [e for e in exported if "debug" in e.lower()], [e for e in exported if "debug" in e.lower()]
    assert not [e for e in exported if e.startswith("_Z")], "C++ internals exported"
    assert probe["abi"] == _lib.EXPECTED_ABI and probe["debug_build"] == 0


def test_debug_library_is_a_superset_with_the_debug_header():
    """libcbas_mi355x_debug.so = the product boundary + include/cbas_mi355x_debug.h (what the GPU suite loads)."""
    path = B.build_library(debug=True)
    product, debug = _declared("cbas_mi355x.h"), _declared("cbas_mi355x_debug.h")
    probe = B.probe_library(path, sorted(product | debug))
    assert debug == set(_lib.DEBUG_SIGNATURES), debug ^ set(_lib.DEBUG_SIGNATURES)
    assert not (product & debug)
    assert {e for e in _exported(path) if e.startswith("cbas_")} == product | debug
    assert probe["debug_build"] == 1 and probe["abi"] == _lib.EXPECTED_ABI and not probe["missing"]
    assert all("debug" in d for d in debug), [d for d in debug if "debug" not in d]


def test_weight_counts_agree_between_host_and_library():
    from cbas_amd.encoder import pack_encoder_weights
    from cbas_amd.head import pack_head_weights
    lib = _lib.load()
    cfg = Cfg.VIT_TINY
    blob = pack_encoder_weights(cfg, W.synth_encoder_weights(cfg, 1))
    cc = _lib.EncConfig(cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers, cfg.num_attention_heads,
                        cfg.num_register_tokens, 16, 1e-5, 100.0, 8, 64, 64, 0, 1, 0)
    assert lib.cbas_enc_weights_count(C.byref(cc)) == blob.shape[0]
    v2 = Cfg.DINOV2_REG_TINY                              # position embedding + key bias in the blob
    blob2 = pack_encoder_weights(v2, W.synth_encoder_weights(v2, 1))
    cc2 = _lib.EncConfig(v2.hidden_size, v2.intermediate_size, v2.num_hidden_layers, v2.num_attention_heads,
                         v2.num_register_tokens, 14, 1e-6, 100.0, 8, 70, 70, 0, 0, v2.pos_embed_grid)
    assert lib.cbas_enc_weights_count(C.byref(cc2)) == blob2.shape[0]
    hc2 = Cfg.HeadConfig(lstm_layers=2)
    hb2 = pack_head_weights(hc2, W.synth_head_weights(hc2, 2))
    hcc2 = _lib.HeadConfigC(hc2.in_features, hc2.out_features, hc2.seq_len, hc2.bottleneck_dim, hc2.lin0_dim,
                            hc2.lstm_hidden_size, hc2.center_window_size, hc2.ema_alpha, 2, 1)
    assert lib.cbas_head_weights_count(C.byref(hcc2)) == hb2.shape[0]
    for h, acc in ((64, True), (128, True), (48, True), (64, False)):
        hc = Cfg.HeadConfig(lstm_hidden_size=h, out_features=7, use_acceleration=acc)
        hb = pack_head_weights(hc, W.synth_head_weights(hc, 2))
        hcc = _lib.HeadConfigC(hc.in_features, hc.out_features, hc.seq_len, hc.bottleneck_dim, hc.lin0_dim,
                               hc.lstm_hidden_size, hc.center_window_size, hc.ema_alpha, 1, int(acc))
        assert lib.cbas_head_weights_count(C.byref(hcc)) == hb.shape[0]


def test_create_rejects_bad_arguments_without_gpu():
    lib = _lib.load()
    cc = _lib.EncConfig(100, 400, 2, 2, 4, 16, 1e-5, 100.0, 8, 64, 64, 0, 1, 0)      # hidden_size not /128
    h = C.c_void_p()
    dummy = np.zeros(4, np.float32)
    rc = lib.cbas_enc_create(C.byref(cc), dummy.ctypes.data, 4, 0, C.byref(h))
    assert rc == -1 and b"hidden_size" in lib.cbas_last_error()
    assert lib.cbas_enc_forward_u8(None, None, 1, 16, 16, 0, 0, 0, None, None, None) == -1
    assert lib.cbas_head_infer_f16(None, None, 1, 1.0, None, None, None) == -1


def test_precision_argument_checks_without_gpu(tmp_path, monkeypatch):
    """precision outside 0..4 is refused by the library before anything touches a device; the reference-style constructor
    takes its mode from CBAS_PRECISION and refuses MX-fp8 there (its rows need their own heads)."""
    lib = _lib.load()
    h = C.c_void_p()
    dummy = np.zeros(4, np.float32)
    for bad in (-1, 5):
        cc = _lib.EncConfig(128, 512, 2, 2, 4, 16, 1e-5, 100.0, 8, 64, 64, bad, 1, 0)
        rc = lib.cbas_enc_create(C.byref(cc), dummy.ctypes.data, 4, 0, C.byref(h))
        assert rc == -1 and b"precision" in lib.cbas_last_error()
    from cbas_amd.encoder import DinoEncoder
    ck = str(tmp_path / "ck")
    W.save_encoder_checkpoint(ck, Cfg.VIT_TINY, W.synth_encoder_weights(Cfg.VIT_TINY, 3))
    monkeypatch.setenv("CBAS_PRECISION", "2")
    with pytest.raises(ValueError, match="MX-fp8"):
        DinoEncoder(ck, device="cuda")


def test_reference_style_construction_defaults_to_the_contract_complete_mode(tmp_path, monkeypatch):
    """DinoEncoder(model_identifier, device) - how CBAS builds its encoder (startup_page.py:66-69), hence what
    integration.install() hands an unmodified checkout - and `python -m cbas_amd.encode_files` compute in precision 4 (CLS
    ~1e-6 from the reference's fp32 CPU rows, every argmax label the reference's) unless CBAS_PRECISION / --precision says
    otherwise; the fp16-operand mode is the explicit opt-in."""
    from cbas_amd import encoder as E
    assert E.DEFAULT_PRECISION == 4
    ck = str(tmp_path / "ck")
    W.save_encoder_checkpoint(ck, Cfg.VIT_TINY, W.synth_encoder_weights(Cfg.VIT_TINY, 3))
    seen = []
    monkeypatch.setattr(E.DinoEncoder, "_init", lambda self, cfg, w, dev, mb, mf, precision: seen.append(precision))
    monkeypatch.delenv("CBAS_PRECISION", raising=False)
    E.DinoEncoder(ck, device="cuda")
    monkeypatch.setenv("CBAS_PRECISION", "0")
    E.DinoEncoder(ck, device="cuda")
    E.DinoEncoder(ck, device="cuda", precision=3)
    assert seen == [4, 0, 3]
    # the command-line tool resolves its default the same way
    import cbas_amd.encode_files as EF
    src = open(EF.__file__).read()
    assert 'default=None, choices=(0, 1, 2, 3, 4)' in src and "DEFAULT_PRECISION" in src


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from cbas_amd.encoder import DinoEncoder
    cfg = Cfg.VIT_TINY
    with pytest.raises(RuntimeError):
        DinoEncoder.from_weights(cfg, W.synth_encoder_weights(cfg, 1), "cpu")
    from cbas_amd.head import ClassifierLSTMDeltas
    m = ClassifierLSTMDeltas(768, 9)
    m.load_state_dict(W.synth_head_weights(Cfg.HeadConfig(), 1))
    m.to("cpu")
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 31, 768))


def test_head_state_dict_validation():
    from cbas_amd.head import pack_head_weights
    hc = Cfg.HeadConfig()
    w = W.synth_head_weights(hc, 1)
    del w["lin2.bias"]
    with pytest.raises(KeyError):
        pack_head_weights(hc, w)
    assert W.infer_head_config(W.synth_head_weights(Cfg.HeadConfig(lstm_hidden_size=128, out_features=4), 1)) \
        .lstm_hidden_size == 128


def test_cls_h5_roundtrip_and_layout(tmp_path):
    p = str(tmp_path / "v_cls.h5")
    rows = synth.cls_walk(1, 700, 768)
    with h5io.ClsWriter(p, 768, {"encoder_model_identifier": "facebook/dinov3-vitb16-pretrain-lvd1689m",
                                 "schema_version": "1.0"}) as w:
        w.append(rows[:512]); w.flush(); w.append(rows[512:].astype(np.float32)); w.flush()
    with h5io.ClsReader(p) as r:
        assert r.shape == (700, 768) and r.itemsize == 2
        assert r.attrs == {"encoder_model_identifier": "facebook/dinov3-vitb16-pretrain-lvd1689m",
                           "schema_version": "1.0"}
        assert np.array_equal(r.read(0, 700), rows)
        assert np.array_equal(r.read(690, 9999), rows[690:])
    h5dump = "/opt/conda/bin/h5dump"
    if os.path.exists(h5dump):
        import subprocess
        txt = subprocess.run([h5dump, "-H", "-p", p], capture_output=True, text=True).stdout
        assert "16-bit little-endian floating-point" in txt
        assert "( 700, 768 ) / ( H5S_UNLIMITED, 768 )" in txt and "CHUNKED ( 8192, 768 )" in txt
        assert "H5T_VARIABLE" in txt and "H5T_CSET_UTF8" in txt
    with h5io.ClsWriter(str(tmp_path / "e_cls.h5"), 384) as w:
        pass
    with h5io.ClsReader(str(tmp_path / "e_cls.h5")) as r:
        assert r.shape == (0, 384) and r.attrs == {}


def test_csv_text_matches_pandas_golden(golden_dir, tmp_path):
    from cbas_amd.pipeline import format_probs_csv, write_probs_csv
    g = np.load(os.path.join(golden_dir, "infer_file.npz"))
    names = ["eating", "drinking", "rearing", "climbing", "digging", "nesting", "resting", "grooming", "exploring"]
    for n in (1, 10, 31, 64, 40):
        want = bytes(g[f"csv_{n}"]).decode()
        assert format_probs_csv(g[f"probs_{n}"], names) == want
        p = str(tmp_path / f"o{n}.csv")
        write_probs_csv(p, g[f"probs_{n}"], names)
        assert open(p).read() == want
    assert format_probs_csv(np.array([[1e-5, 0.5]], np.float32), ["a,b", 'q"']).splitlines()[0] == '"a,b","q"""'
    assert format_probs_csv(np.array([[1.2e-5, 1.0]], np.float32), ["a", "b"]).splitlines()[1] == "1.2e-05,1.0"


def test_checkpoint_dir_roundtrip(tmp_path):
    cfg = Cfg.VIT_TINY
    w = W.synth_encoder_weights(cfg, 5)
    W.save_encoder_checkpoint(str(tmp_path / "ck"), cfg, w)
    cfg2, w2 = W.load_encoder_checkpoint(Cfg.find_checkpoint_dir(str(tmp_path / "ck")))
    assert cfg2 == cfg and all(np.array_equal(w[k], w2[k]) for k in w)
    with pytest.raises(FileNotFoundError):
        Cfg.find_checkpoint_dir("facebook/definitely-not-cached")
    with pytest.raises(NotImplementedError):
        Cfg.ViTConfig(use_gated_mlp=True).validate()


def test_frame_sources(tmp_path):
    from cbas_amd import pipeline as P
    fr = synth.cage_frames(2, 5, 32, 32)
    np.save(str(tmp_path / "clip.npy"), fr)
    src = P.open_video(str(tmp_path / "clip.npy"))
    assert len(src) == 5 and np.array_equal(src.get_batch(range(1, 4)), fr[1:4])
    with pytest.raises(RuntimeError):
        P.open_video(str(tmp_path / "clip.mp4"))          # decord absent, no reader registered


def test_cls_file_bytes_do_not_depend_on_the_append_granularity(tmp_path):
    """encode_file appends 512 rows per chunk and flushes (backend/cbas.py:437-440); the multi-GPU writer and
    encode_infer_file write a clip in one append.  Across several 8192-row HDF5 chunks the files are byte-identical."""
    import hashlib
    from cbas_amd import pipeline as P
    rows = np.random.default_rng(0).standard_normal((20_000, 64)).astype(np.float16)
    attrs = {"encoder_model_identifier": "facebook/dinov3-vitb16-pretrain-lvd1689m", "schema_version": "1.0"}
    a = str(tmp_path / "chunked.h5")
    with h5io.ClsWriter(a, 64, attrs) as w:
        for i in range(0, rows.shape[0], 512):
            w.append(rows[i:i + 512])
            w.flush()
    P.set_project_stamp(attrs["encoder_model_identifier"])
    try:
        b = P.write_cls_file(str(tmp_path / "video.mp4"), rows)
    finally:
        P.set_project_stamp(None)
    sha = lambda p: hashlib.sha256(open(p, "rb").read()).hexdigest()      # noqa: E731
    assert sha(a) == sha(b) and not os.path.exists(b + ".tmp")


def test_cls_reader_hands_out_half_rows_as_they_are_and_anything_else_as_float32(tmp_path):
    """The reference reads any `cls` dtype and converts with .float() (backend/cbas.py:507-508)."""
    x = np.random.default_rng(1).standard_normal((300, 32))
    for dt, want in (("f2", np.float16), ("f4", np.float32), ("f8", np.float32)):
        p = str(tmp_path / f"{dt}.h5")
        with h5io.ClsWriter(p, 32, {}, dtype=dt) as w:
            w.append(x)
        with h5io.ClsReader(p) as r:
            got = r.read(17, 203)
            assert r.is_half == (dt == "f2") and got.dtype == want and got.shape == (186, 32)
            assert np.array_equal(got, x[17:203].astype(np.dtype(dt)).astype(want))


def test_fp8_rows_are_stamped_as_such(tmp_path):
    """MX-fp8 rows must be distinguishable on disk from fp16 rows."""
    from cbas_amd import pipeline as P
    from cbas_amd.encode_files import _needs_encoding
    enc16 = type("E", (), {"precision": 0})()
    enc8 = type("E", (), {"precision": 2})()
    P.set_project_stamp("facebook/dinov3-vitb16-pretrain-lvd1689m")
    try:
        a16, a8 = P.file_attrs(enc16), P.file_attrs(enc8)
    finally:
        P.set_project_stamp(None)
    assert a16 == {"encoder_model_identifier": "facebook/dinov3-vitb16-pretrain-lvd1689m", "schema_version": "1.0"}
    assert a8["encoder_model_identifier"] == "facebook/dinov3-vitb16-pretrain-lvd1689m#mx-fp8" and a8["encoder_precision"] == "mx-fp8"
    assert P.file_attrs(enc8) == {"encoder_precision": "mx-fp8"}                 # no project: still marked
    rows = np.zeros((4, 16), np.float16)
    v8 = str(tmp_path / "v8.mp4")
    P.write_cls_file(v8, rows, a8)
    with h5io.ClsReader(os.path.splitext(v8)[0] + "_cls.h5") as r:
        assert r.attrs["encoder_precision"] == "mx-fp8"
    # a project on the fp16 encoder re-encodes such a file; an fp8 run finds it up to date
    assert _needs_encoding(v8, "facebook/dinov3-vitb16-pretrain-lvd1689m")
    assert not _needs_encoding(v8, "facebook/dinov3-vitb16-pretrain-lvd1689m#mx-fp8")


def test_checkpoint_loader_reads_bf16_and_sharded_saves(tmp_path):
    """A checkpoint someone re-saved in bfloat16 (numpy cannot hold it) or in shards (`max_shard_size`) loads as float32."""
    import json
    import torch
    from safetensors.torch import save_file
    cfg = Cfg.VIT_TINY
    w = W.synth_encoder_weights(cfg, 1234)
    plain = str(tmp_path / "plain")
    W.save_encoder_checkpoint(plain, cfg, w)
    _, ref = W.load_encoder_checkpoint(plain)
    assert all(np.array_equal(ref[k], np.asarray(w[k], np.float32)) for k in w)
    b16 = tmp_path / "bf16"
    b16.mkdir()
    (b16 / "config.json").write_text(open(os.path.join(plain, "config.json")).read())
    save_file({k: torch.from_numpy(np.asarray(v, np.float32)).to(torch.bfloat16).contiguous() for k, v in w.items()},
              str(b16 / "model.safetensors"))
    _, got = W.load_encoder_checkpoint(str(b16))
    for k, v in w.items():
        want = torch.from_numpy(np.asarray(v, np.float32)).to(torch.bfloat16).float().numpy()
        assert got[k].dtype == np.float32 and np.array_equal(got[k], want), k
    sh = tmp_path / "sharded"
    sh.mkdir()
    (sh / "config.json").write_text(open(os.path.join(plain, "config.json")).read())
    names = sorted(w)
    parts = [names[0::2], names[1::2]]
    wm = {}
    for i, part in enumerate(parts):
        fn = f"model-0000{i + 1}-of-00002.safetensors"
        save_file({k: torch.from_numpy(np.ascontiguousarray(np.asarray(w[k], np.float32))) for k in part}, str(sh / fn))
        wm.update({k: fn for k in part})
    (sh / "model.safetensors.index.json").write_text(json.dumps({"metadata": {}, "weight_map": wm}))
    _, got = W.load_encoder_checkpoint(str(sh))
    assert all(np.array_equal(got[k], ref[k]) for k in ref)
    with pytest.raises(FileNotFoundError):
        W.load_encoder_checkpoint(str(tmp_path))
    # a backbone saved under a task model's prefix
    pre = tmp_path / "prefixed"
    pre.mkdir()
    (pre / "config.json").write_text(open(os.path.join(plain, "config.json")).read())
    tensors = {"dinov3_vit." + k: torch.from_numpy(np.ascontiguousarray(np.asarray(v, np.float32))) for k, v in w.items()}
    tensors["classifier.weight"] = torch.zeros(3, cfg.hidden_size)
    save_file(tensors, str(pre / "model.safetensors"))
    _, got = W.load_encoder_checkpoint(str(pre))
    assert all(np.array_equal(got[k], ref[k]) for k in ref)



def test_asmcheck_bans_the_packed_form_round5_identified():
    """cbas_amd/asmcheck.py: the rule on a hand-written disassembly, then on the device code of every kernel of the product."""
    from cbas_amd import asmcheck as A
    text = """
0000000000001c00 <kern_a>:
	v_pk_add_f32 v[8:9], v[10:11], v[8:9] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]// 000000001C00: D3B24008 5A021114
	v_pk_add_f32 v[4:5], v[6:7], v[4:5] neg_lo:[0,1] neg_hi:[0,1]// 000000001C08: D3B24004 1A020906
	v_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel_hi:[1,0]        // 000000001C10: D3B10000 08020902
	v_pk_fma_f32 v[0:1], v[2:3], s[4:5], v[6:7] op_sel_hi:[1,0,1]// 000000001C18: D3B00000 0C180902
	v_sub_f32_e32 v18, v9, v8                                  // 000000001C20: 04241109
0000000000001d00 <kern_b>:
	v_pk_fma_f32 v[36:37], v[36:37], v[0:1], v[2:3] op_sel:[0,1,0]// 000000001D00: D3B04024 1C0A0124
"""
    ks = A.parse_disassembly(text)
    assert set(ks) == {"kern_a", "kern_b"} and len(ks["kern_a"]) == 5
    fa, fb = A.check_kernel(ks["kern_a"]), A.check_kernel(ks["kern_b"])
    assert [f["rule"] for f in fa] == ["R1", "R2"], fa            # the scalar-pair broadcast (s[4:5]) is not a VGPR pair
    assert [f["rule"] for f in fb] == ["R1"], fb
    B.build_library(debug=False)
    rep = A.check_library(enforce=False)
    assert rep["kernels"] > 150 and rep["packed_f32_ops"] > 10000
    assert rep["R1"] == [], rep["R1"][:5]
    # the head's files are compiled without packed fp32 altogether
    for obj in ("head_kernels.o", "head_train_kernels.o"):
        ks = A.disassemble_object(os.path.join(B.HERE, "build", obj))
        assert ks and not [mn for insns in ks.values() for mn, _, _ in insns if mn.startswith("v_pk_") and mn.endswith("_f32")], obj


def test_asmcheck_catches_the_form_in_real_compiler_output(tmp_path):
    """The checker against hipcc's own output: two subtractions whose operands sit in opposite halves of their register pairs
    compile to `v_pk_add_f32 ... op_sel:[0,1] op_sel_hi:[1,0]` - the banned form - and the same source with
    common.h's `keep_scalar` on one of them does not."""
    import shutil
    import subprocess
    from cbas_amd import asmcheck as A
    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not present")
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = tmp_path / "k.hip"
    src.write_text('''
#include "common.h"
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void crossed(const f32x2* a, const f32x2* b, f32x2* c) {
    const int i = threadIdx.x;
    const f32x2 x = a[i], y = b[i];
    c[i] = f32x2{x[0] - y[1], x[1] - y[0]};
}
__global__ void pinned(const f32x2* a, const f32x2* b, f32x2* c) {
    const int i = threadIdx.x;
    const f32x2 x = a[i], y = b[i];
    c[i] = f32x2{keep_scalar(x[0] - y[1]), x[1] - y[0]};
}
''')
    obj = tmp_path / "k.o"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(B.HERE, "csrc"), "-c", str(src), "-o", str(obj)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    rep = A.check_objects([str(obj)])
    assert rep["kernels"] == 2
    kernels_with_r1 = {f["kernel"] for f in rep["R1"]}
    assert any("crossed" in k for k in kernels_with_r1), rep
    assert not any("pinned" in k for k in kernels_with_r1), rep["R1"]

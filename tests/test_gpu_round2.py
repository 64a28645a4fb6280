"""Round-2 GPU tests of the encoder's host/runtime behaviour (all through the C ABI):

* the pruned last layer (query/attention/MLP on the CLS rows only) is bit-identical to the full one,
  for the LDS-resident and the streaming attention kernels;
* cbas_enc_submit_u8_host ships the caller's bytes as they are (RGB interleaved, packed plane, strided
  frames, pinned or pageable memory) and always gives the device-resident result bit for bit;
* a queue that mixes resolutions needs no device synchronisation and stays correct;
* two synchronous forwards on different caller streams are ordered by the library.
"""
import os

import numpy as np
import pytest
import torch

from cbas_amd import config as C, weights as W, synth

pytestmark = pytest.mark.gpu


def _enc(cfgname, max_batch, hw):
    from cbas_amd.encoder import DinoEncoder
    cfg = C.NAMED_VIT[cfgname]
    return cfg, DinoEncoder.from_weights(cfg, W.synth_encoder_weights(cfg, 1234), "cuda", max_batch=max_batch,
                                         max_frame=hw)


@pytest.mark.parametrize("cfgname,n,hw", [("vitb16", 64, (224, 224)),      # bench shape, resident attention (T=201)
                                          ("vits16", 5, (320, 336)),       # T=425: streaming attention
                                          ("tiny", 7, (64, 64)),
                                          ("dinov2regtiny", 3, (70, 70))])
def test_pruned_last_layer_is_bit_identical(cfgname, n, hw):
    cfg, enc = _enc(cfgname, n, hw)
    try:
        fr = torch.from_numpy(synth.noise_frames(3, n, *hw)).cuda()
        enc.set_prune_last_layer(False)
        f16, f32 = enc.encode_u8(fr)
        enc.set_prune_last_layer(True)
        p16, p32 = enc.encode_u8(fr)
        torch.cuda.synchronize()
        assert torch.isfinite(f32).all()
        assert torch.equal(f32, p32) and torch.equal(f16, p16)
        # and the float-input entry point (DinoEncoder.forward) takes the same pruned path
        x = (fr[:, :, :, 1].float() / 255.0).unsqueeze(1)
        a = enc(x)
        enc.set_prune_last_layer(False)
        b = enc(x)
        torch.cuda.synchronize()
        assert torch.equal(a, b)
    finally:
        enc.close()


def test_host_submission_layouts_match_device_path():
    cfg, enc = _enc("tiny", 8, (64, 64))
    try:
        rgb = synth.cage_frames(21, 8, 64, 64)                            # (8,64,64,3) uint8, pageable
        ref16, ref32 = enc.encode_u8(torch.from_numpy(rgb).cuda())
        torch.cuda.synchronize()
        ref16, ref32 = ref16.cpu().numpy(), ref32.cpu().numpy()

        def run(arr, **kw):
            enc.submit_host(0, arr, **kw)
            o16, o32 = enc.wait(0, want_f32=True)
            assert np.array_equal(o16.view(np.uint16), ref16.view(np.uint16)) and np.array_equal(o32, ref32)

        run(rgb)                                                          # pageable interleaved RGB: one span
        pinned = torch.from_numpy(rgb).pin_memory()
        run(pinned.numpy())                                               # pinned: direct DMA from the caller's bytes
        run(np.ascontiguousarray(rgb[:, :, :, 1]))                        # packed green plane
        # frames strided apart (a view into a larger buffer): one span per frame
        wide = np.zeros((8, 2, 64, 64, 3), np.uint8)
        wide[:, 0] = rgb
        from cbas_amd import _lib
        fs = wide.strides[0]
        _lib.check(enc._lib.cbas_enc_submit_u8_host(enc._h, 1, wide.ctypes.data + 1, 8, 64, 64, fs, 64 * 3, 3),
                   "cbas_enc_submit_u8_host")
        enc._slot_n = {1: 8}
        o16, o32 = enc.wait(1, want_f32=True)
        assert np.array_equal(o32, ref32)
        # sparse layout (pixel stride 16): falls back to the host gather, same result
        sparse = np.zeros((8, 64, 64, 16), np.uint8)
        sparse[..., 5] = rgb[..., 1]
        _lib.check(enc._lib.cbas_enc_submit_u8_host(enc._h, 2, sparse.ctypes.data + 5, 8, 64, 64, 64 * 64 * 16, 64 * 16, 16),
                   "cbas_enc_submit_u8_host")
        enc._slot_n = {2: 8}
        assert np.array_equal(enc.wait(2, want_f32=True)[1], ref32)
    finally:
        enc.close()


def test_mixed_resolution_queue():
    """Batches of different frame sizes alternate on the two lanes; each resolution's RoPE table lives in its own
    buffers, so no batch reads a table that a later batch's resolution overwrote."""
    from cbas_amd import _lib as L
    cfg, enc = _enc("tiny", 8, (96, 96))
    try:
        sizes = [(64, 64), (96, 96), (64, 96), (80, 48), (64, 64), (96, 96), (48, 80), (64, 96), (32, 32), (96, 64)]
        frames = [torch.from_numpy(synth.cage_frames(30 + i, 8, h, w)).cuda() for i, (h, w) in enumerate(sizes)]
        ref = []
        for f in frames:
            ref.append(enc.encode_u8(f, want_f32=False)[0].clone())
        torch.cuda.synchronize()
        for rep in range(3):
            outs = [torch.zeros_like(r) for r in ref]
            busy = []
            for i, f in enumerate(frames):
                slot = i % L.ENC_SLOTS
                if slot in busy:
                    enc.wait_stream(slot)
                    busy.remove(slot)
                enc.submit_dev(slot, f, outs[i])
                busy.append(slot)
            for slot in busy:
                enc.wait_stream(slot)
            torch.cuda.synchronize()
            for i in range(len(frames)):
                assert torch.equal(outs[i], ref[i]), (rep, i, sizes[i])
    finally:
        enc.close()


def test_two_synchronous_forwards_on_different_streams():
    cfg, enc = _enc("tiny", 16, (64, 64))
    try:
        fa = torch.from_numpy(synth.cage_frames(41, 16, 64, 64)).cuda()
        fb = torch.from_numpy(synth.cage_frames(42, 16, 64, 64)).cuda()
        ra = enc.encode_u8(fa, want_f32=False)[0].clone()
        rb = enc.encode_u8(fb, want_f32=False)[0].clone()
        torch.cuda.synchronize()
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        for rep in range(8):
            with torch.cuda.stream(s1):
                a = enc.encode_u8(fa, want_f32=False)[0]
            with torch.cuda.stream(s2):
                b = enc.encode_u8(fb, want_f32=False)[0]
            torch.cuda.synchronize()
            assert torch.equal(a, ra) and torch.equal(b, rb), rep
    finally:
        enc.close()


class _RefLikeHead(torch.nn.Module):
    """The attribute / state_dict surface of the reference's classifier_head.ClassifierLSTMDeltas (what
    from_reference_module reads); no forward - the MI355X head does the arithmetic."""

    def __init__(self, I=768, Cn=9, T=31, h=64):
        super().__init__()
        nn = torch.nn
        self.in_features, self.out_features, self.seq_len, self.sw, self.ema_alpha = I, Cn, T, 5, 0.3
        self.gate = nn.Parameter(torch.zeros(()))
        self.attention_temp = nn.Parameter(torch.zeros(()))
        for s in ("cls", "delta", "acc"):
            setattr(self, f"{s}_bottleneck", nn.Sequential(nn.Linear(I, 128)))
            setattr(self, f"{s}_ln", nn.LayerNorm(128))
        self.lin0 = nn.Sequential(nn.Linear(384, 256))
        self.lstm = nn.LSTM(256, h, num_layers=1, batch_first=True, bidirectional=True)
        self.attention_head = nn.Linear(2 * h, 1)
        self.lin1 = nn.Linear(I, Cn)
        self.lin2 = nn.Linear(2 * h, Cn)


def test_infer_file_follows_weight_updates_of_a_reference_module(tmp_path):
    """the device copy of a reference torch head is rebuilt when the module is trained further or
    reloaded in place, or when a new module reuses a freed module's id."""
    from cbas_amd import pipeline as P, h5io
    from oracle import pipeline_oracle as PO
    names = [f"b{i}" for i in range(9)]
    hcfg = C.HeadConfig()
    w1, w2 = W.synth_head_weights(hcfg, 4321), W.synth_head_weights(hcfg, 99)
    cls = (np.random.default_rng(0).standard_normal((300, 768)) * 2).astype(np.float16)
    path = str(tmp_path / "v_cls.h5")
    with h5io.ClsWriter(path, 768, {}) as w:
        w.append(cls)

    def run(model):
        out = P.infer_file(path, model, "m", names, 31, device="cuda")
        assert out is not None
        return np.loadtxt(out, delimiter=",", skiprows=1, dtype=np.float32)

    m = _RefLikeHead()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w1.items()})
    p1 = run(m)
    np.testing.assert_allclose(p1, PO.classify_cls(cls, w1, 31, 1.0), atol=1e-4)
    assert np.array_equal(run(m), p1)                                   # cached copy reused: same result
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w2.items()})  # in-place update of the same module
    p2 = run(m)
    np.testing.assert_allclose(p2, PO.classify_cls(cls, w2, 31, 1.0), atol=1e-4)
    assert np.abs(p2 - p1).max() > 1e-3
    with torch.no_grad():                                               # one optimiser-style in-place step
        m.lin1.bias.add_(0.5)
    w3 = {k: v.detach().numpy().copy() for k, v in m.state_dict().items()}
    np.testing.assert_allclose(run(m), PO.classify_cls(cls, w3, 31, 1.0), atol=1e-4)
    m.lin2.bias.data.mul_(-1.0)                                         # a write through .data bumps no version counter
    m.gate.data.fill_(1.5)
    w4 = {k: v.detach().numpy().copy() for k, v in m.state_dict().items()}
    p4 = run(m)
    np.testing.assert_allclose(p4, PO.classify_cls(cls, w4, 31, 1.0), atol=1e-4)
    assert np.abs(p4 - PO.classify_cls(cls, w3, 31, 1.0)).max() > 1e-3


def test_fused_session_host_and_device_pushes_equal_two_step_path():
    """cbas_fused_* (encode -> head with the rows resident in HBM) == cbas_enc_forward_u8 over the clip followed by
    cbas_head_infer_f16 over the rows, bit for bit: ragged pushes larger and smaller than max_batch, host (pageable
    and pinned) and device frames, several clips through one session, capacity overflow is an error."""
    from cbas_amd.head import ClassifierLSTMDeltas
    from cbas_amd.stream import ClipStream
    cfg, enc = _enc("tiny", 16, (64, 64))
    hcfg = C.HeadConfig(in_features=cfg.hidden_size, out_features=5, seq_len=31)
    head = ClassifierLSTMDeltas(cfg.hidden_size, 5, seq_len=31)
    head.load_state_dict(W.synth_head_weights(hcfg, 7))
    head.to("cuda")
    try:
        frames = synth.cage_frames(51, 331, 64, 64)
        fd = torch.from_numpy(frames).cuda()
        ref16, _ = enc.encode_u8(fd, want_f32=False)
        refp = head.infer_clip(ref16, 0.8)
        torch.cuda.synchronize()
        st = ClipStream(enc, head, capacity=400, temperature=0.8, classify_every=48)
        pinned = torch.from_numpy(frames).pin_memory().numpy()
        for src in ("dev", "host", "pinned", "dev"):
            st.reset()
            o = 0
            for n in (16, 40, 3, 100, 16, 1, 155):             # 331 frames
                if src == "dev":
                    st.push_u8(fd[o:o + n])
                else:
                    st.push_host((frames if src == "host" else pinned)[o:o + n])
                o += n
            if src == "dev":
                c16, pr = st.finish()
                torch.cuda.synchronize()
                assert torch.equal(c16, ref16) and torch.equal(pr, refp), src
            else:
                c16, pr = st.finish_host()
                assert np.array_equal(c16.view(np.uint16), ref16.cpu().numpy().view(np.uint16)), src
                assert np.array_equal(pr, refp.cpu().numpy()), src
        st.reset()
        st.push_u8(fd[:300])
        with pytest.raises(RuntimeError, match="capacity"):
            st.push_u8(fd[:101])
        c16, pr = st.finish()
        torch.cuda.synchronize()
        assert c16.shape[0] == 300 and torch.equal(c16, ref16[:300])
        st.close()
    finally:
        head.close()
        enc.close()


HEAD_VARIANTS = [("h64_t63", dict(seq_len=63)), ("h64_t95", dict(seq_len=95)), ("h32", dict(lstm_hidden_size=32)),
                 ("h96_t63", dict(lstm_hidden_size=96, seq_len=63)), ("h64_noacc", dict(use_acceleration=False)),
                 ("h48_noacc_l2_t15", dict(use_acceleration=False, lstm_hidden_size=48, lstm_layers=2, seq_len=15))]


@pytest.mark.parametrize("tag,kw", HEAD_VARIANTS, ids=[t for t, _ in HEAD_VARIANTS])
def test_head_variants_against_reference_goldens(golden_dir, tag, kw):
    """The head configurations the reference constructor / loader accept beyond the defaults: seq_len 63 and 95
    (sweep_runner.py:110), lstm_hidden_size other than 64 / 128 (workthreads.py:418-421), use_acceleration=False
    (classifier_head.py:74-84,158-162) - logits and latent vs the reference module's outputs."""
    import os
    from cbas_amd.head import ClassifierLSTMDeltas
    g = np.load(os.path.join(golden_dir, f"head_{tag}.npz"))
    hc = C.HeadConfig(in_features=768, out_features=9, **kw)
    m = ClassifierLSTMDeltas(768, 9, **kw)
    m.load_state_dict(W.synth_head_weights(hc, 4321))
    m.to("cuda")
    T = hc.seq_len
    seq = synth.cls_walk(21, 48 + T - 1, 768).astype(np.float32)
    x = torch.from_numpy(np.stack([seq[i:i + T] for i in range(48)])).cuda()
    logits, latent = m(x)
    logits, latent = logits.cpu().numpy(), latent.cpu().numpy()
    m.close()
    np.testing.assert_allclose(logits, g["logits"], atol=1e-4)
    np.testing.assert_allclose(latent, g["latent"], atol=5e-5)
    assert (logits.argmax(1) == g["logits"].argmax(1)).all()


@pytest.mark.parametrize("n,T", [(40, 63), (300, 63), (260, 95)])
def test_infer_clip_long_windows(golden_dir, n, T):
    """infer_file semantics at the sweep's longer windows, incl. a clip shorter than the window (all-replicate padding)."""
    import os
    from cbas_amd.head import ClassifierLSTMDeltas
    g = np.load(os.path.join(golden_dir, "infer_file_seq.npz"))
    m = ClassifierLSTMDeltas(768, 9, seq_len=T)
    m.load_state_dict(W.synth_head_weights(C.HeadConfig(seq_len=T), 4321))
    m.to("cuda")
    cls = torch.from_numpy(synth.cls_walk(500 + n + T, n, 768)).cuda()
    probs = m.infer_clip(cls, float(g[f"temp_{n}_{T}"])).cpu().numpy()
    m.close()
    ref = g[f"probs_{n}_{T}"]
    np.testing.assert_allclose(probs, ref, atol=1e-4)
    n_flip = int((probs.argmax(1) != ref.argmax(1)).sum())
    print(f"\ninfer_clip n={n} seq_len={T}: |dp|max {np.abs(probs - ref).max():.2e}, label flips {n_flip}")
    assert n_flip == 0


def test_model_bundle_round_trip_through_the_head(tmp_path):
    """model.pth + config.yaml + model_meta.json (workthreads.py:856-886) -> load_model_bundle (:372-451 rules) ->
    infer_file with the bundle's calibration temperature (:484) == the oracle on the same rows."""
    from cbas_amd import bundle as B, pipeline as P, h5io
    from cbas_amd.head import ClassifierLSTMDeltas
    from oracle import pipeline_oracle as PO
    names = [f"b{i}" for i in range(9)]
    enc_id = "facebook/dinov3-vitb16-pretrain-lvd1689m"
    hcfg = C.HeadConfig(lstm_hidden_size=96, seq_len=63)
    w = W.synth_head_weights(hcfg, 17)
    src = ClassifierLSTMDeltas(768, 9, seq_len=63, lstm_hidden_size=96)
    src.load_state_dict(w)
    B.save_model_bundle(str(tmp_path / "m"), src, names, "m", enc_id, temperature=1.4)
    head, meta = B.load_model_bundle(str(tmp_path / "m"), device="cuda", project_encoder=enc_id)
    assert head is not None and meta["hyperparameters"]["seq_len"] == 63
    cls = synth.cls_walk(9, 400, 768)
    path = str(tmp_path / "v_cls.h5")
    with h5io.ClsWriter(path, 768, {}) as wr:
        wr.append(cls)
    T = float(meta["calibration"]["temperature"])
    out = P.infer_file(path, head, "m", meta["hyperparameters"]["behaviors"], meta["hyperparameters"]["seq_len"],
                       device="cuda", temperature=T)
    assert out == str(tmp_path / "v_m_outputs.csv")
    got = np.loadtxt(out, delimiter=",", skiprows=1, dtype=np.float32)
    ref = PO.classify_cls(cls, w, 63, T)
    np.testing.assert_allclose(got, ref, atol=1e-4)
    assert (got.argmax(1) == ref.argmax(1)).all()
    head.close()


def test_encode_files_world1_equals_encode_file(tmp_path):
    """The multi-GPU driver with one rank writes the same bytes as encode_file / infer_file (the world-2/3 equality
    is checked on CPU ranks in tests/test_dist_encode_files.py)."""
    import hashlib, os, shutil
    from cbas_amd import dist as cdist, pipeline as P
    from cbas_amd.head import ClassifierLSTMDeltas
    cfg, enc = _enc("tiny", 16, (64, 64))
    hcfg = C.HeadConfig(in_features=cfg.hidden_size, out_features=4)
    head = ClassifierLSTMDeltas(cfg.hidden_size, 4)
    head.load_state_dict(W.synth_head_weights(hcfg, 3))
    head.to("cuda")
    names = ["a", "b", "c", "d"]
    try:
        sha = lambda p: hashlib.sha256(open(p, "rb").read()).hexdigest()      # noqa: E731
        a, b = tmp_path / "a", tmp_path / "b"
        a.mkdir(); b.mkdir()
        for i, n in enumerate((70, 600, 31)):
            np.save(str(a / f"v{i}.npy"), synth.cage_frames(60 + i, n, 64, 64))
            shutil.copy(str(a / f"v{i}.npy"), str(b / f"v{i}.npy"))
        P.set_project_stamp("enc-id")
        exp = []
        for i in range(3):
            h5 = P.encode_file(enc, str(a / f"v{i}.npy"))
            exp.append((sha(h5), sha(P.infer_file(h5, head, "ds", names, 31, device="cuda", temperature=0.7))))
        recs = cdist.encode_files([str(b / f"v{i}.npy") for i in range(3)], enc, head=head, dataset_name="ds",
                                  behaviors=names, temperature=0.7)
        assert [r["status"] for r in recs] == ["ok"] * 3 and [r["frames"] for r in recs] == [70, 600, 31]
        assert [(sha(r["cls_file"]), sha(r["csv_file"])) for r in recs] == exp
    finally:
        P.set_project_stamp(None)
        head.close()
        enc.close()


def test_encode_files_cli_single_process(tmp_path, capsys):
    """`python -m cbas_amd.encode_files --encoder <ckpt> --model-bundle <dir> videos...` (the N-GPU queue entry point)
    run in-process at world 1: checkpoint dir + bundle in, `_cls.h5` with the encoder stamp and `_outputs.csv` with the
    bundle's behaviours and calibration temperature out; `--dir` picks up only videos lacking an up-to-date `_cls.h5`."""
    import os
    from cbas_amd import bundle as B, encode_files as EF, h5io, pipeline as P
    from cbas_amd.head import ClassifierLSTMDeltas
    from oracle import pipeline_oracle as PO
    cfg = C.ViTConfig(hidden_size=768, intermediate_size=1536, num_hidden_layers=1, num_attention_heads=12, image_size=32)
    enc_w = W.synth_encoder_weights(cfg, 1234)
    ck = str(tmp_path / "ckpt")
    W.save_encoder_checkpoint(ck, cfg, enc_w)
    names = ["a", "b", "c"]
    hcfg = C.HeadConfig(out_features=3)
    hw = W.synth_head_weights(hcfg, 5)
    head = ClassifierLSTMDeltas(768, 3)
    head.load_state_dict(hw)
    B.save_model_bundle(str(tmp_path / "mymodel"), head, names, "mymodel", ck, temperature=1.3)
    rec = tmp_path / "rec"
    rec.mkdir()
    fr = synth.cage_frames(8, 90, 32, 32)
    np.save(str(rec / "v0.npy"), fr)
    try:
        rc = EF.main(["--encoder", ck, "--model-bundle", str(tmp_path / "mymodel"), "--max-batch", "32",
                      "--max-frame", "32", "32", str(rec / "v0.npy")])
        assert rc == 0
        with h5io.ClsReader(str(rec / "v0_cls.h5")) as r:
            assert r.shape == (90, 768) and r.attrs["encoder_model_identifier"] == ck
            rows = r.read(0, 90)
        got = np.loadtxt(str(rec / "v0_mymodel_outputs.csv"), delimiter=",", skiprows=1, dtype=np.float32)
        assert open(str(rec / "v0_mymodel_outputs.csv")).readline().strip() == "a,b,c"
        np.testing.assert_allclose(got, PO.classify_cls(rows, hw, 31, 1.3), atol=1e-4)
        ref = PO.encode_frames(fr, enc_w, cfg, batch=8)
        rel = np.linalg.norm(rows.astype(np.float32) - ref, axis=1) / np.linalg.norm(ref, axis=1)
        assert rel.max() < 1.5e-3
        # --dir semantics (startup_page.py:92-117): an encoded, correctly stamped video is not queued again
        assert EF.find_videos(str(rec), ck) == []
        os.rename(str(rec / "v0.npy"), str(rec / "v0.mp4"))
        os.rename(str(rec / "v0_cls.h5"), str(rec / "v0_cls.h5.bak"))
        assert EF.find_videos(str(rec), ck) == [str(rec / "v0.mp4")]
        os.rename(str(rec / "v0_cls.h5.bak"), str(rec / "v0_cls.h5"))
        assert EF.find_videos(str(rec), ck) == [] and EF.find_videos(str(rec), "another-encoder") == [str(rec / "v0.mp4")]
    finally:
        P.set_project_stamp(None)
        head.close()


def _dist_rank(rank, world, port, td, q, backend="gloo", device_rows=False):
    import os
    local = str(rank) if backend == "nccl" else "0"
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=local,
                      LOCAL_WORLD_SIZE=str(world))
    if device_rows:
        os.environ["CBAS_DIST_DEVICE_ROWS"] = "1"
    import faulthandler
    faulthandler.dump_traceback_later(100, exit=False)      # a rank still here after 100 s says where (first contact with RCCL)
    import torch.distributed as dist
    from cbas_amd import dist as cdist, pipeline as P
    from cbas_amd.head import ClassifierLSTMDeltas
    # gloo: two ranks share the one GPU of the test box, rows travel through host memory; nccl: one GPU per rank over RCCL
    cdist.init_from_env(backend)
    if backend == "nccl":
        torch.cuda.set_device(rank)
    cfg, enc = _enc("tiny", 16, (64, 64))
    head = ClassifierLSTMDeltas(cfg.hidden_size, 4)
    head.load_state_dict(W.synth_head_weights(C.HeadConfig(in_features=cfg.hidden_size, out_features=4), 3))
    head.to("cuda")
    P.set_project_stamp("enc-id")
    paths = [os.path.join(td, f"v{i}.npy") for i in range(5)]
    recs = cdist.encode_files(paths, enc, head=head, dataset_name="ds", behaviors=["a", "b", "c", "d"], temperature=0.7)
    if rank == 0:
        q.put(recs)
    dist.barrier()
    head.close()
    enc.close()
    dist.destroy_process_group()
    faulthandler.cancel_dump_traceback_later()


@pytest.mark.parametrize("backend,device_rows", [("gloo", False), ("gloo", True)])   # the RCCL form: tests/test_zz_rccl_two_gpus.py (needs two GPUs)
def test_encode_files_two_ranks_real_kernels(tmp_path, backend, device_rows):
    """Two processes driving the real encoder / head: rank 0's files are byte-identical to the single-process encode_file /
    infer_file results.  gloo: both ranks on the one GPU.  nccl (RCCL over xGMI, device-to-device point-to-point transfers
    from a receiver thread): needs two GPUs - skipped on the one-GPU test box, so THIS PATH HAS NOT RUN YET.
    device_rows (r5): the RCCL ranks' CONTROL FLOW on gloo (CBAS_DIST_DEVICE_ROWS=1: rows left in the session's device buffers,
    two sessions alternating, the next clip started while the sends drain) - everything of that path but the transport."""
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("the RCCL path needs two GPUs")
    import hashlib, os, shutil, socket
    import torch.multiprocessing as mp
    from cbas_amd import pipeline as P
    from cbas_amd.head import ClassifierLSTMDeltas
    sha = lambda p: hashlib.sha256(open(p, "rb").read()).hexdigest()          # noqa: E731
    a, b = tmp_path / "a", tmp_path / "b"
    a.mkdir(); b.mkdir()
    for i, n in enumerate((50, 0, 530, 31, 64)):
        np.save(str(a / f"v{i}.npy"), synth.cage_frames(70 + i, n, 64, 64))
        shutil.copy(str(a / f"v{i}.npy"), str(b / f"v{i}.npy"))
    cfg, enc = _enc("tiny", 16, (64, 64))
    head = ClassifierLSTMDeltas(cfg.hidden_size, 4)
    head.load_state_dict(W.synth_head_weights(C.HeadConfig(in_features=cfg.hidden_size, out_features=4), 3))
    head.to("cuda")
    P.set_project_stamp("enc-id")
    exp = []
    for i in range(5):
        h5 = P.encode_file(enc, str(a / f"v{i}.npy"))
        exp.append(None if h5 is None else (sha(h5), sha(P.infer_file(h5, head, "ds", ["a", "b", "c", "d"], 31, device="cuda", temperature=0.7))))
    P.set_project_stamp(None)
    head.close(); enc.close()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dist_rank, args=(r, 2, port, str(b), q, backend, device_rows)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        recs = q.get(timeout=240)
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
    finally:                                     # a rank that hangs (first contact with RCCL) must not outlive the test
        for p in procs:
            if p.is_alive():
                p.terminate()
                p.join(10)
    assert [r["status"] for r in recs] == ["ok", "empty", "ok", "ok", "ok"]
    for r, e in zip(recs, exp):
        if e is not None:
            assert (sha(r["cls_file"]), sha(r["csv_file"])) == e, r["path"]


@pytest.mark.parametrize("hw", [(160, 224), (224, 96), (64, 16), (16, 3088)])       # last: 1 + 193 table rows > ROPE_LDS_ROWS
def test_rope_table_in_lds_equals_global_table_and_the_oracle(hw):
    """The q|k|v GEMM of the ping-pong kernel rotates with the RoPE angles factorised by axis and held in LDS
    (gemm_epilogue.h); the [P][64] table in global memory is the fallback (one patch column, or more than
    ROPE_LDS_ROWS rows).  Same numbers either way: bit-identical outputs on non-square grids, and within the CLS
    tolerance of the CPU restatement of [tf] modeling_dinov3_vit.py:96-121,168-200."""
    from cbas_amd.encoder import DinoEncoder
    from oracle import pipeline_oracle as PO
    cfg = C.NAMED_VIT["vitb16"]
    w = W.synth_encoder_weights(cfg, 1234)
    fr = synth.cage_frames(5, 3, *hw)
    enc = DinoEncoder.from_weights(cfg, w, "cuda", max_batch=4, max_frame=hw)
    try:
        a16, a32 = enc.encode_u8(torch.from_numpy(fr).cuda())
        enc.debug_option("rope_lds", 0)
        try:
            b16, b32 = enc.encode_u8(torch.from_numpy(fr).cuda())
        finally:
            enc.debug_option("rope_lds", 1)
        torch.cuda.synchronize()
        assert torch.equal(a32, b32) and torch.equal(a16, b16)
        ref = PO.encode_frames(fr[:2], w, cfg, batch=2)
        rel = np.linalg.norm(a32[:2].cpu().numpy() - ref, axis=1) / np.linalg.norm(ref, axis=1)
        assert rel.max() < 1e-3, rel
    finally:
        enc.close()


@pytest.mark.parametrize("n,h,w", [(1, 16, 16), (3, 50, 70), (5, 33, 97), (2, 128, 48), (7, 80, 80), (4, 17, 200)])
def test_tiny_odd_frame_sizes_against_the_oracle(n, h, w):
    """Frame sizes that are not multiples of the patch (the conv drops the remainder, [tf]:82-89), single-row / single-column
    patch grids and ragged batch sizes: CLS rows against the CPU restatement of the whole encoder."""
    from cbas_amd.encoder import DinoEncoder
    from oracle import pipeline_oracle as PO
    cfg = C.VIT_TINY
    wts = W.synth_encoder_weights(cfg, 99)
    fr = synth.noise_frames(1000 + h * w + n, n, h, w)
    enc = DinoEncoder.from_weights(cfg, wts, "cuda", max_batch=8, max_frame=(h, w))
    try:
        c16, c32 = enc.encode_u8(torch.from_numpy(fr).cuda())
        got = c32.cpu().numpy().astype(np.float64)
        assert np.array_equal(c16.cpu().numpy(), c32.cpu().numpy().astype(np.float16))
    finally:
        enc.close()
    ref = PO.encode_frames(fr, wts, cfg, batch=n).astype(np.float64)
    rel = np.linalg.norm(got - ref, axis=1) / np.linalg.norm(ref, axis=1)
    assert rel.max() < 1e-3, rel

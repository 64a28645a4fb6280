"""Round-3 GPU tests (all through the C ABI):

* infer_file reads the `cls` dataset in 20 000-frame halo chunks like the reference (backend/cbas.py:497-525) - bit-identical
  to whole-clip classification - and accepts float32 / float64 datasets (:507-508);
* encode_infer_file (one pass through the fused session, frames DMA'd from the page-locked decode-ahead ring) writes the
  same bytes as encode_file followed by infer_file; an encode-only session (no head) equals the chunk loop;
* the head's workspace is allocated on demand;
* dist.encode_files with one rank: files byte-identical, writer threads off the encode loop.
"""
import hashlib
import os

import numpy as np
import pytest
import torch

from cbas_amd import config as C, weights as W, synth

pytestmark = pytest.mark.gpu


def _sha(p):
    return hashlib.sha256(open(p, "rb").read()).hexdigest()


def _tiny(max_batch=16, hw=(64, 64), ncls=5, seq_len=31):
    from cbas_amd.encoder import DinoEncoder
    from cbas_amd.head import ClassifierLSTMDeltas
    cfg = C.VIT_TINY
    enc = DinoEncoder.from_weights(cfg, W.synth_encoder_weights(cfg, 1234), "cuda", max_batch=max_batch, max_frame=hw)
    hcfg = C.HeadConfig(in_features=cfg.hidden_size, out_features=ncls, seq_len=seq_len)
    head = ClassifierLSTMDeltas(cfg.hidden_size, ncls, seq_len=seq_len)
    head.load_state_dict(W.synth_head_weights(hcfg, 7))
    head.to("cuda")
    return cfg, enc, head


@pytest.mark.parametrize("seq_len", [31, 63])
def test_chunked_infer_file_is_bit_identical_to_whole_clip_and_reads_any_float_dataset(tmp_path, monkeypatch, seq_len):
    from cbas_amd import h5io, pipeline as P
    from cbas_amd.head import ClassifierLSTMDeltas
    D, ncls, n = 128, 6, 5003
    hcfg = C.HeadConfig(in_features=D, out_features=ncls, seq_len=seq_len)
    head = ClassifierLSTMDeltas(D, ncls, seq_len=seq_len)
    head.load_state_dict(W.synth_head_weights(hcfg, 11))
    head.to("cuda")
    names = [f"b{i}" for i in range(ncls)]
    rows = synth.cls_walk(5, n, D)                                   # float16 random walk
    try:
        whole = head.infer_clip(torch.from_numpy(rows).cuda(), 0.9).cpu().numpy()
        files = {}
        for dt in ("f2", "f4", "f8"):
            p = str(tmp_path / f"clip_{dt}_cls.h5")
            with h5io.ClsWriter(p, D, {}, dtype=dt) as w:
                w.append(rows)
            files[dt] = p
        want = None
        for chunk in (20000, 1000, 64, 17):                          # 1 piece ... 295 pieces (17 < half the 63-frame window)
            monkeypatch.setattr(P, "INFER_CHUNK", chunk)
            with h5io.ClsReader(files["f2"]) as r:
                got = P.classify_cls_file(r, head, 0.9, "cuda", chunk=chunk)
            assert np.array_equal(got, whole), chunk
            out = P.infer_file(files["f2"], head, "m", names, seq_len, device="cuda", temperature=0.9)
            assert out == str(tmp_path / "clip_f2_m_outputs.csv")
            want = want or _sha(out)
            assert _sha(out) == want
        # the same numbers stored as float32 / float64: the reference's .float() of exactly representable values
        for dt in ("f4", "f8"):
            out = P.infer_file(files[dt], head, "m", names, seq_len, device="cuda", temperature=0.9)
            assert out is not None and _sha(out) == want, dt
        # genuinely float32 rows (not representable in half) go through the float32 entry point: against the fp32 oracle
        from oracle import pipeline_oracle as PO
        r32 = (rows.astype(np.float32) * 1.0001 + 3e-5).astype(np.float32)
        p32 = str(tmp_path / "foreign_cls.h5")
        with h5io.ClsWriter(p32, D, {}, dtype="f4") as w:
            w.append(r32)
        with h5io.ClsReader(p32) as r:
            got = P.classify_cls_file(r, head, 1.0, "cuda", chunk=1200)
        ref = PO.classify_cls(r32[:200], W.synth_head_weights(hcfg, 11), seq_len, 1.0)       # fp32 restatement of infer_file
        keep = 200 - seq_len // 2                                     # later windows of the prefix see rows the prefix lacks
        np.testing.assert_allclose(got[:keep], ref[:keep], atol=2e-5)
        assert (got[:keep].argmax(1) == ref[:keep].argmax(1)).all()
        assert np.isfinite(got).all() and np.abs(got.sum(1) - 1).max() < 1e-5
        # a dataset of the wrong width is the reference's shape error: infer_file returns None, never raises
        bad = str(tmp_path / "bad_cls.h5")
        with h5io.ClsWriter(bad, 64, {}) as w:
            w.append(np.zeros((10, 64), np.float16))
        assert P.infer_file(bad, head, "m", names, seq_len, device="cuda") is None
    finally:
        head.close()


def test_encode_infer_file_equals_encode_file_then_infer_file(tmp_path):
    from cbas_amd import pipeline as P
    cfg, enc, head = _tiny()
    names = list("abcde")
    try:
        P.set_project_stamp("enc-id")
        for i, n in enumerate((1, 30, 513, 1700)):                   # shorter than a window, ragged, > 3 chunks
            a, b = tmp_path / f"a{i}", tmp_path / f"b{i}"
            a.mkdir(); b.mkdir()
            fr = synth.cage_frames(80 + i, n, 64, 64)
            np.save(str(a / "v.npy"), fr)
            np.save(str(b / "v.npy"), fr)
            h5 = P.encode_file(enc, str(a / "v.npy"))
            csv = P.infer_file(h5, head, "ds", names, 31, device="cuda", temperature=0.8)
            seen = []
            h5b, csvb = P.encode_infer_file(enc, head, str(b / "v.npy"), "ds", names, temperature=0.8,
                                            progress_callback=seen.append)
            assert (os.path.basename(h5b), os.path.basename(csvb)) == ("v_cls.h5", "v_ds_outputs.csv")
            assert _sha(h5) == _sha(h5b) and _sha(csv) == _sha(csvb), n
            assert not os.path.exists(h5b + ".tmp")
            assert seen == [min(100.0, (min(k + 512, n) / n) * 100) for k in range(0, n, 512)]
        # a video without frames: nothing is written
        np.save(str(tmp_path / "empty.npy"), np.empty((0, 64, 64, 3), np.uint8))
        assert P.encode_infer_file(enc, head, str(tmp_path / "empty.npy"), "ds", names) == (None, None)
        assert not os.path.exists(str(tmp_path / "empty_cls.h5"))
        # a reader that fails mid-clip: the error propagates, nothing is left behind, the encoder and runner still work
        class Breaks(P.NpyFrameSource):
            def read_into(self, start, stop, out):
                if start >= 1024:
                    raise IOError("decode failed at 1024")
                super().read_into(start, stop, out)

            def read_channel_into(self, start, stop, channel, out):      # r4: the ring is filled with the green plane only
                if start >= 1024:
                    raise IOError("decode failed at 1024")
                super().read_channel_into(start, stop, channel, out)
        with pytest.raises(IOError, match="1024"):
            P.encode_infer_file(enc, head, str(tmp_path / "b3" / "v.npy"), "zz", names, reader=Breaks(str(tmp_path / "b3" / "v.npy")))
        assert not os.path.exists(str(tmp_path / "b3" / "v_zz_outputs.csv"))
        h5c, csvc = P.encode_infer_file(enc, head, str(tmp_path / "b3" / "v.npy"), "ds", names, temperature=0.8)
        assert _sha(h5c) == _sha(str(tmp_path / "a3" / "v_cls.h5")) and _sha(csvc) == _sha(str(tmp_path / "a3" / "v_ds_outputs.csv"))
    finally:
        P.set_project_stamp(None)
        head.close()
        enc.close()


def test_encode_only_session_and_runner_device_outputs(tmp_path):
    from cbas_amd import pipeline as P
    cfg, enc, head = _tiny()
    try:
        fr = synth.noise_frames(4, 700, 64, 64)
        np.save(str(tmp_path / "v.npy"), fr)
        ref = P.encode_rows(enc, str(tmp_path / "v.npy"))
        r0 = P.ClipRunner(enc, None)
        res = r0.run(str(tmp_path / "v.npy"))
        assert res.probs is None and np.array_equal(res.rows.view(np.uint16), ref.view(np.uint16))
        r0.close()
        r2 = P.ClipRunner(enc, head, 0.7, sessions=2)
        refp = head.infer_clip(torch.from_numpy(ref).cuda(), 0.7)
        outs = [r2.run(str(tmp_path / "v.npy"), device_out=True) for _ in range(3)]       # sessions alternate
        assert outs[0].on_device and outs[0].rows.data_ptr() == outs[2].rows.data_ptr() != outs[1].rows.data_ptr()
        for o in outs[1:]:
            assert torch.equal(o.rows, torch.from_numpy(ref).cuda()) and torch.equal(o.probs, refp)
        rows_view = outs[2].rows
        r2.close()
        del outs, rows_view                                           # views keep their session alive until dropped
    finally:
        head.close()
        enc.close()


def test_head_workspace_is_allocated_on_demand():
    from cbas_amd.head import ClassifierLSTMDeltas
    D = 768
    torch.cuda.synchronize()
    hcfg = C.HeadConfig(in_features=D, out_features=9)
    head = ClassifierLSTMDeltas(D, 9)
    head.load_state_dict(W.synth_head_weights(hcfg, 4321))
    head.to("cuda")
    free0 = torch.cuda.mem_get_info()[0]
    head._ensure()
    small = torch.from_numpy(synth.cls_walk(1, 200, D)).cuda()
    p_small = head.infer_clip(small).cpu()
    torch.cuda.synchronize()
    used_small = free0 - torch.cuda.mem_get_info()[0]
    big = torch.from_numpy(synth.cls_walk(2, 9000, D)).cuda()
    p_big = head.infer_clip(big).cpu()
    torch.cuda.synchronize()
    used_big = free0 - torch.cuda.mem_get_info()[0]
    print(f"head workspace: {used_small / 2**20:.0f} MiB after a 200-frame clip, {used_big / 2**20:.0f} MiB after a 9 000-frame clip")
    assert used_small < 200 * 2 ** 20 < used_big                      # was ~1 GiB at create in round 2
    assert torch.equal(head.infer_clip(small).cpu(), p_small)         # growing the workspace changed nothing
    assert torch.equal(head.infer_clip(big[:200]).cpu()[:150], head.infer_clip(big).cpu()[:150])
    x = torch.randn(3000, 31, D, device="cuda")
    lo, la = head(x)                                                  # explicit windows after sliding ones: another growth
    lo2, _ = head(x[:10])
    assert torch.equal(lo[:10], lo2)
    head.close()


def test_encode_files_world1_never_writes_on_the_encode_thread(tmp_path, monkeypatch):
    import threading
    from cbas_amd import dist as cdist, pipeline as P
    cfg, enc, head = _tiny()
    names = list("abcde")
    writers = set()
    real_h5, real_csv = P.write_cls_file, P.write_probs_csv
    monkeypatch.setattr(P, "write_cls_file", lambda *a, **k: (writers.add(threading.current_thread().name), real_h5(*a, **k))[1])
    monkeypatch.setattr(P, "write_probs_csv", lambda *a, **k: (writers.add(threading.current_thread().name), real_csv(*a, **k))[1])
    try:
        paths = []
        for i, n in enumerate((70, 0, 600, 31)):
            p = str(tmp_path / f"v{i}.npy")
            np.save(p, synth.cage_frames(60 + i, n, 64, 64) if n else np.empty((0, 64, 64, 3), np.uint8))
            paths.append(p)
        recs = cdist.encode_files(paths, enc, head=head, dataset_name="ds", behaviors=names, temperature=0.7)
        assert [r["status"] for r in recs] == ["ok", "empty", "ok", "ok"] and [r["frames"] for r in recs] == [70, 0, 600, 31]
        assert writers and threading.current_thread().name not in writers, writers
        assert all(w.startswith(("cbas-h5-writer", "cbas-csv")) for w in writers), writers
        assert all(os.path.exists(r["cls_file"]) and os.path.exists(r["csv_file"]) for r in recs if r["status"] == "ok")
    finally:
        head.close()
        enc.close()


def test_encode_files_survives_bad_clips_opened_ahead(tmp_path):
    """Clips are opened one ahead on a helper thread (ClipRunner.prepare): a missing file, a file that is not a video and a
    Motion-JPEG clip with a damaged frame fail at THEIR turn - as EncodeThread logs and skips (workthreads.py:334-336) - and
    the clips around them come out byte-identical to the same clips run alone."""
    from cbas_amd import dist as cdist, framesource as F, pipeline as P
    cfg, enc, head = _tiny()
    names = list("abcde")
    try:
        good = []
        for i, n in enumerate((300, 130, 45)):
            d = tmp_path / f"g{i}"
            d.mkdir()
            p = str(d / ("clip.avi" if i == 1 else "clip.npy"))
            fr = synth.cage_frames(80 + i, n, 64, 64)
            F.write_mjpeg_avi(p, fr, quality=90) if p.endswith(".avi") else np.save(p, fr)
            good.append(p)
        alone = [tuple(_sha(x) for x in P.encode_infer_file(enc, head, p, "ds", names)) for p in good]
        junk = str(tmp_path / "junk.avi")
        open(junk, "wb").write(b"RIFF" + bytes(range(200)))
        broken = str(tmp_path / "broken.avi")
        raw = bytearray(open(good[1], "rb").read())
        src = F.MJPEGAviSource(good[1])
        off, size = src._frames[100]                                    # a frame past the first pieces: its header destroyed
        src.close()
        raw[off:off + 64] = bytes(64)
        open(broken, "wb").write(bytes(raw))
        paths = [good[0], str(tmp_path / "missing.npy"), good[1], junk, broken, good[2]]
        recs = cdist.encode_files(paths, enc, head=head, dataset_name="ds", behaviors=names)
        assert [r["status"] for r in recs] == ["ok", "failed", "ok", "failed", "failed", "ok"], [r["status"] for r in recs]
        for r, want in zip([recs[0], recs[2], recs[5]], alone):
            assert (_sha(r["cls_file"]), _sha(r["csv_file"])) == want
        assert not os.path.exists(str(tmp_path / "broken_cls.h5")) and not os.path.exists(str(tmp_path / "broken_cls.h5.tmp"))
        # and the encoder is still usable afterwards
        again = tuple(_sha(x) for x in P.encode_infer_file(enc, head, good[0], "ds", names))
        assert again == alone[0]
    finally:
        head.close()
        enc.close()


def test_rows_leave_while_the_clip_runs(tmp_path, monkeypatch):
    """encode_file / encode_infer_file append the rows to the `.tmp` HDF5 file from a helper thread while the clip is encoded
    (cbas_fused_stream_rows): same bytes as the one-append form (CBAS_ROWS_STREAM=0), several appends actually happen, and a
    writer that fails in the middle fails the call the reference's way - exception out, no `.tmp`, no `_cls.h5`, encoder
    usable again."""
    from cbas_amd import h5io, pipeline as P
    cfg, enc, head = _tiny()
    try:
        p = str(tmp_path / "clip.npy")
        np.save(p, synth.cage_frames(21, 3000, 64, 64))
        appends = []
        real_append = h5io.ClsWriter.append

        def counting(self, rows):
            appends.append(len(rows))
            return real_append(self, rows)
        monkeypatch.setattr(h5io.ClsWriter, "append", counting)
        a = _sha(P.encode_file(enc, p))
        assert sum(appends) == 3000 and len(appends) >= 3, appends          # 512-row pieces while the clip ran + the tail
        h5, csv = P.encode_infer_file(enc, head, p, "m", list("abcde"))
        pair = (_sha(h5), _sha(csv))
        monkeypatch.setenv("CBAS_ROWS_STREAM", "0")
        appends.clear()
        assert _sha(P.encode_file(enc, p)) == a and appends == [3000]
        h5, csv = P.encode_infer_file(enc, head, p, "m", list("abcde"))
        assert (_sha(h5), _sha(csv)) == pair and pair[0] == a
        monkeypatch.delenv("CBAS_ROWS_STREAM")
        os.remove(h5)

        def failing(self, rows):
            if sum(appends) >= 1024:
                raise OSError("disk full")
            appends.append(len(rows))
            return real_append(self, rows)
        appends.clear()
        monkeypatch.setattr(h5io.ClsWriter, "append", failing)
        with pytest.raises(OSError, match="disk full"):
            P.encode_file(enc, p)
        assert not os.path.exists(str(tmp_path / "clip_cls.h5")) and not os.path.exists(str(tmp_path / "clip_cls.h5.tmp"))
        monkeypatch.setattr(h5io.ClsWriter, "append", real_append)
        assert _sha(P.encode_file(enc, p)) == a
    finally:
        head.close()
        enc.close()


def test_encode_file_on_a_compressed_video_equals_its_decoded_frames(tmp_path):
    """Motion-JPEG AVI (real decoder work on the decode-ahead thread, frames landing in the page-locked ring) against the same
    decoded frames stored as .npy: identical `_cls.h5` rows and identical probabilities."""
    from cbas_amd import framesource as F, h5io, pipeline as P
    cfg, enc, head = _tiny()
    try:
        fr = synth.cage_frames(12, 300, 64, 64)
        avi = str(tmp_path / "rec.avi")
        F.write_mjpeg_avi(avi, fr, quality=90)
        with F.MJPEGAviSource(avi) if hasattr(F.MJPEGAviSource, "__enter__") else _closing(F.MJPEGAviSource(avi)) as r:
            dec = r.get_batch(range(len(r)))
        assert dec.shape == fr.shape and not np.array_equal(dec, fr)          # lossy: the decoded frames are the reference point
        np.save(str(tmp_path / "dec.npy"), dec)
        a = P.encode_file(enc, avi)
        b = P.encode_file(enc, str(tmp_path / "dec.npy"))
        assert os.path.basename(a) == "rec_cls.h5"
        with h5io.ClsReader(a) as ra, h5io.ClsReader(b) as rb:
            assert ra.shape == (300, cfg.hidden_size)
            assert np.array_equal(ra.read(0, 300).view(np.uint16), rb.read(0, 300).view(np.uint16))
        h5c, csvc = P.encode_infer_file(enc, head, avi, "m", list("abcde"))
        csvd = P.infer_file(b, head, "m", list("abcde"), 31, device="cuda")
        assert _sha(csvc) == _sha(csvd)
    finally:
        head.close()
        enc.close()


def _closing(obj):
    import contextlib
    return contextlib.closing(obj)


# ------------------------------------------------------------------------------------------------------------------
# LayerNorm folded into the GEMMs around it (cbas_enc_debug_option "ln_fold"; csrc/gemm_epilogue.h, api_enc.hip run_blocks).
# Off by default (a measured wash with two batches in flight); when switched on it must meet the same bars.
# ------------------------------------------------------------------------------------------------------------------
def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b, axis=1) / np.linalg.norm(b, axis=1)


@pytest.mark.parametrize("name,cfgname,hw,batch", [("vitb16_224_noise", "vitb16", 224, 64), ("vitb16_256", "vitb16", 256, 8),
                                                   ("vitl16_224", "vitl16", 224, 8), ("vitl16_518", "vitl16", 518, 4)])
def test_ln_fold_meets_the_cls_bar_and_is_batch_invariant(golden_dir, name, cfgname, hw, batch):
    from cbas_amd.encoder import DinoEncoder
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    cfg = C.NAMED_VIT[cfgname]
    n = min(int(g["n"]), batch)
    mk = synth.noise_frames if str(g["kind"]) == "noise" else synth.cage_frames
    gold = mk(int(g["frame_seed"]), int(g["n"]), hw, hw)[:n]
    fill = synth.noise_frames(77, batch - n, hw, hw) if batch > n else gold[:0]
    fr = torch.from_numpy(np.concatenate([gold, fill])).cuda()
    enc = DinoEncoder.from_weights(cfg, W.synth_encoder_weights(cfg, 1234), "cuda", max_batch=batch, max_frame=(hw, hw))
    try:
        _, off32 = enc.encode_u8(fr)
        enc.debug_option("ln_fold", 1)
        on16, on32 = enc.encode_u8(fr)
        _, small32 = enc.encode_u8(fr[:n])                      # the golden frames alone: other tiles, other tile counts
        perm = torch.from_numpy(np.random.default_rng(1).permutation(batch)).cuda()
        _, perm32 = enc.encode_u8(fr[perm].contiguous())
        torch.cuda.synchronize()
        r_on, r_off = _rel(on32[:n].cpu().numpy(), g["cls"][:n]), _rel(off32[:n].cpu().numpy(), g["cls"][:n])
        d = _rel(on32.cpu().numpy(), off32.cpu().numpy())
        print(f"ln_fold {name} batch {batch}: CLS rel err vs reference golden {r_on.max():.3e} (separate LayerNorm kernels: "
              f"{r_off.max():.3e}); folded vs separate {d.max():.3e}")
        assert r_on.max() < 1e-3 and r_off.max() < 1e-3
        assert d.max() < 1.5e-3
        assert torch.equal(on32[:n], small32) and torch.equal(on32[perm], perm32)          # bit-exact batch invariance
        assert np.array_equal(on16.cpu().numpy(), on32.cpu().numpy().astype(np.float16))
        enc.debug_option("ln_fold", 0)
        _, again = enc.encode_u8(fr)
        torch.cuda.synchronize()
        assert torch.equal(again, off32)
    finally:
        enc.close()


def test_ln_fold_with_massive_activations_and_large_row_means():
    """What real checkpoints do and random weights do not: a few residual-stream channels hundreds of times larger than the rest,
    and rows whose MEAN is far from zero (the case where a one-pass variance would cancel catastrophically: the fold pools
    block sums of squares about block means instead).  4-layer ViT-B against the fp32 oracle, both settings."""
    from cbas_amd.encoder import DinoEncoder
    from oracle import pipeline_oracle as PO
    cfg = C.ViTConfig(hidden_size=768, intermediate_size=3072, num_hidden_layers=4, num_attention_heads=12, image_size=224)
    w = {k: v.copy() for k, v in W.synth_encoder_weights(cfg, 1234).items()}
    hot = [7, 100, 300, 640]
    w["model.layer.0.mlp.down_proj.bias"][hot] += 80.0 / np.abs(w["model.layer.0.layer_scale2.lambda1"][hot])
    w["model.layer.1.mlp.down_proj.bias"] += 25.0 / np.abs(w["model.layer.1.layer_scale2.lambda1"])      # every channel: row mean >> row std
    for nme in ("norm1", "norm2"):
        w[f"model.layer.1.{nme}.weight"][hot] *= 5.0
    w["embeddings.register_tokens"] = w["embeddings.register_tokens"] * 20.0
    fr = synth.cage_frames(5, 3, 224, 224)
    ref = PO.encode_frames(fr, w, cfg, batch=3)
    enc = DinoEncoder.from_weights(cfg, w, "cuda", max_batch=4, max_frame=(224, 224))
    try:
        for fold in (0, 1):
            enc.debug_option("ln_fold", fold)
            _, c32 = enc.encode_u8(torch.from_numpy(fr).cuda())
            torch.cuda.synchronize()
            r = _rel(c32.cpu().numpy(), ref)
            print(f"massive activations + large row means, ln_fold={fold}: CLS rel err {r.max():.3e}")
            assert np.isfinite(c32.cpu().numpy()).all() and r.max() < 1e-3, (fold, r)
    finally:
        enc.close()


def test_ln_fold_dinov2_and_unsupported_widths(golden_dir):
    """DINOv2-with-registers (key bias, LayerNorm eps 1e-6, learned position table) through the fold; a ViT-S (D = 384: no
    whole 256-column statistics block) silently keeps the LayerNorm kernels."""
    from cbas_amd.encoder import DinoEncoder
    g = np.load(os.path.join(golden_dir, "dinov2reg_b14.npz"))
    cfg = C.DINOV2_REG_B14
    enc = DinoEncoder.from_weights(cfg, W.synth_encoder_weights(cfg, 1234), "cuda", max_batch=4, max_frame=(224, 224))
    try:
        enc.debug_option("ln_fold", 1)
        _, c32 = enc.encode_u8(torch.from_numpy(synth.cage_frames(31, 4, 224, 224)).cuda())
        torch.cuda.synchronize()
        assert _rel(c32.cpu().numpy(), g["cls224"]).max() < 1e-3
    finally:
        enc.close()
    cfg = C.NAMED_VIT["vits16"]
    enc = DinoEncoder.from_weights(cfg, W.synth_encoder_weights(cfg, 1234), "cuda", max_batch=2, max_frame=(64, 64))
    try:
        fr = torch.from_numpy(synth.noise_frames(1, 2, 64, 64)).cuda()
        _, a = enc.encode_u8(fr)
        enc.debug_option("ln_fold", 1)
        _, b = enc.encode_u8(fr)
        torch.cuda.synchronize()
        assert torch.equal(a, b)
        with pytest.raises(RuntimeError, match="unknown debug option"):
            enc.debug_option("no_such_option", 1)
    finally:
        enc.close()


def test_file_paths_survive_a_workspace_rebuild_and_mixed_clip_sizes(tmp_path):
    """Frames larger than the encoder's workspace rebuild the handle once (DinoEncoder._fit_frame), which closes the open
    fused sessions; the cached ClipRunner must notice and carry on.  Then clips of other sizes, shorter and longer than the
    session's capacity, through the same runner: every result equals the slot loop's (CBAS_ENCODE_FILE_SLOTS=1)."""
    from cbas_amd import h5io, pipeline as P
    cfg, enc, head = _tiny(max_batch=16, hw=(32, 32))
    names = list("abcde")
    try:
        def rows_of(p):
            with h5io.ClsReader(p) as r:
                return r.read(0, r.shape[0])
        for i, (n, hw) in enumerate([(40, (32, 32)), (70, (64, 48)), (5000, (32, 32)), (33, (64, 48)), (9, (16, 80))]):
            d = tmp_path / f"c{i}"
            d.mkdir()
            fr = synth.noise_frames(200 + i, n, *hw)
            np.save(str(d / "v.npy"), fr)
            h5, csv = P.encode_infer_file(enc, head, str(d / "v.npy"), "m", names)
            got = rows_of(h5)
            os.environ["CBAS_ENCODE_FILE_SLOTS"] = "1"
            try:
                os.remove(h5)
                ref = rows_of(P.encode_file(enc, str(d / "v.npy")))
            finally:
                del os.environ["CBAS_ENCODE_FILE_SLOTS"]
            assert got.shape == (n, cfg.hidden_size) and np.array_equal(got.view(np.uint16), ref.view(np.uint16)), (n, hw)
            fused = rows_of(P.encode_file(enc, str(d / "v.npy")))          # the default encode_file: encode-only session
            assert np.array_equal(fused.view(np.uint16), ref.view(np.uint16))
            csv2 = P.infer_file(h5, head, "m2", names, 31, device="cuda")
            assert open(csv, "rb").read() == open(csv2, "rb").read()
        assert enc.max_frame[0] >= 64 and enc.max_frame[1] >= 80
    finally:
        head.close()
        enc.close()


# ------------------------------------------------------------------------------------------------------------------
# One clip split over the ranks (cbas_amd.dist.encode_infer_file_sharded; SURVEY section 8(e), last sentence) on the real kernels
# ------------------------------------------------------------------------------------------------------------------
def _sharded_rank(rank, world, port, td, q, backend):
    import os
    local = str(rank) if backend == "nccl" else "0"
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=local,
                      LOCAL_WORLD_SIZE=str(world))
    import torch.distributed as dist
    from cbas_amd import dist as cdist, pipeline as P
    cdist.init_from_env(backend)
    if backend == "nccl":
        torch.cuda.set_device(rank)
    cfg, enc, head = _tiny()
    P.set_project_stamp("enc-id")
    out = [cdist.encode_infer_file_sharded(os.path.join(td, f), enc, head=head, dataset_name="ds", behaviors=list("abcde"),
                                           temperature=0.8) for f in ("long.avi", "short.npy")]
    if rank == 0:
        q.put(out)
    dist.barrier()
    head.close()
    enc.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("backend,world", [("gloo", 2), ("gloo", 3)])      # RCCL form: tests/test_zz_rccl_two_gpus.py
def test_one_clip_split_over_ranks_real_kernels(tmp_path, backend, world):
    """Every rank decodes and encodes a frame range of the SAME video (Motion-JPEG AVI: random access into a compressed
    file; and a clip shorter than the head's halo on some ranks), halos are exchanged, rank 0 writes: byte-identical to
    encode_infer_file on one process.  gloo ranks share the test box's one GPU; the RCCL form needs two."""
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("the RCCL path needs two GPUs")
    import shutil
    import socket
    import torch.multiprocessing as mp
    from cbas_amd import framesource as F, pipeline as P
    a, b = tmp_path / "a", tmp_path / "b"
    a.mkdir()
    b.mkdir()
    F.write_mjpeg_avi(str(a / "long.avi"), synth.cage_frames(31, 700, 64, 64), quality=90)
    np.save(str(a / "short.npy"), synth.cage_frames(32, 20, 64, 64))
    for f in ("long.avi", "short.npy"):
        shutil.copy(str(a / f), str(b / f))
    cfg, enc, head = _tiny()
    P.set_project_stamp("enc-id")
    try:
        want = [tuple(_sha(x) for x in P.encode_infer_file(enc, head, str(a / f), "ds", list("abcde"), temperature=0.8))
                for f in ("long.avi", "short.npy")]
    finally:
        P.set_project_stamp(None)
        head.close()
        enc.close()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_rank, args=(r, world, port, str(b), q, backend)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        out = q.get(timeout=240)
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
    finally:                                     # a rank that hangs (first contact with RCCL) must not outlive the test
        for p in procs:
            if p.is_alive():
                p.terminate()
                p.join(10)
    assert [tuple(_sha(x) for x in o) for o in out] == want


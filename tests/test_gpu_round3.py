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

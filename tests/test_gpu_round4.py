"""Round-4 GPU tests: failure handling on the streamed-rows file path, the untested BASELINE configs on one
GPU (cfg3's 18 000-frame clip, cfg5's batch 128) and the green-only staging of RGB sources."""
import os
import threading

import numpy as np
import pytest
import torch

from cbas_amd import config as C, weights as W, synth

pytestmark = pytest.mark.gpu

NAMES = ["eating", "drinking", "rearing", "climbing", "digging", "nesting", "resting", "grooming", "exploring"]


def _tiny_pair(max_batch=16, hw=64):
    from cbas_amd.encoder import DinoEncoder
    from cbas_amd.head import ClassifierLSTMDeltas
    cfg = C.VIT_TINY
    enc = DinoEncoder.from_weights(cfg, W.synth_encoder_weights(cfg, 1234), "cuda", max_batch=max_batch, max_frame=(hw, hw))
    head = ClassifierLSTMDeltas(cfg.hidden_size, 9)
    head.load_state_dict(W.synth_head_weights(C.HeadConfig(in_features=cfg.hidden_size), 4321))
    head.to("cuda")
    return cfg, enc, head


class _FailingReader:
    """A frame source that raises in the middle of the clip (a truncated file, a decoder error): frames before `fail_at`
    are delivered, the read that touches `fail_at` raises."""

    def __init__(self, frames, fail_at):
        self.frames, self.fail_at = frames, fail_at

    def __len__(self):
        return len(self.frames)

    def get_batch(self, indices):
        idx = np.asarray(list(indices))
        if len(idx) and idx.max() >= self.fail_at:
            raise IOError(f"simulated decode error at frame {self.fail_at}")
        return self.frames[idx]


def _rows_threads():
    return [t for t in threading.enumerate() if t.name == "cbas-rows-out" and t.is_alive()]


def test_reader_failing_mid_clip_leaves_no_rows_thread_and_the_next_clip_is_identical(tmp_path):
    """encode_file / encode_infer_file write rows WHILE the clip runs (a 'cbas-rows-out' helper follows
    cbas_fused_rows_ready).  A reader that raises at chunk k, or a progress callback that raises, must stop that helper
    before the error propagates: it would otherwise keep polling a session the next clip resets and could still be inside
    libhdf5 while the .tmp file is removed.  After each failure: no helper thread alive, no .tmp file, and the next clip's
    files are byte-identical to the ones made before any failure."""
    from cbas_amd import pipeline as P
    cfg, enc, head = _tiny_pair()
    try:
        frames = synth.cage_frames(11, 1500, 64, 64)
        good = str(tmp_path / "good.npy")
        np.save(good, frames)
        h5, csv = P.encode_infer_file(enc, head, good, "ds", NAMES)
        want_h5, want_csv = open(h5, "rb").read(), open(csv, "rb").read()
        os.remove(h5); os.remove(csv)

        def progress_bomb(pct):
            if pct > 40.0:
                raise RuntimeError("progress callback failed")

        bad = str(tmp_path / "bad.npy")
        for how in ("reader_encode_file", "reader_encode_infer_file", "callback"):
            if how == "reader_encode_file":
                with pytest.raises(IOError):
                    P.encode_file(enc, bad, reader=_FailingReader(frames, 900))
            elif how == "reader_encode_infer_file":
                with pytest.raises(IOError):
                    P.encode_infer_file(enc, head, bad, "ds", NAMES, reader=_FailingReader(frames, 700))
            else:
                with pytest.raises(RuntimeError):
                    P.encode_file(enc, good, progress_callback=progress_bomb)
            assert _rows_threads() == [], how
            left = [f for f in os.listdir(tmp_path) if f.endswith(".tmp") or f.endswith("_cls.h5") or f.endswith(".csv")]
            assert left == [], (how, left)
            h5, csv = P.encode_infer_file(enc, head, good, "ds", NAMES)
            assert open(h5, "rb").read() == want_h5 and open(csv, "rb").read() == want_csv, how
            os.remove(h5); os.remove(csv)
    finally:
        enc.close(); head.close()


# ------------------------------------------------------------------------------------------------------------------
# Green-only staging of RGB sources
# ------------------------------------------------------------------------------------------------------------------
def test_green_only_staging_writes_the_same_bytes(tmp_path, monkeypatch):
    """`.npy` RGB clips: the decode-ahead thread keeps channel 1 while it fills the page-locked ring (a third of the staged
    and copied bytes).  Files are byte-identical to the r3 form (whole RGB frames staged, CBAS_STAGE_GREEN=0), and the ring
    pieces really are (n, H, W)."""
    from cbas_amd import pipeline as P
    cfg, enc, head = _tiny()
    try:
        frames = synth.cage_frames(12, 1100, 64, 64)
        assert not np.array_equal(frames[..., 0], frames[..., 1])          # the decoy channels differ from green
        vid = str(tmp_path / "v.npy")
        np.save(vid, frames)
        seen = []
        real = P._ChunkStream._decode

        def spy(self, i, end, r=None):
            out = real(self, i, end, r)
            if out[0] is not None:
                seen.append(out[0].shape)
            return out
        monkeypatch.setattr(P._ChunkStream, "_decode", spy)
        h5, csv = P.encode_infer_file(enc, head, vid, "ds", NAMES)
        a = open(h5, "rb").read(), open(csv, "rb").read()
        assert seen and all(len(sh) == 3 and sh[1:] == (64, 64) for sh in seen), seen[:3]
        os.remove(h5); os.remove(csv)
        seen.clear()
        monkeypatch.setenv("CBAS_STAGE_GREEN", "0")
        h5, csv = P.encode_infer_file(enc, head, vid, "ds", NAMES)
        b = open(h5, "rb").read(), open(csv, "rb").read()
        assert seen and all(len(sh) == 4 and sh[3] == 3 for sh in seen), seen[:3]
        assert a == b
    finally:
        enc.close(); head.close()


def _tiny():
    return _tiny_pair()


# ------------------------------------------------------------------------------------------------------------------
# BASELINE configs[2] (cfg3) on one GPU: the per-GPU workload is ONE 18 000-frame clip (30 min at 10 fps)
# ------------------------------------------------------------------------------------------------------------------
N3 = 18000


def _vitb_pair(max_batch=64):
    from cbas_amd.encoder import DinoEncoder
    from cbas_amd.head import ClassifierLSTMDeltas
    cfg = C.VIT_B16
    enc_w = W.synth_encoder_weights(cfg, 1234)
    head_w = W.synth_head_weights(C.HeadConfig(), 4321)
    enc = DinoEncoder.from_weights(cfg, enc_w, "cuda", max_batch=max_batch, max_frame=(224, 224))
    head = ClassifierLSTMDeltas(768, 9)
    head.load_state_dict(head_w)
    head.to("cuda")
    return cfg, enc, head, enc_w, head_w


def _cfg3_rank(rank, world, port, td, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      LOCAL_WORLD_SIZE=str(world))
    import torch.distributed as dist
    from cbas_amd import dist as cdist, pipeline as P
    cdist.init_from_env("gloo")                       # two ranks share the test box's one GPU; rows travel through host memory
    cfg, enc, head, _, _ = _vitb_pair()
    P.set_project_stamp("facebook/dinov3-vitb16-pretrain-lvd1689m")
    recs = cdist.encode_files([os.path.join(td, "clipA.npy"), os.path.join(td, "clipB.npy")], enc, head=head, dataset_name="ds",
                              behaviors=NAMES)
    if rank == 0:
        q.put(recs)
    dist.barrier()
    head.close(); enc.close()
    dist.destroy_process_group()


def test_cfg3_one_18000_frame_clip_through_the_file_paths(tmp_path):
    """ViT-B/16, one 18 000-frame 224 x 224 clip (2.7 GB of RGB frames as decord would hand them over):
    (1) encode_infer_file in one process - the files; (2) the same clip under two names through dist.encode_files on two
    gloo ranks sharing the GPU (one clip per rank, as cfg3 has one per GPU): rank 0's four files byte-identical to (1);
    (3) sampled oracle rows: the numpy ViT on 3 frames (1e-3) and the reference head semantics on sampled windows of the
    produced fp16 rows (every label)."""
    import hashlib
    import socket
    import torch.multiprocessing as mp
    from cbas_amd import pipeline as P, h5io
    from conftest import assert_labels_match
    from oracle import head_oracle as H
    from oracle import pipeline_oracle as PO
    root = "/dev/shm" if os.path.isdir("/dev/shm") else str(tmp_path)
    import tempfile
    td = tempfile.mkdtemp(prefix="cbas_cfg3_", dir=root)
    sha = lambda p: hashlib.sha256(open(p, "rb").read()).hexdigest()          # noqa: E731
    try:
        # a clip with temporal structure (16 noise 'scenes', cross-faded) built on the GPU, green on all three channels +- decoys
        gen = torch.Generator(device="cuda")
        gen.manual_seed(7)
        base = torch.randint(0, 256, (16, 224, 224), dtype=torch.uint8, device="cuda", generator=gen).float()
        path = os.path.join(td, "clipA.npy")
        mm = np.lib.format.open_memmap(path, mode="w+", dtype=np.uint8, shape=(N3, 224, 224, 3))
        picks = [0, 7777, N3 - 1]
        kept = {}
        for s in range(0, N3, 1000):
            e = min(s + 1000, N3)
            idx = torch.arange(s, e, device="cuda")
            seg, frac = (idx // 600) % 16, ((idx % 600).float() / 600.0)[:, None, None]
            g8 = (base[seg] * (1 - frac) + base[(seg + 1) % 16] * frac + (idx % 29)[:, None, None].float()).clamp(0, 255).to(torch.uint8)
            rgb = torch.stack([255 - g8, g8, g8.flip(2)], dim=3)              # decoys on channels 0 and 2
            mm[s:e] = rgb.cpu().numpy()
            for p_ in picks:
                if s <= p_ < e:
                    kept[p_] = g8[p_ - s].cpu().numpy()
        mm.flush()
        del mm
        os.link(path, os.path.join(td, "clipB.npy"))
        cfg, enc, head, enc_w, head_w = _vitb_pair()
        P.set_project_stamp("facebook/dinov3-vitb16-pretrain-lvd1689m")
        try:
            import time
            P.encode_infer_file(enc, head, path, "warm", NAMES)                # first call: sessions, ring, page cache
            t0 = time.perf_counter()
            h5, csv = P.encode_infer_file(enc, head, path, "ds", NAMES)
            dt = time.perf_counter() - t0
            print(f"[cfg3] encode_infer_file on one 18 000-frame clip: {dt:.3f} s = {N3 / dt:.0f} frames/s")
            want = (sha(h5), sha(csv))
            with h5io.ClsReader(h5) as r:
                assert r.shape == (N3, 768)
                rows = r.read(0, N3)
            probs = np.loadtxt(csv, delimiter=",", skiprows=1, dtype=np.float32)
            assert probs.shape == (N3, 9) and np.isfinite(probs).all() and np.abs(probs.sum(1) - 1).max() < 1e-5
            assert len(np.unique(probs.argmax(1))) >= 2
        finally:
            P.set_project_stamp(None)
            head.close(); enc.close()
        os.remove(h5); os.remove(csv)
        # (3) sampled oracle agreement
        fr = np.stack([kept[p_] for p_ in picks])
        ref = PO.encode_frames(np.repeat(fr[..., None], 3, axis=-1), enc_w, cfg, batch=3)
        got = rows[picks].astype(np.float32)
        rel = np.linalg.norm(got - ref, axis=1) / np.linalg.norm(ref, axis=1)
        assert rel.max() < 1.5e-3, rel.max()
        sel = np.r_[0:16, 9000:9016, N3 - 16:N3]
        idx = H.infer_windows(rows, 31)[sel]
        logits, _ = H.head_forward(rows.astype(np.float32)[idx], head_w, 31)
        n_mis, _ = assert_labels_match(probs[sel], H.softmax_T(logits, 1.0), 1e-4, margin=0.0)
        assert n_mis == 0
        # (2) two ranks, one 18 000-frame clip each
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_cfg3_rank, args=(r, 2, port, td, q)) for r in range(2)]
        for p in procs:
            p.start()
        try:
            recs = q.get(timeout=420)
            for p in procs:
                p.join(120)
                assert p.exitcode == 0
        finally:
            for p in procs:
                if p.is_alive():
                    p.terminate()
                    p.join(10)
        assert [r["status"] for r in recs] == ["ok", "ok"] and sorted(r["rank"] for r in recs) == [0, 1]
        assert all(r["frames"] == N3 for r in recs)
        for r in recs:
            assert (sha(r["cls_file"]), sha(r["csv_file"])) == want, r["path"]
    finally:
        import shutil
        shutil.rmtree(td, ignore_errors=True)


# ------------------------------------------------------------------------------------------------------------------
# BASELINE configs[4] (cfg5) on one GPU: MX-fp8 at batch 128
# ------------------------------------------------------------------------------------------------------------------
def test_cfg5_fp8_batch_128(golden_dir):
    """precision 2 with max_batch = 128 (configs[4]'s batch; r3 only ever built fp8 encoders with max_batch = 8 under -m gpu):
    a 128-frame batch gives, frame for frame, the bits of the same frames in batches of 8 (block scales are per row, tiles
    do not mix rows), permuted batches give permuted rows, and the golden frames' CLS sits where the MX restatement
    (oracle/mx_oracle.py) and the fp32 reference say it should."""
    from cbas_amd.encoder import DinoEncoder
    from oracle import mx_oracle as MX
    cfg = C.VIT_B16
    w = W.synth_encoder_weights(cfg, 1234)
    g = np.load(os.path.join(golden_dir, "vitb16_224_noise.npz"))
    gold = synth.noise_frames(int(g["frame_seed"]), int(g["n"]), 224, 224)[:4]
    fr = np.concatenate([gold, synth.cage_frames(21, 124, 224, 224)])
    big = DinoEncoder.from_weights(cfg, w, "cuda", max_batch=128, max_frame=(224, 224), precision=2)
    small = DinoEncoder.from_weights(cfg, w, "cuda", max_batch=8, max_frame=(224, 224), precision=2)
    try:
        fd = torch.from_numpy(fr).cuda()
        a16, a32 = big.encode_u8(fd)
        b16, _ = small.encode_u8(fd, want_f32=False)          # 16 passes of 8 frames
        perm = np.random.default_rng(0).permutation(128)
        p16, _ = big.encode_u8(torch.from_numpy(fr[perm]).cuda(), want_f32=False)
        torch.cuda.synchronize()
        assert torch.isfinite(a32).all()
        assert torch.equal(a16, b16)
        assert torch.equal(a16[perm], p16)
        # host-streamed, two batches in flight, 128 frames per submission
        big.submit_host(0, fr)
        big.submit_host(1, fr[perm])
        h0, _ = big.wait(0)
        h1, _ = big.wait(1)
        assert np.array_equal(h0, a16.cpu().numpy()) and np.array_equal(h1, a16.cpu().numpy()[perm])
        got = a32[:4].cpu().numpy().astype(np.float64)
    finally:
        big.close(); small.close()

    def rel(x, y):
        return (np.linalg.norm(x - y, axis=1) / np.linalg.norm(y, axis=1)).max()
    ref32 = g["cls"][:4].astype(np.float64)
    emu = MX.encode_frames_mx(gold, w, cfg).astype(np.float64)
    r_emu, r_ref, r_emu_ref = rel(got, emu), rel(got, ref32), rel(emu, ref32)
    print(f"[cfg5 batch 128] CLS rel err: GPU vs MX restatement {r_emu:.3e} | GPU vs fp32 reference {r_ref:.3e} | restatement vs "
          f"reference {r_emu_ref:.3e}")
    assert r_emu < 8e-2 and r_ref < 1.0e-1 and r_ref < 1.5 * r_emu_ref + 1e-2


# ------------------------------------------------------------------------------------------------------------------
# encode_files(local_writes=True): every rank writes its own clips' files
# ------------------------------------------------------------------------------------------------------------------
def _local_writes_rank(rank, world, port, td, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      LOCAL_WORLD_SIZE=str(world))
    import torch.distributed as dist
    from cbas_amd import dist as cdist, pipeline as P
    cdist.init_from_env("gloo")
    cfg, enc, head = _tiny_pair()
    P.set_project_stamp("enc-id")
    paths = [os.path.join(td, f"v{i}.npy") for i in range(5)]
    recs = cdist.encode_files(paths, enc, head=head, dataset_name="ds", behaviors=NAMES, temperature=0.7, local_writes=True)
    if rank == 0:
        q.put(recs)
    dist.barrier()
    head.close(); enc.close()
    dist.destroy_process_group()


def test_encode_files_local_writes_two_ranks_real_kernels(tmp_path):
    """Two gloo ranks on the real kernels, each writing the files of the clips it encoded: the same bytes as the
    single-process encode_file / infer_file, records assembled on rank 0 from the ranks' outcome reports."""
    import hashlib, shutil, socket
    import torch.multiprocessing as mp
    from cbas_amd import pipeline as P
    sha = lambda p: hashlib.sha256(open(p, "rb").read()).hexdigest()          # noqa: E731
    a, b = tmp_path / "a", tmp_path / "b"
    a.mkdir(); b.mkdir()
    for i, n in enumerate((50, 0, 530, 31, 64)):
        np.save(str(a / f"v{i}.npy"), synth.cage_frames(70 + i, n, 64, 64))
        shutil.copy(str(a / f"v{i}.npy"), str(b / f"v{i}.npy"))
    cfg, enc, head = _tiny_pair()
    P.set_project_stamp("enc-id")
    exp = []
    try:
        for i in range(5):
            h5 = P.encode_file(enc, str(a / f"v{i}.npy"))
            exp.append(None if h5 is None else (sha(h5), sha(P.infer_file(h5, head, "ds", NAMES, 31, device="cuda", temperature=0.7))))
    finally:
        P.set_project_stamp(None)
        head.close(); enc.close()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_local_writes_rank, args=(r, 2, port, str(b), q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        recs = q.get(timeout=240)
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
                p.join(10)
    assert [r["status"] for r in recs] == ["ok", "empty", "ok", "ok", "ok"]
    assert {r["rank"] for r in recs if r["status"] == "ok"} <= {0, 1}
    for r, e in zip(recs, exp):
        if e is not None:
            assert (sha(r["cls_file"]), sha(r["csv_file"])) == e, r["path"]


@pytest.mark.parametrize("precision", [0, 4])
def test_head_beside_the_encoder_is_bit_stable(precision):
    """The file path runs the head on its own stream while the encoder's lanes run the next batches.  A head inference
    beside encoder passes (cut after attention, after the MLP, and whole) and beside a dense 32x32x16 MFMA loop must
    return the bytes it returns on an idle device: round 4's file-path soak found neighbours beside which the head's
    expand kernel did not (scripts/head_beside_encoder.py tells the story); this is the guard."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "head_beside_encoder", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "head_beside_encoder.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    res = mod.run(precision, 1.5, stages=((0, 3), (0, 7), (-1, -1), ("mfma", 0)))
    print(res)
    for st in res:
        assert st["head_runs"] > 10, st
        assert st["head_runs_differing"] == 0, st


def test_training_beside_the_encoder_is_bit_stable():
    """CBAS starts TrainingThread, EncodeThread and ClassificationThread together on one device
    (backend/workthreads.py:1256-1267): a 40-step training run on its own stream beside precision-4 encoder passes, beside
    default-precision passes and beside the dense 32x32x16 MFMA loop must produce the losses and the trained weights of the
    idle-device run, bit for bit (scripts/train_beside_encoder.py).  The training kernels are compiled without packed-fp32
    instructions (build.py) and asmcheck bans the instruction form round 5 identified from every kernel."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "train_beside_encoder", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "train_beside_encoder.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    res = mod.run(4.0)
    print(res)
    for nb in res:
        assert nb["runs"] >= (1 if nb["neighbour"] == "mfma" else 3), nb      # the MFMA loop leaves training little of the device
        assert nb["runs_differing"] == 0, nb


@pytest.mark.parametrize("precision", [0, 4])
def test_pipelined_clips_write_the_bytes_of_clips_run_alone(tmp_path, precision):
    """A short form of scripts/soak_files.py (the test that found round 4's head / neighbour interference) at the real
    model size: ragged clips through `encode_files` - decode-ahead, both encoder lanes, the head on its own stream beside
    them, clip n + 1 starting under clip n's tail - must write the files the same clips write when run alone."""
    import hashlib
    from cbas_amd import dist as cdist, pipeline as P
    from cbas_amd.encoder import DinoEncoder
    from cbas_amd.head import ClassifierLSTMDeltas
    cfg = C.VIT_B16
    enc = DinoEncoder.from_weights(cfg, W.synth_encoder_weights(cfg, 1234), "cuda", max_batch=64, max_frame=(224, 224), precision=precision)
    head = ClassifierLSTMDeltas(768, 9)
    head.load_state_dict(W.synth_head_weights(C.HeadConfig(), 4321))
    head.to("cuda")
    sha = lambda p: hashlib.sha256(open(p, "rb").read()).hexdigest()
    try:
        base = synth.cage_frames(3, 300, 224, 224)
        lengths = [700, 1, 513, 31, 1030, 128]
        protos = []
        for k, n in enumerate(lengths):
            p = tmp_path / f"proto{k}.npy"
            np.save(p, base[(np.arange(n) * (k + 1)) % 300])
            protos.append(str(p))
        alone = []
        for k, p in enumerate(protos):
            d = tmp_path / f"alone{k}"
            d.mkdir()
            q = d / os.path.basename(p)
            os.symlink(p, q)
            h5, csv = P.encode_infer_file(enc, head, str(q), "soak", NAMES)
            alone.append((sha(h5), sha(csv)))
        paths = []
        for r in range(3):
            for k in np.random.default_rng(r).permutation(len(protos)):
                d = tmp_path / f"r{r}_{k}"
                d.mkdir()
                q = d / os.path.basename(protos[k])
                os.symlink(protos[k], q)
                paths.append((int(k), str(q)))
        recs = cdist.encode_files([q for _, q in paths], enc, head=head, dataset_name="soak", behaviors=NAMES)
        for (k, _q), r in zip(paths, recs):
            assert r["status"] == "ok", r
            assert (sha(r["cls_file"]), sha(r["csv_file"])) == alone[k], (precision, k, lengths[k])
    finally:
        head.close()
        enc.close()

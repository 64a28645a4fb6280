"""cbas_amd.dist.encode_files on CPU ranks (gloo): the multi-GPU product path - clip sharding, per-rank encode (+
classify), gather to rank 0, rank 0 writes `_cls.h5` / `_outputs.csv` in clip order - must produce files byte for byte
equal to the single-process `encode_file` / `infer_file` results.  The encoder and head are CPU stand-ins with the
product classes' interfaces (there is no CPU path of the real ones); the GPU suite runs the real ones at world 1."""
import hashlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cbas_amd import dist as cdist, pipeline as P
from cbas_amd.encoder import DinoEncoder
from cbas_amd.head import ClassifierLSTMDeltas

D, C_ = 64, 4
NAMES = ["rest", "groom", "eat", "dig"]


class StubEncoder(DinoEncoder):
    """DinoEncoder's streaming interface (submit_host / wait, 3 slots) over a deterministic per-frame function."""

    def __init__(self):                                   # no library, no GPU
        self.config = type("Cfg", (), {"hidden_size": D})()
        self.max_batch = 16
        self.device = torch.device("cpu")
        self._slots = {}
        self._slot_n = {}

    def submit_host(self, slot, frames, channel=1):
        assert slot not in self._slots
        g = frames[:, :, :, channel].astype(np.float32)
        if g.mean() > 250:
            raise RuntimeError("decoder produced a saturated frame")          # lets a test inject a failing clip
        feat = np.stack([g.mean((1, 2)), g.std((1, 2)), g[:, 0, 0], g[:, -1, -1]], 1) / 255.0
        proj = np.cos(np.arange(D, dtype=np.float32)[None, :] * (1.0 + feat @ np.array([1.0, 2.0, 3.0, 5.0], np.float32))[:, None])
        self._slots[slot] = proj.astype(np.float16)
        self._slot_n[slot] = frames.shape[0]

    def wait(self, slot, want_f32=False):
        self._slot_n.pop(slot)
        return self._slots.pop(slot), None

    def close(self):
        pass


class StubHead(ClassifierLSTMDeltas):
    def __init__(self):
        self.in_features, self.out_features, self.seq_len = D, C_, 31

    def to(self, device):
        return self

    def infer_clip(self, cls_f16, temperature=1.0, want_logits=False):
        x = cls_f16.float()
        n = x.shape[0]
        idx = (torch.arange(n)[:, None] + torch.arange(-15, 16)[None, :]).clamp(0, n - 1)      # replicate-padded windows
        win = x[idx].mean(1)
        return torch.softmax(win[:, :C_] * 4.0 / max(1e-3, temperature), dim=1)

    def close(self):
        pass


def _make_clips(td):
    rng = np.random.default_rng(5)
    lengths = [40, 700, 0, 33, 513, 90, 17]              # ragged, an empty video, > one 512-frame chunk
    paths = []
    for i, n in enumerate(lengths):
        fr = rng.integers(0, 200, (n, 8, 8, 3), dtype=np.uint8)
        p = os.path.join(td, f"clip{i}.npy")
        np.save(p, fr)
        paths.append(p)
    bad = os.path.join(td, "clip_bad.npy")               # a clip whose encode raises
    np.save(bad, np.full((20, 8, 8, 3), 255, np.uint8))
    paths.insert(4, bad)
    return paths


def _sha(p):
    return hashlib.sha256(open(p, "rb").read()).hexdigest()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, td, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    cdist.init_from_env("gloo")
    P.set_project_stamp("facebook/dinov3-vitb16-pretrain-lvd1689m")
    paths = sorted(p for p in (os.path.join(td, f) for f in os.listdir(td)) if p.endswith(".npy"))
    recs = cdist.encode_files(paths, StubEncoder(), head=StubHead(), dataset_name="gold", behaviors=NAMES, temperature=0.9)
    if rank == 0:
        q.put(recs)
    else:
        assert recs is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_encode_files_matches_single_process_byte_for_byte(tmp_path, world):
    ref_dir, run_dir = str(tmp_path / "ref"), str(tmp_path / "run")
    os.makedirs(ref_dir)
    os.makedirs(run_dir)
    ref_paths = _make_clips(ref_dir)
    _make_clips(run_dir)
    # single-process reference: the product's encode_file / infer_file on each clip
    P.set_project_stamp("facebook/dinov3-vitb16-pretrain-lvd1689m")
    enc, head = StubEncoder(), StubHead()
    expected = {}
    for p in sorted(ref_paths):
        try:
            out = P.encode_file(enc, p)
        except RuntimeError:
            enc = StubEncoder()
            expected[os.path.basename(p)] = ("failed", None, None)
            continue
        if out is None:
            expected[os.path.basename(p)] = ("empty", None, None)
            continue
        csv = P.infer_file(out, head, "gold", NAMES, 31, device="cpu", temperature=0.9)
        assert csv is not None
        expected[os.path.basename(p)] = ("ok", _sha(out), _sha(csv))
    P.set_project_stamp(None)

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, run_dir, q)) for r in range(world)]
    for p in procs:
        p.start()
    recs = q.get(timeout=180)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert [os.path.basename(r["path"]) for r in recs] == sorted(expected)           # clip order
    for r in recs:
        status, h5_sha, csv_sha = expected[os.path.basename(r["path"])]
        assert r["status"] == status, r
        if status == "ok":
            assert _sha(r["cls_file"]) == h5_sha and _sha(r["csv_file"]) == csv_sha, r["path"]
            assert not os.path.exists(r["cls_file"] + ".tmp")
        else:
            assert r["cls_file"] is None and r["csv_file"] is None
            assert not os.path.exists(os.path.splitext(r["path"])[0] + "_cls.h5")
    assert sum(r["frames"] for r in recs) == 40 + 700 + 33 + 513 + 90 + 17

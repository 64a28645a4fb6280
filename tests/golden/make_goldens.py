#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE itself.

Runs only in the build container (needs /root/reference and the ``transformers`` package); the
GPU box never sees either — it gets the small ``.npz`` / ``.csv`` fixtures this script writes.

What is imported from the reference (nothing is copied):
  * ``backend/classifier_head.py``  -> ``ClassifierLSTMDeltas``            (head goldens)
  * ``backend/cbas.py``             -> ``DinoEncoder``, ``encode_file``, ``infer_file``
    (needs stub modules for the absent ``cv2`` / ``decord`` / ``h5py``; functional fakes for
    ``decord.VideoReader`` and ``h5py.File`` are supplied here, backed by numpy)
  * ``transformers`` ``DINOv3ViTModel`` (the third-party package holding the ViT arithmetic)

Weights and input frames come from the counter-based generators in ``cbas_amd.weights`` /
``cbas_amd.synth`` so they never need committing.

Usage:  python tests/golden/make_goldens.py [--only NAME] [--out tests/golden]
"""
from __future__ import annotations

import argparse
import hashlib
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

REF = "/root/reference"

import torch  # noqa: E402

from cbas_amd import config as C  # noqa: E402
from cbas_amd import weights as W  # noqa: E402
from cbas_amd import synth  # noqa: E402

torch.manual_seed(0)
torch.set_grad_enabled(False)

BEHAVIORS = ["eating", "drinking", "rearing", "climbing", "digging", "nesting", "resting", "grooming",
             "exploring"]     # reference models/JonesLabModel/config.yaml:1-10

ENC_SEED, HEAD_SEED = 1234, 4321


# --------------------------------------------------------------------------------------------
# numpy-backed fakes for the reference's missing IO dependencies
# --------------------------------------------------------------------------------------------
class _FakeAttrs(dict):
    pass


class _FakeDataset:
    def __init__(self, shape, dtype):
        self._a = np.zeros(shape, dtype=dtype)

    @property
    def shape(self):
        return self._a.shape

    def resize(self, size, axis=0):
        new = np.zeros((size,) + self._a.shape[1:], dtype=self._a.dtype)
        n = min(size, self._a.shape[0])
        new[:n] = self._a[:n]
        self._a = new

    def __setitem__(self, k, v):
        self._a[k] = v          # casts to the dataset dtype (f2), like h5py

    def __getitem__(self, k):
        return self._a[k]

    def __len__(self):
        return self._a.shape[0]


_FAKE_FS: dict = {}


class _FakeH5File:
    def __init__(self, path, mode="r"):
        self.path, self.mode = path, mode
        if "w" in mode:
            _FAKE_FS[path] = {"attrs": _FakeAttrs(), "dsets": {}}
            open(path, "wb").close()         # so os.replace / os.path.exists behave
        elif path not in _FAKE_FS:
            raise OSError(f"no such fake h5 file: {path}")
        self._n = _FAKE_FS[path]
        self.attrs = self._n["attrs"]

    def create_dataset(self, name, shape, maxshape=None, dtype="f4", chunks=None):
        d = _FakeDataset(shape, np.dtype(dtype))
        d.maxshape, d.chunks = maxshape, chunks
        self._n["dsets"][name] = d
        return d

    def __getitem__(self, name):
        return self._n["dsets"][name]

    def flush(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class _FakeBatch:
    def __init__(self, a):
        self._a = a

    def asnumpy(self):
        return self._a


class _FakeVideoReader:
    """decord.VideoReader(path, ctx) fake: 'path' indexes a registry of in-memory clips."""
    CLIPS: dict = {}

    def __init__(self, path, ctx=None):
        self._f = self.CLIPS[path]

    def __len__(self):
        return self._f.shape[0]

    def get_batch(self, idx):
        return _FakeBatch(self._f[list(idx)])


def import_reference():
    for name in ("cv2", "decord", "h5py"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["cv2"].VideoCapture = object
    sys.modules["decord"].VideoReader = _FakeVideoReader
    sys.modules["decord"].cpu = lambda i=0: None
    sys.modules["h5py"].File = _FakeH5File
    for p in (REF, os.path.join(REF, "backend")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import classifier_head  # noqa
    import cbas  # noqa
    return cbas, classifier_head


def _real_replace(src, dst):
    if src in _FAKE_FS:
        _FAKE_FS[dst] = _FAKE_FS.pop(src)
    _os_replace(src, dst)


_os_replace = os.replace


# --------------------------------------------------------------------------------------------
def hf_model(cfg: C.ViTConfig, weights):
    from transformers import DINOv3ViTConfig, DINOv3ViTModel
    hcfg = DINOv3ViTConfig(
        patch_size=cfg.patch_size, image_size=cfg.image_size, hidden_size=cfg.hidden_size,
        intermediate_size=cfg.intermediate_size, num_hidden_layers=cfg.num_hidden_layers,
        num_attention_heads=cfg.num_attention_heads, num_register_tokens=cfg.num_register_tokens,
        layer_norm_eps=cfg.layer_norm_eps, rope_theta=cfg.rope_theta)
    hcfg._attn_implementation = "eager"
    m = DINOv3ViTModel(hcfg).eval()
    sd = {k: torch.from_numpy(v.copy()) for k, v in weights.items()}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("inv_freq" in k for k in missing), missing
    return m


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def g_tiny(out):
    """Tiny ViT with per-stage taps (via HF hidden_states) — kernel-by-kernel bring-up."""
    cfg = C.VIT_TINY
    w = W.synth_encoder_weights(cfg, ENC_SEED)
    m = hf_model(cfg, w)
    frames = synth.cage_frames(7, 4, 64, 64)
    g = torch.from_numpy(frames[:, :, :, 1] / 255.0).float()
    px = g.unsqueeze(1).repeat(1, 3, 1, 1)
    o = m(px, output_hidden_states=True)
    hs = [h.numpy() for h in o.hidden_states]
    np.savez_compressed(os.path.join(out, "vit_tiny.npz"),
                        frames_sha=sha(frames), n=4, height=64, width=64, frame_seed=7,
                        embeddings=hs[0], layer0=hs[1], layer1=hs[2],
                        last_hidden=o.last_hidden_state.numpy())


def _cls_golden(out, name, cfg, n, H, W_, frame_seed, via_wrapper=False, kind="cage"):
    w = W.synth_encoder_weights(cfg, ENC_SEED)
    frames = (synth.cage_frames if kind == "cage" else synth.noise_frames)(frame_seed, n, H, W_)
    g = torch.from_numpy(frames[:, :, :, 1] / 255.0).float()           # cbas.py:431
    if via_wrapper:
        cbas, _ = import_reference()
        with tempfile.TemporaryDirectory() as td:
            m = hf_model(cfg, w)
            m.save_pretrained(td)
            enc = cbas.DinoEncoder(td, device="cpu")                   # reference wrapper, cbas.py:650-677
            cls = enc(g.unsqueeze(1)).squeeze(1).numpy()               # cbas.py:435-436
    else:
        m = hf_model(cfg, w)
        cls = torch.cat([m(g[i:i + 2].unsqueeze(1).repeat(1, 3, 1, 1)).last_hidden_state[:, 0]
                         for i in range(0, n, 2)]).numpy()
    np.savez_compressed(os.path.join(out, f"{name}.npz"), cls=cls.astype(np.float32),
                        frames_sha=sha(frames), n=n, height=H, width=W_, frame_seed=frame_seed, kind=kind)
    print(name, cls.shape, float(np.abs(cls).mean()))


def g_vits(out):
    _cls_golden(out, "vits16_224", C.VIT_S16, 8, 224, 224, 11)


def g_vitb(out):
    _cls_golden(out, "vitb16_224", C.VIT_B16, 8, 224, 224, 12, via_wrapper=True)


def g_vitb_noise(out):
    _cls_golden(out, "vitb16_224_noise", C.VIT_B16, 4, 224, 224, 0, kind="noise")


def g_vitb256(out):
    _cls_golden(out, "vitb16_256", C.VIT_B16, 4, 256, 256, 13)


def g_vitl(out):
    _cls_golden(out, "vitl16_224", C.VIT_L16, 4, 224, 224, 14)


def g_vitl518(out):
    _cls_golden(out, "vitl16_518", C.VIT_L16, 2, 518, 518, 15)


def ref_head(classifier_head, hcfg: C.HeadConfig, hw):
    m = classifier_head.ClassifierLSTMDeltas(
        in_features=hcfg.in_features, out_features=hcfg.out_features, seq_len=hcfg.seq_len,
        use_acceleration=hcfg.use_acceleration,
        lstm_hidden_size=hcfg.lstm_hidden_size, lstm_layers=hcfg.lstm_layers).eval()
    res = m.load_state_dict({k: torch.from_numpy(np.asarray(v).copy()) for k, v in hw.items()}, strict=True)
    return m


def g_head(out):
    _, classifier_head = import_reference()
    for tag, h, C_, I, nl in (("h64", 64, 9, 768, 1), ("h128", 128, 5, 768, 1), ("h64_d384", 64, 9, 384, 1),
                              ("h64_l2", 64, 9, 768, 2)):           # stacked BiLSTM (sweep_runner.py:108 lists [1, 2])
        hcfg = C.HeadConfig(in_features=I, out_features=C_, lstm_hidden_size=h, lstm_layers=nl)
        hw = W.synth_head_weights(hcfg, HEAD_SEED)
        m = ref_head(classifier_head, hcfg, hw)
        seq = synth.cls_walk(21, 64 + 30, I).astype(np.float32)
        x = np.stack([seq[i:i + 31] for i in range(64)])
        logits, latent = m(torch.from_numpy(x))
        np.savez_compressed(os.path.join(out, f"head_{tag}.npz"), logits=logits.numpy(), latent=latent.numpy(),
                            x_sha=sha(x), walk_seed=21)
        print("head", tag, logits.shape, latent.shape)


def g_head_variants(out):
    """Round 2: the constructor arguments round 1 left out - the sweep's sequence lengths (sweep_runner.py:110 lists
    [31, 63, 95]), hidden sizes other than 64 / 128 (the loader infers any size: workthreads.py:418-421) and
    use_acceleration=False (classifier_head.py:74-84,158-162) - forward goldens, plus infer_file at seq_len 63 / 95."""
    cbas, classifier_head = import_reference()
    for tag, kw in (("h64_t63", dict(seq_len=63)), ("h64_t95", dict(seq_len=95)), ("h32", dict(lstm_hidden_size=32)),
                    ("h96_t63", dict(lstm_hidden_size=96, seq_len=63)), ("h64_noacc", dict(use_acceleration=False)),
                    ("h48_noacc_l2_t15", dict(use_acceleration=False, lstm_hidden_size=48, lstm_layers=2, seq_len=15))):
        hcfg = C.HeadConfig(in_features=768, out_features=9, **kw)
        hw = W.synth_head_weights(hcfg, HEAD_SEED)
        m = ref_head(classifier_head, hcfg, hw)
        T = hcfg.seq_len
        seq = synth.cls_walk(21, 48 + T - 1, 768).astype(np.float32)
        x = np.stack([seq[i:i + T] for i in range(48)])
        logits, latent = m(torch.from_numpy(x))
        np.savez_compressed(os.path.join(out, f"head_{tag}.npz"), logits=logits.detach().numpy(), latent=latent.detach().numpy(),
                            x_sha=sha(x), walk_seed=21)
        print("head variant", tag, logits.shape, latent.shape)
    os.replace = _real_replace
    res = {}
    with tempfile.TemporaryDirectory() as td:
        for n, T, temp in ((40, 63, 1.0), (300, 63, 0.9), (260, 95, 1.0)):        # n < seq_len and n > seq_len
            hcfg = C.HeadConfig(seq_len=T)
            m = ref_head(classifier_head, hcfg, W.synth_head_weights(hcfg, HEAD_SEED))
            p = os.path.join(td, f"clip{n}_{T}_cls.h5")
            cls = synth.cls_walk(500 + n + T, n, 768)
            with _FakeH5File(p, "w") as f:
                d = f.create_dataset("cls", shape=(n, 768), dtype="f2")
                d[:] = cls
            o = cbas.infer_file(p, m, "gold", BEHAVIORS, T, device=torch.device("cpu"), temperature=temp)
            assert o is not None
            import pandas as pd
            res[f"probs_{n}_{T}"] = pd.read_csv(o).to_numpy(dtype=np.float64).astype(np.float32)
            res[f"temp_{n}_{T}"] = np.float64(temp)
            res[f"cls_sha_{n}_{T}"] = sha(cls)
            print("infer variant", n, T)
    np.savez_compressed(os.path.join(out, "infer_file_seq.npz"), **res)


class _MaskDropout(torch.nn.Module):
    """Stand-in for an nn.Dropout of the reference module: same arithmetic (x * keep / (1-p)) with the
    keep-mask taken from the shared counter-based hash instead of torch's global RNG."""

    def __init__(self, p, stream, bank):
        super().__init__()
        self.p, self.stream, self.bank = p, stream, bank

    def forward(self, x):
        if not self.training:
            return x
        keep = self.bank["masks"][self.stream]
        return x * keep.to(x.dtype) / (1.0 - self.p)


TRAIN_SAMPLE_STRIDE = 13


def _pack(res, key, a):
    """Keep the fixtures small: tensors over 4096 elements are stored as every 13th element (flat
    order) plus their L2 norm and sum; small tensors in full."""
    a = np.asarray(a, np.float32)
    if a.size > 4096:
        res[key + "#sample"] = a.reshape(-1)[::TRAIN_SAMPLE_STRIDE].copy()
        res[key + "#norm"] = np.float64(np.linalg.norm(a.astype(np.float64)))
        res[key + "#sum"] = np.float64(a.astype(np.float64).sum())
    else:
        res[key] = a.copy()


def train_problem(hcfg: C.HeadConfig, B: int, seed: int):
    """Synthetic labelled windows: a class-dependent offset on top of a CLS-like walk."""
    return synth.train_windows(seed, B, hcfg.in_features, hcfg.out_features, hcfg.seq_len)


def g_head_train(out):
    """Reference ClassifierLSTMDeltas in train() mode + the loss / optimiser lines of
    train_lstm_model (cbas.py:1305-1311, :1331-1348) on fixed batches, dropout masks injected."""
    _, classifier_head = import_reference()
    from oracle import head_train_oracle as HT
    torch.set_grad_enabled(True)                 # the rest of this script runs under no-grad
    for tag, h, nl, B, wd, ls, use_cw, acc in (("h64", 64, 1, 48, 0.0, 0.0, False, True), ("h64_l2", 64, 2, 32, 1e-2, 0.1, True, True),
                                               ("h128", 128, 1, 32, 0.0, 0.05, True, True),
                                               # what the UI's free "LSTM hidden size" field and a 2-stream checkpoint lead to
                                               ("h32", 32, 1, 24, 0.0, 0.0, False, True), ("h96_noacc", 96, 1, 24, 1e-3, 0.05, True, False),
                                               ("h48_noacc_l2", 48, 2, 20, 0.0, 0.0, False, False)):
        if getattr(g_head_train, "only", None) and tag not in g_head_train.only:
            continue
        hcfg = C.HeadConfig(in_features=768, out_features=9, lstm_hidden_size=h, lstm_layers=nl, use_acceleration=acc)
        hw = W.synth_head_weights(hcfg, HEAD_SEED)
        m = classifier_head.ClassifierLSTMDeltas(in_features=768, out_features=9, seq_len=31, lstm_hidden_size=h,
                                                 lstm_layers=nl, use_acceleration=acc)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v).copy()) for k, v in hw.items()}, strict=True)
        bank = {"masks": None}
        m.cls_bottleneck[2] = _MaskDropout(0.1, "cls", bank)
        m.delta_bottleneck[2] = _MaskDropout(0.1, "delta", bank)
        if acc:
            m.acc_bottleneck[2] = _MaskDropout(0.1, "acc", bank)
        m.lin0[2] = _MaskDropout(0.15, "lin0", bank)
        m.train()
        lr, seed, n_steps = 1e-3, 77, 3
        optimizer = torch.optim.Adam([                                               # cbas.py:1305-1308
            {"params": [p for name, p in m.named_parameters() if name != "gate"]},
            {"params": m.gate, "weight_decay": 1e-3}], lr=lr, weight_decay=wd)
        cw = np.linspace(0.5, 1.5, 9).astype(np.float32) if use_cw else None
        criterion = torch.nn.CrossEntropyLoss(weight=None if cw is None else torch.from_numpy(cw), label_smoothing=ls)
        x, y = train_problem(hcfg, B, 5)
        res = {"lr": lr, "seed": seed, "weight_decay": wd, "label_smoothing": ls, "B": B}
        if cw is not None:
            res["class_weights"] = cw
        for s in range(n_steps):
            bank["masks"] = HT.make_masks(seed, s, B, 31, 128, 256)
            optimizer.zero_grad()
            final_logits, rawm = m(torch.from_numpy(x))
            inv_loss = criterion(final_logits, torch.from_numpy(y))
            rawm_centered = rawm - rawm.mean(dim=0)                                  # cbas.py:1338-1343
            covm = (rawm_centered.T @ rawm_centered) / (rawm_centered.shape[0] - 1)
            n_ = covm.shape[0]
            covm_loss = torch.sum(torch.pow(covm.flatten()[:-1].view(n_ - 1, n_ + 1)[:, 1:].flatten(), 2))
            loss = inv_loss + covm_loss
            loss.backward()
            if s == 0:
                res["loss0"], res["ce0"], res["cov0"] = float(loss), float(inv_loss), float(covm_loss)
                res["logits0"], res["latent0"] = final_logits.detach().numpy(), rawm.detach().numpy()
                for name, p in m.named_parameters():
                    _pack(res, "grad0/" + name, p.grad.detach().numpy())
            res[f"loss{s}"] = float(loss)
            optimizer.step()
        for name, p in m.named_parameters():
            _pack(res, "final/" + name, p.detach().numpy())
        np.savez_compressed(os.path.join(out, f"head_train_{tag}.npz"), **res)
        print("head_train", tag, [res[f"loss{s}"] for s in range(n_steps)])
    torch.set_grad_enabled(False)


def g_infer(out):
    """cbas.infer_file end to end on synthetic _cls.h5 contents: edge padding, halo chunking,
    temperature clamp, and the CSV text pandas writes."""
    cbas, classifier_head = import_reference()
    hcfg = C.HeadConfig()
    hw = W.synth_head_weights(hcfg, HEAD_SEED)
    m = ref_head(classifier_head, hcfg, hw)
    os.replace = _real_replace
    res = {}
    with tempfile.TemporaryDirectory() as td:
        for n, temp in ((1, 1.0), (10, 1.0), (31, 1.0), (64, 0.7), (700, 1.0), (20017, 1.3), (40, 1e-6)):
            p = os.path.join(td, f"clip{n}_cls.h5")
            cls = synth.cls_walk(100 + n, n, 768)
            with _FakeH5File(p, "w") as f:
                d = f.create_dataset("cls", shape=(n, 768), dtype="f2")
                d[:] = cls
            o = cbas.infer_file(p, m, "gold", BEHAVIORS, 31, device=torch.device("cpu"), temperature=temp)
            assert o is not None
            import pandas as pd
            probs = pd.read_csv(o).to_numpy(dtype=np.float64)
            with open(o, "r") as fh:
                text = fh.read()
            res[f"probs_{n}"] = probs.astype(np.float32)
            res[f"temp_{n}"] = np.float64(temp)
            res[f"cls_sha_{n}"] = sha(cls)
            if n <= 64:
                res[f"csv_{n}"] = np.frombuffer(text.encode(), dtype=np.uint8)
            print("infer", n, probs.shape)
    np.savez_compressed(os.path.join(out, "infer_file.npz"), **res)


def g_e2e(out):
    """BASELINE config 1: ViT-S/16, 64 frames 224^2, batch 8 -> CLS -> f16 -> infer_file, C=9."""
    cbas, classifier_head = import_reference()
    cfg = C.VIT_S16
    w = W.synth_encoder_weights(cfg, ENC_SEED)
    m = hf_model(cfg, w)
    frames = synth.cage_frames(3, 64, 224, 224)
    g = torch.from_numpy(frames[:, :, :, 1] / 255.0).float()
    cls = torch.cat([m(g[i:i + 8].unsqueeze(1).repeat(1, 3, 1, 1)).last_hidden_state[:, 0]
                     for i in range(0, 64, 8)]).numpy()
    hcfg = C.HeadConfig(in_features=384)
    hw = W.synth_head_weights(hcfg, HEAD_SEED)
    hm = ref_head(classifier_head, hcfg, hw)
    os.replace = _real_replace
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "e2e_cls.h5")
        with _FakeH5File(p, "w") as f:
            d = f.create_dataset("cls", shape=(64, 384), dtype="f2")
            d[:] = cls                                   # f32 -> f2 cast on write, cbas.py:438
            cls16 = d[:].copy()
        # infer_file hard-codes nothing about 768 (the head is built with in_features=384 here)
        o = cbas.infer_file(p, hm, "gold", BEHAVIORS, 31, device=torch.device("cpu"), temperature=1.0)
        import pandas as pd
        probs = pd.read_csv(o).to_numpy(dtype=np.float64).astype(np.float32)
    np.savez_compressed(os.path.join(out, "e2e_vits16.npz"), cls=cls.astype(np.float32), cls_f16=cls16,
                        probs=probs, labels=probs.argmax(1), frames_sha=sha(frames), frame_seed=3, n=64)
    print("e2e labels", np.bincount(probs.argmax(1), minlength=9))


def g_e2e_vitb(out):
    """BASELINE config 2's model end to end: ViT-B/16 through the reference's OWN DinoEncoder wrapper (cbas.py:642-678, loaded
    from a local save_pretrained directory), 256 frames 224^2 in the reference's 8-frame calls -> CLS -> f16 -> infer_file,
    C = 9.  The label-parity fixture for the headline configuration (e2e_vits16 is config 1)."""
    cbas, classifier_head = import_reference()
    cfg = C.VIT_B16
    w = W.synth_encoder_weights(cfg, ENC_SEED)
    n = 256
    frames = synth.cage_frames(4, n, 224, 224)
    os.replace = _real_replace
    with tempfile.TemporaryDirectory() as td:
        hf_model(cfg, w).save_pretrained(td)
        enc = cbas.DinoEncoder(td, device="cpu")
        g = torch.from_numpy(frames[:, :, :, 1] / 255.0).float()                      # cbas.py:431
        cls = torch.cat([enc(g[i:i + 8].unsqueeze(1)).squeeze(1) for i in range(0, n, 8)]).numpy()      # cbas.py:435-436
        hcfg = C.HeadConfig(in_features=768)
        hm = ref_head(classifier_head, hcfg, W.synth_head_weights(hcfg, HEAD_SEED))
        p = os.path.join(td, "e2e_cls.h5")
        with _FakeH5File(p, "w") as f:
            d = f.create_dataset("cls", shape=(n, 768), dtype="f2")
            d[:] = cls
            cls16 = d[:].copy()
        o = cbas.infer_file(p, hm, "gold", BEHAVIORS, 31, device=torch.device("cpu"), temperature=1.0)
        import pandas as pd
        probs = pd.read_csv(o).to_numpy(dtype=np.float64).astype(np.float32)
    np.savez_compressed(os.path.join(out, "e2e_vitb16.npz"), cls=cls.astype(np.float32), cls_f16=cls16, probs=probs,
                        labels=probs.argmax(1), frames_sha=sha(frames), frame_seed=4, n=n)
    top2 = np.sort(probs, axis=1)[:, -2:]
    print("e2e vitb labels", np.bincount(probs.argmax(1), minlength=9), "smallest top-2 margins", np.sort(top2[:, 1] - top2[:, 0])[:6])


def g_e2e_vitb_long(out):
    """The headline model over a LONG clip (r4): 2 048 frames = 93 scene changes of synth.cage_frames, ViT-B/16 through the
    reference's own DinoEncoder wrapper in 8-frame calls -> f16 -> the reference's infer_file.  Stored: the f16 rows the
    reference wrote (all frames), its f32 CLS for every 8th frame (the fp32-mode bar is checked on those), probabilities,
    labels.  This is the fixture the fp16 default's flip RATE and the fp32 mode's zero-flip gate are measured on."""
    cbas, classifier_head = import_reference()
    cfg = C.VIT_B16
    w = W.synth_encoder_weights(cfg, ENC_SEED)
    n = 2048
    frames = synth.cage_frames(6, n, 224, 224)
    os.replace = _real_replace
    with tempfile.TemporaryDirectory() as td:
        hf_model(cfg, w).save_pretrained(td)
        enc = cbas.DinoEncoder(td, device="cpu")
        g = torch.from_numpy(frames[:, :, :, 1] / 255.0).float()                      # cbas.py:431
        cls = torch.cat([enc(g[i:i + 8].unsqueeze(1)).squeeze(1) for i in range(0, n, 8)]).numpy()      # cbas.py:435-436
        hcfg = C.HeadConfig(in_features=768)
        hm = ref_head(classifier_head, hcfg, W.synth_head_weights(hcfg, HEAD_SEED))
        p = os.path.join(td, "e2e_cls.h5")
        with _FakeH5File(p, "w") as f:
            d = f.create_dataset("cls", shape=(n, 768), dtype="f2")
            d[:] = cls
            cls16 = d[:].copy()
        o = cbas.infer_file(p, hm, "gold", BEHAVIORS, 31, device=torch.device("cpu"), temperature=1.0)
        import pandas as pd
        probs = pd.read_csv(o).to_numpy(dtype=np.float64).astype(np.float32)
    labels = probs.argmax(1)
    np.savez_compressed(os.path.join(out, "e2e_vitb16_long.npz"), cls_every8=cls[::8].astype(np.float32), cls_f16=cls16,
                        probs=probs, labels=labels, frames_sha=sha(frames), frame_seed=6, n=n)
    top2 = np.sort(probs, axis=1)[:, -2:]
    print("e2e vitb long labels", np.bincount(labels, minlength=9), "transitions", int((labels[1:] != labels[:-1]).sum()),
          "smallest top-2 margins", np.sort(top2[:, 1] - top2[:, 0])[:8])


def g_encode_file(out):
    """cbas.encode_file on a fake 'video' (decord/h5py fakes): pins chunking (CHUNK_SIZE=512 with a
    ragged tail), the f2 cast and the returned path.  Uses the tiny-D wrapper-compatible config
    (the reference hard-codes 768, so D must be 768: use ViT-B width with 1 layer to stay small)."""
    cbas, _ = import_reference()
    cfg = C.ViTConfig(hidden_size=768, intermediate_size=1536, num_hidden_layers=1, num_attention_heads=12,
                      image_size=32)
    w = W.synth_encoder_weights(cfg, ENC_SEED)
    os.replace = _real_replace
    with tempfile.TemporaryDirectory() as td:
        m = hf_model(cfg, w)
        m.save_pretrained(td)
        enc = cbas.DinoEncoder(td, device="cpu")
        frames = synth.cage_frames(5, 600, 32, 32)
        vp = os.path.join(td, "vid.mp4")
        _FakeVideoReader.CLIPS[vp] = frames
        ticks = []
        o = cbas.encode_file(enc, vp, progress_callback=ticks.append)
        assert o == os.path.join(td, "vid_cls.h5"), o
        d = _FAKE_FS[o]["dsets"]["cls"]
        np.savez_compressed(os.path.join(out, "encode_file_b1layer.npz"), cls_f16=d[:].copy(),
                            ticks=np.array(ticks), frames_sha=sha(frames), frame_seed=5, n=600,
                            chunks=np.array(d.chunks), maxshape1=d.maxshape[1])
        print("encode_file", d.shape, ticks)


def hf_dinov2(cfg: C.ViTConfig, weights):
    from transformers import Dinov2WithRegistersConfig, Dinov2WithRegistersModel
    hcfg = Dinov2WithRegistersConfig(
        hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
        num_attention_heads=cfg.num_attention_heads, mlp_ratio=cfg.intermediate_size // cfg.hidden_size,
        image_size=cfg.image_size, patch_size=cfg.patch_size, num_register_tokens=cfg.num_register_tokens,
        layer_norm_eps=cfg.layer_norm_eps, qkv_bias=True)
    hcfg._attn_implementation = "eager"
    m = Dinov2WithRegistersModel(hcfg).eval()
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in weights.items()}, strict=True)
    return m


def g_dinov2(out):
    """CBAS's default encoder family (facebook/dinov2-with-registers-base, backend/cbas.py:1030-1033):
    HF Dinov2WithRegistersModel with synthetic weights.  Tiny config at the native grid (70x70, no
    interpolation), down-sampled (56x56: 5x5 -> 4x4) and up-sampled (84x84: 5x5 -> 6x6) position
    embeddings; ViT-B/14 at 224 (37x37 -> 16x16) and at CBAS's 256x256 (-> 18x18, 4 px dropped),
    the latter through the reference's own DinoEncoder wrapper."""
    cfg = C.DINOV2_REG_TINY
    w = W.synth_encoder_weights(cfg, ENC_SEED)
    m = hf_dinov2(cfg, w)
    res = {}
    for hw in (70, 56, 84):
        frames = synth.cage_frames(20 + hw, 3, hw, hw)
        g = torch.from_numpy(frames[:, :, :, 1] / 255.0).float()
        o = m(g.unsqueeze(1).repeat(1, 3, 1, 1), output_hidden_states=True)
        res[f"emb_{hw}"] = o.hidden_states[0].numpy()
        res[f"last_{hw}"] = o.last_hidden_state.numpy()
        res[f"sha_{hw}"] = sha(frames)
    np.savez_compressed(os.path.join(out, "dinov2reg_tiny.npz"), **res)
    cfg = C.DINOV2_REG_B14
    w = W.synth_encoder_weights(cfg, ENC_SEED)
    m = hf_dinov2(cfg, w)
    frames = synth.cage_frames(31, 4, 224, 224)
    g = torch.from_numpy(frames[:, :, :, 1] / 255.0).float()
    cls224 = torch.cat([m(g[i:i + 2].unsqueeze(1).repeat(1, 3, 1, 1)).last_hidden_state[:, 0] for i in (0, 2)]).numpy()
    cbas, _ = import_reference()
    frames256 = synth.cage_frames(32, 2, 256, 256)
    g = torch.from_numpy(frames256[:, :, :, 1] / 255.0).float()
    with tempfile.TemporaryDirectory() as td:
        m.save_pretrained(td)
        enc = cbas.DinoEncoder(td, device="cpu")                     # reference wrapper, AutoModel -> Dinov2WithRegistersModel
        cls256 = enc(g.unsqueeze(1)).squeeze(1).numpy()
    np.savez_compressed(os.path.join(out, "dinov2reg_b14.npz"), cls224=cls224, sha224=sha(frames), cls256=cls256,
                        sha256=sha(frames256))
    print("dinov2", cls224.shape, cls256.shape)


def g_e2e_dinov2(out):
    """CBAS's DEFAULT encoder family end to end (r4): DINOv2-with-registers ViT-B/14 (backend/cbas.py:1030-1033) through the
    reference's own DinoEncoder wrapper at CBAS's standard 256 x 256 video size (18 x 18 patches, T = 329), 512 frames in
    8-frame calls -> f16 -> the reference's infer_file, C = 9.  Stored like the long DINOv3 fixture: f16 rows for every frame,
    f32 CLS for every 8th, probabilities, labels."""
    cbas, classifier_head = import_reference()
    cfg = C.DINOV2_REG_B14
    w = W.synth_encoder_weights(cfg, ENC_SEED)
    n = 512
    frames = synth.cage_frames(8, n, 256, 256)
    os.replace = _real_replace
    with tempfile.TemporaryDirectory() as td:
        hf_dinov2(cfg, w).save_pretrained(td)
        enc = cbas.DinoEncoder(td, device="cpu")
        g = torch.from_numpy(frames[:, :, :, 1] / 255.0).float()
        cls = torch.cat([enc(g[i:i + 8].unsqueeze(1)).squeeze(1) for i in range(0, n, 8)]).numpy()
        hcfg = C.HeadConfig(in_features=768)
        hm = ref_head(classifier_head, hcfg, W.synth_head_weights(hcfg, HEAD_SEED))
        p = os.path.join(td, "e2e_cls.h5")
        with _FakeH5File(p, "w") as f:
            d = f.create_dataset("cls", shape=(n, 768), dtype="f2")
            d[:] = cls
            cls16 = d[:].copy()
        o = cbas.infer_file(p, hm, "gold", BEHAVIORS, 31, device=torch.device("cpu"), temperature=1.0)
        import pandas as pd
        probs = pd.read_csv(o).to_numpy(dtype=np.float64).astype(np.float32)
    labels = probs.argmax(1)
    np.savez_compressed(os.path.join(out, "e2e_dinov2reg_b14.npz"), cls_every8=cls[::8].astype(np.float32), cls_f16=cls16,
                        probs=probs, labels=labels, frames_sha=sha(frames), frame_seed=8, n=n, hw=256)
    top2 = np.sort(probs, axis=1)[:, -2:]
    print("e2e dinov2 labels", np.bincount(labels, minlength=9), "transitions", int((labels[1:] != labels[:-1]).sum()),
          "smallest top-2 margins", np.sort(top2[:, 1] - top2[:, 0])[:8])


ALL = {"dinov2": g_dinov2, "tiny": g_tiny, "vits": g_vits, "vitb": g_vitb, "vitb_noise": g_vitb_noise, "vitb256": g_vitb256,
       "vitl": g_vitl, "vitl518": g_vitl518, "head": g_head, "head_variants": g_head_variants, "head_train": g_head_train, "infer": g_infer, "e2e": g_e2e, "e2e_vitb": g_e2e_vitb, "e2e_vitb_long": g_e2e_vitb_long, "e2e_dinov2": g_e2e_dinov2,
       "encode_file": g_encode_file}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--out", default=HERE)
    ap.add_argument("--train-tags", default=None, help="head_train: only these fixtures (e.g. h32,h96_noacc)")
    a = ap.parse_args()
    g_head_train.only = a.train_tags.split(",") if a.train_tags else None
    for k, fn in ALL.items():
        if a.only and k not in a.only.split(","):
            continue
        fn(a.out)

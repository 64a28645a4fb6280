"""Label study for the MX-fp8 throughput mode (BASELINE.json configs[4]) through a head TRAINED on the device.

With seeded random head weights a random BiLSTM turns a 6 % CLS perturbation into |dp| ~ 0.5, so label agreement between
the fp8 and fp16 encoders says nothing ("what's weak" 1).  A trained head has margins.  This script

  1. renders labelled synthetic clips: C behaviours = C scene textures + a blob whose motion depends on the behaviour,
     in segments of random length with cross-fades, sensor noise on every frame;
  2. encodes the training clips with the fp16 encoder (the contract path) and trains ClassifierLSTMDeltas on the device
     with cbas_head_train_* (backend/cbas.py:1326-1348 semantics) on 31-frame windows of those rows;
  3. encodes HELD-OUT clips with the fp16 encoder and with every fp8 plan, classifies them with the trained head
     (infer_file semantics) and reports accuracy against the true labels, label agreement with the fp16 path, |dp| and
     the fp16 top-2 margin at the flips.

    python scripts/fp8_label_study.py [--model vitb16] [--hw 224] [--epochs 30] [--json out.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbas_amd import config as C, weights as W  # noqa: E402
from cbas_amd.encoder import DinoEncoder  # noqa: E402
from cbas_amd.head import ClassifierLSTMDeltas  # noqa: E402
from cbas_amd.train import HeadTrainer, initial_head_weights  # noqa: E402


def render_clip(seed: int, n: int, hw: int, n_classes: int, device="cuda"):
    """(frames (n, hw, hw) uint8 green planes on the device, labels (n,) int64).  Scene c: a smooth texture with
    class-specific spatial frequencies; the blob circles with a class-specific radius / speed; segments of 40-120
    frames in random order, 6-frame cross-fades, +-18 grey levels of per-pixel noise."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    rng = np.random.default_rng(seed)
    yy, xx = torch.meshgrid(torch.arange(hw, device=device, dtype=torch.float32),
                            torch.arange(hw, device=device, dtype=torch.float32), indexing="ij")
    scenes = []
    for c in range(n_classes):
        fx, fy, fd = 9.0 + 5.0 * c, 31.0 - 3.0 * c, 4.0 + 1.5 * c
        scenes.append(110.0 + 10.0 * c + (30.0 + 3.0 * c) * torch.sin(xx / fx) + 25.0 * torch.cos(yy / fy) + 12.0 * torch.sin((xx + yy) / fd))
    scenes = torch.stack(scenes)
    labels = np.empty(n, np.int64)
    seg_of = np.empty(n, np.int64)
    fade = np.zeros(n, np.float32)
    nxt_cls = np.empty(n, np.int64)
    t, cur = 0, int(rng.integers(n_classes))
    while t < n:
        ln = int(rng.integers(40, 121))
        nx = int((cur + 1 + rng.integers(n_classes - 1)) % n_classes)
        for k in range(ln):
            if t + k >= n:
                break
            f = max(0.0, (k - (ln - 6)) / 6.0)
            labels[t + k] = cur if f < 0.5 else nx
            seg_of[t + k], nxt_cls[t + k], fade[t + k] = cur, nx, f
        t += ln
        cur = nx
    frames = torch.empty((n, hw, hw), dtype=torch.uint8, device=device)
    for i0 in range(0, n, 256):
        i1 = min(n, i0 + 256)
        idx = torch.arange(i0, i1, device=device, dtype=torch.float32)
        a = torch.from_numpy(seg_of[i0:i1]).to(device)
        b = torch.from_numpy(nxt_cls[i0:i1]).to(device)
        fd = torch.from_numpy(fade[i0:i1]).to(device)[:, None, None]
        bg = scenes[a] * (1 - fd) + scenes[b] * fd
        cls_f = torch.from_numpy(labels[i0:i1]).to(device).float()
        rad = hw * (0.12 + 0.04 * cls_f)
        om = 0.05 + 0.03 * cls_f
        cx = hw * 0.5 + rad * torch.cos(om * idx)
        cy = hw * 0.5 + rad * torch.sin(om * idx * (1.0 + 0.1 * cls_f))
        sig = 0.09 * hw
        blob = 90.0 * torch.exp(-((xx[None] - cx[:, None, None]) ** 2 + (yy[None] - cy[:, None, None]) ** 2) / (2 * sig * sig))
        noise = (torch.rand((i1 - i0, hw, hw), device=device, generator=g) - 0.5) * 36.0
        frames[i0:i1] = (bg + blob + noise).round().clamp(0, 255).to(torch.uint8)
    return frames, labels


def windows_of(rows16: torch.Tensor, seq_len: int) -> torch.Tensor:
    """(N, D) fp16 rows -> (N, seq_len, D) fp32 replicate-padded windows (infer_file's / the training set's windows)."""
    n = rows16.shape[0]
    half = seq_len // 2
    idx = (torch.arange(n, device=rows16.device)[:, None] + torch.arange(-half, seq_len - half, device=rows16.device)[None, :]).clamp(0, n - 1)
    return rows16.float()[idx]


def study(model="vitb16", hw=224, n_classes=6, epochs=30, train_clips=6, train_len=1200, test_clips=3, test_len=1024,
          plans=(2,), lr=1e-3, seed=0, verbose=True):
    cfg = C.NAMED_VIT[model]
    enc_w = W.synth_encoder_weights(cfg, 1234)
    seq_len = 31
    hcfg = C.HeadConfig(in_features=cfg.hidden_size, out_features=n_classes, seq_len=seq_len)
    log = print if verbose else (lambda *a, **k: None)

    def encode(prec, clips):
        enc = DinoEncoder.from_weights(cfg, enc_w, "cuda", max_batch=64, max_frame=(hw, hw), precision=prec)
        out = [enc.encode_u8(fr, want_f32=False)[0].clone() for fr, _ in clips]
        torch.cuda.synchronize()
        enc.close()
        return out

    t0 = time.perf_counter()
    train = [render_clip(1000 + seed * 100 + i, train_len, hw, n_classes) for i in range(train_clips)]
    test = [render_clip(5000 + seed * 100 + i, test_len, hw, n_classes) for i in range(test_clips)]
    rows_train = encode(0, train)
    rows_test = {0: encode(0, test)}
    rows_train_p = {}
    for p in plans:
        rows_test[p] = encode(p, test)
        rows_train_p[p] = encode(p, train)
    log(f"rendered + encoded {train_clips * train_len + (1 + len(plans)) * test_clips * test_len} frames in {time.perf_counter() - t0:.1f} s")

    # ---- train a head on the rows of one encoder (the same recipe for every encoder) ---------------------------------
    Y = torch.cat([torch.from_numpy(l) for _, l in train]).cuda()

    def train_head(rows, tag):
        X = torch.cat([windows_of(r, seq_len) for r in rows])
        trainer = HeadTrainer(hcfg, initial_head_weights(hcfg, seed), "cuda", lr=lr, max_batch=512, seed=seed)
        gen = torch.Generator(device="cuda")
        gen.manual_seed(seed)
        t0 = time.perf_counter()
        for ep in range(epochs):
            perm = torch.randperm(X.shape[0], device="cuda", generator=gen)
            loss = None
            for b in range(0, X.shape[0], 512):
                sel = perm[b:b + 512]
                loss = trainer.step(X[sel].contiguous(), Y[sel], want_loss=(b + 512 >= X.shape[0]))
            if ep % 10 == 9 or ep == epochs - 1:
                log(f"[{tag}] epoch {ep + 1}: loss {loss[0]:.4f} (ce {loss[1]:.4f}, cov {loss[2]:.4f})")
        h = ClassifierLSTMDeltas(cfg.hidden_size, n_classes, seq_len=seq_len)
        h.load_state_dict(trainer.weights())
        h.to("cuda")
        trainer.close()
        log(f"[{tag}] trained {epochs} epochs on {X.shape[0]} windows in {time.perf_counter() - t0:.1f} s")
        return h

    head = train_head(rows_train, "fp16 rows")

    # ---- held-out clips: fp16 vs each fp8 plan ---------------------------------------------------------------------
    truth = np.concatenate([l for _, l in test])
    probs = {p: np.concatenate([head.infer_clip(r, 1.0).cpu().numpy() for r in rows]) for p, rows in rows_test.items()}
    p16 = probs[0]
    s = np.sort(p16, axis=1)
    margin = s[:, -1] - s[:, -2]
    out = {"model": model, "hw": hw, "classes": n_classes, "train_windows": int(train_clips * train_len),
           "held_out_frames": int(len(truth)), "fp16_accuracy": float((p16.argmax(1) == truth).mean()),
           "fp16_margin_median": float(np.median(margin)), "plans": {}}
    log(f"held-out: {len(truth)} frames; fp16 accuracy {out['fp16_accuracy']:.4f}; fp16 top-2 margin median {np.median(margin):.3f}, "
        f"{int((margin < 0.2).sum())} frames under 0.2")
    for p in plans:
        pp = probs[p]
        cls_rel = max(float((torch.linalg.norm(a.float() - b.float(), dim=1) / torch.linalg.norm(b.float(), dim=1)).max())
                      for a, b in zip(rows_test[p], rows_test[0]))
        adp = np.abs(pp - p16).max(1)
        flips = pp.argmax(1) != p16.argmax(1)
        band = margin <= 2.0 * adp.max()
        rec = {"cls_rel_err_max": cls_rel, "accuracy": float((pp.argmax(1) == truth).mean()),
               "agreement": float(1.0 - flips.mean()), "flips": int(flips.sum()),
               "dp_median": float(np.median(adp)), "dp_p99": float(np.quantile(adp, 0.99)), "dp_max": float(adp.max()),
               "flip_margin_max": float(margin[flips].max()) if flips.any() else 0.0,
               "flips_outside_near_tie_band": int((flips & ~band).sum()), "band_frames": int(band.sum()),
               "flips_at_margin_over_0.2": int((flips & (margin > 0.2)).sum())}
        # the mode's own contract: its rows are a DIFFERENT encoder's rows (files are stamped '#mx-fp8', a bundle trained on
        # fp16 rows is refused) - a head trained on this mode's rows, evaluated on this mode's held-out rows
        own = train_head(rows_train_p[p], f"precision {p} rows")
        po = np.concatenate([own.infer_clip(r, 1.0).cpu().numpy() for r in rows_test[p]])
        again = np.concatenate([own.infer_clip(r, 1.0).cpu().numpy() for r in encode(p, test)])
        own.close()
        rec["own_head_accuracy"] = float((po.argmax(1) == truth).mean())
        rec["own_head_bit_reproducible"] = bool(np.array_equal(po, again))
        out["plans"][str(p)] = rec
        log(f"precision {p}: a head trained on THIS mode's rows scores {rec['own_head_accuracy']:.4f} on its held-out rows "
            f"(fp16 pipeline: {out['fp16_accuracy']:.4f}); encode + classify twice bit-identical: {rec['own_head_bit_reproducible']}")
        log(f"precision {p}: CLS rel err max {cls_rel:.3e}; accuracy {rec['accuracy']:.4f}; agreement with fp16 {rec['agreement']:.4f} "
            f"({rec['flips']} flips, largest fp16 margin at a flip {rec['flip_margin_max']:.3f}); |dp| median {rec['dp_median']:.2e} "
            f"p99 {rec['dp_p99']:.2e} max {rec['dp_max']:.2e}; flips outside the near-tie band: {rec['flips_outside_near_tie_band']}")
    head.close()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="vitb16")
    ap.add_argument("--hw", type=int, default=224)
    ap.add_argument("--epochs", type=int, default=30)
    ap.add_argument("--classes", type=int, default=6)
    ap.add_argument("--plans", default="2")
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    res = study(a.model, a.hw, a.classes, a.epochs, plans=tuple(int(x) for x in a.plans.split(",")), lr=a.lr)
    if a.json:
        json.dump(res, open(a.json, "w"), indent=1)

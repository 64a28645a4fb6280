"""Is head TRAINING bit-stable beside other work on the device?  In CBAS TrainingThread, EncodeThread and ClassificationThread
are started together and share one device (backend/workthreads.py:1256-1267), so the training kernels run beside the encoder.
A 40-step run (dropout on, Adam) on its own stream is repeated while the main thread keeps the device busy with (a) precision-4
encoder passes (the mode whose attention kernel exposed round 4's interference), (b) default-precision passes, (c) the
register-only v_mfma_f32_32x32x16_f16 loop (cbas_debug_mfma_neighbor); every loss and every trained weight must equal the
idle-device run bit for bit.

    python scripts/train_beside_encoder.py [seconds_per_neighbour [out.json]]
"""
import os as _os
_os.environ.setdefault("CBAS_BUILD_DEBUG", "1")      # the MFMA neighbour is a debug-build entry point
import json, os, sys, threading, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbas_amd import config as C, weights as W, synth, _lib  # noqa: E402
from cbas_amd.encoder import DinoEncoder  # noqa: E402
from cbas_amd.train import HeadTrainer  # noqa: E402

STEPS = 40


def train_run(hcfg, hw, x, y, stream=None):
    """40 Adam steps from the same initial weights and seed -> (losses, weight blob bytes)."""
    tr = HeadTrainer(hcfg, hw, "cuda", lr=1e-3, weight_decay=1e-4, label_smoothing=0.05, max_batch=int(x.shape[0]), seed=7, dropout=True)
    try:
        losses = [tr.step(x, y) for _ in range(STEPS)]
        wts = tr.weights()
    finally:
        tr.close()
    blob = np.concatenate([np.asarray(wts[k], np.float32).reshape(-1) for k in sorted(wts)])
    return np.asarray(losses, np.float32), blob


def run(seconds: float = 6.0, neighbours=("p4", "p0", "mfma")):
    hcfg = C.HeadConfig(in_features=768, out_features=9)
    hw = W.synth_head_weights(hcfg, 4321)
    xs, ys = synth.train_windows(5, 64, 768, 9, 31)
    x, y = torch.from_numpy(xs).cuda(), torch.from_numpy(ys).cuda()
    ref_losses, ref_blob = train_run(hcfg, hw, x, y)
    again_losses, again_blob = train_run(hcfg, hw, x, y)
    assert np.array_equal(ref_losses.view(np.uint32), again_losses.view(np.uint32)) and np.array_equal(ref_blob.view(np.uint32), again_blob.view(np.uint32)), \
        "training is not deterministic on an idle device"
    cfg = C.VIT_B16
    fr = torch.from_numpy(synth.cage_frames(2, 64, 224, 224)[:, :, :, 1].copy()).cuda()
    lib = _lib.load()
    s2 = torch.cuda.Stream()
    dev = torch.cuda.current_device()
    out = []
    for nb in neighbours:
        enc = None
        if nb in ("p4", "p0"):
            enc = DinoEncoder.from_weights(cfg, W.synth_encoder_weights(cfg, 1234), "cuda", max_batch=64, max_frame=(224, 224),
                                           precision=4 if nb == "p4" else 0)
        stop, res = threading.Event(), {"runs": 0, "runs_differing": 0, "first_difference": None}

        def loop():
            torch.cuda.set_device(dev)
            with torch.cuda.stream(s2):
                while not stop.is_set():
                    losses, blob = train_run(hcfg, hw, x, y)
                    res["runs"] += 1
                    dl = np.nonzero(losses.view(np.uint32) != ref_losses.view(np.uint32))[0]
                    dw = int((blob.view(np.uint32) != ref_blob.view(np.uint32)).sum())
                    if len(dl) or dw:
                        res["runs_differing"] += 1
                        if res["first_difference"] is None:
                            res["first_difference"] = {"first_step_with_another_loss": int(dl[0]) if len(dl) else None, "weights_differing": dw}
        th = threading.Thread(target=loop)
        th.start()
        t0, k = time.time(), 0
        try:
            while time.time() - t0 < seconds:
                if nb == "mfma":
                    _lib.check(lib.cbas_debug_mfma_neighbor(5000, None), "mfma_neighbor")      # ~2.5 ms launches: the trainer gets its turns
                else:
                    enc.encode_u8(fr, want_f32=False)
                torch.cuda.synchronize()
                k += 1
        finally:
            stop.set()
            th.join()
            if enc is not None:
                enc.close()
        out.append(dict(neighbour=nb, neighbour_launches=k, steps_per_run=STEPS, **res))
    return out


if __name__ == "__main__":
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
    res = {"seconds_per_neighbour": secs, "neighbours": run(secs)}
    print(json.dumps(res, indent=1))
    if len(sys.argv) > 2:
        json.dump(res, open(sys.argv[2], "w"), indent=1)
    sys.exit(1 if any(r["runs_differing"] for r in res["neighbours"]) else 0)

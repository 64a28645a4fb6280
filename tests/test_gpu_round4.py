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

"""BASELINE.json configs[3] ("cfg4"): DINOv3 ViT-L/16, 518x518 frames, batch 32 on one MI355X - the MFMA-bound
configuration.  T = 1029 tokens per frame (streaming attention kernel), M = 32 928 rows per batch.  The oracle
takes minutes per frame at this size, so the batch-32 result is tied to the reference through the 2-frame golden
(made by transformers' DINOv3ViTModel, tests/golden/make_goldens.py) plus size-independent properties:
a frame's CLS is bit-identical whatever batch it rides in and wherever it sits, duplicates give duplicates,
the pruned last layer equals the full one, asynchronous lanes equal the synchronous call."""
import os

import numpy as np
import pytest
import torch

from cbas_amd import config as C, weights as W, synth

pytestmark = pytest.mark.gpu


def test_vitl16_518_batch32(golden_dir):
    from cbas_amd.encoder import DinoEncoder
    g = np.load(os.path.join(golden_dir, "vitl16_518.npz"))
    cfg = C.VIT_L16
    enc = DinoEncoder.from_weights(cfg, W.synth_encoder_weights(cfg, 1234), "cuda", max_batch=32, max_frame=(518, 518))
    try:
        gold = synth.cage_frames(int(g["frame_seed"]), 2, 518, 518)
        rest = synth.noise_frames(77, 14, 518, 518)
        batch = np.concatenate([gold, rest, gold[::-1], rest])            # 32 frames: the golden pair twice + 2 x 14
        fd = torch.from_numpy(batch).cuda()
        b16, b32 = enc.encode_u8(fd)
        s16, s32 = enc.encode_u8(fd[:2])                                  # the same two frames as a batch of 2
        torch.cuda.synchronize()
        assert torch.isfinite(b32).all()
        # reference parity on the golden frames, at batch 32
        ref = g["cls"].astype(np.float64)
        got = b32[:2].cpu().numpy().astype(np.float64)
        rel = np.linalg.norm(got - ref, axis=1) / np.linalg.norm(ref, axis=1)
        print(f"cfg4 ViT-L/16 518^2 batch 32: CLS rel err vs reference golden {rel.max():.3e}")
        assert rel.max() < 1e-3, rel
        # batch / position invariance, bit for bit
        assert torch.equal(b32[:2], s32) and torch.equal(b16[:2], s16)
        assert torch.equal(b32[16], b32[1]) and torch.equal(b32[17], b32[0])          # reversed duplicates
        assert torch.equal(b32[2:16], b32[18:32])
        perm = torch.from_numpy(np.random.default_rng(3).permutation(32)).cuda()
        p16, _ = enc.encode_u8(fd[perm].contiguous(), want_f32=False)
        torch.cuda.synchronize()
        assert torch.equal(p16, b16[perm])
        # full last layer == pruned last layer
        enc.set_prune_last_layer(False)
        f16, f32 = enc.encode_u8(fd)
        enc.set_prune_last_layer(True)
        torch.cuda.synchronize()
        assert torch.equal(f32, b32) and torch.equal(f16, b16)
        # two batches in flight on the two compute lanes == synchronous
        out = torch.zeros((64, cfg.hidden_size), dtype=torch.float16, device="cuda")
        enc.submit_dev(0, fd, out[:32])
        enc.submit_dev(1, fd[perm].contiguous(), out[32:])
        enc.wait_stream(0)
        enc.wait_stream(1)
        torch.cuda.synchronize()
        assert torch.equal(out[:32], b16) and torch.equal(out[32:], b16[perm])
    finally:
        enc.close()

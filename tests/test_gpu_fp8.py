"""precision = 2: the MX-fp8 throughput mode (BASELINE.json configs[4], "DINOv3 ViT-B/16 fp8 weights").

The reference has no fp8 arithmetic; its low-precision site is the fp16 autocast (backend/cbas.py:433-434).  e4m3 keeps
3 mantissa bits, so this mode cannot meet the 1e-3 CLS bar and is held to what SURVEY.md section 7 states for it:
argmax-label parity (outside the near-tie band its own probability error implies).  The tests pin, in order:
  1. the GEMM kernel: product of exactly the quantised bytes / scales it was given (layout, scale routing, every tile);
  2. the quantiser: e4m3 bytes and E8M0 scales equal the CPU restatement (oracle/mx_oracle.py) bit for bit;
  3. the encoder: CLS vs the CPU restatement of the same quantised arithmetic, and vs the fp32 reference golden - the
     tolerance it actually achieves is printed and asserted;
  4. labels: fp8 clip -> head vs the fp16 path (itself pinned to the reference), flips counted and printed.
"""
import ctypes as C_
import os

import numpy as np
import pytest
import torch

from cbas_amd import config as C, weights as W, synth, _lib
from conftest import assert_labels_match

pytestmark = pytest.mark.gpu


def _gemm_f8(A, Wt, tile=0):
    lib = _lib.load()
    M, K = A.shape
    N = Wt.shape[0]
    M_pad = (M + 255) // 256 * 256
    out = np.empty((M, N), np.float32)
    A8, W8 = np.empty((M, K), np.uint8), np.empty((N, K), np.uint8)
    Asc, Wsc = np.empty((K // 128, M_pad), np.uint32), np.empty((K // 128, N), np.uint32)
    A, Wt = np.ascontiguousarray(A, np.float32), np.ascontiguousarray(Wt, np.float32)
    _lib.check(lib.cbas_debug_gemm_f8(M, N, K, tile, A.ctypes.data, Wt.ctypes.data, out.ctypes.data, A8.ctypes.data,
                                      Asc.ctypes.data, W8.ctypes.data, Wsc.ctypes.data), "cbas_debug_gemm_f8")
    return out, A8, Asc, W8, Wsc


def _dequant(b8, sc, rows):
    """bytes [R][K] + scales [K/128][ld] dwords -> float64 [R][K]."""
    from oracle import mx_oracle as MX
    R, K = b8.shape
    sb = sc.view(np.uint8).reshape(sc.shape[0], sc.shape[1], 4)[:, :rows, :]           # [kt][row][block]
    sb = sb.transpose(1, 0, 2).reshape(R, K // 32).astype(np.float64)
    return MX.e4m3_decode(b8).astype(np.float64) * np.repeat(2.0 ** (sb - 127.0), 32, axis=1)


@pytest.mark.parametrize("M,N,K,tile", [(300, 512, 256, 16), (1000, 768, 768, 13), (777, 256, 1024, 14),
                                        (520, 512, 512, 15), (12864, 768, 3072, 0), (201, 2304, 768, 0)])
def test_fp8_gemm_equals_product_of_its_quantised_operands(M, N, K, tile):
    from oracle import mx_oracle as MX
    rng = np.random.default_rng(M + N + K)
    A = (rng.standard_normal((M, K)) * np.exp(rng.standard_normal((M, 1)))).astype(np.float32)     # rows of very different scale
    A[:, ::7] *= 30.0
    A[3] = 0.0                                                                                     # an all-zero row (scale byte 0)
    Wt = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    out, A8, Asc, W8, Wsc = _gemm_f8(A, Wt, tile)
    # (2) the quantiser against the CPU restatement, bit for bit
    if M <= 1000:
        _, qa, sa = MX.mx_quant(A)
        assert np.array_equal(MX.e4m3_decode(A8), qa)
        got_sa = Asc.view(np.uint8).reshape(K // 128, -1, 4)[:, :M].transpose(1, 0, 2).reshape(M, K // 32)
        assert np.array_equal(got_sa.astype(np.int32), sa)
    # (1) the kernel: product of exactly those operands, fp32 accumulation
    ref = _dequant(A8, Asc, M) @ _dequant(W8, Wsc, N).T
    err = np.abs(out - ref).max() / np.abs(ref).max()
    # the scaled MFMA does not accumulate its 128 products as an fp32 fma chain: on wide-dynamic-range operands it is
    # exact only to ~1e-4 of sum|a b| (measured with scripts/probes/probe_mfma_scale.hip), far below the e4m3 quantisation
    assert err < 5e-4, err
    assert np.all(out[3] == 0.0)


@pytest.mark.parametrize("cfgname,gold,hw", [("vitb16", "vitb16_224_noise", 224), ("vitb16", "vitb16_256", 256)])
def test_fp8_encoder_against_its_restatement_and_the_reference(golden_dir, cfgname, gold, hw):
    from cbas_amd.encoder import DinoEncoder
    from oracle import mx_oracle as MX
    g = np.load(os.path.join(golden_dir, gold + ".npz"))
    cfg = C.NAMED_VIT[cfgname]
    w = W.synth_encoder_weights(cfg, 1234)
    mk = synth.noise_frames if str(g["kind"]) == "noise" else synth.cage_frames
    fr = mk(int(g["frame_seed"]), int(g["n"]), hw, hw)[:4]
    enc = DinoEncoder.from_weights(cfg, w, "cuda", max_batch=8, max_frame=(hw, hw), precision=2)
    try:
        fd = torch.from_numpy(fr).cuda()
        c16, c32 = enc.encode_u8(fd)
        enc.set_prune_last_layer(False)
        f16, f32 = enc.encode_u8(fd)
        torch.cuda.synchronize()
        got = c32.cpu().numpy().astype(np.float64)
    finally:
        enc.close()

    def rel(a, b):
        return (np.linalg.norm(a - b, axis=1) / np.linalg.norm(b, axis=1)).max()
    ref32 = g["cls"][:4].astype(np.float64)
    emu = MX.encode_frames_mx(fr, w, cfg).astype(np.float64)
    r_emu, r_ref, r_emu_ref = rel(got, emu), rel(got, ref32), rel(emu, ref32)
    r_full = rel(f32.cpu().numpy().astype(np.float64), ref32)
    print(f"\nfp8 {gold}: CLS rel err  GPU vs CPU restatement {r_emu:.3e} | GPU vs fp32 reference {r_ref:.3e} "
          f"(full last layer in fp8: {r_full:.3e}) | restatement vs reference {r_emu_ref:.3e}")
    assert np.isfinite(got).all()
    # same quantised arithmetic, but e4m3 rounding decisions flip on last-bit differences of the inputs (GELU rounded
    # through fp16, fp16 attention, MFMA accumulation order), so the two fp8 results are nearly as far from each other
    # as each is from the fp32 reference: a sanity bound, not a tight one
    assert r_emu < 8e-2, r_emu
    assert r_ref < 1.0e-1, r_ref               # the tolerance this mode achieves on synthetic (unstructured) weights
    assert r_ref < 1.5 * r_emu_ref + 1e-2      # and no worse than the stated arithmetic implies


def test_fp8_labels_against_the_fp16_path():
    """A 1 024-frame clip with temporal structure through encoder (fp8 vs fp16) -> fp16 rows -> head: probabilities
    within the error the CLS perturbation implies and no label flip outside the near-tie band; counts printed."""
    from cbas_amd.encoder import DinoEncoder
    from cbas_amd.head import ClassifierLSTMDeltas
    cfg = C.VIT_B16
    w = W.synth_encoder_weights(cfg, 1234)
    head = ClassifierLSTMDeltas(768, 9)
    head.load_state_dict(W.synth_head_weights(C.HeadConfig(), 4321))
    head.to("cuda")
    gen = torch.Generator(device="cuda")
    gen.manual_seed(11)
    N = 1024
    base = torch.randint(0, 256, (6, 224, 224), dtype=torch.uint8, device="cuda", generator=gen).float()
    idx = torch.arange(N, device="cuda")
    seg, frac = (idx // 200) % 6, ((idx % 200).float() / 200.0)
    clip = (base[seg] * (1 - frac[:, None, None]) + base[(seg + 1) % 6] * frac[:, None, None]).clamp(0, 255).to(torch.uint8)
    probs = {}
    for prec in (0, 2):
        enc = DinoEncoder.from_weights(cfg, w, "cuda", max_batch=64, max_frame=(224, 224), precision=prec)
        c16, _ = enc.encode_u8(clip, want_f32=False)
        probs[prec] = head.infer_clip(c16, 1.0).cpu().numpy()
        cls = c16.float().cpu().numpy()
        if prec == 0:
            cls0, c16_0 = cls, c16.clone()
        enc.close()
    # calibration: white noise of the SAME per-row norm as the fp8 error, added to the fp16 rows, through the same head
    err = torch.from_numpy(np.linalg.norm(cls - cls0, axis=1)).to("cuda")
    noise = torch.randn(c16_0.shape, device="cuda", generator=gen)
    noise = noise / noise.norm(dim=1, keepdim=True) * err[:, None]
    probs["noise"] = head.infer_clip((c16_0.float() + noise).half(), 1.0).cpu().numpy()
    head.close()
    rel = (np.linalg.norm(cls - cls0, axis=1) / np.linalg.norm(cls0, axis=1)).max()
    adp = np.abs(probs[2] - probs[0]).max(1)
    adp_n = np.abs(probs["noise"] - probs[0]).max(1)
    agree = float((probs[2].argmax(1) == probs[0].argmax(1)).mean())
    agree_n = float((probs["noise"].argmax(1) == probs[0].argmax(1)).mean())
    s0 = np.sort(probs[0], axis=1)
    margin = s0[:, -1] - s0[:, -2]
    flips = probs[2].argmax(1) != probs[0].argmax(1)
    print(f"\nfp8 vs fp16 on a {N}-frame clip (synthetic weights): CLS rel err max {rel:.3e}; |dp| median {np.median(adp):.3e} "
          f"p99 {np.quantile(adp, 0.99):.3e} max {adp.max():.3e}; label agreement {agree:.4f} ({int(flips.sum())} flips); "
          f"fp16 top-2 margin at the flips: median {np.median(margin[flips]) if flips.any() else 0:.3e} max "
          f"{margin[flips].max() if flips.any() else 0:.3e}\n"
          f"white noise of the same per-row norm on the fp16 rows: |dp| median {np.median(adp_n):.3e}; label agreement {agree_n:.4f}")
    # The label bar, stated exactly.  With SYNTHETIC weights the head is hypersensitive: a random BiLSTM over time
    # differences of the CLS rows of a random ViT turns a 6 % row perturbation into |dp| ~ 0.5, so "no flip outside
    # the near-tie band" is weak by construction (the band is almost the whole simplex) and the raw agreement moves by
    # +-0.1 with ANY rounding-level change upstream (it read 0.84 and 0.73 for two GELU formulations 3e-7 apart).
    # What can be asserted is relative: the fp8 error flips about as many labels as white noise of the same size does
    # (measured: 0.73 - 0.84 for fp8 against 0.86 for the noise; the fp8 error is correlated along the row, white
    # noise is not).
    # Real checkpoints cannot be fetched here (gated, no network).
    n_mis, n_near = assert_labels_match(probs[2], probs[0], prob_tol=1.0)
    assert agree > agree_n - 0.2, (agree, agree_n)
    assert np.median(adp) < 2.5 * np.median(adp_n) + 1e-3
    assert rel < 1.2e-1


def test_fp8_rejects_unsupported_shapes():
    from cbas_amd.encoder import DinoEncoder
    cfg = C.NAMED_VIT["vits16"]                      # D = 384 is not a multiple of 256
    with pytest.raises(RuntimeError, match="multiples of 256"):
        DinoEncoder.from_weights(cfg, W.synth_encoder_weights(cfg, 1234), "cuda", max_batch=2, max_frame=(64, 64), precision=2)

"""precision = 2: the MX-fp8 throughput mode (BASELINE.json configs[4], "DINOv3 ViT-B/16 fp8 weights").

The reference has no fp8 arithmetic; its low-precision site is the fp16 autocast (backend/cbas.py:433-434).  e4m3 keeps
3 mantissa bits, so this mode cannot meet the 1e-3 CLS bar and is held to what SURVEY.md section 7 states for it:
argmax-label parity (outside the near-tie band its own probability error implies).  The tests pin, in order:
  1. the GEMM kernel: product of exactly the quantised bytes / scales it was given (layout, scale routing, every tile);
  2. the quantiser: e4m3 bytes and E8M0 scales equal the CPU restatement (oracle/mx_oracle.py) bit for bit;
  3. the encoder: CLS vs the CPU restatement of the same quantised arithmetic, and vs the fp32 reference golden - the
     tolerance it actually achieves is printed and asserted;
  4. labels: through heads TRAINED on the device on labelled synthetic clips - the fp8 pipeline (fp8 rows + a head trained on
     fp8 rows) must classify held-out clips within 0.10 of the fp16 pipeline's accuracy; the agreement of an fp16-trained
     head fed fp8 rows (70-80 %) is printed: the two kinds of rows are not interchangeable, which is why fp8 files are stamped.
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


def test_fp8_label_gate_through_a_head_trained_on_the_device(capsys):
    """The label bar of the MX-fp8 throughput mode (BASELINE.json configs[4]; reference low-precision site:
    backend/cbas.py:433-434), measured through heads TRAINED on the device (cbas_head_train_*) on labelled synthetic clips
    (scripts/fp8_label_study.py): 6 behaviours = scene texture + blob motion, 7 200 training windows, 3 072 held-out frames.

    What is asserted, and why it is this and not "labels identical to the fp16 path":
    * MX-fp8 rows are ~6e-2 from the fp16 rows on unstructured (synthetic) weights: every fp8 GEMM adds ~2-3 % relative
      noise to its output (3 mantissa bits; the block scales only fix the range).  A head trained on fp16 rows therefore
      agrees with itself on only 70-80 % of the held-out frames when fed fp8 rows, with flips at fp16 margins up to 0.9 - this
      is measured here and PRINTED, and it is why fp8 files are stamped as a different encoder (pipeline.file_attrs) and a
      bundle trained on fp16 rows is refused for an fp8 run (encode_files.py): the two kinds of rows are not interchangeable.
    * Used as what it is - a different encoder, with a head trained on ITS rows - the mode must classify held-out clips
      nearly as well as the fp16 pipeline does: accuracy against the TRUE labels within 0.10 of the fp16 pipeline's and
      far above chance (1/6), bit-reproducibly.  A kernel that corrupted, dropped or mis-scaled rows fails this."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import fp8_label_study as S
    with capsys.disabled():
        res = S.study("vitb16", 224, n_classes=6, epochs=30, plans=(2,), verbose=True)
    r = res["plans"]["2"]
    assert res["fp16_accuracy"] > 0.85                                   # the task is learnable from fp16 rows
    assert r["own_head_bit_reproducible"]
    assert r["own_head_accuracy"] > 0.70 and r["own_head_accuracy"] > res["fp16_accuracy"] - 0.10, r
    assert r["accuracy"] > 0.5                                           # even the mismatched pairing is far from chance
    assert 4e-2 < r["cls_rel_err_max"] < 1.0e-1                          # the format's error, neither better nor worse
    assert r["flips_outside_near_tie_band"] == 0                         # (the band is wide here: printed above)


def test_fp8_rejects_unsupported_shapes():
    from cbas_amd.encoder import DinoEncoder
    cfg = C.NAMED_VIT["vits16"]                      # D = 384 is not a multiple of 256
    with pytest.raises(RuntimeError, match="multiples of 256"):
        DinoEncoder.from_weights(cfg, W.synth_encoder_weights(cfg, 1234), "cuda", max_batch=2, max_frame=(64, 64), precision=2)

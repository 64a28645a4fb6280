"""Does anything that runs BESIDE a head inference (another stream, same device) change its result?
The head classifies fixed rows in a loop on its own stream and every result is compared, bit for bit, with the result
of the idle device, while the main thread keeps the device busy with (a) encoder passes cut off after a given stage
(0 patch, 1 LayerNorm, 2 q|k|v, 3 attention, 4 o_proj, 5 LayerNorm, 6 up, 7 down of layer 0; -1 = the whole encoder) or
(b) a register-only v_mfma_f32_32x32x16_f16 loop on every SIMD (cbas_debug_mfma_neighbor; stage "mfma").

    python scripts/head_beside_encoder.py [precision [seconds [out.json]]]      (CBAS_SPLIT_FORMS=<0..3>: precision 4's GEMM forms)

Why this exists (round 4).  The precision-4 soak of the file path reported CSV files that differed, in single frames by
~1e-2, from the same clips run alone, while the encoder's rows were identical.  Narrowed down on the GPU:
  * the head's result changed only when a NEW form of precision 4's attention kernel ran beside it (bit-identical rows, one
    basic block per key block, 126 VGPRs) - although that kernel only reads: the effect stayed with its LDS-DMA, its P.V
    half and its stores compiled out; rocBLAS / elementwise torch kernels beside it stayed bit-stable; 2 GiB of canary
    buffers saw no stray write; it needed co-residence on a CU;
  * the same happened beside a pure v_mfma_f32_32x32x16_f16 loop in ANOTHER PROCESS (not beside 16x16x32 loops);
  * the wrong values were always in `head_expand_kernel`: ONE LayerNorm input row of the third stream, columns 48-63 or
    112-127 = lanes 48-63 (the last of a wave64 VALU instruction's four passes) of that stream's two waves, holding values
    that match none of the kernel's intermediate quantities;
  * what removed it: no divergent (EXEC-masked) code in the kernel's arithmetic - the device library's erff picks one of two
    formulas with a branch; branch-free erf: 15 % -> 0.3 % of runs beside the MFMA loop - and no LDS read-back of values the
    same thread wrote a few iterations earlier (the EMA written in one pass and re-read three at a time in the next; now
    carried in registers): 0 of 10 170 runs beside the attention kernel, 0 of 381 beside the MFMA loop.
Whether the hardware or the generated code is at fault was not established.  tests/test_gpu_round4.py runs this check."""
import os as _os
_os.environ.setdefault("CBAS_BUILD_DEBUG", "1")      # bring-up entry points: the debug build of the library
import json, os, sys, threading, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbas_amd import config as C, weights as W, synth, _lib  # noqa: E402
from cbas_amd.encoder import DinoEncoder  # noqa: E402
from cbas_amd.head import ClassifierLSTMDeltas  # noqa: E402


def run(precision: int, seconds: float, stages=((0, 3), (0, 5), (0, 7), (-1, -1), ("mfma", 0)), forms: int = -1):
    cfg = C.VIT_B16
    enc = DinoEncoder.from_weights(cfg, W.synth_encoder_weights(cfg, 1234), "cuda", max_batch=64, max_frame=(224, 224), precision=precision)
    if precision == 4:
        enc.debug_option("split_kernels", forms)
    head = ClassifierLSTMDeltas(768, 9)
    head.load_state_dict(W.synth_head_weights(C.HeadConfig(), 4321))
    head.to("cuda")
    fr = torch.from_numpy(synth.cage_frames(2, 64, 224, 224)[:, :, :, 1].copy()).cuda()
    rows = torch.from_numpy(np.random.default_rng(5).standard_normal((3000, 768)).astype(np.float16)).cuda()
    ref = head.infer_clip(rows)
    torch.cuda.synchronize()
    s2 = torch.cuda.Stream()
    dev = torch.cuda.current_device()

    def head_loop(stop, res):
        torch.cuda.set_device(dev)
        with torch.cuda.stream(s2):
            while not stop.is_set():
                o = head.infer_clip(rows)
                s2.synchronize()
                res[0] += 1
                res[1] += int(not torch.equal(o, ref))
    if precision == 2:                                    # MX-fp8 has no debug taps: whole passes and the MFMA loop only
        stages = tuple(st for st in stages if st[0] in (-1, "mfma"))
    out = []
    try:
        for layer, stage in stages:
            stop, res = threading.Event(), [0, 0]
            th = threading.Thread(target=head_loop, args=(stop, res))
            th.start()
            t0, k = time.time(), 0
            try:
                while time.time() - t0 < seconds:
                    if layer == "mfma":                   # ~10 ms of dense 32x32x16 MFMAs per launch on the default stream
                        _lib.check(enc._lib.cbas_debug_mfma_neighbor(20000, None), "mfma_neighbor")
                        torch.cuda.synchronize()
                    elif layer == -1:                     # the whole encoder (every precision)
                        enc.encode_u8(fr, want_f32=False)
                        torch.cuda.synchronize()
                    else:
                        _lib.check(enc._lib.cbas_enc_debug_forward_u8(enc._h, fr.data_ptr(), 64, 224, 224, 224 * 224, 224, 1, layer, stage), "debug_forward")
                    k += 1
            finally:
                stop.set()
                th.join()
            out.append({"stop_layer": layer, "stop_stage": stage, "encoder_passes": k, "head_runs": res[0], "head_runs_differing": res[1]})
    finally:
        if precision == 4:
            enc.debug_option("split_kernels", -1)
        head.close()
        enc.close()
    return out


if __name__ == "__main__":
    precision = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 2.5
    forms = int(os.environ.get("CBAS_SPLIT_FORMS", "-1"))
    res = {"precision": precision, "split_forms": forms, "seconds_per_stage": seconds, "stages": run(precision, seconds, forms=forms)}
    print(json.dumps(res, indent=1))
    if len(sys.argv) > 3:
        json.dump(res, open(sys.argv[3], "w"), indent=1)
    sys.exit(1 if any(s["head_runs_differing"] for s in res["stages"]) else 0)

"""Yardstick, not product code: what the vendor GEMM library (hipBLASLt / rocBLAS through torch.matmul) reaches on the
encoder's four GEMM shapes, next to this repo's fused kernel (the planner's choice, i.e. gemm_f16_8ph) on the same shapes.
The library numbers are for a PLAIN fp16 GEMM (fp16 out; no bias / GELU / LayerScale / residual / RoPE epilogue), so they
bound what a fused kernel could hope for; they are not a like-for-like timing.  The MX-fp8 form of the fused kernel rides
along (no library counterpart through torch).

    python scripts/lib_gemm_ref.py [out.json]        -> profiles/r04_lib_gemm_ref.json (ViT-B at M = 12 864, ViT-L at M = 32 928)
"""
import os as _os
_os.environ.setdefault("CBAS_BUILD_DEBUG", "1")      # bring-up entry points: the debug build of the library
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbas_amd import _lib  # noqa: E402

lib = _lib.load()
fn = lib.cbas_debug_gemm_bench
fn.restype = C.c_int
fn.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_float), C.POINTER(C.c_ulonglong)]
ITERS = 200


def fused(m, n, k, code):
    ms, cs = C.c_float(), C.c_ulonglong()
    rc = fn(m, n, k, code, ITERS, C.byref(ms), C.byref(cs))
    return ms.value * 1e3 if rc == 0 else float("nan")


def library(m, n, k, dev):
    a = torch.randn(m, k, device=dev, dtype=torch.float16)
    w = torch.randn(n, k, device=dev, dtype=torch.float16)
    for _ in range(10):
        torch.matmul(a, w.t())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(ITERS):
        torch.matmul(a, w.t())
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / ITERS * 1e3


def main():
    dev = torch.device("cuda")
    out = {"what": "plain fp16 GEMM of the vendor library (torch.matmul -> hipBLASLt / rocBLAS) against this repo's fused GEMM "
                   "kernels on the encoder's shapes; microseconds per launch, 200 back-to-back launches each, random operands",
           "sets": []}
    for tag, M, D, F in (("ViT-B/16 224^2 batch 64", 12864, 768, 3072), ("ViT-L/16 518^2 batch 32", 32928, 1024, 4096)):
        rows = []
        for name, n, k, eoff in (("qkv", 3 * D, D, 200), ("o_proj", D, D, 100), ("up", F, D, 0), ("down", D, F, 100)):
            fl = 2.0 * M * n * k
            lib_us = library(M, n, k, dev)
            f16_us = fused(M, n, k, 0 + eoff)
            f8_us = fused(M, n, k, 500 + eoff) if eoff != 200 else fused(M, n, k, 500 + 200)
            r = {"gemm": name, "M": M, "N": n, "K": k,
                 "library_plain_us": round(lib_us, 2), "library_plain_tflops": round(fl / lib_us / 1e6, 1),
                 "fused_f16_us": round(f16_us, 2), "fused_f16_tflops": round(fl / f16_us / 1e6, 1),
                 "fused_over_library": round(lib_us / f16_us, 3),
                 "fused_mxfp8_us": round(f8_us, 2), "fused_mxfp8_tflops": round(fl / f8_us / 1e6, 1)}
            rows.append(r)
            print(f"{tag} {name:6s} {M}x{n}x{k}: library {lib_us:7.1f} us {r['library_plain_tflops']:7.1f} TF/s | fused fp16 "
                  f"{f16_us:7.1f} us {r['fused_f16_tflops']:7.1f} TF/s ({r['fused_over_library']:.2f}x) | fused MX-fp8 {f8_us:7.1f} us "
                  f"{r['fused_mxfp8_tflops']:7.1f} TF/s", flush=True)
        fl_all = sum(2.0 * r["M"] * r["N"] * r["K"] for r in rows)
        out["sets"].append({"shapes": tag, "rows": rows,
                            "layer_library_tflops": round(fl_all / sum(r["library_plain_us"] for r in rows) / 1e6, 1),
                            "layer_fused_f16_tflops": round(fl_all / sum(r["fused_f16_us"] for r in rows) / 1e6, 1),
                            "layer_fused_mxfp8_tflops": round(fl_all / sum(r["fused_mxfp8_us"] for r in rows) / 1e6, 1)})
    dst = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                             "gpurun_out", "r04_lib_gemm_ref.json")
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    json.dump(out, open(dst, "w"), indent=1)
    print("wrote", dst)


if __name__ == "__main__":
    main()

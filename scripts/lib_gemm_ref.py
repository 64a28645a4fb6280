"""Yardstick, not product code: what the vendor GEMM library (hipBLASLt/rocBLAS through torch.matmul)
reaches on the encoder's four GEMM shapes, next to this repo's gemm_f16 kernel on the same shapes.
The library numbers are for a *plain* fp16 GEMM (fp16 out, no bias/GELU/LayerScale/residual/RoPE
epilogue), so they bound what a fused kernel could hope for; they are not a like-for-like timing."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbas_amd import _lib  # noqa: E402

lib = _lib.load()
fn = lib.cbas_debug_gemm_bench
fn.restype = C.c_int
fn.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_float), C.POINTER(C.c_ulonglong)]

M = int(sys.argv[1]) if len(sys.argv) > 1 else 12864
shapes = [("qkv", M, 2304, 768, 0), ("oproj", M, 768, 768, 100), ("up", M, 3072, 768, 0), ("down", M, 768, 3072, 100)]
dev = torch.device("cuda")
for name, m, n, k, eoff in shapes:
    a = torch.randn(m, k, device=dev, dtype=torch.float16)
    w = torch.randn(n, k, device=dev, dtype=torch.float16)
    for _ in range(5):
        torch.matmul(a, w.t())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        torch.matmul(a, w.t())
    e1.record()
    torch.cuda.synchronize()
    lib_us = e0.elapsed_time(e1) / 50 * 1e3
    ms, cs = C.c_float(), C.c_ulonglong()
    rc = fn(m, n, k, 0 + eoff, 50, C.byref(ms), C.byref(cs))
    mine_us = ms.value * 1e3 if rc == 0 else float("nan")
    fl = 2.0 * m * n * k
    print(f"{name:6s} {m}x{n}x{k}: library {lib_us:7.1f} us {fl / lib_us / 1e6:7.1f} TF/s | "
          f"gemm_f16 (fused epilogue) {mine_us:7.1f} us {fl / mine_us / 1e6:7.1f} TF/s", flush=True)

"""Block timeline (s_memtime stamps: prologue / K loop / epilogue cycles, in-kernel clock) of the fp16 and MX-fp8 ping-pong GEMMs
on the four encoder shapes at M = 12 864 (ViT-B/16, 64 frames), each with its own epilogue, through cbas_debug_gemm_bench's
stamp option (tile = 1000 + 100 residual | 200 q|k|v | 0 GELU, + 500 MX-fp8; tile id 0 = the planner's choice).
usage: python scripts/gemm_stamps.py [iters [M]]            (the library prints the stamp lines on stdout)"""
import os as _os
_os.environ.setdefault("CBAS_BUILD_DEBUG", "1")      # bring-up entry points: the debug build of the library
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbas_amd import _lib
lib = _lib.load()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
M = int(sys.argv[2]) if len(sys.argv) > 2 else 12864
for f8 in (0, 500):
    for name, n, k, code in [("up + GELU", 3072, 768, 0), ("q|k|v + RoPE", 2304, 768, 200), ("o_proj + residual", 768, 768, 100),
                             ("down + residual", 768, 3072, 100)]:
        ms, cs = C.c_float(), C.c_ulonglong()
        print(f"{'MX-fp8' if f8 else 'fp16'} {name} ({M} x {n} x {k}):", flush=True)
        rc = lib.cbas_debug_gemm_bench(M, n, k, 1000 + code + f8, iters, C.byref(ms), C.byref(cs))
        kt = k // (128 if f8 else 64)
        print(f"   rc={rc}  {ms.value * 1e3:.1f} us per launch = {2 * M * n * k / ms.value / 1e9:.0f} TF/s; {kt} K-tiles per tile", flush=True)

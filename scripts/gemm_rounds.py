"""Launch time of the ping-pong GEMM against the number of tile rounds (256 CUs): fixed cost vs per-round cost.

    python scripts/gemm_rounds.py [N K eoff]      (eoff 0: GELU epilogue, 100: residual, 200: q|k|v; + 500: MX-fp8 operands)
"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbas_amd import _lib
lib = _lib.load()
fn = lib.cbas_debug_gemm_bench
fn.restype = C.c_int
fn.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_float), C.POINTER(C.c_ulonglong)]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
K = int(sys.argv[2]) if len(sys.argv) > 2 else 768
eoff = int(sys.argv[3]) if len(sys.argv) > 3 else 0
tn = N // 256
for panels in (1, 5, 10, 21, 22, 32, 42, 43, 50, 51, 64, 65, 85, 86, 128):
    M = panels * 256
    ms, cs = C.c_float(), C.c_ulonglong()
    rc = fn(M, N, K, 13 + eoff, 50, C.byref(ms), C.byref(cs))
    tiles = panels * tn
    print(f"panels {panels:4d} tiles {tiles:5d} rounds {tiles/256:6.2f}  {ms.value*1e3:8.1f} us  {2.0*M*N*K/ms.value/1e9:7.1f} TF/s rc={rc}", flush=True)

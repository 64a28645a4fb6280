"""Block timeline (s_memtime stamps: prologue / K loop / epilogue) of GEMM tile variants."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbas_amd import _lib
lib = _lib.load()
fn = lib.cbas_debug_gemm_bench
fn.restype = C.c_int
fn.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_float), C.POINTER(C.c_ulonglong)]
tiles = [int(t) for t in sys.argv[1].split(",")]           # usage: gemm_stamps.py 13,17 [iters [M]]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
M = int(sys.argv[3]) if len(sys.argv) > 3 else 12864
shapes = os.environ.get("CBAS_STAMP_SHAPES", "up,qkv,oproj,down").split(",")
f8 = 500 if os.environ.get("CBAS_STAMP_F8") == "1" else 0           # the MX-fp8 kernels instead
f8 += 2000 if os.environ.get("CBAS_STAMP_LN") == "1" else 0          # the LayerNorm-fold forms of the epilogues
for name, m, n, k, eoff in [("up (gelu)", M, 3072, 768, 0), ("qkv (rope)", M, 2304, 768, 200), ("oproj", M, 768, 768, 100), ("down", M, 768, 3072, 100)]:
    if name.split()[0] not in shapes:
        continue
    for t in tiles:
        ms, cs = C.c_float(), C.c_ulonglong()
        print(f"{name} tile {t}:", flush=True)
        rc = fn(m, n, k, 1000 + t + eoff + f8, iters, C.byref(ms), C.byref(cs))
        print(f"   {ms.value*1e3:.1f} us rc={rc}", flush=True)

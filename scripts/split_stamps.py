"""Block timeline (s_memtime stamps: prologue / K loop / epilogue) of the precision-4 GEMM forms.
usage: split_stamps.py TILE[,TILE...] [iters [M]]   TILE = 0 planner, 128/160/192/256 rows, -1 = the 128 x 128 8-wave kernel"""
import os as _os
_os.environ.setdefault("CBAS_BUILD_DEBUG", "1")      # bring-up entry points: the debug build of the library
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbas_amd import _lib
lib = _lib.load()
fn = lib.cbas_debug_gemm_split_bench
fn.restype = C.c_int
fn.argtypes = [C.c_int] * 6 + [C.POINTER(C.c_float)]
tiles = [int(t) for t in sys.argv[1].split(",")]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
M = int(sys.argv[3]) if len(sys.argv) > 3 else 12864
shapes = os.environ.get("CBAS_STAMP_SHAPES", "up,qkv,oproj,down").split(",")
for name, n, k, epi in [("up (gelu)", 3072, 768, 3), ("qkv (rope)", 2304, 768, 1), ("oproj", 768, 768, 2), ("down", 768, 3072, 2)]:
    if name.split()[0] not in shapes:
        continue
    for t in tiles:
        ms = C.c_float()
        print(f"{name} tile {t}:", flush=True)
        rc = fn(M, n, k, epi, t, iters, C.byref(ms))
        print(f"   {ms.value * 1e3:.1f} us rc={rc}  {3 * 2 * M * n * k / ms.value / 1e9:.0f} TF/s executed", flush=True)

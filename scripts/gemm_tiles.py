"""GEMM tile-variant micro-benchmark on the encoder's real shapes (bit-exactness vs tile 1)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbas_amd import _lib
lib = _lib.load()
fn = lib.cbas_debug_gemm_bench
fn.restype = C.c_int
fn.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_float), C.POINTER(C.c_ulonglong)]
tiles = [int(t) for t in (sys.argv[1].split(",") if len(sys.argv) > 1 else "1,2,3,4".split(","))]
M = int(sys.argv[2]) if len(sys.argv) > 2 else 12864
shapes = [("qkv", M, 2304, 768, 0), ("oproj", M, 768, 768, 100), ("up", M, 3072, 768, 0), ("down", M, 768, 3072, 100)]
tot = {t: 0.0 for t in tiles}
for name, m, n, k, eoff in shapes:
    ref = None
    for t in tiles:
        ms, cs = C.c_float(), C.c_ulonglong()
        rc = fn(m, n, k, t + eoff, 20, C.byref(ms), C.byref(cs))
        if rc:
            print(name, t, "rc", rc, lib.cbas_last_error()); continue
        if ref is None: ref = cs.value
        tot[t] += ms.value
        print(f"{name:6s} {m}x{n}x{k} tile={t}: {ms.value*1e3:8.1f} us  {2.0*m*n*k/ms.value/1e9:7.1f} TFLOP/s  bitexact={cs.value == ref}")
print("sum per layer (us):", {t: round(v * 1e3, 1) for t, v in tot.items()})

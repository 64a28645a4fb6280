"""How kernels of two compute lanes share the chip: each component alone, then pairs concurrently on separate streams.

    python scripts/overlap_kernels.py [iters]
perfect overlap: the pair's wall time = max of the two alone; time slicing: = their sum."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbas_amd import _lib
lib = _lib.load()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
names = ["up", "ln", "attn", "down"]


def run(mode):
    ms = (C.c_float * 5)()
    _lib.check(lib.cbas_debug_overlap(mode, iters, ms), "cbas_debug_overlap")
    return [ms[i] * 1e3 for i in range(4)], ms[4] * 1e3 / iters


alone = {}
for b in range(4):
    per, wall = run(1 << b)
    alone[b] = wall
    print(f"{names[b]:5s} alone: {wall:7.1f} us per launch")
for a, b in [(0, 1), (0, 2), (3, 1), (3, 2), (0, 3), (1, 2)]:
    per, wall = run((1 << a) | (1 << b))
    s, m = alone[a] + alone[b], max(alone[a], alone[b])
    print(f"{names[a]:5s} + {names[b]:5s}: wall {wall:7.1f} us per pair   (alone sum {s:6.1f}, max {m:6.1f}) -> "
          f"hidden {100 * (s - wall) / (s - m):5.1f} % of the shorter one;  per-stream {per[a]:6.1f} / {per[b]:6.1f}")

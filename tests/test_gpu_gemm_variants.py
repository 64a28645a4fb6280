"""Every fp16 GEMM tile variant computes bit-identical results (same MFMAs in the same k order): the
ping-pong kernel's tile heights, its planner (uniform / tail-balanced grids) and the 16-wave kernels against
the 128x128 baseline, on shapes with ragged M, the minimum K (two K-tiles), long K and many column tiles."""
import ctypes as C

import pytest

from cbas_amd import _lib

pytestmark = pytest.mark.gpu

SHAPES = [(300, 256, 128), (1000, 512, 256), (257, 256, 4096), (5000, 1024, 1024), (12864, 768, 768), (6432, 3072, 768),
          (12864, 2304, 768), (32928, 1024, 1024), (1, 256, 768), (64, 768, 3072), (37, 3072, 768)]
TILES = [4, 7, 13, 14, 15, 16, 17]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_tile_variants_are_bit_identical(shape):
    lib = _lib.load()
    fn = lib.cbas_debug_gemm_bench
    m, n, k = shape

    def run(tile):
        ms, cs = C.c_float(), C.c_ulonglong()
        _lib.check(fn(m, n, k, tile, 1, C.byref(ms), C.byref(cs)), f"gemm tile {tile}")
        return cs.value

    ref = run(1)
    assert ref != 0
    for t in TILES + ([8, 9] if m <= 64 else []):            # 64x128 and the skinny ring kernel for the CLS-row GEMMs
        assert run(t) == ref, (shape, t)

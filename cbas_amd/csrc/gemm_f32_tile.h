// Tile geometry shared by the exact-fp32 MFMA GEMMs (gemm_f32.hip: classifier head; vit_f32.hip: the precision-3 encoder):
// 128x128 output tile per 256-thread workgroup, 128-byte LDS rows (32 floats of K), 16-byte LDS-DMA staging with the
// XOR swizzle applied on the SOURCE address (the LDS side of the DMA is lane-linear), conflict-free ds_read_b128 fragments.
#pragma once
#include "kernels.h"

namespace f32tile {

constexpr int BM = 128, BN = 128, BKF = 32;      // BKF floats = 128 bytes per LDS row
constexpr int TILE_BYTES = BM * 128;

static __device__ __forceinline__ int xcd_remap(int orig, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (orig >> 3);
}

static __device__ __forceinline__ void stage_tile32(const float* __restrict__ g, int64_t ld, int64_t row0, int64_t max_row,
                                             int k0, char* lds_tile, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int piece = wave * 4 + i;
        const int r = piece * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        int64_t row = row0 + r;
        row = row < max_row ? row : max_row;          // clamp: rows past the end are never stored
        const float* src = g + row * ld + k0 + chunk * 4;
        __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(lds_tile + piece * 1024), 16, 0, 0);
    }
}

static __device__ __forceinline__ f32x4 read_frag32(const char* lds_tile, int row, int chunk) {
    const int off = row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
    return *reinterpret_cast<const f32x4*>(lds_tile + off);
}

}  // namespace f32tile

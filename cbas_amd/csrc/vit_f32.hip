// Precision 3 (and 4): the encoder in the reference's own CPU arithmetic - fp32 operands, fp32 products, fp32 sums - on
// gfx950; precision 4 keeps this file's buffers, attention, LayerNorm and epilogues and forms the GEMM products on the fp16
// matrix pipe from split operands (see "precision 4" below).
//
// The reference's parity target is its CPU path: autocast is disabled there (backend/cbas.py:433-434), so every
// nn.Linear of [tf] modeling_dinov3_vit.py runs as an fp32 GEMM, attention as fp32 SDPA, LayerNorm / GELU / RoPE in
// fp32.  The fp16-operand kernels meet the 1e-3 CLS bar but move probabilities by ~1e-2, which flips near-tie labels;
// this file is the mode that does not: every contraction on v_mfma_f32_16x16x4_f32 (f32 in / f32 accumulate, bit for
// bit a k-ordered fmaf chain: MI355X_MICROARCH.md "Matrix cores"), every activation buffer fp32, and the element-wise
// steps written with the reference's rounding points (no FMA contraction across what are separate torch ops).
//
//   gemm_f32_vit_kernel<EPI>   every nn.Linear of the ViT with its epilogue fused:
//        EPI_PATCH  Conv2d k=s=16 as im2col GEMM + bias (+ DINOv2 position embedding), token scatter   [tf]:82-89
//        EPI_QKV    + bias, RoPE on the patch rows of q and k ([tf]:238-268, rotate_half :203-207), q * 1/8 (exact)
//        EPI_RESID  x += (acc + bias) * lambda                                        [tf]:342-343, :432-443
//        EPI_GELU   exact-erf GELU(acc + bias)                                        [tf]:356
//   attention_f32_kernel       softmax(q k^T) v per (frame, head), online softmax over 64-key blocks   [tf]:210-234
//   layernorm_f32_kernel       [tf]:404,410
//   im2col_*_f32, pack_patch_weight_f32   backend/cbas.py:431 (green / 255.0 in double, then float), :674
//
// The matrix pipe runs at 1/16 of the fp16 rate here (157 TFLOP/s peak), so the GEMM is MFMA-bound by a wide margin
// and the geometry is the simple one of gemm_f32.hip: 128x128 tiles, 4 waves, two LDS buffers, LDS-DMA staging.
#include <stdlib.h>
#include <type_traits>
#include "kernels.h"
#include "gemm_f32_tile.h"
#include "vit32_epilogue.h"

namespace {

using namespace f32tile;

template <int EPI, bool SPLIT>
__global__ __launch_bounds__(256, 2) void gemm_f32_vit_kernel(Gemm32VitParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BUF_BYTES = 2 * TILE_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    const int tiles_n = p.N / BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int64_t row0 = (int64_t)tm * BM;
    const int col0 = tn * BN;
    const int nk = p.K / BKF;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * BUF_BYTES;
        stage_tile32(p.A, p.lda, row0, p.M - 1, kt * BKF, base, wave, lane);
        stage_tile32(p.W, p.K, col0, p.N - 1, kt * BKF, base + TILE_BYTES, wave, lane);
    };
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int frow = lane & 15, fchunk = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const char* At = smem + cur * BUF_BYTES;
        const char* Wt = At + TILE_BYTES;
        if constexpr (SPLIT) {
            // one K-tile = 32 k = ONE fp16 MFMA k-step; both operands arrive split (see above): chunk g = hi, 4 + g = lo
            f16x8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ah[i] = __builtin_bit_cast(f16x8, read_frag32(At, wr * 64 + i * 16 + frow, fchunk));
                al[i] = __builtin_bit_cast(f16x8, read_frag32(At, wr * 64 + i * 16 + frow, 4 + fchunk));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bh[j] = __builtin_bit_cast(f16x8, read_frag32(Wt, wc * 64 + j * 16 + frow, fchunk));
                bl[j] = __builtin_bit_cast(f16x8, read_frag32(Wt, wc * 64 + j * 16 + frow, 4 + fchunk));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j], ah[i], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], al[i], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], ah[i], acc[i][j], 0, 0, 0);
                }
        } else {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                f32x4 a[4], b[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = read_frag32(At, wr * 64 + i * 16 + frow, kk * 4 + fchunk);
#pragma unroll
                for (int j = 0; j < 4; ++j) b[j] = read_frag32(Wt, wc * 64 + j * 16 + frow, kk * 4 + fchunk);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j][e], a[i][e], acc[i][j], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    if constexpr (SPLIT) {
        const float unscale = 1.0f / (p.a_scale * p.w_scale);     // powers of two: exact
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] *= unscale;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t m = row0 + wr * 64 + i * 16 + (lane & 15);
        if (m < p.M) vit32_epilogue_row<EPI>(p, (int)m, col0 + wc * 64, lane, acc[i]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// precision 4, large M.  The split loop is 48 MFMAs (768 cycles) per K-tile and wave: the kernel above, 4 waves per
// workgroup, spends most of a K-tile waiting for the LDS-DMA round trip of the one K-tile it has in flight (PMC: matrix
// pipes 35-39 % busy at 2.2 GHz).  What helped, measured (ViT-B/16 batch 64, ms per step, same device, bit-identical
// results): more waves per CU - THIS form: the same 128 x 128 tile and two LDS buffers, hence still two workgroups per CU,
// but 8 waves as 4 (M) x 2 (N) with 32 x 64 per wave (a 64-column group per wave: the RoPE partner stays in the lane), one
// raw s_barrier per K-tile: 7.64 against 7.96.  What did not: more K-tiles in flight at one workgroup per CU - a
// 256 x 128 tile with a 3-stage ring: 9.90 against 9.35; this tile with a 4-stage ring (three in flight, counted vmcnt):
// 9.90 against 8.70 - a single workgroup in barrier lockstep loses more than the deeper prefetch wins.
//   iteration kt:  wait own pieces of tile kt (vmcnt 0) -> barrier (everybody's pieces landed AND everybody is past its
//                  reads of tile kt-1) -> issue tile kt+1 into tile kt-1's buffer -> 12 fragment reads + 24 MFMAs
// Same products in the same order per output element as the 4-wave kernel: bit-identical results, so the choice by M
// keeps batch invariance exact.  (The fp32 loop of precision 3, 4 096 MFMA cycles per K-tile, does NOT gain from 8 waves:
// 21.19 against 20.41 ms per step, measured with the same kernel template; it stays on the 4-wave kernel.)
// ---------------------------------------------------------------------------------------------------------------------
constexpr int S8_STAGES = 2;

__device__ __forceinline__ void stage_half8(const float* __restrict__ g, int64_t ld, int64_t row0, int64_t max_row, int k0,
                                            char* lds_tile, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {                          // 128 rows = 16 pieces of 8 rows over 8 waves
        const int piece = wave * 2 + i;
        const int r = piece * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        int64_t row = row0 + r;
        row = row < max_row ? row : max_row;
        const float* src = g + row * ld + k0 + chunk * 4;
        __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(lds_tile + piece * 1024), 16, 0, 0);
    }
}

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_split8_kernel(Gemm32VitParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE = 2 * TILE_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int tiles_n = p.N / BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int64_t row0 = (int64_t)tm * BM;
    const int col0 = tn * BN;
    const int nk = p.K / BKF;

    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * STAGE;
        stage_half8(p.A, p.lda, row0, p.M - 1, kt * BKF, base, wave, lane);
        stage_half8(p.W, p.K, col0, p.N - 1, kt * BKF, base + TILE_BYTES, wave, lane);
    };
    stage(0, 0);

    const int frow = lane & 15, fchunk = lane >> 4;
    int cur = 0, nxt = 1;                                   // buffer of tile kt / of tile kt + 1
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk) stage(nxt, kt + 1);
        const char* At = smem + cur * STAGE;
        const char* Wt = At + TILE_BYTES;
        f16x8 ah[2], al[2], bh[4], bl[4];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ah[i] = __builtin_bit_cast(f16x8, read_frag32(At, wr * 32 + i * 16 + frow, fchunk));
            al[i] = __builtin_bit_cast(f16x8, read_frag32(At, wr * 32 + i * 16 + frow, 4 + fchunk));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bh[j] = __builtin_bit_cast(f16x8, read_frag32(Wt, wc * 64 + j * 16 + frow, fchunk));
            bl[j] = __builtin_bit_cast(f16x8, read_frag32(Wt, wc * 64 + j * 16 + frow, 4 + fchunk));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j], ah[i], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], al[i], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], ah[i], acc[i][j], 0, 0, 0);
            }
        cur ^= 1;
        nxt ^= 1;
    }

    const float unscale = 1.0f / (p.a_scale * p.w_scale);     // powers of two: exact
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] *= unscale;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int64_t m = row0 + wr * 32 + i * 16 + (lane & 15);
        if (m < p.M) vit32_epilogue_row<EPI>(p, (int)m, col0 + wc * 64, lane, acc[i]);
    }
}

template <int EPI>
int launch_split8(const Gemm32VitParams& p, hipStream_t stream) {
    constexpr int lds = S8_STAGES * 2 * TILE_BYTES;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_split8_kernel<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                lds) != hipSuccess)
            return -2;
        attr_set = true;
    }
    const int64_t grid = ((int64_t)(p.M + BM - 1) / BM) * (p.N / BN);
    if (grid <= 0 || grid > 0x7fffffff) return -1;
    hipLaunchKernelGGL((gemm_split8_kernel<EPI>), dim3((unsigned)grid), dim3(512), lds, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---------------------------------------------------------------------------------------------------------------------
// precision 4, M <= 256: the CLS rows of the pruned last layer (q, o_proj, up, down on n frames; [tf]:540-541,
// backend/cbas.py:677).  There are only N / 64 x ceil(M / 64) tiles of work and K is up to 3 072: on the 2-buffer kernel
// above every K-tile pays one global -> LDS round trip (K = 3 072: 96 of them, 63 us).  Here, as in gemm_f16_skinny.hip, a
// workgroup is 4 waves x (16 rows x 64 columns) over an 8-slot LDS ring filled by LDS-DMA with 6 K-tiles in flight under
// a counted vmcnt and ONE raw barrier per K-tile.  Same products in the same order per output element as the other
// split kernels: bit-identical rows, so a frame's CLS row does not depend on its batch.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int SS_NS = 8;                                // ring slots
constexpr int SS_STAGE = 2 * 64 * 128;                  // A 64 rows + W 64 rows, 128 bytes (one K-tile: 32 k, hi | lo) each

template <int EPI>
__global__ __launch_bounds__(256) void gemm_split_skinny_kernel(Gemm32VitParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col0 = blockIdx.x * 64, row0 = blockIdx.y * 64;
    const int nk = p.K / BKF;

    const int lrow = lane >> 3;
    const float* a_src[2];
    const float* b_src[2];
    int lds_off[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int r = (wave + 4 * s) * 8 + lrow;
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        int64_t ar = row0 + r;
        ar = ar < p.M ? ar : p.M - 1;                   // rows past the end are never stored
        a_src[s] = p.A + ar * p.lda + chunk * 4;
        b_src[s] = p.W + (int64_t)(col0 + r) * p.K + chunk * 4;
        lds_off[s] = (wave + 4 * s) * 1024;
    }
    auto stage = [&](int kt) {
        char* base = smem + (kt % SS_NS) * SS_STAGE;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            __builtin_amdgcn_global_load_lds(GLB_PTR(a_src[s] + kt * BKF), LDS_PTR(base + lds_off[s]), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(GLB_PTR(b_src[s] + kt * BKF), LDS_PTR(base + 8192 + lds_off[s]), 16, 0, 0);
        }
    };
    const int frow = lane & 15, fchunk = lane >> 4;
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int kt = 0; kt < SS_NS - 1 && kt < nk; ++kt) stage(kt);           // 7 K-tiles in flight
    for (int kt = 0; kt < nk; ++kt) {
        // K-tile kt has landed once at most the 6 younger stages (4 DMAs each per wave) are outstanding
        if (kt + SS_NS - 2 < nk) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                      // ... for every wave's pieces; and every wave
        if (kt + SS_NS - 1 < nk) stage(kt + SS_NS - 1);                    // is done with slot (kt-1) % NS, refilled here
        const char* At = smem + (kt % SS_NS) * SS_STAGE;
        const char* Wt = At + 8192;
        const f16x8 ah = __builtin_bit_cast(f16x8, read_frag32(At, wave * 16 + frow, fchunk));
        const f16x8 al = __builtin_bit_cast(f16x8, read_frag32(At, wave * 16 + frow, 4 + fchunk));
        f16x8 bh[4], bl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bh[j] = __builtin_bit_cast(f16x8, read_frag32(Wt, j * 16 + frow, fchunk));
            bl[j] = __builtin_bit_cast(f16x8, read_frag32(Wt, j * 16 + frow, 4 + fchunk));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j], ah, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], al, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], ah, acc[j], 0, 0, 0);
    }
    const float unscale = 1.0f / (p.a_scale * p.w_scale);     // powers of two: exact
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] *= unscale;
    const int m = row0 + wave * 16 + (lane & 15);
    if (m < p.M) vit32_epilogue_row<EPI>(p, m, col0, lane, acc);
}

template <int EPI>
int launch_split_skinny(const Gemm32VitParams& p, hipStream_t stream) {
    constexpr int lds = SS_NS * SS_STAGE;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_split_skinny_kernel<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                lds) != hipSuccess)
            return -2;
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_split_skinny_kernel<EPI>), dim3((unsigned)(p.N / 64), (unsigned)((p.M + 63) / 64)), dim3(256), lds, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

template <int EPI, bool SPLIT>
int launch_vit32s(const Gemm32VitParams& p, hipStream_t stream) {
    constexpr int lds = 4 * TILE_BYTES;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_vit_kernel<EPI, SPLIT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return -2;
        attr_set = true;
    }
    const int64_t grid = ((int64_t)(p.M + BM - 1) / BM) * (p.N / BN);
    if (grid <= 0 || grid > 0x7fffffff) return -1;
    hipLaunchKernelGGL((gemm_f32_vit_kernel<EPI, SPLIT>), dim3((unsigned)grid), dim3(256), lds, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// which precision-4 GEMM forms are in use: bit 0 the ping-pong form (M > 256), bit 1 the skinny form (M <= 256); the
// rest falls to the 128 x 128 kernels.  -1 = the environment's choice (CBAS_SPLIT_PP / CBAS_SPLIT_SKINNY, default on);
// set by cbas_enc_debug_option("split_kernels") - process-wide, for the bit-identity tests
int g_split_forms = -1;
int split_forms() {
    static const int env = [] {
        const char* a = getenv("CBAS_SPLIT_PP");
        const char* b = getenv("CBAS_SPLIT_SKINNY");
        return ((!a || a[0] != '0') ? 1 : 0) | ((!b || b[0] != '0') ? 2 : 0);
    }();
    return g_split_forms >= 0 ? g_split_forms : env;
}

template <int EPI>
int launch_vit32(const Gemm32VitParams& p, hipStream_t stream) {
    if (!p.split) return launch_vit32s<EPI, false>(p, stream);
    if (p.M <= 256) {
        const bool skinny = split_forms() & 2;
        return skinny && EPI != EPI_PATCH ? launch_split_skinny<EPI>(p, stream) : launch_vit32s<EPI, true>(p, stream);
    }
    // large M: the ping-pong kernel's split form (gemm_f16_8ph.hip) where the shape is one of its; all three kernels form
    // the same products in the same order, so the choice does not show in the results
    if (split_forms() & 1) {
        const int rc = launch_gemm_split_pp((GemmEpilogue)EPI, p, stream);
        if (rc != -1) return rc;
    }
    return launch_split8<EPI>(p, stream);
}

// ---------------------------------------------------------------------------------------------
// ingest / weights / LayerNorm
// ---------------------------------------------------------------------------------------------
// One thread = one patch row of `ps` pixels; A[m][i*16 + j] with the 16x16 slot layout of im2col_u8_kernel (slots
// with i >= ps or j >= ps stay zero: rows i >= ps are never written, the buffer is zeroed at allocation).
template <typename SRC>
__global__ void im2col_f32out_kernel(const SRC* __restrict__ frames, int n, int64_t frame_stride, int64_t row_stride,
                                     int64_t pixel_stride, float* __restrict__ A, int nh, int nw, int ps, int split) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)n * nh * ps * nw;
    if (gid >= total) return;
    const int px = gid % nw;
    const int64_t t1 = gid / nw;
    const int y = t1 % (nh * ps);
    const int b = t1 / (nh * ps);
    const int py = y / ps, i = y - py * ps;
    const SRC* src = frames + b * frame_stride + (int64_t)y * row_stride + (int64_t)px * ps * pixel_stride;
    float* dst = A + ((int64_t)b * nh * nw + (int64_t)py * nw + px) * 256 + i * 16;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = q * 4 + e;
            float f = 0.f;
            if (j < ps) {
                if (sizeof(SRC) == 1) f = (float)((double)src[j * pixel_stride] / 255.0);     // numpy: uint8 / 255.0 -> float64; .float()
                else f = (float)src[j * pixel_stride];
            }
            v[e] = f;
        }
        if (split) store_split4(dst - i * 16, i * 16 + q * 4, v, 1.0f);
        else reinterpret_cast<f32x4*>(dst)[q] = v;
    }
}

__global__ void write_prefix32_kernel(float* __restrict__ x, const float* __restrict__ prefix, int n, int n_prefix, int D, int T) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t per = (int64_t)n_prefix * (D / 4);
    if (gid >= (int64_t)n * per) return;
    const int b = gid / per;
    const int rem = gid - (int64_t)b * per;
    const int r = rem / (D / 4), c = rem - r * (D / 4);
    reinterpret_cast<f32x4*>(x + ((int64_t)b * T + r) * D)[c] = reinterpret_cast<const f32x4*>(prefix + (int64_t)r * D)[c];
}

__global__ void pack_patch_weight_f32_kernel(const float* __restrict__ w, float* __restrict__ out, int D, int ps) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;     // over D*256
    if (idx >= D * 256) return;
    const int d = idx >> 8, k = idx & 255, i = k >> 4, j = k & 15;
    float v = 0.f;
    if (i < ps && j < ps) {
        const float* base = w + (size_t)d * 3 * ps * ps + i * ps + j;
        v = (float)(((double)base[0] + (double)base[ps * ps]) + (double)base[2 * ps * ps]);     // the 3 identical input channels
    }
    out[idx] = v;
}

// precision 4: fp32 weight [N][K] -> split format (same byte size), values scaled by `scale` (a power of two); K % 32 == 0
__global__ void pack_split_weight_kernel(const float* __restrict__ w, float* __restrict__ out, int64_t N, int K, float scale) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;          // one thread = 4 consecutive k of one row
    const int per_row = K >> 2;
    if (i >= N * per_row) return;
    const int64_t row = i / per_row;
    const int c = (int)(i - row * per_row) * 4;
    store_split4<true>(out + row * K, c, *reinterpret_cast<const f32x4*>(w + row * K + c), scale);
}

// one wave per row, two-pass statistics in registers
template <int NV>
__global__ __launch_bounds__(256) void layernorm_f32_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ out, int M, int D,
                                                            float eps, int split) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (size_t)row * ldx;
    const int nvec = D >> 2;
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int idx = lane + 64 * k;
        v[k] = (idx < nvec) ? reinterpret_cast<const f32x4*>(xr)[idx] : f32x4{0.f, 0.f, 0.f, 0.f};
        // add_np: the same sums in the same order, but not paired through a cross-half v_pk_mov_b32 / v_pk_add_f32 (common.h)
        s += add_np(add_np(v[k][0], v[k][1]), add_np(v[k][2], v[k][3]));
    }
    const float mean = wave_sum_dpp(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int idx = lane + 64 * k;
        if (idx < nvec) {
#pragma clang fp contract(off)            // four squares, three adds - as before (the packed form had no FMA either)
            const f32x4 d = v[k] - mean;
            const f32x4 sq = d * d;
            q += add_np(add_np(sq[0], sq[1]), add_np(sq[2], sq[3]));
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum_dpp(q) / (float)D + eps);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int idx = lane + 64 * k;
        if (idx < nvec) {
            const f32x4 g = reinterpret_cast<const f32x4*>(gamma)[idx];
            const f32x4 bb = reinterpret_cast<const f32x4*>(beta)[idx];
            const f32x4 y = (v[k] - mean) * rstd * g + bb;
            if (split) store_split4(out + (size_t)row * D, idx * 4, y, 1.0f);
            else reinterpret_cast<f32x4*>(out + (size_t)row * D)[idx] = y;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Attention, fp32.  One workgroup = `nwaves` 16-query tiles of one (frame, head); K and V stream through LDS in blocks
// of 64 keys (two buffers, staged by 16-byte LDS-DMA with the swizzles applied on the source address).
//   S^T = K Q^T   on v_mfma_f32_16x16x4_f32 with the key on the MFMA row: a lane holds, for ONE query (lane & 15), the
//                 keys 16 kt + 4 (lane >> 4) + r - the softmax is in-lane plus two cross-lane steps, and those registers
//                 ARE the B operand of the next product (k-slot (lane >> 4) of step (kt, r) <-> key 16 kt + 4 (lane >> 4) + r;
//                 the A operand reads V with the same map, so the sum over keys is unchanged)
//   O^T = V^T P^T one ds_read_b32 of V per MFMA
// Online softmax over the key blocks (what torch's CPU flash kernel does too), everything fp32, expf from the device
// library (accurate form).  K image: chunk c (16 B) of row r at position c ^ (r & 15) - ds_read_b128 of one chunk
// column over 16 rows is conflict-free.  V image: chunk c of row r at c ^ (4 * ((r >> 2) & 1)) - the ds_read_b32 of 16
// consecutive floats from rows 4g + r, g = 0..3, lands the two 16-lane halves of a 32-lane group on different banks.
// ---------------------------------------------------------------------------------------------
constexpr int AKB = 64;                     // keys per block
constexpr int AIMG = AKB * 256;             // bytes of one K or V block image

__global__ __launch_bounds__(512, 2) void attention_f32_kernel(const float* __restrict__ qkv, const float* __restrict__ q_cls,
                                                               float* __restrict__ out, int T, int D, int n_heads, int qblocks,
                                                               float split_scale, int npairs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = blockDim.x >> 6;
    // XCD-aware workgroup -> (frame, head, query block) map (r5).  The hardware hands consecutive workgroup ids to the eight XCDs
    // round-robin, so with the plain map (id = pair * qblocks + qb) the query blocks of one (frame, head) land on DIFFERENT
    // XCDs and each XCD's L2 fetches that head's K and V from the fabric again - and the ablation of this kernel
    // shows 60 % of its time is exactly that stream.  Here XCD x's k-th workgroup takes pair x + 8 (k / qblocks), query block
    // k % qblocks: the query blocks of a pair run back to back on ONE XCD and the second one finds K and V in its L2.
    const int xcd_k = blockIdx.x >> 3;
    const int pair = (blockIdx.x & 7) + 8 * (xcd_k / qblocks), qb = xcd_k % qblocks;
    if (pair >= npairs) return;                                        // the grid is rounded up to 8 x qblocks (whole workgroups leave)
    const int b = pair / n_heads, hd = pair - b * n_heads;
    const size_t ld = (size_t)3 * D;
    const float* qbase = qkv + (size_t)b * T * ld + hd * 64;
    const float* kbase = qbase + D;
    const float* vbase = qbase + 2 * D;
    const int g = lane >> 4, li = lane & 15;
    const int nq = q_cls ? 1 : T;
    const float* qsrc = q_cls ? q_cls + (size_t)b * D + hd * 64 : qbase;
    const size_t qld = q_cls ? 0 : ld;
    const int qt = qb * nwaves + wave;
    const bool active = qt * 16 < nq;                      // wave-uniform
    const int q = qt * 16 + li;
    const int qrow = q < nq ? q : nq - 1;
    f32x4 qf[4];                                           // Q[q][16c + 4g + e]
#pragma unroll
    for (int c = 0; c < 4; ++c) qf[c] = *reinterpret_cast<const f32x4*>(qsrc + (size_t)qrow * qld + 16 * c + 4 * g);

    const int nkb = (T + AKB - 1) / AKB;
    auto stage = [&](int buf, int kb) {
        char* base = smem + buf * 2 * AIMG;
        for (int p = wave; p < 32; p += nwaves) {          // 16 pieces (4 rows x 256 B) per image
            const int img = p >> 4, piece = p & 15;
            const int r = piece * 4 + g;
            const int key = kb * AKB + r;
            const int ks = key < T ? key : T - 1;          // rows past T re-read row T-1: finite, masked below
            const int chunk = img ? (li ^ (((r >> 2) & 1) << 2)) : (li ^ (r & 15));
            const float* src = (img ? vbase : kbase) + (size_t)ks * ld + chunk * 4;
            __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(base + img * AIMG + piece * 1024), 16, 0, 0);
        }
    };
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    float mrun = -INFINITY, lrun = 0.f;
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int kb = 0; kb < nkb; ++kb) {
        const int cur = kb & 1;
        if (kb + 1 < nkb) stage(cur ^ 1, kb + 1);
        if (active) {
            const char* Ks = smem + cur * 2 * AIMG;
            const char* Vs = Ks + AIMG;
            // key tiles of this block that hold at least one real key (wave-uniform; 4 except in the last block: T = 201
            // leaves 9 keys = one tile there, so three quarters of that block's MFMAs are skipped)
            const int left = T - kb * AKB;
            const int nkt = left >= AKB ? 4 : (left + 15) >> 4;
            f32x4 s[4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                if (kt >= nkt) { s[kt] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY}; continue; }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const f32x4 kf = *reinterpret_cast<const f32x4*>(Ks + (kt * 16 + li) * 256 + (((4 * c + g) ^ li) << 4));
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[e], qf[c][e], acc, 0, 0, 0);
                }
                s[kt] = acc;
            }
            // s[kt][r] = S[q][key = kb*64 + kt*16 + 4g + r]   (q already carries the 1/8 scale)
            float bm = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (kb * AKB + kt * 16 + 4 * g + r >= T) s[kt][r] = -INFINITY;
                    bm = fmaxf(bm, s[kt][r]);
                }
            bm = xor16_max(bm);
            bm = xor32_max(bm);
            const float mnew = fmaxf(mrun, bm);             // finite: every block holds at least one real key
            const float alpha = expf(mrun - mnew);          // first block: exp(-inf) = 0
            float psum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s[kt][r] = expf(s[kt][r] - mnew);
                    psum += s[kt][r];
                }
            lrun = lrun * alpha + psum;                     // per-lane partial; the four g lanes are pooled at the end
            mrun = mnew;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[dt] *= alpha;
            const int vsw = (g & 1) << 2;                   // ((row >> 2) & 1) << 2 for row = 16 kt + 4g + r
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                if (kt >= nkt) continue;                    // probabilities of a padded tile are exactly 0
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const char* vrow = Vs + (kt * 16 + 4 * g + r) * 256 + ((li & 3) << 2);
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        const float vf = *reinterpret_cast<const float*>(vrow + (((dt * 4 + (li >> 2)) ^ vsw) << 4));
                        o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf, s[kt][r], o[dt], 0, 0, 0);
                    }
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (!active) return;
    lrun = xor16_add(lrun);
    lrun = xor32_add(lrun);
    if (q < nq) {
        float* row = out + (q_cls ? (size_t)b : (size_t)b * T + q) * D;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const f32x4 v = o[dt] / lrun;
            if (split_scale > 0.f) store_split4(row, hd * 64 + 16 * dt + 4 * g, v, split_scale);   // precision 4: o_proj's A operand
            else *reinterpret_cast<f32x4*>(row + hd * 64 + 16 * dt + 4 * g) = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Attention, precision 4: the structure of attention_f32_kernel (64-key blocks through a 2-deep LDS ring, S^T = K Q^T so
// that a lane holds its query's keys, online softmax in fp32 with the accurate expf) with both products on the fp16 pipe
// from split operands: S^T = K_hi Q_hi + K_hi Q_lo + K_lo Q_hi (6 MFMAs per 16-key tile), O^T = V_hi P_hi + V_hi P_lo +
// V_lo P_hi (12 per 32-key group), 48 x 16 cycles per block and wave against 128 x 32.  q, k, v arrive split from the
// q|k|v epilogue (store_head_split4); the probabilities are split in registers (x 1024 first: a softmax tail would
// otherwise sit in fp16's subnormals).  K / V images and fragment reads are those of the fp16 kernels (vit_kernels.hip:
// 128-byte rows, k_off / v_off swizzles applied on the DMA source, V through ds_read_b64_tr_b16), once for the hi and
// once for the lo halves: 4 images x 8 KiB per buffer.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int sk_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int sv_off(int row, int col) {   // col in halves
    return row * 128 + ((((col >> 4) ^ ((row >> 1) & 3))) << 5) + ((col & 15) << 1);
}
constexpr int SIMG = AKB * 128;             // bytes of one hi or lo image of a 64-key block

// exp(x) for x <= 0 (softmax arguments), 7 VALU instructions against the device library's 13 (no overflow / underflow
// branches: the argument is clamped at -100, where v_exp_f32 flushes to 0): exp2 of the rounded product x log2(e), corrected
// by the product's exact rounding residual and the low part of log2(e) - as accurate as expf (v_exp_f32 is 1 ulp).
__device__ __forceinline__ float exp_neg(float x) {
    const float v = fmaxf(x, -100.f);
    const float t = v * 1.44269504f;
    float r = fmaf(v, 1.44269504f, -t);
    r = fmaf(v, 1.92596299e-8f, r);
    const float e = __builtin_amdgcn_exp2f(t);
    return fmaf(e, r * 0.693147181f, e);
}

// This form (r4: full key blocks without per-tile tests - one basic block per block, four independent MFMA chains - and the
// score scale folded into the exponential's constants; 67 -> 64 us per layer, rows bit-identical to the earlier form) is
// the kernel beside which the head's expand kernel first returned wrong values (scripts/head_beside_encoder.py, common.h):
// it was the trigger, not the fault - it only reads, and the head kernel was the one that needed changing.
// Measured and removed (r4): a resident form for T <= 256 - one workgroup per (frame, head), one wave per query tile, every
// key block staged up front (4 x 32 KiB), one barrier, then free-running waves: 71-73 us per layer against this ring's
// 64-68 (ViT-B/16 batch 64, bit-identical rows): the kernel is not bound by its barriers or by staging K / V twice.
// exp(d / (ATT_QS ATT_KS)) for d <= 0, finite: exp_neg with the power-of-two score scale folded into its constants (the
// same roundings: scaling by 2^-6 commutes with each of them) and without the clamp (v_exp_f32 flushes a hugely negative
// argument to 0, and the correction term is then 0 x finite).
__device__ __forceinline__ float exp_raw(float d) {
    constexpr float K_HI = 1.44269504f / (ATT_QS * ATT_KS), K_LO = 1.92596299e-8f / (ATT_QS * ATT_KS);
    const float t = d * K_HI;
    float r = fmaf(d, K_HI, -t);
    r = fmaf(d, K_LO, r);
    const float e = __builtin_amdgcn_exp2f(t);
    return fmaf(e, r * 0.693147181f, e);
}

// Measurement aid (scripts/attn_ablate.sh): -DCBAS_ATTN_ABLATE=<bits> compiles parts of a key block OUT - the
// results are then garbage, the point is what each part costs: 1 the softmax's exponentials and hi / lo split, 2 the P.V MFMAs
// and V's LDS reads, 4 V's LDS reads only (the MFMAs run on constant fragments), 8 the S MFMAs and K's LDS reads, 16 the K / V
// stream itself (no LDS-DMA; barriers stay), 32 the final stores, 64 HALF of the K and V fragment reads (what a form with two query
// tiles per wave would save).  0 (default).
#ifndef CBAS_ATTN_ABLATE
#define CBAS_ATTN_ABLATE 0
#endif
// Wave priority raised (s_setprio 1) around: 1 the S MFMAs, 2 the P.V MFMAs, 4 the softmax, 8 the K / V staging - so that the
// waves of a SIMD, which leave every barrier in the same phase, fall out of step.  Default 5 (r5, scripts/attn_variants.sh,
// profiles/r05_attention_priority_ab.json): the softmax - the VALU phase, the kernel's largest pipe demand - is what gains
// (3-4 us of 56-62 per layer on three boxes, rows identical); priority on the MFMA clusters alone does nothing.  The STEP
// does not move with it: at precision 4 the board is at its power limit and the time comes back as clock.
#ifndef CBAS_ATTN_PRIO
#define CBAS_ATTN_PRIO 5
#endif
#ifndef CBAS_ATTN_PRIO_LEVEL
#define CBAS_ATTN_PRIO_LEVEL 1
#endif
#define ATTN_PRIO_ON(bit)  do { if (CBAS_ATTN_PRIO & (bit)) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_setprio(CBAS_ATTN_PRIO_LEVEL); } } while (0)
#define ATTN_PRIO_OFF(bit) do { if (CBAS_ATTN_PRIO & (bit)) { __builtin_amdgcn_s_setprio(0); __builtin_amdgcn_sched_barrier(0); } } while (0)
// Experiment switch: -DCBAS_ATTN_WIDE=1 = ONE workgroup per (frame, head) with a wave per query tile (up to 16 waves; T <= 256),
// K / V staged once per pair instead of once per query block; one such workgroup per CU.  0 (default): measured 59.6 against
// 61.3 us alone, 60.0 against 57.3 with the priorities (profiles/r05_attention_priority_ab.json).
#ifndef CBAS_ATTN_WIDE
#define CBAS_ATTN_WIDE 0
#endif
#if CBAS_ATTN_WIDE
__global__ __launch_bounds__(1024, 1) void attention_split_kernel(
#else
__global__ __launch_bounds__(512, 2) void attention_split_kernel(
#endif
    const float* __restrict__ qkv, const float* __restrict__ q_cls, float* __restrict__ out, int T, int D, int n_heads, int qblocks,
    float out_scale, int npairs, int total_items) {
    extern __shared__ __attribute__((aligned(16))) char smem[];       // [buf][K_hi | K_lo | V_hi | V_lo][64 keys][128 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = blockDim.x >> 6;
    // XCD-aware workgroup -> (frame, head, query block) map (r5).  The hardware hands consecutive workgroup ids to the eight XCDs
    // round-robin, so with the plain map (id = pair * qblocks + qb) the query blocks of one (frame, head) land on DIFFERENT
    // XCDs and each XCD's L2 fetches that head's K and V from the fabric again - and the ablation of this kernel
    // shows 60 % of its time is exactly that stream.  Here XCD x's k-th workgroup takes pair x + 8 (k / qblocks), query block
    // k % qblocks: the query blocks of a pair run back to back on ONE XCD and the second one finds K and V in its L2.
    // PERSISTENT form (r5): the grid may be smaller than the number of (frame, head, query block) items - a multiple of 8, so that
    // a workgroup's items all keep its XCD - and every workgroup walks items id, id + gridDim.x, ...  (grid == total_items: one
    // item each, the earlier form.)
    // The K / V ring runs ON across items: the slot index is a block counter of the workgroup (gb), and during an item's last
    // key block the free slot receives the NEXT item's first block - the stream has no bubble at an item boundary and the
    // first wait of an item finds its block landed under the previous item's tail.
    const size_t ldb = (size_t)3 * D * 4;                              // bytes per token row
    const int g = lane >> 4, li = lane & 15;
    const int nkb = (T + AKB - 1) / AKB;
    const int pr = lane >> 3, pos = lane & 7;
    auto item_pair = [&](int it) { const int k_ = it >> 3; return (it & 7) + 8 * (k_ / qblocks); };
    auto stage = [&](int buf, int pair_, int kb) {                     // block kb of (frame, head) pair_ into ring slot buf
        const int b_ = pair_ / n_heads, hd_ = pair_ - b_ * n_heads;
        const char* kb_ = reinterpret_cast<const char*>(qkv) + (size_t)b_ * T * ldb + (size_t)hd_ * 256 + (size_t)D * 4;
        const char* vb_ = kb_ + (size_t)D * 4;
        char* base = smem + buf * 4 * SIMG;
        for (int p = wave; p < 32; p += nwaves) {                      // 4 images x 8 pieces of 8 rows
            const int img = p >> 3, piece = p & 7;
            const int r = piece * 8 + pr;
            int gr = kb * AKB + r;
            gr = gr < T ? gr : T - 1;                                  // rows past T re-read row T-1: finite, masked below
            const int chunk = img < 2 ? (pos ^ ((r >> 1) & 7)) : ((((pos >> 1) ^ ((r >> 1) & 3)) << 1) | (pos & 1));
            const char* src = (img < 2 ? kb_ : vb_) + (size_t)gr * ldb + (img & 1) * 128 + chunk * 16;
#if !(CBAS_ATTN_ABLATE & 16)
            __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(base + img * SIMG + piece * 1024), 16, 0, 0);
#else
            (void)src; (void)base;
#endif
        }
    };
    // first item with work of this workgroup (items past npairs * qblocks are padding of the XCD map)
    auto next_item = [&](int it) { while (it < total_items && item_pair(it) >= npairs) it += gridDim.x; return it; };
    int item = next_item(blockIdx.x);
    int gb = 0;                                                        // blocks consumed so far: ring slot = gb & 1
    if (item < total_items) stage(0, item_pair(item), 0);
    for (; item < total_items;) {
        const int following = next_item(item + gridDim.x);
        const int xcd_k = item >> 3;
        const int pair = item_pair(item), qb = xcd_k % qblocks;
        const int b = pair / n_heads, hd = pair - b * n_heads;
        const char* qbase = reinterpret_cast<const char*>(qkv) + (size_t)b * T * ldb + (size_t)hd * 256;
        const int nq = q_cls ? 1 : T;
        const int qt = qb * nwaves + wave;
        const bool active = qt * 16 < nq;                                  // wave-uniform
        const int q = qt * 16 + li;
        const int qrow = q < nq ? q : nq - 1;
        const char* qsrc = q_cls ? reinterpret_cast<const char*>(q_cls) + (size_t)b * D * 4 + (size_t)hd * 256 : qbase + (size_t)qrow * ldb;
        f16x8 qh[2], ql[2];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {                                   // k-half h2: d in [32 h2, 32 h2 + 32): chunk 4 h2 + g
            qh[h2] = *reinterpret_cast<const f16x8*>(qsrc + (4 * h2 + g) * 16);
            ql[h2] = *reinterpret_cast<const f16x8*>(qsrc + 128 + (4 * h2 + g) * 16);
        }

        // scores stay in the accumulator's units (x ATT_QS ATT_KS: max and differences scale exactly); the factor is undone
        // inside the exponential's constant.  Masked scores are a large finite negative, not -inf: exp_raw needs no clamp.
        constexpr float NEG = -1.0e30f;
        float mrun = NEG, lrun = 0.f;
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

        auto block = [&](int kb, auto full_c) {
            constexpr bool FULL = decltype(full_c)::value;
            const char* Kh = smem + ((gb + kb) & 1) * 4 * SIMG;
            const char* Kl = Kh + SIMG;
            const char* Vh = Kl + SIMG;
            const char* Vl = Vh + SIMG;
            const int left = T - kb * AKB;
            const int nkt = FULL ? 4 : (left + 15) >> 4;              // key tiles with at least one real key (wave-uniform)
            f32x4 s[4];
#if CBAS_ATTN_ABLATE & 64
            f16x8 kkeep[2][2];
#endif
            ATTN_PRIO_ON(1);
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                if (!FULL && kt >= nkt) { s[kt] = f32x4{NEG, NEG, NEG, NEG}; continue; }
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#if CBAS_ATTN_ABLATE & 8
                acc = f32x4{(float)(kt + li), (float)g, (float)(kb & 3), 1.0f};
#else
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
#if CBAS_ATTN_ABLATE & 64
                    // half the K fragment reads: an odd tile re-uses the even tile's fragments (through an opaque copy, so that its
                    // MFMAs are not folded into the even tile's)
                    if (!(kt & 1)) {
                        kkeep[h2][0] = *reinterpret_cast<const f16x8*>(Kh + sk_off(kt * 16 + li, 4 * h2 + g));
                        kkeep[h2][1] = *reinterpret_cast<const f16x8*>(Kl + sk_off(kt * 16 + li, 4 * h2 + g));
                    }
                    f16x8 kh = kkeep[h2][0], kl = kkeep[h2][1];
                    if (kt & 1) asm volatile("" : "+v"(kh), "+v"(kl));
#else
                    const f16x8 kh = *reinterpret_cast<const f16x8*>(Kh + sk_off(kt * 16 + li, 4 * h2 + g));
                    const f16x8 kl = *reinterpret_cast<const f16x8*>(Kl + sk_off(kt * 16 + li, 4 * h2 + g));
#endif
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl, qh[h2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, ql[h2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, qh[h2], acc, 0, 0, 0);
                }
#endif
                s[kt] = acc;
            }
            ATTN_PRIO_OFF(1);
            ATTN_PRIO_ON(4);
            if (!FULL) {                                              // only a partial block can hold keys past T
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (kb * AKB + kt * 16 + 4 * g + r >= T) s[kt][r] = NEG;
            }
            float bm = NEG;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) bm = fmaxf(bm, s[kt][r]);
            bm = xor16_max(bm);
            bm = xor32_max(bm);
            const float mnew = fmaxf(mrun, bm);
            const float alpha = exp_raw(mrun - mnew);
            float psum = 0.f;
            f16x8 ph[2], pl[2];                                       // P^T of the two 32-key groups, hi and lo halves
#pragma unroll
            for (int grp = 0; grp < 2; ++grp)
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
#if CBAS_ATTN_ABLATE & 1
                        const f16 h = (f16)fminf(fabsf(s[2 * grp + u][r]), 1.0f);      // bounded (the range guard stays quiet), still depends on S
                        ph[grp][4 * u + r] = h;
                        pl[grp][4 * u + r] = h;
                        psum += 1.0f;
#else
                        const float pv = exp_raw(s[2 * grp + u][r] - mnew);
                        psum += pv;
                        const float x = pv * ATT_PS;
                        const f16 h = (f16)x;
                        ph[grp][4 * u + r] = h;
                        pl[grp][4 * u + r] = (f16)(x - (float)h);
#endif
                    }
            lrun = lrun * alpha + psum;
            mrun = mnew;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[dt] *= alpha;
            ATTN_PRIO_OFF(4);
            ATTN_PRIO_ON(2);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                if (!FULL && 2 * s2 >= nkt) continue;                 // both tiles of the group are padding
                const int krow = 32 * s2 + 4 * g + (li >> 2);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
#if CBAS_ATTN_ABLATE & 64
                    const int col = 16 * (dt & ~1) + 4 * (li & 3);        // half the V fragment reads
#else
                    const int col = 16 * dt + 4 * (li & 3);
#endif
                    union { struct { s16x4 a, b; } s; f16x8 v; } uh, ul;
#if CBAS_ATTN_ABLATE & 2
                    o[dt][0] += (float)ph[s2][0] + (float)pl[s2][1];               // keeps P alive; no reads, no MFMAs
                    (void)krow; (void)col; (void)uh; (void)ul;
#else
#if CBAS_ATTN_ABLATE & 4
                    uh.v = qh[s2]; ul.v = ql[s2];                                  // constant fragments instead of V's LDS reads
                    (void)krow; (void)col;
#else
                    uh.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(Vh + sv_off(krow, col)));
                    uh.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(Vh + sv_off(krow + 16, col)));
                    ul.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(Vl + sv_off(krow, col)));
                    ul.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(Vl + sv_off(krow + 16, col)));
#endif
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ul.v, ph[s2], o[dt], 0, 0, 0);
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(uh.v, pl[s2], o[dt], 0, 0, 0);
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(uh.v, ph[s2], o[dt], 0, 0, 0);
#endif
                }
            }
            ATTN_PRIO_OFF(2);
        };
        for (int kb = 0; kb < nkb; ++kb) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's pieces of block kb
            __builtin_amdgcn_s_barrier();                                 // everyone's; and the other buffer is free
            ATTN_PRIO_ON(8);
            if (kb + 1 < nkb) stage((gb + kb + 1) & 1, pair, kb + 1);
            else if (following < total_items) stage((gb + kb + 1) & 1, item_pair(following), 0);      // the next item's first block
            ATTN_PRIO_OFF(8);
            if (active) {
                // a full block (64 real keys: every block but the last) runs without the per-tile tests, so that its four key
                // tiles are four independent MFMA chains in one basic block; same operations per query either way
                if (T - kb * AKB >= AKB) block(kb, std::integral_constant<bool, true>{});
                else block(kb, std::integral_constant<bool, false>{});
            }
        }
        if (active) {
            lrun = xor16_add(lrun);
            lrun = xor32_add(lrun);
            if (q < nq) {
                float* row = out + (q_cls ? (size_t)b : (size_t)b * T + q) * D;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const f32x4 v = (o[dt] * (1.0f / (ATT_VS * ATT_PS))) / lrun;
#if CBAS_ATTN_ABLATE & 32
                    if (v[0] == 123.456f) store_split4(row, hd * 64 + 16 * dt + 4 * g, v, out_scale);      // never true: no stores
#else
                    store_split4(row, hd * 64 + 16 * dt + 4 * g, v, out_scale);
#endif
                }
            }
        }                                                                  // active
        gb += nkb;
        item = following;                                                  // (no barrier here: the next item's first block waits + barriers as every block does)
    }                                                                  // items
}


}  // namespace

#define CHECK_LAUNCH() (hipGetLastError() == hipSuccess ? 0 : -2)

void vit32_split_set_forms(int forms) { g_split_forms = forms; }

int launch_gemm_f32_vit(GemmEpilogue epi, const Gemm32VitParams& p, hipStream_t stream) {
    if (p.M <= 0 || p.N <= 0 || p.N % BN || p.K % BKF || p.lda % 4 || p.ldo % 4) return -1;
    if (epi == EPI_QKV && (p.D % 64 || p.N % p.D || p.sec0 < 0 || p.N / p.D + p.sec0 > 3)) return -1;
    if (p.split && !(p.a_scale > 0.f && p.w_scale > 0.f)) return -1;
    switch (epi) {
        case EPI_PATCH: return launch_vit32<EPI_PATCH>(p, stream);
        case EPI_QKV:   return launch_vit32<EPI_QKV>(p, stream);
        case EPI_RESID: return launch_vit32<EPI_RESID>(p, stream);
        case EPI_GELU:  return launch_vit32<EPI_GELU>(p, stream);
        default: return -1;
    }
}

int launch_im2col_u8_f32(const uint8_t* frames, int n, int height, int width, int64_t frame_stride, int64_t row_stride,
                         int64_t pixel_stride, float* A, float* x, const float* prefix_tokens, int n_prefix, int D, int T,
                         int ps, int split, hipStream_t stream) {
    const int nh = height / ps, nw = width / ps;
    const int64_t total = (int64_t)n * nh * ps * nw;
    hipLaunchKernelGGL(im2col_f32out_kernel<uint8_t>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, frames, n,
                       frame_stride, row_stride, pixel_stride, A, nh, nw, ps, split);
    const int64_t tp = (int64_t)n * n_prefix * (D / 4);
    hipLaunchKernelGGL(write_prefix32_kernel, dim3((unsigned)((tp + 255) / 256)), dim3(256), 0, stream, x, prefix_tokens, n,
                       n_prefix, D, T);
    return CHECK_LAUNCH();
}

int launch_im2col_f32_f32(const float* frames, int n, int height, int width, float* A, float* x,
                          const float* prefix_tokens, int n_prefix, int D, int T, int ps, int split, hipStream_t stream) {
    const int nh = height / ps, nw = width / ps;
    const int64_t total = (int64_t)n * nh * ps * nw;
    hipLaunchKernelGGL(im2col_f32out_kernel<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, frames, n,
                       (int64_t)height * width, (int64_t)width, (int64_t)1, A, nh, nw, ps, split);
    const int64_t tp = (int64_t)n * n_prefix * (D / 4);
    hipLaunchKernelGGL(write_prefix32_kernel, dim3((unsigned)((tp + 255) / 256)), dim3(256), 0, stream, x, prefix_tokens, n,
                       n_prefix, D, T);
    return CHECK_LAUNCH();
}

int launch_pack_split_weight(const float* w, float* out, int64_t N, int K, float scale, hipStream_t stream) {
    if (K % 32 || N <= 0) return -1;
    const int64_t n4 = N * (K / 4);
    hipLaunchKernelGGL(pack_split_weight_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, w, out, N, K, scale);
    return CHECK_LAUNCH();
}

int launch_pack_patch_weight_f32(const float* w, float* out, int D, int ps, hipStream_t stream) {
    hipLaunchKernelGGL(pack_patch_weight_f32_kernel, dim3((D * 256 + 255) / 256), dim3(256), 0, stream, w, out, D, ps);
    return CHECK_LAUNCH();
}

int launch_layernorm_f32(const float* x, int64_t ldx, const float* gamma, const float* beta, float* out, int M, int D,
                         float eps, int split, hipStream_t stream) {
    if (split && D % 32) return -1;
    const int nv = (D / 4 + 63) / 64;
    const dim3 grid((M + 3) / 4), block(256);
    switch (nv) {
        case 1: hipLaunchKernelGGL(layernorm_f32_kernel<1>, grid, block, 0, stream, x, ldx, gamma, beta, out, M, D, eps, split); break;
        case 2: hipLaunchKernelGGL(layernorm_f32_kernel<2>, grid, block, 0, stream, x, ldx, gamma, beta, out, M, D, eps, split); break;
        case 3: hipLaunchKernelGGL(layernorm_f32_kernel<3>, grid, block, 0, stream, x, ldx, gamma, beta, out, M, D, eps, split); break;
        case 4: hipLaunchKernelGGL(layernorm_f32_kernel<4>, grid, block, 0, stream, x, ldx, gamma, beta, out, M, D, eps, split); break;
        default: return -1;
    }
    return CHECK_LAUNCH();
}

int launch_attention_f32(const float* qkv, const float* q_cls, float* out, int n, int T, int D, int n_heads,
                         float split_scale, hipStream_t stream) {
    if (n <= 0 || T <= 0 || D != n_heads * 64) return -1;
    static bool attr_set = false;
    constexpr int lds = 4 * AIMG;                           // two buffers x (K + V) = 64 KiB: two workgroups per CU
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_f32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                lds) != hipSuccess)
            return -2;
        attr_set = true;
    }
    // waves per workgroup = query tiles per workgroup: the count in 5..8 that wastes the fewest tile slots (T = 201:
    // 13 tiles -> 7 waves x 2 workgroups; T = 261: 17 -> 6 x 3; T = 1029: 65 -> 5 x 13), larger on ties
    const int ntiles = q_cls ? 1 : (T + 15) / 16;
    int nw = 8, best = 1 << 30;
    for (int w = 8; w >= 5; --w) {
        const int waste = (ntiles + w - 1) / w * w - ntiles;
        if (waste < best) { best = waste; nw = w; }
    }
    if (q_cls) nw = 4;
#if CBAS_ATTN_WIDE
    const bool wide = split_scale > 0.f && !q_cls && ntiles <= 16;
    if (wide) nw = ntiles;
#else
    const bool wide = false;
#endif
    const int qblocks = (ntiles + nw - 1) / nw;
    const int npairs = n * n_heads;
    const int64_t grid = (int64_t)((npairs + 7) / 8) * 8 * qblocks;        // whole groups of 8 pairs: see the kernels' workgroup map
    if (grid > 0x7fffffff) return -1;
    if (split_scale > 0.f) {                                // precision 4: q | k | v arrive split (store_head_split4)
        static bool attr2 = false;
        constexpr int lds2 = 2 * 4 * SIMG;
        if (!attr2) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_split_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    lds2) != hipSuccess)
                return -2;
            attr2 = true;
        }
        // persistent launch (default; CBAS_ATTN_PERSIST=0: one workgroup per item): at most two workgroups per CU - what fits -
        // each walking several items with its K / V ring running on across them; a multiple of 8 keeps the XCD map.  Rows are
        // bit-identical either way; 59.5 -> 56.0 us per layer at ViT-B batch 64 on one lease (profiles/r05_attention_ablation.json)
        static const int persist = [] { const char* e = getenv("CBAS_ATTN_PERSIST"); return e ? atoi(e) : 1; }();
        int64_t launch = grid;
        if (persist) {
            static int cus = 0;
            if (!cus) { int dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev); }
            const int64_t cap = (int64_t)((wide ? 1 : 2) * cus) / 8 * 8;
            if (cap > 0 && launch > cap) launch = cap;
        }
        hipLaunchKernelGGL(attention_split_kernel, dim3((unsigned)launch), dim3(nw * 64), lds2, stream, qkv, q_cls, out, T, D, n_heads,
                           qblocks, split_scale, npairs, (int)grid);
        return CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(attention_f32_kernel, dim3((unsigned)grid), dim3(nw * 64), lds, stream, qkv, q_cls, out, T, D, n_heads, qblocks, split_scale,
                       npairs);
    return CHECK_LAUNCH();
}

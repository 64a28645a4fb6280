// fp16 MFMA GEMM, deep-prefetch variant (gfx950):  C[M,N] = A[M,K] * W[N,K]^T + fused epilogue.
//
// Why it exists: the 2-buffer kernel of gemm_f16.hip waits for the NEXT K-tile's LDS-DMA at the end
// of every K-step (prefetch distance 1, vmcnt(0) + barrier).  Under load that DMA takes a few
// thousand cycles, so its waves sit in s_waitcnt ~50 % of the time and the matrix pipe is ~26 %
// busy (rocprofv3 SQ_WAIT_ANY / SQ_VALU_MFMA_BUSY_CYCLES, profiles/).  Here:
//   * 256x256 workgroup tile (128 FLOP per byte staged into LDS, twice the 128x128 kernel),
//   * BK = 32 so that FOUR K-tiles fit the LDS (4 x 32 KiB ring): the DMA for K-tile kt+3 is issued
//     while K-tile kt is being multiplied,
//   * a counted s_waitcnt vmcnt(N) that only retires K-tile kt+1 and a raw s_barrier, so two tiles
//     stay in flight across every barrier (never vmcnt(0) in the loop),
//   * 64-byte LDS rows with the XOR swizzle chunk ^ (3 * ((row >> 2) & 1)) on the DMA source address
//     and on the ds_read_b128 address (conflict-free for the 16-row x 4-chunk fragment read).
// K order per output element is identical to gemm_f16.hip, so results are bit-identical.
#include <stdlib.h>
#include "gemm_epilogue.h"

namespace {

constexpr int RBK = 32;          // halves of K per ring slot

__device__ __forceinline__ int swz64(int row, int chunk) { return chunk ^ (((row >> 2) & 1) * 3); }

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// WM x WN waves; each wave owns (TM*16) rows x 64 columns; RING LDS slots, prefetch distance RING-1.
template <int EPI, int WM, int WN, int TM, int RING, int WPE>
__global__ __launch_bounds__(WM * WN * 64, WPE) void gemm_f16_ring_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = WM * WN;
    constexpr int BM = WM * TM * 16, BN = WN * 64;
    constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64, SLOT_BYTES = A_BYTES + B_BYTES;
    constexpr int A_PIECES = BM / 16, B_PIECES = BN / 16;
    constexpr int PER = (A_PIECES + B_PIECES) / NW;           // LDS-DMA instructions per wave per K-tile
    static_assert(PER * NW == A_PIECES + B_PIECES, "pieces must split evenly over the waves");
    static_assert(RING * SLOT_BYTES <= 160 * 1024, "ring does not fit the LDS");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WN, wc = wave % WN;

    const int tiles_n = p.N / BN;
    const int bid = gemm_xcd_remap(blockIdx.x, gridDim.x);
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int row0 = tm * BM, col0 = tn * BN;
    const int nk = p.K / RBK;

    f32x4 acc[TM][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // this lane's part of the staging addresses (constant over K except for the k offset)
    const f16* src[PER];
    int dst[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int q = wave * PER + i;                        // piece id: [0, A_PIECES) are A, the rest W
        const bool isA = q < A_PIECES;
        const int piece = isA ? q : q - A_PIECES;
        const int r = piece * 16 + (lane >> 2);
        const int chunk = swz64(r, lane & 3);
        int grow = (isA ? row0 : col0) + r;
        const int last = isA ? p.M_pad - 1 : p.N - 1;
        grow = grow < last ? grow : last;
        src[i] = (isA ? p.A : p.W) + (size_t)grow * p.K + chunk * 8;
        dst[i] = (isA ? 0 : A_BYTES) + piece * 1024;
    }
    constexpr int DIST = RING - 1;
    auto stage = [&](int kt) {
        char* base = smem + (kt % RING) * SLOT_BYTES;
#pragma unroll
        for (int i = 0; i < PER; ++i)
            __builtin_amdgcn_global_load_lds(GLB_PTR(src[i] + kt * RBK), LDS_PTR(base + dst[i]), 16, 0, 0);
    };

    // prologue: DIST K-tiles in flight, retire the first
    auto wait_leave = [&](int tiles_in_flight) {             // wave-uniform; counts must be immediates
        if (tiles_in_flight >= 2 && DIST >= 3) wait_vmcnt<2 * PER>();
        else if (tiles_in_flight >= 1 && DIST >= 2) wait_vmcnt<PER>();
        else wait_vmcnt<0>();
    };
#pragma unroll
    for (int t = 0; t < DIST; ++t)
        if (t < nk) stage(t);
    wait_leave((nk < DIST ? nk : DIST) - 1);
    __builtin_amdgcn_s_barrier();

    const int frow = lane & 15, fchunk = lane >> 4;
    int a_off[TM], b_off[4];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = wr * TM * 16 + i * 16 + frow;
        a_off[i] = r * 64 + (swz64(r, fchunk) << 4);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = wc * 64 + j * 16 + frow;
        b_off[j] = A_BYTES + r * 64 + (swz64(r, fchunk) << 4);
    }

    for (int kt = 0; kt < nk; ++kt) {
        if (kt + DIST < nk) stage(kt + DIST);
        const char* slot = smem + (kt % RING) * SLOT_BYTES;
        f16x8 a[TM], b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const f16x8*>(slot + b_off[j]);
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f16x8*>(slot + a_off[i]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j], a[i], acc[i][j], 0, 0, 0);
        // retire K-tile kt+1 (read in the NEXT step, one barrier after this wait); leave the rest in flight
        {
            const int last = kt + DIST < nk - 1 ? kt + DIST : nk - 1;   // newest K-tile staged so far
            wait_leave(last - (kt + 1));
        }
        __builtin_amdgcn_s_barrier();
    }

    gemm_epilogue_tile<EPI, TM>(p, row0 + wr * TM * 16, col0 + wc * 64, lane, acc, smem + wave * 8192);
}

template <int EPI, int WM, int WN, int TM, int RING, int WPE>
int launch_ring(const GemmParams& p, hipStream_t stream) {
    constexpr int BM = WM * TM * 16, BN = WN * 64;
    constexpr int ring_bytes = RING * (BM + BN) * 64, scratch_bytes = WM * WN * 8192;   // epilogue: 8 KiB per wave
    constexpr int lds = ring_bytes > scratch_bytes ? ring_bytes : scratch_bytes;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16_ring_kernel<EPI, WM, WN, TM, RING, WPE>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return -2;
        attr_set = true;
    }
    if (p.N % BN || p.K % RBK || p.W_lo) return -1;
    const int grid = ((p.M + BM - 1) / BM) * (p.N / BN);
    hipLaunchKernelGGL((gemm_f16_ring_kernel<EPI, WM, WN, TM, RING, WPE>), dim3(grid), dim3(WM * WN * 64), lds, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

template <int EPI>
int launch_ring_epi(const GemmParams& p, int tile, hipStream_t stream) {
    switch (tile) {
        case GEMM_TILE_RING_256x256_W16: return launch_ring<EPI, 4, 4, 4, 4, 4>(p, stream);
        case GEMM_TILE_RING_256x256_W8:  return launch_ring<EPI, 2, 4, 8, 4, 2>(p, stream);
        case 8:  return launch_ring<EPI, 2, 4, 4, 2, 4>(p, stream);    // 128x256, 2 slots (48 KB): 2-3 WGs/CU
        case 9:  return launch_ring<EPI, 2, 4, 4, 3, 4>(p, stream);    // 128x256, 3 slots (72 KB): 2 WGs/CU
        case 10: return launch_ring<EPI, 4, 2, 4, 3, 4>(p, stream);    // 256x128, 3 slots (72 KB): 2 WGs/CU
        case 11: return launch_ring<EPI, 2, 2, 4, 4, 4>(p, stream);    // 128x128, 4 slots (64 KB): 2 WGs/CU, 4 waves
        default: return -1;
    }
}

}  // namespace

int launch_gemm_ring(GemmEpilogue epi, const GemmParams& p, int tile, hipStream_t stream) {
    switch (epi) {
        case EPI_PATCH: return launch_ring_epi<EPI_PATCH>(p, tile, stream);
        case EPI_QKV:   return launch_ring_epi<EPI_QKV>(p, tile, stream);
        case EPI_RESID: return launch_ring_epi<EPI_RESID>(p, tile, stream);
        case EPI_GELU:  return launch_ring_epi<EPI_GELU>(p, tile, stream);
    }
    return -1;
}

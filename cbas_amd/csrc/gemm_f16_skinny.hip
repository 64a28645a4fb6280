// fp16 MFMA GEMM for M <= 64 rows: the CLS rows of the pruned last layer (q, o_proj, up, down on n frames).
// Same arithmetic, fragment layout, swizzle and fused epilogues as gemm_f16.hip / gemm_f16_8ph.hip (bit-identical:
// every accumulator sees the same MFMAs in the same k order); what differs is the shape of the problem.  With 64
// rows there are only N/64 tiles of work (12 for N = 768), far too few to hide anything behind other workgroups,
// and the 2-buffer kernels pay one global->LDS round trip per K-tile (K = 3072: 48 of them, 30 us).  Here a
// workgroup is 4 waves x (16 rows x 64 columns), the K loop runs over an 8-deep LDS ring filled by 16-byte LDS-DMA
// with 6 K-tiles in flight under a counted vmcnt and ONE raw barrier per K-tile, so the round-trip latency is paid
// once per tile, not once per K-tile.
//
// Reference arithmetic replaced: the nn.Linear calls of the last DINOv3ViTLayer on the CLS row ([tf]:307-309, :331,
// :356-357; only last_hidden_state[:, 0] is consumed: [tf]:540-541, backend/cbas.py:677).
#include "gemm_epilogue.h"

namespace {

constexpr int SK_BK = 64, SK_NS = 8;                   // K-tile (elements), ring slots
constexpr int SK_STAGE = 2 * 64 * 128;                 // A 64 rows + W 64 rows, 128 bytes each

template <int EPI>
__global__ __launch_bounds__(256) void gemm_f16_skinny_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col0 = blockIdx.x * 64;
    const int nk = p.K / SK_BK;

    // staging: piece q = wave + 4*s (s = 0,1) of each 8-piece operand image; 128-byte rows, 16-byte chunks swizzled on
    // the source address (chunk c of row r sits at position c ^ ((r >> 1) & 7)), exactly as the other GEMM kernels
    const int lrow = lane >> 3;
    unsigned a_src[2], b_src[2];
    int lds_off[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int r = (wave + 4 * s) * 8 + lrow;
        const int sw = ((lane & 7) ^ ((r >> 1) & 7)) << 3;
        const int ar = r < p.M_pad ? r : p.M_pad - 1;
        a_src[s] = (unsigned)ar * (unsigned)p.lda + sw;
        b_src[s] = (unsigned)(col0 + r) * (unsigned)p.K + sw;
        lds_off[s] = (wave + 4 * s) * 1024;
    }
    auto stage = [&](int kt) {
        char* base = smem + (kt % SK_NS) * SK_STAGE;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            __builtin_amdgcn_global_load_lds(GLB_PTR(p.A + a_src[s] + kt * SK_BK), LDS_PTR(base + lds_off[s]), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(GLB_PTR(p.W + b_src[s] + kt * SK_BK), LDS_PTR(base + 8192 + lds_off[s]), 16, 0, 0);
        }
    };
    const int frow = lane & 15, fchunk = lane >> 4;
    int a_off[2], b_off[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int sw = ((kk * 4 + fchunk) ^ ((frow >> 1) & 7)) << 4;
        a_off[kk] = (wave * 16 + frow) * 128 + sw;
        b_off[kk] = 8192 + frow * 128 + sw;
    }
    f32x4 acc[1][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[0][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int kt = 0; kt < SK_NS - 1 && kt < nk; ++kt) stage(kt);            // 7 K-tiles in flight
    for (int kt = 0; kt < nk; ++kt) {
        // K-tile kt has landed once at most the 6 younger stages (4 DMAs each per wave) are outstanding
        if (kt + SK_NS - 2 < nk) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                       // ... for every wave's pieces; and every wave
        if (kt + SK_NS - 1 < nk) stage(kt + SK_NS - 1);                     // is done with slot (kt-1) % NS, refilled here
        const char* buf = smem + (kt % SK_NS) * SK_STAGE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const f16x8 a = *reinterpret_cast<const f16x8*>(buf + a_off[kk]);
            f16x8 b[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const f16x8*>(buf + b_off[kk] + j * 2048);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j], a, acc[0][j], 0, 0, 0);
        }
    }
    __syncthreads();                                                        // the ring becomes epilogue scratch
    gemm_epilogue_tile<EPI, 1>(p, wave * 16, col0, lane, acc, smem + wave * 8192);
}

template <int EPI>
int launch_skinny(const GemmParams& p, hipStream_t stream) {
    constexpr int lds = SK_NS * SK_STAGE;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16_skinny_kernel<EPI>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return -2;
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_f16_skinny_kernel<EPI>), dim3(p.N / 64), dim3(256), lds, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace

int launch_gemm_skinny(GemmEpilogue epi, const GemmParams& p, hipStream_t stream) {
    if (p.M > 64 || p.M <= 0 || p.W_lo || p.A8 || p.N % 64 || p.K % SK_BK || (long long)p.M_pad * p.lda >= (1ll << 31) ||
        (long long)p.N * p.K >= (1ll << 31))
        return -1;
    switch (epi) {
        case EPI_QKV:   return launch_skinny<EPI_QKV>(p, stream);
        case EPI_RESID: return launch_skinny<EPI_RESID>(p, stream);
        case EPI_GELU:  return launch_skinny<EPI_GELU>(p, stream);
        default: return -1;
    }
}

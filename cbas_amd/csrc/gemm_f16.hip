// fp16 MFMA GEMM for the ViT projections on gfx950 (MI355X).
//
//   C[M,N] = A[M,K] * W[N,K]^T,  A/W fp16 row-major (K contiguous), fp32 accumulate,
//   epilogue fused per GemmEpilogue (bias, RoPE, LayerScale+residual, exact GELU).
//
// Structure (v1): 128x128x64 tile per 256-thread workgroup (2x2 waves, each 64x64 = 4x4 MFMA
// 16x16x32 tiles), both operands staged global->LDS with 16-byte LDS-DMA (global_load_lds), two
// LDS buffers (next K-tile's DMA is in flight under the current tile's MFMAs), XOR-swizzled 128-B
// LDS rows so every ds_read_b128 fragment read is bank-conflict-free, XCD-aware tile order.
//
// The MFMAs are issued with the operands swapped (W fragment as "A", activation fragment as "B"),
// so the accumulator holds C^T tiles: lane = output row m, 4 registers = 4 consecutive output
// columns n.  That makes every epilogue store an 8/16-byte vector store and puts the RoPE partner
// (column d +- 32 of the same head) in the same lane.
//
// Reference arithmetic replaced: nn.Linear calls at [tf] modeling_dinov3_vit.py:307-309 (q/k/v),
// :331 (o_proj), :356-357 (MLP), conv patch embedding :82, RoPE :238-268, LayerScale :342-343,
// residual adds :432-443.
#include "kernels.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;      // 16 KiB per operand tile

__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
    // blocks b and b+8 share an XCD (round-robin dispatch); give each XCD a contiguous range of
    // logical tile ids so neighbouring tiles (same A row-panel) hit one L2.  Bijective for any nwg.
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (orig >> 3);
}

// Stage a 128-row x 64-half tile: 16 pieces of 1 KiB, 4 per wave.  LDS image: row r at r*128 B,
// 16-B slot s of row r holds global chunk s ^ ((r>>1)&7)  (swizzle applied on the SOURCE address;
// the LDS-DMA destination is lane-linear).
__device__ __forceinline__ void stage_tile(const f16* __restrict__ g, int ld, int row0, int k0,
                                           char* lds_tile, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int piece = wave * 4 + i;
        const int r = piece * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        const f16* src = g + (size_t)(row0 + r) * ld + k0 + chunk * 8;
        __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(lds_tile + piece * 1024), 16, 0, 0);
    }
}

__device__ __forceinline__ f16x8 read_frag(const char* lds_tile, int row, int chunk) {
    const int off = row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
    return *reinterpret_cast<const f16x8*>(lds_tile + off);
}

template <int EPI, int NSPLIT>
__global__ __launch_bounds__(256, 2) void gemm_f16_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BUF_BYTES = TILE_BYTES * (1 + NSPLIT);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    const int tiles_n = p.N / BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int row0 = tm * BM, col0 = tn * BN;
    const int nk = p.K / BK;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * BUF_BYTES;
        stage_tile(p.A, p.K, row0, kt * BK, base, wave, lane);
        stage_tile(p.W, p.K, col0, kt * BK, base + TILE_BYTES, wave, lane);
        if (NSPLIT == 2) stage_tile(p.W_lo, p.K, col0, kt * BK, base + 2 * TILE_BYTES, wave, lane);
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
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            f16x8 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = read_frag(At, wr * 64 + i * 16 + frow, kk * 4 + fchunk);
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = read_frag(Wt, wc * 64 + j * 16 + frow, kk * 4 + fchunk);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j], a[i], acc[i][j], 0, 0, 0);
            if (NSPLIT == 2) {
                const char* Wl = Wt + TILE_BYTES;
#pragma unroll
                for (int j = 0; j < 4; ++j) b[j] = read_frag(Wl, wc * 64 + j * 16 + frow, kk * 4 + fchunk);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j], a[i], acc[i][j], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- epilogue: lane owns row m (per i) and columns n0..n0+3 (per j) -------------------------
    const int ncol = col0 + wc * 64 + (lane >> 4) * 4;     // + j*16
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = row0 + wr * 64 + i * 16 + (lane & 15);
        if (m >= p.M) continue;
        if (EPI == EPI_PATCH) {
            const int b = m / p.patches_per_frame;
            const int orow = b * p.tokens_per_frame + p.n_prefix + (m - b * p.patches_per_frame);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = ncol + j * 16;
                const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + n);
                f32x4 v = acc[i][j] * p.in_scale + bv;
                *reinterpret_cast<f32x4*>(p.out_f32 + (size_t)orow * p.ldo + n) = v;
            }
        } else if (EPI == EPI_QKV) {
            const int sec = (col0 + wc * 64) / p.D;        // 0 q, 1 k, 2 v: wave-uniform (64 | D)
            const int t = m % p.tokens_per_frame;
            const bool rope = (sec < 2) && (t >= p.n_prefix);
            f32x4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                v[j] = acc[i][j] + *reinterpret_cast<const f32x4*>(p.bias + ncol + j * 16);
            if (rope) {
                const size_t ro = (size_t)(t - p.n_prefix) * 64 + (lane >> 4) * 4;
                f32x4 o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 c = *reinterpret_cast<const f32x4*>(p.rope_cos + ro + j * 16);
                    const f32x4 s = *reinterpret_cast<const f32x4*>(p.rope_sin + ro + j * 16);
                    // rotate_half(x)[d] = -x[d+32] (d < 32), x[d-32] (d >= 32)
                    o[j] = (j < 2) ? (v[j] * c - v[j + 2] * s) : (v[j] * c + v[j - 2] * s);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = o[j];
            }
            const float qs = (sec == 0) ? 0.125f : 1.0f;   // head_dim^-0.5, exact power of two
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 w = v[j] * qs;
                f16x4 hv = {(f16)w[0], (f16)w[1], (f16)w[2], (f16)w[3]};
                *reinterpret_cast<f16x4*>(p.out_f16 + (size_t)m * p.ldo + ncol + j * 16) = hv;
            }
        } else if (EPI == EPI_RESID) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = ncol + j * 16;
                const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + n);
                const f32x4 lv = *reinterpret_cast<const f32x4*>(p.lambda + n);
                float* xp = p.out_f32 + (size_t)m * p.ldo + n;
                const f32x4 xv = *reinterpret_cast<const f32x4*>(xp);
                *reinterpret_cast<f32x4*>(xp) = (acc[i][j] + bv) * lv + xv;
            }
        } else {  // EPI_GELU
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = ncol + j * 16;
                const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + n);
                const f32x4 w = acc[i][j] + bv;
                f16x4 hv = {(f16)gelu_erf(w[0]), (f16)gelu_erf(w[1]), (f16)gelu_erf(w[2]), (f16)gelu_erf(w[3])};
                *reinterpret_cast<f16x4*>(p.out_f16 + (size_t)m * p.ldo + n) = hv;
            }
        }
    }
}

template <int EPI, int NSPLIT>
int launch_one(const GemmParams& p, hipStream_t stream) {
    constexpr int lds = 2 * TILE_BYTES * (1 + NSPLIT);
    static bool attr_set = false;   // per-instantiation; idempotent, racing callers set the same value
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16_kernel<EPI, NSPLIT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return -2;
        attr_set = true;
    }
    const int grid = (p.M_pad / BM) * (p.N / BN);
    hipLaunchKernelGGL((gemm_f16_kernel<EPI, NSPLIT>), dim3(grid), dim3(256), lds, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace

int launch_gemm(GemmEpilogue epi, const GemmParams& p, hipStream_t stream) {
    if (p.M_pad % BM || p.N % BN || p.K % BK || p.M > p.M_pad || p.M <= 0) return -1;
    if (epi == EPI_QKV && (p.D % 64 || p.N != 3 * p.D)) return -1;
    const bool split = p.W_lo != nullptr;
    switch (epi) {
        case EPI_PATCH: return split ? launch_one<EPI_PATCH, 2>(p, stream) : launch_one<EPI_PATCH, 1>(p, stream);
        case EPI_QKV:   return split ? launch_one<EPI_QKV, 2>(p, stream)   : launch_one<EPI_QKV, 1>(p, stream);
        case EPI_RESID: return split ? launch_one<EPI_RESID, 2>(p, stream) : launch_one<EPI_RESID, 1>(p, stream);
        case EPI_GELU:  return split ? launch_one<EPI_GELU, 2>(p, stream)  : launch_one<EPI_GELU, 1>(p, stream);
    }
    return -1;
}

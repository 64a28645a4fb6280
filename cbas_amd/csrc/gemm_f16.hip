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
#include <stdlib.h>
#include "gemm_epilogue.h"

namespace {

constexpr int BK = 64;

// Stage a ROWS-row x 64-half tile as ROWS/8 pieces of 1 KiB, ROWS/8/NW per wave.  LDS image: row r
// at r*128 B, 16-B slot s of row r holds global chunk s ^ ((r>>1)&7)  (swizzle applied on the
// SOURCE address; the LDS-DMA destination is lane-linear).  Rows past `last_row` are clamped (their
// products land in output rows/columns that are never stored).
template <int ROWS, int NW>
__device__ __forceinline__ void stage_tile(const f16* __restrict__ g, int ld, int row0, int last_row, int k0,
                                           char* lds_tile, int wave, int lane) {
    constexpr int PIECES = ROWS / 8, PER = (PIECES + NW - 1) / NW;     // uneven split allowed (12 waves)
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int piece = wave + i * NW;
        if (PIECES % NW != 0 && piece >= PIECES) break;                // wave-uniform
        const int r = piece * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        int row = row0 + r;
        row = row < last_row ? row : last_row;
        const f16* src = g + (size_t)row * ld + k0 + chunk * 8;
        __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(lds_tile + piece * 1024), 16, 0, 0);
    }
}

__device__ __forceinline__ f16x8 read_frag(const char* lds_tile, int row, int chunk) {
    const int off = row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
    return *reinterpret_cast<const f16x8*>(lds_tile + off);
}

// Workgroup tile = (64*WM) x (64*WN), one 64x64 sub-tile per wave.  The larger tiles exist because
// at 128x128 the kernel stages 1 byte per 64 FLOP from L2 / Infinity Cache into LDS, which caps it
// near the measured L2->LDS rate; 256x256 halves that traffic.
template <int EPI, int NSPLIT, int WM, int WN, int TM = 4>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN >= 16 ? 4 : 2)) void gemm_f16_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 16 * TM * WM, BN = 64 * WN, NW = WM * WN;      // each wave owns (16*TM) rows x 64 columns
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
    constexpr int BUF_BYTES = A_BYTES + B_BYTES * NSPLIT;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WN, wc = wave % WN;

    const int tiles_n = p.N / BN;
    const int bid = gemm_xcd_remap(blockIdx.x, gridDim.x);
    int tm, tn;
    if (p.group_m <= 1) {
        tm = bid / tiles_n; tn = bid - tm * tiles_n;
    } else {
        // grouped raster: GM row-panels share each W column tile while it is hot in the XCD's L2
        const int tiles_m = gridDim.x / tiles_n;
        const int gsz = p.group_m * tiles_n, gid = bid / gsz, first = gid * p.group_m;
        const int gm = tiles_m - first < p.group_m ? tiles_m - first : p.group_m;
        const int in = bid - gid * gsz;
        tm = first + in % gm; tn = in / gm;
    }
    const int row0 = tm * BM, col0 = tn * BN;
    const int nk = p.K / BK;

    f32x4 acc[TM][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * BUF_BYTES;
        stage_tile<BM, NW>(p.A, p.lda, row0, p.M_pad - 1, kt * BK, base, wave, lane);
        stage_tile<BN, NW>(p.W, p.K, col0, p.N - 1, kt * BK, base + A_BYTES, wave, lane);
        if (NSPLIT == 2) stage_tile<BN, NW>(p.W_lo, p.K, col0, p.N - 1, kt * BK, base + A_BYTES + B_BYTES, wave, lane);
    };

    unsigned long long t_start = 0, t_pro = 0, t_loop = 0;
    if (p.stamps) t_start = __builtin_amdgcn_s_memtime();
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (p.stamps) t_pro = __builtin_amdgcn_s_memtime();

    const int frow = lane & 15, fchunk = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const char* At = smem + cur * BUF_BYTES;
        const char* Wt = At + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            f16x8 a[TM], b[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = read_frag(Wt, wc * 64 + j * 16 + frow, kk * 4 + fchunk);
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = read_frag(At, wr * TM * 16 + i * 16 + frow, kk * 4 + fchunk);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j], a[i], acc[i][j], 0, 0, 0);
            if (NSPLIT == 2) {
                const char* Wl = Wt + B_BYTES;
#pragma unroll
                for (int j = 0; j < 4; ++j) b[j] = read_frag(Wl, wc * 64 + j * 16 + frow, kk * 4 + fchunk);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j], a[i], acc[i][j], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    if (p.stamps) t_loop = __builtin_amdgcn_s_memtime();
    // ---- epilogue: lane owns row m (per i) and 4 consecutive columns in each of the 4 n-tiles ---
    gemm_epilogue_tile<EPI, TM>(p, row0 + wr * TM * 16, col0 + wc * 64, lane, acc, smem + wave * 8192);
    if (p.stamps && tid == 0) {       // diagnostic builds only: block timeline (start, prologue, loop, end)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long* o = p.stamps + (size_t)blockIdx.x * 4;
        o[0] = t_start; o[1] = t_pro; o[2] = t_loop; o[3] = __builtin_amdgcn_s_memtime();
    }
}

template <int EPI, int NSPLIT, int WM, int WN, int TM = 4>
int launch_one(const GemmParams& p, hipStream_t stream) {
    constexpr int BM = 16 * TM * WM, BN = 64 * WN;
    constexpr int lds = 2 * (BM * 128 + BN * 128 * NSPLIT);
    static_assert(lds <= 160 * 1024, "tile does not fit the 160 KiB LDS");
    static bool attr_set = false;   // per-instantiation; idempotent, racing callers set the same value
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16_kernel<EPI, NSPLIT, WM, WN, TM>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return -2;
        attr_set = true;
    }
    if (p.N % BN) return -1;
    const int grid = ((p.M + BM - 1) / BM) * (p.N / BN);
    hipLaunchKernelGGL((gemm_f16_kernel<EPI, NSPLIT, WM, WN, TM>), dim3(grid), dim3(WM * WN * 64), lds, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

template <int EPI>
int launch_epi(const GemmParams& p, int tile, hipStream_t stream) {
    const bool split = p.W_lo != nullptr;
    if (split) {      // the hi+lo weight tiles need 50 % more LDS: 128x128 and 256x128 only
        if (tile == GEMM_TILE_256x128 || tile == GEMM_TILE_256x256) return launch_one<EPI, 2, 4, 2>(p, stream);
        return launch_one<EPI, 2, 2, 2>(p, stream);
    }
    switch (tile) {
        case GEMM_TILE_256x256: return launch_one<EPI, 1, 4, 4>(p, stream);
        case GEMM_TILE_256x128: return launch_one<EPI, 1, 4, 2>(p, stream);
        case GEMM_TILE_128x256: return launch_one<EPI, 1, 2, 4>(p, stream);
        case GEMM_TILE_192x256: return launch_one<EPI, 1, 3, 4>(p, stream);
        case 12: return launch_one<EPI, 1, 2, 4, 8>(p, stream);      // 256x256, 8 waves x (128x64): experiment
        case GEMM_TILE_64x128: return launch_one<EPI, 1, 1, 2>(p, stream);   // M <= 64: the CLS rows of the last layer
        default: return launch_one<EPI, 1, 2, 2>(p, stream);
    }
}

// Tile choice: estimated time = rounds over the CU slots x cost of one tile-round.  The larger tiles
// halve the L2->LDS traffic but quantise worse; costs are relative per-round times measured on
// MI355X with scripts/gemm_tiles.py.
int pick_tile(const GemmParams& p) {
    static const int forced_longk = [] { const char* e = getenv("CBAS_GEMM_TILE_LONGK"); return e ? atoi(e) : 0; }();
    if (forced_longk && p.K >= 2048) return forced_longk;          // experiments on the down projection only
    static const int forced = [] { const char* e = getenv("CBAS_GEMM_TILE"); return e ? atoi(e) : 0; }();
    if (forced) return forced;
    if (p.tile) return p.tile;
    static const bool skinny_off = [] { const char* e = getenv("CBAS_GEMM_SKINNY"); return e && atoi(e) == 0; }();
    if (p.M <= 64 && !p.W_lo) return skinny_off ? GEMM_TILE_64x128 : GEMM_TILE_SKINNY;
    // Measured with scripts/gemm_tiles.py at M = 12 864 (64 frames x 201 tokens), ViT-B shapes: the
    // 256x256 tile (16 waves, 1 workgroup per CU, 83 % MFMA-efficient main loop) wins on every
    // projection once there are enough tiles to occupy the chip; below that the 128x128 tile
    // (2 workgroups per CU, 4x the tiles) does.
    if (!p.W_lo && p.N % 256 == 0) {
        const long t256 = (long)((p.M + 255) / 256) * (p.N / 256);
        const long t192 = (long)((p.M + 191) / 192) * (p.N / 256);
        if (t256 >= 120) {
            // one workgroup per CU for both: time ~ rounds over the 256 CUs x tile rows
            const long c256 = ((t256 + 255) / 256) * 256, c192 = ((t192 + 255) / 256) * 192;
            // the 8-wave ping-pong kernel (gemm_f16_8ph.hip) computes the same tiles bit-identically and
            // is 4-8 % faster on every projection; it consumes K-tiles in pairs and uses 32-bit offsets
            static const bool pp_off = [] { const char* e = getenv("CBAS_GEMM_PP"); return e && atoi(e) == 0; }();
            const bool pp = !pp_off && p.K % 128 == 0 && (long long)p.M_pad * p.lda < (1ll << 31) &&
                            (long long)p.N * p.K < (1ll << 31);
            if (pp) return GEMM_TILE_PP_AUTO;          // its planner also knows 160-row tiles and 128-row tails
            return c192 < c256 ? GEMM_TILE_192x256 : GEMM_TILE_256x256;
        }
    }
    return GEMM_TILE_128x128;
}

}  // namespace

static int dispatch_gemm(GemmEpilogue epi, const GemmParams& p, int tile, hipStream_t stream) {
    if (tile >= GEMM_TILE_PP_256x256 && tile <= GEMM_TILE_PP_AUTO) return launch_gemm_8ph(epi, p, tile, stream);
    if (epi_base(epi) != epi) return -3;              // the LayerNorm fold exists in the ping-pong kernel only
    if (tile == GEMM_TILE_SKINNY) return launch_gemm_skinny(epi, p, stream);
    switch (epi) {
        case EPI_PATCH: return launch_epi<EPI_PATCH>(p, tile, stream);
        case EPI_QKV:   return launch_epi<EPI_QKV>(p, tile, stream);
        case EPI_RESID: return launch_epi<EPI_RESID>(p, tile, stream);
        case EPI_GELU:  return launch_epi<EPI_GELU>(p, tile, stream);
    }
    return -1;
}

int launch_gemm(GemmEpilogue epi, const GemmParams& p_in, hipStream_t stream) {
    GemmParams p = p_in;
    static const int gm_env = [] { const char* e = getenv("CBAS_GEMM_GM"); return e ? atoi(e) : 0; }();
    static const int gmw_env = [] { const char* e = getenv("CBAS_GEMM_GM_WIDE"); return e ? atoi(e) : 0; }();      // N >= 2048 only (experiment)
    if (!p.group_m) p.group_m = gm_env > 0 ? gm_env : (gmw_env > 0 && p.N / 256 >= 8 ? gmw_env : 1);
    if (!p.lda) p.lda = p.K;
    if (p.N % 128 || p.K % BK || p.M > p.M_pad || p.M <= 0) return -1;
    if (epi_base(epi) == EPI_QKV && (p.D % 64 || p.N % p.D || p.sec0 < 0 || p.N / p.D + p.sec0 > 3)) return -1;
    if (p.A8) {           // MX-fp8 operands exist only in the ping-pong kernel; small problems take its 128-row tile
        if (p.N % 256) return -1;
        const long t256 = (long)((p.M + 255) / 256) * (p.N / 256);
        if (p.tile >= GEMM_TILE_PP_256x256 && p.tile <= GEMM_TILE_PP_AUTO) return launch_gemm_8ph(epi, p, p.tile, stream);   // bring-up
        return launch_gemm_8ph(epi, p, t256 >= 120 ? GEMM_TILE_PP_AUTO : GEMM_TILE_PP_128x256, stream);
    }
    if (epi == EPI_GELU_F8) return -1;
    const int tile = pick_tile(p);
    int rc = dispatch_gemm(epi, p, tile, stream);
    if (rc == -3) return -1;
    if (rc == -1 && tile != GEMM_TILE_128x128 && epi_base(epi) == epi) rc = dispatch_gemm(epi, p, GEMM_TILE_128x128, stream);   // shape not tileable that way
    return rc;
}

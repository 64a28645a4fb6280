// Exact-fp32 MFMA GEMM for the classifier head on gfx950:  C[M,N] = A[M,K] * W[N,K]^T (+bias, +GELU).
//
// The head is 0.1 % of the encoder's FLOPs but its labels must match the reference's fp32 CPU
// arithmetic, so it runs on v_mfma_f32_16x16x4_f32 (f32 in / f32 accumulate: bit-for-bit an fmaf
// chain, MI355X_MICROARCH.md "Matrix cores") instead of the fp16 path.
//
// Same geometry as gemm_f16.hip: 128x128 tile per 256-thread workgroup, 128-byte LDS rows (here
// 32 floats of K), 16-byte LDS-DMA staging with the XOR swizzle on the source address, two LDS
// buffers.  A lane reads 4 consecutive k with one ds_read_b128 and feeds them to 4 MFMAs; the
// k-slot permutation this implies is the same for both operands, so the sum is unchanged.
//
// Replaces nn.Linear calls of backend/classifier_head.py:72-75 (bottlenecks, via the per-frame
// projection of SURVEY.md §8(a) H1), :83-87 (lin0 + GELU), :96 (lin1) and the input half of
// nn.LSTM :100 (W_ih x + b_ih + b_hh for both directions as one GEMM).
#include "kernels.h"
#include "gemm_f32_tile.h"

namespace {

using namespace f32tile;

template <int GELU>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(Gemm32Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BUF_BYTES = 2 * TILE_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    const int tiles_n = (p.N + BN - 1) / BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int64_t row0 = (int64_t)tm * BM;
    const int col0 = tn * BN;
    const int nk = p.K / BKF;
    const int64_t ldw = p.ldw ? p.ldw : p.K;
    const int64_t koff = (int64_t)blockIdx.z * p.K;            // split-K: this block's k range starts here
    const float* __restrict__ Ap = p.A + koff;
    const float* __restrict__ Wp = p.W + koff;
    float* __restrict__ outp = p.out + (int64_t)blockIdx.z * p.split_stride;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * BUF_BYTES;
        stage_tile32(Ap, p.lda, row0, p.M - 1, kt * BKF, base, wave, lane);
        stage_tile32(Wp, ldw, col0, p.N_alloc - 1, kt * BKF, base + TILE_BYTES, wave, lane);
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
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    const int ncol = col0 + wc * 64 + (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t m = row0 + wr * 64 + i * 16 + (lane & 15);
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = ncol + j * 16;
            if (n >= p.N) continue;                  // N % 4 == 0, so a 4-vector is all-in or all-out
            f32x4 v = acc[i][j];
            if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
            if (GELU) v = f32x4{gelu_erf(v[0]), gelu_erf(v[1]), gelu_erf(v[2]), gelu_erf(v[3])};
            *reinterpret_cast<f32x4*>(outp + m * p.ldo + n) = v;
        }
    }
}

template <int GELU>
int launch32(const Gemm32Params& p, hipStream_t stream) {
    constexpr int lds = 4 * TILE_BYTES;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_kernel<GELU>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return -2;
        attr_set = true;
    }
    const int64_t grid = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    if (grid <= 0 || grid > 0x7fffffff) return -1;
    const unsigned gz = p.splits > 1 ? (unsigned)p.splits : 1u;
    hipLaunchKernelGGL((gemm_f32_kernel<GELU>), dim3((unsigned)grid, 1, gz), dim3(256), lds, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace

namespace {
__global__ void splitk_reduce_kernel(const float* __restrict__ partial, int splits, int64_t stride, int64_t n,
                                     float* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = partial[i];
    for (int z = 1; z < splits; ++z) s += partial[(int64_t)z * stride + i];
    dst[i] = s;
}
}  // namespace

int launch_splitk_reduce(const float* partial, int splits, int64_t stride, int64_t n, float* dst, hipStream_t stream) {
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, partial, splits, stride, n, dst);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_gemm_f32(const Gemm32Params& p, int gelu, hipStream_t stream) {
    if (p.M <= 0 || p.N <= 0 || p.N % 4 || p.K % BKF || p.N_alloc < p.N || p.lda % 4 || p.ldo % 4 || p.ldw % 4) return -1;
    if (p.splits > 1 && (p.bias || gelu || p.split_stride % 4)) return -1;
    return gelu ? launch32<1>(p, stream) : launch32<0>(p, stream);
}

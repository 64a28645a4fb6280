// Bring-up, test and measurement harnesses of the DEBUG build (include/cbas_mi355x_debug.h): stand-alone GEMM timing and
// bit-exactness entry points, the lane-overlap probe, the MX-fp8 GEMM in isolation, the MFMA neighbour.  Compiled only with
// -DCBAS_BUILD_DEBUG=1 (python -m cbas_amd.build --debug); the product library does not contain this file.
#include <math.h>
#include <string.h>
#include <new>
#include <vector>
#include <algorithm>
#include <chrono>

#include "api_common.h"
#include "kernels.h"

#if !CBAS_BUILD_DEBUG
#error "api_debug.hip belongs to the debug build only (-DCBAS_BUILD_DEBUG=1)"
#endif

extern "C" int cbas_debug_build(void) { return 1; }

// ---- bring-up: stand-alone GEMM timing / bit-exactness harness ---------------------------------
namespace {
__global__ void fill_random_f16(f16* p, int64_t n, uint32_t seed, float scale) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t x = (uint32_t)i * 2654435761u ^ seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    p[i] = (f16)(((float)(x & 0xFFFF) / 32768.0f - 1.0f) * scale);
}
__global__ void mask_bytes_kernel(uint32_t* p, int64_t n) {       // clear bit 3 of every byte: no e4m3 NaN codes
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] &= 0x77777777u;
}
__global__ void checksum_u16(const uint16_t* p, int64_t n, unsigned long long* out) {
    unsigned long long s = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        s += (unsigned long long)p[i] * (unsigned long long)((i % 1021) + 1);
    atomicAdd(out, s);
}
}  // namespace

// bring-up / tests: a register-only v_mfma_f32_32x32x16_f16 loop on every SIMD (two waves each), queued on `stream` - the
// neighbour beside which round 4's head kernels returned wrong values in lanes 48-63 (common.h); tests run the head beside it
namespace {
__global__ __launch_bounds__(512) void mfma_neighbor_kernel(int iters, float* sink) {
    typedef float f16v __attribute__((ext_vector_type(16)));
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    f16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 8; ++e) {
            unsigned x = (t * 64 + i * 8 + e) * 2654435761u; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
            a[i][e] = (f16)(((int)(x & 0xffff) - 32768) * (1.0f / 32768.0f));
            b[i][e] = (f16)(((int)(x >> 16) - 32768) * (1.0f / 32768.0f));
        }
    f16v c[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) c[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + r) & 3], b[(i * 2 + r) & 3], c[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) s += c[i][e];
    if (s == 123.456f) sink[t] = s;
}
}  // namespace

extern "C" int cbas_debug_mfma_neighbor(int iters, void* stream) {
    static float* sink = nullptr;
    int dev = 0, cus = 256;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (!sink) HIP_TRY(hipMalloc(&sink, (size_t)cus * 512 * sizeof(float)));
    if (iters <= 0) return cbas_fail(CBAS_EINVAL, "iters=%d", iters);
    hipLaunchKernelGGL(mfma_neighbor_kernel, dim3((unsigned)cus), dim3(512), 0, (hipStream_t)stream, iters, sink);
    return hipGetLastError() == hipSuccess ? CBAS_OK : cbas_fail(CBAS_EHIP, "mfma_neighbor launch failed");
}

// precision-4 GEMM (split operands) alone: epi = 1 q|k|v (RoPE), 2 residual, 3 GELU; tile = 0 (planner) / 128 / 160 / 192 / 256
// rows of the ping-pong form, -1 = the 128 x 128 8-wave kernel; prints the block timeline.  Timing only (random operands).
extern "C" int cbas_debug_gemm_split_bench(int M, int N, int K, int epi, int tile, int iters, float* ms_out) {
    if (M <= 256 || N % 256 || K % 64 || iters <= 0 || epi < 1 || epi > 3) return cbas_fail(CBAS_EINVAL, "bad split GEMM bench shape");
    if (epi == 1 && (N % 3 || (N / 3) % 64)) return cbas_fail(CBAS_EINVAL, "q|k|v bench needs N = 3 D, D a multiple of 64");
    float *A = nullptr, *Wt = nullptr, *out = nullptr, *bias = nullptr, *rope = nullptr;
    HIP_TRY(hipMalloc(&A, (int64_t)M * K * 4));
    HIP_TRY(hipMalloc(&Wt, (int64_t)N * K * 4));
    HIP_TRY(hipMalloc(&out, (int64_t)M * N * 4));
    HIP_TRY(hipMalloc(&bias, (int64_t)N * 4));
    HIP_TRY(hipMalloc(&rope, 2 * 196 * 64 * 4));
    HIP_TRY(hipMemset(bias, 0, (int64_t)N * 4));
    HIP_TRY(hipMemset(out, 0, (int64_t)M * N * 4));
    HIP_TRY(hipMemset(rope, 0, 2 * 196 * 64 * 4));
    hipLaunchKernelGGL(fill_random_f16, dim3((unsigned)(((int64_t)M * K * 2 + 255) / 256)), dim3(256), 0, 0, (f16*)A, (int64_t)M * K * 2, 1u, 1.0f);
    hipLaunchKernelGGL(fill_random_f16, dim3((unsigned)(((int64_t)N * K * 2 + 255) / 256)), dim3(256), 0, 0, (f16*)Wt, (int64_t)N * K * 2, 2u, 0.05f);
    Gemm32VitParams p{};
    p.A = A; p.lda = K; p.W = Wt; p.M = M; p.N = N; p.K = K; p.bias = bias; p.lambda = bias; p.out = out; p.ldo = N;
    p.tokens_per_frame = 201; p.n_prefix = 5; p.patches_per_frame = 196; p.rope_cos = rope; p.rope_sin = rope + 196 * 64; p.D = N / 3;
    p.rope_fac = rope; p.rope_nh = 14; p.rope_nw = 14; p.rope_magic = (unsigned)((1ull << 32) / 14u) + 1u;   // zeros: timing only
    p.split = 1; p.a_scale = 1.f; p.w_scale = 1.f; p.out_scale = 1.f;
    const GemmEpilogue e = (GemmEpilogue)epi;
    auto run = [&]() { return tile < 0 ? launch_gemm_f32_vit(e, p, 0) : launch_gemm_split_pp(e, p, 0); };
    vit32_split_set_forms(tile < 0 ? 2 : -1);
    gemm_split_pp_set_tile(tile > 0 ? tile : 0, nullptr);
    int rc = run();
    if (rc) return cbas_fail(CBAS_EINVAL, "split GEMM launch failed (rc=%d)", rc);
    HIP_TRY(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) run();
    HIP_TRY(hipEventRecord(e1, 0));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (ms_out) *ms_out = ms / iters;
    if (tile >= 0) {
        const int nblk = GEMM_STAMP_BLOCKS;
        unsigned long long* st = nullptr;
        HIP_TRY(hipMalloc(&st, (size_t)nblk * 72));
        HIP_TRY(hipMemset(st, 0, (size_t)nblk * 72));
        gemm_split_pp_set_tile(tile > 0 ? tile : 0, st);
        run();
        HIP_TRY(hipDeviceSynchronize());
        gemm_split_pp_set_tile(0, nullptr);
        std::vector<unsigned long long> hs((size_t)nblk * 9);
        HIP_TRY(hipMemcpy(hs.data(), st, (size_t)nblk * 72, hipMemcpyDeviceToHost));
        double pro = 0, loop = 0, epi_c = 0, real = 0; int n = 0;
        for (int b = 0; b < nblk; ++b) {
            if (!hs[4 * b + 3]) continue;
            pro += (double)(hs[4 * b + 1] - hs[4 * b]); loop += (double)(hs[4 * b + 2] - hs[4 * b + 1]);
            epi_c += (double)(hs[4 * b + 3] - hs[4 * b + 2]); real += (double)hs[(size_t)nblk * 4 + b]; ++n;
        }
        if (n) printf("  stamps (first tile of each workgroup): %d workgroups; avg prologue %.0f, K loop %.0f (%.0f per K-tile), epilogue %.0f cycles; "
                      "in-kernel clock %.2f GHz\n", n, pro / n, loop / n, loop / n / (K / 32), epi_c / n, real > 0 ? loop / real * 0.1 : 0.0);
        double lp = 0, ll = 0, le = 0; int ln = 0;
        for (int b = 0; b < nblk; ++b) {
            const unsigned long long* o = &hs[(size_t)nblk * 5 + 4 * (size_t)b];
            if (!o[3]) continue;
            lp += (double)(o[1] - o[0]); ll += (double)(o[2] - o[1]); le += (double)(o[3] - o[2]); ++ln;
        }
        if (ln) printf("  last tile of the %d workgroups that ran more than one: prologue %.0f, K loop %.0f, epilogue %.0f cycles\n",
                       ln, lp / ln, ll / ln, le / ln);
        fflush(stdout);
        hipFree(st);
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(A); hipFree(Wt); hipFree(out); hipFree(bias); hipFree(rope);
    return CBAS_OK;
}

// precision-4 GEMM forms against each other on the same random split operands: the ping-pong form at `tile` rows (0 = planner)
// and the 128 x 128 kernels; n_diff = output 32-bit words that differ (the forms promise 0).  epi as above; the residual
// form starts both runs from the same x.
namespace {
__global__ void count_diff_u32(const uint32_t* a, const uint32_t* b, int64_t n, unsigned long long* out) {
    unsigned long long c = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) c += a[i] != b[i];
    if (c) atomicAdd(out, c);
}
}  // namespace
extern "C" int cbas_debug_gemm_split_compare(int M, int N, int K, int epi, int tile, int64_t* n_diff) {
    if (M <= 0 || N % 256 || K % 64 || epi < 1 || epi > 3 || !n_diff) return cbas_fail(CBAS_EINVAL, "bad split GEMM compare shape");
    if (epi == 1 && (N % 3 || (N / 3) % 64)) return cbas_fail(CBAS_EINVAL, "q|k|v compare needs N = 3 D, D a multiple of 64");
    float *A = nullptr, *Wt = nullptr, *o1 = nullptr, *o2 = nullptr, *x0 = nullptr, *bias = nullptr, *rope = nullptr;
    unsigned long long* cnt = nullptr;
    const int64_t no = (int64_t)M * N;
    HIP_TRY(hipMalloc(&A, (int64_t)M * K * 4));
    HIP_TRY(hipMalloc(&Wt, (int64_t)N * K * 4));
    HIP_TRY(hipMalloc(&o1, no * 4)); HIP_TRY(hipMalloc(&o2, no * 4)); HIP_TRY(hipMalloc(&x0, no * 4));
    HIP_TRY(hipMalloc(&bias, (int64_t)N * 4 * 2));
    HIP_TRY(hipMalloc(&rope, 2 * 196 * 64 * 4));
    HIP_TRY(hipMalloc(&cnt, 8));
    HIP_TRY(hipMemset(cnt, 0, 8));
    auto fill = [&](float* p, int64_t n_f16, unsigned seed, float sc) {
        hipLaunchKernelGGL(fill_random_f16, dim3((unsigned)((n_f16 + 255) / 256)), dim3(256), 0, 0, (f16*)p, n_f16, seed, sc);
    };
    fill(A, (int64_t)M * K * 2, 1u, 1.0f);
    fill(Wt, (int64_t)N * K * 2, 2u, 0.05f);
    // fp32 side data: bias / lambda, x, cos / sin - any finite numbers do (random halves widened by the conversion kernel)
    f16* tmp = nullptr;
    HIP_TRY(hipMalloc(&tmp, (no > 2 * 196 * 64 ? no : 2 * 196 * 64) * 2));
    auto fill32 = [&](float* p, int64_t n, unsigned seed, float sc) {
        hipLaunchKernelGGL(fill_random_f16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, tmp, n, seed, sc);
        return launch_f16_to_f32(tmp, p, n, 0);
    };
    if (fill32(bias, (int64_t)N * 2, 3u, 0.5f) || fill32(x0, no, 4u, 2.0f) || fill32(rope, 2 * 196 * 64, 5u, 1.0f)) return cbas_fail(CBAS_EHIP, "fill failed");
    // the tables are angles.tile(2) ([tf]:190): columns d and d + 32 of a row hold the same number - the tile epilogue relies on it
    HIP_TRY(hipMemcpy2D(rope + 32, 64 * 4, rope, 64 * 4, 32 * 4, 2 * 196, hipMemcpyDeviceToDevice));
    Gemm32VitParams p{};
    p.A = A; p.lda = K; p.W = Wt; p.M = M; p.N = N; p.K = K; p.bias = bias; p.lambda = bias + N; p.ldo = N;
    p.tokens_per_frame = 201; p.n_prefix = 5; p.patches_per_frame = 196; p.rope_cos = rope; p.rope_sin = rope + 196 * 64; p.D = N / 3;
    p.split = 1; p.a_scale = 2.f; p.w_scale = 4.f; p.out_scale = 4.f;
    const GemmEpilogue e = (GemmEpilogue)epi;
    int rc = 0;
    for (int form = 0; form < 2 && !rc; ++form) {
        float* o = form ? o2 : o1;
        HIP_TRY(hipMemcpy(o, x0, no * 4, hipMemcpyDeviceToDevice));
        p.out = o;
        if (form == 0) { gemm_split_pp_set_tile(tile > 0 ? tile : 0, nullptr); vit32_split_set_forms(3); }
        else vit32_split_set_forms(0);
        rc = launch_gemm_f32_vit(e, p, 0);
        HIP_TRY(hipDeviceSynchronize());
    }
    gemm_split_pp_set_tile(0, nullptr);
    vit32_split_set_forms(-1);
    if (rc) return cbas_fail(CBAS_EINVAL, "split GEMM launch failed (rc=%d)", rc);
    hipLaunchKernelGGL(count_diff_u32, dim3(1024), dim3(256), 0, 0, (const uint32_t*)o1, (const uint32_t*)o2, no, cnt);
    unsigned long long hc = 0;
    HIP_TRY(hipMemcpy(&hc, cnt, 8, hipMemcpyDeviceToHost));
    *n_diff = (int64_t)hc;
    hipFree(A); hipFree(Wt); hipFree(o1); hipFree(o2); hipFree(x0); hipFree(bias); hipFree(rope); hipFree(cnt); hipFree(tmp);
    return CBAS_OK;
}

extern "C" int cbas_debug_gemm_bench(int M, int N, int K, int tile, int iters, float* ms_out,
                                     unsigned long long* checksum_out) {
    // tile >= 100: residual epilogue (o_proj/down_proj style, fp32 in/out) with tile id = tile - 100;
    // tile >= 200: q|k|v epilogue (RoPE tables of 196 patches, 201 tokens per frame, D = N / 3) with tile id = tile - 200
    // + 500: MX-fp8 operands (random e4m3 bytes, unit scales; the GELU form then writes fp8 + scales): timing only
    const bool ln = tile >= 2000;            // 2000 + ...: the LayerNorm-fold form of the epilogue (timing only: zero statistics)
    if (ln) tile -= 2000;
    const bool want_stamps = tile >= 1000;   // 1000 + tile: also print the block timeline statistics
    tile %= 1000;
    const bool f8 = tile >= 500;
    if (f8) tile -= 500;
    const bool qkv = tile >= 200;
    const bool resid = !qkv && tile >= 100;
    if (qkv) tile -= 200;
    if (resid) tile -= 100;
    if (qkv && (N % 3 || (N / 3) % 64)) return cbas_fail(CBAS_EINVAL, "q|k|v bench needs N = 3 D, D a multiple of 64");
    if (M <= 0 || N % 128 || K % 64 || iters <= 0) return cbas_fail(CBAS_EINVAL, "bad GEMM bench shape");
    const int64_t M_pad = round_up(M, 256);
    f16 *A = nullptr, *Wt = nullptr, *out = nullptr;
    float* bias = nullptr;
    unsigned long long* cs = nullptr;
    HIP_TRY(hipMalloc(&A, M_pad * (int64_t)K * 2));
    HIP_TRY(hipMalloc(&Wt, (int64_t)N * K * 2));
    HIP_TRY(hipMalloc(&out, M_pad * (int64_t)N * 2));
    HIP_TRY(hipMalloc(&bias, (int64_t)N * 4));
    HIP_TRY(hipMalloc(&cs, 8));
    HIP_TRY(hipMemset(bias, 0, (int64_t)N * 4));
    HIP_TRY(hipMemset(out, 0, M_pad * (int64_t)N * 2));
    HIP_TRY(hipMemset(cs, 0, 8));
    hipLaunchKernelGGL(fill_random_f16, dim3((unsigned)((M_pad * K + 255) / 256)), dim3(256), 0, 0, A, M_pad * K, 1u, 1.0f);
    hipLaunchKernelGGL(fill_random_f16, dim3((unsigned)(((int64_t)N * K + 255) / 256)), dim3(256), 0, 0, Wt, (int64_t)N * K, 2u, 0.05f);
    GemmParams p{};
    p.tile = tile; p.A = A; p.W = Wt; p.M = M; p.M_pad = (int)M_pad; p.N = N; p.K = K; p.bias = bias; p.out_f16 = out; p.ldo = N;
    uint32_t* sc8 = nullptr;
    if (f8) {                                    // the fp16 buffers reinterpreted as bytes (first half used); NaN codes masked out
        if (K % 256 || N % 256) return cbas_fail(CBAS_EINVAL, "fp8 GEMM bench needs N, K multiples of 256");
        hipLaunchKernelGGL(mask_bytes_kernel, dim3(4096), dim3(256), 0, 0, (uint32_t*)A, M_pad * (int64_t)K / 4);
        hipLaunchKernelGGL(mask_bytes_kernel, dim3(4096), dim3(256), 0, 0, (uint32_t*)Wt, (int64_t)N * K / 4);
        const int64_t sc_ld = M_pad > N ? M_pad : N;
        HIP_TRY(hipMalloc(&sc8, (size_t)(K / 128 + N / 128) * sc_ld * 4));
        HIP_TRY(hipMemset(sc8, 0x7a, (size_t)(K / 128 + N / 128) * sc_ld * 4));     // E8M0 2^-5 per block: products stay finite
        p.A8 = (const uint8_t*)A; p.W8 = (const uint8_t*)Wt; p.A_sc = sc8; p.W_sc = sc8; p.sc_lda = (int)sc_ld; p.sc_ldw = (int)sc_ld;
        p.out_f8 = (uint8_t*)out; p.out_sc = sc8 + (size_t)(K / 128) * sc_ld; p.sc_ldo = (int)sc_ld;
    }
    float* x32 = nullptr;
    if (resid) {
        HIP_TRY(hipMalloc(&x32, M_pad * (int64_t)N * 4));
        HIP_TRY(hipMemset(x32, 0, M_pad * (int64_t)N * 4));
        p.out_f32 = x32; p.lambda = bias;      // lambda = 0: x stays 0, timing only
    }
    float* rope = nullptr;
    if (qkv) {
        HIP_TRY(hipMalloc(&rope, 2 * 196 * 64 * 4));
        HIP_TRY(hipMemset(rope, 0, 2 * 196 * 64 * 4));
        p.rope_cos = rope; p.rope_sin = rope + 196 * 64; p.D = N / 3; p.tokens_per_frame = 201; p.n_prefix = 5;
        p.rope_fac = rope; p.rope_nh = 14; p.rope_nw = 14; p.rope_magic = (unsigned)((1ull << 32) / 14u) + 1u;   // zeros: timing only
    }
    float2* lnst = nullptr;
    f16* x16 = nullptr;
    float* colsum = nullptr;
    if (ln) {
        if (f8) return cbas_fail(CBAS_EINVAL, "the LayerNorm fold is an fp16 form");
        HIP_TRY(hipMalloc(&lnst, 4 * M_pad * sizeof(float2)));
        HIP_TRY(hipMemset(lnst, 0, 4 * M_pad * sizeof(float2)));
        HIP_TRY(hipMalloc(&colsum, (int64_t)N * 4));
        HIP_TRY(hipMemset(colsum, 0, (int64_t)N * 4));
        if (resid) { HIP_TRY(hipMalloc(&x16, M_pad * (int64_t)N * 2)); p.x16_out = x16; p.ln_out = lnst; }
        else { p.ln_in = lnst; p.ln_colsum = colsum; p.ln_parts = K / 256; p.ln_eps = 1.0f; }
        p.ln_ld = (int)M_pad;
        if (!p.tile) p.tile = GEMM_TILE_PP_AUTO;
    }
    const GemmEpilogue epi = qkv ? (ln ? EPI_QKV_LN : EPI_QKV) : resid ? (ln ? EPI_RESID_LN : EPI_RESID) : f8 ? EPI_GELU_F8 : (ln ? EPI_GELU_LN : EPI_GELU);
    int rc = launch_gemm(epi, p, 0);
    if (rc) return cbas_fail(CBAS_EINVAL, "launch_gemm failed for tile %d (rc=%d)", tile, rc);
    HIP_TRY(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) launch_gemm(epi, p, 0);
    HIP_TRY(hipEventRecord(e1, 0));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (ms_out) *ms_out = ms / iters;
    if (want_stamps) {
        const int nblk = GEMM_STAMP_BLOCKS;
        unsigned long long* st = nullptr;
        HIP_TRY(hipMalloc(&st, (size_t)nblk * 72));
        HIP_TRY(hipMemset(st, 0, (size_t)nblk * 72));
        p.stamps = st;
        launch_gemm(epi, p, 0);                  // straight after the timed launches: the clock is the loaded one
        HIP_TRY(hipDeviceSynchronize());
        std::vector<unsigned long long> hs((size_t)nblk * 9);
        HIP_TRY(hipMemcpy(hs.data(), st, (size_t)nblk * 72, hipMemcpyDeviceToHost));
        double pro = 0, loop = 0, epi_c = 0, real = 0; int n = 0;
        for (int b = 0; b < nblk; ++b) {
            if (!hs[4 * b + 3]) continue;
            pro += (double)(hs[4 * b + 1] - hs[4 * b]); loop += (double)(hs[4 * b + 2] - hs[4 * b + 1]);
            epi_c += (double)(hs[4 * b + 3] - hs[4 * b + 2]); real += (double)hs[(size_t)nblk * 4 + b]; ++n;
        }
        // s_memtime ticks are shader cycles; the K loop's span in s_memrealtime (100 MHz) ticks gives the in-kernel clock
        printf("  stamps (first tile of each workgroup): %d workgroups; avg prologue %.0f, K loop %.0f, epilogue %.0f cycles; "
               "in-kernel clock %.2f GHz\n", n, pro / n, loop / n, epi_c / n, real > 0 ? loop / real * 0.1 : 0.0);
        double lp = 0, ll = 0, le = 0; int ln = 0;
        for (int b = 0; b < nblk; ++b) {
            const unsigned long long* o = &hs[(size_t)nblk * 5 + 4 * (size_t)b];
            if (!o[3]) continue;
            lp += (double)(o[1] - o[0]); ll += (double)(o[2] - o[1]); le += (double)(o[3] - o[2]); ++ln;
        }
        if (ln) printf("  last tile of the %d workgroups that ran more than one: prologue %.0f, K loop %.0f, epilogue %.0f cycles\n",
                       ln, lp / ln, ll / ln, le / ln);
        fflush(stdout);
        p.stamps = nullptr;
        hipFree(st);
    }
    hipLaunchKernelGGL(checksum_u16, dim3(1024), dim3(256), 0, 0, (const uint16_t*)out, (int64_t)M * N, cs);
    if (checksum_out) HIP_TRY(hipMemcpy(checksum_out, cs, 8, hipMemcpyDeviceToHost));
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(A); hipFree(Wt); hipFree(out); hipFree(bias); hipFree(cs); if (x32) hipFree(x32); if (rope) hipFree(rope); if (sc8) hipFree(sc8);
    if (lnst) hipFree(lnst); if (x16) hipFree(x16); if (colsum) hipFree(colsum);
    return CBAS_OK;
}

// Bring-up: how kernels of two compute lanes share the chip.  Runs, each on its own stream and concurrently, `iters`
// launches of: bit 0 the up projection (12 864 x 3072 x 768, GELU epilogue), bit 1 LayerNorm (12 864 x 768), bit 2 the
// resident attention kernel (64 frames x 201 tokens, 12 heads), bit 3 the down projection (residual epilogue);
// ms_out[b] = average time per launch of component b as seen on its stream, ms_out[4] = wall time of the whole run.
extern "C" int cbas_debug_overlap(int mode, int iters, float* ms_out) {
    if (iters <= 0 || !ms_out) return cbas_fail(CBAS_EINVAL, "bad arguments");
    const char* te = getenv("CBAS_OVL_T");                   // tokens per frame of the attention component (default 201)
    const int Ta = te ? atoi(te) : 201;
    const int n = 64, T = 201, D = 768, F = 3072, M = n * T;
    const int64_t M_pad = round_up(M, 256);
    f16 *h16 = nullptr, *u16 = nullptr, *qkv = nullptr, *ctx = nullptr, *Wu = nullptr, *Wd = nullptr, *ln_out = nullptr;
    float *x = nullptr, *x2 = nullptr, *vec = nullptr;
    HIP_TRY(hipMalloc(&h16, M_pad * D * 2)); HIP_TRY(hipMalloc(&u16, M_pad * (int64_t)F * 2));
    HIP_TRY(hipMalloc(&qkv, M_pad * 3 * D * 2)); HIP_TRY(hipMalloc(&ctx, M_pad * D * 2)); HIP_TRY(hipMalloc(&ln_out, M_pad * D * 2));
    HIP_TRY(hipMalloc(&Wu, (int64_t)F * D * 2)); HIP_TRY(hipMalloc(&Wd, (int64_t)F * D * 2));
    HIP_TRY(hipMalloc(&x, M_pad * D * 4)); HIP_TRY(hipMalloc(&x2, M_pad * D * 4)); HIP_TRY(hipMalloc(&vec, F * 4));
    HIP_TRY(hipMemset(vec, 0, F * 4)); HIP_TRY(hipMemset(x, 0, M_pad * D * 4)); HIP_TRY(hipMemset(x2, 0, M_pad * D * 4));
    auto fill = [&](f16* p, int64_t cnt, unsigned seed, float sc) {
        hipLaunchKernelGGL(fill_random_f16, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, 0, p, cnt, seed, sc);
    };
    fill(h16, M_pad * D, 1u, 1.0f); fill(u16, M_pad * (int64_t)F, 2u, 1.0f); fill(qkv, M_pad * 3 * D, 3u, 1.0f);
    fill(Wu, (int64_t)F * D, 4u, 0.05f); fill(Wd, (int64_t)F * D, 5u, 0.02f);
    HIP_TRY(hipDeviceSynchronize());
    GemmParams up{}; up.A = h16; up.W = Wu; up.M = M; up.M_pad = (int)M_pad; up.N = F; up.K = D; up.bias = vec; up.out_f16 = u16; up.ldo = F;
    GemmParams dn{}; dn.A = u16; dn.W = Wd; dn.M = M; dn.M_pad = (int)M_pad; dn.N = D; dn.K = F; dn.bias = vec; dn.lambda = vec; dn.out_f32 = x2; dn.ldo = D;
    hipStream_t st[4]; hipEvent_t e0[4], e1[4];
    for (int b = 0; b < 4; ++b) { HIP_TRY(hipStreamCreateWithFlags(&st[b], hipStreamNonBlocking)); HIP_TRY(hipEventCreate(&e0[b])); HIP_TRY(hipEventCreate(&e1[b])); }
    auto launch = [&](int b) -> int {
        switch (b) {
            case 0: return launch_gemm(EPI_GELU, up, st[0]);
            case 1: return launch_layernorm_f16(x, D, vec, vec, ln_out, M, D, 1e-5f, st[1]);
            case 2: return launch_attention(qkv, nullptr, ctx, nullptr, 0, n * T / Ta, Ta, D, 12, st[2]);
            default: return launch_gemm(EPI_RESID, dn, st[3]);
        }
    };
    for (int b = 0; b < 4; ++b) if (mode & (1 << b)) for (int i = 0; i < 3; ++i) if (launch(b)) return cbas_fail(CBAS_EINVAL, "launch %d failed", b);
    HIP_TRY(hipDeviceSynchronize());
    const auto w0 = std::chrono::steady_clock::now();
    for (int b = 0; b < 4; ++b) if (mode & (1 << b)) HIP_TRY(hipEventRecord(e0[b], st[b]));
    for (int i = 0; i < iters; ++i)                       // interleaved submission, like two lanes queueing their kernels
        for (int b = 0; b < 4; ++b) if (mode & (1 << b)) launch(b);
    for (int b = 0; b < 4; ++b) if (mode & (1 << b)) HIP_TRY(hipEventRecord(e1[b], st[b]));
    HIP_TRY(hipDeviceSynchronize());
    ms_out[4] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - w0).count();
    for (int b = 0; b < 4; ++b) {
        ms_out[b] = 0.f;
        if (mode & (1 << b)) { float ms = 0; HIP_TRY(hipEventElapsedTime(&ms, e0[b], e1[b])); ms_out[b] = ms / iters; }
        hipStreamDestroy(st[b]); hipEventDestroy(e0[b]); hipEventDestroy(e1[b]);
    }
    hipFree(h16); hipFree(u16); hipFree(qkv); hipFree(ctx); hipFree(ln_out); hipFree(Wu); hipFree(Wd); hipFree(x); hipFree(x2); hipFree(vec);
    return CBAS_OK;
}

// Bring-up / test harness of the MX-fp8 GEMM: quantise A [M][K] and W [N][K] (fp32, host) with the library's own
// block quantiser, run out = A_q * W_q^T through the fp8 ping-pong kernel (residual epilogue on a zero stream with
// bias 0 and lambda 1), and hand back the product together with the quantised operands, so a test can recompute it
// from exactly those bytes and scales.  A_sc / W_sc: [K/128][M_pad resp. N] dwords (GemmParams layout), M_pad = M up to 256.
extern "C" int cbas_debug_gemm_f8(int M, int N, int K, int tile, const float* A_host, const float* W_host, float* out_host,
                                  uint8_t* A8_host, uint32_t* Asc_host, uint8_t* W8_host, uint32_t* Wsc_host) {
    if (M <= 0 || N % 256 || K % 256 || !A_host || !W_host || !out_host) return cbas_fail(CBAS_EINVAL, "bad fp8 GEMM test shape");
    const int64_t M_pad = round_up(M, 256);
    float *A = nullptr, *W = nullptr, *x = nullptr, *bias = nullptr, *lam = nullptr;
    uint8_t *A8 = nullptr, *W8 = nullptr;
    uint32_t *Asc = nullptr, *Wsc = nullptr;
    HIP_TRY(hipMalloc(&A, M_pad * (int64_t)K * 4));
    HIP_TRY(hipMalloc(&W, (int64_t)N * K * 4));
    HIP_TRY(hipMalloc(&x, M_pad * (int64_t)N * 4));
    HIP_TRY(hipMalloc(&bias, (int64_t)N * 4));
    HIP_TRY(hipMalloc(&lam, (int64_t)N * 4));
    HIP_TRY(hipMalloc(&A8, M_pad * (int64_t)K));
    HIP_TRY(hipMalloc(&W8, (int64_t)N * K));
    HIP_TRY(hipMalloc(&Asc, M_pad * (int64_t)K / 32));
    HIP_TRY(hipMalloc(&Wsc, (int64_t)N * K / 32));
    HIP_TRY(hipMemset(A, 0, M_pad * (int64_t)K * 4));
    HIP_TRY(hipMemset(x, 0, M_pad * (int64_t)N * 4));
    HIP_TRY(hipMemset(bias, 0, (int64_t)N * 4));
    HIP_TRY(hipMemcpy(A, A_host, (int64_t)M * K * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(W, W_host, (int64_t)N * K * 4, hipMemcpyHostToDevice));
    std::vector<float> ones((size_t)N, 1.0f);
    HIP_TRY(hipMemcpy(lam, ones.data(), (int64_t)N * 4, hipMemcpyHostToDevice));
    LAUNCH_TRY(launch_pack_fp8_weight(A, A8, Asc, (int)M_pad, K, (int)M_pad, 0, 0));
    LAUNCH_TRY(launch_pack_fp8_weight(W, W8, Wsc, N, K, N, 0, 0));
    GemmParams p{};
    p.tile = tile; p.A8 = A8; p.W8 = W8; p.A_sc = Asc; p.W_sc = Wsc; p.sc_lda = (int)M_pad;
    p.M = M; p.M_pad = (int)M_pad; p.N = N; p.K = K; p.lda = K; p.bias = bias; p.lambda = lam; p.out_f32 = x; p.ldo = N;
    int rc = tile ? launch_gemm_8ph(EPI_RESID, p, tile, 0) : launch_gemm(EPI_RESID, p, 0);
    if (rc) return cbas_fail(CBAS_EINVAL, "fp8 GEMM launch failed (rc=%d)", rc);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out_host, x, (int64_t)M * N * 4, hipMemcpyDeviceToHost));
    if (A8_host) HIP_TRY(hipMemcpy(A8_host, A8, (int64_t)M * K, hipMemcpyDeviceToHost));
    if (Asc_host) HIP_TRY(hipMemcpy(Asc_host, Asc, M_pad * (int64_t)K / 32, hipMemcpyDeviceToHost));
    if (W8_host) HIP_TRY(hipMemcpy(W8_host, W8, (int64_t)N * K, hipMemcpyDeviceToHost));
    if (Wsc_host) HIP_TRY(hipMemcpy(Wsc_host, Wsc, (int64_t)N * K / 32, hipMemcpyDeviceToHost));
    hipFree(A); hipFree(W); hipFree(x); hipFree(bias); hipFree(lam); hipFree(A8); hipFree(W8); hipFree(Asc); hipFree(Wsc);
    return CBAS_OK;
}

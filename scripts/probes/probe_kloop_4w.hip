// A K loop with fewer LDS bytes per MFMA ("what the K loops are really bound by"): 256 x 256 x 64 fp16
// tile, FOUR waves per workgroup as 2 x 2, each wave a 128 x 128 block of C (64 accumulator tiles = 256 registers), one wave
// per SIMD.  Per K-tile a wave reads 16 KB of A + 16 KB of B fragments (the 8-wave ping-pong kernels: 16 + 8 KB per wave,
// 192 KB per workgroup against 128 KB here) and the workgroup stages the same 64 KB by LDS-DMA into a two-deep ring.
// Schedule per K-tile t (two k32 steps, fragment sets F[0] / F[1]):
//   step 0: 64 MFMAs on F[0]; in the gaps between them the 16 ds_read_b128 of F[1] <- (stage t & 1, step 1)
//   s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier        K-tile t+1 has landed everywhere; stage t & 1 is free
//   step 1: 64 MFMAs on F[1]; beside them the 16 LDS-DMA of K-tile t+2 -> stage t & 1 (two per group of 8 MFMAs) and the reads F[0] <- (stage (t+1) & 1, 0)
// No epilogue: the probe measures the loop alone (cycles per K-tile, s_memtime) on L2-resident operands (tiles_m = 4) and on
// the whole o_proj panel (tiles_m = 51), and checks one tile's sum against a host dot product of the same fp16 data.
//      usage: probe_kloop_4w [tile_steps = 40 [tiles_m = 4]]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <type_traits>
#include <vector>

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int K = 768, KT = K / 64, M_ROWS = 13056, N_ROWS = 768, TILES_N = 3, STAGE = 65536;

__device__ __forceinline__ f16x8 read_frag(const char* tile, int row, int chunk) {
    return *reinterpret_cast<const f16x8*>(tile + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}

__global__ __launch_bounds__(256, 1) void kloop_4w(const f16* __restrict__ A, const f16* __restrict__ W, int steps, int tiles_m,
                                                   unsigned long long* out, float* csum) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int frow = lane & 15, fq = lane >> 4;
    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f16x8 fa[2][8], fb[2][8];

    // staging: piece = wave + 4 i (8 rows x 128 B each), i = 0..7 for A and for B; the swizzled chunk does not depend on i
    const int r_in = wave * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((r_in >> 1) & 7);
    const f16 *dma_a = nullptr, *dma_b = nullptr;
    char* dma_base = nullptr;
    auto stage_begin = [&](int g) {                   // global K-tile index g -> tile (g / KT), k-tile (g % KT)
        const int s = g / KT, kt = g - s * KT;
        const int tile = (blockIdx.x + s * gridDim.x) % (tiles_m * TILES_N);
        dma_a = A + (size_t)((tile / TILES_N) * 256 + r_in) * K + kt * 64 + chunk * 8;
        dma_b = W + (size_t)((tile % TILES_N) * 256 + r_in) * K + kt * 64 + chunk * 8;
        dma_base = smem + (g & 1) * STAGE + wave * 1024;
    };
    auto stage_piece = [&](int i) {
        __builtin_amdgcn_global_load_lds(GLB_PTR(dma_a + (size_t)i * 32 * K), LDS_PTR(dma_base + i * 4096), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(GLB_PTR(dma_b + (size_t)i * 32 * K), LDS_PTR(dma_base + 32768 + i * 4096), 16, 0, 0);
    };
    auto stage = [&](int g) {
        stage_begin(g);
#pragma unroll
        for (int i = 0; i < 8; ++i) stage_piece(i);
    };
    // The accumulators live in AGPRs for the whole kernel ("+a": left to itself the compiler keeps them in VGPRs, uses the
    // AGPRs as spill space and wraps every MFMA in v_accvgpr moves).  One k32 step = 8 groups (A row tile i) of 8 MFMAs; the
    // reads of the NEXT fragment set are issued two per group.
    auto step = [&](auto cur, auto nxt, const char* tile, int kstep, bool dma) {
        constexpr int C = decltype(cur)::value, N = decltype(nxt)::value;
        const char* At = tile;
        const char* Bt = tile + 32768;
        // Memory instructions go one at a time into the gaps between MFMAs (16 cycles of matrix pipe each).  Measured forms
        // (cycles per K-tile, 2 048 of them MFMA; operands L2-resident):
        //   reads and DMA issues lumped between groups of 8 MFMAs                                         2 733
        //   THIS FORM: per group of 8 MFMAs (one A row tile x B 0-7) one A read, one B read, two DMA issues,
        //     each in a gap of its own                                                                    2 470
        //   all 16 reads in the first 16 gaps of a step (4 waves x 1 KB per 16 cycles > the LDS rate: the
        //     queue fills and the in-order wave stops issuing MFMAs)                                      2 823
        //   one read every third gap, MFMAs walking C in quadrants so the last reads are needed last     2 554
        // i.e. with ONE wave per SIMD every memory instruction costs ~9 cycles of matrix pipe whatever its place: nobody else
        // issues MFMAs meanwhile.  The 8-wave ping-pong kernels (two waves per SIMD, 2 360 cycles) stay ahead.
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(fb[C][j]), "v"(fa[C][i]));
                if (j == 0) fa[N][i] = read_frag(At, wr * 128 + i * 16 + frow, kstep * 4 + fq);
                if (j == 2) fb[N][i] = read_frag(Bt, wc * 128 + i * 16 + frow, kstep * 4 + fq);
                if (dma && j == 4)
                    __builtin_amdgcn_global_load_lds(GLB_PTR(dma_a + (size_t)i * 32 * K), LDS_PTR(dma_base + i * 4096), 16, 0, 0);
                if (dma && j == 6)
                    __builtin_amdgcn_global_load_lds(GLB_PTR(dma_b + (size_t)i * 32 * K), LDS_PTR(dma_base + 32768 + i * 4096), 16, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;

    const int G = steps * KT;
    stage(0);
    stage(1);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");            // K-tile 0 has landed (this wave's part)
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        fa[0][i] = read_frag(smem, wr * 128 + i * 16 + frow, fq);
        fb[0][i] = read_frag(smem + 32768, wc * 128 + i * 16 + frow, fq);
    }
    __builtin_amdgcn_s_barrier();
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int g = 0; g < G; ++g) {
        // step 0 of K-tile g: F[0] is in registers; fetch F[1] <- (stage g & 1, step 1) beside the MFMAs
        step(S0{}, S1{}, smem + (g & 1) * STAGE, 1, false);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                            // K-tile g+1 landed everywhere; every read of stage g & 1 retired
        // step 1: refill stage g & 1 with K-tile g+2, fetch the next K-tile's F[0] <- (stage (g+1) & 1, step 0)
        // (past the end both wrap onto valid K-tiles: two harmless extra refills, no branches in the loop)
        stage_begin(g + 2);
        step(S1{}, S0{}, smem + ((g + 1) & 1) * STAGE, 0, true);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");           // the last MFMAs' results, before the accumulators are read
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    if (tid == 0) { out[blockIdx.x * 2] = c1 - c0; out[blockIdx.x * 2 + 1] = w1 - w0; }
    // checksum: sum over this wave's accumulators (C^T layout is irrelevant for a sum)
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    atomicAdd(&csum[blockIdx.x], s);
}

int main(int argc, char** argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 40;
    const int tiles_m = argc > 2 ? atoi(argv[2]) : 4;
    if (steps < 1 || steps > 4000 || tiles_m < 1 || tiles_m > 51) { printf("arguments out of range\n"); return 1; }
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    std::vector<f16> hA((size_t)M_ROWS * K), hW((size_t)N_ROWS * K);
    unsigned x = 12345u;
    auto rnd = [&]() { x = x * 1664525u + 1013904223u; return (float)((int)(x >> 16) - 32768) * (1.0f / 32768.0f); };
    for (auto& v : hA) v = (f16)rnd();
    for (auto& v : hW) v = (f16)rnd();
    f16 *A, *W;
    unsigned long long* d_out;
    float* d_sum;
    CK(hipMalloc(&A, hA.size() * 2)); CK(hipMalloc(&W, hW.size() * 2));
    CK(hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(W, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_out, n_cu * 16)); CK(hipMalloc(&d_sum, n_cu * 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kloop_4w), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE));

    // correctness: one workgroup, one tile (steps = 1): sum of C over the 256 x 256 tile 0 = sum_k (sum_m A[m][k]) (sum_n W[n][k])
    CK(hipMemset(d_sum, 0, n_cu * 4));
    hipLaunchKernelGGL(kloop_4w, dim3(1), dim3(256), 2 * STAGE, 0, A, W, 1, tiles_m, d_out, d_sum);
    CK(hipDeviceSynchronize());
    float got = 0.f;
    CK(hipMemcpy(&got, d_sum, 4, hipMemcpyDeviceToHost));
    double want = 0.0, mag = 0.0;
    for (int k = 0; k < K; ++k) {
        double sa = 0.0, sw = 0.0;
        for (int m = 0; m < 256; ++m) sa += (double)hA[(size_t)m * K + k];
        for (int n = 0; n < 256; ++n) sw += (double)hW[(size_t)n * K + k];
        want += sa * sw;
        mag += fabs(sa * sw);
    }
    printf("{\"check\": \"sum of tile 0\", \"device\": %.4f, \"host\": %.4f, \"rel_to_magnitude\": %.2e}\n", got, want, fabs(got - want) / mag);

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kloop_4w, dim3(n_cu), dim3(256), 2 * STAGE, 0, A, W, steps, tiles_m, d_out, d_sum);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    std::vector<unsigned long long> h(n_cu * 2);
    CK(hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost));
    double cyc = 0, wall = 0;
    for (int i = 0; i < n_cu; ++i) { cyc += h[i * 2]; wall += h[i * 2 + 1]; }
    cyc /= n_cu; wall /= n_cu;
    const double ktiles = (double)steps * KT, flop = ktiles * 256.0 * 256.0 * 64.0 * 2.0;
    printf("{\"kernel\": \"kloop_4w (4 waves x 128x128)\", \"a_rows\": %d, \"k_tiles\": %.0f, \"cycles_per_k_tile\": %.0f, \"mfma_cycles_per_k_tile\": 2048, "
           "\"core_clock_GHz\": %.3f, \"chip_TFLOPs\": %.0f, \"kernel_ms\": %.3f}\n",
           tiles_m * 256, ktiles, cyc / ktiles, cyc / (wall * 10.0), flop * n_cu / (ms * 1e-3) / 1e12, ms);
    return 0;
}

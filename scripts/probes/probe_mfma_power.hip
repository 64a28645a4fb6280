// What the matrix pipe sustains with NOTHING else going on: register-resident MFMA loops (no LDS, no memory) on every SIMD of
// the chip, run for a few seconds each while scripts/mfma_power.sh samples rocm-smi.  The TFLOP/s it prints is the ceiling the
// board's power limit leaves for dense fp16 / MX-fp8 MFMA work on non-trivial operands - the number to hold the GEMM family
// against, next to the spec-sheet peak.      usage: probe_mfma_power <shape 0|1|2> <seconds> [waves per SIMD = 2]
//   0: v_mfma_f32_16x16x32_f16   1: v_mfma_f32_32x32x16_f16   2: v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 x e4m3)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef int i8v __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned mix(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int SHAPE>
__global__ __launch_bounds__(512) void mfma_loop(int iters, float* sink) {
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    // operands with full-entropy mantissas in a sane range (random sign, exponent near 1)
    h8 a[4], b[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 8; ++e) {
            a[i][e] = (_Float16)(((int)(mix(t * 64 + i * 8 + e) & 0xffff) - 32768) * (1.0f / 32768.0f));
            b[i][e] = (_Float16)(((int)(mix(t * 64 + 32 + i * 8 + e) & 0xffff) - 32768) * (1.0f / 32768.0f));
        }
    if (SHAPE == 0) {
        f4 c[16];
        for (int i = 0; i < 16; ++i) c[i] = f4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[i >> 2], c[i], 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < 16; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
        if (s == 123.456f) sink[t] = s;
    } else if (SHAPE == 1) {
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
    } else {
        i8v qa[2], qb[2];
        for (int i = 0; i < 2; ++i)
            for (int e = 0; e < 8; ++e) {
                qa[i][e] = (int)(mix(t * 32 + i * 8 + e) & 0x77777777u);      // e4m3 bytes, exponents below the NaN code
                qb[i][e] = (int)(mix(t * 32 + 16 + i * 8 + e) & 0x77777777u);
            }
        f4 c[16];
        for (int i = 0; i < 16; ++i) c[i] = f4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                c[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(qa[i & 1], qb[i >> 1], c[i], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
        float s = 0.f;
        for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
        if (s == 123.456f) sink[t] = s;
    }
}

int main(int argc, char** argv) {
    const int shape = argc > 1 ? atoi(argv[1]) : 0;
    const double seconds = argc > 2 ? atof(argv[2]) : 5.0;
    const int wps = argc > 3 ? atoi(argv[3]) : 2;
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const int blocks = cus * wps / 2;                      // 512 threads = 8 waves = 2 per SIMD
    float* sink;
    hipMalloc(&sink, (size_t)blocks * 512 * 4);
    const int iters = 20000;
    // FLOPs per wave per iteration: 16 x (16x16x32x2) = 262 144 | 8 x (32x32x16x2) = 262 144 | 4 x (16x16x128x2) = 262 144
    const double flop_per_launch = (double)blocks * 8 * iters * 262144.0;
    auto launch = [&] {
        if (shape == 0) hipLaunchKernelGGL(mfma_loop<0>, dim3(blocks), dim3(512), 0, 0, iters, sink);
        else if (shape == 1) hipLaunchKernelGGL(mfma_loop<1>, dim3(blocks), dim3(512), 0, 0, iters, sink);
        else hipLaunchKernelGGL(mfma_loop<2>, dim3(blocks), dim3(512), 0, 0, iters, sink);
    };
    launch();
    hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    long n = 0;
    double el = 0;
    while (el < seconds) {
        for (int i = 0; i < 8; ++i) launch();
        hipDeviceSynchronize();
        n += 8;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    const char* names[3] = {"v_mfma_f32_16x16x32_f16", "v_mfma_f32_32x32x16_f16", "v_mfma_scale_f32_16x16x128_f8f6f4"};
    printf("%s: %d CUs x %d waves per SIMD, %.1f s: %.1f TFLOP/s\n", names[shape], cus, wps, el, n * flop_per_launch / el / 1e12);
    return 0;
}

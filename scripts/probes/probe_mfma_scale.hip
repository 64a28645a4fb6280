// Bring-up probe: operand / scale layout of v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands on gfx950.
// Tries candidate lane->k maps against a CPU reference on exactly representable data and prints which match.
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/probe_mfma_scale.hip -o gpurun_out/probe_mfma_scale
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <math.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __host__ inline int kmap(int layout, int lane, int j) {   // k index of byte j (0..31) of lane's fragment
    const int q = lane >> 4;
    switch (layout) {
        case 0: return 32 * q + j;                            // 32 consecutive k per lane
        case 1: return 16 * q + (j & 15) + 64 * (j >> 4);     // two K=64 halves
        case 2: return 8 * q + (j & 7) + 32 * (j >> 3);       // four K=32 quarters
        default: return 4 * q + (j & 3) + 16 * (j >> 2);
    }
}

__global__ void probe(const uint8_t* A, const uint8_t* B, const uint8_t* sa, const uint8_t* sb, float* C, int layout,
                      int use_scale) {
    const int lane = threadIdx.x;
    union { v8i v; uint8_t b[32]; } a, b;
    for (int j = 0; j < 32; ++j) {
        const int k = kmap(layout, lane, j);
        a.b[j] = A[(lane & 15) * 128 + k];
        b.b[j] = B[k * 16 + (lane & 15)];
    }
    // scale of the lane's own 32-block (layout 0: block = lane>>4); E8M0 in byte 0 of the scale register
    int s_a = 127, s_b = 127;
    if (use_scale) { s_a = sa[(lane & 15) * 4 + (lane >> 4)]; s_b = sb[(lane >> 4) * 16 + (lane & 15)]; }
    v4f c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a.v, b.v, c, 0, 0, 0, s_a, 0, s_b);
    for (int r = 0; r < 4; ++r) C[((lane >> 4) * 4 + r) * 16 + (lane & 15)] = c[r];
}

static float e4m3(uint8_t v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float x = e == 0 ? ldexpf((float)m / 8.f, -6) : ldexpf(1.f + m / 8.f, e - 7);
    return s ? -x : x;
}

int main() {
    const uint8_t vals[] = {0x00, 0x30, 0x38, 0x3C, 0x40, 0x44, 0x48, 0xB0, 0xB8, 0xBC, 0xC0, 0xC4, 0x28, 0xA8};
    uint8_t hA[16 * 128], hB[128 * 16], hsa[16 * 4], hsb[4 * 16];
    srand(1);
    for (auto& v : hA) v = vals[rand() % 14];
    for (auto& v : hB) v = vals[rand() % 14];
    for (auto& v : hsa) v = 125 + rand() % 5;
    for (auto& v : hsb) v = 125 + rand() % 5;
    uint8_t *dA, *dB, *dsa, *dsb; float* dC;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dsa, sizeof hsa); hipMalloc(&dsb, sizeof hsb);
    hipMalloc(&dC, 256 * 4);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipMemcpy(dsa, hsa, sizeof hsa, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, sizeof hsb, hipMemcpyHostToDevice);
    for (int use_scale = 0; use_scale < 2; ++use_scale) {
        float ref[256];
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                double s = 0;
                for (int k = 0; k < 128; ++k) {
                    double fa = use_scale ? ldexp(1.0, hsa[i * 4 + k / 32] - 127) : 1.0;
                    double fb = use_scale ? ldexp(1.0, hsb[(k / 32) * 16 + j] - 127) : 1.0;
                    s += e4m3(hA[i * 128 + k]) * fa * e4m3(hB[k * 16 + j]) * fb;
                }
                ref[i * 16 + j] = (float)s;
            }
        for (int layout = 0; layout < 4; ++layout) {
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dC, layout, use_scale);
            float hC[256];
            hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
            double md = 0, mdT = 0;
            for (int i = 0; i < 16; ++i)
                for (int j = 0; j < 16; ++j) {
                    md = fmax(md, fabs(hC[i * 16 + j] - ref[i * 16 + j]));
                    mdT = fmax(mdT, fabs(hC[j * 16 + i] - ref[i * 16 + j]));
                }
            printf("scale=%d layout=%d  max|C-ref|=%g  max|C^T-ref|=%g  %s\n", use_scale, layout, md, mdT,
                   md == 0 ? "MATCH" : (mdT == 0 ? "MATCH-TRANSPOSED" : ""));
        }
    }
    // ---- accumulation precision: full-range random e4m3 bytes and scales, one instruction, exact reference in double
    {
        srand(7);
        double worst = 0, worst_narrow = 0;
        for (int trial = 0; trial < 20; ++trial) {
            const bool narrow = trial >= 10;                  // narrow: all scales equal, elements in [0.5, 2)
            for (auto& v : hA) { do { v = (uint8_t)(rand() & 0xff); } while ((v & 0x7f) == 0x7f); if (narrow) v = (v & 0x87) | 0x30 | (rand() & 8); }
            for (auto& v : hB) { do { v = (uint8_t)(rand() & 0xff); } while ((v & 0x7f) == 0x7f); if (narrow) v = (v & 0x87) | 0x30 | (rand() & 8); }
            for (auto& v : hsa) v = narrow ? 127 : 120 + rand() % 15;
            for (auto& v : hsb) v = narrow ? 127 : 120 + rand() % 15;
            hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
            hipMemcpy(dsa, hsa, sizeof hsa, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, sizeof hsb, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dC, 1, 1);
            float hC[256];
            hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
            for (int i = 0; i < 16; ++i)
                for (int j = 0; j < 16; ++j) {
                    double s = 0, sabs = 0;
                    for (int k = 0; k < 128; ++k) {
                        const double t = (double)e4m3(hA[i * 128 + k]) * ldexp(1.0, hsa[i * 4 + k / 32] - 127) *
                                         (double)e4m3(hB[k * 16 + j]) * ldexp(1.0, hsb[(k / 32) * 16 + j] - 127);
                        s += t; sabs += fabs(t);
                    }
                    const double e = fabs(hC[i * 16 + j] - s) / sabs;          // relative to sum |a b|
                    if (narrow) worst_narrow = fmax(worst_narrow, e); else worst = fmax(worst, e);
                }
        }
        printf("accumulation: max |C - exact| / sum|a*b|  wide dynamic range %.3e   narrow %.3e   (fp32 chain would be ~1e-7)\n",
               worst, worst_narrow);
    }
    return 0;
}

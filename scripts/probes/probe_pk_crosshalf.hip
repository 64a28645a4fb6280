// Stand-alone reproducer for DESIGN.md section 4 ("The co-residency corruption, root-caused"): does
//     v_pk_add_f32 vD[0:1], vA[0:1], vB[0:1] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]      ; {A.lo - B.hi, A.hi - B.lo}
// compute its LOW half wrong beside waves that keep the matrix pipe busy with v_mfma_f32_32x32x16_f16?  A victim kernel executes
// that instruction (inline asm, nothing for the compiler to rearrange) in a loop on per-lane operands and compares both halves
// with scalar v_sub_f32 results; a neighbour kernel (the library's cbas_debug_mfma_neighbor loop) runs on another stream, two
// waves per SIMD on every CU.  Forms: 0 the cross-half form above, 1 the plain form (operands moved into place first: control),
// 2 the cross-half selection done by v_pk_mov_b32 and a plain v_pk_add_f32.  Prints one JSON line per form: instruction
// executions (per wave), wrong low / high halves by 16-lane group, and the first wrong sample.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/probes/bin/probe_pk_crosshalf scripts/probes/probe_pk_crosshalf.hip
//   scripts/probes/bin/probe_pk_crosshalf [seconds per form] [neighbour: 0 none | 1 v_mfma_f32_32x32x16_f16 | 2 v_mfma_f32_16x16x32_f16]
//   PK_FORMS=3,4,5,6,7 (digits; 8 = the DPP wave sum, 9 = the mirror multiply, a = SDWA half reads, b / c = the selection on src0 / on an FMA's addend) selects further forms: the low-from-high selection without negation / on a multiply / on an FMA, and the
//   MIRROR (high from low: the scalar-broadcast form the compiler uses throughout the GEMM epilogues) on an add and on an FMA
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

typedef float f32x4v __attribute__((ext_vector_type(4)));
// the same with v_mfma_f32_16x16x32_f16 (the instruction of the library's own fp16 GEMMs and attention kernels)
__global__ __launch_bounds__(512) void mfma16_neighbor_kernel(int iters, float* sink) {
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    f16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 8; ++e) {
            unsigned x = (t * 64 + i * 8 + e) * 2654435761u; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
            a[i][e] = (f16)(((int)(x & 0xffff) - 32768) * (1.0f / 32768.0f));
            b[i][e] = (f16)(((int)(x >> 16) - 32768) * (1.0f / 32768.0f));
        }
    f32x4v c[8];
    for (int i = 0; i < 8; ++i) c[i] = f32x4v{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i + r) & 3], b[(i * 2 + r) & 3], c[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][3];
    if (s == 123.456f) sink[t] = s;
}

__global__ __launch_bounds__(512) void mfma_neighbor_kernel(int iters, float* sink) {
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

typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
template <int CTRL> static __device__ __forceinline__ float dpp_f32(float v) {
    return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xf, 0xf, false));
}
// the wave sums of the library's LayerNorm kernels (csrc/common.h: wave_sum_dpp) and the __shfl_xor butterfly they replaced
static __device__ __forceinline__ float wave_sum_dpp(float v) {
    u32x2_t r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    v += dpp_f32<0x128>(v); v += dpp_f32<0x124>(v); v += dpp_f32<0x122>(v); v += dpp_f32<0x121>(v);
    return v;
}
static __device__ __forceinline__ float wave_sum_shfl(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

struct Counts { unsigned long long lo_wrong[4], hi_wrong[4], execs; float sample[8]; int have_sample; };

template <int FORM>
__global__ __launch_bounds__(256) void victim_kernel(int iters, Counts* out) {
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned x = t * 2654435761u + 12345u;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return ((int)(x & 0xffffff) - 0x800000) * (1.0f / 0x100000); };
    // operands live in register PAIRS for the whole loop and are updated by packed ops without operand selection, so that the
    // only cross-half selection in the loop is the instruction under test (building {b, a} from scalars each round made the
    // compiler emit v_pk_mov_b32 ... op_sel:[1,0] in every form, the control included)
    float a = rnd(), b = rnd(), c = rnd();
    f32x2 A = {b, a}, B = {b, c}, Bs = {c, b};                 // A = v[10:11] of the kernel, B = v[8:9]; Bs = B's halves in place for the control
    const f32x2 mA = {1.001f, 0.999f}, cA = {-0.11f, 0.37f}, mB = {1.001f, 0.998f}, cB = {-0.11f, 0.23f};
    const f32x2 mS = {0.998f, 1.001f}, cS = {0.23f, -0.11f};
    unsigned lo_bad = 0, hi_bad = 0;
    float s_lo = 0, s_hi = 0, s_a = 0, s_b = 0, s_c = 0, g_lo = 0, g_hi = 0;
    for (int it = 0; it < iters; ++it) {
        f32x2 R;
        float want_lo, want_hi;
        if (FORM == 8) {                                        // not a packed op: the permlane-swap + DPP wave sum against the shuffle butterfly
            want_lo = wave_sum_shfl(A[0]); want_hi = wave_sum_shfl(A[1]);
            R[0] = wave_sum_dpp(A[0]); R[1] = wave_sum_dpp(A[1]);
        } else
        if (FORM <= 2) {
            asm volatile("v_sub_f32 %0, %1, %2" : "=v"(want_lo) : "v"(A[0]), "v"(B[1]));       // b - c
            asm volatile("v_sub_f32 %0, %1, %2" : "=v"(want_hi) : "v"(A[1]), "v"(B[0]));       // a - b
        }
        if (FORM == 0) {
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(R) : "v"(A), "v"(B));
        } else if (FORM == 1) {
            asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(R) : "v"(A), "v"(Bs));
        } else if (FORM == 2) {
            f32x2 Bm;
            asm volatile("v_pk_mov_b32 %0, %1, %1 op_sel:[1,0]\n\ts_nop 0" : "=v"(Bm) : "v"(B));      // {B.hi, B.lo} = {c, b}
            asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(R) : "v"(A), "v"(Bm));
        } else if (FORM == 3) {                                 // low half from src1's HIGH half, no negation: {A.lo + B.hi, A.hi + B.hi}
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(want_lo) : "v"(A[0]), "v"(B[1]));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(want_hi) : "v"(A[1]), "v"(B[1]));
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(R) : "v"(A), "v"(B));
        } else if (FORM == 4) {                                 // the same selection on a multiply
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(want_lo) : "v"(A[0]), "v"(B[1]));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(want_hi) : "v"(A[1]), "v"(B[1]));
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(R) : "v"(A), "v"(B));
        } else if (FORM == 5) {                                 // ... on a fused multiply-add (src1's high half for both halves)
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(want_lo) : "v"(A[0]), "v"(B[1]), "v"(Bs[0]));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(want_hi) : "v"(A[1]), "v"(B[1]), "v"(Bs[1]));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(R) : "v"(A), "v"(B), "v"(Bs));
        } else if (FORM == 6) {                                 // the MIRROR: high half from src1's LOW half (the broadcast form): {A.lo + B.lo, A.hi + B.lo}
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(want_lo) : "v"(A[0]), "v"(B[0]));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(want_hi) : "v"(A[1]), "v"(B[0]));
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(R) : "v"(A), "v"(B));
        } else if (FORM == 7) {                                 // the mirror on a fused multiply-add: the GEMM epilogues' form
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(want_lo) : "v"(A[0]), "v"(B[0]), "v"(Bs[0]));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(want_hi) : "v"(A[1]), "v"(B[0]), "v"(Bs[1]));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(R) : "v"(A), "v"(B), "v"(Bs));
        } else if (FORM == 9) {                                 // the mirror on a multiply, src0's LOW half for both halves (3 029 sites in the library)
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(want_lo) : "v"(A[0]), "v"(B[0]));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(want_hi) : "v"(A[0]), "v"(B[1]));
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(R) : "v"(A), "v"(B));
        } else if (FORM == 11) {                                // the selection on SRC0: low half from src0's high half: {A.hi + B.lo, A.hi + B.hi}
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(want_lo) : "v"(A[1]), "v"(B[0]));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(want_hi) : "v"(A[1]), "v"(B[1]));
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(R) : "v"(A), "v"(B));
        } else if (FORM == 12) {                                // the selection on an FMA's ADDEND (src2): {A.lo B.lo + C.hi, A.hi B.hi + C.hi}
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(want_lo) : "v"(A[0]), "v"(B[0]), "v"(Bs[1]));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(want_hi) : "v"(A[1]), "v"(B[1]), "v"(Bs[1]));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]" : "=v"(R) : "v"(A), "v"(B), "v"(Bs));
        } else if (FORM == 10) {                                // half selection INSIDE a register: SDWA reads of the high / low fp16 of a dword (412 sites)
            unsigned pk, hi16;
            asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(pk) : "v"(A[0]), "v"(B[1]));      // {f16(A.lo), f16(B.hi)}
            asm volatile("v_lshrrev_b32 %0, 16, %1" : "=v"(hi16) : "v"(pk));
            asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(want_lo) : "v"(hi16));
            asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(want_hi) : "v"(pk));                        // reads bits 15:0
            asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(R[0]) : "v"(pk));
            asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(R[1]) : "v"(pk));
        }
        asm volatile("s_nop 0");
        const bool bl = __float_as_uint(R[0]) != __float_as_uint(want_lo), bh = __float_as_uint(R[1]) != __float_as_uint(want_hi);
        if (bl || bh) {
            if (!lo_bad && !hi_bad) { s_lo = want_lo; s_hi = want_hi; s_a = A[1]; s_b = A[0]; s_c = B[1]; g_lo = R[0]; g_hi = R[1]; }
            lo_bad += bl; hi_bad += bh;
        }
        A = A * mA + cA; B = B * mB + cB; Bs = Bs * mS + cS;           // new operands every round (b identical in all three pairs)
        if (!(fabsf(A[0]) < 64.f && fabsf(A[1]) < 64.f && fabsf(B[1]) < 64.f)) {
            a = rnd(); b = rnd(); c = rnd();
            A = f32x2{b, a}; B = f32x2{b, c}; Bs = f32x2{c, b};
        }
    }
    const int grp = (threadIdx.x & 63) >> 4;
    if (lo_bad) atomicAdd(&out->lo_wrong[grp], (unsigned long long)lo_bad);
    if (hi_bad) atomicAdd(&out->hi_wrong[grp], (unsigned long long)hi_bad);
    if ((lo_bad || hi_bad) && atomicCAS(&out->have_sample, 0, 1) == 0) {
        out->sample[0] = s_a; out->sample[1] = s_b; out->sample[2] = s_c; out->sample[3] = s_lo; out->sample[4] = g_lo;
        out->sample[5] = s_hi; out->sample[6] = g_hi; out->sample[7] = (float)(threadIdx.x & 63);
    }
    if ((threadIdx.x & 63) == 0) atomicAdd(&out->execs, (unsigned long long)iters);
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 20.0;
    const int with_nb = argc > 2 ? atoi(argv[2]) : 1;
    int cus = 256; (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    hipStream_t sa, sb; (void)hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    float* sink; (void)hipMalloc(&sink, (size_t)cus * 512 * 4);
    Counts* dc; (void)hipMalloc(&dc, sizeof(Counts));
    const char* names[13] = {"cross-half op_sel", "plain (control)", "v_pk_mov_b32 cross-half + plain add",
                            "v_pk_add_f32 op_sel:[0,1] (low from src1.hi, no negation)", "v_pk_mul_f32 op_sel:[0,1]", "v_pk_fma_f32 op_sel:[0,1,0]",
                            "MIRROR v_pk_add_f32 op_sel_hi:[1,0] (high from src1.lo: broadcast)", "MIRROR v_pk_fma_f32 op_sel_hi:[1,0,1]",
                            "wave sum by v_permlane32/16_swap + v_add_f32_dpp row_ror (LayerNorm, r5) against the __shfl_xor butterfly",
                            "MIRROR v_pk_mul_f32 op_sel_hi:[0,1]",
                            "v_cvt_f32_f16_sdwa src0_sel:WORD_1 / WORD_0 (half selection inside a register)",
                            "v_pk_add_f32 op_sel:[1,0] (low half from SRC0's high half)",
                            "v_pk_fma_f32 op_sel:[0,0,1] (low half's ADDEND from src2's high half)"};
    const char* forms_env = getenv("PK_FORMS");                         // e.g. PK_FORMS=3,4,5,6,7 ; default 0,1,2
    bool want[13] = {!forms_env, !forms_env, !forms_env, false, false, false, false, false, false, false, false, false, false};
    if (forms_env) for (const char* p = forms_env; *p; ++p) {                  // digits 0-9, 'a' = 10, 'b' = 11, 'c' = 12
        if (*p >= '0' && *p <= '9') want[*p - '0'] = true;
        if (*p >= 'a' && *p <= 'c') want[10 + *p - 'a'] = true;
    }
    for (int form = 0; form < 13; ++form) {
        if (!want[form]) continue;
        (void)hipMemset(dc, 0, sizeof(Counts));
        (void)hipDeviceSynchronize();
        const auto t0 = std::chrono::steady_clock::now();
        long launches = 0;
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
            if (with_nb == 1) hipLaunchKernelGGL(mfma_neighbor_kernel, dim3(cus), dim3(512), 0, sa, 20000, sink);
            if (with_nb == 2) hipLaunchKernelGGL(mfma16_neighbor_kernel, dim3(cus), dim3(512), 0, sa, 20000, sink);
            for (int k = 0; k < 8; ++k) {
                if (form == 0) hipLaunchKernelGGL(victim_kernel<0>, dim3(cus * 2), dim3(256), 0, sb, 20000, dc);
                if (form == 1) hipLaunchKernelGGL(victim_kernel<1>, dim3(cus * 2), dim3(256), 0, sb, 20000, dc);
                if (form == 2) hipLaunchKernelGGL(victim_kernel<2>, dim3(cus * 2), dim3(256), 0, sb, 20000, dc);
                if (form == 3) hipLaunchKernelGGL(victim_kernel<3>, dim3(cus * 2), dim3(256), 0, sb, 20000, dc);
                if (form == 4) hipLaunchKernelGGL(victim_kernel<4>, dim3(cus * 2), dim3(256), 0, sb, 20000, dc);
                if (form == 5) hipLaunchKernelGGL(victim_kernel<5>, dim3(cus * 2), dim3(256), 0, sb, 20000, dc);
                if (form == 6) hipLaunchKernelGGL(victim_kernel<6>, dim3(cus * 2), dim3(256), 0, sb, 20000, dc);
                if (form == 7) hipLaunchKernelGGL(victim_kernel<7>, dim3(cus * 2), dim3(256), 0, sb, 20000, dc);
                if (form == 8) hipLaunchKernelGGL(victim_kernel<8>, dim3(cus * 2), dim3(256), 0, sb, 4000, dc);
                if (form == 9) hipLaunchKernelGGL(victim_kernel<9>, dim3(cus * 2), dim3(256), 0, sb, 20000, dc);
                if (form == 10) hipLaunchKernelGGL(victim_kernel<10>, dim3(cus * 2), dim3(256), 0, sb, 20000, dc);
                if (form == 11) hipLaunchKernelGGL(victim_kernel<11>, dim3(cus * 2), dim3(256), 0, sb, 20000, dc);
                if (form == 12) hipLaunchKernelGGL(victim_kernel<12>, dim3(cus * 2), dim3(256), 0, sb, 20000, dc);
                ++launches;
            }
            (void)hipStreamSynchronize(sb);
            (void)hipStreamSynchronize(sa);
        }
        Counts h; (void)hipMemcpy(&h, dc, sizeof(h), hipMemcpyDeviceToHost);
        printf("{\"form\": \"%s\", \"neighbour\": %d, \"victim_launches\": %ld, \"wave_executions\": %llu, \"low_half_wrong_by_lane_group\": [%llu, %llu, %llu, %llu], "
               "\"high_half_wrong_by_lane_group\": [%llu, %llu, %llu, %llu]", names[form], with_nb, launches, h.execs,
               h.lo_wrong[0], h.lo_wrong[1], h.lo_wrong[2], h.lo_wrong[3], h.hi_wrong[0], h.hi_wrong[1], h.hi_wrong[2], h.hi_wrong[3]);
        if (h.have_sample)
            printf(", \"first_wrong\": {\"a\": %.9g, \"b\": %.9g, \"c\": %.9g, \"low_wanted_b_minus_c\": %.9g, \"low_got\": %.9g, \"high_wanted_a_minus_b\": %.9g, \"high_got\": %.9g, \"lane\": %d}",
                   h.sample[0], h.sample[1], h.sample[2], h.sample[3], h.sample[4], h.sample[5], h.sample[6], (int)h.sample[7]);
        printf("}\n");
        fflush(stdout);
    }
    return 0;
}

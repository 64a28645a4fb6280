// Shared device/host helpers for libcbas_mi355x (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 f16;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

static __device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
static __device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// exact-erf GELU, the reference's ACT2FN["gelu"] / nn.GELU() default
static __device__ __forceinline__ float gelu_erf(float x) {
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

// Branch-free GELU for the fp16 encoder path: erfc by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7,
// three orders below the fp16 rounding applied to the result), one v_rcp + one v_exp + ~12 VALU
// instead of the ~50-instruction divergent libm erff.  gelu(x) = x * Phi(x),
// Phi(|x|) = 1 - q/2, Phi(-|x|) = q/2, q = erfc(|x|/sqrt2).
static __device__ __forceinline__ float gelu_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    poly *= t;
    const float e = __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);
    const float hq = 0.5f * poly * e;                 // erfc(z) / 2
    return x * (x >= 0.f ? 1.0f - hq : hq);
}

// ---- MX-fp8 (OCP e4m3 elements, E8M0 scale per 32-element block) -------------------------------------------------
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
// E8M0 byte of the smallest power-of-two scale s with amax / s <= 448 (the largest e4m3 value): no element clips.
// amax = 1.f * 2^(E-127); 448 = 1.75 * 2^8  =>  byte = E - 8 (+1 when the fraction exceeds .75), clamped to [0, 253].
static __device__ __forceinline__ int mx_scale_byte(float amax) {
    const unsigned u = __float_as_uint(amax);
    int b = (int)((u >> 23) & 0xffu) - 8 + ((u & 0x7fffffu) > 0x600000u ? 1 : 0);
    return b < 0 ? 0 : (b > 253 ? 253 : b);
}
// 2^-(byte-127): what the elements are multiplied by before the e4m3 conversion
static __device__ __forceinline__ float mx_inv_scale(int byte) { return __uint_as_float((unsigned)(254 - byte) << 23); }
// four floats -> four e4m3 bytes (round to nearest even; |v| <= 448 by construction)
static __device__ __forceinline__ unsigned cvt4_e4m3(float a, float b, float c, float d) {
    int v = 0;
    v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, v, false);
    v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
    return (unsigned)v;
}

static inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

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

// Sum / max over the 64 lanes of a wave, every lane gets the result: the xor butterfly 32, 16, 8, 4, 2, 1 - the order and the
// pairs of the __shfl_xor loops below (wave_sum / wave_max), hence the same bits (r5) - without its six LDS round trips (ds_bpermute):
// v_permlane32_swap / v_permlane16_swap for the two cross-row steps (xor32_* / xor16_* below), then v_add_f32_dpp row_ror:8 / 4 /
// 2 / 1 inside a row of 16.  row_ror:o hands lane i the value of lane (i + o) mod 16; after the step before it the values are
// 2o-periodic within the row, so that lane holds what lane i ^ o holds.
template <int CTRL> static __device__ __forceinline__ float dpp_f32(float v) {
    return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xf, 0xf, false));
}
static __device__ __forceinline__ float xor16_max(float a);
static __device__ __forceinline__ float xor32_max(float a);
static __device__ __forceinline__ float xor16_add(float a);
static __device__ __forceinline__ float xor32_add(float a);
static __device__ __forceinline__ float wave_sum_dpp(float v) {
    v = xor32_add(v);
    v = xor16_add(v);
    v += dpp_f32<0x128>(v);      // row_ror:8
    v += dpp_f32<0x124>(v);      // row_ror:4
    v += dpp_f32<0x122>(v);      // row_ror:2
    v += dpp_f32<0x121>(v);      // row_ror:1
    return v;
}
static __device__ __forceinline__ float wave_max_dpp(float v) {
    v = xor32_max(v);
    v = xor16_max(v);
    v = fmaxf(v, dpp_f32<0x128>(v));
    v = fmaxf(v, dpp_f32<0x124>(v));
    v = fmaxf(v, dpp_f32<0x122>(v));
    v = fmaxf(v, dpp_f32<0x121>(v));
    return v;
}
// the __shfl_xor forms: everywhere but the encoder's LayerNorm kernels (whose waves are whole rows: EXEC is full at the call, which
// the DPP forms need - a row_ror reads its neighbour lane whether or not that lane is active)
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

// ROUND 4's CO-RESIDENCY CORRUPTION, ROOT-CAUSED IN ROUND 5 (scripts/expand_rootcause.py;
// profiles/r05_expand_rootcause.json).  The head's expand kernel returned wrong values in lanes 48-63 of a wave when waves of
// ANOTHER kernel on the same CU kept the matrix pipe busy with 32x32x16 MFMAs.  Nine single-edit variants of that kernel's
// assembly, ~78 000 launches each beside the same neighbour, every wrong row captured: the wrong value was always
// (a - b) - b where (a - b) - (b - c) was due - the LOW half of
//     v_pk_add_f32 v[8:9], v[10:11], v[8:9] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]        ; {b - c, a - b}
// computed as b - 0 in the instruction's last 16-lane pass.  Independent of what surrounds it (8 wait states before or after,
// fresh destination pair, operands from registers instead of LDS, branch-free code around it: still wrong; the same
// subtraction as scalar v_sub_f32 or as a v_pk_add_f32 WITHOUT cross-half operand selection: 0 of 651 000 launches against
// 39 of 653 000).  So one instruction FORM is banned from the library - a packed-fp32 op whose low result half reads the high
// half of a source pair (op_sel with a 1) - and cbas_amd/asmcheck.py fails the build if the compiler emits one anywhere.
// keep_scalar / add_np below are the source-level way out: an empty asm that pins one of the two scalar results the compiler
// would have paired.  scripts/probes/probe_pk_crosshalf.hip reproduces the fault outside the library in forty lines (288 wrong low
// halves, all in lanes 48-63, in 2.5e12 executions beside the MFMA loop; 0 for the plain form, 0 on an idle device; the selection on
// SRC1 fails on add, multiply and FMA alike, the selection on src0 or on the FMA's addend did not in 1.5e12 executions each - the
// ban covers every source).  (Round 4's empirical fix - branch-free erf / tanh, registers instead of an LDS read-back - worked
// because it changed which operations the vectoriser could pair; the branch-free forms stay: they are also faster.)
static __device__ __forceinline__ float keep_scalar(float r) { asm("" : "+v"(r)); return r; }
static __device__ __forceinline__ float add_np(float a, float b) { return keep_scalar(a + b); }

// BRANCH-FREE elementary functions (r4; kept for their speed and because divergent library forms invite the pairing above).
//   erf: two minimax polynomials (|z| <= 0.9277: z + z P(z^2); beyond: 1 - exp(Q(|z|)), copysign; after N. Juffa's erff),
//        both evaluated, one selected by v_cndmask; 1.5 ulp against erf in double over [-9, 9] (libm erff: 1.3 ulp)
static __device__ __forceinline__ float erf_bf(float a) {
    const float t = fabsf(a), s = a * a;
    float r = fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
    const float u = fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
    r = fmaf(r, s, u);
    r = fmaf(r, t, -1.06777877e-1f);
    r = fmaf(r, t, -6.34846687e-1f);
    r = fmaf(r, t, -1.28717512e-1f);
    r = fmaf(r, t, -t);
    const float big = 1.0f - __builtin_amdgcn_exp2f(r * 1.44269504088896341f);
    float q = fmaf(-5.96761703e-4f, s, 4.99119423e-3f);
    q = fmaf(q, s, -2.67681349e-2f);
    q = fmaf(q, s, 1.12819925e-1f);
    q = fmaf(q, s, -3.76125336e-1f);
    q = fmaf(q, s, 1.28379166e-1f);
    q = fmaf(q, a, a);
    return t > 0.927734375f ? __builtin_copysignf(big, a) : q;
}
// exact-erf GELU, the reference's ACT2FN["gelu"] / nn.GELU() default
static __device__ __forceinline__ float gelu_erf(float x) {
    return 0.5f * x * (1.0f + erf_bf(x * 0.70710678118654752440f));
}
//   tanh = sign(x) (1 - e) / (1 + e), e = exp(-2 |x|) with the product's rounding residual carried into the result
//        (absolute error ~6e-8, all the LSTM's gates need)
static __device__ __forceinline__ float tanh_bf(float x) {
    const float v = fmaxf(-2.0f * fabsf(x), -100.f);
    const float t = v * 1.44269504f;
    float r = fmaf(v, 1.44269504f, -t);
    r = fmaf(v, 1.92596299e-8f, r);
    float e = __builtin_amdgcn_exp2f(t);
    e = fmaf(e, r * 0.693147181f, e);
    return __builtin_copysignf((1.0f - e) / (1.0f + e), x);
}

// Reductions over the lanes l ^ 16 and l ^ 32 (the four lanes that hold one row of an MFMA accumulator tile) through
// v_permlane16_swap / v_permlane32_swap (gfx950): plain VALU instructions instead of the LDS round trip of ds_bpermute
// that __shfl_xor compiles to - these sit in the softmax / amax dependency chains.  The swap returns {own value, partner's
// value} in a lane-dependent order (scripts/probes/probe_permlane_swap.hip), so only commutative operations may use it;
// max and a single fp32 add are bit-identical to the shuffle forms.
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ float xor16_max(float a) {
    const u32x2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(a), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
static __device__ __forceinline__ float xor32_max(float a) {
    const u32x2_t r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(a), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
static __device__ __forceinline__ float xor16_add(float a) {
    const u32x2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(a), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
static __device__ __forceinline__ float xor32_add(float a) {
    const u32x2_t r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(a), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// Branch-free GELU for the fp16 encoder path, four values at once: erfc by Abramowitz-Stegun 7.1.26 (|abs err| <= 3.5e-7
// over the whole line, three orders below the fp16 rounding applied to the result).  The epilogue that calls this is
// VALU-bound (128 values per lane), so the form is chosen by instruction count:
//   a = |x| / sqrt2 * sqrt(log2 e) serves the rational argument, exp(-z^2) = exp2(-a^2) and the final product;
//   gelu(x) = max(x, 0) - |x| * erfc(z) / 2 needs no compare / select; the 1/2 and |x| / a sit in the coefficients;
//   everything after `a` is sign-free and written on vectors, so it is packed math (v_pk_fma/mul_f32) by construction.
// Per value: 2.75 packed + 2 plain VALU + v_rcp + v_exp (was 5 + 6 + 2 in the x * Phi(x) form with a select).
static __device__ __forceinline__ f32x4 gelu_fast4(f32x4 x) {
    const f32x4 a = __builtin_elementwise_abs(x) * 0.84932180028801904272f;
    const f32x4 d = a * 0.2727374809f + 1.0f;
    const f32x4 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1]), __builtin_amdgcn_rcpf(d[2]), __builtin_amdgcn_rcpf(d[3])};
    f32x4 poly = t * 0.6248546950f + -0.8554778804f;
    poly = poly * t + 0.8367933924f;
    poly = poly * t + -0.1674846542f;
    poly = poly * t + 0.1500194578f;
    const f32x4 s = a * a;
    const f32x4 e = {__builtin_amdgcn_exp2f(-s[0]), __builtin_amdgcn_exp2f(-s[1]), __builtin_amdgcn_exp2f(-s[2]), __builtin_amdgcn_exp2f(-s[3])};
    const f32x4 r = __builtin_elementwise_max(x, f32x4{0.f, 0.f, 0.f, 0.f});
    return r - a * (poly * t * e);
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

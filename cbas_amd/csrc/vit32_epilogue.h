// Epilogues of the fp32-schedule ViT GEMMs (precision 3 and 4) and the split-operand stores of precision 4: shared by
// vit_f32.hip (128 x 128 tiles) and gemm_f16_8ph.hip (the ping-pong kernel's split-operand form).
#pragma once
#include "kernels.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------------
// precision 4: the same fp32 schedule with every GEMM's products on the fp16 matrix pipe.  An fp32 value x (times a power
// of two) is kept as hi + lo, hi = fp16(x), lo = fp16(x - hi): 22 significant bits;
//      a w  ~  a_hi w_hi + a_hi w_lo + a_lo w_hi        on v_mfma_f32_16x16x32_f16
// (fp16 products are exact in fp32; the dropped a_lo w_lo term and the two representation residues are ~2^-22 of the
// product each; the MFMA sums a 32-product block before it rounds into the fp32 accumulator, which makes the result
// CLOSER to the reference's blocked CPU GEMM than the k-ordered fmaf chain of precision 3: measured).
//
// Split operands live in memory in the GEMM's LDS image order, at the byte size of the fp32 array they replace: the 32
// k-values of a K-tile of a row are 128 bytes = [hi: 32 x fp16 | lo: 32 x fp16], and inside each half the value of
// k = 16 h + 4 g + e (h = 0, 1; g, e = 0..3) sits at position 8 g + 4 h + e - so the 16-byte chunk g of the hi half (chunk
// 4 + g for lo) is exactly the 8 k-values lane group g feeds one MFMA, and the staging / swizzle / fragment reads are the
// fp32 kernel's own.  Producers (LayerNorm, attention, the GELU epilogue, ingest, the weight packer) write this format,
// each value split ONCE; the GEMM loop is fragment reads + 48 MFMAs per K-tile per wave, no conversions.
// ---------------------------------------------------------------------------------------------------------------------
typedef _Float16 f16x4v __attribute__((ext_vector_type(4)));
// four consecutive columns c .. c+3 (c % 4 == 0) of a row that starts at `row` (float-sized slots): hi and lo halves
// NP: keep the four low-half subtractions scalar (the weight packer's instantiation paired them into a v_pk_add_f32 with
// cross-half operand selection, the form common.h bans)
template <bool NP = false>
__device__ __forceinline__ void store_split4(float* row, int c, f32x4 v, float scale) {
    const int kk = c & 31;
    char* tile = reinterpret_cast<char*>(row + (c - kk));                       // the K-tile's 128 bytes
    const int pos = (((kk & 15) >> 2) << 3) + ((kk >> 4) << 2);                 // 8 g + 4 h
    f16x4v hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x = v[e] * scale;
        const f16 h = (f16)x;
        hi[e] = h;
        lo[e] = (f16)(NP ? keep_scalar(x - (float)h) : x - (float)h);
    }
    *reinterpret_cast<f16x4v*>(tile + pos * 2) = hi;
    *reinterpret_cast<f16x4v*>(tile + 64 + pos * 2) = lo;
}

// Attention operands of precision 4: the 64 values of one head of q, k or v (256 bytes as fp32) become
// [hi: 64 x fp16 | lo: 64 x fp16] in natural d order - 128-byte rows, the fp16 attention kernels' K / V row geometry.
// Scales (powers of two, undone exactly in the kernel): q x 16 on top of its 1/8, k x 4, v x 4; probabilities x 1024.
constexpr float ATT_QS = 16.f, ATT_KS = 4.f, ATT_VS = 4.f, ATT_PS = 1024.f;
__device__ __forceinline__ void store_head_split4(float* head, int d, f32x4 v, float scale) {
    f16x4v hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x = v[e] * scale;
        const f16 h = (f16)x;
        hi[e] = h;
        lo[e] = (f16)(x - (float)h);
    }
    char* b = reinterpret_cast<char*>(head);
    *reinterpret_cast<f16x4v*>(b + d * 2) = hi;
    *reinterpret_cast<f16x4v*>(b + 128 + d * 2) = lo;
}

// a*c + b*s with every product and the sum rounded on its own, as the reference's `(q * cos) + (rotate_half(q) * sin)`
__device__ __forceinline__ f32x4 rope_rot32(f32x4 a, f32x4 c, f32x4 b, f32x4 s) {
#pragma clang fp contract(off)
    const f32x4 t0 = a * c;
    const f32x4 t1 = b * s;
    return t0 + t1;
}

// exact-erf GELU ([tf]:356, nn.GELU() default) on four values, branch-free: erf by two minimax polynomials (|z| <= 0.9277:
// z + z P(z^2); beyond: 1 - exp(Q(|z|)), copysign; both forms after N. Juffa's single-precision erff), 1.5 ulp against
// erf in double over [-9, 9] (libm's erff: 1.3 ulp; checked on the host with the same operations).  The epilogue that
// calls this is VALU-bound (128 values per lane), so everything is written on vectors - v_pk_fma_f32 by construction, about
// 13 VALU instructions per value where the library erff takes about 30.
__device__ __forceinline__ f32x4 erf4(f32x4 a) {
    const f32x4 t = __builtin_elementwise_abs(a), s = a * a;
    const auto K = [](float c) { return f32x4{c, c, c, c}; };
    f32x4 r = __builtin_elementwise_fma(K(-1.72853470e-5f), t, K(3.83197126e-4f));
    const f32x4 u = __builtin_elementwise_fma(K(-3.88396438e-3f), t, K(2.42546219e-2f));
    r = __builtin_elementwise_fma(r, s, u);
    r = __builtin_elementwise_fma(r, t, K(-1.06777877e-1f));
    r = __builtin_elementwise_fma(r, t, K(-6.34846687e-1f));
    r = __builtin_elementwise_fma(r, t, K(-1.28717512e-1f));
    r = __builtin_elementwise_fma(r, t, -t);
    const f32x4 x2 = r * 1.44269504088896341f;
    const f32x4 e = {__builtin_amdgcn_exp2f(x2[0]), __builtin_amdgcn_exp2f(x2[1]), __builtin_amdgcn_exp2f(x2[2]), __builtin_amdgcn_exp2f(x2[3])};
    const f32x4 big = 1.0f - e;
    f32x4 q = __builtin_elementwise_fma(K(-5.96761703e-4f), s, K(4.99119423e-3f));
    q = __builtin_elementwise_fma(q, s, K(-2.67681349e-2f));
    q = __builtin_elementwise_fma(q, s, K(1.12819925e-1f));
    q = __builtin_elementwise_fma(q, s, K(-3.76125336e-1f));
    q = __builtin_elementwise_fma(q, s, K(1.28379166e-1f));
    q = __builtin_elementwise_fma(q, a, a);
    f32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = t[i] > 0.927734375f ? __builtin_copysignf(big[i], a[i]) : q[i];
    return o;
}
__device__ __forceinline__ f32x4 gelu_erf4(f32x4 x) {
    const f32x4 z = x * 0.70710678118654752440f;
    return (0.5f * x) * (1.0f + erf4(z));
}

// hi / lo halves of four values (x scale): x ~ hi + lo to 22 bits
__device__ __forceinline__ void split4(f32x4 v, float scale, f16x4v& hi, f16x4v& lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x = v[e] * scale;
        const f16 h = (f16)x;
        hi[e] = h;
        lo[e] = (f16)(x - (float)h);
    }
}

template <int EPI>
__device__ __forceinline__ void vit32_epilogue_row(const Gemm32VitParams& p, int m, int head_col0, int lane,
                                                   const f32x4 (&acc)[4]) {
#pragma clang fp contract(off)
    const int ncol = head_col0 + (lane >> 4) * 4;     // + j*16
    if (EPI == EPI_PATCH) {
        const int b = m / p.patches_per_frame;
        const int pp = m - b * p.patches_per_frame;
        const int64_t orow = (int64_t)b * p.tokens_per_frame + p.n_prefix + pp;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = ncol + j * 16;
            f32x4 v = acc[j] + *reinterpret_cast<const f32x4*>(p.bias + n);
            if (p.pos) v = v + *reinterpret_cast<const f32x4*>(p.pos + (size_t)pp * p.N + n);
            *reinterpret_cast<f32x4*>(p.out + orow * p.ldo + n) = v;
        }
    } else if (EPI == EPI_QKV) {
        const int sec = head_col0 / p.D + p.sec0;      // 0 q, 1 k, 2 v: uniform over the 64-column group
        const int t = m % p.tokens_per_frame;
        const bool rope = p.rope_cos && (sec < 2) && (t >= p.n_prefix);
        f32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = acc[j] + *reinterpret_cast<const f32x4*>(p.bias + ncol + j * 16);
        if (rope) {
            const size_t ro = (size_t)(t - p.n_prefix) * 64 + (lane >> 4) * 4;
            f32x4 o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 c = *reinterpret_cast<const f32x4*>(p.rope_cos + ro + j * 16);
                const f32x4 s = *reinterpret_cast<const f32x4*>(p.rope_sin + ro + j * 16);
                // rotate_half(x)[d] = -x[d+32] (d < 32), x[d-32] (d >= 32)
                o[j] = (j < 2) ? rope_rot32(v[j], c, -v[j + 2], s) : rope_rot32(v[j], c, v[j - 2], s);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = o[j];
        }
        const float qs = (sec == 0) ? 0.125f : 1.0f;   // head_dim^-0.5, an exact power of two: commutes with every rounding
        if (p.split) {                                  // precision 4: the attention kernel's split operands
            const float sc = sec == 0 ? 0.125f * ATT_QS : (sec == 1 ? ATT_KS : ATT_VS);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                store_head_split4(p.out + (int64_t)m * p.ldo + head_col0, j * 16 + (lane >> 4) * 4, v[j], sc);
            return;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<f32x4*>(p.out + (int64_t)m * p.ldo + ncol + j * 16) = v[j] * qs;
    } else if (EPI == EPI_RESID) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = ncol + j * 16;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + n);
            const f32x4 lv = *reinterpret_cast<const f32x4*>(p.lambda + n);
            float* xp = p.out + (int64_t)m * p.ldo + n;
            const f32x4 xv = *reinterpret_cast<const f32x4*>(xp);
            const f32x4 hsc = (acc[j] + bv) * lv;       // layer_scale(linear(.)), rounded as its own op
            *reinterpret_cast<f32x4*>(xp) = hsc + xv;   // + residual
        }
    } else {  // EPI_GELU
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = ncol + j * 16;
            const f32x4 w = acc[j] + *reinterpret_cast<const f32x4*>(p.bias + n);
            const f32x4 gv = gelu_erf4(w);
            if (p.split) store_split4(p.out + (int64_t)m * p.ldo, n, gv, p.out_scale);      // the down projection's A operand
            else *reinterpret_cast<f32x4*>(p.out + (int64_t)m * p.ldo + n) = gv;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Tile form for the ping-pong kernel (gemm_f16_8ph.hip, precision 4): the same values as vit32_epilogue_row - every
// element-wise step is taken in the accumulator layout, with the same operations - but the global traffic goes through a
// wave-private LDS scratch (8 KiB: two 16-row x 256-byte slabs, alternating; LDS is in order per wave: no barrier) so
// that a wave's 64 columns of a row leave as ONE 256-byte segment: the accumulator layout spreads a store over 16 rows x
// 32-64 bytes, which made this epilogue 38-64k cycles of a 130k-cycle tile (s_memtime stamps).
//   EPI_RESID  fp32 x read / written in 256-byte row segments, rows prefetched two slabs ahead
//   EPI_GELU   the split image of 64 columns = two K-tiles of the down projection = 256 contiguous bytes per row
//   EPI_QKV    one head of q, k or v = [hi 128 B | lo 128 B] = 256 contiguous bytes per row
// `pre` is called once after the first global loads are issued (the persistent kernel's next-tile LDS-DMA).
// ---------------------------------------------------------------------------------------------------------------------
template <int EPI, int TM, typename Pre>
__device__ __forceinline__ void vit32_epilogue_tile(const Gemm32VitParams& p, int row_base, int head_col0, int lane,
                                                    f32x4 (&acc)[TM][4], char* scratch, Pre pre, const float* rope_lds = nullptr) {
#pragma clang fp contract(off)
    const int li = lane & 15, g = lane >> 4;
    const float unscale = 1.0f / (p.a_scale * p.w_scale);     // powers of two: exact
    if (EPI == EPI_PATCH) {
        pre();
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] *= unscale;
            const int m = row_base + i * 16 + li;
            if (m < p.M) vit32_epilogue_row<EPI_PATCH>(p, m, head_col0, lane, acc[i]);
        }
        return;
    }
    // transposed side: lane (li, g) owns the 16-byte chunk li of rows it * 4 + g of a slab
    auto out_row = [&](int m) { return reinterpret_cast<char*>(p.out + (int64_t)m * p.ldo + head_col0) + li * 16; };
    if (EPI == EPI_RESID) {
        const f32x4 bvt = *reinterpret_cast<const f32x4*>(p.bias + head_col0 + li * 4);
        const f32x4 lvt = *reinterpret_cast<const f32x4*>(p.lambda + head_col0 + li * 4);
        f32x4 xs[2][4];
        auto load_x = [&](int i, f32x4 (&dst)[4]) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                int m = row_base + i * 16 + it * 4 + g;
                m = m < p.M ? m : p.M - 1;                             // clamp: never read past the valid rows
                dst[it] = *reinterpret_cast<const f32x4*>(out_row(m));
            }
        };
        load_x(0, xs[0]);
        if (TM > 1) load_x(1, xs[1]);
        pre();
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            char* sc = scratch + (i & 1) * 4096;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int q = (4 * j + g) ^ li;                        // 16-byte chunk swizzle by row
                *reinterpret_cast<f32x4*>(sc + li * 256 + q * 16) = acc[i][j] * unscale;
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int r = it * 4 + g;
                const f32x4 y = *reinterpret_cast<const f32x4*>(sc + r * 256 + ((li ^ r) << 4));
                const int m = row_base + i * 16 + r;
                const f32x4 hsc = (y + bvt) * lvt;                     // layer_scale(linear(.)), rounded as its own op
                const f32x4 xn = hsc + xs[i & 1][it];                  // + residual
                if (m < p.M) *reinterpret_cast<f32x4*>(out_row(m)) = xn;
            }
            if (i + 2 < TM) load_x(i + 2, xs[i & 1]);
            asm volatile("" ::: "memory");
        }
        return;
    }
    f32x4 bv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bv[j] = *reinterpret_cast<const f32x4*>(p.bias + head_col0 + j * 16 + g * 4);
    const int sec = EPI == EPI_QKV ? head_col0 / p.D + p.sec0 : 2;      // 0 q, 1 k, 2 v: uniform over the 64-column group
    const bool rope_sec = EPI == EPI_QKV && p.rope_cos && sec < 2;
    const float qsc = sec == 0 ? 0.125f * ATT_QS : (sec == 1 ? ATT_KS : ATT_VS);
    // token index of this lane's row in the slab at hand: one division per tile, then + 16 with a wrap
    int t_row = rope_sec ? (row_base + li) % p.tokens_per_frame : 0;
    pre();
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        char* sc = scratch + (i & 1) * 4096;
        f32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = acc[i][j] * unscale + bv[j];
        if (EPI == EPI_QKV) {
            if (rope_sec) {
                const int pr = t_row - p.n_prefix;
                const bool rot = pr >= 0;                              // prefix rows (cls, registers) are not rotated
                const int pp = pr > 0 ? pr : 0;
                f32x4 c[2], s[2];                                      // columns g*4.. of the 16-column groups 0 and 1; 2, 3 repeat them ([tf]:190)
                if (rope_lds) {
                    // the angles by axis, in LDS: group 0 depends on the patch row only, group 1 on the patch column ([tf]:96-121)
                    const int iy = (int)__umulhi((unsigned)pp, p.rope_magic), ix = pp - iy * p.rope_nw;
                    const float* ry = rope_lds + iy * 32 + g * 4;
                    const float* rx = rope_lds + (p.rope_nh + ix) * 32 + g * 4;
                    c[0] = *reinterpret_cast<const f32x4*>(ry);
                    s[0] = *reinterpret_cast<const f32x4*>(ry + 16);
                    c[1] = *reinterpret_cast<const f32x4*>(rx);
                    s[1] = *reinterpret_cast<const f32x4*>(rx + 16);
                } else {
                    const size_t ro = (size_t)pp * 64 + g * 4;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        c[j] = *reinterpret_cast<const f32x4*>(p.rope_cos + ro + j * 16);
                        s[j] = *reinterpret_cast<const f32x4*>(p.rope_sin + ro + j * 16);
                    }
                }
                f32x4 o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    o[j] = (j < 2) ? rope_rot32(v[j], c[j], -v[j + 2], s[j]) : rope_rot32(v[j], c[j - 2], v[j - 2], s[j - 2]);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = rot ? o[j] : v[j];
                t_row += 16;
                if (p.tokens_per_frame >= 16) t_row -= t_row >= p.tokens_per_frame ? p.tokens_per_frame : 0;
                else t_row %= p.tokens_per_frame;
            }
            // [hi: d = 0..63 | lo]: the 8-byte piece of (j, g) is half g & 1 of chunk 2 j + (g >> 1) (+ 8 for lo)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f16x4v hi, lo;
                split4(v[j], qsc, hi, lo);
                const int c = 2 * j + (g >> 1);
                *reinterpret_cast<f16x4v*>(sc + li * 256 + ((c ^ li) << 4) + (g & 1) * 8) = hi;
                *reinterpret_cast<f16x4v*>(sc + li * 256 + (((8 + c) ^ li) << 4) + (g & 1) * 8) = lo;
            }
        } else {  // EPI_GELU: K-tile t2 = j >> 1 of the 64 columns; chunk g = hi (k = 4 g + e | 16 + 4 g + e), chunk 4 + g = lo
            f16x4v hi[4], lo[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) split4(gelu_erf4(v[j]), p.out_scale, hi[j], lo[j]);
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                const f16x8 h8 = __builtin_shufflevector(hi[2 * t2], hi[2 * t2 + 1], 0, 1, 2, 3, 4, 5, 6, 7);
                const f16x8 l8 = __builtin_shufflevector(lo[2 * t2], lo[2 * t2 + 1], 0, 1, 2, 3, 4, 5, 6, 7);
                *reinterpret_cast<f16x8*>(sc + li * 256 + (((t2 * 8 + g) ^ li) << 4)) = h8;
                *reinterpret_cast<f16x8*>(sc + li * 256 + (((t2 * 8 + 4 + g) ^ li) << 4)) = l8;
            }
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int r = it * 4 + g;
            const f32x4 y = *reinterpret_cast<const f32x4*>(sc + r * 256 + ((li ^ r) << 4));
            const int m = row_base + i * 16 + r;
            if (m < p.M) *reinterpret_cast<f32x4*>(out_row(m)) = y;
        }
        asm volatile("" ::: "memory");
    }
}

}  // namespace

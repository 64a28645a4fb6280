// Fused GEMM epilogues shared by the fp16 GEMM kernels.  The kernels issue their MFMAs with the
// operands swapped, so a lane holds, for ONE output row m, 4 consecutive columns in each of the 4
// accumulator tiles that make up a 64-column group (= one attention head for the QKV projection):
//     v[j][e]  <->  C[m][head_col0 + j*16 + (lane>>4)*4 + e],   j = 0..3, e = 0..3
// Reference arithmetic: bias adds of nn.Linear, RoPE [tf]:238-268 (rotate_half :203-207),
// LayerScale :342-343 + residual :432-443, exact-erf GELU :356, conv bias / token scatter :82-89.
#pragma once
#include "kernels.h"

// a * c + b * s with the rounding fixed - one product rounded, then one fma - so that every kernel variant and both
// epilogue forms give the same bits whatever the compiler would have contracted
__device__ __forceinline__ f32x4 rope_rot(f32x4 a, f32x4 c, f32x4 b, f32x4 s) {
    f32x4 t;
    {
#pragma clang fp contract(off)
        t = b * s;
    }
    return __builtin_elementwise_fma(a, c, t);
}

// ---- LayerNorm fold helpers -----------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int LN_BLOCK = 256;                  // columns per statistics block = one N tile of the producer
// sum over the 16 lanes of a DPP row (lanes 16r .. 16r+15), every lane gets the total; fixed order, plain VALU
// (dpp_f32<CTRL>: common.h)
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f32<0xB1>(v);       // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E>(v);       // quad_perm [2,3,0,1]
    v += dpp_f32<0x141>(v);      // row_half_mirror: the other quad of the 8
    v += dpp_f32<0x140>(v);      // row_mirror: the other half of the 16
    return v;
}
// pooled statistics of nb equal blocks of n values each, from {sum_b, M2_b about the block mean}:
//   sum = sum_b sum_b,   M2 = sum_b M2_b + n sum_b (sum_b / n - sum / (nb n))^2          (exact; Chan et al.)
template <int NB>
__device__ __forceinline__ f32x2 ln_pool(const f32x2 (&b)[NB], int nb, float n) {
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int k = 0; k < NB; ++k)
        if (k < nb) { s += b[k][0]; q += b[k][1]; }
    const float mean = s / ((float)nb * n);
    float between = 0.f;
#pragma unroll
    for (int k = 0; k < NB; ++k)
        if (k < nb) {
            const float d = b[k][0] * (1.0f / n) - mean;
            between = fmaf(d, d, between);
        }
    return f32x2{s, fmaf(n, between, q)};
}

template <int EPI>
__device__ __forceinline__ void gemm_epilogue_row(const GemmParams& p, int m, int head_col0, int lane,
                                                  const f32x4 (&acc)[4]) {
    const int ncol = head_col0 + (lane >> 4) * 4;     // + j*16
    if (EPI == EPI_PATCH) {
        const int b = m / p.patches_per_frame;
        const int orow = b * p.tokens_per_frame + p.n_prefix + (m - b * p.patches_per_frame);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = ncol + j * 16;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + n);
            f32x4 v = acc[j] * p.in_scale + bv;
            if (p.pos)    // DINOv2: learned position embedding of patch (m - b*P), [v2]:161
                v += *reinterpret_cast<const f32x4*>(p.pos + (size_t)(m - b * p.patches_per_frame) * p.N + n);
            *reinterpret_cast<f32x4*>(p.out_f32 + (size_t)orow * p.ldo + n) = v;
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
                o[j] = (j < 2) ? rope_rot(v[j], c, v[j + 2], -s) : rope_rot(v[j], c, v[j - 2], s);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = o[j];
        }
        const float qs = (sec == 0) ? 0.125f : 1.0f;   // head_dim^-0.5, exact power of two
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 w = v[j] * qs;
            const f16x4 hv = {(f16)w[0], (f16)w[1], (f16)w[2], (f16)w[3]};
            *reinterpret_cast<f16x4*>(p.out_f16 + (size_t)m * p.ldo + ncol + j * 16) = hv;
        }
    } else if (EPI == EPI_RESID) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = ncol + j * 16;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + n);
            const f32x4 lv = *reinterpret_cast<const f32x4*>(p.lambda + n);
            float* xp = p.out_f32 + (size_t)m * p.ldo + n;
            const f32x4 xv = *reinterpret_cast<const f32x4*>(xp);
            *reinterpret_cast<f32x4*>(xp) = (acc[j] + bv) * lv + xv;
        }
    } else {  // EPI_GELU
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = ncol + j * 16;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + n);
            const f32x4 w = acc[j] + bv;
            const f32x4 gv = gelu_fast4(w);
            const f16x4 hv = {(f16)gv[0], (f16)gv[1], (f16)gv[2], (f16)gv[3]};
            *reinterpret_cast<f16x4*>(p.out_f16 + (size_t)m * p.ldo + n) = hv;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Coalesced tile epilogue.  The accumulator layout above gives each lane 4-column fragments at a
// row stride of ldo: a wave store touches 16 rows x 32-64 bytes (quarter/half cache lines) and the
// fp32 residual update reads AND writes x that way; measured with s_memtime stamps this took
// 13-17k cycles per 128x128 tile against a 20k-cycle main loop.  Here each wave bounces its 64-wide
// tile through a private LDS scratch (the staging buffers are idle after the K loop) so that every
// global access is whole 128-byte (fp16) / 256-byte (fp32) row segments.
//   scratch: 8 KiB per wave, 16-byte aligned, wave-private (in-order LDS => no barrier needed).
// ---------------------------------------------------------------------------------------------
#ifndef CBAS_EPI_PASS_SLABS
#define CBAS_EPI_PASS_SLABS 1
#endif
struct NoPrefetch { __device__ __forceinline__ void operator()() const {} };

// `pre` is called once, after the epilogue's first global loads have been issued (they would otherwise queue behind
// it): the persistent kernel issues the next tile's first LDS-DMAs there.
// EPI_RESID_LN: ln_red = this wave's slot in the workgroup's statistics image, f32x2 [rows of the wave][4 column waves]
// (entry (r, wc) at ln_red[r * 4]); the caller pools the four 64-column blocks of a row after a barrier.
// EPI_QKV_LN / EPI_GELU_LN: ln_red = (mean, rstd) of this wave's rows, f32x2 [rows of the wave], written by the caller.
template <int EPIX, int TM, typename Pre = NoPrefetch>
__device__ __forceinline__ void gemm_epilogue_tile(const GemmParams& p, int row_base, int head_col0, int lane,
                                                   const f32x4 (&acc)[TM][4], char* scratch, Pre pre = Pre(),
                                                   const float* rope_lds = nullptr, f32x2* ln_red = nullptr) {
    constexpr int EPI = epi_base(EPIX);
    constexpr bool LN = EPIX != EPI;
    const int li = lane & 15, g = lane >> 4;
    if (EPI == EPI_PATCH) {                      // small GEMM with a row scatter: keep the direct form
        pre();
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = row_base + i * 16 + li;
            if (m < p.M) gemm_epilogue_row<EPI_PATCH>(p, m, head_col0, lane, acc[i]);
        }
    } else if (EPI == EPI_RESID) {
        // The raw accumulators are transposed 16 rows at a time (two 4 KiB scratch halves, alternating);
        // in the transposed layout a lane owns ONE 4-column chunk for every row, so bias and lambda are
        // a single vector each, and x is read/written in whole 256-byte row segments.  The x rows are
        // prefetched two slabs ahead: the read-modify-write is otherwise a serial latency chain.
        const f32x4 bvt = *reinterpret_cast<const f32x4*>(p.bias + head_col0 + li * 4);
        const f32x4 lvt = *reinterpret_cast<const f32x4*>(p.lambda + head_col0 + li * 4);
        f32x4 xs[2][4];
        auto load_x = [&](int i, f32x4 (&dst)[4]) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                int m = row_base + i * 16 + it * 4 + g;
                m = m < p.M ? m : p.M - 1;                             // clamp: never read past the valid rows
                dst[it] = *reinterpret_cast<const f32x4*>(p.out_f32 + (size_t)m * p.ldo + head_col0 + li * 4);
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
                *reinterpret_cast<f32x4*>(sc + li * 256 + q * 16) = acc[i][j];
            }
            asm volatile("" ::: "memory");
            uint2 h16r[4];                                              // LN: the fp16 row segments of this slab
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int r = it * 4 + g;                              // row within the 16-row slab
                const f32x4 y = *reinterpret_cast<const f32x4*>(sc + r * 256 + ((li ^ r) << 4));
                const int m = row_base + i * 16 + r;
                const f32x4 xn = (y + bvt) * lvt + xs[i & 1][it];
                if (m < p.M) *reinterpret_cast<f32x4*>(p.out_f32 + (size_t)m * p.ldo + head_col0 + li * 4) = xn;
                if (LN) {
                    // LayerNorm fold, producer side: the statistics of this row's 64 columns - they sit in the 16 lanes of
                    // one DPP row: block sum, then M2 about the block mean (two dependent reductions: as robust as
                    // LayerNorm's own two passes) - and the fp16 copy of the row segment (the next GEMM's A operand)
                    const float bs = row16_sum((xn[0] + xn[1]) + (xn[2] + xn[3]));
                    const f32x4 d = xn - bs * (1.0f / 64.0f);
                    const float bq = row16_sum((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]));
                    if (li == 0) ln_red[(i * 16 + r) * 4] = f32x2{bs, bq};
                    const f32x4 c = __builtin_elementwise_max(__builtin_elementwise_min(xn, f32x4{65504.f, 65504.f, 65504.f, 65504.f}),
                                                              f32x4{-65504.f, -65504.f, -65504.f, -65504.f});
                    union { f16x4 h; uint2 u; } cv;
                    cv.h = f16x4{(f16)c[0], (f16)c[1], (f16)c[2], (f16)c[3]};
                    h16r[it] = cv.u;
                }
            }
            if (LN) {
                // The epilogue is store-ISSUE bound: two 8-byte stores per row pair would cost as much as two more 16-byte
                // ones.  Lanes li and li ^ 1 trade halves instead (one DPP quad swap): the even lane stores 8 columns of row
                // `it`, the odd lane 8 columns of row `it + 1` - one 16-byte store per lane and row PAIR.
                const bool odd = li & 1;
#pragma unroll
                for (int it = 0; it < 4; it += 2) {
                    const uint2 send = odd ? h16r[it] : h16r[it + 1];      // what the partner's row needs from this lane
                    uint2 recv;
                    recv.x = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send.x, 0xB1, 0xf, 0xf, true);
                    recv.y = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send.y, 0xB1, 0xf, 0xf, true);
                    const uint2 own = odd ? h16r[it + 1] : h16r[it];
                    const uint4 outv = odd ? uint4{recv.x, recv.y, own.x, own.y} : uint4{own.x, own.y, recv.x, recv.y};
                    const int m = row_base + i * 16 + (it + (odd ? 1 : 0)) * 4 + g;
                    if (m < p.M) *reinterpret_cast<uint4*>(p.x16_out + (size_t)m * p.N + head_col0 + (li & ~1) * 4) = outv;
                }
            }
            if (i + 2 < TM) load_x(i + 2, xs[i & 1]);
            asm volatile("" ::: "memory");
        }
    } else if (EPI == EPI_GELU_F8) {
        // MX-fp8 output (precision 2): quantise in the ACCUMULATOR layout, then transpose bytes.  A lane holds, for row
        // li of a 16-row tile, columns j*16 + g*4 + e (j, e = 0..3): the 32-column scale block b is j in {2b, 2b+1} over the
        // four lanes g = 0..3 with equal li, i.e. an in-lane max over 8 values and two shuffles.  The wave's 64 columns
        // are blocks (head_col0 % 128) / 32 + {0, 1} of K-tile head_col0 / 128 of the consumer (the down projection).
        f32x4 bv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[j] = *reinterpret_cast<const f32x4*>(p.bias + head_col0 + j * 16 + g * 4);
        pre();
        uint8_t* const sc_bytes = reinterpret_cast<uint8_t*>(p.out_sc) + (size_t)(head_col0 >> 7) * p.sc_ldo * 4 + ((head_col0 & 127) >> 5);
#pragma unroll
        for (int half = 0; half < (TM + 3) / 4; ++half) {              // up to 64 rows per pass: 64 x 64 bytes of scratch
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int i = half * 4 + ii;
                if (i >= TM) break;
                f32x4 v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 w = acc[i][j] + bv[j];
                    v[j] = gelu_fast4(w);
                }
                int sb[2];
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    float a = 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) a = fmaxf(a, fmaxf(fabsf(v[2 * b][e]), fabsf(v[2 * b + 1][e])));
                    a = xor16_max(a);
                    a = xor32_max(a);
                    sb[b] = mx_scale_byte(a);
                }
                const int rl = ii * 16 + li;                            // row within the 64-row pass
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float inv = mx_inv_scale(sb[j >> 1]);
                    // 16-byte chunk j of the row sits at position j ^ ((rl >> 1) & 3): ds_write_b32 at most 2-way conflicted
                    *reinterpret_cast<unsigned*>(scratch + rl * 64 + ((j ^ ((rl >> 1) & 3)) << 4) + g * 4) =
                        cvt4_e4m3(v[j][0] * inv, v[j][1] * inv, v[j][2] * inv, v[j][3] * inv);
                }
                const int m = row_base + i * 16 + li;
                if (g == 0 && m < p.M) *reinterpret_cast<uint16_t*>(sc_bytes + (size_t)m * 4) = (uint16_t)(sb[0] | (sb[1] << 8));
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                if (it * 16 >= (TM - half * 4) * 16) break;             // rows of this pass (compile-time)
                const int r = it * 16 + (lane >> 2), c = lane & 3;      // 16 rows x 64 bytes per wave store
                const uint4 q = *reinterpret_cast<const uint4*>(scratch + r * 64 + ((c ^ ((r >> 1) & 3)) << 4));
                const int m = row_base + half * 64 + r;
                if (m < p.M) *reinterpret_cast<uint4*>(p.out_f8 + (size_t)m * p.ldo + head_col0 + c * 16) = q;
            }
            asm volatile("" ::: "memory");
        }
    } else {                                     // EPI_QKV / EPI_GELU: fp16 out, whole 64x(TM*16) tile at once
        f32x4 bv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[j] = *reinterpret_cast<const f32x4*>(p.bias + head_col0 + j * 16 + g * 4);
        const int sec = (EPI == EPI_QKV) ? head_col0 / p.D + p.sec0 : 2;
        const float qs = (sec == 0) ? 0.125f : 1.0f;
        // RoPE (q and k sections).  From the global [P][64] table this was 256 KB of L2 reads per 256x256 tile (every wave
        // of a row group fetches the same rows, and 8 waves x 4 KB per slab do not live in the 16 KB L1): 7-9k of the
        // epilogue's 16k cycles, bound by the ~30 B/cycle/CU L2 rate, not by latency (fetching two slabs ahead changed
        // nothing).  The angles factorise by axis ([tf]:96-121: columns 0-15 use the patch row, 16-31 the patch column),
        // so the ping-pong kernel keeps (nh + nw) x 128 bytes in LDS and reads the same numbers from there (rope_lds).
        // The table is angles.tile(2) ([tf]:190): columns d and d+32 hold the same value, so two loads per table serve
        // all four 16-column groups.  Prefix rows (cls + registers) are not rotated: they read row 0, keep their values.
        const bool rope = EPI == EPI_QKV && p.rope_cos && sec < 2;
        f32x4 rcs[2][2], rsn[2][2];
        bool rot[2] = {false, false};
        int t_row = rope ? (row_base + li) % p.tokens_per_frame : 0;   // token index of this lane's row in the slab being fetched
        auto load_rope = [&](int slot) {
            const int pr = t_row - p.n_prefix;
            rot[slot] = pr >= 0;
            const int pp = pr > 0 ? pr : 0;
            if (rope_lds) {
                const int iy = (int)__umulhi((unsigned)pp, p.rope_magic), ix = pp - iy * p.rope_nw;
                const float* ry = rope_lds + iy * 32 + g * 4;
                const float* rx = rope_lds + (p.rope_nh + ix) * 32 + g * 4;
                rcs[slot][0] = *reinterpret_cast<const f32x4*>(ry);
                rsn[slot][0] = *reinterpret_cast<const f32x4*>(ry + 16);
                rcs[slot][1] = *reinterpret_cast<const f32x4*>(rx);
                rsn[slot][1] = *reinterpret_cast<const f32x4*>(rx + 16);
            } else {
                const size_t ro = (size_t)pp * 64 + g * 4;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    rcs[slot][j] = *reinterpret_cast<const f32x4*>(p.rope_cos + ro + j * 16);
                    rsn[slot][j] = *reinterpret_cast<const f32x4*>(p.rope_sin + ro + j * 16);
                }
            }
            t_row += 16;                                                // next slab: 16 rows further
            if (p.tokens_per_frame >= 16) t_row -= t_row >= p.tokens_per_frame ? p.tokens_per_frame : 0;
            else t_row %= p.tokens_per_frame;
        };
        if (rope) {
            load_rope(0);
            if (TM > 1) load_rope(1);
        }
        // LayerNorm fold, consumer side: A was raw fp16 x and W = fp16(gamma o W).  (mean, rstd) of the tile's rows were
        // pooled from the producer's per-N-tile partials once per tile, at its top, and sit in LDS (ln_red = this wave's rows)
        f32x4 lnc[4];
        if (LN) {
#pragma unroll
            for (int j = 0; j < 4; ++j) lnc[j] = *reinterpret_cast<const f32x4*>(p.ln_colsum + head_col0 + j * 16 + g * 4);
        }
        pre();
        // PS slabs of 16 rows per pass through the scratch (two alternating regions: the LDS is in order per wave, so a
        // pass may overwrite what the pass before last has read).  PS = 1: the stores of a slab are in flight while
        // the next slab's arithmetic runs, and only the last slab's stores are exposed.
        constexpr int PS = CBAS_EPI_PASS_SLABS;
#pragma unroll
        for (int half = 0; half < (TM + PS - 1) / PS; ++half) {
            char* const sc = scratch + (half & 1) * (PS * 2048);
#pragma unroll
            for (int ii = 0; ii < PS; ++ii) {
                const int i = half * PS + ii;
                if (i >= TM) break;                                     // compile-time after unrolling
                f32x4 v[4];
                if (LN) {
                    const f32x2 st = ln_red[i * 16 + li];                  // (mean, rstd) of row row_base + i*16 + li
                    // own registers for both: broadcasting rs from the HIGH register of the loaded pair is the banned
                    // packed form (v_pk_fma_f32 ... op_sel:[0,1,0]; common.h)
                    const float mu = keep_scalar(st[0]), rs = keep_scalar(st[1]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = __builtin_elementwise_fma(acc[i][j] - lnc[j] * mu, f32x4{rs, rs, rs, rs}, bv[j]);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = acc[i][j] + bv[j];
                }
                if (EPI == EPI_QKV) {
                    if (rope) {
                        const int sl = i & 1;
                        f32x4 o[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            o[j] = (j < 2) ? rope_rot(v[j], rcs[sl][j], v[j + 2], -rsn[sl][j]) : rope_rot(v[j], rcs[sl][j - 2], v[j - 2], rsn[sl][j - 2]);
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = (rot[sl] ? o[j] : v[j]) * qs;
                        if (i + 2 < TM) load_rope(sl);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = v[j] * qs;
                    }
                } else {
                    {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = gelu_fast4(v[j]);
                    }
                }
                const int rl = ii * 16 + li;                            // row within the 64-row pass
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int q = (4 * j + g) ^ ((rl & 7) << 1);        // 8-byte granule swizzle, pairs stay adjacent
                    const f16x4 hv = {(f16)v[j][0], (f16)v[j][1], (f16)v[j][2], (f16)v[j][3]};
                    *reinterpret_cast<f16x4*>(sc + rl * 128 + q * 8) = hv;
                }
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                if (it * 8 >= PS * 16 || it * 8 >= (TM - half * PS) * 16) break;   // rows of this pass (compile-time)
                const int r = it * 8 + (lane >> 3), c = lane & 7;      // 8 rows x 128 bytes per wave store
                const f16x8 hv = *reinterpret_cast<const f16x8*>(sc + r * 128 + ((c ^ (r & 7)) << 4));
                const int m = row_base + half * (PS * 16) + r;
                {
                if (m < p.M) __builtin_nontemporal_store(hv, reinterpret_cast<f16x8*>(p.out_f16 + (size_t)m * p.ldo + head_col0 + c * 8));
                }
            }
            asm volatile("" ::: "memory");
        }
    }
}

__device__ __forceinline__ int gemm_xcd_remap(int orig, int nwg) {
    // blocks b and b+8 share an XCD (round-robin dispatch); give each XCD a contiguous range of
    // logical tile ids so neighbouring tiles (same A row-panel) hit one L2.  Bijective for any nwg.
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (orig >> 3);
}

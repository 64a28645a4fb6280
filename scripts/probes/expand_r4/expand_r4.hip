// The head's expand kernel as it stood at commit 93483c0 (round 4, before the hardening of 1f51523): the form that returned
// wrong values in lanes 48-63 of its third stream beside MFMA-heavy neighbours.  Kept as a PROBE: scripts/expand_rootcause.py
// compiles it to gfx950 assembly, makes instruction-level variants of that assembly, and runs each in place of the library's
// kernel (cbas_head_debug_expand_module, debug build only) beside cbas_debug_mfma_neighbor.  Not part of the library.
//   EXPAND_ERF:   0 = the device library's erff (divergent two-formula form), 1 = branch-free erf_bf (common.h)
//   EXPAND_NOPK:  1 = a compiler barrier between the two differences of the third stream, so they cannot pair into v_pk_add_f32
#include "kernels.h"
#ifndef EXPAND_ERF
#define EXPAND_ERF 0
#endif
#ifndef EXPAND_NOPK
#define EXPAND_NOPK 0
#endif

static __device__ __forceinline__ float gelu_probe(float x) {
#if EXPAND_ERF
    return 0.5f * x * (1.0f + erf_bf(x * 0.70710678118654752440f));
#else
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
#endif
}

extern "C" __global__ void head_expand_r4(const float* __restrict__ proj, HeadDims d, const float* __restrict__ b_bott,
                                          const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                          const float* __restrict__ b_lin1, int sliding, int64_t w0, int64_t r0,
                                          int64_t n_frames, float* __restrict__ aug, float* __restrict__ lin_logits) {
    extern __shared__ __attribute__((aligned(16))) float sbuf[];   // [T][3Bn]
    const int W3 = d.NS * d.Bn, T = d.T, half = T / 2;
    const int tid = threadIdx.x;
    const int64_t w = blockIdx.x;
    const int stream = tid / d.Bn;
    const float alpha = d.alpha;

    auto row_of = [&](int t) -> int64_t {
        if (!sliding) return w * T + t;
        int64_t f = w0 + w + t - half;
        f = f < 0 ? 0 : (f > n_frames - 1 ? n_frames - 1 : f);
        return f - r0;
    };

    // pass 1: EMA
    float s_prev = 0.f;
    for (int t = 0; t < T; ++t) {
        const float x = proj[row_of(t) * d.NPROJ + tid];
        s_prev = (t == 0) ? x : s_prev + alpha * (x - s_prev);
        sbuf[t * W3 + tid] = s_prev;
    }
    // linear branch
    if (tid < d.C) {
        float e = 0.f, accum = 0.f;
        for (int t = 0; t < d.hi; ++t) {
            const float x = proj[row_of(t) * d.NPROJ + W3 + tid];
            e = (t == 0) ? x : e + alpha * (x - e);
            if (t >= d.lo) accum += e;
        }
        lin_logits[w * d.C + tid] = accum / (float)(d.hi - d.lo) + b_lin1[tid];
    }
    // pass 2 (own column only, back to front): stream value + bias, GELU
    const float s0 = sbuf[tid], s1 = sbuf[W3 + tid], s2 = sbuf[2 * W3 + tid];
    const float bb = b_bott[tid];
    for (int t = T - 1; t >= 0; --t) {
        float a, b, c;
        if (t >= 3) { a = sbuf[t * W3 + tid]; b = sbuf[(t - 1) * W3 + tid]; c = sbuf[(t - 2) * W3 + tid]; }
        else if (t == 2) { a = s2; b = s1; c = s0; }
        else if (t == 1) { a = s1; b = s0; c = s1; }
        else { a = s0; b = s1; c = s2; }
        float v;
        if (stream == 0) v = a;
        else if (stream == 1) v = a - b;
        else {
#if EXPAND_NOPK
            float d1 = a - b;
            asm volatile("" : "+v"(d1));
            v = d1 - (b - c);
#else
            v = (a - b) - (b - c);
#endif
        }
        sbuf[t * W3 + tid] = gelu_probe(v + bb);
    }
    __syncthreads();
    // pass 3: LayerNorm(Bn) per (t, stream) row, one wave per row
    const int lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const int per = d.Bn >> 6;
    for (int r = wave; r < T * d.NS; r += nwaves) {
        const int t = r / d.NS, st = r - t * d.NS;
        const float* yrow = sbuf + t * W3 + st * d.Bn;
        float vals[4];
        float sum = 0.f;
        for (int k = 0; k < per; ++k) { vals[k] = yrow[lane + 64 * k]; sum += vals[k]; }
        const float mean = wave_sum(sum) / (float)d.Bn;
        float q = 0.f;
        for (int k = 0; k < per; ++k) { const float dv = vals[k] - mean; q += dv * dv; }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d.Bn + 1e-5f);
        float* orow = aug + (w * T + t) * W3 + st * d.Bn;
        for (int k = 0; k < per; ++k) {
            const int cidx = st * d.Bn + lane + 64 * k;
            orow[lane + 64 * k] = (vals[k] - mean) * rstd * ln_w[cidx] + ln_b[cidx];
        }
    }
}

// Classifier-head kernels (fp32) for gfx950: the per-window part of ClassifierLSTMDeltas.forward
// (reference backend/classifier_head.py:102-172).  The big matrix products run in gemm_f32.hip;
// these kernels do what is sequential or per-window:
//
//   head_expand   :102-117 EMA / delta / acceleration (on the per-frame PROJECTED rows: the three
//                 bottleneck Linears and lin1 commute with EMA and with the zero-sum delta
//                 stencils), :155-160 bias + GELU + LayerNorm, :119-129 linear branch
//   head_centre   :166-167 subtract the per-window time mean
//   head_lstm     :133 recurrent half of nn.LSTM (gate order i,f,g,o): the four gates of 16 hidden
//                 units for 16 windows are ONE chain of v_mfma_f32_16x16x4_f32 per wave, W_hh
//                 fragments stay in registers across the time loop, gates + cell update are
//                 fused element-wise on the accumulator
//   head_pool     :140-148 attention pooling over the centre window, lin2; :171 gate lerp;
//                 backend/cbas.py:545-546 softmax(logits / max(1e-3, T))
#include "kernels.h"

namespace {

__global__ void f16_to_f32_kernel(const f16* __restrict__ src, float* __restrict__ dst, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const f16x4 v = reinterpret_cast<const f16x4*>(src)[i];
    reinterpret_cast<f32x4*>(dst)[i] = f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---------------------------------------------------------------------------------------------
// expand: one workgroup (NS*Bn threads) per window; NS = 3 streams (cls, delta, acc) or 2 (use_acceleration = False)
// ---------------------------------------------------------------------------------------------
__global__ void head_expand_kernel(const float* __restrict__ proj, HeadDims d, const float* __restrict__ b_bott,
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

    // passes 1 + 2, fused (r4): EMA (torch.lerp(prev, x, alpha) = prev + alpha*(x - prev), alpha < 0.5) forward in time with
    // the last three values in registers; stream value (x, dx, ddx with the reflect padding [s2, s1 | s0, s1, ...]) + bias,
    // GELU, ONE LDS write per t and no LDS read: this column's earlier form wrote the EMA, then read three values back per t -
    // the part of this kernel that returned wrong values in lanes 48-63 beside MFMA-heavy neighbours (common.h).  Selects,
    // not branches: `stream` is wave-uniform but the compiler cannot know.
    {
        const float bb = b_bott[tid];
        auto emit = [&](int t, float a, float b, float c) {     // a, b, c = s_t, s_{t-1}, s_{t-2}
            const float d1 = a - b, d2 = d1 - (b - c);
            const float v = stream == 0 ? a : (stream == 1 ? d1 : d2);
            sbuf[t * W3 + tid] = gelu_erf(v + bb);
        };
        const float s0 = proj[row_of(0) * d.NPROJ + tid];
        const float s1 = s0 + alpha * (proj[row_of(1) * d.NPROJ + tid] - s0);
        const float s2 = s1 + alpha * (proj[row_of(2) * d.NPROJ + tid] - s1);
        emit(0, s0, s1, s2);
        emit(1, s1, s0, s1);
        emit(2, s2, s1, s0);
        float e2 = s0, e1 = s1, e0 = s2;                        // s_{t-3}, s_{t-2}, s_{t-1} entering t = 3
        for (int t = 3; t < T; ++t) {
            const float x = proj[row_of(t) * d.NPROJ + tid];
            const float e = e0 + alpha * (x - e0);
            emit(t, e, e0, e1);
            e2 = e1; e1 = e0; e0 = e;
        }
        (void)e2;
    }
    // linear branch: mean over the centre window of EMA(lin1 projection) + bias
    if (tid < d.C) {
        float e = 0.f, accum = 0.f;
        for (int t = 0; t < d.hi; ++t) {
            const float x = proj[row_of(t) * d.NPROJ + W3 + tid];
            e = (t == 0) ? x : e + alpha * (x - e);
            if (t >= d.lo) accum += e;
        }
        lin_logits[w * d.C + tid] = accum / (float)(d.hi - d.lo) + b_lin1[tid];
    }
    __syncthreads();
    // pass 3: LayerNorm(Bn) per (t, stream) row, one wave per row
    const int lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const int per = d.Bn >> 6;     // values per lane (Bn multiple of 64)
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

// ---------------------------------------------------------------------------------------------
__global__ void head_centre_kernel(float* __restrict__ xl, int T, int L0) {
    const int64_t w = blockIdx.x;
    for (int c = threadIdx.x; c < L0; c += blockDim.x) {
        float* p = xl + w * T * L0 + c;
        float s = 0.f;
        for (int t = 0; t < T; ++t) s += p[(int64_t)t * L0];
        const float m = s / (float)T;
        for (int t = 0; t < T; ++t) p[(int64_t)t * L0] -= m;
    }
}

// ---------------------------------------------------------------------------------------------
// recurrent LSTM: workgroup = 16 windows x one direction; wave v owns hidden units [16v, 16v+16)
// ---------------------------------------------------------------------------------------------
template <int H>
__global__ __launch_bounds__(H * 4) void head_lstm_kernel(const float* __restrict__ gin, const float* __restrict__ w_hh,
                                                          int T, int lo, int hi, int64_t nw, float* __restrict__ hout) {
    constexpr int KS = H / 4, HS = H + 4;
    __shared__ float hbuf[2][16][HS];
    const int dir = blockIdx.y;
    const int64_t wbase = (int64_t)blockIdx.x * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, g4 = lane >> 4;
    const int unit = wave * 16 + li;
    const float* whh = w_hh + (size_t)dir * 4 * H * H;

    float Bf[4][KS];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) Bf[g][ks] = whh[(size_t)(g * H + unit) * H + 4 * ks + g4];

    for (int i = tid; i < 2 * 16 * HS; i += blockDim.x) (&hbuf[0][0][0])[i] = 0.f;
    __syncthreads();

    int64_t wrow[4];
    bool wok[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t w = wbase + 4 * g4 + r;
        wok[r] = w < nw;
        wrow[r] = wok[r] ? w : nw - 1;
    }
    const int ld = 8 * H;
    const int gcol = dir * 4 * H + unit;
    auto load_gin = [&](int t, f32x4 (&dst)[4]) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[g][r] = gin[(wrow[r] * T + t) * ld + gcol + g * H];
    };

    const int nsteps = dir == 0 ? hi : T - lo;
    const int nc = hi - lo;
    float c[4] = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc[4], nxt[4];
    load_gin(dir == 0 ? 0 : T - 1, nxt);
    for (int s = 0; s < nsteps; ++s) {
        const int t = dir == 0 ? s : T - 1 - s;
        const int cur = s & 1;
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = nxt[g];
        if (s + 1 < nsteps) load_gin(dir == 0 ? s + 1 : T - 2 - s, nxt);   // prefetch under the MFMAs
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const float a = hbuf[cur][li][4 * ks + g4];
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, Bf[g][ks], acc[g], 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float ig = sigmoidf_(acc[0][r]);
            const float fg = sigmoidf_(acc[1][r]);
            const float gg = tanh_bf(acc[2][r]);
            const float og = sigmoidf_(acc[3][r]);
            c[r] = fg * c[r] + ig * gg;
            const float hn = og * tanh_bf(c[r]);
            hbuf[cur ^ 1][4 * g4 + r][unit] = hn;
            if (t >= lo && t < hi && wok[r])
                hout[((wbase + 4 * g4 + r) * nc + (t - lo)) * (2 * H) + dir * H + unit] = hn;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// pooling: one wave per window
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void head_pool_kernel(const float* __restrict__ hout, const float* __restrict__ lin_logits,
                                                        HeadDims d, const float* __restrict__ w_att, float b_att,
                                                        float att_temp, const float* __restrict__ w_lin2,
                                                        const float* __restrict__ b_lin2, float gate_sigmoid,
                                                        float temperature, int64_t nw, float* __restrict__ probs,
                                                        float* __restrict__ logits, float* __restrict__ latent) {
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= nw) return;
    const int H2 = 2 * d.h, nc = d.hi - d.lo, per = (H2 + 63) >> 6;   // lane holds columns lane + 64k < H2 (H2 <= 256)
    const float* hw = hout + w * nc * H2;

    float wa[4];
    for (int k = 0; k < per; ++k) wa[k] = lane + 64 * k < H2 ? w_att[lane + 64 * k] : 0.f;
    // scores (two passes: max, then exp-sum and weighted latent)
    float mx = -INFINITY;
    for (int t = 0; t < nc; ++t) {
        float p = 0.f;
        for (int k = 0; k < per; ++k) p += (lane + 64 * k < H2 ? hw[t * H2 + lane + 64 * k] : 0.f) * wa[k];
        const float sc = (wave_sum(p) + b_att) / att_temp;
        mx = fmaxf(mx, sc);
    }
    float den = 0.f, lat[4] = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < nc; ++t) {
        float p = 0.f, hv[4];
        for (int k = 0; k < per; ++k) { hv[k] = lane + 64 * k < H2 ? hw[t * H2 + lane + 64 * k] : 0.f; p += hv[k] * wa[k]; }
        const float e = expf((wave_sum(p) + b_att) / att_temp - mx);
        den += e;
        for (int k = 0; k < per; ++k) lat[k] += e * hv[k];
    }
    for (int k = 0; k < per; ++k) {
        lat[k] /= den;
        if (latent && lane + 64 * k < H2) latent[w * H2 + lane + 64 * k] = lat[k];
    }
    // lin2, gate lerp, softmax over classes (every lane ends up holding every logit via wave_sum)
    const float tdiv = fmaxf(1e-3f, temperature);
    float mine = 0.f, zmax = -INFINITY;
    for (int cidx = 0; cidx < d.C; ++cidx) {
        float p = 0.f;
        for (int k = 0; k < per; ++k) p += lane + 64 * k < H2 ? lat[k] * w_lin2[cidx * H2 + lane + 64 * k] : 0.f;
        const float lstm_logit = wave_sum(p) + b_lin2[cidx];
        const float lin = lin_logits[w * d.C + cidx];
        // torch.lerp(lin, lstm, g): g < 0.5 ? lin + g*(lstm-lin) : lstm - (lstm-lin)*(1-g)
        const float fin = gate_sigmoid < 0.5f ? lin + gate_sigmoid * (lstm_logit - lin)
                                              : lstm_logit - (lstm_logit - lin) * (1.0f - gate_sigmoid);
        const float z = fin / tdiv;
        zmax = fmaxf(zmax, z);
        if ((cidx & 63) == lane) mine = fin;      // lane c keeps logit c (C <= 64 enforced by the launcher)
    }
    if (lane < d.C) {
        if (logits) logits[w * d.C + lane] = mine;
    }
    const float e = lane < d.C ? expf(mine / tdiv - zmax) : 0.f;
    const float esum = wave_sum(e);
    if (lane < d.C && probs) probs[w * d.C + lane] = e / esum;
}

}  // namespace

#define CHECK_LAUNCH() (hipGetLastError() == hipSuccess ? 0 : -2)

int launch_f16_to_f32(const f16* src, float* dst, int64_t n, hipStream_t stream) {
    if (n % 4) return -1;
    const int64_t n4 = n / 4;
    hipLaunchKernelGGL(f16_to_f32_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, src, dst, n4);
    return CHECK_LAUNCH();
}

int launch_head_expand(const float* proj, const HeadDims& d, const float* b_bott, const float* ln_w,
                       const float* ln_b, const float* b_lin1, int64_t n_windows, int sliding, int64_t w0,
                       int64_t r0, int64_t n_frames, float* aug, float* lin_logits, hipStream_t stream) {
    const int threads = d.NS * d.Bn;
    if (d.Bn % 64 || threads > 1024 || d.Bn > 256 || d.C > threads || d.T < 3 || d.NS < 2 || d.NS > 3) return -1;
    const size_t lds = (size_t)d.T * threads * sizeof(float);
    if (lds > 160 * 1024) return -1;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&head_expand_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return -2;
        attr_set = true;
    }
    hipLaunchKernelGGL(head_expand_kernel, dim3((unsigned)n_windows), dim3(threads), lds, stream, proj, d, b_bott,
                       ln_w, ln_b, b_lin1, sliding, w0, r0, n_frames, aug, lin_logits);
    return CHECK_LAUNCH();
}

int launch_head_centre(float* xl, int64_t n_windows, int T, int L0, hipStream_t stream) {
    hipLaunchKernelGGL(head_centre_kernel, dim3((unsigned)n_windows), dim3(256), 0, stream, xl, T, L0);
    return CHECK_LAUNCH();
}

int launch_head_lstm(const float* gin, const float* w_hh, const HeadDims& d, int olo, int ohi, int64_t n_windows,
                     float* hout, hipStream_t stream) {
    const dim3 grid((unsigned)((n_windows + 15) / 16), 2);
    // one wave per 16 hidden units: any multiple of 16 up to 128 (W_hh fragments are H registers per lane)
#define CBAS_LSTM_CASE(H) case H: hipLaunchKernelGGL(head_lstm_kernel<H>, grid, dim3(4 * H), 0, stream, gin, w_hh, d.T, olo, ohi, n_windows, hout); break;
    switch (d.h) {
        CBAS_LSTM_CASE(16) CBAS_LSTM_CASE(32) CBAS_LSTM_CASE(48) CBAS_LSTM_CASE(64) CBAS_LSTM_CASE(80) CBAS_LSTM_CASE(96)
        CBAS_LSTM_CASE(112) CBAS_LSTM_CASE(128)
        default: return -1;
    }
#undef CBAS_LSTM_CASE
    return CHECK_LAUNCH();
}

int launch_head_pool(const float* hout, const float* lin_logits, const HeadDims& d, const float* w_att,
                     float b_att, float att_temp, const float* w_lin2, const float* b_lin2, float gate_sigmoid,
                     float temperature, int64_t n_windows, float* probs, float* logits, float* latent,
                     hipStream_t stream) {
    if (d.C > 64 || 2 * d.h > 256) return -1;
    hipLaunchKernelGGL(head_pool_kernel, dim3((unsigned)((n_windows + 3) / 4)), dim3(256), 0, stream, hout, lin_logits,
                       d, w_att, b_att, att_temp, w_lin2, b_lin2, gate_sigmoid, temperature, n_windows, probs, logits,
                       latent);
    return CHECK_LAUNCH();
}

// C-ABI: classifier-head handle (ClassifierLSTMDeltas + the window loop of infer_file).
// See include/cbas_mi355x.h for the contract and the reference lines each entry point replaces.
#include <math.h>
#include <string.h>
#include <new>
#include <vector>

#include "api_common.h"
#include "kernels.h"

namespace {
constexpr int64_t WCHUNK = 4096;     // most windows per pass (bounds the workspace at ~1 GB for T = 31, h = 64)
}

struct cbas_head {
    cbas_head_config cfg;
    int device;
    HeadDims d;
    // device weights
    float* wbuf = nullptr;           // one arena
    const float *w_proj, *b_bott, *ln_w, *ln_b, *b_lin1, *w_lin0, *b_lin0, *w_att, *w_lin2, *b_lin2;
    int n_layers = 1;
    const float *w_ih[4], *b_gate[4], *w_hh[4];       // per stacked LSTM layer: [8h][in], [8h], [2][4h][h]
    float b_att, gate_sigmoid, att_temp;
    // workspaces
    // workspaces: sized by the largest pass seen so far (grow-only; a 4096-window pass at T = 31, h = 64 takes ~1 GB,
    // a 64-window call a few MB), allocated by ensure_workspace on the first call that needs them
    float *rows32 = nullptr, *proj = nullptr, *aug = nullptr, *xl = nullptr, *gin = nullptr, *hout = nullptr,
          *lin_logits = nullptr, *hfull = nullptr;     // hfull: [w][T][2h], inner-layer outputs of a stacked LSTM
    int64_t win_cap = 0;             // windows per pass the per-window buffers hold
    int64_t proj_rows_cap = 0;       // rows of `proj` (explicit windows: w*T; sliding: w + T)
    int64_t rows32_cap = 0;          // rows of `rows32` (sliding mode with half-precision input only)
#if CBAS_BUILD_DEBUG
    hipModule_t expand_module = nullptr;     // cbas_head_debug_expand_module: a probe kernel run in place of head_expand_kernel
    hipFunction_t expand_fn = nullptr;
    // cbas_head_debug_expand_repeat: the expand stage launched `expand_repeat` times per pass, every launch's rows compared
    // with a reference copy on the device; differing rows are appended to `cap` ([cap_rows][2 + Bn] floats)
    int expand_repeat = 1;
    float* aug_ref = nullptr;                // [win_cap * T * NS * Bn], captured by mode 1
    int64_t aug_ref_elems = 0;
    int capture_mode = 0;                    // 1: the next pass copies its rows to aug_ref, 2: compare
    unsigned long long* dbg_counts = nullptr;   // [0] launches compared, [1] launches with a differing row, [2] differing rows, [3] rows captured
    float* cap = nullptr;
    int cap_rows = 0;
#endif
};

#if CBAS_BUILD_DEBUG
namespace {
// one wave per (window, t, stream) row of Bn floats: bitwise compare with the reference copy; a differing row is appended
// (row index, launch number, Bn values) to the capture buffer while it has room
__global__ void debug_compare_rows_kernel(const float* __restrict__ aug, const float* __restrict__ ref, int64_t n_rows, int Bn,
                                          unsigned long long* counts, float* cap, int cap_rows, unsigned launch_no) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&counts[0], 1ull);
    if (row >= n_rows) return;
    bool diff = false;
    for (int k = lane; k < Bn; k += 64)
        diff |= __float_as_uint(aug[row * Bn + k]) != __float_as_uint(ref[row * Bn + k]);
    const unsigned long long any = __ballot(diff);
    if (!any) return;
    unsigned long long slot = 0;
    if (lane == 0) {
        atomicAdd(&counts[2], 1ull);
        if (atomicAdd(&counts[4 + (launch_no & 1023)], 1ull) == 0) atomicAdd(&counts[1], 1ull);    // first differing row of this launch
        slot = atomicAdd(&counts[3], 1ull);
    }
    slot = __shfl(slot, 0, 64);
    if (slot >= (unsigned long long)cap_rows) return;
    float* o = cap + slot * (2 + Bn);
    if (lane == 0) { o[0] = (float)row; o[1] = (float)launch_no; }
    for (int k = lane; k < Bn; k += 64) o[2 + k] = aug[row * Bn + k];
}
}  // namespace
#endif

namespace {

int64_t head_weights_count(const cbas_head_config& c) {
    const int64_t I = c.in_features, C = c.out_features, Bn = c.bottleneck_dim, L0 = c.lin0_dim, h = c.lstm_hidden_size;
    int64_t lstm = 0;
    for (int l = 0; l < c.lstm_layers; ++l) lstm += 2 * (4 * h * (l == 0 ? L0 : 2 * h) + 4 * h * h + 8 * h);
    const int64_t NS = c.use_acceleration ? 3 : 2;
    return 2 + NS * (Bn * I + Bn) + NS * 2 * Bn + (L0 * NS * Bn + L0) + (C * I + C) + lstm + (2 * h + 1) + (C * 2 * h + C);
}

// One chunk of windows through expand -> lin0 -> centre -> in-proj -> recurrent -> pool.
int run_chunk(cbas_head* h, int64_t nw, int sliding, int64_t w0, int64_t r0, int64_t n_frames, float temperature,
              float* probs, float* logits, float* latent, hipStream_t st) {
    const HeadDims& d = h->d;
#if CBAS_BUILD_DEBUG
    {   // root-cause probe (scripts/expand_rootcause.py): optionally a separately built code object in place of the library's
        // kernel, optionally launched several times per pass with every launch's rows compared with a reference on the device
        HeadDims dd = d;
        const float *proj = h->proj, *b_bott = h->b_bott, *ln_w = h->ln_w, *ln_b = h->ln_b, *b_lin1 = h->b_lin1;
        float *aug = h->aug, *lin_logits = h->lin_logits;
        void* args[] = {&proj, &dd, &b_bott, &ln_w, &ln_b, &b_lin1, &sliding, &w0, &r0, &n_frames, &aug, &lin_logits};
        const unsigned threads = (unsigned)(d.NS * d.Bn);
        const size_t lds = (size_t)d.T * threads * sizeof(float);
        if (h->expand_fn && lds > 64 * 1024)
            return cbas_fail(CBAS_EINVAL, "expand probe: %zu bytes of LDS (64 KiB at most through a module launch)", lds);
        const int64_t elems = nw * d.T * d.NS * d.Bn;
        const bool compare = h->capture_mode == 2 && h->aug_ref && elems <= h->aug_ref_elems;
        const int reps = compare ? h->expand_repeat : 1;
        for (int rep = 0; rep < reps; ++rep) {
            if (h->expand_fn)
                HIP_TRY(hipModuleLaunchKernel(h->expand_fn, (unsigned)nw, 1, 1, threads, 1, 1, (unsigned)lds, st, args, nullptr));
            else
                LAUNCH_TRY(launch_head_expand(h->proj, d, h->b_bott, h->ln_w, h->ln_b, h->b_lin1, nw, sliding, w0, r0, n_frames,
                                              h->aug, h->lin_logits, st));
            if (compare) {
                static unsigned launch_no = 0;
                const int64_t n_rows = nw * d.T * d.NS;
                HIP_TRY(hipMemsetAsync(h->dbg_counts + 4 + (launch_no & 1023), 0, sizeof(unsigned long long), st));
                hipLaunchKernelGGL(debug_compare_rows_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, st, h->aug, h->aug_ref,
                                   n_rows, d.Bn, h->dbg_counts, h->cap, h->cap_rows, launch_no);
                ++launch_no;
            }
        }
        if (h->capture_mode == 1) {
            if (elems > h->aug_ref_elems) {
                HIP_TRY(hipStreamSynchronize(st));
                if (h->aug_ref) (void)hipFree(h->aug_ref);
                h->aug_ref = nullptr; h->aug_ref_elems = 0;
                HIP_TRY(hipMalloc(&h->aug_ref, (size_t)elems * sizeof(float)));
                h->aug_ref_elems = elems;
            }
            HIP_TRY(hipMemcpyAsync(h->aug_ref, h->aug, (size_t)elems * sizeof(float), hipMemcpyDeviceToDevice, st));
        }
    }
#else
    LAUNCH_TRY(launch_head_expand(h->proj, d, h->b_bott, h->ln_w, h->ln_b, h->b_lin1, nw, sliding, w0, r0, n_frames,
                                  h->aug, h->lin_logits, st));
#endif
    Gemm32Params g{};
    g.A = h->aug; g.lda = d.NS * d.Bn; g.W = h->w_lin0; g.bias = h->b_lin0; g.out = h->xl; g.ldo = d.L0;
    g.M = nw * d.T; g.N = d.L0; g.N_alloc = d.L0; g.K = d.NS * d.Bn;
    LAUNCH_TRY(launch_gemm_f32(g, 1, st));
    LAUNCH_TRY(launch_head_centre(h->xl, nw, d.T, d.L0, st));
    for (int l = 0; l < h->n_layers; ++l) {
        const bool last = l == h->n_layers - 1;
        Gemm32Params q{};                                   // W_ih x + b_ih + b_hh for both directions as one GEMM
        q.A = l == 0 ? h->xl : h->hfull; q.lda = l == 0 ? d.L0 : 2 * d.h; q.K = (int)q.lda;
        q.W = h->w_ih[l]; q.bias = h->b_gate[l]; q.out = h->gin; q.ldo = 8 * d.h;
        q.M = nw * d.T; q.N = 8 * d.h; q.N_alloc = 8 * d.h;
        LAUNCH_TRY(launch_gemm_f32(q, 0, st));
        // inner layers feed the next layer at every time step; only the last one can use the truncated ranges
        LAUNCH_TRY(launch_head_lstm(h->gin, h->w_hh[l], d, last ? d.lo : 0, last ? d.hi : d.T, nw,
                                    last ? h->hout : h->hfull, st));
    }
    LAUNCH_TRY(launch_head_pool(h->hout, h->lin_logits, d, h->w_att, h->b_att, h->att_temp, h->w_lin2, h->b_lin2,
                                h->gate_sigmoid, temperature, nw, probs, logits, latent, st));
    return CBAS_OK;
}

// Make the workspace hold one pass of `nw` windows (sliding = rows shared between windows: nw + T projected rows;
// explicit windows: nw * T).  Buffers only grow; growing synchronises the device first (earlier passes may still read
// the old buffers), which happens at most a few times in a handle's life.
int ensure_workspace(cbas_head* h, int64_t nw, bool sliding) {
    const HeadDims& d = h->d;
    const int64_t T = d.T, rows = sliding ? nw + T : nw * T;
    const bool grow_w = nw > h->win_cap, grow_p = rows > h->proj_rows_cap, grow_r = sliding && rows > h->rows32_cap;
    if (!grow_w && !grow_p && !grow_r) return CBAS_OK;
    HIP_TRY(hipDeviceSynchronize());
    auto regrow = [](float** p, int64_t elems) -> hipError_t {
        if (*p) { (void)hipFree(*p); *p = nullptr; }
        return hipMalloc(p, (size_t)elems * sizeof(float));
    };
    if (grow_w) {
        int64_t w = 2 * h->win_cap < WCHUNK ? 2 * h->win_cap : WCHUNK;      // geometric: few regrowths
        if (w < nw) w = nw;
        if (w < 256) w = 256;
        h->win_cap = 0;
        HIP_TRY(regrow(&h->aug, w * T * d.NS * d.Bn));
        HIP_TRY(regrow(&h->xl, w * T * d.L0));
        HIP_TRY(regrow(&h->gin, w * T * 8 * d.h));
        HIP_TRY(regrow(&h->hout, w * (d.hi - d.lo) * 2 * d.h));
        HIP_TRY(regrow(&h->lin_logits, w * d.C));
        if (h->n_layers > 1) HIP_TRY(regrow(&h->hfull, w * T * 2 * d.h));
        h->win_cap = w;
    }
    if (grow_p) {
        int64_t r = 2 * h->proj_rows_cap < WCHUNK * T ? 2 * h->proj_rows_cap : WCHUNK * T;
        if (r < rows) r = rows;
        h->proj_rows_cap = 0;
        HIP_TRY(regrow(&h->proj, r * d.NPROJ));
        h->proj_rows_cap = r;
    }
    if (grow_r) {
        int64_t r = 2 * h->rows32_cap < WCHUNK + T ? 2 * h->rows32_cap : WCHUNK + T;
        if (r < rows) r = rows;
        h->rows32_cap = 0;
        HIP_TRY(regrow(&h->rows32, r * d.I));
        h->rows32_cap = r;
    }
    return CBAS_OK;
}

int project(cbas_head* h, const float* rows32, int64_t n_rows, hipStream_t st) {
    const HeadDims& d = h->d;
    Gemm32Params p{};
    p.A = rows32; p.lda = d.I; p.W = h->w_proj; p.bias = nullptr; p.out = h->proj; p.ldo = d.NPROJ;
    p.M = n_rows; p.N = d.NPROJ; p.N_alloc = d.NPROJ; p.K = d.I;
    LAUNCH_TRY(launch_gemm_f32(p, 0, st));
    return CBAS_OK;
}

}  // namespace

extern "C" int64_t cbas_head_weights_count(const cbas_head_config* cfg) { return cfg ? head_weights_count(*cfg) : -1; }

extern "C" int cbas_head_get_config(const cbas_head* h, cbas_head_config* out) {
    if (!h || !out) return cbas_fail(CBAS_EINVAL, "null argument");
    *out = h->cfg;
    return CBAS_OK;
}

extern "C" void cbas_head_destroy(cbas_head* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    void* bufs[] = {h->wbuf, h->rows32, h->proj, h->aug, h->xl, h->gin, h->hout, h->lin_logits, h->hfull};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
#if CBAS_BUILD_DEBUG
    if (h->expand_module) (void)hipModuleUnload(h->expand_module);
    if (h->aug_ref) (void)hipFree(h->aug_ref);
    if (h->dbg_counts) (void)hipFree(h->dbg_counts);
    if (h->cap) (void)hipFree(h->cap);
#endif
    delete h;
}

extern "C" int cbas_head_create(const cbas_head_config* cfg, const float* weights_host, int64_t n_weights,
                                int device_id, cbas_head** out) {
    if (!cfg || !weights_host || !out) return cbas_fail(CBAS_EINVAL, "null argument");
    *out = nullptr;
    const cbas_head_config& c = *cfg;
    const int64_t I = c.in_features, C = c.out_features, Bn = c.bottleneck_dim, L0 = c.lin0_dim, hh = c.lstm_hidden_size;
    const int T = c.seq_len;
    if (I <= 0 || I % 32) return cbas_fail(CBAS_EINVAL, "in_features=%lld must be a positive multiple of 32", (long long)I);
    if (C <= 0 || C > 64) return cbas_fail(CBAS_EINVAL, "out_features=%lld outside [1,64]", (long long)C);
    const int64_t NS = c.use_acceleration ? 3 : 2;       // bottleneck streams (classifier_head.py:74-84,158-162)
    if (Bn % 64 || Bn <= 0 || Bn > 256 || (NS * Bn) % 32) return cbas_fail(CBAS_EINVAL, "bottleneck_dim=%lld unsupported", (long long)Bn);
    if (L0 % 32 || L0 <= 0) return cbas_fail(CBAS_EINVAL, "lin0_dim=%lld must be a multiple of 32", (long long)L0);
    if (hh < 16 || hh > 128 || hh % 16)
        return cbas_fail(CBAS_EINVAL, "lstm_hidden_size=%lld: multiples of 16 up to 128 are built", (long long)hh);
    if (T < 3 || T > 101) return cbas_fail(CBAS_EINVAL, "seq_len=%d outside [3,101]", T);
    if (c.lstm_layers < 1 || c.lstm_layers > 4) return cbas_fail(CBAS_EINVAL, "lstm_layers=%d outside [1,4]", c.lstm_layers);
    const int hsl = T / 2, sw = c.center_window_size;
    const int lo = hsl - sw > 0 ? hsl - sw : 0, hi = hsl + sw + 1 < T ? hsl + sw + 1 : T;
    if (lo >= hi) return cbas_fail(CBAS_EINVAL, "empty centre window (seq_len=%d, center_window_size=%d)", T, sw);
    if (n_weights != head_weights_count(c))
        return cbas_fail(CBAS_EINVAL, "weights blob has %lld floats, config needs %lld", (long long)n_weights,
                         (long long)head_weights_count(c));
    HIP_TRY(hipSetDevice(device_id));

    cbas_head* h = new (std::nothrow) cbas_head();
    if (!h) return cbas_fail(CBAS_ENOMEM, "out of host memory");
    h->cfg = c; h->device = device_id;
    HeadDims& d = h->d;
    d.I = (int)I; d.C = (int)C; d.T = T; d.Bn = (int)Bn; d.L0 = (int)L0; d.h = (int)hh; d.lo = lo; d.hi = hi;
    d.NS = (int)NS;
    d.NPROJ = (int)round_up(NS * Bn + C, 4);
    d.alpha = c.ema_alpha;

    // ---- repack the state dict into the layouts the kernels read --------------------------------
    const float* p = weights_host;
    const float gate = p[0], att_temp_raw = p[1];
    p += 2;
    const float* bw[3]; const float* bb[3];
    for (int s = 0; s < NS; ++s) { bw[s] = p; p += Bn * I; bb[s] = p; p += Bn; }
    const float* lnw[3]; const float* lnb[3];
    for (int s = 0; s < NS; ++s) { lnw[s] = p; p += Bn; lnb[s] = p; p += Bn; }
    const float* lin0_w = p; p += L0 * NS * Bn;
    const float* lin0_b = p; p += L0;
    const float* lin1_w = p; p += C * I;
    const float* lin1_b = p; p += C;
    const int NL = c.lstm_layers;
    h->n_layers = NL;
    const float *wih[4][2], *whh[4][2], *bih[4][2], *bhh[4][2];
    for (int l = 0; l < NL; ++l)
        for (int dir = 0; dir < 2; ++dir) {
            wih[l][dir] = p; p += 4 * hh * (l == 0 ? L0 : 2 * hh);
            whh[l][dir] = p; p += 4 * hh * hh;
            bih[l][dir] = p; p += 4 * hh;
            bhh[l][dir] = p; p += 4 * hh;
        }
    const float* att_w = p; p += 2 * hh;
    const float att_b = *p; p += 1;
    const float* lin2_w = p; p += C * 2 * hh;
    const float* lin2_b = p; p += C;
    if (p - weights_host != n_weights) { delete h; return cbas_fail(CBAS_EINVAL, "internal blob layout mismatch"); }

    std::vector<float> arena;
    auto put = [&](const float* src, int64_t n) { size_t o = arena.size(); arena.insert(arena.end(), src, src + n); return o; };
    auto pad4 = [&]() { while (arena.size() % 4) arena.push_back(0.f); };
    // w_proj rows: cls | delta | (acc) | lin1 | zero rows up to NPROJ
    const size_t o_proj = arena.size();
    for (int s = 0; s < NS; ++s) put(bw[s], Bn * I);
    put(lin1_w, C * I);
    arena.resize(o_proj + (size_t)d.NPROJ * I, 0.f);
    const size_t o_bbott = arena.size(); for (int s = 0; s < NS; ++s) put(bb[s], Bn);
    const size_t o_lnw = arena.size(); for (int s = 0; s < NS; ++s) put(lnw[s], Bn);
    const size_t o_lnb = arena.size(); for (int s = 0; s < NS; ++s) put(lnb[s], Bn);
    const size_t o_blin1 = put(lin1_b, C); pad4();
    const size_t o_wlin0 = put(lin0_w, L0 * NS * Bn);
    const size_t o_blin0 = put(lin0_b, L0); pad4();
    size_t o_wih[4], o_bgate[4], o_whh[4];
    for (int l = 0; l < NL; ++l) {
        const int64_t in = l == 0 ? L0 : 2 * hh;
        o_wih[l] = arena.size(); put(wih[l][0], 4 * hh * in); put(wih[l][1], 4 * hh * in);
        o_bgate[l] = arena.size();
        for (int dir = 0; dir < 2; ++dir)
            for (int64_t i = 0; i < 4 * hh; ++i) arena.push_back(bih[l][dir][i] + bhh[l][dir][i]);
        o_whh[l] = arena.size(); put(whh[l][0], 4 * hh * hh); put(whh[l][1], 4 * hh * hh);
    }
    const size_t o_watt = put(att_w, 2 * hh); pad4();
    const size_t o_wlin2 = put(lin2_w, C * 2 * hh); pad4();
    const size_t o_blin2 = put(lin2_b, C); pad4();

    h->gate_sigmoid = (float)(1.0 / (1.0 + exp(-(double)gate)));
    // F.softplus(x) + 1e-3 (beta=1, threshold=20), classifier_head.py:142
    h->att_temp = (att_temp_raw > 20.f ? att_temp_raw : (float)log1p(exp((double)att_temp_raw))) + 1e-3f;
    h->b_att = att_b;

#define CREATE_TRY(expr)                                                                                  \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess) {                                                                           \
            cbas_fail(CBAS_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            cbas_head_destroy(h);                                                                         \
            return _e == hipErrorOutOfMemory ? CBAS_ENOMEM : CBAS_EHIP;                                   \
        }                                                                                                 \
    } while (0)
    CREATE_TRY(hipMalloc(&h->wbuf, arena.size() * sizeof(float)));
    CREATE_TRY(hipMemcpy(h->wbuf, arena.data(), arena.size() * sizeof(float), hipMemcpyHostToDevice));
    h->w_proj = h->wbuf + o_proj; h->b_bott = h->wbuf + o_bbott; h->ln_w = h->wbuf + o_lnw; h->ln_b = h->wbuf + o_lnb;
    h->b_lin1 = h->wbuf + o_blin1; h->w_lin0 = h->wbuf + o_wlin0; h->b_lin0 = h->wbuf + o_blin0;
    for (int l = 0; l < NL; ++l) { h->w_ih[l] = h->wbuf + o_wih[l]; h->b_gate[l] = h->wbuf + o_bgate[l]; h->w_hh[l] = h->wbuf + o_whh[l]; }
    h->w_att = h->wbuf + o_watt; h->w_lin2 = h->wbuf + o_wlin2; h->b_lin2 = h->wbuf + o_blin2;

    // activations are allocated on demand (ensure_workspace): nothing here
#undef CREATE_TRY
    *out = h;
    return CBAS_OK;
}

extern "C" int cbas_head_forward_windows(cbas_head* h, const float* x_dev, int64_t n_windows, float* logits_dev,
                                         float* latent_dev, void* stream) {
    if (!h) return cbas_fail(CBAS_EINVAL, "null head handle");
    if (!x_dev || n_windows <= 0) return cbas_fail(CBAS_EINVAL, "x_dev NULL or n_windows <= 0");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    const HeadDims& d = h->d;
    for (int64_t w0 = 0; w0 < n_windows; w0 += WCHUNK) {
        const int64_t nw = n_windows - w0 < WCHUNK ? n_windows - w0 : WCHUNK;
        int rc = ensure_workspace(h, nw, false);
        if (rc) return rc;
        rc = project(h, x_dev + w0 * d.T * d.I, nw * d.T, st);
        if (rc) return rc;
        rc = run_chunk(h, nw, 0, 0, 0, 0, 1.0f, nullptr, logits_dev ? logits_dev + w0 * d.C : nullptr,
                       latent_dev ? latent_dev + w0 * 2 * d.h : nullptr, st);
        if (rc) return rc;
    }
    return CBAS_OK;
}

// rows: the clip's CLS rows as IEEE half (what _cls.h5 holds) or float32 (a foreign `cls` dataset, which the reference
// reads with .float(): backend/cbas.py:507-508)
static int infer_range(cbas_head* h, const void* cls_dev, bool half_rows, int64_t n_frames, int64_t first, int64_t count,
                       float temperature, float* probs_dev, float* logits_dev, void* stream) {
    if (!h) return cbas_fail(CBAS_EINVAL, "null head handle");
    if (!cls_dev || n_frames <= 0) return cbas_fail(CBAS_EINVAL, "cls rows NULL or n_frames <= 0");
    if (first < 0 || count <= 0 || first + count > n_frames)
        return cbas_fail(CBAS_EINVAL, "range [%lld, %lld) outside the clip of %lld frames", (long long)first,
                         (long long)(first + count), (long long)n_frames);
    if (!probs_dev && !logits_dev) return cbas_fail(CBAS_EINVAL, "no output requested");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    const HeadDims& d = h->d;
    const int half = d.T / 2;
    for (int64_t w0 = first; w0 < first + count; w0 += WCHUNK) {
        const int64_t nw = first + count - w0 < WCHUNK ? first + count - w0 : WCHUNK;
        // rows needed: frames [w0-half, w0+nw+half) clipped to the clip (edge replicate = clamped index)
        const int64_t r0 = w0 - half > 0 ? w0 - half : 0;
        const int64_t r1 = w0 + nw + half < n_frames ? w0 + nw + half : n_frames;
        const int64_t nr = r1 - r0;
        int rc = ensure_workspace(h, nw, true);
        if (rc) return rc;
        const float* rows32 = (const float*)cls_dev + r0 * d.I;
        if (half_rows) {
            LAUNCH_TRY(launch_f16_to_f32((const f16*)cls_dev + r0 * d.I, h->rows32, nr * d.I, st));
            rows32 = h->rows32;
        }
        rc = project(h, rows32, nr, st);
        if (rc) return rc;
        const int64_t o = w0 - first;
        rc = run_chunk(h, nw, 1, w0, r0, n_frames, temperature, probs_dev ? probs_dev + o * d.C : nullptr,
                       logits_dev ? logits_dev + o * d.C : nullptr, nullptr, st);
        if (rc) return rc;
    }
    return CBAS_OK;
}

#if CBAS_BUILD_DEBUG
// bring-up: copy the first `n_floats` floats of a workspace buffer of the last pass to the host (device synchronised first).
// which: 0 rows32, 1 proj, 2 aug, 3 xl, 4 gin, 5 hout, 6 lin_logits
extern "C" int cbas_head_debug_read(cbas_head* h, int which, float* host_out, int64_t n_floats) {
    if (!h || !host_out) return cbas_fail(CBAS_EINVAL, "null argument");
    const float* src = which == 0 ? h->rows32 : which == 1 ? h->proj : which == 2 ? h->aug : which == 3 ? h->xl : which == 4 ? h->gin
                       : which == 5 ? h->hout : which == 6 ? h->lin_logits : nullptr;
    if (!src) return cbas_fail(CBAS_EINVAL, "buffer %d not available", which);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(host_out, src, (size_t)n_floats * sizeof(float), hipMemcpyDeviceToHost));
    return CBAS_OK;
}

// root-cause probe (round 5): run `kernel_name` of the code object at `hsaco_path` in place of head_expand_kernel on this handle
extern "C" int cbas_head_debug_expand_module(cbas_head* h, const char* hsaco_path, const char* kernel_name) {
    if (!h) return cbas_fail(CBAS_EINVAL, "null head handle");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    if (h->expand_module) { (void)hipModuleUnload(h->expand_module); h->expand_module = nullptr; h->expand_fn = nullptr; }
    if (!hsaco_path) return CBAS_OK;
    if (!kernel_name) return cbas_fail(CBAS_EINVAL, "kernel_name is NULL");
    HIP_TRY(hipModuleLoad(&h->expand_module, hsaco_path));
    hipError_t e = hipModuleGetFunction(&h->expand_fn, h->expand_module, kernel_name);
    if (e != hipSuccess) {
        (void)hipModuleUnload(h->expand_module); h->expand_module = nullptr; h->expand_fn = nullptr;
        return cbas_fail(CBAS_EHIP, "no kernel '%s' in %s: %s", kernel_name, hsaco_path, hipGetErrorString(e));
    }
    return CBAS_OK;
}

// amplification for the probe: mode 1 = the next pass (run it on an idle device) copies its expand output to a reference
// buffer; mode 2 = every pass launches the probe kernel `repeat` times and compares every launch's rows with that reference on
// the device (differing rows are captured); mode 0 = off.  cbas_head_debug_expand_stats: counts[0..3] = launches compared,
// launches with a differing row, differing rows, rows offered to the capture buffer; rows_out receives up to max_rows captured
// rows of 2 + Bn floats (row index = (window * T + t) * NS + stream, launch number, the row); reset != 0 clears the counters.
extern "C" int cbas_head_debug_expand_repeat(cbas_head* h, int mode, int repeat) {
    if (!h) return cbas_fail(CBAS_EINVAL, "null head handle");
    if (mode < 0 || mode > 2 || repeat < 1) return cbas_fail(CBAS_EINVAL, "mode %d / repeat %d", mode, repeat);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    if (!h->dbg_counts) {
        HIP_TRY(hipMalloc(&h->dbg_counts, (4 + 1024) * sizeof(unsigned long long)));
        HIP_TRY(hipMemset(h->dbg_counts, 0, (4 + 1024) * sizeof(unsigned long long)));
        HIP_TRY(hipDeviceSynchronize());             // a null-stream fill may still be in flight: the counters are used on other streams
        h->cap_rows = 512;
        HIP_TRY(hipMalloc(&h->cap, (size_t)h->cap_rows * (2 + h->d.Bn) * sizeof(float)));
    }
    h->capture_mode = mode;
    h->expand_repeat = repeat;
    return CBAS_OK;
}

extern "C" int cbas_head_debug_expand_stats(cbas_head* h, uint64_t* counts4, float* rows_out, int max_rows, int reset) {
    if (!h || !counts4) return cbas_fail(CBAS_EINVAL, "null argument");
    if (!h->dbg_counts) return cbas_fail(CBAS_ESTATE, "cbas_head_debug_expand_repeat was not called");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(counts4, h->dbg_counts, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    int n = (int)(counts4[3] < (uint64_t)h->cap_rows ? counts4[3] : (uint64_t)h->cap_rows);
    if (n > max_rows) n = max_rows;
    if (rows_out && n > 0) HIP_TRY(hipMemcpy(rows_out, h->cap, (size_t)n * (2 + h->d.Bn) * sizeof(float), hipMemcpyDeviceToHost));
    if (reset) {
        HIP_TRY(hipMemset(h->dbg_counts, 0, (4 + 1024) * sizeof(unsigned long long)));
        HIP_TRY(hipDeviceSynchronize());
    }
    return CBAS_OK;
}
#endif

extern "C" int cbas_head_infer_f16_range(cbas_head* h, const uint16_t* cls_f16_dev, int64_t n_frames,
                                         int64_t first, int64_t count, float temperature, float* probs_dev,
                                         float* logits_dev, void* stream) {
    return infer_range(h, cls_f16_dev, true, n_frames, first, count, temperature, probs_dev, logits_dev, stream);
}

extern "C" int cbas_head_infer_f32_range(cbas_head* h, const float* cls_f32_dev, int64_t n_frames,
                                         int64_t first, int64_t count, float temperature, float* probs_dev,
                                         float* logits_dev, void* stream) {
    return infer_range(h, cls_f32_dev, false, n_frames, first, count, temperature, probs_dev, logits_dev, stream);
}

extern "C" int cbas_head_infer_f16(cbas_head* h, const uint16_t* cls_f16_dev, int64_t n_frames, float temperature,
                                   float* probs_dev, float* logits_dev, void* stream) {
    return cbas_head_infer_f16_range(h, cls_f16_dev, n_frames, 0, n_frames, temperature, probs_dev, logits_dev, stream);
}

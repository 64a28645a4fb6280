// C-ABI: classifier-head trainer (one optimisation step of train_lstm_model, backend/cbas.py:1326-1348).
// See include/cbas_mi355x.h for the contract.  Kernels: head_train_kernels.hip, GEMMs: gemm_f32.hip.
//
// Parameters, gradients and the two Adam moments live in four device arrays of one "train layout":
//   w_proj [NPROJ][I] (rows: cls | delta | acc bottleneck weights, lin1 weight, zero rows)
//   b_bott [3Bn]  ln_w [3Bn]  ln_b [3Bn]  b_lin1 [C^]            (^ = padded to a multiple of 4)
//   w_lin0 [L0][3Bn]  b_lin0 [L0]
//   per LSTM layer: w_ih [8h][in]  b_ih [8h]  b_hh [8h]  w_hh [2][4h][h]   (forward rows, then reverse)
//   w_att [2h]  b_att [4]  gate [4]  att_temp [4]  w_lin2 [C][2h]  b_lin2 [C^]
// so that every weight gradient is the direct output of one GEMM or one column sum.  Padding elements
// have zero gradients and stay zero.  `map` translates to and from the state-dict blob order.
#include <math.h>
#include <string.h>
#include <new>
#include <vector>

#include "api_common.h"
#include "kernels.h"

namespace {
struct MapEntry { int64_t blob_off, train_off, n; };
inline int64_t pad4(int64_t n) { return (n + 3) / 4 * 4; }
inline unsigned long long mix64h(unsigned long long z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
}  // namespace

struct cbas_head_trainer {
    cbas_head_config cfg;
    cbas_train_config tcfg;
    int device = 0;
    int I, C, T, Bn, L0, h, NL, lo, hi, NPROJ, F, H2, NS;
    int64_t n_blob = 0, n_train = 0;
    std::vector<MapEntry> map;
    // segment offsets (floats) in the train layout
    int64_t o_wproj, o_bbott, o_lnw, o_lnb, o_blin1, o_wlin0, o_blin0, o_wih[4], o_bih[4], o_bhh[4], o_whh[4], o_watt, o_batt,
        o_gate, o_temp, o_wlin2, o_blin2;
    float *P = nullptr, *G = nullptr, *M = nullptr, *V = nullptr;       // parameters, gradients, Adam moments
    float* cw = nullptr;                                                 // class weights or nullptr
    float *tmat = nullptr, *lin_vec = nullptr;
    int step = 0;
    int64_t Bcap = 0, Rcap = 0, Rp = 0, Bp = 0;
    // activations kept for the backward pass
    float *proj, *Y, *aug, *Z, *xl, *gin, *act[4], *cst[4], *hout[4], *attw, *scores, *latent, *lstm_logits, *final_logits,
        *lin_logits, *b_gate;
    // backward workspaces
    float *terms, *sums, *dfinal, *dlstm, *dlin, *part_pool, *part_exp, *cs_tmp, *dhA, *dhB, *dgin, *hprev, *dxl, *daug, *dproj,
        *XT, *dprojT, *augT, *dZT, *dginT, *xinT, *hprevT, *latentT, *dlogT, *Rc, *RcT, *cov, *Gm, *sq, *dlat_cov, *wlin0T, *wihT,
        *skbuf;                                                          // split-K partial tiles
    std::vector<void*> allocs;
};

namespace {

int gemm_nt(const float* A, int64_t lda, const float* W, int n_alloc, const float* bias, float* out, int64_t ldo, int64_t M,
            int N, int K, hipStream_t st) {
    Gemm32Params g{};
    g.A = A; g.lda = lda; g.W = W; g.bias = bias; g.out = out; g.ldo = ldo; g.M = M; g.N = N; g.N_alloc = n_alloc; g.K = K;
    return launch_gemm_f32(g, 0, st);
}

// Weight-gradient GEMM out[M][N] = A[M][K] W[N][K]^T with K = (windows x seq_len) and a handful of output
// tiles: split K over grid.z and add the partial tiles in a fixed order (deterministic).  K % (32*splits) == 0.
constexpr int K_PAD = 512;             // transposed activations are zero-padded to a multiple of this
int gemm_nt_longk(const float* A, const float* W, int n_alloc, float* out, int M, int N, int64_t K, float* skbuf,
                  hipStream_t st) {
    int splits = 1;
    while (splits < 16 && K / (splits * 2) >= 512 && K % (64 * splits) == 0) splits *= 2;
    if (splits == 1) return gemm_nt(A, K, W, n_alloc, nullptr, out, N, M, N, (int)K, st);
    Gemm32Params g{};
    g.A = A; g.lda = K; g.W = W; g.ldw = K; g.bias = nullptr; g.out = skbuf; g.ldo = N; g.M = M; g.N = N; g.N_alloc = n_alloc;
    g.K = (int)(K / splits); g.splits = splits; g.split_stride = (int64_t)M * N;
    int rc = launch_gemm_f32(g, 0, st);
    if (rc) return rc;
    return launch_splitk_reduce(skbuf, splits, (int64_t)M * N, (int64_t)M * N, out, st);
}

int build_layout(cbas_head_trainer* t) {
    const int64_t I = t->I, C = t->C, Bn = t->Bn, L0 = t->L0, h = t->h;
    int64_t o = 0;
    auto seg = [&](int64_t n) { const int64_t at = o; o += pad4(n); return at; };
    t->o_wproj = seg((int64_t)t->NPROJ * I);
    const int64_t NS = t->NS;
    // b_bott | ln_w | ln_b must be contiguous (one column sum over part_exp fills all three gradients): NS * Bn is a multiple of 4
    t->o_bbott = seg(NS * Bn); t->o_lnw = seg(NS * Bn); t->o_lnb = seg(NS * Bn); t->o_blin1 = seg(C);
    t->o_wlin0 = seg(L0 * NS * Bn); t->o_blin0 = seg(L0);
    for (int l = 0; l < t->NL; ++l) {
        const int64_t in = l == 0 ? L0 : 2 * h;
        t->o_wih[l] = seg(8 * h * in); t->o_bih[l] = seg(8 * h); t->o_bhh[l] = seg(8 * h); t->o_whh[l] = seg(8 * h * h);
    }
    t->o_watt = seg(2 * h); t->o_batt = seg(1); t->o_gate = seg(1); t->o_temp = seg(1);
    t->o_wlin2 = seg(C * 2 * h); t->o_blin2 = seg(C);
    t->n_train = o;
    // blob order (include/cbas_mi355x.h, cbas_head_create) -> train layout
    int64_t b = 0;
    auto put = [&](int64_t train_off, int64_t n) { t->map.push_back({b, train_off, n}); b += n; };
    put(t->o_gate, 1); put(t->o_temp, 1);
    for (int s = 0; s < NS; ++s) { put(t->o_wproj + s * Bn * I, Bn * I); put(t->o_bbott + s * Bn, Bn); }
    for (int s = 0; s < NS; ++s) { put(t->o_lnw + s * Bn, Bn); put(t->o_lnb + s * Bn, Bn); }
    put(t->o_wlin0, L0 * NS * Bn); put(t->o_blin0, L0);
    put(t->o_wproj + NS * Bn * I, C * I); put(t->o_blin1, C);
    for (int l = 0; l < t->NL; ++l) {
        const int64_t in = l == 0 ? L0 : 2 * h;
        for (int dir = 0; dir < 2; ++dir) {
            put(t->o_wih[l] + dir * 4 * h * in, 4 * h * in);
            put(t->o_whh[l] + dir * 4 * h * h, 4 * h * h);
            put(t->o_bih[l] + dir * 4 * h, 4 * h);
            put(t->o_bhh[l] + dir * 4 * h, 4 * h);
        }
    }
    put(t->o_watt, 2 * h); put(t->o_batt, 1); put(t->o_wlin2, C * 2 * h); put(t->o_blin2, C);
    t->n_blob = b;
    return 0;
}

// temporal operators of _calculate_robust_deltas (classifier_head.py:102-117) as T x T matrices, in double
void build_temporal(int T, double alpha, int lo, int hi, std::vector<float>& tmat, std::vector<float>& lin_vec) {
    std::vector<double> E((size_t)T * T, 0.0), D((size_t)T * T, 0.0), A((size_t)T * T, 0.0);
    E[0] = 1.0;
    for (int t = 1; t < T; ++t) {
        for (int s = 0; s < T; ++s) E[(size_t)t * T + s] = (1.0 - alpha) * E[(size_t)(t - 1) * T + s];
        E[(size_t)t * T + t] += alpha;
    }
    auto row = [&](std::vector<double>& m, int t) { return &m[(size_t)t * T]; };
    for (int s = 0; s < T; ++s) {
        row(D, 0)[s] = row(E, 0)[s] - row(E, 1)[s];                                     // reflect pad: s0 - s1
        row(A, 0)[s] = row(E, 0)[s] - 2.0 * row(E, 1)[s] + row(E, 2)[s];
        row(A, 1)[s] = 2.0 * (row(E, 1)[s] - row(E, 0)[s]);
        for (int t = 1; t < T; ++t) row(D, t)[s] = row(E, t)[s] - row(E, t - 1)[s];
        for (int t = 2; t < T; ++t) row(A, t)[s] = row(E, t)[s] - 2.0 * row(E, t - 1)[s] + row(E, t - 2)[s];
    }
    tmat.resize(3 * (size_t)T * T);
    for (size_t i = 0; i < (size_t)T * T; ++i) { tmat[i] = (float)E[i]; tmat[(size_t)T * T + i] = (float)D[i]; tmat[2 * (size_t)T * T + i] = (float)A[i]; }
    lin_vec.assign(T, 0.f);
    for (int s = 0; s < T; ++s) {
        double v = 0.0;
        for (int t = lo; t < hi; ++t) v += E[(size_t)t * T + s];
        lin_vec[s] = (float)(v / (double)(hi - lo));
    }
}

}  // namespace

extern "C" void cbas_head_train_destroy(cbas_head_trainer* t) {
    if (!t) return;
    (void)hipSetDevice(t->device);
    (void)hipDeviceSynchronize();
    for (void* p : t->allocs)
        if (p) (void)hipFree(p);
    delete t;
}

extern "C" int cbas_head_train_create(const cbas_head_config* cfg, const cbas_train_config* tcfg, const float* weights_host,
                                      int64_t n_weights, const float* class_weights_host, int device_id,
                                      cbas_head_trainer** out) {
    if (!cfg || !tcfg || !weights_host || !out) return cbas_fail(CBAS_EINVAL, "null argument");
    *out = nullptr;
    const cbas_head_config& c = *cfg;
    if (c.in_features <= 0 || c.in_features % 32) return cbas_fail(CBAS_EINVAL, "in_features=%d must be a positive multiple of 32", c.in_features);
    if (c.out_features <= 0 || c.out_features > 64) return cbas_fail(CBAS_EINVAL, "out_features=%d outside [1,64]", c.out_features);
    if (c.bottleneck_dim % 64 || c.bottleneck_dim <= 0 || c.bottleneck_dim > 256) return cbas_fail(CBAS_EINVAL, "bottleneck_dim=%d unsupported", c.bottleneck_dim);
    if (c.lin0_dim % 32 || c.lin0_dim <= 0) return cbas_fail(CBAS_EINVAL, "lin0_dim=%d must be a multiple of 32", c.lin0_dim);
    if (c.lstm_hidden_size < 16 || c.lstm_hidden_size > 128 || c.lstm_hidden_size % 16)
        return cbas_fail(CBAS_EINVAL, "lstm_hidden_size=%d: multiples of 16 from 16 to 128 are built", c.lstm_hidden_size);
    if (c.seq_len < 3 || c.seq_len > 101) return cbas_fail(CBAS_EINVAL, "seq_len=%d outside [3,101]", c.seq_len);
    if (c.lstm_layers < 1 || c.lstm_layers > 4) return cbas_fail(CBAS_EINVAL, "lstm_layers=%d outside [1,4]", c.lstm_layers);
    if (tcfg->max_batch < 1 || tcfg->max_batch > 65536) return cbas_fail(CBAS_EINVAL, "max_batch=%d outside [1,65536]", tcfg->max_batch);
    if (!(tcfg->lr > 0.f) || tcfg->weight_decay < 0.f || tcfg->label_smoothing < 0.f || tcfg->label_smoothing >= 1.f)
        return cbas_fail(CBAS_EINVAL, "bad hyper-parameters (lr=%g weight_decay=%g label_smoothing=%g)", tcfg->lr, tcfg->weight_decay, tcfg->label_smoothing);
    const int T = c.seq_len, hsl = T / 2, sw = c.center_window_size;
    const int lo = hsl - sw > 0 ? hsl - sw : 0, hi = hsl + sw + 1 < T ? hsl + sw + 1 : T;
    if (lo >= hi) return cbas_fail(CBAS_EINVAL, "empty centre window (seq_len=%d, center_window_size=%d)", T, sw);
    if (n_weights != cbas_head_weights_count(cfg))
        return cbas_fail(CBAS_EINVAL, "weights blob has %lld floats, config needs %lld", (long long)n_weights, (long long)cbas_head_weights_count(cfg));
    HIP_TRY(hipSetDevice(device_id));

    cbas_head_trainer* t = new (std::nothrow) cbas_head_trainer();
    if (!t) return cbas_fail(CBAS_ENOMEM, "out of host memory");
    t->cfg = c; t->tcfg = *tcfg; t->device = device_id;
    t->I = c.in_features; t->C = c.out_features; t->T = T; t->Bn = c.bottleneck_dim; t->L0 = c.lin0_dim; t->h = c.lstm_hidden_size;
    t->NS = c.use_acceleration ? 3 : 2;                        // bottleneck streams (classifier_head.py:74-84)
    t->NL = c.lstm_layers; t->lo = lo; t->hi = hi; t->F = t->NS * t->Bn; t->H2 = 2 * t->h;
    t->NPROJ = (int)round_up(t->F + t->C, 4);
    build_layout(t);
    if (t->n_blob != n_weights) { delete t; return cbas_fail(CBAS_EINVAL, "internal blob layout mismatch"); }
    if (train_expand_lds_bytes(T, t->Bn, t->NPROJ) > 160 * 1024) {
        delete t;
        return cbas_fail(CBAS_EINVAL, "seq_len=%d too long for the training kernels (window does not fit the 160 KiB LDS)", T);
    }

#define TRY_HIP(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess) {                                                                           \
            cbas_fail(CBAS_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            cbas_head_train_destroy(t);                                                                   \
            return _e == hipErrorOutOfMemory ? CBAS_ENOMEM : CBAS_EHIP;                                   \
        }                                                                                                 \
    } while (0)
    auto dalloc = [&](float** p, int64_t n) -> hipError_t {
        hipError_t e = hipMalloc((void**)p, (size_t)(n > 0 ? n : 1) * sizeof(float));
        if (e == hipSuccess) { t->allocs.push_back(*p); e = hipMemset(*p, 0, (size_t)(n > 0 ? n : 1) * sizeof(float)); }
        return e;
    };
    const int64_t B = tcfg->max_batch, R = B * T, Rp = round_up(R, K_PAD), Bp = round_up(B, 32);
    t->Bcap = B; t->Rcap = R; t->Rp = Rp; t->Bp = Bp;
    const int64_t I = t->I, C = t->C, F = t->F, L0 = t->L0, h = t->h, H2 = t->H2, NP = t->NPROJ, nc = hi - lo;
    TRY_HIP(dalloc(&t->P, t->n_train)); TRY_HIP(dalloc(&t->G, t->n_train)); TRY_HIP(dalloc(&t->M, t->n_train)); TRY_HIP(dalloc(&t->V, t->n_train));
    {   // parameters: blob -> train layout
        std::vector<float> host((size_t)t->n_train, 0.f);
        for (const MapEntry& e : t->map) memcpy(host.data() + e.train_off, weights_host + e.blob_off, (size_t)e.n * sizeof(float));
        TRY_HIP(hipMemcpy(t->P, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    if (class_weights_host) {
        TRY_HIP(dalloc(&t->cw, C));
        TRY_HIP(hipMemcpy(t->cw, class_weights_host, (size_t)C * sizeof(float), hipMemcpyHostToDevice));
    }
    {
        std::vector<float> tm, lv;
        build_temporal(T, (double)c.ema_alpha, lo, hi, tm, lv);
        TRY_HIP(dalloc(&t->tmat, (int64_t)tm.size())); TRY_HIP(dalloc(&t->lin_vec, T));
        TRY_HIP(hipMemcpy(t->tmat, tm.data(), tm.size() * sizeof(float), hipMemcpyHostToDevice));
        TRY_HIP(hipMemcpy(t->lin_vec, lv.data(), lv.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    TRY_HIP(dalloc(&t->proj, R * NP)); TRY_HIP(dalloc(&t->Y, R * F)); TRY_HIP(dalloc(&t->aug, R * F));
    TRY_HIP(dalloc(&t->Z, R * L0)); TRY_HIP(dalloc(&t->xl, R * L0)); TRY_HIP(dalloc(&t->gin, R * 8 * h));
    for (int l = 0; l < t->NL; ++l) {
        TRY_HIP(dalloc(&t->act[l], R * 8 * h)); TRY_HIP(dalloc(&t->cst[l], R * H2)); TRY_HIP(dalloc(&t->hout[l], R * H2));
    }
    TRY_HIP(dalloc(&t->attw, B * nc)); TRY_HIP(dalloc(&t->scores, B * nc)); TRY_HIP(dalloc(&t->latent, B * H2));
    TRY_HIP(dalloc(&t->lstm_logits, B * C)); TRY_HIP(dalloc(&t->final_logits, B * C)); TRY_HIP(dalloc(&t->lin_logits, B * C));
    TRY_HIP(dalloc(&t->b_gate, 8 * h));
    TRY_HIP(dalloc(&t->terms, B * 2)); TRY_HIP(dalloc(&t->sums, 8)); TRY_HIP(dalloc(&t->dfinal, B * C)); TRY_HIP(dalloc(&t->dlstm, B * C));
    TRY_HIP(dalloc(&t->dlin, B * C)); TRY_HIP(dalloc(&t->part_pool, B * (H2 + 12))); TRY_HIP(dalloc(&t->part_exp, B * 3 * F));
    const int64_t max_cols = 8 * h > 3 * F ? 8 * h : 3 * F;
    TRY_HIP(dalloc(&t->cs_tmp, COLSUM_CHUNKS * max_cols));
    TRY_HIP(dalloc(&t->dhA, R * H2)); TRY_HIP(dalloc(&t->dhB, R * H2)); TRY_HIP(dalloc(&t->dgin, R * 8 * h)); TRY_HIP(dalloc(&t->hprev, R * H2));
    TRY_HIP(dalloc(&t->dxl, R * L0)); TRY_HIP(dalloc(&t->daug, R * F)); TRY_HIP(dalloc(&t->dproj, R * NP));
    TRY_HIP(dalloc(&t->XT, I * Rp)); TRY_HIP(dalloc(&t->dprojT, NP * Rp)); TRY_HIP(dalloc(&t->augT, F * Rp)); TRY_HIP(dalloc(&t->dZT, L0 * Rp));
    TRY_HIP(dalloc(&t->dginT, 8 * h * Rp)); TRY_HIP(dalloc(&t->xinT, (L0 > H2 ? L0 : H2) * Rp)); TRY_HIP(dalloc(&t->hprevT, H2 * Rp));
    TRY_HIP(dalloc(&t->latentT, H2 * Bp)); TRY_HIP(dalloc(&t->dlogT, pad4(C) * Bp)); TRY_HIP(dalloc(&t->Rc, B * H2)); TRY_HIP(dalloc(&t->RcT, H2 * Bp));
    TRY_HIP(dalloc(&t->cov, H2 * H2)); TRY_HIP(dalloc(&t->Gm, H2 * H2)); TRY_HIP(dalloc(&t->sq, H2)); TRY_HIP(dalloc(&t->dlat_cov, B * H2));
    TRY_HIP(dalloc(&t->wlin0T, F * L0)); TRY_HIP(dalloc(&t->wihT, (L0 > H2 ? L0 : H2) * 8 * h));
    {
        int64_t mx = (int64_t)NP * I;
        if (L0 * F > mx) mx = L0 * F;
        if (8 * h * (L0 > H2 ? L0 : H2) > mx) mx = 8 * h * (L0 > H2 ? L0 : H2);
        TRY_HIP(dalloc(&t->skbuf, 16 * mx));
    }
    // dalloc zero-fills with hipMemset on the NULL stream, which may return before the fill has run, and the training step is
    // queued on the caller's stream - in CBAS a torch stream, NON-BLOCKING, hence not ordered after the null stream: beside a
    // busy encoder the fills of G / M / V landed after the first steps had written them (r5: 114 of 125 forty-step runs
    // beside encoder passes ended with other weights than the idle-device run; scripts/train_beside_encoder.py).  The handle
    // is only handed out once every fill has completed.
    TRY_HIP(hipStreamSynchronize(nullptr));
#undef TRY_HIP
    *out = t;
    return CBAS_OK;
}

extern "C" int cbas_head_train_step(cbas_head_trainer* t, const float* x_dev, const int32_t* labels_dev, int32_t n_windows,
                                    int32_t update, float* loss_host, void* stream) {
    if (!t) return cbas_fail(CBAS_EINVAL, "null trainer handle");
    if (!x_dev || !labels_dev) return cbas_fail(CBAS_EINVAL, "x_dev / labels_dev NULL");
    if (n_windows < 1 || n_windows > t->Bcap) return cbas_fail(CBAS_EINVAL, "n_windows=%d outside [1, max_batch=%lld]", n_windows, (long long)t->Bcap);
    HIP_TRY(hipSetDevice(t->device));
    hipStream_t st = (hipStream_t)stream;
    const int I = t->I, C = t->C, T = t->T, Bn = t->Bn, L0 = t->L0, h = t->h, F = t->F, H2 = t->H2, NP = t->NPROJ, NL = t->NL;
    const int64_t B = n_windows, R = B * T, Rp = round_up(R, K_PAD), Bp = round_up(B, 32);
    float* P = t->P;
    float* G = t->G;

    // dropout streams of this step (oracle/head_train_oracle.py: dropout_keep)
    const bool drop = t->tcfg.dropout != 0;
    unsigned long long key[4];
    for (int s = 0; s < 4; ++s) key[s] = mix64h(t->tcfg.seed ^ mix64h((unsigned long long)t->step * 4ull + (unsigned long long)s));
    const unsigned thr_b = drop ? (unsigned)floor(0.1 * 16777216.0) : 0u, thr_l = drop ? (unsigned)floor(0.15 * 16777216.0) : 0u;
    const float sc_b = drop ? (float)(1.0 / (1.0 - 0.1)) : 1.0f, sc_l = drop ? (float)(1.0 / (1.0 - 0.15)) : 1.0f;

    // ---------------- forward ----------------
    LAUNCH_TRY(gemm_nt(x_dev, I, P + t->o_wproj, NP, nullptr, t->proj, NP, R, NP, I, st));
    TrainExpandParams ep{};
    ep.proj = t->proj; ep.tmat = t->tmat; ep.lin_vec = t->lin_vec; ep.b_bott = P + t->o_bbott; ep.ln_w = P + t->o_lnw;
    ep.ln_b = P + t->o_lnb; ep.b_lin1 = P + t->o_blin1; ep.T = T; ep.Bn = Bn; ep.NPROJ = NP; ep.C = C; ep.NS = t->NS;
    for (int s = 0; s < 3; ++s) ep.key[s] = key[s];
    ep.thr = thr_b; ep.scale = sc_b;
    LAUNCH_TRY(launch_train_expand_fwd(ep, B, t->Y, t->aug, t->lin_logits, st));
    LAUNCH_TRY(gemm_nt(t->aug, F, P + t->o_wlin0, L0, P + t->o_blin0, t->Z, L0, R, L0, F, st));
    LAUNCH_TRY(launch_gelu_dropout(t->Z, t->xl, R * L0, key[3], thr_l, sc_l, 0, st));
    LAUNCH_TRY(launch_head_centre(t->xl, B, T, L0, st));
    for (int l = 0; l < NL; ++l) {
        const float* xin = l == 0 ? t->xl : t->hout[l - 1];
        const int in = l == 0 ? L0 : H2;
        LAUNCH_TRY(launch_add_vec(P + t->o_bih[l], P + t->o_bhh[l], t->b_gate, 8 * h, st));
        LAUNCH_TRY(gemm_nt(xin, in, P + t->o_wih[l], 8 * h, t->b_gate, t->gin, 8 * h, R, 8 * h, in, st));
        LAUNCH_TRY(launch_lstm_train_fwd(t->gin, P + t->o_whh[l], h, T, B, t->act[l], t->cst[l], t->hout[l], st));
    }
    TrainPoolParams pp{};
    pp.hout = t->hout[NL - 1]; pp.lin_logits = t->lin_logits; pp.w_att = P + t->o_watt; pp.b_att = P + t->o_batt;
    pp.att_temp = P + t->o_temp; pp.w_lin2 = P + t->o_wlin2; pp.b_lin2 = P + t->o_blin2; pp.gate = P + t->o_gate;
    pp.T = T; pp.H2 = H2; pp.C = C; pp.lo = t->lo; pp.hi = t->hi;
    LAUNCH_TRY(launch_pool_train_fwd(pp, B, t->attw, t->scores, t->latent, t->lstm_logits, t->final_logits, st));

    // ---------------- loss ----------------
    const float eps = t->tcfg.label_smoothing;
    LAUNCH_TRY(launch_ce_terms(t->final_logits, labels_dev, t->cw, B, C, eps, t->terms, st));
    LAUNCH_TRY(launch_colsum(t->terms, B, 2, 2, 1.0f, t->cs_tmp, t->sums, st));                 // sums[0..1]
    LAUNCH_TRY(launch_ce_grad(t->final_logits, labels_dev, t->cw, t->sums, B, C, eps, t->dfinal, st));
    const bool use_cov = B > 1;                                                                // cbas.py:1340
    if (use_cov) {
        LAUNCH_TRY(launch_colsum(t->latent, B, H2, H2, 1.0f, t->cs_tmp, t->sq, st));            // column sums (sq as scratch)
        LAUNCH_TRY(launch_sub_colmean(t->latent, t->sq, B, H2, t->Rc, st));
        LAUNCH_TRY(launch_transpose_pad(t->Rc, B, H2, H2, t->RcT, Bp, st));
        LAUNCH_TRY(gemm_nt(t->RcT, Bp, t->RcT, H2, nullptr, t->cov, H2, H2, H2, (int)Bp, st));
        const float inv = 1.0f / (float)(B - 1);
        LAUNCH_TRY(launch_cov_offdiag(t->cov, H2, inv, 4.0f * inv, t->Gm, t->sq, st));
        LAUNCH_TRY(launch_colsum(t->sq, H2, 1, 1, 1.0f, t->cs_tmp, t->sums + 2, st));           // sums[2] = covariance penalty
        LAUNCH_TRY(gemm_nt(t->Rc, H2, t->Gm, H2, nullptr, t->dlat_cov, H2, B, H2, H2, st));
    } else {
        HIP_TRY(hipMemsetAsync(t->sums + 2, 0, sizeof(float), st));
    }

    // ---------------- backward ----------------
    LAUNCH_TRY(launch_pool_train_bwd(pp, B, t->attw, t->scores, t->lstm_logits, t->dfinal, use_cov ? t->dlat_cov : nullptr, t->dhA,
                                     t->dlstm, t->dlin, t->part_pool, st));
    // w_att | b_att | gate | att_temp are contiguous in the layout and in part_pool's row
    LAUNCH_TRY(launch_colsum(t->part_pool, B, H2 + 12, H2 + 12, 1.0f, t->cs_tmp, G + t->o_watt, st));
    LAUNCH_TRY(launch_colsum(t->dlstm, B, C, C, 1.0f, t->cs_tmp, G + t->o_blin2, st));
    LAUNCH_TRY(launch_colsum(t->dlin, B, C, C, 1.0f, t->cs_tmp, G + t->o_blin1, st));
    LAUNCH_TRY(launch_transpose_pad(t->dlstm, B, C, C, t->dlogT, Bp, st));
    LAUNCH_TRY(launch_transpose_pad(t->latent, B, H2, H2, t->latentT, Bp, st));
    LAUNCH_TRY(gemm_nt(t->dlogT, Bp, t->latentT, H2, nullptr, G + t->o_wlin2, H2, C, H2, (int)Bp, st));

    float* dh_cur = t->dhA;
    float* dh_next = t->dhB;
    for (int l = NL - 1; l >= 0; --l) {
        const float* xin = l == 0 ? t->xl : t->hout[l - 1];
        const int in = l == 0 ? L0 : H2;
        LAUNCH_TRY(launch_lstm_train_bwd(dh_cur, t->act[l], t->cst[l], t->hout[l], P + t->o_whh[l], h, T, B, t->dgin, t->hprev, st));
        LAUNCH_TRY(launch_colsum(t->dgin, R, 8 * h, 8 * h, 1.0f, t->cs_tmp, G + t->o_bih[l], st));
        LAUNCH_TRY(launch_add_vec(G + t->o_bih[l], nullptr, G + t->o_bhh[l], 8 * h, st));
        LAUNCH_TRY(launch_transpose_pad(t->dgin, R, 8 * h, 8 * h, t->dginT, Rp, st));
        LAUNCH_TRY(launch_transpose_pad(t->hprev, R, H2, H2, t->hprevT, Rp, st));
        LAUNCH_TRY(launch_transpose_pad(xin, R, in, in, t->xinT, Rp, st));
        for (int dir = 0; dir < 2; ++dir)
            LAUNCH_TRY(gemm_nt_longk(t->dginT + (int64_t)dir * 4 * h * Rp, t->hprevT + (int64_t)dir * h * Rp, h,
                                     G + t->o_whh[l] + (int64_t)dir * 4 * h * h, 4 * h, h, Rp, t->skbuf, st));
        LAUNCH_TRY(gemm_nt_longk(t->dginT, t->xinT, in, G + t->o_wih[l], 8 * h, in, Rp, t->skbuf, st));
        LAUNCH_TRY(launch_transpose_pad(P + t->o_wih[l], 8 * h, in, in, t->wihT, 8 * h, st));   // [in][8h]
        float* dx = l == 0 ? t->dxl : dh_next;
        LAUNCH_TRY(gemm_nt(t->dgin, 8 * h, t->wihT, in, nullptr, dx, in, R, in, 8 * h, st));
        float* tmp = dh_cur; dh_cur = dh_next; dh_next = tmp;
    }
    // centring (classifier_head.py:166-167) is its own adjoint: d x = d xc - mean_t(d xc)
    LAUNCH_TRY(launch_head_centre(t->dxl, B, T, L0, st));
    LAUNCH_TRY(launch_gelu_dropout(t->Z, t->dxl, R * L0, key[3], thr_l, sc_l, 1, st));       // dxl is now dZ
    LAUNCH_TRY(launch_colsum(t->dxl, R, L0, L0, 1.0f, t->cs_tmp, G + t->o_blin0, st));
    LAUNCH_TRY(launch_transpose_pad(t->dxl, R, L0, L0, t->dZT, Rp, st));
    LAUNCH_TRY(launch_transpose_pad(t->aug, R, F, F, t->augT, Rp, st));
    LAUNCH_TRY(gemm_nt_longk(t->dZT, t->augT, F, G + t->o_wlin0, L0, F, Rp, t->skbuf, st));
    LAUNCH_TRY(launch_transpose_pad(P + t->o_wlin0, L0, F, F, t->wlin0T, L0, st));             // [F][L0]
    LAUNCH_TRY(gemm_nt(t->dxl, L0, t->wlin0T, F, nullptr, t->daug, F, R, F, L0, st));
    LAUNCH_TRY(launch_train_expand_bwd(ep, B, t->Y, t->daug, t->dlin, t->dproj, t->part_exp, st));
    // b_bott | ln_w | ln_b are contiguous in the layout and in part_exp's row
    LAUNCH_TRY(launch_colsum(t->part_exp, B, 3 * F, 3 * F, 1.0f, t->cs_tmp, G + t->o_bbott, st));
    LAUNCH_TRY(launch_transpose_pad(t->dproj, R, NP, NP, t->dprojT, Rp, st));
    LAUNCH_TRY(launch_transpose_pad(x_dev, R, I, I, t->XT, Rp, st));
    LAUNCH_TRY(gemm_nt_longk(t->dprojT, t->XT, I, G + t->o_wproj, NP, I, Rp, t->skbuf, st));

    if (update) {
        LAUNCH_TRY(launch_adam_step(P, G, t->M, t->V, t->n_train, t->tcfg.lr, t->tcfg.weight_decay, t->o_gate, t->o_gate + 1,
                                    1e-3f /* cbas.py:1307 */, t->step + 1, st));
        t->step += 1;
    }
    if (loss_host) {
        float s[3];
        HIP_TRY(hipMemcpyAsync(s, t->sums, sizeof(s), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        loss_host[1] = s[0] / s[1];
        loss_host[2] = s[2];
        loss_host[0] = loss_host[1] + loss_host[2];
    }
    return CBAS_OK;
}

extern "C" int cbas_head_train_read(cbas_head_trainer* t, int32_t what, float* blob_host, int64_t n) {
    if (!t || !blob_host) return cbas_fail(CBAS_EINVAL, "null argument");
    if (n != t->n_blob) return cbas_fail(CBAS_EINVAL, "blob has %lld floats, config needs %lld", (long long)n, (long long)t->n_blob);
    if (what != 0 && what != 1) return cbas_fail(CBAS_EINVAL, "what=%d (0 parameters, 1 gradients)", what);
    HIP_TRY(hipSetDevice(t->device));
    HIP_TRY(hipDeviceSynchronize());
    std::vector<float> host((size_t)t->n_train);
    HIP_TRY(hipMemcpy(host.data(), what == 0 ? t->P : t->G, host.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (const MapEntry& e : t->map) memcpy(blob_host + e.blob_off, host.data() + e.train_off, (size_t)e.n * sizeof(float));
    return CBAS_OK;
}

extern "C" int cbas_head_train_last_outputs(cbas_head_trainer* t, float* logits_host, float* latent_host, int32_t n_windows) {
    if (!t) return cbas_fail(CBAS_EINVAL, "null trainer handle");
    if (n_windows < 1 || n_windows > t->Bcap) return cbas_fail(CBAS_EINVAL, "n_windows=%d outside [1, max_batch]", n_windows);
    HIP_TRY(hipSetDevice(t->device));
    HIP_TRY(hipDeviceSynchronize());
    if (logits_host) HIP_TRY(hipMemcpy(logits_host, t->final_logits, (size_t)n_windows * t->C * sizeof(float), hipMemcpyDeviceToHost));
    if (latent_host) HIP_TRY(hipMemcpy(latent_host, t->latent, (size_t)n_windows * t->H2 * sizeof(float), hipMemcpyDeviceToHost));
    return CBAS_OK;
}

// C-ABI: encoder handle (DinoEncoder replacement).  See include/cbas_mi355x.h for the contract and
// the reference lines each entry point stands in for.
#include <math.h>
#include <string.h>
#include <new>
#include <vector>
#include <algorithm>

#include "api_common.h"
#include "kernels.h"
#include <chrono>

thread_local char g_cbas_err[512] = {0};

extern "C" const char* cbas_last_error(void) { return g_cbas_err; }
extern "C" int cbas_abi_version(void) { return CBAS_ABI_VERSION; }

extern "C" int cbas_device_info(int device_id, char* arch_out, int arch_cap, int32_t* n_cu, int64_t* hbm_bytes) {
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    if (arch_out && arch_cap > 0) {
        strncpy(arch_out, prop.gcnArchName, arch_cap - 1);
        arch_out[arch_cap - 1] = 0;
    }
    if (n_cu) *n_cu = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    return CBAS_OK;
}

namespace {

struct LayerW {
    const float *ln1_w, *ln1_b, *ln2_w, *ln2_b, *o_b, *ls1, *up_b, *down_b, *ls2;
    float* qkv_b;                       // [3D] = q.b | 0 | v.b
    f16 *wqkv, *wo, *wup, *wdown;       // fp16 (hi)
    f16 *wqkv_lo, *wo_lo, *wup_lo, *wdown_lo;
    // precision 2: MX-fp8 copies (e4m3 bytes, same [N][K] layout) + E8M0 block scales [K/128][N] dwords
    uint8_t *wqkv8 = nullptr, *wo8 = nullptr, *wup8 = nullptr, *wdown8 = nullptr;
    uint32_t *sqkv = nullptr, *so = nullptr, *sup = nullptr, *sdown = nullptr;
    // LayerNorm fold (precision 0): fp16(gamma o W) of the two GEMMs that consume a LayerNorm output, the column sums of
    // those rounded weights and the folded biases beta W^T + b
    f16 *wqkv_f = nullptr, *wup_f = nullptr;
    float *qkv_cs = nullptr, *qkv_bf = nullptr, *up_cs = nullptr, *up_bf = nullptr;
    // precision 3: the fp32 weights themselves ([N][K] as stored in the blob; q | k | v packed into one [3D][D] copy)
    const float *wqkv32 = nullptr, *wo32 = nullptr, *wup32 = nullptr, *wdown32 = nullptr;
    // precision 4: power-of-two scale per weight tensor that brings max |w| into [1, 2) before the fp16 split
    float sc_qkv = 1.f, sc_o = 1.f, sc_up = 1.f, sc_down = 1.f;
};

struct Slot {
    uint8_t* in_host = nullptr;   // pinned, green planes
    uint8_t* in_dev = nullptr;
    uint16_t* out16_host = nullptr;
    float* out32_host = nullptr;
    f16* out16_dev = nullptr;
    float* out32_dev = nullptr;
    hipEvent_t ev_copied = nullptr, ev_done = nullptr, ev_in = nullptr;
    int n = 0;
    bool busy = false;
    bool used = false;            // ev_copied / ev_done have been recorded at least once
    bool dev_mode = false;        // submitted by cbas_enc_submit_u8 (device in/out) rather than ..._host
};

}  // namespace

struct cbas_enc {
    cbas_enc_config cfg;
    int device;
    int D, F, L, NH, R, NP;            // NP = prefix tokens
    float* blob = nullptr;             // all fp32 parameters on device
    std::vector<LayerW> layers;
    const float *prefix, *patch_b, *norm_w, *norm_b;
    f16 *w16 = nullptr, *w16_lo = nullptr;     // all fp16 weights (hi / lo)
    uint8_t* w8 = nullptr;                     // precision 2: all MX-fp8 weights
    uint32_t* w8_sc = nullptr;                 //              and their block scales
    uint32_t *sc_h = nullptr, *sc_u = nullptr; // precision 2: block scales of the fp8 activations in h16 / u16 ([K/128][rows_cap])
    f16 *wpatch, *wpatch2, *wpatch_lo, *wpatch2_lo;
    float* w32 = nullptr;                      // precision 3 / 4: packed q|k|v weights of every layer + the (D,256) patch weight
    const float* wpatch32 = nullptr;
    float sc_patch = 1.f;                      // precision 4: scale of the patch weight (see LayerW)
    float* qkv_bias_all = nullptr;
    unsigned* nonfinite_dev = nullptr;  // frames whose CLS row came out non-finite since the last cbas_enc_check_finite
    float* prefix_dev = nullptr;        // (1+R, D): cls (+ its position embedding for DINOv2) | registers
    float* pos_tab = nullptr;           // DINOv2: (Pmax, D) position embedding interpolated to the current grid
    std::vector<float> pos_host;        // DINOv2: raw (1+G*G, D) table
    // RoPE (DINOv3) / interpolated position-embedding (DINOv2) tables, one set per patch grid seen so far.  A new
    // grid gets FRESH device buffers filled by a blocking copy, so no stream has to be drained when a queue mixes
    // resolutions; rope_cos / rope_sin / pos_tab point at the set of the batch being queued (kernel arguments are
    // captured at launch).  Only when POS_TABLES_MAX grids are cached is the oldest one recycled behind a device sync.
    struct PosTable { int nh = 0, nw = 0; float *cos = nullptr, *sin = nullptr, *fac = nullptr, *pos = nullptr; uint64_t last_use = 0; };
    static constexpr int POS_TABLES_MAX = 8;
    std::vector<PosTable> pos_tables;
    uint64_t pos_clock = 0;
    float *rope_cos = nullptr, *rope_sin = nullptr;
    float* rope_fac = nullptr;          // the same angles by axis, [nh + nw][cos(16) | sin(16)] (GemmParams::rope_fac)
    int rope_nh = 0, rope_nw = 0;
    int rope_cap = 0;
    // workspaces
    int64_t rows_cap = 0, prow_cap = 0;
    f16 *A_patch = nullptr, *h16 = nullptr, *qkv16 = nullptr, *u16 = nullptr;
    float* x = nullptr;
    // compact per-frame rows of the pruned last layer (only the CLS row is consumed: [tf]:540-541, cbas.py:677):
    // cls16 = [q | ctx | LN2 | (pad)] x [max_batch][D] fp16, then GELU(up) [max_batch][F]
    f16* cls16 = nullptr;
    bool prune_last = true;
    bool rope_in_lds = true;           // cbas_enc_debug_option("rope_lds"): q|k|v epilogue reads the by-axis RoPE table from LDS
    // LayerNorm fold: see run_blocks.  fold_ok = the folded weights exist; ln_fold = use them (debug option "ln_fold").
    // Off by default: with two batches in flight it measured +0 ... +1 % (the LayerNorm kernels already hide under the other
    // lane's GEMMs, the fold moves their work into epilogues that do not); -5.4 % of kernel time with one batch in flight.
    bool fold_ok = false, ln_fold = false;
    f16* w16_fold = nullptr;
    float* fold_vec = nullptr;
    f16* x16 = nullptr;                // [rows_cap][D] fp16 copy of the residual stream (the folded GEMMs' A operand)
    float2* lnst = nullptr;            // [4][rows_cap] per-row LayerNorm statistics by 256-column block
    int last_rows = 0;
    hipStream_t compute = nullptr, copy = nullptr;
    hipStream_t aux = nullptr;          // cbas_enc_check_finite's 4-byte copies: not the NULL stream (torch's default stream is one: a copy there would wait for the caller's own work)
    Slot slots[CBAS_ENC_SLOTS];
    int64_t slot_bytes = 0;             // pinned staging / device input bytes per slot: max_batch x H x W x 4 channels
    // Two batches in flight: the asynchronous entry points (cbas_enc_submit_u8 / ..._host) alternate two
    // compute lanes, each a full workspace + its own stream, so that one batch's partial tile rounds,
    // LayerNorm and attention run under the other batch's GEMMs (+10 % measured; outputs bit-identical).
    // lane 0 = the buffers above on `compute`; the synchronous cbas_enc_forward_* always use lane 0.
    struct Lane { f16 *A_patch, *h16, *qkv16, *u16, *cls16; float* x; uint32_t *sc_h, *sc_u; hipStream_t stream; f16* x16; float2* lnst; };
    Lane lanes[2] = {};
    int n_lanes = 1;
    uint64_t submit_count = 0;
    // ordering between the synchronous calls (lane 0 workspace on the CALLER's stream) and asynchronous batches
    // on lane 0's own stream, so that mixing the two forms on one handle never races on the workspace
    hipEvent_t lane0_async_done = nullptr, sync_done = nullptr;
    bool lane0_async_used = false, sync_used = false;
    // optional per-kernel-category timing (HIP events on the launch stream)
    bool prof_on = false;
    struct ProfRec { hipEvent_t a, b; int cat; double flops; };
    std::vector<ProfRec> prof;
    size_t prof_used = 0;
};

namespace {

int64_t weights_count(const cbas_enc_config& c) {
    const int64_t D = c.hidden_size, F = c.intermediate_size, R = c.num_register_tokens, p = c.patch_size;
    const int64_t G = c.pos_embed_grid;
    int64_t n = D + R * D + (G > 0 ? (1 + G * G) * D : 0) + D * 3 * p * p + D;
    const int64_t per = 2 * D + (D * D + D) + (D * D + D) + (D * D + D) + (D * D + D) + D + 2 * D + (F * D + F) + (D * F + D) + D;
    n += per * c.num_layers;
    n += 2 * D;
    return n;
}

// Keys cubic kernel, a = -0.5: the coefficient ATen's ANTIALIASED bicubic uses (UpSampleKernel.cpp aa_filter)
static inline double cubic_aa(double x) {
    const double a = -0.5;
    x = fabs(x);
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0;
    if (x < 2.0) return (((x - 5.0) * x + 8.0) * x - 4.0) * a;
    return 0.0;
}

// (out, in) weights of F.interpolate(mode="bicubic", align_corners=False, antialias=True) along one axis
static std::vector<float> aa_bicubic_matrix(int in_size, int out_size) {
    std::vector<float> W((size_t)out_size * in_size, 0.f);
    const double scale = (double)in_size / out_size;
    const double support = scale >= 1.0 ? 2.0 * scale : 2.0, invscale = scale >= 1.0 ? 1.0 / scale : 1.0;
    for (int i = 0; i < out_size; ++i) {
        const double center = scale * (i + 0.5);
        int xmin = (int)(center - support + 0.5); if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5); if (xmax > in_size) xmax = in_size;
        double tot = 0.0;
        for (int j = xmin; j < xmax; ++j) tot += cubic_aa((j - center + 0.5) * invscale);
        for (int j = xmin; j < xmax; ++j) W[(size_t)i * in_size + j] = (float)(cubic_aa((j - center + 0.5) * invscale) / tot);
    }
    return W;
}

// Find or create the table set of an nh x nw patch grid and point the handle at it.
int acquire_pos_table(cbas_enc* h, int nh, int nw, cbas_enc::PosTable** out) {
    const int P = nh * nw;
    if (P > h->rope_cap) return cbas_fail(CBAS_EINVAL, "frame has %d patches, workspace holds %d", P, h->rope_cap);
    *out = nullptr;
    for (auto& t : h->pos_tables)
        if (t.nh == nh && t.nw == nw) { t.last_use = ++h->pos_clock; *out = &t; return CBAS_OK; }
    cbas_enc::PosTable* t = nullptr;
    if ((int)h->pos_tables.size() < cbas_enc::POS_TABLES_MAX) {
        h->pos_tables.emplace_back();
        t = &h->pos_tables.back();
        const size_t Pmax = (size_t)h->rope_cap;
        if (h->cfg.use_rope) {
            HIP_TRY(hipMalloc(&t->cos, Pmax * 64 * sizeof(float)));
            HIP_TRY(hipMalloc(&t->sin, Pmax * 64 * sizeof(float)));
            HIP_TRY(hipMalloc(&t->fac, (Pmax + 1) * 32 * sizeof(float)));     // nh + nw <= nh * nw + 1
        } else {
            HIP_TRY(hipMalloc(&t->pos, Pmax * h->D * sizeof(float)));
        }
    } else {                                   // recycle the least recently used set: its readers must have finished
        t = &h->pos_tables[0];
        for (auto& c : h->pos_tables) if (c.last_use < t->last_use) t = &c;
        HIP_TRY(hipDeviceSynchronize());
    }
    t->nh = -1; t->nw = -1;                    // not valid until filled
    t->last_use = ++h->pos_clock;
    *out = t;
    return 1;                                  // caller fills it
}

// DINOv2: position embedding of the patch tokens for an nh x nw grid ([v2] interpolate_pos_encoding :93-145)
int ensure_pos_embed(cbas_enc* h, int nh, int nw) {
    cbas_enc::PosTable* t = nullptr;
    const int rc = acquire_pos_table(h, nh, nw, &t);
    if (rc < 0) return rc;
    h->pos_tab = t->pos;
    if (rc == 0) return CBAS_OK;
    const int P = nh * nw, G = h->cfg.pos_embed_grid, D = h->D;
    const float* src = h->pos_host.data() + D;               // skip the cls position
    std::vector<float> out((size_t)P * D);
    if (nh == G && nw == G) {
        memcpy(out.data(), src, out.size() * 4);
    } else {
        const std::vector<float> Wh = aa_bicubic_matrix(G, nh), Ww = aa_bicubic_matrix(G, nw);
        std::vector<float> tmp((size_t)G * nw * D, 0.f);     // width pass, then height pass (separable)
        for (int i = 0; i < G; ++i)
            for (int x = 0; x < nw; ++x) {
                float* tp = &tmp[((size_t)i * nw + x) * D];
                for (int j = 0; j < G; ++j) {
                    const float w = Ww[(size_t)x * G + j];
                    if (w == 0.f) continue;
                    const float* sp = src + ((size_t)i * G + j) * D;
                    for (int d = 0; d < D; ++d) tp[d] += w * sp[d];
                }
            }
        std::fill(out.begin(), out.end(), 0.f);
        for (int y = 0; y < nh; ++y)
            for (int i = 0; i < G; ++i) {
                const float w = Wh[(size_t)y * G + i];
                if (w == 0.f) continue;
                for (int x = 0; x < nw; ++x) {
                    float* o = &out[((size_t)y * nw + x) * D];
                    const float* tp = &tmp[((size_t)i * nw + x) * D];
                    for (int d = 0; d < D; ++d) o[d] += w * tp[d];
                }
            }
    }
    HIP_TRY(hipMemcpy(t->pos, out.data(), out.size() * 4, hipMemcpyHostToDevice));   // blocking; nothing reads t yet
    t->nh = nh; t->nw = nw;
    return CBAS_OK;
}

int ensure_rope(cbas_enc* h, int nh, int nw) {
    if (!h->cfg.use_rope) return ensure_pos_embed(h, nh, nw);
    cbas_enc::PosTable* t = nullptr;
    const int rc = acquire_pos_table(h, nh, nw, &t);
    if (rc < 0) return rc;
    h->rope_cos = t->cos; h->rope_sin = t->sin; h->rope_fac = t->fac; h->rope_nh = nh; h->rope_nw = nw;
    if (rc == 0) return CBAS_OK;
    const int P = nh * nw;
    // [tf]:96-121 patch-centre coordinates, :153-200 angles; float32 throughout like the reference
    std::vector<float> c((size_t)P * 64), s((size_t)P * 64);
    float inv_freq[16];
    for (int j = 0; j < 16; ++j) inv_freq[j] = 1.0f / powf(h->cfg.rope_theta, (float)j * (4.0f / 64.0f));
    const float two_pi = 6.283185307179586f;
    for (int iy = 0; iy < nh; ++iy)
        for (int ix = 0; ix < nw; ++ix) {
            const float cy = 2.0f * (((float)iy + 0.5f) / (float)nh) - 1.0f;
            const float cx = 2.0f * (((float)ix + 0.5f) / (float)nw) - 1.0f;
            float* cr = &c[(size_t)(iy * nw + ix) * 64];
            float* sr = &s[(size_t)(iy * nw + ix) * 64];
            for (int d = 0; d < 32; ++d) {
                const float coord = d < 16 ? cy : cx;
                const float ang = (two_pi * coord) * inv_freq[d & 15];
                cr[d] = cr[d + 32] = cosf(ang);
                sr[d] = sr[d + 32] = sinf(ang);
            }
        }
    HIP_TRY(hipMemcpy(t->cos, c.data(), c.size() * 4, hipMemcpyHostToDevice));       // blocking; nothing reads t yet
    HIP_TRY(hipMemcpy(t->sin, s.data(), s.size() * 4, hipMemcpyHostToDevice));
    // the same numbers by axis: columns 0-15 of patch (iy, ix) are those of any patch in row iy, 16-31 of column ix
    std::vector<float> fac((size_t)(nh + nw) * 32);
    for (int iy = 0; iy < nh; ++iy)
        for (int d = 0; d < 16; ++d) {
            fac[(size_t)iy * 32 + d] = c[(size_t)(iy * nw) * 64 + d];
            fac[(size_t)iy * 32 + 16 + d] = s[(size_t)(iy * nw) * 64 + d];
        }
    for (int ix = 0; ix < nw; ++ix)
        for (int d = 0; d < 16; ++d) {
            fac[(size_t)(nh + ix) * 32 + d] = c[(size_t)ix * 64 + 16 + d];
            fac[(size_t)(nh + ix) * 32 + 16 + d] = s[(size_t)ix * 64 + 16 + d];
        }
    HIP_TRY(hipMemcpy(t->fac, fac.data(), fac.size() * 4, hipMemcpyHostToDevice));
    t->nh = nh; t->nw = nw;
    return CBAS_OK;
}

// RoPE tables of the grid ensure_rope() selected, into the q|k|v GEMM's parameters
void set_rope(const cbas_enc* h, GemmParams& p) {
    if (!h->cfg.use_rope) return;
    p.rope_cos = h->rope_cos; p.rope_sin = h->rope_sin;
    p.rope_fac = h->rope_in_lds ? h->rope_fac : nullptr; p.rope_nh = h->rope_nh; p.rope_nw = h->rope_nw;
    p.rope_magic = (unsigned)((1ull << 32) / (unsigned)h->rope_nw) + 1u;
}

int check_frame(cbas_enc* h, int n, int height, int width) {
    if (!h) return cbas_fail(CBAS_EINVAL, "null encoder handle");
    if (n <= 0 || n > h->cfg.max_batch) return cbas_fail(CBAS_EINVAL, "n=%d outside (0, max_batch=%d]", n, h->cfg.max_batch);
    const int ps = h->cfg.patch_size;
    if (height < ps || width < ps) return cbas_fail(CBAS_EINVAL, "frame %dx%d smaller than one patch", height, width);
    const int64_t P = (int64_t)(height / ps) * (width / ps);
    if ((int64_t)n * (P + h->NP) > h->rows_cap || (int64_t)n * P > h->prow_cap)
        return cbas_fail(CBAS_EINVAL, "%d frames of %dx%d exceed the workspace (max_batch=%d, %dx%d)", n, height,
                         width, h->cfg.max_batch, h->cfg.max_height, h->cfg.max_width);
    return CBAS_OK;
}

// Bracket one launch with events when profiling is on (events are created lazily and reused).
struct ProfScope {
    cbas_enc* h; hipStream_t st; cbas_enc::ProfRec* r = nullptr;
    ProfScope(cbas_enc* h_, hipStream_t st_, int cat, double flops) : h(h_), st(st_) {
        if (!h->prof_on) return;
        if (h->prof_used == h->prof.size()) {
            cbas_enc::ProfRec n{};
            if (hipEventCreate(&n.a) != hipSuccess || hipEventCreate(&n.b) != hipSuccess) return;
            h->prof.push_back(n);
        }
        r = &h->prof[h->prof_used++];
        r->cat = cat; r->flops = flops;
        (void)hipEventRecord(r->a, st);
    }
    ~ProfScope() { if (r) (void)hipEventRecord(r->b, st); }
};
#define PROF(cat, flops) ProfScope _prof_scope_##cat(h, st, cat, flops)


// Last transformer layer when only the CLS rows are consumed.  K and V are still projected for every row
// (the CLS query attends to all tokens); the query, attention, o_proj, LayerNorm 2 and the MLP run on the n CLS
// rows, read and written in place in the residual stream with a row stride of T*D.
int run_last_layer_cls(cbas_enc* h, const LayerW& w, int n, int T, hipStream_t st, bool fold) {
    const int D = h->D, F = h->F, M = n * T, M_pad = (int)round_up(M, 128);
    const int64_t cap = round_up(h->cfg.max_batch, 128);
    f16* qc = h->cls16;                     // [n][D] CLS queries (bias added, scaled by 1/8; no RoPE on prefix rows)
    f16* cc = qc + cap * D;                 // [n][D] attention context of the CLS rows
    f16* hc = cc + cap * D;                 // [n][D] LayerNorm 2 of the CLS rows
    f16* uc = hc + cap * D;                 // [n][F] GELU(up_proj)
    const bool split = h->cfg.precision == 1, f8 = h->cfg.precision == 2;
    const int sc_ld = (int)h->rows_cap;
    GemmParams kv{};                        // k | v sections of the fused QKV weight, all rows
    if (f8) {
        { PROF(CBAS_PROF_LAYERNORM, 0.0); LAUNCH_TRY(launch_layernorm_f8(h->x, D, w.ln1_w, w.ln1_b, (uint8_t*)h->h16, h->sc_h, sc_ld, M, D, h->cfg.layer_norm_eps, st)); }
        // the CLS tail of this layer stays fp16 (fp16 weights, n rows): its LayerNorm 1 rows go to the LN2 slot for now
        { PROF(CBAS_PROF_LAYERNORM, 0.0); LAUNCH_TRY(launch_layernorm_f16(h->x, (int64_t)T * D, w.ln1_w, w.ln1_b, hc, n, D, h->cfg.layer_norm_eps, st)); }
        kv.A8 = (const uint8_t*)h->h16; kv.A_sc = h->sc_h; kv.sc_lda = sc_ld;
        kv.W8 = w.wqkv8 + (size_t)D * D; kv.W_sc = w.sqkv + D; kv.sc_ldw = 3 * D;
    } else if (fold) {
        // LayerNorm fold: the k | v rows of the folded weight on the raw fp16 residual stream; the n CLS rows' LayerNorm 1
        // (for the query) is computed on its own, into the LN2 slot for now
        { PROF(CBAS_PROF_LAYERNORM, 0.0); LAUNCH_TRY(launch_layernorm_f16(h->x, (int64_t)T * D, w.ln1_w, w.ln1_b, hc, n, D, h->cfg.layer_norm_eps, st)); }
        kv.A = h->x16; kv.W = w.wqkv_f + (size_t)D * D;
        kv.tile = (long)((M + 255) / 256) * (2 * D / 256) >= 120 ? GEMM_TILE_PP_AUTO : GEMM_TILE_PP_128x256;
        kv.ln_in = h->lnst; kv.ln_colsum = w.qkv_cs + D; kv.ln_parts = D / 256; kv.ln_ld = (int)h->rows_cap; kv.ln_eps = h->cfg.layer_norm_eps;
    } else {
        { PROF(CBAS_PROF_LAYERNORM, 0.0); LAUNCH_TRY(launch_layernorm_f16(h->x, D, w.ln1_w, w.ln1_b, h->h16, M, D, h->cfg.layer_norm_eps, st)); }
        kv.A = h->h16; kv.W = w.wqkv + (size_t)D * D; kv.W_lo = split ? w.wqkv_lo + (size_t)D * D : nullptr;
    }
    kv.M = M; kv.M_pad = M_pad; kv.N = 2 * D; kv.K = D; kv.bias = (fold ? w.qkv_bf : w.qkv_b) + D; kv.out_f16 = h->qkv16 + D; kv.ldo = 3 * D;
    kv.tokens_per_frame = T; kv.n_prefix = h->NP; kv.D = D; kv.sec0 = 1;
    set_rope(h, kv);
    { PROF(CBAS_PROF_QKV, 2.0 * M * 2.0 * D * D); LAUNCH_TRY(launch_gemm(fold ? EPI_QKV_LN : EPI_QKV, kv, st)); }
    const bool compact_q = f8 || fold;      // the CLS rows' LayerNorm 1 sits in hc
    GemmParams q{};                         // q section, CLS rows only (row b*T of h16; the compact fp16 rows when f8 / folded)
    q.A = compact_q ? hc : h->h16; q.lda = compact_q ? D : T * D; q.W = w.wqkv; q.W_lo = split ? w.wqkv_lo : nullptr;
    q.M = n; q.M_pad = n; q.N = D; q.K = D; q.bias = w.qkv_b; q.out_f16 = qc; q.ldo = D;
    q.tokens_per_frame = 1; q.n_prefix = 1; q.D = D; q.sec0 = 0;      // every row is token 0: no RoPE
    { PROF(CBAS_PROF_QKV, 2.0 * n * (double)D * D); LAUNCH_TRY(launch_gemm(EPI_QKV, q, st)); }
    { PROF(CBAS_PROF_ATTENTION, 4.0 * n * (double)T * D); LAUNCH_TRY(launch_attention(h->qkv16, qc, cc, nullptr, 0, n, T, D, h->NH, st)); }
    GemmParams o{};
    o.A = cc; o.W = w.wo; o.W_lo = split ? w.wo_lo : nullptr;
    o.M = n; o.M_pad = n; o.N = D; o.K = D; o.bias = w.o_b; o.lambda = w.ls1; o.out_f32 = h->x; o.ldo = T * D;
    { PROF(CBAS_PROF_OPROJ, 2.0 * n * (double)D * D); LAUNCH_TRY(launch_gemm(EPI_RESID, o, st)); }
    { PROF(CBAS_PROF_LAYERNORM, 0.0); LAUNCH_TRY(launch_layernorm_f16(h->x, (int64_t)T * D, w.ln2_w, w.ln2_b, hc, n, D, h->cfg.layer_norm_eps, st)); }
    GemmParams u{};
    u.A = hc; u.W = w.wup; u.W_lo = split ? w.wup_lo : nullptr;
    u.M = n; u.M_pad = n; u.N = F; u.K = D; u.bias = w.up_b; u.out_f16 = uc; u.ldo = F;
    { PROF(CBAS_PROF_UP, 2.0 * n * (double)F * D); LAUNCH_TRY(launch_gemm(EPI_GELU, u, st)); }
    GemmParams d{};
    d.A = uc; d.W = w.wdown; d.W_lo = split ? w.wdown_lo : nullptr;
    d.M = n; d.M_pad = n; d.N = D; d.K = F; d.bias = w.down_b; d.lambda = w.ls2; d.out_f32 = h->x; d.ldo = T * D;
    { PROF(CBAS_PROF_DOWN, 2.0 * n * (double)F * D); LAUNCH_TRY(launch_gemm(EPI_RESID, d, st)); }
    return CBAS_OK;
}

// precision 3: the same schedule with every buffer and every contraction in fp32 (vit_f32.hip).  The workspace pointers
// (A_patch, h16, qkv16, u16, cls16) are allocated at 4 bytes per element in this mode and hold floats.
int run_blocks_f32(cbas_enc* h, int n, int height, int width, float* cls_f32, f16* cls_f16, hipStream_t st,
                   int stop_layer, int stop_stage) {
    const int ps = h->cfg.patch_size;
    const int nh = height / ps, nw = width / ps, P = nh * nw, T = P + h->NP;
    const int D = h->D, F = h->F;
    const int M = n * T;
    h->last_rows = M;
    int rc = ensure_rope(h, nh, nw);
    if (rc) return rc;
    float* const A32 = reinterpret_cast<float*>(h->A_patch);
    float* const h32 = reinterpret_cast<float*>(h->h16);
    float* const qkv32 = reinterpret_cast<float*>(h->qkv16);
    float* const u32 = reinterpret_cast<float*>(h->u16);
    const float eps = h->cfg.layer_norm_eps;

    // precision 4: the same schedule with every GEMM's products on the fp16 pipe as three-term splits; operand scales:
    // LayerNorm rows and pixels as they are, attention context x 16, GELU output x 4 (typical magnitudes ~0.05 / ~0.3:
    // keeps their low halves out of fp16's subnormal range), weights by their per-tensor power of two
    const int split = h->cfg.precision == 4;
    auto sp = [&](Gemm32VitParams& q, float a_scale, float w_scale) { q.split = split; q.a_scale = a_scale; q.w_scale = w_scale; };
    Gemm32VitParams g{};
    sp(g, 1.f, h->sc_patch);
    g.A = A32; g.lda = 256; g.W = h->wpatch32; g.M = n * P; g.N = D; g.K = 256; g.bias = h->patch_b; g.out = h->x; g.ldo = D;
    g.patches_per_frame = P; g.tokens_per_frame = T; g.n_prefix = h->NP; g.pos = h->cfg.use_rope ? nullptr : h->pos_tab;
    { PROF(CBAS_PROF_PATCH, 2.0 * g.M * g.N * g.K); LAUNCH_TRY(launch_gemm_f32_vit(EPI_PATCH, g, st)); }
    if (stop_layer == 0 && stop_stage == 0) return CBAS_OK;

    const bool prune = h->prune_last && stop_layer < 0 && (cls_f32 || cls_f16);
    auto qkv_params = [&](const LayerW& w, Gemm32VitParams& q) {
        q.K = D; q.D = D; q.tokens_per_frame = T; q.n_prefix = h->NP;
        if (h->cfg.use_rope) {
            q.rope_cos = h->rope_cos; q.rope_sin = h->rope_sin;
            if (h->rope_in_lds) {
                q.rope_fac = h->rope_fac; q.rope_nh = h->rope_nh; q.rope_nw = h->rope_nw;
                q.rope_magic = (unsigned)((1ull << 32) / (unsigned)h->rope_nw) + 1u;
            }
        }
    };
    for (int l = 0; l < h->L; ++l) {
        const LayerW& w = h->layers[l];
        auto stop = [&](int stage) { return stop_layer == l && stop_stage == stage; };
        if (prune && l == h->L - 1) {
            // the last layer feeds only the final norm of the CLS rows (see run_last_layer_cls): K | V for every row,
            // everything else for the n CLS rows, read and written in place with a row stride of T*D
            const int64_t cap = round_up(h->cfg.max_batch, 128);
            float* qc = reinterpret_cast<float*>(h->cls16);      // [n][D] CLS queries
            float* cc = qc + cap * D;                            // [n][D] attention context
            float* hc = cc + cap * D;                            // [n][D] LayerNorm rows
            float* uc = hc + cap * D;                            // [n][F] GELU(up_proj)
            { PROF(CBAS_PROF_LAYERNORM, 0.0); LAUNCH_TRY(launch_layernorm_f32(h->x, D, w.ln1_w, w.ln1_b, h32, M, D, eps, split, st)); }
            Gemm32VitParams kv{};
            qkv_params(w, kv);
            sp(kv, 1.f, w.sc_qkv);
            kv.A = h32; kv.lda = D; kv.W = w.wqkv32 + (size_t)D * D; kv.M = M; kv.N = 2 * D; kv.bias = w.qkv_b + D;
            kv.out = qkv32 + D; kv.ldo = 3 * D; kv.sec0 = 1;
            { PROF(CBAS_PROF_QKV, 2.0 * M * 2.0 * D * D); LAUNCH_TRY(launch_gemm_f32_vit(EPI_QKV, kv, st)); }
            Gemm32VitParams q{};
            qkv_params(w, q);
            sp(q, 1.f, w.sc_qkv);
            q.A = h32; q.lda = (int64_t)T * D; q.W = w.wqkv32; q.M = n; q.N = D; q.bias = w.qkv_b; q.out = qc; q.ldo = D;
            q.tokens_per_frame = 1; q.n_prefix = 1; q.sec0 = 0;       // every row is token 0: no RoPE
            { PROF(CBAS_PROF_QKV, 2.0 * n * (double)D * D); LAUNCH_TRY(launch_gemm_f32_vit(EPI_QKV, q, st)); }
            { PROF(CBAS_PROF_ATTENTION, 4.0 * n * (double)T * D); LAUNCH_TRY(launch_attention_f32(qkv32, qc, cc, n, T, D, h->NH, split ? 16.f : 0.f, st)); }
            Gemm32VitParams o{};
            sp(o, 16.f, w.sc_o);
            o.A = cc; o.lda = D; o.W = w.wo32; o.M = n; o.N = D; o.K = D; o.bias = w.o_b; o.lambda = w.ls1; o.out = h->x; o.ldo = (int64_t)T * D;
            { PROF(CBAS_PROF_OPROJ, 2.0 * n * (double)D * D); LAUNCH_TRY(launch_gemm_f32_vit(EPI_RESID, o, st)); }
            { PROF(CBAS_PROF_LAYERNORM, 0.0); LAUNCH_TRY(launch_layernorm_f32(h->x, (int64_t)T * D, w.ln2_w, w.ln2_b, hc, n, D, eps, split, st)); }
            Gemm32VitParams u{};
            sp(u, 1.f, w.sc_up);
            u.out_scale = 4.f;
            u.A = hc; u.lda = D; u.W = w.wup32; u.M = n; u.N = F; u.K = D; u.bias = w.up_b; u.out = uc; u.ldo = F;
            { PROF(CBAS_PROF_UP, 2.0 * n * (double)F * D); LAUNCH_TRY(launch_gemm_f32_vit(EPI_GELU, u, st)); }
            Gemm32VitParams d{};
            sp(d, 4.f, w.sc_down);
            d.A = uc; d.lda = F; d.W = w.wdown32; d.M = n; d.N = D; d.K = F; d.bias = w.down_b; d.lambda = w.ls2; d.out = h->x; d.ldo = (int64_t)T * D;
            { PROF(CBAS_PROF_DOWN, 2.0 * n * (double)F * D); LAUNCH_TRY(launch_gemm_f32_vit(EPI_RESID, d, st)); }
            break;
        }
        { PROF(CBAS_PROF_LAYERNORM, 0.0); LAUNCH_TRY(launch_layernorm_f32(h->x, D, w.ln1_w, w.ln1_b, h32, M, D, eps, split, st)); }
        if (stop(1)) return CBAS_OK;
        Gemm32VitParams q{};
        qkv_params(w, q);
        sp(q, 1.f, w.sc_qkv);
        q.A = h32; q.lda = D; q.W = w.wqkv32; q.M = M; q.N = 3 * D; q.bias = w.qkv_b; q.out = qkv32; q.ldo = 3 * D;
        { PROF(CBAS_PROF_QKV, 2.0 * M * 3.0 * D * D); LAUNCH_TRY(launch_gemm_f32_vit(EPI_QKV, q, st)); }
        if (stop(2)) return CBAS_OK;
        { PROF(CBAS_PROF_ATTENTION, 4.0 * n * (double)T * T * D); LAUNCH_TRY(launch_attention_f32(qkv32, nullptr, h32, n, T, D, h->NH, split ? 16.f : 0.f, st)); }
        if (stop(3)) return CBAS_OK;
        Gemm32VitParams o{};
        sp(o, 16.f, w.sc_o);
        o.A = h32; o.lda = D; o.W = w.wo32; o.M = M; o.N = D; o.K = D; o.bias = w.o_b; o.lambda = w.ls1; o.out = h->x; o.ldo = D;
        { PROF(CBAS_PROF_OPROJ, 2.0 * M * (double)D * D); LAUNCH_TRY(launch_gemm_f32_vit(EPI_RESID, o, st)); }
        if (stop(4)) return CBAS_OK;
        { PROF(CBAS_PROF_LAYERNORM, 0.0); LAUNCH_TRY(launch_layernorm_f32(h->x, D, w.ln2_w, w.ln2_b, h32, M, D, eps, split, st)); }
        if (stop(5)) return CBAS_OK;
        Gemm32VitParams u{};
        sp(u, 1.f, w.sc_up);
        u.out_scale = 4.f;
        u.A = h32; u.lda = D; u.W = w.wup32; u.M = M; u.N = F; u.K = D; u.bias = w.up_b; u.out = u32; u.ldo = F;
        { PROF(CBAS_PROF_UP, 2.0 * M * (double)F * D); LAUNCH_TRY(launch_gemm_f32_vit(EPI_GELU, u, st)); }
        if (stop(6)) return CBAS_OK;
        Gemm32VitParams d{};
        sp(d, 4.f, w.sc_down);
        d.A = u32; d.lda = F; d.W = w.wdown32; d.M = M; d.N = D; d.K = F; d.bias = w.down_b; d.lambda = w.ls2; d.out = h->x; d.ldo = D;
        { PROF(CBAS_PROF_DOWN, 2.0 * M * (double)F * D); LAUNCH_TRY(launch_gemm_f32_vit(EPI_RESID, d, st)); }
        if (stop(7)) return CBAS_OK;
    }
    if (cls_f32 || cls_f16)
        LAUNCH_TRY(launch_final_norm_cls(h->x, h->norm_w, h->norm_b, cls_f32, cls_f16, n, T, D, eps, st, h->nonfinite_dev));
    return CBAS_OK;
}

// Everything after ingest: patch GEMM, L transformer blocks, final CLS norm.
int run_blocks(cbas_enc* h, int n, int height, int width, int patch_k, float in_scale, float* cls_f32,
               f16* cls_f16, hipStream_t st, int stop_layer, int stop_stage) {
    if (h->cfg.precision >= 3) return run_blocks_f32(h, n, height, width, cls_f32, cls_f16, st, stop_layer, stop_stage);
    const int ps = h->cfg.patch_size;
    const int nh = height / ps, nw = width / ps, P = nh * nw, T = P + h->NP;
    const int D = h->D, F = h->F;
    const int M = n * T, M_pad = (int)round_up(M, 128);
    h->last_rows = M;
    int rc = ensure_rope(h, nh, nw);
    if (rc) return rc;

    GemmParams g{};
    g.A = h->A_patch;
    g.W = patch_k == 256 ? h->wpatch : h->wpatch2;
    g.W_lo = h->cfg.precision == 1 ? (patch_k == 256 ? h->wpatch_lo : h->wpatch2_lo) : nullptr;
    g.M = n * P; g.M_pad = (int)round_up(n * P, 128); g.N = D; g.K = patch_k;
    g.bias = h->patch_b; g.out_f32 = h->x; g.ldo = D;
    g.patches_per_frame = P; g.tokens_per_frame = T; g.n_prefix = h->NP; g.in_scale = in_scale;
    g.pos = h->cfg.use_rope ? nullptr : h->pos_tab;
    { PROF(CBAS_PROF_PATCH, 2.0 * g.M * g.N * g.K); LAUNCH_TRY(launch_gemm(EPI_PATCH, g, st)); }
    if (stop_layer == 0 && stop_stage == 0) return CBAS_OK;

    // The last layer feeds only the final norm of the CLS rows, so everything after its K/V projection is
    // done for n rows instead of n*T (rows are independent: bit-identical CLS).  Debug taps run it in full.
    const bool prune = h->prune_last && stop_layer < 0 && (cls_f32 || cls_f16);
    const bool split = h->cfg.precision == 1, f8 = h->cfg.precision == 2;
    // LayerNorm fold.  LN(x) W^T + b = rstd (x (gamma o W)^T - mean colsum(gamma o W)) + (beta W^T + b): the two GEMMs that
    // consume a LayerNorm output ([tf]:404-445: q|k|v after norm1, up_proj after norm2) run on the raw fp16 residual stream
    // with gamma folded into their weights and apply mean / rstd in their epilogues; the two GEMMs that produce the residual
    // stream (o_proj, down_proj) write that fp16 copy and the row statistics in theirs.  22 of the 24 LayerNorm launches
    // of a ViT-B step and their 114 MB per layer of traffic disappear.  The fold lives in the ping-pong kernel only, so in this
    // mode all four GEMMs run there at EVERY batch size (128-row tiles for small problems, as precision 2 does): a frame's
    // CLS row must not depend on how many frames shared its batch.  Debug taps and precision 1 / 2 keep the LayerNorm kernels.
    const bool fold = h->ln_fold && h->fold_ok && stop_layer < 0;
    const int ln_ld = (int)h->rows_cap;
    auto pp_tile = [&](int N) { return (long)((M + 255) / 256) * (N / 256) >= 120 ? GEMM_TILE_PP_AUTO : GEMM_TILE_PP_128x256; };
    if (fold) { PROF(CBAS_PROF_LAYERNORM, 0.0); LAUNCH_TRY(launch_ln_stats_x16(h->x, h->x16, h->lnst, ln_ld, M, D, st)); }
    if (f8 && stop_layer >= 0) return cbas_fail(CBAS_EINVAL, "debug taps read fp16 buffers; not available with precision 2");
    const int sc_ld = (int)h->rows_cap;
    uint8_t* const h8 = reinterpret_cast<uint8_t*>(h->h16);      // fp8 activations live in the fp16 buffers' memory
    uint8_t* const u8 = reinterpret_cast<uint8_t*>(h->u16);
    for (int l = 0; l < h->L; ++l) {
        const LayerW& w = h->layers[l];
        auto stop = [&](int stage) { return stop_layer == l && stop_stage == stage; };
        if (prune && l == h->L - 1) {
            rc = run_last_layer_cls(h, w, n, T, st, fold);
            if (rc) return rc;
            break;
        }
        if (fold) {
            auto ln_consumer = [&](GemmParams& g, const float* colsum) {
                g.ln_in = h->lnst; g.ln_colsum = colsum; g.ln_parts = D / 256; g.ln_ld = ln_ld; g.ln_eps = h->cfg.layer_norm_eps;
            };
            auto ln_producer = [&](GemmParams& g) { g.x16_out = h->x16; g.ln_out = h->lnst; g.ln_ld = ln_ld; };
            const bool more = l + 1 < h->L;             // the last layer's down_proj feeds the final norm only
            GemmParams q{};
            q.tile = pp_tile(3 * D);
            q.A = h->x16; q.W = w.wqkv_f; q.M = M; q.M_pad = M_pad; q.N = 3 * D; q.K = D; q.bias = w.qkv_bf; q.out_f16 = h->qkv16; q.ldo = 3 * D;
            q.tokens_per_frame = T; q.n_prefix = h->NP; q.D = D;
            ln_consumer(q, w.qkv_cs);
            set_rope(h, q);
            { PROF(CBAS_PROF_QKV, 2.0 * M * 3.0 * D * D); LAUNCH_TRY(launch_gemm(EPI_QKV_LN, q, st)); }
            { PROF(CBAS_PROF_ATTENTION, 4.0 * n * (double)T * T * D);
              LAUNCH_TRY(launch_attention(h->qkv16, nullptr, h->h16, nullptr, sc_ld, n, T, D, h->NH, st)); }
            GemmParams o{};
            o.tile = pp_tile(D);
            o.A = h->h16; o.W = w.wo; o.M = M; o.M_pad = M_pad; o.N = D; o.K = D; o.bias = w.o_b; o.lambda = w.ls1; o.out_f32 = h->x; o.ldo = D;
            ln_producer(o);
            { PROF(CBAS_PROF_OPROJ, 2.0 * M * (double)D * D); LAUNCH_TRY(launch_gemm(EPI_RESID_LN, o, st)); }
            GemmParams u{};
            u.tile = pp_tile(F);
            u.A = h->x16; u.W = w.wup_f; u.out_f16 = h->u16; u.M = M; u.M_pad = M_pad; u.N = F; u.K = D; u.bias = w.up_bf; u.ldo = F;
            ln_consumer(u, w.up_cs);
            { PROF(CBAS_PROF_UP, 2.0 * M * (double)F * D); LAUNCH_TRY(launch_gemm(EPI_GELU_LN, u, st)); }
            GemmParams d{};
            d.tile = pp_tile(D);
            d.A = h->u16; d.W = w.wdown; d.M = M; d.M_pad = M_pad; d.N = D; d.K = F; d.bias = w.down_b; d.lambda = w.ls2; d.out_f32 = h->x; d.ldo = D;
            if (more) ln_producer(d);
            { PROF(CBAS_PROF_DOWN, 2.0 * M * (double)F * D); LAUNCH_TRY(launch_gemm(more ? EPI_RESID_LN : EPI_RESID, d, st)); }
            continue;
        }
        if (f8) { PROF(CBAS_PROF_LAYERNORM, 0.0); LAUNCH_TRY(launch_layernorm_f8(h->x, D, w.ln1_w, w.ln1_b, h8, h->sc_h, sc_ld, M, D, h->cfg.layer_norm_eps, st)); }
        else { PROF(CBAS_PROF_LAYERNORM, 0.0); LAUNCH_TRY(launch_layernorm_f16(h->x, D, w.ln1_w, w.ln1_b, h->h16, M, D, h->cfg.layer_norm_eps, st)); }
        if (stop(1)) return CBAS_OK;

        GemmParams q{};
        if (f8) { q.A8 = h8; q.A_sc = h->sc_h; q.sc_lda = sc_ld; q.W8 = w.wqkv8; q.W_sc = w.sqkv; }
        else { q.A = h->h16; q.W = w.wqkv; q.W_lo = split ? w.wqkv_lo : nullptr; }
        q.M = M; q.M_pad = M_pad; q.N = 3 * D; q.K = D; q.bias = w.qkv_b; q.out_f16 = h->qkv16; q.ldo = 3 * D;
        q.tokens_per_frame = T; q.n_prefix = h->NP; q.D = D;
        set_rope(h, q);
        { PROF(CBAS_PROF_QKV, 2.0 * M * 3.0 * D * D); LAUNCH_TRY(launch_gemm(EPI_QKV, q, st)); }
        if (stop(2)) return CBAS_OK;

        { PROF(CBAS_PROF_ATTENTION, 4.0 * n * (double)T * T * D);
          LAUNCH_TRY(launch_attention(h->qkv16, nullptr, h->h16, f8 ? h->sc_h : nullptr, sc_ld, n, T, D, h->NH, st)); }
        if (stop(3)) return CBAS_OK;

        GemmParams o{};
        if (f8) { o.A8 = h8; o.A_sc = h->sc_h; o.sc_lda = sc_ld; o.W8 = w.wo8; o.W_sc = w.so; }
        else { o.A = h->h16; o.W = w.wo; o.W_lo = split ? w.wo_lo : nullptr; }
        o.M = M; o.M_pad = M_pad; o.N = D; o.K = D; o.bias = w.o_b; o.lambda = w.ls1; o.out_f32 = h->x; o.ldo = D;
        { PROF(CBAS_PROF_OPROJ, 2.0 * M * (double)D * D); LAUNCH_TRY(launch_gemm(EPI_RESID, o, st)); }
        if (stop(4)) return CBAS_OK;

        if (f8) { PROF(CBAS_PROF_LAYERNORM, 0.0); LAUNCH_TRY(launch_layernorm_f8(h->x, D, w.ln2_w, w.ln2_b, h8, h->sc_h, sc_ld, M, D, h->cfg.layer_norm_eps, st)); }
        else { PROF(CBAS_PROF_LAYERNORM, 0.0); LAUNCH_TRY(launch_layernorm_f16(h->x, D, w.ln2_w, w.ln2_b, h->h16, M, D, h->cfg.layer_norm_eps, st)); }
        if (stop(5)) return CBAS_OK;

        GemmParams u{};
        if (f8) {
            u.A8 = h8; u.A_sc = h->sc_h; u.sc_lda = sc_ld; u.W8 = w.wup8; u.W_sc = w.sup;
            u.out_f8 = u8; u.out_sc = h->sc_u; u.sc_ldo = sc_ld;
        } else { u.A = h->h16; u.W = w.wup; u.W_lo = split ? w.wup_lo : nullptr; u.out_f16 = h->u16; }
        u.M = M; u.M_pad = M_pad; u.N = F; u.K = D; u.bias = w.up_b; u.ldo = F;
        { PROF(CBAS_PROF_UP, 2.0 * M * (double)F * D); LAUNCH_TRY(launch_gemm(f8 ? EPI_GELU_F8 : EPI_GELU, u, st)); }
        if (stop(6)) return CBAS_OK;

        GemmParams d{};
        if (f8) { d.A8 = u8; d.A_sc = h->sc_u; d.sc_lda = sc_ld; d.W8 = w.wdown8; d.W_sc = w.sdown; }
        else { d.A = h->u16; d.W = w.wdown; d.W_lo = split ? w.wdown_lo : nullptr; }
        d.M = M; d.M_pad = M_pad; d.N = D; d.K = F; d.bias = w.down_b; d.lambda = w.ls2; d.out_f32 = h->x; d.ldo = D;
        { PROF(CBAS_PROF_DOWN, 2.0 * M * (double)F * D); LAUNCH_TRY(launch_gemm(EPI_RESID, d, st)); }
        if (stop(7)) return CBAS_OK;
    }
    if (cls_f32 || cls_f16)
        LAUNCH_TRY(launch_final_norm_cls(h->x, h->norm_w, h->norm_b, cls_f32, cls_f16, n, T, D,
                                         h->cfg.layer_norm_eps, st, h->nonfinite_dev));
    return CBAS_OK;
}

int forward_u8_one(cbas_enc* h, const uint8_t* frames_dev, int n, int height, int width, int64_t frame_stride,
                   int64_t row_stride, int64_t pixel_stride, float* cls_f32, f16* cls_f16, hipStream_t st,
                   int stop_layer, int stop_stage) {
    const int ps = h->cfg.patch_size;
    const int T = (height / ps) * (width / ps) + h->NP;
    if (h->cfg.precision >= 3)
        LAUNCH_TRY(launch_im2col_u8_f32(frames_dev, n, height, width, frame_stride, row_stride, pixel_stride,
                                        reinterpret_cast<float*>(h->A_patch), h->x, h->prefix, h->NP, h->D, T, ps,
                                        h->cfg.precision == 4, st));
    else
        LAUNCH_TRY(launch_im2col_u8(frames_dev, n, height, width, frame_stride, row_stride, pixel_stride, h->A_patch,
                                    h->x, h->prefix, h->NP, h->D, T, ps, st));
    return run_blocks(h, n, height, width, 256, 1.0f / 255.0f, cls_f32, cls_f16, st, stop_layer, stop_stage);
}

// Point the handle's workspace pointers at lane `l` (kernel arguments are captured at launch, so the
// host-side switch is safe while the other lane's kernels are still running).
void use_lane(cbas_enc* h, int l) {
    const cbas_enc::Lane& L = h->lanes[l];
    h->A_patch = L.A_patch; h->x = L.x; h->h16 = L.h16; h->qkv16 = L.qkv16; h->u16 = L.u16; h->cls16 = L.cls16;
    h->sc_h = L.sc_h; h->sc_u = L.sc_u; h->x16 = L.x16; h->lnst = L.lnst;
}

int forward_u8(cbas_enc* h, const uint8_t* frames_dev, int n, int height, int width, int64_t frame_stride,
               int64_t row_stride, int64_t pixel_stride, float* cls_f32, f16* cls_f16, hipStream_t st,
               int stop_layer, int stop_stage) {
    int rc = check_frame(h, n, height, width);
    if (rc) return rc;
    if (!frames_dev) return cbas_fail(CBAS_EINVAL, "frames_dev is NULL");
    HIP_TRY(hipSetDevice(h->device));
    return forward_u8_one(h, frames_dev, n, height, width, frame_stride, row_stride, pixel_stride, cls_f32, cls_f16, st,
                          stop_layer, stop_stage);
}

}  // namespace

extern "C" int64_t cbas_enc_weights_count(const cbas_enc_config* cfg) { return cfg ? weights_count(*cfg) : -1; }

extern "C" void cbas_enc_destroy(cbas_enc* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->compute) (void)hipStreamSynchronize(h->compute);
    if (h->copy) (void)hipStreamSynchronize(h->copy);
    if (h->nonfinite_dev) (void)hipFree(h->nonfinite_dev);
    for (Slot& s : h->slots) {
        if (s.in_host) (void)hipHostFree(s.in_host);
        if (s.out16_host) (void)hipHostFree(s.out16_host);
        if (s.out32_host) (void)hipHostFree(s.out32_host);
        if (s.in_dev) (void)hipFree(s.in_dev);
        if (s.out16_dev) (void)hipFree(s.out16_dev);
        if (s.out32_dev) (void)hipFree(s.out32_dev);
        if (s.ev_copied) (void)hipEventDestroy(s.ev_copied);
        if (s.ev_done) (void)hipEventDestroy(s.ev_done);
        if (s.ev_in) (void)hipEventDestroy(s.ev_in);
    }
    for (auto& r : h->prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    if (h->lanes[0].x) use_lane(h, 0);
    if (h->lanes[1].stream) { (void)hipStreamSynchronize(h->lanes[1].stream); (void)hipStreamDestroy(h->lanes[1].stream); }
    if (h->lane0_async_done) (void)hipEventDestroy(h->lane0_async_done);
    if (h->sync_done) (void)hipEventDestroy(h->sync_done);
    {
        void* b1[] = {h->lanes[1].A_patch, h->lanes[1].x, h->lanes[1].h16, h->lanes[1].qkv16, h->lanes[1].u16, h->lanes[1].cls16,
                      h->lanes[1].sc_h, h->lanes[1].sc_u, h->lanes[1].x16, h->lanes[1].lnst};
        for (void* b : b1) if (b) (void)hipFree(b);
    }
    for (auto& t : h->pos_tables) { if (t.cos) (void)hipFree(t.cos); if (t.sin) (void)hipFree(t.sin); if (t.fac) (void)hipFree(t.fac); if (t.pos) (void)hipFree(t.pos); }
    void* bufs[] = {h->blob, h->w16, h->w16_lo, h->qkv_bias_all, h->prefix_dev,
                    h->A_patch, h->h16, h->qkv16, h->u16, h->x, h->cls16, h->w8, h->w8_sc, h->sc_h, h->sc_u,
                    h->w16_fold, h->fold_vec, h->x16, h->lnst, h->w32};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    if (h->compute) (void)hipStreamDestroy(h->compute);
    if (h->copy) (void)hipStreamDestroy(h->copy);
    if (h->aux) (void)hipStreamDestroy(h->aux);
    delete h;
}

extern "C" int cbas_enc_create(const cbas_enc_config* cfg, const float* weights_host, int64_t n_weights,
                               int device_id, cbas_enc** out) {
    if (!cfg || !weights_host || !out) return cbas_fail(CBAS_EINVAL, "null argument");
    *out = nullptr;
    const cbas_enc_config& c = *cfg;
    if (c.hidden_size <= 0 || c.hidden_size % 128 || c.num_heads * 64 != c.hidden_size)
        return cbas_fail(CBAS_EINVAL, "hidden_size=%d must be a multiple of 128 with head_dim 64 (num_heads=%d)",
                         c.hidden_size, c.num_heads);
    if (c.intermediate_size <= 0 || c.intermediate_size % 128)
        return cbas_fail(CBAS_EINVAL, "intermediate_size=%d must be a multiple of 128", c.intermediate_size);
    if (c.hidden_size > 1024) return cbas_fail(CBAS_EINVAL, "hidden_size > 1024 not supported");
    if (c.patch_size != 16 && c.patch_size != 14) return cbas_fail(CBAS_EINVAL, "patch_size must be 14 or 16");
    if (c.precision < 0 || c.precision > 4)
        return cbas_fail(CBAS_EINVAL, "precision=%d: 0 (fp16), 1 (fp16 hi+lo weights), 2 (MX-fp8), 3 (fp32, the reference's CPU arithmetic) "
                                      "or 4 (fp32 storage, GEMM products as three-term fp16 splits)", c.precision);
    if (c.precision == 2 && (c.hidden_size % 256 || c.intermediate_size % 256))
        return cbas_fail(CBAS_EINVAL, "precision 2 (MX-fp8) needs hidden_size and intermediate_size to be multiples of 256 "
                                      "(K-tiles of 128 consumed in pairs); got %d / %d", c.hidden_size, c.intermediate_size);
    if ((c.use_rope != 0) == (c.pos_embed_grid > 0))
        return cbas_fail(CBAS_EINVAL, "exactly one of use_rope / pos_embed_grid must be set");
    if (c.num_layers <= 0 || c.num_register_tokens < 0 || c.max_batch <= 0 || c.max_height < c.patch_size || c.max_width < c.patch_size)
        return cbas_fail(CBAS_EINVAL, "bad layer/register/batch/frame-size field");
    if (n_weights != weights_count(c))
        return cbas_fail(CBAS_EINVAL, "weights blob has %lld floats, config needs %lld", (long long)n_weights,
                         (long long)weights_count(c));
    HIP_TRY(hipSetDevice(device_id));

    cbas_enc* h = new (std::nothrow) cbas_enc();
    if (!h) return cbas_fail(CBAS_ENOMEM, "out of host memory");
    h->cfg = c; h->device = device_id;
    h->D = c.hidden_size; h->F = c.intermediate_size; h->L = c.num_layers; h->NH = c.num_heads;
    h->R = c.num_register_tokens; h->NP = 1 + h->R;
    const int64_t D = h->D, F = h->F;

#define CREATE_TRY(expr)                                                                                  \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess) {                                                                           \
            cbas_fail(CBAS_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            cbas_enc_destroy(h);                                                                          \
            return _e == hipErrorOutOfMemory ? CBAS_ENOMEM : CBAS_EHIP;                                   \
        }                                                                                                 \
    } while (0)

    CREATE_TRY(hipStreamCreateWithFlags(&h->compute, hipStreamNonBlocking));
    CREATE_TRY(hipStreamCreateWithFlags(&h->copy, hipStreamNonBlocking));
    CREATE_TRY(hipStreamCreateWithFlags(&h->aux, hipStreamNonBlocking));
    CREATE_TRY(hipMalloc(&h->blob, n_weights * sizeof(float)));
    CREATE_TRY(hipMemcpy(h->blob, weights_host, n_weights * sizeof(float), hipMemcpyHostToDevice));

    // fp16 weight arena: patch (D*256 + D*512) + per layer (3DD + DD + FD + DF)
    const int64_t n16 = D * 256 + D * 512 + (int64_t)h->L * (4 * D * D + 2 * F * D);
    const bool p3 = c.precision >= 3;         // fp32 end to end (3: fp32 MFMA; 4: three-term fp16 split of the same operands): no fp16 copies, every workspace 4 bytes per element
    if (!p3) CREATE_TRY(hipMalloc(&h->w16, n16 * sizeof(f16)));
    if (c.precision == 1) CREATE_TRY(hipMalloc(&h->w16_lo, n16 * sizeof(f16)));
    float* w32p = nullptr;
    if (c.precision == 4) {
        // every weight once more in the split hi | lo format (same byte size as fp32) + a scratch patch weight
        CREATE_TRY(hipMalloc(&h->w32, (2 * D * 256 + (int64_t)h->L * (4 * D * D + 2 * F * D)) * sizeof(float)));
        w32p = h->w32;
    } else if (p3) {
        CREATE_TRY(hipMalloc(&h->w32, (D * 256 + (int64_t)h->L * 3 * D * D) * sizeof(float)));
        w32p = h->w32;
    }
    uint8_t* w8p = nullptr;
    uint32_t* s8p = nullptr;
    if (c.precision == 2) {
        const int64_t n8 = (int64_t)h->L * (4 * D * D + 2 * F * D);
        CREATE_TRY(hipMalloc(&h->w8, n8));
        CREATE_TRY(hipMalloc(&h->w8_sc, n8 / 32));              // one E8M0 byte per 32 elements
        w8p = h->w8; s8p = h->w8_sc;
    }
    CREATE_TRY(hipMalloc(&h->nonfinite_dev, sizeof(unsigned)));
    CREATE_TRY(hipMemsetAsync(h->nonfinite_dev, 0, sizeof(unsigned), h->compute));
    CREATE_TRY(hipMalloc(&h->qkv_bias_all, (int64_t)h->L * 3 * D * sizeof(float)));
    // on the stream the packing kernels and copies below run on: a null-stream hipMemset may still be in flight when work on a
    // NON-BLOCKING stream (every stream of this library) touches the buffer - found by running training beside the encoder (r5)
    CREATE_TRY(hipMemsetAsync(h->qkv_bias_all, 0, (int64_t)h->L * 3 * D * sizeof(float), h->compute));

    // LayerNorm fold: folded copies of the q|k|v and up_proj weights (+ column sums and biases), fp16 path only
    h->fold_ok = c.precision == 0 && D % 256 == 0 && D <= 1024 && F % 256 == 0;
    f16* wf = nullptr;
    float* fv = nullptr;
    if (h->fold_ok) {
        CREATE_TRY(hipMalloc(&h->w16_fold, (int64_t)h->L * (3 * D * D + F * D) * sizeof(f16)));
        CREATE_TRY(hipMalloc(&h->fold_vec, (int64_t)h->L * 2 * (3 * D + F) * sizeof(float)));
        wf = h->w16_fold; fv = h->fold_vec;
    }

    hipStream_t st = h->compute;
    const float* p = h->blob;
    f16* w = h->w16;
    f16* wl = h->w16_lo;
    auto lo = [&](f16* hi_ptr) -> f16* { return wl ? wl + (hi_ptr - h->w16) : nullptr; };

    // prefix rows of every frame: cls_token (+ its position embedding, DINOv2 [v2]:158-161) | register_tokens
    {
        const int64_t G = c.pos_embed_grid;
        std::vector<float> pre(weights_host, weights_host + (1 + h->R) * D);
        if (G > 0) {
            const float* pos = weights_host + (1 + h->R) * D;
            h->pos_host.assign(pos, pos + (1 + G * G) * D);
            for (int64_t d = 0; d < D; ++d) pre[d] += pos[d];
        }
        CREATE_TRY(hipMalloc(&h->prefix_dev, pre.size() * sizeof(float)));
        CREATE_TRY(hipMemcpy(h->prefix_dev, pre.data(), pre.size() * sizeof(float), hipMemcpyHostToDevice));
        h->prefix = h->prefix_dev;
        p += (1 + h->R) * D + (G > 0 ? (1 + G * G) * D : 0);
    }
    const float* patch_w = p; p += D * 3 * c.patch_size * c.patch_size;
    h->patch_b = p; p += D;
    h->wpatch = w; w += D * 256;
    h->wpatch2 = w; w += D * 512;
    h->wpatch_lo = lo(h->wpatch); h->wpatch2_lo = lo(h->wpatch2);
    // precision 4: per-tensor power-of-two scales from the HOST copy of the weights (same offsets as the device blob)
    auto host_max = [&](const float* dev_ptr, int64_t n) {
        const float* hp = weights_host + (dev_ptr - h->blob);
        float m = 0.f;
        for (int64_t i = 0; i < n; ++i) m = std::max(m, fabsf(hp[i]));
        return m;
    };
    auto pow2_scale = [](float maxabs) {                     // 2^s with maxabs * 2^s in [1, 2); 1 for an all-zero tensor
        if (!(maxabs > 0.f) || !std::isfinite(maxabs)) return 1.0f;
        int e = 0;
        (void)frexpf(maxabs, &e);                            // maxabs = f * 2^e, f in [0.5, 1)
        e = std::min(20, std::max(-20, 1 - e));
        return ldexpf(1.0f, e);
    };
    int rc = 0;
    if (c.precision == 4) {
        const int pp = c.patch_size * c.patch_size;
        const float* hw = weights_host + (patch_w - h->blob);
        float m = 0.f;
        for (int64_t d = 0; d < D; ++d)
            for (int k = 0; k < pp; ++k)
                m = std::max(m, fabsf((hw[d * 3 * pp + k] + hw[d * 3 * pp + pp + k]) + hw[d * 3 * pp + 2 * pp + k]));
        h->sc_patch = pow2_scale(m);
    }
    if (c.precision == 4) {
        h->wpatch32 = w32p;
        rc = launch_pack_patch_weight_f32(patch_w, w32p + D * 256, (int)D, c.patch_size, st);      // fp32 sum over channels
        rc |= launch_pack_split_weight(w32p + D * 256, w32p, D, 256, h->sc_patch, st);
        w32p += 2 * D * 256;
    } else if (p3) {
        h->wpatch32 = w32p;
        rc = launch_pack_patch_weight_f32(patch_w, w32p, (int)D, c.patch_size, st);
        w32p += D * 256;
    } else {
        rc = launch_pack_patch_weight(patch_w, h->wpatch, h->wpatch_lo, h->wpatch2, h->wpatch2_lo, (int)D, c.patch_size, st);
    }

    h->layers.resize(h->L);
    for (int l = 0; l < h->L && !rc; ++l) {
        LayerW& lw = h->layers[l];
        lw.qkv_b = h->qkv_bias_all + (int64_t)l * 3 * D;
        lw.ln1_w = p; p += D;
        lw.ln1_b = p; p += D;
        const float* qw = p; p += D * D;
        const float* qb = p; p += D;
        const float* kw = p; p += D * D;
        const float* kb = p; p += D;
        const float* vw = p; p += D * D;
        const float* vb = p; p += D;
        const float* ow = p; p += D * D;
        lw.o_b = p; p += D;
        lw.ls1 = p; p += D;
        lw.ln2_w = p; p += D;
        lw.ln2_b = p; p += D;
        const float* uw = p; p += F * D;
        lw.up_b = p; p += F;
        const float* dw = p; p += D * F;
        lw.down_b = p; p += D;
        lw.ls2 = p; p += D;
        lw.wqkv = w; w += 3 * D * D;
        lw.wo = w; w += D * D;
        lw.wup = w; w += F * D;
        lw.wdown = w; w += D * F;
        lw.wqkv_lo = lo(lw.wqkv); lw.wo_lo = lo(lw.wo); lw.wup_lo = lo(lw.wup); lw.wdown_lo = lo(lw.wdown);
        if (c.precision == 4) {
            lw.sc_qkv = pow2_scale(std::max(host_max(qw, D * D), std::max(host_max(kw, D * D), host_max(vw, D * D))));
            lw.sc_o = pow2_scale(host_max(ow, D * D));
            lw.sc_up = pow2_scale(host_max(uw, F * D));
            lw.sc_down = pow2_scale(host_max(dw, D * F));
        }
        if (c.precision == 4) {
            rc |= launch_pack_split_weight(qw, w32p, D, (int)D, lw.sc_qkv, st);
            rc |= launch_pack_split_weight(kw, w32p + D * D, D, (int)D, lw.sc_qkv, st);
            rc |= launch_pack_split_weight(vw, w32p + 2 * D * D, D, (int)D, lw.sc_qkv, st);
            lw.wqkv32 = w32p; w32p += 3 * D * D;
            rc |= launch_pack_split_weight(ow, w32p, D, (int)D, lw.sc_o, st);
            lw.wo32 = w32p; w32p += D * D;
            rc |= launch_pack_split_weight(uw, w32p, F, (int)D, lw.sc_up, st);
            lw.wup32 = w32p; w32p += F * D;
            rc |= launch_pack_split_weight(dw, w32p, D, (int)F, lw.sc_down, st);
            lw.wdown32 = w32p; w32p += D * F;
        } else if (p3) {
            // q, k, v sit in the blob with their biases between them: one packed [3D][D] copy; the rest is used in place
            CREATE_TRY(hipMemcpyAsync(w32p, qw, D * D * sizeof(float), hipMemcpyDeviceToDevice, st));
            CREATE_TRY(hipMemcpyAsync(w32p + D * D, kw, D * D * sizeof(float), hipMemcpyDeviceToDevice, st));
            CREATE_TRY(hipMemcpyAsync(w32p + 2 * D * D, vw, D * D * sizeof(float), hipMemcpyDeviceToDevice, st));
            lw.wqkv32 = w32p; w32p += 3 * D * D;
            lw.wo32 = ow; lw.wup32 = uw; lw.wdown32 = dw;
        } else {
            rc |= launch_convert_f16(qw, lw.wqkv, lw.wqkv_lo, D * D, st);
            rc |= launch_convert_f16(kw, lw.wqkv + D * D, lw.wqkv_lo ? lw.wqkv_lo + D * D : nullptr, D * D, st);
            rc |= launch_convert_f16(vw, lw.wqkv + 2 * D * D, lw.wqkv_lo ? lw.wqkv_lo + 2 * D * D : nullptr, D * D, st);
            rc |= launch_convert_f16(ow, lw.wo, lw.wo_lo, D * D, st);
            rc |= launch_convert_f16(uw, lw.wup, lw.wup_lo, F * D, st);
            rc |= launch_convert_f16(dw, lw.wdown, lw.wdown_lo, D * F, st);
        }
        if (c.precision == 2) {
            lw.wqkv8 = w8p; w8p += 3 * D * D;  lw.sqkv = s8p; s8p += 3 * D * D / 128;
            lw.wo8 = w8p; w8p += D * D;        lw.so = s8p; s8p += D * D / 128;
            lw.wup8 = w8p; w8p += F * D;       lw.sup = s8p; s8p += F * D / 128;
            lw.wdown8 = w8p; w8p += D * F;     lw.sdown = s8p; s8p += D * F / 128;
            rc |= launch_pack_fp8_weight(qw, lw.wqkv8, lw.sqkv, (int)D, (int)D, (int)(3 * D), 0, st);
            rc |= launch_pack_fp8_weight(kw, lw.wqkv8 + D * D, lw.sqkv, (int)D, (int)D, (int)(3 * D), (int)D, st);
            rc |= launch_pack_fp8_weight(vw, lw.wqkv8 + 2 * D * D, lw.sqkv, (int)D, (int)D, (int)(3 * D), (int)(2 * D), st);
            rc |= launch_pack_fp8_weight(ow, lw.wo8, lw.so, (int)D, (int)D, (int)D, 0, st);
            rc |= launch_pack_fp8_weight(uw, lw.wup8, lw.sup, (int)F, (int)D, (int)F, 0, st);
            rc |= launch_pack_fp8_weight(dw, lw.wdown8, lw.sdown, (int)D, (int)F, (int)D, 0, st);
        }
        if (h->fold_ok) {
            lw.wqkv_f = wf; wf += 3 * D * D;
            lw.wup_f = wf; wf += F * D;
            lw.qkv_cs = fv; fv += 3 * D;
            lw.qkv_bf = fv; fv += 3 * D;
            lw.up_cs = fv; fv += F;
            lw.up_bf = fv; fv += F;
            rc |= launch_fold_ln_weight(qw, lw.ln1_w, lw.ln1_b, qb, lw.wqkv_f, lw.qkv_cs, lw.qkv_bf, (int)D, (int)D, st);
            rc |= launch_fold_ln_weight(kw, lw.ln1_w, lw.ln1_b, kb, lw.wqkv_f + D * D, lw.qkv_cs + D, lw.qkv_bf + D, (int)D, (int)D, st);
            rc |= launch_fold_ln_weight(vw, lw.ln1_w, lw.ln1_b, vb, lw.wqkv_f + 2 * D * D, lw.qkv_cs + 2 * D, lw.qkv_bf + 2 * D, (int)D, (int)D, st);
            rc |= launch_fold_ln_weight(uw, lw.ln2_w, lw.ln2_b, lw.up_b, lw.wup_f, lw.up_cs, lw.up_bf, (int)F, (int)D, st);
        }
        CREATE_TRY(hipMemcpyAsync(lw.qkv_b, qb, D * sizeof(float), hipMemcpyDeviceToDevice, st));
        CREATE_TRY(hipMemcpyAsync(lw.qkv_b + D, kb, D * sizeof(float), hipMemcpyDeviceToDevice, st));
        CREATE_TRY(hipMemcpyAsync(lw.qkv_b + 2 * D, vb, D * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    h->norm_w = p; p += D;
    h->norm_b = p; p += D;
    if (rc || (p - h->blob) != n_weights) {
        cbas_fail(CBAS_EHIP, "weight packing failed (rc=%d, consumed %lld of %lld)", rc, (long long)(p - h->blob),
                  (long long)n_weights);
        cbas_enc_destroy(h);
        return CBAS_EHIP;
    }

    // workspaces
    const int64_t Pmax = (int64_t)(c.max_height / c.patch_size) * (c.max_width / c.patch_size);
    const int64_t Tmax = Pmax + h->NP;
    h->rows_cap = round_up((int64_t)c.max_batch * Tmax, 128);
    h->prow_cap = round_up((int64_t)c.max_batch * Pmax, 128);
    h->rope_cap = (int)Pmax;
    h->pos_tables.reserve(cbas_enc::POS_TABLES_MAX);     // entries are handed out by pointer: never reallocate
    const size_t esz = p3 ? sizeof(float) : sizeof(f16);      // activation element size (A_patch is 512 f16 = 256 f32 per row)
    CREATE_TRY(hipMalloc(&h->A_patch, h->prow_cap * 512 * sizeof(f16)));
    CREATE_TRY(hipMalloc(&h->x, h->rows_cap * D * sizeof(float)));
    CREATE_TRY(hipMalloc(&h->h16, h->rows_cap * D * esz));
    CREATE_TRY(hipMalloc(&h->qkv16, h->rows_cap * 3 * D * esz));
    CREATE_TRY(hipMalloc(&h->u16, h->rows_cap * F * esz));
    const int64_t sch_elems = (D / 128) * h->rows_cap, scu_elems = (F / 128) * h->rows_cap;    // dwords
    if (c.precision == 2) {
        CREATE_TRY(hipMalloc(&h->sc_h, sch_elems * 4));
        CREATE_TRY(hipMalloc(&h->sc_u, scu_elems * 4));
        CREATE_TRY(hipMemsetAsync(h->sc_h, 0, sch_elems * 4, st));
        CREATE_TRY(hipMemsetAsync(h->sc_u, 0, scu_elems * 4, st));
    }
    if (h->fold_ok) {
        CREATE_TRY(hipMalloc(&h->x16, h->rows_cap * D * sizeof(f16)));
        CREATE_TRY(hipMalloc(&h->lnst, 4 * h->rows_cap * sizeof(float2)));
        CREATE_TRY(hipMemsetAsync(h->x16, 0, h->rows_cap * D * sizeof(f16), st));
        CREATE_TRY(hipMemsetAsync(h->lnst, 0, 4 * h->rows_cap * sizeof(float2), st));
    }
    const int64_t cls_elems = round_up(c.max_batch, 128) * (3 * D + F);
    CREATE_TRY(hipMalloc(&h->cls16, cls_elems * esz));
    CREATE_TRY(hipMemsetAsync(h->cls16, 0, cls_elems * esz, st));
    CREATE_TRY(hipMemsetAsync(h->A_patch, 0, h->prow_cap * 512 * sizeof(f16), st));
    CREATE_TRY(hipMemsetAsync(h->x, 0, h->rows_cap * D * sizeof(float), st));
    CREATE_TRY(hipMemsetAsync(h->h16, 0, h->rows_cap * D * esz, st));
    CREATE_TRY(hipMemsetAsync(h->qkv16, 0, h->rows_cap * 3 * D * esz, st));
    CREATE_TRY(hipMemsetAsync(h->u16, 0, h->rows_cap * F * esz, st));
    // second compute lane (see cbas_enc::Lane); CBAS_LANES=1 (read here, once per handle) keeps a single lane
    {
        const char* e = getenv("CBAS_LANES");
        h->n_lanes = (e && atoi(e) == 1) ? 1 : 2;
        cbas_enc::Lane& L0 = h->lanes[0];
        L0.A_patch = h->A_patch; L0.x = h->x; L0.h16 = h->h16; L0.qkv16 = h->qkv16; L0.u16 = h->u16; L0.cls16 = h->cls16; L0.sc_h = h->sc_h; L0.sc_u = h->sc_u; L0.stream = h->compute;
        L0.x16 = h->x16; L0.lnst = h->lnst;
        if (h->n_lanes == 2) {
            cbas_enc::Lane& L1 = h->lanes[1];
            CREATE_TRY(hipMalloc(&L1.A_patch, h->prow_cap * 512 * sizeof(f16)));
            CREATE_TRY(hipMalloc(&L1.x, h->rows_cap * D * sizeof(float)));
            CREATE_TRY(hipMalloc(&L1.h16, h->rows_cap * D * esz));
            CREATE_TRY(hipMalloc(&L1.qkv16, h->rows_cap * 3 * D * esz));
            CREATE_TRY(hipMalloc(&L1.u16, h->rows_cap * F * esz));
            if (c.precision == 2) {
                CREATE_TRY(hipMalloc(&L1.sc_h, sch_elems * 4));
                CREATE_TRY(hipMalloc(&L1.sc_u, scu_elems * 4));
                CREATE_TRY(hipMemsetAsync(L1.sc_h, 0, sch_elems * 4, st));
                CREATE_TRY(hipMemsetAsync(L1.sc_u, 0, scu_elems * 4, st));
            }
            if (h->fold_ok) {
                CREATE_TRY(hipMalloc(&L1.x16, h->rows_cap * D * sizeof(f16)));
                CREATE_TRY(hipMalloc(&L1.lnst, 4 * h->rows_cap * sizeof(float2)));
                CREATE_TRY(hipMemsetAsync(L1.x16, 0, h->rows_cap * D * sizeof(f16), st));
                CREATE_TRY(hipMemsetAsync(L1.lnst, 0, 4 * h->rows_cap * sizeof(float2), st));
            }
            CREATE_TRY(hipMalloc(&L1.cls16, cls_elems * esz));
            CREATE_TRY(hipMemsetAsync(L1.cls16, 0, cls_elems * esz, st));
            CREATE_TRY(hipMemsetAsync(L1.A_patch, 0, h->prow_cap * 512 * sizeof(f16), st));
            CREATE_TRY(hipMemsetAsync(L1.x, 0, h->rows_cap * D * sizeof(float), st));
            CREATE_TRY(hipMemsetAsync(L1.h16, 0, h->rows_cap * D * esz, st));
            CREATE_TRY(hipMemsetAsync(L1.qkv16, 0, h->rows_cap * 3 * D * esz, st));
            CREATE_TRY(hipMemsetAsync(L1.u16, 0, h->rows_cap * F * esz, st));
            CREATE_TRY(hipStreamCreateWithFlags(&L1.stream, hipStreamNonBlocking));
        }
        CREATE_TRY(hipEventCreateWithFlags(&h->lane0_async_done, hipEventDisableTiming));
        CREATE_TRY(hipEventCreateWithFlags(&h->sync_done, hipEventDisableTiming));
    }

    // host-streaming slots
    h->slot_bytes = (int64_t)c.max_batch * c.max_height * c.max_width * 4;
    for (Slot& s : h->slots) {
        CREATE_TRY(hipHostMalloc(&s.in_host, h->slot_bytes, hipHostMallocDefault));
        CREATE_TRY(hipHostMalloc(&s.out16_host, (int64_t)c.max_batch * D * 2, hipHostMallocDefault));
        CREATE_TRY(hipHostMalloc(&s.out32_host, (int64_t)c.max_batch * D * 4, hipHostMallocDefault));
        CREATE_TRY(hipMalloc(&s.in_dev, h->slot_bytes));
        CREATE_TRY(hipMalloc(&s.out16_dev, (int64_t)c.max_batch * D * 2));
        CREATE_TRY(hipMalloc(&s.out32_dev, (int64_t)c.max_batch * D * 4));
        CREATE_TRY(hipEventCreateWithFlags(&s.ev_copied, hipEventDisableTiming));
        CREATE_TRY(hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming));
        CREATE_TRY(hipEventCreateWithFlags(&s.ev_in, hipEventDisableTiming));
    }
    CREATE_TRY(hipStreamSynchronize(st));
#undef CREATE_TRY
    *out = h;
    return CBAS_OK;
}

// a synchronous call is about to use lane 0's workspace on stream st
static int sync_enter(cbas_enc* h, hipStream_t st) {
    if (h->lane0_async_used) HIP_TRY(hipStreamWaitEvent(st, h->lane0_async_done, 0));
    // two synchronous calls on DIFFERENT caller streams share lane 0's workspace too (free when st is the same stream)
    if (h->sync_used) HIP_TRY(hipStreamWaitEvent(st, h->sync_done, 0));
    return CBAS_OK;
}
static int sync_leave(cbas_enc* h, hipStream_t st) {
    HIP_TRY(hipEventRecord(h->sync_done, st));
    h->sync_used = true;
    return CBAS_OK;
}
// an asynchronous batch is about to use `lane` on its own stream ls / has been queued there
static int async_enter(cbas_enc* h, int lane, hipStream_t ls) {
    if (lane == 0 && h->sync_used) HIP_TRY(hipStreamWaitEvent(ls, h->sync_done, 0));
    return CBAS_OK;
}
static int async_leave(cbas_enc* h, int lane, hipStream_t ls) {
    if (lane == 0) { HIP_TRY(hipEventRecord(h->lane0_async_done, ls)); h->lane0_async_used = true; }
    return CBAS_OK;
}

extern "C" int cbas_enc_forward_u8(cbas_enc* h, const uint8_t* frames_dev, int n, int height, int width,
                                   int64_t frame_stride, int64_t row_stride, int64_t pixel_stride,
                                   float* cls_f32_dev, uint16_t* cls_f16_dev, void* stream) {
    if (!h) return cbas_fail(CBAS_EINVAL, "null encoder handle");
    hipStream_t st = (hipStream_t)stream;
    int rc = sync_enter(h, st);
    if (rc) return rc;
    rc = forward_u8(h, frames_dev, n, height, width, frame_stride, row_stride, pixel_stride, cls_f32_dev,
                    (f16*)cls_f16_dev, st, -1, -1);
    if (rc) return rc;
    return sync_leave(h, st);
}

extern "C" int cbas_enc_forward_f32(cbas_enc* h, const float* x_dev, int n, int height, int width,
                                    float* cls_f32_dev, uint16_t* cls_f16_dev, void* stream) {
    int rc = check_frame(h, n, height, width);
    if (rc) return rc;
    if (!x_dev) return cbas_fail(CBAS_EINVAL, "x_dev is NULL");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    const int ps = h->cfg.patch_size;
    const int T = (height / ps) * (width / ps) + h->NP;
    rc = sync_enter(h, st);
    if (rc) return rc;
    if (h->cfg.precision >= 3)
        LAUNCH_TRY(launch_im2col_f32_f32(x_dev, n, height, width, reinterpret_cast<float*>(h->A_patch), h->x, h->prefix, h->NP, h->D, T, ps,
                                         h->cfg.precision == 4, st));
    else
        LAUNCH_TRY(launch_im2col_f32(x_dev, n, height, width, h->A_patch, h->x, h->prefix, h->NP, h->D, T, ps, st));
    rc = run_blocks(h, n, height, width, 512, 1.0f, cls_f32_dev, (f16*)cls_f16_dev, st, -1, -1);
    if (rc) return rc;
    return sync_leave(h, st);
}

#if CBAS_BUILD_DEBUG      // stage taps: debug build only (include/cbas_mi355x_debug.h)
extern "C" int cbas_enc_debug_forward_u8(cbas_enc* h, const uint8_t* frames_dev, int n, int height, int width,
                                         int64_t frame_stride, int64_t row_stride, int64_t pixel_stride,
                                         int stop_layer, int stop_stage) {
    if (!h) return cbas_fail(CBAS_EINVAL, "null encoder handle");
    int rc = forward_u8(h, frames_dev, n, height, width, frame_stride, row_stride, pixel_stride, nullptr, nullptr,
                        h->compute, stop_layer, stop_stage);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(h->compute));
    return CBAS_OK;
}

extern "C" int cbas_enc_debug_read(cbas_enc* h, int which, void* host_out, int64_t n_bytes) {
    if (!h || !host_out) return cbas_fail(CBAS_EINVAL, "null argument");
    const void* src = nullptr;
    int64_t cap = 0;
    switch (which) {
        case 0: src = h->x; cap = h->rows_cap * h->D * 4; break;
        // precision 3 keeps these as fp32 (4 bytes per element)
        case 1: src = h->h16; cap = h->rows_cap * h->D * (h->cfg.precision >= 3 ? 4 : 2); break;
        case 2: src = h->qkv16; cap = h->rows_cap * 3 * h->D * (h->cfg.precision >= 3 ? 4 : 2); break;
        case 3: src = h->u16; cap = h->rows_cap * h->F * (h->cfg.precision >= 3 ? 4 : 2); break;
        default: return cbas_fail(CBAS_EINVAL, "unknown buffer %d", which);
    }
    if (n_bytes < 0 || n_bytes > cap) return cbas_fail(CBAS_EINVAL, "read of %lld bytes exceeds buffer (%lld)", (long long)n_bytes, (long long)cap);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->compute));
    HIP_TRY(hipMemcpy(host_out, src, n_bytes, hipMemcpyDeviceToHost));
    return CBAS_OK;
}
#endif

// cls_*_dev == nullptr: rows go to the slot's own buffers and on to pinned host memory (cbas_enc_wait);
// otherwise rows are written to the caller's device buffers and the slot is released by cbas_enc_wait_stream.
static int submit_u8_host_impl(cbas_enc* h, int slot, const uint8_t* frames_host, int n, int height, int width,
                               int64_t frame_stride, int64_t row_stride, int64_t pixel_stride, float* cls_f32_dev,
                               f16* cls_f16_dev, bool to_device) {
    int rc = check_frame(h, n, height, width);
    if (rc) return rc;
    if (slot < 0 || slot >= CBAS_ENC_SLOTS) return cbas_fail(CBAS_EINVAL, "slot %d out of range", slot);
    if (!frames_host) return cbas_fail(CBAS_EINVAL, "frames_host is NULL");
    Slot& s = h->slots[slot];
    if (s.busy) return cbas_fail(CBAS_ESTATE, "slot %d is busy; call cbas_enc_wait first", slot);
    const int64_t plane = (int64_t)height * width;
    if (frame_stride < 0 || row_stride < 0 || pixel_stride < 1) return cbas_fail(CBAS_EINVAL, "negative stride");
    HIP_TRY(hipSetDevice(h->device));
    // Bytes of the caller's layout are shipped AS THEY ARE (for decord's (n,H,W,3) RGB: 3 bytes per pixel, trivial on
    // PCIe 5) and the consumed channel is picked by the ingest kernel through the strides, so the host does no
    // per-pixel work.  `ext` = bytes from a frame's first consumed pixel to its last.
    if (s.used) {
        // A slot released by cbas_enc_wait_stream was never waited for on the HOST: its previous H2D copy may still be
        // reading the pinned staging, and its previous batch may still be reading in_dev.
        HIP_TRY(hipEventSynchronize(s.ev_copied));
        HIP_TRY(hipStreamWaitEvent(h->copy, s.ev_done, 0));
    }
    const int64_t ext = (int64_t)(height - 1) * row_stride + (int64_t)(width - 1) * pixel_stride + 1;
    const bool dense = n == 1 || frame_stride >= ext;               // frames do not interleave
    int64_t dev_frame_stride = plane, dev_row_stride = width, dev_pixel_stride = 1, bytes = (int64_t)n * plane;
    const uint8_t* src_host = s.in_host;
    if (dense && frame_stride <= ext + 64 && (int64_t)(n - 1) * frame_stride + ext <= h->slot_bytes) {
        // one contiguous span holds the whole chunk
        bytes = (int64_t)(n - 1) * frame_stride + ext;
        dev_frame_stride = frame_stride; dev_row_stride = row_stride; dev_pixel_stride = pixel_stride;
        hipPointerAttribute_t attr;
        const bool pinned = hipPointerGetAttributes(&attr, frames_host) == hipSuccess && attr.type == hipMemoryTypeHost;
        if (!pinned) (void)hipGetLastError();                         // pageable memory: the query fails, clear it
        if (pinned) src_host = frames_host;                           // direct DMA; caller keeps it valid until wait
        else memcpy(s.in_host, frames_host, bytes);
    } else if (dense && (int64_t)n * ext <= h->slot_bytes) {
        // strided frames: one span per frame, packed back to back
        for (int f = 0; f < n; ++f) memcpy(s.in_host + (int64_t)f * ext, frames_host + (int64_t)f * frame_stride, ext);
        bytes = (int64_t)n * ext;
        dev_frame_stride = ext; dev_row_stride = row_stride; dev_pixel_stride = pixel_stride;
    } else {
        // sparse layouts (huge strides): gather the consumed channel into packed planes on the host
        if ((int64_t)n * plane > h->slot_bytes) return cbas_fail(CBAS_EINVAL, "chunk exceeds slot staging size");
        for (int f = 0; f < n; ++f) {
            const uint8_t* src = frames_host + (int64_t)f * frame_stride;
            uint8_t* dst = s.in_host + (int64_t)f * plane;
            for (int y = 0; y < height; ++y) {
                const uint8_t* r = src + (int64_t)y * row_stride;
                uint8_t* d = dst + (int64_t)y * width;
                if (pixel_stride == 1) memcpy(d, r, width);
                else for (int xx = 0; xx < width; ++xx) d[xx] = r[(int64_t)xx * pixel_stride];
            }
        }
    }
    HIP_TRY(hipMemcpyAsync(s.in_dev, src_host, bytes, hipMemcpyHostToDevice, h->copy));
    HIP_TRY(hipEventRecord(s.ev_copied, h->copy));
    const int lane = (int)(h->submit_count++ % (uint64_t)h->n_lanes);
    hipStream_t ls = h->lanes[lane].stream;
    HIP_TRY(hipStreamWaitEvent(ls, s.ev_copied, 0));
    rc = async_enter(h, lane, ls);
    if (rc) return rc;
    use_lane(h, lane);
    rc = forward_u8(h, s.in_dev, n, height, width, dev_frame_stride, dev_row_stride, dev_pixel_stride,
                    to_device ? cls_f32_dev : s.out32_dev, to_device ? cls_f16_dev : s.out16_dev, ls, -1, -1);
    use_lane(h, 0);
    if (rc) return rc;
    rc = async_leave(h, lane, ls);
    if (rc) return rc;
    if (!to_device) {
        HIP_TRY(hipMemcpyAsync(s.out16_host, s.out16_dev, (int64_t)n * h->D * 2, hipMemcpyDeviceToHost, ls));
        HIP_TRY(hipMemcpyAsync(s.out32_host, s.out32_dev, (int64_t)n * h->D * 4, hipMemcpyDeviceToHost, ls));
    }
    HIP_TRY(hipEventRecord(s.ev_done, ls));
    s.n = n;
    s.busy = true;
    s.used = true;
    s.dev_mode = to_device;
    return CBAS_OK;
}

extern "C" int cbas_enc_submit_u8_host(cbas_enc* h, int slot, const uint8_t* frames_host, int n, int height,
                                       int width, int64_t frame_stride, int64_t row_stride, int64_t pixel_stride) {
    return submit_u8_host_impl(h, slot, frames_host, n, height, width, frame_stride, row_stride, pixel_stride, nullptr,
                               nullptr, false);
}

extern "C" int cbas_enc_submit_u8_host_dev(cbas_enc* h, int slot, const uint8_t* frames_host, int n, int height,
                                           int width, int64_t frame_stride, int64_t row_stride, int64_t pixel_stride,
                                           float* cls_f32_dev, uint16_t* cls_f16_dev, void* after_stream) {
    if (!h) return cbas_fail(CBAS_EINVAL, "null encoder handle");
    if (!cls_f32_dev && !cls_f16_dev) return cbas_fail(CBAS_EINVAL, "no output requested");
    if (slot >= 0 && slot < CBAS_ENC_SLOTS && !h->slots[slot].busy) {
        // the output rows may still be read by work queued on after_stream (e.g. the head over an earlier clip)
        HIP_TRY(hipSetDevice(h->device));
        HIP_TRY(hipEventRecord(h->slots[slot].ev_in, (hipStream_t)after_stream));
        HIP_TRY(hipStreamWaitEvent(h->copy, h->slots[slot].ev_in, 0));
    }
    return submit_u8_host_impl(h, slot, frames_host, n, height, width, frame_stride, row_stride, pixel_stride,
                               cls_f32_dev, (f16*)cls_f16_dev, true);
}

extern "C" void* cbas_enc_copy_stream(cbas_enc* h) { return h ? (void*)h->copy : nullptr; }

extern "C" int cbas_enc_get_config(const cbas_enc* h, cbas_enc_config* out) {
    if (!h || !out) return cbas_fail(CBAS_EINVAL, "null argument");
    *out = h->cfg;
    return CBAS_OK;
}

extern "C" int cbas_enc_submit_u8(cbas_enc* h, int slot, const uint8_t* frames_dev, int n, int height, int width,
                                  int64_t frame_stride, int64_t row_stride, int64_t pixel_stride, float* cls_f32_dev,
                                  uint16_t* cls_f16_dev, void* after_stream) {
    int rc = check_frame(h, n, height, width);
    if (rc) return rc;
    if (slot < 0 || slot >= CBAS_ENC_SLOTS) return cbas_fail(CBAS_EINVAL, "slot %d out of range", slot);
    if (!frames_dev) return cbas_fail(CBAS_EINVAL, "frames_dev is NULL");
    if (!cls_f32_dev && !cls_f16_dev) return cbas_fail(CBAS_EINVAL, "no output requested");
    Slot& s = h->slots[slot];
    if (s.busy) return cbas_fail(CBAS_ESTATE, "slot %d is busy; call cbas_enc_wait_stream first", slot);
    HIP_TRY(hipSetDevice(h->device));
    const int lane = (int)(h->submit_count++ % (uint64_t)h->n_lanes);
    hipStream_t ls = h->lanes[lane].stream;
    HIP_TRY(hipEventRecord(s.ev_in, (hipStream_t)after_stream));      // the frames (and the output rows) are ready
    HIP_TRY(hipStreamWaitEvent(ls, s.ev_in, 0));
    rc = async_enter(h, lane, ls);
    if (rc) return rc;
    use_lane(h, lane);
    rc = forward_u8(h, frames_dev, n, height, width, frame_stride, row_stride, pixel_stride, cls_f32_dev,
                    (f16*)cls_f16_dev, ls, -1, -1);
    use_lane(h, 0);
    if (rc) return rc;
    rc = async_leave(h, lane, ls);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(s.ev_done, ls));
    s.n = n;
    s.busy = true;
    s.dev_mode = true;
    return CBAS_OK;
}

extern "C" int cbas_enc_set_lanes(cbas_enc* h, int n_lanes) {
    if (!h) return cbas_fail(CBAS_EINVAL, "null encoder handle");
    if (n_lanes < 1 || n_lanes > 2 || (n_lanes == 2 && !h->lanes[1].stream))
        return cbas_fail(CBAS_EINVAL, "n_lanes=%d not available (handle was created with %s)", n_lanes,
                         h->lanes[1].stream ? "2 lanes" : "CBAS_LANES=1");
    for (const Slot& s : h->slots)
        if (s.busy) return cbas_fail(CBAS_ESTATE, "cbas_enc_set_lanes with a batch in flight");
    h->n_lanes = n_lanes;
    h->submit_count = 0;
    return CBAS_OK;
}

extern "C" int cbas_enc_set_prune_last_layer(cbas_enc* h, int enable) {
    if (!h) return cbas_fail(CBAS_EINVAL, "null encoder handle");
    h->prune_last = enable != 0;
    return CBAS_OK;
}

#if CBAS_BUILD_DEBUG
extern "C" int cbas_enc_debug_option(cbas_enc* h, const char* name, int value) {
    if (!h || !name) return cbas_fail(CBAS_EINVAL, "null argument");
    if (!strcmp(name, "rope_lds")) { h->rope_in_lds = value != 0; return CBAS_OK; }
    if (!strcmp(name, "ln_fold")) { h->ln_fold = value != 0; return CBAS_OK; }
    if (!strcmp(name, "split_kernels")) { vit32_split_set_forms(value); return CBAS_OK; }       // process-wide (precision 4)
    return cbas_fail(CBAS_EINVAL, "unknown debug option '%s'", name);
}
#endif

extern "C" int cbas_enc_wait_stream(cbas_enc* h, int slot, void* stream) {
    if (!h) return cbas_fail(CBAS_EINVAL, "null encoder handle");
    if (slot < 0 || slot >= CBAS_ENC_SLOTS) return cbas_fail(CBAS_EINVAL, "slot %d out of range", slot);
    Slot& s = h->slots[slot];
    if (!s.busy) return cbas_fail(CBAS_ESTATE, "slot %d has no submitted work", slot);
    if (!s.dev_mode) return cbas_fail(CBAS_ESTATE, "slot %d holds a host submission; use cbas_enc_wait", slot);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, s.ev_done, 0));
    s.busy = false;
    return CBAS_OK;
}

// Frames whose CLS row came out NaN / infinite since the last call (the counter is cleared): CBAS_ERANGE when there are any.
// Meaningful after the batches in question have completed (cbas_enc_wait and the fused session's waits call it themselves).
extern "C" int cbas_enc_check_finite(cbas_enc* h) {
    if (!h) return cbas_fail(CBAS_EINVAL, "null encoder handle");
    HIP_TRY(hipSetDevice(h->device));
    unsigned n = 0;
    HIP_TRY(hipMemcpyAsync(&n, h->nonfinite_dev, sizeof(n), hipMemcpyDeviceToHost, h->aux));
    HIP_TRY(hipStreamSynchronize(h->aux));
    if (!n) return CBAS_OK;
    HIP_TRY(hipMemsetAsync(h->nonfinite_dev, 0, sizeof(unsigned), h->aux));
    HIP_TRY(hipStreamSynchronize(h->aux));
    return cbas_fail(CBAS_ERANGE, "%u frame(s) produced a non-finite CLS row: an activation left the range of precision %d%s",
                     n, h->cfg.precision,
                     h->cfg.precision == 4 ? " (operands are split into fp16 halves after power-of-two scaling: |GELU output| < 16 376, "
                                             "|k|, |v| < 16 376, ... - include/cbas_mi355x.h); precision 3 (CBAS_PRECISION=3) computes "
                                             "the same rows in fp32 without a range limit"
                                           : h->cfg.precision == 3 ? "" : " (fp16 activations: |value| < 65 504); precisions 3 / 4 keep them in fp32");
}

extern "C" int cbas_enc_wait(cbas_enc* h, int slot, uint16_t* cls_f16_host, float* cls_f32_host) {
    if (!h) return cbas_fail(CBAS_EINVAL, "null encoder handle");
    if (slot < 0 || slot >= CBAS_ENC_SLOTS) return cbas_fail(CBAS_EINVAL, "slot %d out of range", slot);
    Slot& s = h->slots[slot];
    if (!s.busy) return cbas_fail(CBAS_ESTATE, "slot %d has no submitted work", slot);
    if (s.dev_mode) return cbas_fail(CBAS_ESTATE, "slot %d holds a device submission; use cbas_enc_wait_stream", slot);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipEventSynchronize(s.ev_done));
    if (cls_f16_host) memcpy(cls_f16_host, s.out16_host, (int64_t)s.n * h->D * 2);
    if (cls_f32_host) memcpy(cls_f32_host, s.out32_host, (int64_t)s.n * h->D * 4);
    s.busy = false;
    return cbas_enc_check_finite(h);
}

extern "C" int cbas_enc_profile(cbas_enc* h, int enable) {
    if (!h) return cbas_fail(CBAS_EINVAL, "null encoder handle");
    h->prof_on = enable != 0;
    return CBAS_OK;
}

extern "C" int cbas_enc_profile_read(cbas_enc* h, double* ms_by_cat, int64_t* launches_by_cat, double* flops_by_cat,
                                     int reset) {
    if (!h || !ms_by_cat || !launches_by_cat || !flops_by_cat) return cbas_fail(CBAS_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    for (int c = 0; c < CBAS_PROF_NCAT; ++c) { ms_by_cat[c] = 0; launches_by_cat[c] = 0; flops_by_cat[c] = 0; }
    for (size_t i = 0; i < h->prof_used; ++i) {
        const auto& r = h->prof[i];
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, r.a, r.b));
        ms_by_cat[r.cat] += ms;
        launches_by_cat[r.cat] += 1;
        flops_by_cat[r.cat] += r.flops;
    }
    if (reset) h->prof_used = 0;
    return CBAS_OK;
}

// Host-callable launchers of the HIP kernels (internal to libcbas_mi355x; not part of the C ABI).
#pragma once
#include "common.h"

// ---------------------------------------------------------------------------------------------
// fp16 x fp16 -> fp32 MFMA GEMM:  C[M,N] = A[M,K] * W[N,K]^T  with fused epilogues.
// A and W are row-major with K contiguous (the HF Linear weight layout, no transpose needed).
// ---------------------------------------------------------------------------------------------
enum GemmEpilogue {
    EPI_PATCH = 0,   // x[b*T + n_prefix + p][n] = acc*in_scale + bias[n]           (fp32 out)
    EPI_QKV   = 1,   // + bias, RoPE on patch rows of q and k, q *= 1/8              (fp16 out)
    EPI_RESID = 2,   // x[m][n] += (acc + bias[n]) * lambda[n]                        (fp32 in/out)
    EPI_GELU  = 3,   // u[m][n] = gelu_erf(acc + bias[n])                             (fp16 out)
    EPI_GELU_F8 = 4, // the same, stored as MX-fp8: e4m3 bytes + one E8M0 scale per 32 columns (precision 2)
    // LayerNorm folded into the GEMMs around it (gemm_f16_8ph.hip only; GemmParams "LayerNorm fold" fields):
    EPI_QKV_LN = 5,   // EPI_QKV on raw fp16 x with W = fp16(gamma o W): rstd * (acc - mean * colsum) + bias', then as EPI_QKV
    EPI_GELU_LN = 6,  // EPI_GELU likewise
    EPI_RESID_LN = 7, // EPI_RESID that also writes fp16 x and the per-row statistics the next EPI_*_LN needs
};
constexpr int epi_base(int epi) { return epi == EPI_QKV_LN ? EPI_QKV : epi == EPI_GELU_LN ? EPI_GELU : epi == EPI_RESID_LN ? EPI_RESID : epi; }

enum GemmTile { GEMM_TILE_AUTO = 0, GEMM_TILE_128x128 = 1, GEMM_TILE_256x128 = 2, GEMM_TILE_128x256 = 3,
                GEMM_TILE_256x256 = 4,
                GEMM_TILE_192x256 = 7, GEMM_TILE_64x128 = 8, GEMM_TILE_SKINNY = 9,
                GEMM_TILE_PP_256x256 = 13,      // gemm_f16_8ph.hip: 8 waves, ping-pong phases, counted vmcnt
                GEMM_TILE_PP_192x256 = 14, GEMM_TILE_PP_160x256 = 15, GEMM_TILE_PP_128x256 = 16,
                GEMM_TILE_PP_AUTO = 17 };       // planner: uniform tile height, or 256-row tiles + 128-row tail

constexpr int GEMM_STAMP_BLOCKS = 16384;     // GemmParams::stamps: [GEMM_STAMP_BLOCKS][4] s_memtime stamps of a workgroup's first tile + [GEMM_STAMP_BLOCKS] loop spans in s_memrealtime ticks + [GEMM_STAMP_BLOCKS][4] stamps of its last tile
constexpr int ROPE_LDS_ROWS = 192;            // (nh + nw) rows of 128 bytes the q|k|v kernel holds in LDS (24 KiB)
struct GemmParams {
    int tile;            // GemmTile (0 = pick by shape)
    int group_m;         // raster: row-panels per group (0/1 = N-fastest order)
    unsigned long long* stamps;   // bring-up only: per-block s_memtime stamps [grid][4], or nullptr
    const f16* A;        // [M_pad][lda]
    int lda;             // row stride of A in elements (0: K).  lda = T*D reads one row per frame (the CLS rows)
    const f16* W;        // [N][K]  (hi part when split)
    const f16* W_lo;     // [N][K]  residual W - fp16(W) as fp16, or nullptr
    // MX-fp8 operands (precision 2; gemm_f16_8ph.hip only): e4m3 bytes, K contiguous, plus E8M0 block scales, one
    // byte per 32 k-elements, stored K-tile-major as dwords: sc[(k / 128) * ld + row] byte (k % 128) / 32.
    // A8 != nullptr selects the fp8 kernel (A / W above are then unused).
    const uint8_t* A8;   // [M_pad][lda]
    const uint8_t* W8;   // [N][K]
    const uint32_t* A_sc;   // [K/128][sc_lda]
    const uint32_t* W_sc;   // [K/128][sc_ldw]
    int sc_lda;
    int sc_ldw;          // 0: N (W_sc may point into a wider array, e.g. the k|v columns of the fused q|k|v scales)
    uint8_t* out_f8;     // EPI_GELU_F8: [M][ldo] bytes
    uint32_t* out_sc;    // EPI_GELU_F8: [N/128][sc_ldo]
    int sc_ldo;
    int M;               // valid rows (stores are skipped for rows >= M)
    int M_pad;           // rows allocated in A (loads of rows >= M_pad are clamped)
    int N;               // multiple of 128
    int K;               // multiple of 64
    const float* bias;   // [N]
    const float* lambda; // [N]   (EPI_RESID)
    float* out_f32;      // EPI_PATCH / EPI_RESID: residual stream x, leading dim ldo
    f16* out_f16;        // EPI_QKV / EPI_GELU
    int ldo;
    // EPI_PATCH
    int patches_per_frame;   // P
    int tokens_per_frame;    // T
    int n_prefix;            // 1 + R
    float in_scale;          // 1/255 for uint8 input, 1 for float input
    const float* pos;        // [P][N] additive position embedding for the frame's patch grid, or nullptr (DINOv3)
    // EPI_QKV
    const float* rope_cos;   // [P][64], or nullptr: no RoPE (DINOv2)
    const float* rope_sin;   // [P][64]
    // the same numbers factorised by axis (columns 0-15 of a table row depend on the patch row only, 16-31 on the patch
    // column only): [nh + nw][32] floats, row r < nh = {cos(16) | sin(16)} of patch row r, row nh + c = those of patch
    // column c.  The ping-pong kernel keeps this copy in LDS (nullptr, or more than ROPE_LDS_ROWS rows: global table).
    const float* rope_fac;
    int rope_nh, rope_nw;
    unsigned rope_magic;     // floor(2^32 / rope_nw) + 1: patch row = umulhi(patch, rope_magic)
    int D;                   // hidden size (q | k | v sections of width D)
    int sec0;                // section of output column 0 (0: the full q|k|v GEMM; 1: W holds only k|v)
    // ---- LayerNorm folded into the GEMMs around it (gemm_f16_8ph.hip only; see "LayerNorm fold" in api_enc.hip) ----
    // producer side, EPI_RESID_LN: besides the fp32 residual stream the epilogue writes its fp16 copy (the NEXT GEMM's A
    // operand, un-normalised) and, per row and 256-column block (= one N tile), the statistics LayerNorm needs of it:
    //   ln_out[(col / 256) * ln_ld + m] = { sum of the block's 256 values, sum of squares about the BLOCK mean }
    f16* x16_out;            // [M][N]
    float2* ln_out;          // [N / 256][ln_ld]
    // consumer side, EPI_QKV_LN / EPI_GELU_LN: A holds raw fp16 x, W = fp16(gamma o W), bias = beta W^T + b; the epilogue
    // forms rstd * (acc - mean * ln_colsum[n]) + bias[n] with mean / rstd pooled from the row's ln_parts blocks
    // (exact pooled-variance formula: as robust as the two-pass form of layernorm_f16_kernel)
    const float2* ln_in;     // [ln_parts][ln_ld]
    const float* ln_colsum;  // [N]: sum over k of the fp16 folded weights
    int ln_parts;            // K / 256 (<= 4)
    int ln_ld;               // row stride of ln_in / ln_out (>= M)
    float ln_eps;
};

int launch_gemm(GemmEpilogue epi, const GemmParams& p, hipStream_t stream);

int launch_gemm_8ph(GemmEpilogue epi, const GemmParams& p, int tile, hipStream_t stream);
int launch_gemm_skinny(GemmEpilogue epi, const GemmParams& p, hipStream_t stream);     // M <= 64 (gemm_f16_skinny.hip)

// ---------------------------------------------------------------------------------------------
// ViT element-wise / attention kernels
// ---------------------------------------------------------------------------------------------
// uint8 pixels -> im2col matrix A[n*P][256] fp16 holding the INTEGER pixel values (exact in fp16);
// also writes the CLS/register prefix rows of the residual stream x.
int launch_im2col_u8(const uint8_t* frames, int n, int height, int width, int64_t frame_stride,
                     int64_t row_stride, int64_t pixel_stride, f16* A, float* x, const float* prefix_tokens,
                     int n_prefix, int D, int T, int ps, hipStream_t stream);
// float32 (n,H,W) in [0,1] -> A[n*P][512] fp16 as hi|lo halves along K (x = hi + lo to ~2^-22)
int launch_im2col_f32(const float* frames, int n, int height, int width, f16* A, float* x,
                      const float* prefix_tokens, int n_prefix, int D, int T, int ps, hipStream_t stream);

// LayerNorm over the last dim (fp32 in, fp16 out), rows = M
// ldx = row stride of x in elements (D for the whole residual stream, T*D for the CLS rows only)
int launch_layernorm_f16(const float* x, int64_t ldx, const float* gamma, const float* beta, f16* out, int M, int D,
                         float eps, hipStream_t stream);
// the same with an MX-fp8 result: out8 [M][D] e4m3 bytes, out_sc [D/128][sc_ld] block scales (GemmParams::A_sc layout)
int launch_layernorm_f8(const float* x, int64_t ldx, const float* gamma, const float* beta, uint8_t* out8,
                        uint32_t* out_sc, int sc_ld, int M, int D, float eps, hipStream_t stream);
// LayerNorm fold, first layer: fp32 x [M][D] -> fp16 copy x16 [M][D] + per 64-column block statistics
// ln_out[(col / 256) * ln_ld + m] = {sum, sum of squares about the block mean} (what EPI_RESID_LN writes for later layers)
int launch_ln_stats_x16(const float* x, f16* x16, float2* ln_out, int ln_ld, int M, int D, hipStream_t stream);
// LayerNorm fold, create time: W [N][K] fp32, gamma / beta [K], b [N] ->
//   Wf [N][K] = fp16(gamma_k W_nk),  colsum[n] = sum_k float(Wf_nk),  biasf[n] = b[n] + sum_k beta_k W_nk
int launch_fold_ln_weight(const float* W, const float* gamma, const float* beta, const float* b, f16* Wf, float* colsum,
                          float* biasf, int N, int K, hipStream_t stream);
// Final LayerNorm on the CLS row of every frame: x[b*T] -> cls_f32[b][D] / cls_f16[b][D]
// `nonfinite` (may be NULL): incremented once per frame whose CLS row holds a NaN / infinity
int launch_final_norm_cls(const float* x, const float* gamma, const float* beta, float* cls_f32,
                          f16* cls_f16, int n, int T, int D, float eps, hipStream_t stream, unsigned* nonfinite = nullptr);

// Multi-head attention over frames: qkv16 [n*T][3D] (q pre-scaled by 1/8, RoPE applied) -> o16 [n*T][D]
// q_cls != nullptr: only the CLS query of every frame (the last layer: [tf]:540-541 + cbas.py:677 consume row 0
// alone): q_cls [n][D] holds the projected CLS queries, out is then [n][D]; K and V still come from qkv.
// out_sc != nullptr: the context is stored as MX-fp8 (out = e4m3 bytes [n*T][D], out_sc [D/128][sc_ld] scales)
int launch_attention(const f16* qkv, const f16* q_cls, void* out, uint32_t* out_sc, int sc_ld, int n, int T, int D,
                     int n_heads, hipStream_t stream);

// fp32 -> fp16 weight conversion (optionally also the fp16 residual), n elements
int launch_convert_f16(const float* src, f16* hi, f16* lo, int64_t n, hipStream_t stream);
// patch weight (D,3,16,16) fp32 -> sum over the 3 identical input channels -> (D,256) fp16 hi (+lo),
// and the same duplicated along K as (D,512) for the float-input path
// fp32 weight [N][K] -> MX-fp8: e4m3 bytes [N][K] + block scales [K/128][N] (GemmParams::W_sc layout); K % 128 == 0
// sc row n of this call is column n0 + n of a scale array with n_total columns (q, k, v packed into one [3D][D] weight)
int launch_pack_fp8_weight(const float* src, uint8_t* w8, uint32_t* sc, int N, int K, int n_total, int n0, hipStream_t stream);
int launch_pack_patch_weight(const float* w, f16* hi, f16* lo, f16* hi2, f16* lo2, int D, int ps, hipStream_t stream);

// ---------------------------------------------------------------------------------------------
// precision 3: the reference's fp32 CPU arithmetic (backend/cbas.py:433-434 autocast off on CPU; [tf]:523-548) on
// v_mfma_f32_16x16x4_f32 - exact fp32 products and sums, k-ordered (vit_f32.hip).  Every buffer is fp32.
// ---------------------------------------------------------------------------------------------
struct Gemm32VitParams {
    const float* A; int64_t lda;   // [M][lda], first K columns used (lda = T*D reads one row per frame: the CLS rows)
    const float* W;                // [N][K], the HF Linear weight as stored
    int M, N, K;                   // N % 128 == 0, K % 32 == 0
    const float* bias;             // [N]
    const float* lambda;           // [N]   (EPI_RESID)
    float* out; int64_t ldo;       // EPI_PATCH / EPI_RESID: the residual stream x; EPI_QKV / EPI_GELU: qkv / u
    // EPI_PATCH
    int patches_per_frame, tokens_per_frame, n_prefix;
    const float* pos;              // DINOv2 position embedding [P][N], or nullptr
    // EPI_QKV
    const float* rope_cos;         // [P][64], or nullptr: no RoPE
    const float* rope_sin;
    // the same numbers by axis (GemmParams::rope_fac): the ping-pong kernel's split form keeps them in LDS (or nullptr)
    const float* rope_fac; int rope_nh, rope_nw; unsigned rope_magic;
    int D;                         // hidden size (q | k | v sections of width D)
    int sec0;                      // section of output column 0
    // precision 4: the same fp32 operands, products on the fp16 matrix pipe as a three-term split (vit_f32.hip).
    // a_scale / w_scale: powers of two the operands are multiplied by before the split (keeps the low halves out of
    // fp16's subnormal range); the accumulators are multiplied by 1 / (a_scale * w_scale) - all exact.
    int split;                     // A and W are in the split format (vit_f32.hip); EPI_GELU then writes its output split too
    float a_scale, w_scale;        // the scales A and W were split with
    float out_scale;               // EPI_GELU with split: scale of the split output (the down projection's a_scale)
};
int launch_gemm_f32_vit(GemmEpilogue epi, const Gemm32VitParams& p, hipStream_t stream);
// precision 4, M > 256, N % 256 == 0: the ping-pong kernel's split-operand form (gemm_f16_8ph.hip); -1 = not its shape
int launch_gemm_split_pp(GemmEpilogue epi, const Gemm32VitParams& p, hipStream_t stream);
void vit32_split_set_forms(int forms);   // bring-up: which precision-4 GEMM forms run (bit 0 ping-pong, bit 1 skinny; -1 = environment)
void gemm_split_pp_set_tile(int tile, unsigned long long* stamps);   // bring-up: forced tile height (0 = planner), block timeline stamps
// uint8 pixels -> A[n*P][256] fp32 = float(double(pixel) / 255.0), the reference's cbas.py:431 value bit for bit
// (+ prefix rows of x, as launch_im2col_u8)
// split != 0 (precision 4): A in the split hi | lo format of vit_f32.hip
int launch_im2col_u8_f32(const uint8_t* frames, int n, int height, int width, int64_t frame_stride, int64_t row_stride,
                         int64_t pixel_stride, float* A, float* x, const float* prefix_tokens, int n_prefix, int D, int T,
                         int ps, int split, hipStream_t stream);
// float32 (n,H,W) -> A[n*P][256] fp32, values as they are
int launch_im2col_f32_f32(const float* frames, int n, int height, int width, float* A, float* x,
                          const float* prefix_tokens, int n_prefix, int D, int T, int ps, int split, hipStream_t stream);
// precision 4: fp32 weight [N][K] -> split format at the same byte size, values multiplied by `scale` first
int launch_pack_split_weight(const float* w, float* out, int64_t N, int K, float scale, hipStream_t stream);
// patch weight (D,3,ps,ps) fp32 -> (D,256) fp32 in the 16x16 slot layout, summed over the 3 identical input channels in
// double and rounded once
int launch_pack_patch_weight_f32(const float* w, float* out, int D, int ps, hipStream_t stream);
int launch_layernorm_f32(const float* x, int64_t ldx, const float* gamma, const float* beta, float* out, int M, int D,
                         float eps, int split, hipStream_t stream);
// qkv [n*T][3D] fp32 (q pre-scaled by 1/8, RoPE applied) -> out [n*T][D] fp32; q_cls as launch_attention
// split_scale > 0 (precision 4): the context is written in the split format, multiplied by split_scale
int launch_attention_f32(const float* qkv, const float* q_cls, float* out, int n, int T, int D, int n_heads,
                         float split_scale, hipStream_t stream);

// ---------------------------------------------------------------------------------------------
// classifier head (all fp32)
// ---------------------------------------------------------------------------------------------
struct Gemm32Params {
    const float* A; int64_t lda;   // [M][lda], first K columns used
    const float* W;                // [N_alloc][K]
    const float* bias;             // [N] or nullptr
    float* out; int64_t ldo;       // [M][ldo]
    int64_t M; int N; int N_alloc; int K;   // N % 4 == 0, K % 32 == 0
    // split-K (head training's weight gradients: K = windows x seq_len, only a handful of output tiles):
    // grid.z = splits, block z accumulates k in [z*K, (z+1)*K) of operands whose row strides are lda / ldw and
    // writes its partial tile to out + z*split_stride (no bias); launch_splitk_reduce adds the partials in order.
    int64_t ldw;          // row stride of W (0: K)
    int splits;           // 0/1: plain GEMM
    int64_t split_stride; // floats between partial outputs
};
int launch_gemm_f32(const Gemm32Params& p, int gelu, hipStream_t stream);
// dst[i] = sum_z partial[z*stride + i], z ascending (deterministic), i < n
int launch_splitk_reduce(const float* partial, int splits, int64_t stride, int64_t n, float* dst, hipStream_t stream);

struct HeadDims {
    int I, C, T, Bn, L0, h;   // in_features, classes, seq_len, bottleneck dim, lin0 dim, lstm hidden
    int NS;                   // bottleneck streams: 3 (cls, delta, acc) or 2 when use_acceleration = False
    int lo, hi;               // centre window [lo, hi) in window time
    int NPROJ;                // projection width: NS*Bn + C rounded up to a multiple of 4
    float alpha;
};

// fp16 rows -> fp32 rows (exact), n elements
int launch_f16_to_f32(const f16* src, float* dst, int64_t n, hipStream_t stream);

// Per-window EMA / delta / acceleration on the PROJECTED rows, + bias, GELU, LayerNorm -> aug[w][t][3Bn];
// also the linear branch lin_logits[w][C] = mean_{t in centre}(EMA(proj_lin1)_t) + b_lin1.
// sliding: proj row of (w,t) = clamp(w0 + w + t - T/2, 0, n_frames-1) - r0 ; else (w*T + t).
int launch_head_expand(const float* proj, const HeadDims& d, const float* b_bott, const float* ln_w,
                       const float* ln_b, const float* b_lin1, int64_t n_windows, int sliding, int64_t w0,
                       int64_t r0, int64_t n_frames, float* aug, float* lin_logits, hipStream_t stream);
// xl[w][t][:] -= mean_t xl[w][t][:]
int launch_head_centre(float* xl, int64_t n_windows, int T, int L0, hipStream_t stream);
// Recurrent half of one BiLSTM layer: gin[w][t][dir*4h + gate*h + unit] holds W_ih x + b; runs fwd steps
// 0..ohi-1 and bwd steps T-1..olo, writes hout[w][t-olo][dir*h + unit] for t in [olo,ohi).  The last
// layer uses the centre window (olo,ohi) = (lo,hi); inner layers of a stacked LSTM use (0,T).
int launch_head_lstm(const float* gin, const float* w_hh, const HeadDims& d, int olo, int ohi, int64_t n_windows,
                     float* hout, hipStream_t stream);
// attention pooling + lin2 + gate lerp + softmax(logits / max(1e-3, T))
int launch_head_pool(const float* hout, const float* lin_logits, const HeadDims& d, const float* w_att,
                     float b_att, float att_temp, const float* w_lin2, const float* b_lin2, float gate_sigmoid,
                     float temperature, int64_t n_windows, float* probs, float* logits, float* latent,
                     hipStream_t stream);

// ---------------------------------------------------------------------------------------------
// classifier-head training (head_train_kernels.hip; orchestration in api_head_train.hip)
// ---------------------------------------------------------------------------------------------
constexpr int COLSUM_CHUNKS = 64;

struct TrainExpandParams {
    const float* proj;       // [w][T][NPROJ] per-position projections (cls | delta | acc | lin1 | pad)
    const float* tmat;       // [3][T][T] temporal operators (EMA, delta o EMA, accel o EMA)
    const float* lin_vec;    // [T] mean over the centre window of the EMA rows
    const float *b_bott, *ln_w, *ln_b, *b_lin1;
    int T, Bn, NPROJ, C;
    int NS;                      // bottleneck streams (3, or 2 without the acceleration stream)
    unsigned long long key[3];   // dropout stream keys of the three bottlenecks for this step
    unsigned thr;                // keep <=> hash24 >= thr
    float scale;                 // 1 / (1 - p)
};
struct TrainPoolParams {
    const float *hout, *lin_logits, *w_att, *b_att, *att_temp, *w_lin2, *b_lin2, *gate;
    int T, H2, C, lo, hi;
};

// dst [cols][rows_pad] = src^T, columns rows..rows_pad zero-filled
int launch_transpose_pad(const float* src, int64_t rows, int cols, int64_t ld, float* dst, int64_t rows_pad, hipStream_t st);
size_t train_expand_lds_bytes(int T, int Bn, int NPROJ);     // dynamic LDS the expand kernels need (<= 160 KiB)
int launch_train_expand_fwd(const TrainExpandParams& p, int64_t n_windows, float* Y, float* aug, float* lin_logits, hipStream_t st);
int launch_train_expand_bwd(const TrainExpandParams& p, int64_t n_windows, const float* Y, const float* daug, const float* dlin,
                            float* dproj, float* part, hipStream_t st);
int launch_gelu_dropout(const float* Z, float* io, int64_t n, unsigned long long key, unsigned thr, float scale, int backward,
                        hipStream_t st);
int launch_lstm_train_fwd(const float* gin, const float* w_hh, int h, int T, int64_t n_windows, float* act, float* cst,
                          float* hout, hipStream_t st);
int launch_lstm_train_bwd(const float* dhout, const float* act, const float* cst, const float* hout, const float* w_hh, int h,
                          int T, int64_t n_windows, float* dgin, float* hprev, hipStream_t st);
int launch_pool_train_fwd(const TrainPoolParams& p, int64_t n_windows, float* attw, float* scores, float* latent,
                          float* lstm_logits, float* final_logits, hipStream_t st);
int launch_pool_train_bwd(const TrainPoolParams& p, int64_t n_windows, const float* attw, const float* scores,
                          const float* lstm_logits, const float* dfinal, const float* dlat_cov, float* dhout,
                          float* dlstm_logits, float* dlin_logits, float* part, hipStream_t st);
int launch_ce_terms(const float* logits, const int* labels, const float* cw, int64_t n, int C, float eps, float* terms,
                    hipStream_t st);
int launch_ce_grad(const float* logits, const int* labels, const float* cw, const float* sums, int64_t n, int C, float eps,
                   float* dlogits, hipStream_t st);
int launch_cov_offdiag(const float* cov, int n, float cscale, float gscale, float* G, float* sq, hipStream_t st);
int launch_sub_colmean(const float* src, const float* colsum, int64_t rows, int cols, float* dst, hipStream_t st);
// deterministic column sums of src [rows][cols] (row stride ld): dst[c] = scale * sum_r src[r][c]; tmp: COLSUM_CHUNKS * cols floats
int launch_colsum(const float* src, int64_t rows, int cols, int64_t ld, float scale, float* tmp, float* dst, hipStream_t st);
int launch_add_vec(const float* a, const float* b, float* out, int n, hipStream_t st);      // b == nullptr: copy
int launch_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float wd, int64_t wd_lo, int64_t wd_hi,
                     float wd_special, int step, hipStream_t st);

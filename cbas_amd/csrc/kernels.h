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
};

struct GemmParams {
    const f16* A;        // [M_pad][K]
    const f16* W;        // [N][K]  (hi part when split)
    const f16* W_lo;     // [N][K]  residual W - fp16(W) as fp16, or nullptr
    int M;               // valid rows (stores are skipped for rows >= M)
    int M_pad;           // rows the grid covers, multiple of 128
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
    // EPI_QKV
    const float* rope_cos;   // [P][64]
    const float* rope_sin;   // [P][64]
    int D;                   // hidden size (q | k | v sections of width D)
};

int launch_gemm(GemmEpilogue epi, const GemmParams& p, hipStream_t stream);

// ---------------------------------------------------------------------------------------------
// ViT element-wise / attention kernels
// ---------------------------------------------------------------------------------------------
// uint8 pixels -> im2col matrix A[n*P][256] fp16 holding the INTEGER pixel values (exact in fp16);
// also writes the CLS/register prefix rows of the residual stream x.
int launch_im2col_u8(const uint8_t* frames, int n, int height, int width, int64_t frame_stride,
                     int64_t row_stride, int64_t pixel_stride, f16* A, float* x, const float* prefix_tokens,
                     int n_prefix, int D, int T, hipStream_t stream);
// float32 (n,H,W) in [0,1] -> A[n*P][512] fp16 as hi|lo halves along K (x = hi + lo to ~2^-22)
int launch_im2col_f32(const float* frames, int n, int height, int width, f16* A, float* x,
                      const float* prefix_tokens, int n_prefix, int D, int T, hipStream_t stream);

// LayerNorm over the last dim (fp32 in, fp16 out), rows = M
int launch_layernorm_f16(const float* x, const float* gamma, const float* beta, f16* out, int M, int D,
                         float eps, hipStream_t stream);
// Final LayerNorm on the CLS row of every frame: x[b*T] -> cls_f32[b][D] / cls_f16[b][D]
int launch_final_norm_cls(const float* x, const float* gamma, const float* beta, float* cls_f32,
                          f16* cls_f16, int n, int T, int D, float eps, hipStream_t stream);

// Multi-head attention over frames: qkv16 [n*T][3D] (q pre-scaled by 1/8, RoPE applied) -> o16 [n*T][D]
int launch_attention(const f16* qkv, f16* out, int n, int T, int D, int n_heads, hipStream_t stream);

// fp32 -> fp16 weight conversion (optionally also the fp16 residual), n elements
int launch_convert_f16(const float* src, f16* hi, f16* lo, int64_t n, hipStream_t stream);
// patch weight (D,3,16,16) fp32 -> sum over the 3 identical input channels -> (D,256) fp16 hi (+lo),
// and the same duplicated along K as (D,512) for the float-input path
int launch_pack_patch_weight(const float* w, f16* hi, f16* lo, f16* hi2, f16* lo2, int D, hipStream_t stream);

// ---------------------------------------------------------------------------------------------
// classifier head kernels (fp32)
// ---------------------------------------------------------------------------------------------
struct HeadDims {
    int I, C, T, Bn, L0, h, sw;   // in_features, classes, seq_len, bottleneck, lin0 dim, lstm hidden, centre half-width
    float alpha;
};
struct HeadWeightsDev {
    // projection matrix for the per-row stage: [NP][I] rows = cls(Bn) | delta(Bn) | acc(Bn) | lin1(C), zero padded to NP
    const float* w_proj; int NP;
    const float* b_bott;      // [3*Bn]   bottleneck biases
    const float* ln_w;        // [3*Bn]
    const float* ln_b;        // [3*Bn]
    const float* b_lin1;      // [C]
    const float* w_lin0;      // [L0][3*Bn]
    const float* b_lin0;      // [L0]
    const float* w_ih;        // [2][4h][L0]
    const float* w_hh;        // [2][4h][h]
    const float* b_gate;      // [2][4h]  (bias_ih + bias_hh)
    const float* w_att;       // [2h]
    float b_att;
    const float* w_lin2;      // [C][2h]
    const float* b_lin2;      // [C]
    float gate_sigmoid;       // sigmoid(gate)
    float att_temp;           // softplus(attention_temp) + 1e-3
};

// rows (fp16 or fp32, [n_rows][I]) -> proj [n_rows][NP] fp32
int launch_head_project(const void* rows, int rows_are_f16, int64_t n_rows, const HeadDims& d,
                        const HeadWeightsDev& w, float* proj, hipStream_t stream);
// per-window stage.  sliding != 0: window i = rows clamp(i-half .. i+half); else window i = rows i*T .. i*T+T-1.
int launch_head_windows(const float* proj, int64_t n_rows, int64_t n_windows, int sliding, const HeadDims& d,
                        const HeadWeightsDev& w, float temperature, float* probs, float* logits,
                        float* latent, hipStream_t stream);

/*
 * cbas_mi355x_debug.h - bring-up, test and measurement-harness entry points of the DEBUG build of the library
 * (libcbas_mi355x_debug.so: `python -m cbas_amd.build --debug`, every source compiled with -DCBAS_BUILD_DEBUG=1).
 *
 * None of this is part of the drop-in boundary: the product library (libcbas_mi355x.so, include/cbas_mi355x.h) exports no
 * symbol declared here (tests/test_host_logic.py checks `nm -D`).  The GPU test suite and scripts/ load the debug build
 * (CBAS_BUILD_DEBUG=1 in the environment, read by cbas_amd/_lib.py); bench.py, __graft_entry__.smoke() and a CBAS
 * installation load the product build.  The debug build is a superset: same kernels, same product entry points.
 */
#ifndef CBAS_MI355X_DEBUG_H
#define CBAS_MI355X_DEBUG_H

#include "cbas_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

/* 1 in the debug build (the product build does not export the symbol at all). */
int cbas_debug_build(void);

/* Bring-up/debug: run the forward pass only up to (layer, stage) and copy an internal buffer to
 * the host.  stage: 0 embeddings (x), then per layer 1 LN1(h16) 2 QKV(qkv16) 3 attention(h16)
 * 4 o_proj residual (x) 5 LN2 (h16) 6 up_proj+GELU (u16) 7 down_proj residual (x).
 * which: 0 x f32 (rows,D)  1 h16 (rows,D)  2 qkv16 (rows,3D)  3 u16 (rows,F); rows = n*T. */
int cbas_enc_debug_forward_u8(cbas_enc* h, const uint8_t* frames_dev, int n, int height, int width,
                              int64_t frame_stride, int64_t row_stride, int64_t pixel_stride,
                              int stop_layer, int stop_stage);
int cbas_enc_debug_read(cbas_enc* h, int which, void* host_out, int64_t n_bytes);

/* Bring-up / tests: switch an implementation detail of the handle.  Options:
 *   "rope_lds"  1 (default): the q|k|v epilogue reads the RoPE angles, factorised by axis, from LDS; 0: from the [P][64]
 *               table in global memory ([tf]:96-121, 168-200 either way).  Bit-identical results.
 *   "ln_fold"   1: LayerNorm ([tf]:404, 410) is folded into the GEMMs around it - o_proj / down_proj write a fp16 copy of
 *               the residual stream and per-row statistics, q|k|v / up_proj run on it with gamma folded into their weights and
 *               apply mean / rstd in their epilogues; 0: separate LayerNorm kernels.  The two settings agree to fp16 rounding
 *               (both within the 1e-3 CLS bar, both batch-invariant); fp16 path with hidden_size a multiple of 256 only.
 *               Default 0: with two batches in flight the separate kernels already hide under the other lane's GEMMs
 *               (measured +0 ... +1 % for the fold; -5 % of kernel time with a single batch in flight).
 *   "split_kernels"  precision 4, PROCESS-WIDE: which GEMM forms run - bit 0 the ping-pong kernel's split-operand form
 *               (M > 256, N a multiple of 256), bit 1 the 8-slot-ring skinny form (M <= 256); cleared bits fall to the
 *               128 x 128 kernels.  -1 (default): both on, or as CBAS_SPLIT_PP=0 / CBAS_SPLIT_SKINNY=0 say.  Every
 *               setting forms the same products in the same order per output element: bit-identical rows. */
int cbas_enc_debug_option(cbas_enc* h, const char* name, int value);

/* Bring-up / tests (round 4).
 *   cbas_debug_gemm_split_bench: precision 4's GEMM alone on random split operands (epi 1 q|k|v, 2 residual, 3 GELU; tile 0 =
 *       planner, 128 / 160 / 192 / 256 rows of the ping-pong form, -1 = the 128 x 128 kernel), prints its block timeline.
 *   cbas_debug_mfma_neighbor: queue a register-only v_mfma_f32_32x32x16_f16 loop (every SIMD, two waves each, `iters` rounds
 *       of 8 MFMAs) on `stream`: the neighbour beside which the head is checked for bit-stability
 *       (scripts/head_beside_encoder.py).
 *   cbas_head_debug_read: copy the first n_floats of a head workspace buffer of the last pass to the host
 *       (0 rows32, 1 proj, 2 aug, 3 xl, 4 gin, 5 hout, 6 lin_logits); the device is synchronised first. */
int cbas_debug_gemm_split_bench(int M, int N, int K, int epi, int tile, int iters, float* ms_out);
/* the same GEMM through the ping-pong / skinny forms and through the 128 x 128 kernels on the same random operands:
 * n_diff = 32-bit output words that differ (0 by construction: same products in the same order) */
int cbas_debug_gemm_split_compare(int M, int N, int K, int epi, int tile, int64_t* n_diff);
int cbas_debug_mfma_neighbor(int iters, void* stream);
int cbas_head_debug_read(cbas_head* h, int which, float* host_out, int64_t n_floats);

/* Bring-up: time the fp16 GEMM kernel alone on random operands (GELU epilogue, M x N x K,
 * tile: 0 auto, 1 128x128, 2 256x128, 3 128x256, 4 256x256, 5+ experimental variants) and return a
 * position-weighted checksum of the fp16 output, so tile variants can be compared bit for bit. */
int cbas_debug_gemm_bench(int M, int N, int K, int tile, int iters, float* ms_out,
                          unsigned long long* checksum_out);

/* Bring-up: how kernels of two compute lanes share the chip.  Concurrently, each on its own stream, `iters` launches of
 * bit 0 the up projection (12 864 x 3072 x 768, GELU), bit 1 LayerNorm (12 864 x 768), bit 2 attention (64 x 201 tokens,
 * 12 heads), bit 3 the down projection (residual epilogue).  ms_out[0..3] = average milliseconds per launch of each
 * component on its stream, ms_out[4] = wall milliseconds of the whole run (scripts/overlap_kernels.py). */
int cbas_debug_overlap(int mode, int iters, float* ms_out);

/* Bring-up / tests: the MX-fp8 GEMM of precision 2 in isolation.  A (M x K) and W (N x K) fp32 host matrices are
 * quantised with the library's block quantiser (e4m3 elements, one E8M0 scale per 32 k-elements), multiplied by the
 * fp8 kernel (tile: 0 = the shape's default, 13..16 = a fixed ping-pong tile) and out = A_q W_q^T (M x N fp32) is
 * returned together with the quantised bytes and scales ([K/128][round_up(M,256)] resp. [K/128][N] dwords, byte b of
 * a dword = block b of that 128-wide K-tile).  N % 256 == 0, K % 256 == 0. */
int cbas_debug_gemm_f8(int M, int N, int K, int tile, const float* A_host, const float* W_host, float* out_host,
                       uint8_t* A8_host, uint32_t* Asc_host, uint8_t* W8_host, uint32_t* Wsc_host);

/* Root-cause probe (round 5): run a kernel from a separately built code object IN PLACE of the library's head_expand_kernel
 * on this handle (same grid, block, dynamic LDS and arguments: scripts/probes/expand_r4/expand_r4.hip has the signature).
 * hsaco_path = NULL restores the library's kernel.  scripts/expand_rootcause.py builds instruction-level variants of the
 * round-4 kernel that returned wrong values beside MFMA-heavy neighbours and runs each one through the head this way. */
int cbas_head_debug_expand_module(cbas_head* h, const char* hsaco_path, const char* kernel_name);
/* Amplification for that probe.  mode 1: the next pass (run it on an idle device) copies its expand output to a reference
 * buffer; mode 2: every pass launches the probe kernel `repeat` times and compares every launch's rows bit for bit with
 * the reference ON THE DEVICE (differing rows are captured, 512 at most); mode 0: off.  cbas_head_debug_expand_stats:
 * counts4 = launches compared, launches with a differing row, differing rows, rows offered to the capture buffer;
 * rows_out receives up to max_rows captured rows of 2 + bottleneck_dim floats (row index = (window * T + t) * NS + stream,
 * launch number, the row's values); reset != 0 clears the counters and the capture buffer's fill. */
int cbas_head_debug_expand_repeat(cbas_head* h, int mode, int repeat);
int cbas_head_debug_expand_stats(cbas_head* h, uint64_t* counts4, float* rows_out, int max_rows, int reset);

#ifdef __cplusplus
}
#endif
#endif /* CBAS_MI355X_DEBUG_H */

// fp16 MFMA GEMM, "ping-pong" schedule: BM x 256 x 64 tile, 8 waves as 2(M) x 4(N), each wave
// 16*(TA+TB) rows x 64 columns of C.  (TA,TB) = (4,4): 256x256 tile, (3,3): 192x256, (3,2): 160x256,
// (2,2): 128x256.  Same arithmetic, operand layout, LDS image and fused epilogues as gemm_f16.hip
// (bit-identical results: every accumulator sees the same MFMAs in the same k order); what differs
// is the K loop:
//
//  * Per-wave tile up to 128 x 64 instead of 64 x 64: 24 ds_read_b128 per 64 MFMAs (16 per 32 before),
//    so the LDS array is no longer a co-limiter of the MFMA pipe.
//  * The two wave groups (wr = 0 / 1; one wave of each per SIMD) run one barrier apart: while one
//    group issues its MFMA cluster the other reads fragments and issues LDS-DMA for a later K-tile,
//    then they swap.  A K-tile is four such phases (one C quadrant of the wave each).
//  * LDS-DMA is never drained inside the loop: each K-tile is staged as four sub-tiles (B-sub0,
//    A-sub0, B-sub1, A-sub1: the rows the phase-1/2/3 fragment reads touch), three sub-tiles stay
//    in flight across the barriers (s_waitcnt vmcnt(6), raw s_barrier), and a sub-tile region is
//    re-staged only after every read of it has been retired (see the table).
//
//  phase of K-tile t (buffer b = t&1)   fragment reads             LDS-DMA issued           MFMA quadrant
//    P1                                 B-sub0 (4), A-sub0 (2*TA)   A-sub1(t+1) -> b^1        (A0,B0)
//    P2                                 B-sub1 (4)                  B-sub0(t+2) -> b          (A0,B1)
//    P3                                 A-sub1 (2*TB)               A-sub0(t+2) -> b          (A1,B1)
//    P4                                 -                           B-sub1(t+2) -> b, vmcnt   (A1,B0)
//  RAW: a K-tile's data is waited for (counted vmcnt by each issuing wave) before P4's first barrier
//  of the previous K-tile and first read in P1, i.e. after two further barriers - one more than the
//  group stagger.  WAR: B-sub0 is re-staged one phase after its reads, which P1 retires with
//  lgkmcnt(2*TA) before its first barrier (reads issue B first); every other region two phases after.
//
// Tail balancing: a launch may carry two kinds of tile.  Blocks [0, main_blocks) cover rows
// [0, tail_row0) with the main tile, the rest cover [tail_row0, M) with 128x256 tiles; the hardware
// hands out workgroups in block order, so the small tiles fill the last, partial round of the big
// ones (612 big tiles on 256 CUs are 3 rounds; 504 big + 204 half-size tiles finish in about 2.65).
//
// MX-fp8 form (F8 = true, precision 2): operands are e4m3 bytes with one E8M0 scale per 32 k-elements.  A K-tile is
// 128 k-elements = the same 128-byte LDS rows, swizzle, staging schedule and fragment reads as the fp16 form; the two
// 16-byte fragment reads of a lane (chunks q and 4+q of a row, q = lane >> 4) are exactly the 32 bytes
// v_mfma_scale_f32_16x16x128_f8f6f4 wants from it (k = 16q..16q+15 and 64+16q..64+16q+15; probed on gfx950 with
// scripts/probes/probe_mfma_scale.hip), and the scale of block b of a row comes from lane q = b.  So the K loop issues
// ONE scaled MFMA where the fp16 form issues two, at twice the k per staged byte.  The block scales of a K-tile (4 bytes
// per row, 2 KiB per tile pair) ride along as one extra 4-byte LDS-DMA per wave issued with the A-sub1 sub-tile, and are
// read with the fragments they belong to.
//
// Split-operand form (SPLIT = true, precision 4; gemm_split_pp_kernel): operands are fp32 values stored as fp16 hi | lo halves,
// 32 k-values per 128-byte LDS row (vit32_epilogue.h describes the format) - again the same rows, swizzle, staging schedule and
// fragment reads; a lane's two 16-byte reads are the hi and the lo halves of its 8 k-values, a phase issues w_lo a_hi, w_hi a_lo,
// w_hi a_hi per accumulator (24 MFMAs where the fp16 form has 16: the loop is MFMA-bound at 256 rows with a third of the
// staging per MFMA), and the epilogues are the fp32-schedule ones of vit32_epilogue.h, bounced through the same per-wave LDS
// scratch.  The fp16 / MX-fp8 instantiations are untouched by it (if constexpr; same instruction streams as before).
//
// Reference arithmetic replaced: the same nn.Linear calls as gemm_f16.hip ([tf] modeling_dinov3_vit.py
// :307-309, :331, :356-357).
#include <stdlib.h>
#include <map>
#include <mutex>
#include <queue>
#include <type_traits>
#include <vector>
#include "gemm_epilogue.h"
#include "vit32_epilogue.h"

namespace {

constexpr int BK = 64;
constexpr int BN = 256;

struct PPGrid {               // tile ids of one launch; workgroup b runs ids b, b + gridDim.x, b + 2 gridDim.x, ...
    int main_blocks;          // ids [0, main_blocks): main tiles over rows [0, tail_row0)
    int tail_row0;            // first row of the tail segment (== M rounded up when there is none)
    int n_tiles;              // ids [main_blocks, n_tiles): 128-row tail tiles
};

__device__ __forceinline__ f16x8 read_frag8(const char* lds_tile, int off) {
    return *reinterpret_cast<const f16x8*>(lds_tile + off);
}

// LDS bytes of a (TA, TB) tile: two staging buffers, with the epilogue's 64 KiB of per-wave scratch laid over
// buffer 1 and whatever follows it (buffer 0 is being refilled with the next tile's first K-tile by then)
template <int TA, int TB>
constexpr int pp_lds_main() {
    constexpr int buf = (32 * (TA + TB) + BN) * 128;
    return 2 * buf > buf + 65536 ? 2 * buf : buf + 65536;
}

// staging buffers (+ MX scale images) of a kernel instantiation: what lies in front of the q|k|v kernel's RoPE table
template <int TA, int TB, int TAIL, bool F8>
constexpr int pp_lds_kernel() {
    return (TAIL && pp_lds_main<2, 2>() > pp_lds_main<TA, TB>() ? pp_lds_main<2, 2>() : pp_lds_main<TA, TB>()) + (F8 ? 4096 : 0);
}

// Runs the tiles `first, first + stride, ...` (< n_kind) of one kind: tile id -> (tm, tn) through the XCD remap over
// n_kind ids; rows start at base_row.  Persistent: while a tile's epilogue runs, the LDS-DMA of the NEXT tile's first
// K-tile is already in flight into buffer 0 (its latency - most of the prologue - hides under the epilogue).
template <int EPI, int TA, int TB, bool F8, bool SPLIT = false, typename SP = int>
__device__ __forceinline__ void pp_tiles(const GemmParams& p, char* smem, int first, int n_kind, int stride, int base_row,
                                         int tiles_n, const float* rope_lds, char* ln_lds, const SP& sp = SP()) {
    static_assert(!(F8 && SPLIT), "one operand form at a time");
    constexpr int TM = TA + TB;                      // 16-row MFMA tiles per wave
    constexpr int WROWS = 16 * TM;                   // rows of C per wave
    constexpr int BM = 2 * WROWS;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, BUF_BYTES = A_BYTES + B_BYTES;
    constexpr int TMAX = TA > TB ? TA : TB;
    int row0 = 0, col0 = 0;

    const int tid = threadIdx.x;
    // `lane` is laundered through an empty asm at the top of every tile and again before the next tile's offsets are
    // formed: everything derived from it (staging / fragment offsets, ~20 VGPRs) is then recomputed per tile instead
    // of being kept alive across the epilogue, which needs those registers (128 accumulators + 32 prefetched x values)
    int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int nk = p.K / (F8 ? 2 * BK : SPLIT ? BK / 2 : BK);   // even, >= 2 (checked by the launcher)
    constexpr int ESZ = F8 ? 1 : SPLIT ? 4 : 2;        // bytes per operand element: a K-tile is always 128 bytes per row
    // operands are addressed through buffer descriptors (SGPRs) + a 32-bit per-lane offset + a scalar K-tile offset:
    // no 64-bit address arithmetic per LDS-DMA, fewer VGPRs (what lets a LayerNorm wave of the other lane share a SIMD)
    const __amdgpu_buffer_rsrc_t Arsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(F8 || SPLIT ? reinterpret_cast<const void*>(p.A8) : reinterpret_cast<const void*>(p.A)), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t Wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(F8 || SPLIT ? reinterpret_cast<const void*>(p.W8) : reinterpret_cast<const void*>(p.W)), 0, 0x7fffffff, 0x00020000);
    char* const sc_lds = smem + pp_lds_main<TA, TB>(); // F8: [buf][A rows 256 x 4 B | B rows 256 x 4 B]

    // ---- LDS-DMA source offsets: two 8-row pieces per wave and sub-tile --------------------------
    // A-sub0 = rows {wr' * WROWS + [0, 16 TA)}, A-sub1 = rows {wr' * WROWS + 16 TA + [0, 16 TB)}, wr' = 0,1;
    // B-sub h = rows with bit 5 == h.  Piece q of a sub-tile (q = wave, wave + 8); a sub-tile with fewer
    // than 16 pieces clamps q: the duplicates rewrite identical bytes (keeps the per-wave DMA count, and
    // so the vmcnt immediates, uniform)
    int a_lds[2][2], b_lds[2][2];
    unsigned a_src[2][2], b_src[2][2];
    // F8: block scales of one K-tile.  Waves 0-3 fetch the dwords of A rows row0 + 64*wave + lane, waves 4-7 those of
    // W rows col0 + 64*(wave-4) + lane (rows past the tile / past M_pad are clamped or belong to a neighbour: never used)
    unsigned sc_src = 0;                               // byte offset into A_sc (waves 0-3) / W_sc (waves 4-7)
    const int sc_step = F8 ? 4 * (wave < 4 ? p.sc_lda : (p.sc_ldw ? p.sc_ldw : p.N)) : 0, sc_dst = wave * 256;   // bytes per K-tile
    const __amdgpu_buffer_rsrc_t Srsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(reinterpret_cast<const void*>(wave < 4 ? p.A_sc : p.W_sc)), 0, 0x7fffffff, 0x00020000);
    auto set_tile = [&](int id) {
        const int lrow = lane >> 3;
        const int bid = gemm_xcd_remap(id, n_kind);
        int tm, tn;
        if (p.group_m > 1) {
            // grouped raster: group_m row panels x all column tiles, row-fastest, so that the ~32 tiles an XCD runs at
            // a time form a squarish block (operand footprint per round (rows + cols) x 256 x K instead of a few
            // rows x every column)
            const int width = p.group_m * tiles_n, grp = bid / width, in = bid - grp * width;
            const int m0 = grp * p.group_m, left = n_kind / tiles_n - m0;
            const int gsz = left < p.group_m ? left : p.group_m;
            tn = in / gsz;
            tm = m0 + in - tn * gsz;
        } else {
            tm = bid / tiles_n;
            tn = bid - tm * tiles_n;
        }
        row0 = base_row + tm * BM;
        col0 = tn * BN;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int T = h ? TB : TA;
                int q = wave + 8 * s;
                q = q < 4 * T ? q : 4 * T - 1;
                const int ar = (q / (2 * T)) * WROWS + h * (16 * TA) + (q % (2 * T)) * 8;   // first row of the piece
                a_lds[h][s] = ar * 128;
                const int r = ar + lrow;
                int grow = row0 + r;
                grow = grow < p.M_pad - 1 ? grow : p.M_pad - 1;
                a_src[h][s] = (unsigned)grow * (unsigned)(p.lda * ESZ) + (((lane & 7) ^ ((r >> 1) & 7)) << 4);     // bytes
                const int qb = wave + 8 * s;
                const int br = (((qb >> 2) << 3) | (h << 2) | (qb & 3)) * 8;
                b_lds[h][s] = A_BYTES + br * 128;
                const int rb = br + lrow;
                b_src[h][s] = (unsigned)(col0 + rb) * (unsigned)(p.K * ESZ) + (((lane & 7) ^ ((rb >> 1) & 7)) << 4);
            }
        if (F8) {
            if (wave < 4) {
                int r = row0 + wave * 64 + lane;
                r = r < p.M_pad - 1 ? r : p.M_pad - 1;
                sc_src = (unsigned)r * 4u;
            } else {
                sc_src = (unsigned)(col0 + (wave - 4) * 64 + lane) * 4u;
            }
        }
    };
    set_tile(first);
    // which: 0 = B-sub0, 1 = A-sub0, 2 = B-sub1, 3 = A-sub1 (the issue order within a K-tile)
    auto stage = [&](int buf, int kt, int which) {
        char* base = smem + buf * BUF_BYTES;
        const int h = which >> 1;
        if (which & 1) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(Arsrc, LDS_PTR(base + a_lds[h][s]), 16, a_src[h][s], kt * 128, 0, 0);
            if (F8 && h == 1)        // rides with A-sub1: retired by the same counted vmcnt, two barriers before its first read
                __builtin_amdgcn_raw_ptr_buffer_load_lds(Srsrc, LDS_PTR(sc_lds + buf * 2048 + sc_dst), 4, sc_src, kt * sc_step, 0, 0);
        } else {
#pragma unroll
            for (int s = 0; s < 2; ++s)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(Wrsrc, LDS_PTR(base + b_lds[h][s]), 16, b_src[h][s], kt * 128, 0, 0);
        }
    };

    // ---- fragment read offsets (swizzled 128-B rows, as gemm_f16.hip) -------------------------------
    int a_off[2], b_off[2];                           // per k-half; + row-block immediates
    int sa_off = 0, sb_off = 0;                       // F8: byte of lane (frow, q = fchunk) in the scale image: row * 4 + q
    auto set_frag_offsets = [&] {
        const int frow = lane & 15, fchunk = lane >> 4;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int sw = ((kk * 4 + fchunk) ^ ((frow >> 1) & 7)) << 4;
            a_off[kk] = (wr * WROWS + frow) * 128 + sw;
            b_off[kk] = A_BYTES + (wc * 64 + frow) * 128 + sw;
        }
        sa_off = (wr * WROWS + frow) * 4 + fchunk;              // + (I0 + i) * 64
        sb_off = 1024 + (wc * 64 + frow) * 4 + fchunk;          // + (h * 2 + j) * 64
    };

    // LayerNorm fold, consumer side: per-N-tile statistics of row row0 + tid (threads of the first four waves)
    constexpr bool LNC = EPI == EPI_QKV_LN || EPI == EPI_GELU_LN;
    f32x2 lnp[4];
    auto load_lnp = [&] {                              // uses row0 as set_tile left it: this tile's, or the NEXT tile's
        if constexpr (LNC) {
            if (tid < BM) {
                int m = row0 + tid;
                m = m < p.M ? m : p.M - 1;
                const f32x2* src = reinterpret_cast<const f32x2*>(p.ln_in) + m;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < p.ln_parts) lnp[k] = src[(size_t)k * p.ln_ld];
            }
        }
    };

    f32x4 acc[TM][4];
    unsigned long long t_start = 0, t_top = 0, t_pro = 0, t_loop = 0, r_pro = 0, r_loop = 0;   // r_*: s_memrealtime (100 MHz)
    if (p.stamps) t_start = __builtin_amdgcn_s_memtime();
    load_lnp();
    stage(0, 0, 0); stage(0, 0, 1); stage(0, 0, 2); stage(0, 0, 3);      // K-tile 0 of the first tile

    // fragments: fp16 form = two 16-byte k-halves per 16-row tile; F8 form = the same two reads joined into the
    // 8-register operand of the scaled MFMA (a register sequence, no copies) + its E8M0 scale byte
    struct Frag { f16x8 h[2]; i32x8 v; int sc; };
    Frag a[TMAX], b0[2], b1[2];

#define CBAS_SEG_BARRIER()                      \
    do {                                        \
        __builtin_amdgcn_sched_barrier(0);      \
        __builtin_amdgcn_s_barrier();           \
        __builtin_amdgcn_sched_barrier(0);      \
    } while (0)

    auto mfma_quadrant = [&](auto ha_c, Frag (&bf)[2], int hb) {
        constexpr int ha = decltype(ha_c)::value;
        constexpr int T = ha ? TB : TA, I0 = ha ? TA : 0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        if constexpr (F8) {
#pragma unroll
            for (int i = 0; i < T; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[I0 + i][hb * 2 + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
                        bf[j].v, a[i].v, acc[I0 + i][hb * 2 + j], 0, 0, 0, bf[j].sc, 0, a[i].sc);
            // hipcc (ROCm 7.2) sinks the scaled MFMAs out of their phase to the end of the loop body (their results are
            // only consumed there); an empty volatile asm that "uses" each accumulator pins them inside the setprio window
#pragma unroll
            for (int i = 0; i < T; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(acc[I0 + i][hb * 2 + j]));
        } else if constexpr (SPLIT) {
            // h[0] = the hi halves of the K-tile's 32 k-values, h[1] = the lo halves; per accumulator the three products in
            // the order of vit_f32.hip's split kernels (w_lo a_hi, w_hi a_lo, w_hi a_hi): bit-identical results
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int i = 0; i < T; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[I0 + i][hb * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                            bf[j].h[term == 0 ? 1 : 0], a[i].h[term == 1 ? 1 : 0], acc[I0 + i][hb * 2 + j], 0, 0, 0);
        } else {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < T; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[I0 + i][hb * 2 + j] =
                            __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j].h[kk], a[i].h[kk], acc[I0 + i][hb * 2 + j], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
    };
    auto read_frag = [&](Frag& f, const char* buf, int off0, int off1, const char* scb, int sc_off) {
        if constexpr (F8) {
            const i32x4 lo = *reinterpret_cast<const i32x4*>(buf + off0), hi = *reinterpret_cast<const i32x4*>(buf + off1);
            f.v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            // The scale byte is read by an asm ds_read_u8: as a C++ byte load hipcc masks it (v_and 0xff) in front of the
            // phase barrier, which puts an lgkmcnt wait ladder - the whole LDS latency of the phase's fragment reads - before
            // every barrier (16 % of the K loop).  Hidden from the compiler the value is only consumed by the MFMAs, behind the
            // explicit lgkmcnt(0) + sched_barrier at the top of mfma_quadrant (the form cdna_hip_programming.md 5.7 (iii) asks for).
            asm volatile("ds_read_u8 %0, %1" : "=v"(f.sc) : "v"((unsigned)(uintptr_t)LDS_PTR(scb + sc_off)));
        } else {
            f.h[0] = read_frag8(buf, off0);
            f.h[1] = read_frag8(buf, off1);
        }
    };
    auto read_a = [&](const char* buf, const char* scb, auto h_c) {
        constexpr int h = decltype(h_c)::value;
        constexpr int T = h ? TB : TA, I0 = h ? TA : 0;
#pragma unroll
        for (int i = 0; i < T; ++i)
            read_frag(a[i], buf, a_off[0] + (I0 + i) * 2048, a_off[1] + (I0 + i) * 2048, scb, sa_off + (I0 + i) * 64);
    };
    auto read_b = [&](const char* buf, const char* scb, int h, Frag (&bf)[2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
            read_frag(bf[j], buf, b_off[0] + (h * 2 + j) * 2048, b_off[1] + (h * 2 + j) * 2048, scb, sb_off + (h * 2 + j) * 64);
    };
    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, 1>;

    auto ktile = [&](int b, int t) {
        const char* buf = smem + b * BUF_BYTES;
        const char* scb = sc_lds + b * 2048;
        // P1
        read_b(buf, scb, 0, b0);
        __builtin_amdgcn_sched_barrier(0);
        read_a(buf, scb, H0{});
        if (t + 1 < nk) stage(b ^ 1, t + 1, 3);
        // the B-sub0 reads (issued first: 4 fragments, + 2 scale bytes when F8) are retired: the A reads issued after
        // them (2*TA fragments, + TA scale bytes when F8) may still be in flight
        if (TA == 4) { if (F8) asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); }
        else if (TA == 3) { if (F8) asm volatile("s_waitcnt lgkmcnt(9)" ::: "memory"); else asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory"); }
        else { if (F8) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); }
        CBAS_SEG_BARRIER();
        mfma_quadrant(H0{}, b0, 0);
        CBAS_SEG_BARRIER();
        // P2
        read_b(buf, scb, 1, b1);
        if (t + 2 < nk) stage(b, t + 2, 0);
        CBAS_SEG_BARRIER();
        mfma_quadrant(H0{}, b1, 1);
        CBAS_SEG_BARRIER();
        // P3
        read_a(buf, scb, H1{});
        if (t + 2 < nk) stage(b, t + 2, 1);
        CBAS_SEG_BARRIER();
        mfma_quadrant(H1{}, b1, 1);
        CBAS_SEG_BARRIER();
        // P4
        if (t + 2 < nk) {
            stage(b, t + 2, 2);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");    // K-tile t+1 has landed; 3 sub-tiles of t+2 in flight
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        CBAS_SEG_BARRIER();
        mfma_quadrant(H1{}, b0, 0);
        CBAS_SEG_BARRIER();
    };

    for (int id = first;;) {
        if (p.stamps) t_top = __builtin_amdgcn_s_memtime();
        asm volatile("" : "+v"(lane));
        set_tile(id);                                     // same values the K-tile 0 staging used; see `lane` above
        set_frag_offsets();
        // ---- rest of the prologue: K-tile 0 is in flight (issued above, or under the previous tile's epilogue);
        // three sub-tiles of K-tile 1 follow.  vmcnt counts in issue order, so vmcnt(6) also retires every store of
        // the previous tile's epilogue.
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        stage(1, 1, 0); stage(1, 1, 1); stage(1, 1, 2);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if constexpr (LNC) {
            // LayerNorm fold, consumer side: thread r of the first four waves pools the statistics of row row0 + r - one
            // record per N tile of the producer, fetched under the previous tile's epilogue (load_lnp) - into (mean, rstd)
            // for this tile's epilogue; every K-loop barrier lies between this write and those reads
            if (tid < BM) {
                const f32x2 st = ln_pool<4>(lnp, p.ln_parts, (float)LN_BLOCK);
                const float inv_d = 1.0f / (float)(p.ln_parts * LN_BLOCK);
                reinterpret_cast<f32x2*>(ln_lds)[tid] = f32x2{st[0] * inv_d, 1.0f / sqrtf(st[1] * inv_d + p.ln_eps)};
            }
        }
        if (p.stamps) { t_pro = __builtin_amdgcn_s_memtime(); r_pro = __builtin_amdgcn_s_memrealtime(); }
        if (wr == 1) __builtin_amdgcn_s_barrier();        // stagger the second wave group by one barrier

        for (int t = 0; t < nk; t += 2) {
            ktile(0, t);
            ktile(1, t + 1);
        }
        if (wr == 0) __builtin_amdgcn_s_barrier();        // re-align the groups: every wave has now left the loop

        if (p.stamps) { t_loop = __builtin_amdgcn_s_memtime(); r_loop = __builtin_amdgcn_s_memrealtime(); }
        const int erow = row0 + wr * WROWS, ecol = col0 + wc * 64;
        const int next = id + stride;
        const bool has_next = next < n_kind;
        asm volatile("" : "+v"(lane));
        if (has_next) set_tile(next);                     // the staging offsets now describe the NEXT tile
        // every wave is past the loop: buffer 0 is free, and the scratch below lies over buffer 1
        if constexpr (SPLIT) {
            // precision 4: the fp32-schedule epilogues of vit32_epilogue.h
            vit32_epilogue_tile<EPI, TM>(sp, erow, ecol, lane, acc, smem + BUF_BYTES + wave * 8192, [&] {
                if (has_next) { stage(0, 0, 0); stage(0, 0, 1); stage(0, 0, 2); stage(0, 0, 3); }
            }, rope_lds);
        } else
        gemm_epilogue_tile<EPI, TM>(p, erow, ecol, lane, acc, smem + BUF_BYTES + wave * 8192, [&] {
            if (has_next) { load_lnp(); stage(0, 0, 0); stage(0, 0, 1); stage(0, 0, 2); stage(0, 0, 3); }
        }, rope_lds, EPI == EPI_RESID_LN ? reinterpret_cast<f32x2*>(ln_lds) + (wr * WROWS) * 4 + wc
                                         : reinterpret_cast<f32x2*>(ln_lds) + wr * WROWS);
        if constexpr (EPI == EPI_RESID_LN) {
            // LayerNorm fold, producer side: the four column waves of a row group have left the statistics of their 64
            // columns in LDS; pool them into ONE record per row and N tile (the consumer then pools N / 256 records).
            // This barrier is also the one that frees the epilogue scratch for the next prologue.
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (tid < BM) {
                const f32x2* e = reinterpret_cast<const f32x2*>(ln_lds) + tid * 4;
                const f32x2 blk[4] = {e[0], e[1], e[2], e[3]};
                const int m = erow - wr * WROWS + tid;
                if (m < p.M) reinterpret_cast<f32x2*>(p.ln_out)[(size_t)((ecol - wc * 64) / LN_BLOCK) * p.ln_ld + m] = ln_pool<4>(blk, 4, 64.0f);
            }
        }
        if (p.stamps && id == first && tid == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            unsigned long long* o = p.stamps + (size_t)blockIdx.x * 4;
            o[0] = t_start; o[1] = t_pro; o[2] = t_loop; o[3] = __builtin_amdgcn_s_memtime();
            p.stamps[(size_t)GEMM_STAMP_BLOCKS * 4 + blockIdx.x] = r_loop - r_pro;     // in-kernel clock = loop ticks / this x 100 MHz
        }
        if (p.stamps && id != first && !has_next && tid == 0) {      // the workgroup's last tile (steady state of a persistent run)
            unsigned long long* o = p.stamps + (size_t)GEMM_STAMP_BLOCKS * 5 + (size_t)blockIdx.x * 4;
            o[0] = t_top; o[1] = t_pro; o[2] = t_loop; o[3] = __builtin_amdgcn_s_memtime();
        }
        if (!has_next) break;
        id = next;
        // the next prologue refills buffer 1: every wave must be done with its scratch (raw barrier: the K-tile 0
        // LDS-DMA stays in flight across it)
        if constexpr (EPI != EPI_RESID_LN) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
#undef CBAS_SEG_BARRIER
}

// TAIL = 1: blocks past g.main_blocks run 128x256 tiles over rows [g.tail_row0, M)
template <int EPI, int TA, int TB, int TAIL, bool F8>
__global__ __launch_bounds__(512, 2) void gemm_f16_8ph_kernel(GemmParams p, PPGrid g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tiles_n = p.N / BN;
    const int G = gridDim.x;                                   // a multiple of 8 whenever a workgroup gets > 1 tile
    // q|k|v: the factorised RoPE table lives in LDS behind the staging buffers for the life of the workgroup
    const float* rope_lds = nullptr;
    // LayerNorm fold: EPI_RESID_LN f32x2 [256 rows][4 column waves]; EPI_QKV_LN / EPI_GELU_LN f32x2 [256 rows] (mean, rstd),
    // behind the q|k|v kernel's RoPE table
    char* const ln_lds = smem + pp_lds_kernel<TA, TB, TAIL, F8>() +
                         (epi_base(EPI) == EPI_QKV && p.rope_fac ? (p.rope_nh + p.rope_nw) * 128 : 0);
    if (epi_base(EPI) == EPI_QKV && p.rope_fac) {
        float* dst = reinterpret_cast<float*>(smem + pp_lds_kernel<TA, TB, TAIL, F8>());
        const int n4 = (p.rope_nh + p.rope_nw) * 8;            // 16-byte pieces
        for (int i = threadIdx.x; i < n4; i += 512)
            reinterpret_cast<f32x4*>(dst)[i] = reinterpret_cast<const f32x4*>(p.rope_fac)[i];
        __syncthreads();
        rope_lds = dst;
    }
    int first = blockIdx.x;
    if (first < g.main_blocks) {
        pp_tiles<EPI, TA, TB, F8>(p, smem, first, g.main_blocks, G, 0, tiles_n, rope_lds, ln_lds);
        first += ((g.main_blocks - 1 - first) / G + 1) * G;    // this workgroup's first id past the main tiles
    }
    if (TAIL && first < g.n_tiles) {
        // the first tail tile's K-tile 0 is not prefetched across the kind switch: drain the LDS before restaging
        __syncthreads();
        pp_tiles<EPI, 2, 2, F8>(p, smem, first - g.main_blocks, g.n_tiles - g.main_blocks, G, g.tail_row0, tiles_n, rope_lds, ln_lds);
    }
}

// precision 4: the same loop on split operands (hi | lo halves of 32 k-values per 128-byte K-tile row), epilogues of the
// fp32 schedule.  p carries the loop's view of the problem (byte pointers in A8 / W8, lda and K in floats), sp the epilogue's.
template <int EPI, int TA, int TB, int TAIL>
__global__ __launch_bounds__(512, 2) void gemm_split_pp_kernel(GemmParams p, PPGrid g, Gemm32VitParams sp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tiles_n = p.N / BN;
    const int G = gridDim.x;
    // q|k|v: the RoPE angles by axis live in LDS behind the staging buffers for the life of the workgroup
    const float* rope_lds = nullptr;
    if (EPI == EPI_QKV && sp.rope_fac) {
        float* dst = reinterpret_cast<float*>(smem + pp_lds_kernel<TA, TB, TAIL, false>());
        const int n4 = (sp.rope_nh + sp.rope_nw) * 8;          // 16-byte pieces
        for (int i = threadIdx.x; i < n4; i += 512)
            reinterpret_cast<f32x4*>(dst)[i] = reinterpret_cast<const f32x4*>(sp.rope_fac)[i];
        __syncthreads();
        rope_lds = dst;
    }
    int first = blockIdx.x;
    if (first < g.main_blocks) {
        pp_tiles<EPI, TA, TB, false, true>(p, smem, first, g.main_blocks, G, 0, tiles_n, rope_lds, nullptr, sp);
        first += ((g.main_blocks - 1 - first) / G + 1) * G;
    }
    if (TAIL && first < g.n_tiles) {
        __syncthreads();
        pp_tiles<EPI, 2, 2, false, true>(p, smem, first - g.main_blocks, g.n_tiles - g.main_blocks, G, g.tail_row0, tiles_n, rope_lds, nullptr, sp);
    }
}

int pp_cus() {
    static const int cus = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    return cus;
}

template <int EPI, int TA, int TB, int TAIL, bool F8>
int launch_8ph(const GemmParams& p, int main_panels, hipStream_t stream) {
    constexpr int BM = 32 * (TA + TB);
    constexpr int lds = pp_lds_kernel<TA, TB, TAIL, F8>() + (epi_base(EPI) == EPI_QKV ? ROPE_LDS_ROWS * 128 : 0) +
                        (EPI == EPI_RESID_LN ? 256 * 4 * 8 : EPI == EPI_QKV_LN || EPI == EPI_GELU_LN ? 256 * 8 : 0);
    static_assert(lds <= 160 * 1024, "tile does not fit the 160 KiB LDS");
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16_8ph_kernel<EPI, TA, TB, TAIL, F8>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return -2;
        attr_set = true;
    }
    const int tiles_n = p.N / BN;
    PPGrid g;
    if (TAIL) {
        g.main_blocks = main_panels * tiles_n;
        g.tail_row0 = main_panels * BM;
        if (g.tail_row0 >= p.M) return -1;
        g.n_tiles = g.main_blocks + ((p.M - g.tail_row0 + 127) / 128) * tiles_n;
    } else {
        g.n_tiles = ((p.M + BM - 1) / BM) * tiles_n;
        g.main_blocks = g.n_tiles;
        g.tail_row0 = p.M;
    }
    // persistent: one workgroup per CU walks the tile ids with a stride of the grid size (a multiple of 8, so a
    // workgroup's tiles stay in its XCD's contiguous range of the remap)
    static const int persist = [] { const char* e = getenv("CBAS_PP_PERSIST"); return e ? atoi(e) : 1; }();
    const int slots = pp_cus() & ~7;
    const int grid = persist && g.n_tiles > slots ? slots : g.n_tiles;
    // the launch asks only for the table rows this grid has (224^2: 3.5 KiB), not the ROPE_LDS_ROWS the attribute allows:
    // what is left of the 160 KiB decides which kernels of the other compute lane can share the CU
    const int lds_launch = pp_lds_kernel<TA, TB, TAIL, F8>() + (epi_base(EPI) == EPI_QKV && p.rope_fac ? (p.rope_nh + p.rope_nw) * 128 : 0) +
                           (EPI == EPI_RESID_LN ? 256 * 4 * 8 : EPI == EPI_QKV_LN || EPI == EPI_GELU_LN ? 256 * 8 : 0);
    hipLaunchKernelGGL((gemm_f16_8ph_kernel<EPI, TA, TB, TAIL, F8>), dim3(grid), dim3(512), lds_launch, stream, p, g);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// Makespan of a launch on `slots` CUs, in units of one 256x256 tile: tiles are handed out in block
// order to whichever CU frees up first.  Relative tile costs measured with scripts/gemm_stamps.py
// (the smaller tiles are bound by the ~30 B/cycle/CU L2->LDS rate, not by the MFMA pipe).
// The split form's loop is three times as long per staged byte and MFMA-bound at every tile height: cost ~ rows.
double pp_tile_cost(int bm, bool split = false) {
    if (split) return 0.05 + 0.95 * bm / 256.0;
    return bm == 256 ? 1.0 : bm == 192 ? 0.86 : bm == 160 ? 0.83 : 0.66;
}

double pp_makespan(int n_main, double c_main, int n_tail, double c_tail, int slots) {
    std::priority_queue<double, std::vector<double>, std::greater<double>> free_at;
    for (int i = 0; i < slots; ++i) free_at.push(0.0);
    double end = 0.0;
    auto run = [&](int n, double c) {
        for (int i = 0; i < n; ++i) {
            const double t = free_at.top() + c;
            free_at.pop();
            free_at.push(t);
            if (t > end) end = t;
        }
    };
    run(n_main, c_main);
    run(n_tail, c_tail);
    return end;
}

struct PPPlan { int bm; int main_panels; };

// plan for an (M, N) problem on this device; computed once per shape
PPPlan pp_plan(int M, int N, bool split = false) {
    static std::mutex mu;
    static std::map<std::pair<int, int>, PPPlan> cache;            // key: (M, +-N)
    static const int cus = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    std::lock_guard<std::mutex> lock(mu);
    const std::pair<int, int> key{M, split ? -N : N};
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    const int tiles_n = N / BN;
    PPPlan best{256, 0};
    double best_t = 1e30;
    for (int bm : {256, 192, 160}) {
        const int n = ((M + bm - 1) / bm) * tiles_n;
        const double t = pp_makespan(n, pp_tile_cost(bm, split), 0, 0.0, cus);
        if (t < best_t - 1e-9) { best = {bm, 0}; best_t = t; }
    }
    static const bool no_tail = [] { const char* e = getenv("CBAS_PP_NO_TAIL"); return e && e[0] == '1'; }();   // experiments
    for (int mp = 1; mp * 256 < M && !no_tail; ++mp) {
        const int n_main = mp * tiles_n, n_tail = ((M - mp * 256 + 127) / 128) * tiles_n;
        const double t = pp_makespan(n_main, 1.0, n_tail, pp_tile_cost(128, split), cus);
        if (t < best_t - 0.02) { best = {256, mp}; best_t = t; }      // prefer a uniform grid on near-ties
    }
    cache[key] = best;
    return best;
}

template <int EPI, bool F8>
int launch_8ph_epi(const GemmParams& p, int tile, hipStream_t stream) {
    if (tile == GEMM_TILE_PP_192x256) return launch_8ph<EPI, 3, 3, 0, F8>(p, 0, stream);
    if (tile == GEMM_TILE_PP_160x256) return launch_8ph<EPI, 3, 2, 0, F8>(p, 0, stream);
    if (tile == GEMM_TILE_PP_128x256) return launch_8ph<EPI, 2, 2, 0, F8>(p, 0, stream);
    if (tile == GEMM_TILE_PP_256x256) return launch_8ph<EPI, 4, 4, 0, F8>(p, 0, stream);
    // GEMM_TILE_PP_AUTO: uniform 256 / 192 / 160-row tiles, or 256-row tiles with a 128-row tail
    PPPlan plan = pp_plan(p.M, p.N);
    static const int mp_env = [] { const char* e = getenv("CBAS_PP_MAIN_PANELS"); return e ? atoi(e) : -1; }();
    if (mp_env >= 0 && mp_env * 256 < p.M) plan = {256, mp_env};                 // experiments only
    if (plan.main_panels) return launch_8ph<EPI, 4, 4, 1, F8>(p, plan.main_panels, stream);
    if (plan.bm == 192) return launch_8ph<EPI, 3, 3, 0, F8>(p, 0, stream);
    if (plan.bm == 160) return launch_8ph<EPI, 3, 2, 0, F8>(p, 0, stream);
    return launch_8ph<EPI, 4, 4, 0, F8>(p, 0, stream);
}

template <int EPI, int TA, int TB, int TAIL>
int launch_split_pp(const GemmParams& p, const Gemm32VitParams& sp, int main_panels, hipStream_t stream) {
    constexpr int BM = 32 * (TA + TB);
    constexpr int lds = pp_lds_kernel<TA, TB, TAIL, false>() + (EPI == EPI_QKV ? ROPE_LDS_ROWS * 128 : 0);
    static_assert(lds <= 160 * 1024, "tile does not fit the 160 KiB LDS");
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_split_pp_kernel<EPI, TA, TB, TAIL>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return -2;
        attr_set = true;
    }
    const int tiles_n = p.N / BN;
    PPGrid g;
    if (TAIL) {
        g.main_blocks = main_panels * tiles_n;
        g.tail_row0 = main_panels * BM;
        if (g.tail_row0 >= p.M) return -1;
        g.n_tiles = g.main_blocks + ((p.M - g.tail_row0 + 127) / 128) * tiles_n;
    } else {
        g.n_tiles = ((p.M + BM - 1) / BM) * tiles_n;
        g.main_blocks = g.n_tiles;
        g.tail_row0 = p.M;
    }
    const int slots = pp_cus() & ~7;
    const int grid = g.n_tiles > slots ? slots : g.n_tiles;
    const int lds_launch = pp_lds_kernel<TA, TB, TAIL, false>() + (EPI == EPI_QKV && sp.rope_fac ? (sp.rope_nh + sp.rope_nw) * 128 : 0);
    hipLaunchKernelGGL((gemm_split_pp_kernel<EPI, TA, TB, TAIL>), dim3(grid), dim3(512), lds_launch, stream, p, g, sp);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int g_split_tile = 0;                          // bring-up (gemm_split_pp_set_tile): forced tile height, stamps buffer
unsigned long long* g_split_stamps = nullptr;

template <int EPI>
int launch_split_pp_epi(const GemmParams& p, const Gemm32VitParams& sp, hipStream_t stream) {
    static const int tile_env0 = [] { const char* e = getenv("CBAS_SPLIT_TILE"); return e ? atoi(e) : 0; }();   // experiments
    const int tile_env = g_split_tile ? g_split_tile : tile_env0;
    if (tile_env == 256) return launch_split_pp<EPI, 4, 4, 0>(p, sp, 0, stream);
    if (tile_env == 192) return launch_split_pp<EPI, 3, 3, 0>(p, sp, 0, stream);
    if (tile_env == 160) return launch_split_pp<EPI, 3, 2, 0>(p, sp, 0, stream);
    if (tile_env == 128) return launch_split_pp<EPI, 2, 2, 0>(p, sp, 0, stream);
    const PPPlan plan = pp_plan(p.M, p.N, true);
    if (plan.main_panels) return launch_split_pp<EPI, 4, 4, 1>(p, sp, plan.main_panels, stream);
    if (plan.bm == 192) return launch_split_pp<EPI, 3, 3, 0>(p, sp, 0, stream);
    if (plan.bm == 160) return launch_split_pp<EPI, 3, 2, 0>(p, sp, 0, stream);
    return launch_split_pp<EPI, 4, 4, 0>(p, sp, 0, stream);
}

}  // namespace

void gemm_split_pp_set_tile(int tile, unsigned long long* stamps) { g_split_tile = tile; g_split_stamps = stamps; }

// precision 4, large M: -1 when the shape is not this kernel's (the caller keeps its 128 x 128 kernels for those)
int launch_gemm_split_pp(GemmEpilogue epi, const Gemm32VitParams& sp_in, hipStream_t stream) {
    Gemm32VitParams sp = sp_in;
    // the LDS copy of the RoPE angles, or the global [P][64] tables
    if (sp.rope_fac && (!sp.rope_cos || sp.rope_nw <= 1 || sp.rope_nh + sp.rope_nw > ROPE_LDS_ROWS)) sp.rope_fac = nullptr;
    if (!sp.split || sp.N % 256 || sp.K % 64 || sp.K < 64 || sp.M < 1) return -1;
    if ((long long)sp.M * sp.lda * 4 >= (1ll << 31) || (long long)sp.N * sp.K * 4 >= (1ll << 31)) return -1;
    GemmParams p{};
    p.A8 = reinterpret_cast<const uint8_t*>(sp.A);
    p.W8 = reinterpret_cast<const uint8_t*>(sp.W);
    p.lda = (int)sp.lda;
    p.M = sp.M;
    p.M_pad = sp.M;
    p.N = sp.N;
    p.K = sp.K;
    p.stamps = g_split_stamps;
    // Raster (r5, scripts/gemm_gm_ab.sh, profiles/r05_gemm_raster_ab.json): where a row of tiles is long (N >= 2048: q|k|v, up) groups
    // of 6 row panels, row-fastest inside a group, so that the ~32 tiles an XCD runs at a time form a 6 x 5 block and share
    // operand panels in its L2 - `up` fetches 20 % less from the fabric; the N = 768 GEMMs (three column tiles per row panel: a
    // compact block already) fetch 18-24 % MORE when grouped and keep N-fastest.  +1.0 % frames/s at precision 4 against
    // N-fastest everywhere, rows identical.  The fp16 kernels lose 0.5 % with groups and keep N-fastest: operands here are twice
    // the bytes and the step is bound by energy - fabric traffic saved comes back as clock.
    // CBAS_GEMM_GM=<n> forces n everywhere (1 = N-fastest), CBAS_GEMM_GM_WIDE=<n> the wide shapes only.
    static const int gm_env = [] { const char* e = getenv("CBAS_GEMM_GM"); return e ? atoi(e) : 0; }();
    static const int gmw_env = [] { const char* e = getenv("CBAS_GEMM_GM_WIDE"); return e ? atoi(e) : 0; }();      // the wide shapes only
    p.group_m = gm_env > 0 ? gm_env : (p.N / 256 >= 8 ? (gmw_env > 0 ? gmw_env : 6) : 1);
    switch (epi) {
        case EPI_PATCH: return launch_split_pp_epi<EPI_PATCH>(p, sp, stream);
        case EPI_QKV:   return launch_split_pp_epi<EPI_QKV>(p, sp, stream);
        case EPI_RESID: return launch_split_pp_epi<EPI_RESID>(p, sp, stream);
        case EPI_GELU:  return launch_split_pp_epi<EPI_GELU>(p, sp, stream);
        default: return -1;
    }
}

int launch_gemm_8ph(GemmEpilogue epi, const GemmParams& p_in, int tile, hipStream_t stream) {
    GemmParams p = p_in;
    // the LDS copy of the RoPE table, or the global [P][64] table (the caller passes rope_fac = nullptr to force the latter)
    if (p.rope_fac && (!p.rope_cos || p.rope_nw <= 1 || p.rope_nh + p.rope_nw > ROPE_LDS_ROWS)) p.rope_fac = nullptr;
    // 32-bit byte offsets into A / W; K-tiles are consumed in pairs
    if (p.W_lo || p.N % 256 || (long long)p.M_pad * p.lda >= (1ll << 31) || (long long)p.N * p.K >= (1ll << 31)) return -1;
    if (p.A8) {                                        // MX-fp8 operands: 128-element K-tiles
        if (p.K % (4 * BK) || !p.W8 || !p.A_sc || !p.W_sc) return -1;
        switch (epi) {
            case EPI_QKV:     return launch_8ph_epi<EPI_QKV, true>(p, tile, stream);
            case EPI_RESID:   return launch_8ph_epi<EPI_RESID, true>(p, tile, stream);
            case EPI_GELU_F8: return launch_8ph_epi<EPI_GELU_F8, true>(p, tile, stream);
            default: return -1;
        }
    }
    if (p.K % (2 * BK)) return -1;
    switch (epi) {
        case EPI_PATCH: return launch_8ph_epi<EPI_PATCH, false>(p, tile, stream);
        case EPI_QKV:   return launch_8ph_epi<EPI_QKV, false>(p, tile, stream);
        case EPI_RESID: return launch_8ph_epi<EPI_RESID, false>(p, tile, stream);
        case EPI_GELU:  return launch_8ph_epi<EPI_GELU, false>(p, tile, stream);
        // LayerNorm fold (fp16 only): the statistics are pooled over N tiles of 256 columns, at most four of them
        case EPI_QKV_LN:   return p.ln_in && p.ln_colsum && p.ln_parts >= 1 && p.ln_parts <= 4 && p.K == p.ln_parts * 256
                                  ? launch_8ph_epi<EPI_QKV_LN, false>(p, tile, stream) : -1;
        case EPI_GELU_LN:  return p.ln_in && p.ln_colsum && p.ln_parts >= 1 && p.ln_parts <= 4 && p.K == p.ln_parts * 256
                                  ? launch_8ph_epi<EPI_GELU_LN, false>(p, tile, stream) : -1;
        case EPI_RESID_LN: return p.x16_out && p.ln_out ? launch_8ph_epi<EPI_RESID_LN, false>(p, tile, stream) : -1;
        default: return -1;
    }
}

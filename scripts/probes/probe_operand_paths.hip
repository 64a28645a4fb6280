// Which path feeds a GEMM tile's operands fastest on one CU while EVERY CU does the same?  The fp16 / fp8 / split ping-pong
// kernels stage both operands by 16-byte LDS-DMA and their K loops sit at ~28 B/cycle/CU; the epilogues'
// plain global loads into registers were measured at ~50 B/cycle/CU.  This probe streams the operands of 256 x 256 tiles of
// the ViT-B o_proj shape (A: 12 864 x 768 fp16 panel, B: 768 x 768 fp16, K-tiles of 64 elements = 128-byte row segments at a
// 1 536-byte pitch: the kernels' own access pattern) with nothing else going on, 8 waves per CU, one workgroup per CU:
//   mode 0: A and B by LDS-DMA (global_load_lds, 16 B per lane) into a two-deep LDS ring       = what the kernels do
//   mode 1: A and B by global_load_dwordx4 into registers (two register sets, software pipelined)
//   mode 2: B by LDS-DMA, A by register loads                                                   = "A operand in registers"
//   mode 3: B by LDS-DMA, A by register loads issued TWICE (each A row block is needed by the two waves of a row group)
// and prints bytes per core-clock cycle per CU (s_memtime) and GB/s per CU (s_memrealtime, 100 MHz).
//      usage: probe_operand_paths [tile_steps = 40 [tiles_m = 4]]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int M_ROWS = 13056, N_ROWS = 768, PITCH = 1536, KT = 12, TILES_N = 3;

// pieces (64 rows x 128 B = 8 KB) per K-tile on each path; LDSR = ds_read_b128 per lane per K-tile of the slot that has landed
// (24 = what a 128 x 64 wave tile reads: 16 KB of A + 8 KB of B fragments per wave)
template <int DMA_A, int DMA_B, int REG_A, int REG_B, int LDSR = 0, int FRAG = 0>
__global__ __launch_bounds__(512) void stream_kernel(const uint8_t* __restrict__ A, const uint8_t* __restrict__ B, int steps, int TILES_M,
                                                     unsigned long long* out, unsigned* sink) {
    extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];
    const int t = threadIdx.x, r = t >> 3, c = t & 7;
    const int wg = blockIdx.x;
    uint4 acc = {0, 0, 0, 0};
    constexpr int NREG = REG_A + REG_B, NDMA = DMA_A + DMA_B;
    uint4 cur[NREG > 0 ? NREG : 1], nxt[NREG > 0 ? NREG : 1];
    auto issue = [&](int it, uint4* regs) {
        const int s = it / KT, kt = it - s * KT;
        const int tile = (wg + s * 256) % (TILES_M * TILES_N);
        const uint8_t* a = A + (size_t)((tile / TILES_N) * 256 + r) * PITCH + kt * 128 + c * 16;
        const uint8_t* b = B + (size_t)((tile % TILES_N) * 256 + r) * PITCH + kt * 128 + c * 16;
        uint8_t* ring = lds + (it & 1) * 65536 + __builtin_amdgcn_readfirstlane(t >> 6) * 1024;   // this wave's 1 KB of a piece
#pragma unroll
        for (int p = 0; p < DMA_A; ++p)
            __builtin_amdgcn_global_load_lds(GLB_PTR(a + (size_t)(p & 3) * 64 * PITCH), LDS_PTR(ring + p * 8192), 16, 0, 0);
#pragma unroll
        for (int p = 0; p < DMA_B; ++p)
            __builtin_amdgcn_global_load_lds(GLB_PTR(b + (size_t)(p & 3) * 64 * PITCH), LDS_PTR(ring + (DMA_A + p) * 8192), 16, 0, 0);
        if (FRAG) {
            // MFMA fragment layout, 8 waves as 4 (M) x 2 (N): wave w owns rows 64 (w >> 1) ... + 63 of the tile and asks for them
            // itself - load p: row tile p >> 1, k-half p & 1; lane -> row lane & 15, 16 bytes at 16 (lane >> 4) + 64 (p & 1):
            // 16 rows x 64 bytes per instruction, every line asked for by two loads of this wave and two of its neighbour
            const uint8_t* af = A + (size_t)((tile / TILES_N) * 256 + (t >> 7) * 64 + (t & 15)) * PITCH + kt * 128 + ((t >> 4) & 3) * 16;
#pragma unroll
            for (int p = 0; p < REG_A; ++p) regs[p] = *reinterpret_cast<const uint4*>(af + (size_t)(p >> 1) * 16 * PITCH + (p & 1) * 64);
        } else {
#pragma unroll
            for (int p = 0; p < REG_A; ++p) regs[p] = *reinterpret_cast<const uint4*>(a + (size_t)(p & 3) * 64 * PITCH);
        }
#pragma unroll
        for (int p = 0; p < REG_B; ++p) regs[REG_A + p] = *reinterpret_cast<const uint4*>(b + (size_t)(p & 3) * 64 * PITCH);
    };
    auto read_lds = [&](int it) {
        const uint8_t* slot = lds + (it & 1) * 65536 + (t & 63) * 16 + ((t >> 6) & 1) * (LDSR <= 16 ? 16384 : 32768);
#pragma unroll
        for (int j = 0; j < LDSR; ++j) {
            const uint4 v = *reinterpret_cast<const uint4*>(slot + j * 1024);
            acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
        }
    };
    auto consume = [&](uint4* regs) {
#pragma unroll
        for (int p = 0; p < NREG; ++p) { acc.x ^= regs[p].x; acc.y ^= regs[p].y; acc.z ^= regs[p].z; acc.w ^= regs[p].w; }
    };
    const int iters = steps * KT;                 // even
    __builtin_amdgcn_s_barrier();
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    issue(0, cur);
    for (int it = 0; it < iters; it += 2) {
        issue(it + 1, nxt);
        if (NREG > 0) consume(cur);
        else if (NDMA > 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NDMA) : "memory");     // K-tile `it` has landed
        __builtin_amdgcn_s_barrier();
        if (LDSR > 0) { read_lds(it); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
        if (it + 2 < iters) issue(it + 2, cur);
        if (NREG > 0) consume(nxt);
        else if (NDMA > 0) { if (it + 2 < iters) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NDMA) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        __builtin_amdgcn_s_barrier();
        if (LDSR > 0) { read_lds(it + 1); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    if (t == 0) { out[wg * 2] = c1 - c0; out[wg * 2 + 1] = w1 - w0; }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[wg * 512 + t] = acc.x + lds[t];
}

template <int DA, int DB, int RA, int RB, int LR = 0, int FR = 0>
static void run(const char* what, const uint8_t* A, const uint8_t* B, int steps, int tiles_m, unsigned long long* d_out, unsigned* sink, int n_cu) {
    auto k = stream_kernel<DA, DB, RA, RB, LR, FR>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(n_cu), dim3(512), 131072, 0, A, B, steps, tiles_m, d_out, sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
    }
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(n_cu * 2);
    CK(hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost));
    double cyc = 0, wall = 0;
    for (int i = 0; i < n_cu; ++i) { cyc += h[i * 2]; wall += h[i * 2 + 1]; }
    cyc /= n_cu; wall /= n_cu;
    const double bytes = (DA + DB + RA + RB) ? (double)steps * KT * (DA + DB + RA + RB) * 8192.0 : (double)steps * KT * LR * 8192.0;
    printf("{\"mode\": \"%s\", \"dma_pieces\": %d, \"reg_pieces\": %d, \"bytes_per_cu\": %.0f, \"B_per_cycle_per_cu\": %.2f, "
           "\"GB_s_per_cu\": %.1f, \"chip_TB_s\": %.2f, \"core_clock_GHz\": %.3f, \"kernel_ms\": %.3f}\n",
           what, DA + DB, RA + RB, bytes, bytes / cyc, bytes / (wall * 10.0), bytes * n_cu / (ms * 1e-3) / 1e12, cyc / (wall * 10.0), ms);
    fflush(stdout);
}

int main(int argc, char** argv) {
    int steps = argc > 1 ? atoi(argv[1]) : 40;
    // rows of A in play = 256 x tiles_m: 51 = the whole 12 864-row panel (19.8 MB: most of it comes over the fabric, and with
    // this probe's plain wg -> tile map the three column tiles of a row block sit on three XCDs), 4 = 1.5 MB (L2-resident on
    // every XCD: the CU-side capacity of each path)
    int tiles_m = argc > 2 ? atoi(argv[2]) : 4;
    if (steps < 1 || steps > 4000 || tiles_m < 1 || tiles_m > 51) { printf("tile_steps / tiles_m out of range\n"); return 1; }
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    uint8_t *A, *B;
    unsigned long long* d_out;
    unsigned* sink;
    CK(hipMalloc(&A, (size_t)M_ROWS * PITCH)); CK(hipMalloc(&B, (size_t)N_ROWS * PITCH));
    CK(hipMemset(A, 1, (size_t)M_ROWS * PITCH)); CK(hipMemset(B, 2, (size_t)N_ROWS * PITCH));
    CK(hipMalloc(&d_out, n_cu * 16)); CK(hipMalloc(&sink, (size_t)n_cu * 512 * 4));
    printf("{\"device\": \"%s\", \"cus\": %d, \"tile_steps\": %d, \"k_tiles_per_step\": %d, \"a_rows\": %d}\n", prop.name, n_cu, steps, KT, tiles_m * 256);
    run<4, 4, 0, 0>("A and B by LDS-DMA (the kernels' staging)", A, B, steps, tiles_m, d_out, sink, n_cu);
    run<4, 4, 0, 0, 24>("A and B by LDS-DMA while every wave reads 24 KB of fragments per K-tile from the landed slot (ds_read_b128)", A, B, steps, tiles_m, d_out, sink, n_cu);
    run<0, 0, 0, 0, 24>("no operand traffic: only the 24 KB of fragment reads per wave per K-tile (bytes = LDS bytes read)", A, B, steps, tiles_m, d_out, sink, n_cu);
    run<4, 4, 0, 0, 12>("the same with 12 KB of fragment reads per wave", A, B, steps, tiles_m, d_out, sink, n_cu);
    run<0, 0, 4, 4>("A and B by global_load_dwordx4 into registers", A, B, steps, tiles_m, d_out, sink, n_cu);
    run<0, 4, 4, 0>("B by LDS-DMA, A into registers", A, B, steps, tiles_m, d_out, sink, n_cu);
    run<0, 4, 8, 0>("B by LDS-DMA, A into registers twice (both waves of a row group)", A, B, steps, tiles_m, d_out, sink, n_cu);
    run<0, 4, 8, 0, 0, 1>("B by LDS-DMA, A into registers in MFMA fragment layout by the wave that uses it (64 KB asked for, 32 KB distinct)", A, B, steps, tiles_m, d_out, sink, n_cu);
    run<0, 4, 8, 0, 16, 1>("the same while every wave reads its 16 KB of B fragments per K-tile from the landed slot", A, B, steps, tiles_m, d_out, sink, n_cu);
    run<0, 4, 0, 0>("B by LDS-DMA alone (half the bytes)", A, B, steps, tiles_m, d_out, sink, n_cu);
    run<0, 0, 4, 0>("A into registers alone (half the bytes)", A, B, steps, tiles_m, d_out, sink, n_cu);
    return 0;
}

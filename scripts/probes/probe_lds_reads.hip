// What does an LDS read instruction cost on gfx950?  One workgroup of 512 threads per CU (8 waves: 2 per SIMD), every wave
// issues `iters` rounds of 16 reads of one kind from conflict-free addresses (the attention kernels' own swizzled patterns),
// results XOR-ed into a sink.  Prints cycles per wave-instruction as seen by the CU's LDS (s_memtime over the loop x
// clock ratio), i.e. the time the CU's LDS pipeline is occupied per instruction when all waves hammer it.
//   kind 0: ds_read_b128 (K fragments: 64 lanes x 16 B)            kind 1: ds_read_b64 (64 lanes x 8 B)
//   kind 2: ds_read_b64_tr_b16 (V fragments of the attention kernels: transposing read, 64 lanes x 8 B)
//   kind 3: ds_read_b32
// Build: hipcc --offload-arch=gfx950 -O3 -o scripts/probes/bin/probe_lds_reads scripts/probes/probe_lds_reads.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int sk_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int sv_off(int row, int col) { return row * 128 + ((((col >> 4) ^ ((row >> 1) & 3))) << 5) + ((col & 15) << 1); }

template <int KIND>
__global__ __launch_bounds__(512) void probe(int iters, unsigned long long* cycles, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];        // 32 KiB: 4 images of 64 rows x 128 B
    const int lane = threadIdx.x & 63, g = lane >> 4, li = lane & 15;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) reinterpret_cast<unsigned*>(smem)[i] = i * 2654435761u;
    __syncthreads();
    unsigned acc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const char* img = smem + ((j + it) & 3) * 8192;          // varies with `it`: the compiler cannot hoist the read out of the loop
            if (KIND == 0) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(img + sk_off((j >> 2) * 16 + li, (4 * (j & 1) + g)));
                acc ^= __float_as_uint(v[0]) ^ __float_as_uint(v[3]);
            } else if (KIND == 1) {
                const f32x2 v = *reinterpret_cast<const f32x2*>(img + sk_off((j >> 2) * 16 + li, (4 * (j & 1) + g)) + 8 * (j & 1));
                acc ^= __float_as_uint(v[0]) ^ __float_as_uint(v[1]);
            } else if (KIND == 2) {
                const int krow = 32 * (j & 1) + 4 * g + (li >> 2), col = 16 * ((j >> 2) & 3) + 4 * (li & 3);
                const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + sv_off(krow, col)));
                acc ^= (unsigned)v[0] ^ ((unsigned)v[3] << 16);
            } else {
                const float v = *reinterpret_cast<const float*>(img + ((j * 64 + lane) & 2047) * 4);
                acc ^= __float_as_uint(v);
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    if (acc == 0x12345678u) sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    int cus = 256; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    unsigned long long* cyc; unsigned* sink;
    hipMalloc(&cyc, cus * 8); hipMalloc(&sink, (size_t)cus * 512 * 4);
    const char* names[4] = {"ds_read_b128", "ds_read_b64", "ds_read_b64_tr_b16", "ds_read_b32"};
    const int bytes[4] = {1024, 512, 512, 256};
    for (int kind = 0; kind < 4; ++kind) {
        for (int waves : {8, 4, 1}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            auto launch = [&]() {
                dim3 grid(cus), block(64 * waves);
                if (kind == 0) hipLaunchKernelGGL(probe<0>, grid, block, 32768, 0, iters, cyc, sink);
                if (kind == 1) hipLaunchKernelGGL(probe<1>, grid, block, 32768, 0, iters, cyc, sink);
                if (kind == 2) hipLaunchKernelGGL(probe<2>, grid, block, 32768, 0, iters, cyc, sink);
                if (kind == 3) hipLaunchKernelGGL(probe<3>, grid, block, 32768, 0, iters, cyc, sink);
            };
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(cus); hipMemcpy(h.data(), cyc, cus * 8, hipMemcpyDeviceToHost);
            double avg = 0; for (auto c : h) avg += (double)c; avg /= cus;
            const double insns = (double)iters * 16 * waves;               // wave-instructions per CU
            // s_memtime ticks at 100 MHz on this part; report wall-time based figures: ns per instruction per CU
            printf("{\"read\": \"%s\", \"waves_per_cu\": %d, \"ns_per_wave_instruction_per_cu\": %.3f, \"bytes_per_ns_per_cu\": %.1f, \"memtime_ticks\": %.0f}\n",
                   names[kind], waves, ms * 1e6 / insns, bytes[kind] / (ms * 1e6 / insns), avg);
        }
    }
    return 0;
}

#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* out) {
    unsigned a = threadIdx.x;
    u32x2 r16 = __builtin_amdgcn_permlane16_swap(a, a, false, false);
    u32x2 r32 = __builtin_amdgcn_permlane32_swap(a, a, false, false);
    out[threadIdx.x * 4 + 0] = r16[0]; out[threadIdx.x * 4 + 1] = r16[1];
    out[threadIdx.x * 4 + 2] = r32[0]; out[threadIdx.x * 4 + 3] = r32[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 64 * 16); k<<<1, 64>>>(d); unsigned h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 8) printf("lane %2d: p16 (%2u,%2u) p32 (%2u,%2u)\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    for (int l = 0; l < 64; ++l) { if ((h[l*4] ^ h[l*4+1]) != 16 || (h[l*4+2] ^ h[l*4+3]) != 32 || (h[l*4] != (unsigned)l && h[l*4+1] != (unsigned)l) || (h[l*4+2] != (unsigned)l && h[l*4+3] != (unsigned)l)) { printf("MISMATCH at %d\n", l); return 1; } }
    printf("OK: {r[0], r[1]} = {lane value, value of lane^16 / lane^32} in some order for every lane\n");
    return 0;
}

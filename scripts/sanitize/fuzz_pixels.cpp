// ASan / UBSan harness for cbas_pick_channel_u8 (csrc/host_pixels.cpp): random pixel counts, channel counts and thread
// counts, source and destination in EXACT-size heap blocks (an over-read or over-write of one byte trips the sanitizer),
// every output byte compared with the plain strided copy.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <random>
extern "C" int cbas_pick_channel_u8(const uint8_t*, int64_t, int32_t, int32_t, uint8_t*, int32_t);
thread_local char g_cbas_err[512];
int main() {
    std::mt19937_64 rng(7);
    long total = 0;
    for (int it = 0; it < 4000; ++it) {
        const int nc = 1 + (int)(rng() % 4);
        const int ch = (int)(rng() % nc);
        int64_t n = (it % 50 == 0) ? (int64_t)(rng() % 3000000) : (int64_t)(rng() % 5000);
        if (it < 70) n = it;                                    // every small size, incl. 0 and the 16-pixel vector boundary
        const int threads = 1 + (int)(rng() % 20);
        uint8_t* src = (uint8_t*)malloc((size_t)(n * nc) + (n == 0));
        uint8_t* dst = (uint8_t*)malloc((size_t)n + (n == 0));
        for (int64_t i = 0; i < n * nc; ++i) src[i] = (uint8_t)rng();
        memset(dst, 0xAB, (size_t)n);
        const int rc = cbas_pick_channel_u8(src, n, nc, ch, dst, threads);
        if (rc != 0) { printf("rc %d at n=%lld nc=%d\n", rc, (long long)n, nc); return 1; }
        for (int64_t i = 0; i < n; ++i)
            if (dst[i] != src[i * nc + ch]) { printf("mismatch at %lld (n=%lld nc=%d ch=%d threads=%d)\n", (long long)i, (long long)n, nc, ch, threads); return 1; }
        total += n;
        free(src); free(dst);
    }
    if (cbas_pick_channel_u8(nullptr, 1, 3, 1, nullptr, 1) == 0 || cbas_pick_channel_u8((const uint8_t*)"x", 1, 3, 3, (uint8_t*)g_cbas_err, 1) == 0) {
        printf("argument checks missing\n"); return 1;
    }
    printf("ok %ld pixels\n", total);
}

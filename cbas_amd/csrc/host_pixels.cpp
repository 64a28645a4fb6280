// Host-side pixel staging of the C ABI: pick ONE channel of interleaved uint8 frames.
//
// The reference decodes RGB frames (decord ... .asnumpy(), backend/cbas.py:425) and keeps channel 1 (:431): two thirds of
// every decoded frame are never used.  The decode-ahead thread calls this while it fills the page-locked ring, so that only
// the consumed plane - 50 176 bytes of a 224 x 224 frame instead of 150 528 - crosses PCIe and sits in the staging buffers
// (SURVEY section 8(d)'s byte count).  3-channel input goes through an SSSE3 byte shuffle (48 bytes in, 16 out per step),
// anything else through the plain strided loop; the work is split over threads by pixel ranges.
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include "api_common.h"

namespace {

#if defined(__x86_64__)
__attribute__((target("ssse3"))) void pick3_ssse3(const uint8_t* src, int64_t n, int channel, uint8_t* dst) {
    // output byte j of a 16-pixel group comes from input byte 3 j + channel: bytes 0-15 / 16-31 / 32-47 of the group
    alignas(16) int8_t m[3][16];
    for (int part = 0; part < 3; ++part)
        for (int j = 0; j < 16; ++j) {
            const int b = 3 * j + channel - 16 * part;
            m[part][j] = (b >= 0 && b < 16) ? (int8_t)b : (int8_t)-128;       // high bit set: pshufb writes zero
        }
    const __m128i m0 = _mm_load_si128((const __m128i*)m[0]), m1 = _mm_load_si128((const __m128i*)m[1]),
                  m2 = _mm_load_si128((const __m128i*)m[2]);
    int64_t i = 0;
    for (; i + 16 <= n; i += 16) {
        const uint8_t* s = src + 3 * i;
        const __m128i a = _mm_loadu_si128((const __m128i*)s), b = _mm_loadu_si128((const __m128i*)(s + 16)),
                      c = _mm_loadu_si128((const __m128i*)(s + 32));
        const __m128i r = _mm_or_si128(_mm_or_si128(_mm_shuffle_epi8(a, m0), _mm_shuffle_epi8(b, m1)), _mm_shuffle_epi8(c, m2));
        _mm_storeu_si128((__m128i*)(dst + i), r);
    }
    for (; i < n; ++i) dst[i] = src[3 * i + channel];
}
#endif

void pick_range(const uint8_t* src, int64_t n, int n_channels, int channel, uint8_t* dst) {
#if defined(__x86_64__)
    if (n_channels == 3 && __builtin_cpu_supports("ssse3")) { pick3_ssse3(src, n, channel, dst); return; }
#endif
    for (int64_t i = 0; i < n; ++i) dst[i] = src[i * n_channels + channel];
}

}  // namespace

extern "C" int cbas_pick_channel_u8(const uint8_t* src, int64_t n_pixels, int32_t n_channels, int32_t channel, uint8_t* dst,
                                    int32_t n_threads) {
    if (!src || !dst) return cbas_fail(CBAS_EINVAL, "null argument");
    if (n_pixels < 0 || n_channels < 1 || channel < 0 || channel >= n_channels)
        return cbas_fail(CBAS_EINVAL, "channel %d of %d over %lld pixels", channel, n_channels, (long long)n_pixels);
    if (n_channels == 1) { std::memcpy(dst, src, (size_t)n_pixels); return CBAS_OK; }
    int nt = n_threads < 1 ? 1 : (n_threads > 16 ? 16 : n_threads);
    if (n_pixels < (int64_t)nt * (1 << 18)) nt = (int)(n_pixels >> 18) + 1;        // >= 256 Ki pixels per thread
    if (nt <= 1) { pick_range(src, n_pixels, n_channels, channel, dst); return CBAS_OK; }
    std::vector<std::thread> th;
    th.reserve(nt - 1);
    const int64_t per = ((n_pixels + nt - 1) / nt + 63) & ~(int64_t)63;
    try {
        for (int t = 1; t < nt; ++t) {
            const int64_t a = t * per, b = a + per < n_pixels ? a + per : n_pixels;
            if (a >= b) break;
            th.emplace_back(pick_range, src + a * n_channels, b - a, n_channels, channel, dst + a);
        }
    } catch (...) {
        for (auto& x : th) x.join();
        return cbas_fail(CBAS_ENOMEM, "could not start a worker thread");
    }
    pick_range(src, per < n_pixels ? per : n_pixels, n_channels, channel, dst);
    for (auto& x : th) x.join();
    return CBAS_OK;
}

// Motion-JPEG frames -> green planes (or RGB) on host threads: the decode stage of the frame source.
//
// The reference decodes with decord.VideoReader(...).get_batch(...).asnumpy() (backend/cbas.py:402,425) and keeps channel 1
// of the RGB frame (:431).  For Motion-JPEG AVI files this file is that decoder: baseline / extended-sequential Huffman
// JPEG, 8 bit, grey or YCbCr with 4:4:4 / 4:2:2 / 4:2:0 sampling, restart intervals, Annex K tables when a frame carries
// no DHT ("AVI1" streams).  The arithmetic restates the published algorithms of the IJG / libjpeg-turbo decoder in its
// default configuration - the "accurate integer" inverse DCT (Loeffler-Ligtenberg-Moschytz, 13-bit constants), triangle
// ("fancy") chroma upsampling, 16-bit fixed-point YCbCr -> RGB - so the planes equal Pillow's bit for bit
// (tests/test_mjpeg_native.py).  Frames are independent: n_threads workers pull frame indices from one atomic counter and
// write straight into the caller's buffer (a page-locked ring piece in cbas_amd/pipeline.py).  Anything outside that
// envelope (progressive, 12 bit, CMYK, non-interleaved scans, other sampling factors) is refused with CBAS_EINVAL and the
// Python side falls back to Pillow for that file.
#include <atomic>
#include <cstdint>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "api_common.h"

namespace {

typedef int32_t v8i __attribute__((vector_size(32)));
typedef int16_t v8s __attribute__((vector_size(16)));
typedef uint8_t v8b __attribute__((vector_size(8)));

const uint8_t ZIGZAG[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                            41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                            30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// ITU T.81 Annex K.3 typical Huffman tables (what an MJPEG frame without DHT segments implies).
const uint8_t STD_DC_LUM_BITS[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
const uint8_t STD_DC_CHR_BITS[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
const uint8_t STD_DC_VALS[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const uint8_t STD_AC_LUM_BITS[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
const uint8_t STD_AC_LUM_VALS[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81,
    0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18,
    0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48,
    0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75,
    0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99,
    0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5,
    0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
const uint8_t STD_AC_CHR_BITS[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
const uint8_t STD_AC_CHR_VALS[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08,
    0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25,
    0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47,
    0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74,
    0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97,
    0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba,
    0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4,
    0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

constexpr int LUT_BITS = 9;

struct Huff {
    uint16_t lut[1 << LUT_BITS];   // (length << 8) | symbol for codes of <= LUT_BITS bits, 0 = longer
    int32_t maxcode[17];           // largest code of each length, -1 when there is none
    int32_t valoff[17];            // index of a length's first symbol minus its first code
    uint8_t vals[256];
    bool present = false;

    bool build(const uint8_t* bits, const uint8_t* v, int nv) {
        int total = 0;
        for (int i = 0; i < 16; ++i) total += bits[i];
        if (total > 256 || total > nv) return false;
        memcpy(vals, v, total);
        memset(lut, 0, sizeof(lut));
        int code = 0, k = 0;
        for (int l = 1; l <= 16; ++l) {
            valoff[l] = k - code;
            for (int i = 0; i < bits[l - 1]; ++i, ++code, ++k) {
                if (code >= (1 << l)) return false;
                if (l <= LUT_BITS) {
                    const int lo = code << (LUT_BITS - l), n = 1 << (LUT_BITS - l);
                    for (int j = 0; j < n; ++j) lut[lo + j] = (uint16_t)((l << 8) | vals[k]);
                }
            }
            maxcode[l] = bits[l - 1] ? code - 1 : -1;
            code <<= 1;
        }
        present = true;
        return true;
    }
};

struct BitReader {
    const uint8_t* p;
    const uint8_t* end;
    uint64_t buf = 0;
    int cnt = 0;           // valid bits at the bottom of buf
    bool marker = false;   // entropy data ended (a marker or the end of the frame): zeros are fed from here on
    int fake = 0;          // bytes of such zeros in buf

    inline void fill() {
        while (cnt <= 56) {
            uint32_t b = 0;
            if (!marker) {
                if (p >= end) marker = true;
                else if (*p != 0xFF) b = *p++;
                else if (p + 1 < end && p[1] == 0) { b = 0xFF; p += 2; }
                else marker = true;
            }
            fake += marker;
            buf = (buf << 8) | b;
            cnt += 8;
        }
    }
    inline uint32_t peek(int n) const { return (uint32_t)(buf >> (cnt - n)) & ((1u << n) - 1); }
    inline void skip(int n) { cnt -= n; }
    inline int decode(const Huff& h) {
        if (cnt < 32) fill();
        const uint16_t e = h.lut[peek(LUT_BITS)];
        if (e) { skip(e >> 8); return e & 0xFF; }
        for (int l = LUT_BITS + 1; l <= 16; ++l) {
            const int32_t c = (int32_t)peek(l);
            if (c <= h.maxcode[l]) { skip(l); return h.vals[(h.valoff[l] + c) & 0xFF]; }
        }
        return -1;
    }
    inline int receive_extend(int s) {
        if (cnt < 32) fill();
        const int r = (int)peek(s);
        skip(s);
        return r < (1 << (s - 1)) ? r - (1 << s) + 1 : r;
    }
    // true when the decoder has consumed bits that were never in the stream
    bool overrun() const { return fake * 8 > cnt; }
    // At a restart boundary: drop the partial byte and step over RSTn.
    bool restart() {
        if (overrun()) return false;
        buf = 0; cnt = 0; marker = false; fake = 0;
        while (p + 1 < end) {
            if (p[0] == 0xFF && p[1] >= 0xD0 && p[1] <= 0xD7) { p += 2; return true; }
            if (p[0] == 0xFF && p[1] != 0 && p[1] != 0xFF) return false;     // some other marker: the scan is over
            ++p;
        }
        return false;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int pred = 0;
    int stride = 0, rows = 0;          // padded plane
    int dw = 0, dh = 0;                // real (downsampled) size
    std::vector<uint8_t> plane;
};

struct Decoder {
    uint16_t quant[4][64];             // natural order
    bool quant_present[4] = {false, false, false, false};
    Huff dc[4], ac[4];
    Component comp[3];
    int ncomp = 0, width = 0, height = 0, restart_interval = 0;
    std::vector<int16_t> colsum_cb, colsum_cr;
    std::vector<uint8_t> up_cb, up_cr;
    std::string err;

    bool fail(const char* what) { err = what; return false; }
    void install_std_tables() {
        if (!dc[0].present) dc[0].build(STD_DC_LUM_BITS, STD_DC_VALS, 12);
        if (!dc[1].present) dc[1].build(STD_DC_CHR_BITS, STD_DC_VALS, 12);
        if (!ac[0].present) ac[0].build(STD_AC_LUM_BITS, STD_AC_LUM_VALS, 162);
        if (!ac[1].present) ac[1].build(STD_AC_CHR_BITS, STD_AC_CHR_VALS, 162);
    }
    bool decode(const uint8_t* data, size_t size, int want_h, int want_w, int channels, uint8_t* out);
    bool scan(const uint8_t* p, const uint8_t* end);
    void emit(int channels, uint8_t* out);
};

// ---- inverse DCT: jidctint's "islow" (8x8, CONST_BITS 13, PASS1_BITS 2), all eight columns / rows of a pass at once ----
constexpr int CB = 13, P1 = 2;
#define FIXC(x) ((int32_t)(x))
constexpr int32_t F_0_298 = 2446, F_0_390 = 3196, F_0_541 = 4433, F_0_765 = 6270, F_0_899 = 7373, F_1_175 = 9633,
                  F_1_501 = 12299, F_1_847 = 15137, F_1_961 = 16069, F_2_053 = 16819, F_2_562 = 20995, F_3_072 = 25172;

template <int SHIFT>
static inline __attribute__((always_inline)) void idct_1d(const v8i* in, v8i* out) {
    v8i z2 = in[2], z3 = in[6];
    v8i z1 = (z2 + z3) * F_0_541;
    v8i t2 = z1 - z3 * F_1_847;
    v8i t3 = z1 + z2 * F_0_765;
    z2 = in[0]; z3 = in[4];
    v8i t0 = (z2 + z3) << CB, t1 = (z2 - z3) << CB;
    const v8i t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
    t0 = in[7]; t1 = in[5]; t2 = in[3]; t3 = in[1];
    z1 = t0 + t3; z2 = t1 + t2; z3 = t0 + t2;
    v8i z4 = t1 + t3;
    const v8i z5 = (z3 + z4) * F_1_175;
    t0 *= F_0_298; t1 *= F_2_053; t2 *= F_3_072; t3 *= F_1_501;
    z1 *= -F_0_899; z2 *= -F_2_562; z3 *= -F_1_961; z4 *= -F_0_390;
    z3 += z5; z4 += z5;
    t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
    const int32_t rnd = 1 << (SHIFT - 1);
    out[0] = (t10 + t3 + rnd) >> SHIFT; out[7] = (t10 - t3 + rnd) >> SHIFT;
    out[1] = (t11 + t2 + rnd) >> SHIFT; out[6] = (t11 - t2 + rnd) >> SHIFT;
    out[2] = (t12 + t1 + rnd) >> SHIFT; out[5] = (t12 - t1 + rnd) >> SHIFT;
    out[3] = (t13 + t0 + rnd) >> SHIFT; out[4] = (t13 - t0 + rnd) >> SHIFT;
}

static inline __attribute__((always_inline)) void transpose8(v8i* r) {
    v8i a0 = __builtin_shufflevector(r[0], r[1], 0, 8, 1, 9, 4, 12, 5, 13);
    v8i a1 = __builtin_shufflevector(r[0], r[1], 2, 10, 3, 11, 6, 14, 7, 15);
    v8i a2 = __builtin_shufflevector(r[2], r[3], 0, 8, 1, 9, 4, 12, 5, 13);
    v8i a3 = __builtin_shufflevector(r[2], r[3], 2, 10, 3, 11, 6, 14, 7, 15);
    v8i a4 = __builtin_shufflevector(r[4], r[5], 0, 8, 1, 9, 4, 12, 5, 13);
    v8i a5 = __builtin_shufflevector(r[4], r[5], 2, 10, 3, 11, 6, 14, 7, 15);
    v8i a6 = __builtin_shufflevector(r[6], r[7], 0, 8, 1, 9, 4, 12, 5, 13);
    v8i a7 = __builtin_shufflevector(r[6], r[7], 2, 10, 3, 11, 6, 14, 7, 15);
    v8i b0 = __builtin_shufflevector(a0, a2, 0, 1, 8, 9, 4, 5, 12, 13);
    v8i b1 = __builtin_shufflevector(a0, a2, 2, 3, 10, 11, 6, 7, 14, 15);
    v8i b2 = __builtin_shufflevector(a1, a3, 0, 1, 8, 9, 4, 5, 12, 13);
    v8i b3 = __builtin_shufflevector(a1, a3, 2, 3, 10, 11, 6, 7, 14, 15);
    v8i b4 = __builtin_shufflevector(a4, a6, 0, 1, 8, 9, 4, 5, 12, 13);
    v8i b5 = __builtin_shufflevector(a4, a6, 2, 3, 10, 11, 6, 7, 14, 15);
    v8i b6 = __builtin_shufflevector(a5, a7, 0, 1, 8, 9, 4, 5, 12, 13);
    v8i b7 = __builtin_shufflevector(a5, a7, 2, 3, 10, 11, 6, 7, 14, 15);
    r[0] = __builtin_shufflevector(b0, b4, 0, 1, 2, 3, 8, 9, 10, 11);
    r[4] = __builtin_shufflevector(b0, b4, 4, 5, 6, 7, 12, 13, 14, 15);
    r[1] = __builtin_shufflevector(b1, b5, 0, 1, 2, 3, 8, 9, 10, 11);
    r[5] = __builtin_shufflevector(b1, b5, 4, 5, 6, 7, 12, 13, 14, 15);
    r[2] = __builtin_shufflevector(b2, b6, 0, 1, 2, 3, 8, 9, 10, 11);
    r[6] = __builtin_shufflevector(b2, b6, 4, 5, 6, 7, 12, 13, 14, 15);
    r[3] = __builtin_shufflevector(b3, b7, 0, 1, 2, 3, 8, 9, 10, 11);
    r[7] = __builtin_shufflevector(b3, b7, 4, 5, 6, 7, 12, 13, 14, 15);
}

// blk: quantised coefficients in natural order; q: the quantisation table; dst: 8 rows of 8 samples.
static void idct_block(const int16_t* blk, const uint16_t* q, uint8_t* dst, int stride) {
    v8i w[8], o[8];
    for (int r = 0; r < 8; ++r) {
        v8s c, qq;
        memcpy(&c, blk + 8 * r, 16);
        memcpy(&qq, q + 8 * r, 16);
        // quantiser values are <= 255 for 8-bit tables and <= 65535 otherwise: widen as unsigned
        typedef uint16_t v8us __attribute__((vector_size(16)));
        v8us qu;
        memcpy(&qu, &qq, 16);
        w[r] = __builtin_convertvector(c, v8i) * __builtin_convertvector(qu, v8i);
    }
    idct_1d<CB - P1>(w, o);          // pass 1: along the rows' index (columns of the block), lanes = column
    transpose8(o);                   // o[c] lanes = row
    idct_1d<CB + P1 + 3>(o, w);      // pass 2: along the columns' index, lanes = row
    transpose8(w);                   // w[r] lanes = column
    for (int r = 0; r < 8; ++r) {
        v8i x = w[r] + 128;
        x = x < 0 ? (v8i){0, 0, 0, 0, 0, 0, 0, 0} : x;
        x = x > 255 ? (v8i){255, 255, 255, 255, 255, 255, 255, 255} : x;
        const v8b b = __builtin_convertvector(x, v8b);
        memcpy(dst + r * stride, &b, 8);
    }
}

static inline void fill_block(uint8_t* dst, int stride, int dcq) {
    // a block with only its DC term: both passes reduce to DESCALE(dc << PASS1_BITS, PASS1_BITS + 3)
    int v = ((dcq * 4 + 16) >> 5) + 128;
    v = v < 0 ? 0 : v > 255 ? 255 : v;
    for (int r = 0; r < 8; ++r) memset(dst + r * stride, v, 8);
}

bool Decoder::scan(const uint8_t* p, const uint8_t* end) {
    BitReader br{p, end};
    const int hmax = comp[0].h, vmax = comp[0].v;
    const int mcux = (width + 8 * hmax - 1) / (8 * hmax), mcuy = (height + 8 * vmax - 1) / (8 * vmax);
    for (int c = 0; c < ncomp; ++c) {
        Component& k = comp[c];
        k.stride = mcux * 8 * k.h;
        k.rows = mcuy * 8 * k.v;
        k.dw = (width * k.h + hmax - 1) / hmax;
        k.dh = (height * k.v + vmax - 1) / vmax;
        if (k.plane.size() < (size_t)k.stride * k.rows) k.plane.resize((size_t)k.stride * k.rows);
        k.pred = 0;
        if (!dc[k.td].present || !ac[k.ta].present) return fail("a scan names a Huffman table the frame never defined");
        if (!quant_present[k.tq]) return fail("a component names a quantisation table the frame never defined");
    }
    alignas(32) int16_t blk[64];
    int until_restart = restart_interval;
    for (int my = 0; my < mcuy; ++my)
        for (int mx = 0; mx < mcux; ++mx) {
            if (restart_interval && until_restart == 0) {
                if (!br.restart()) return fail("restart marker missing or entropy data ends early");
                for (int c = 0; c < ncomp; ++c) comp[c].pred = 0;
                until_restart = restart_interval;
            }
            --until_restart;
            for (int c = 0; c < ncomp; ++c) {
                Component& k = comp[c];
                const Huff& hd = dc[k.td];
                const Huff& ha = ac[k.ta];
                for (int by = 0; by < k.v; ++by)
                    for (int bx = 0; bx < k.h; ++bx) {
                        int s = br.decode(hd);
                        if (s < 0 || s > 11) return fail("bad DC code");
                        if (s) k.pred += br.receive_extend(s);
                        if (k.pred > 32767 || k.pred < -32768) return fail("DC coefficient out of range");
                        int last = 0;
                        bool cleared = false;
                        for (int i = 1; i < 64;) {
                            const int rs = br.decode(ha);
                            if (rs < 0) return fail("bad AC code");
                            const int r = rs >> 4;
                            s = rs & 15;
                            if (s == 0) {
                                if (r != 15) break;
                                i += 16;
                                continue;
                            }
                            i += r;
                            if (i > 63) return fail("AC run past the end of a block");
                            if (!cleared) { memset(blk, 0, sizeof(blk)); cleared = true; }
                            blk[ZIGZAG[i]] = (int16_t)br.receive_extend(s);
                            last = i++;
                        }
                        uint8_t* dst = k.plane.data() + (size_t)(my * k.v + by) * 8 * k.stride + (mx * k.h + bx) * 8;
                        if (last == 0) {
                            fill_block(dst, k.stride, k.pred * (int)quant[k.tq][0]);
                        } else {
                            blk[0] = (int16_t)k.pred;
                            idct_block(blk, quant[k.tq], dst, k.stride);
                        }
                    }
            }
        }
    if (br.overrun()) return fail("entropy data ends early");
    return true;
}

// Planes -> output rows: triangle-filter chroma upsampling (h2v1 / h2v2 "fancy"), then the 16-bit fixed-point colour matrix.
void Decoder::emit(int channels, uint8_t* out) {
    const Component& Y = comp[0];
    if (ncomp == 1) {
        for (int y = 0; y < height; ++y) {
            const uint8_t* s = Y.plane.data() + (size_t)y * Y.stride;
            uint8_t* d = out + (size_t)y * width * channels;
            if (channels == 1) memcpy(d, s, width);
            else for (int x = 0; x < width; ++x) d[3 * x] = d[3 * x + 1] = d[3 * x + 2] = s[x];
        }
        return;
    }
    const Component& B = comp[1];
    const Component& R = comp[2];
    const int h2 = Y.h == 2, v2 = Y.v == 2;
    const int cw = B.dw, ch = B.dh;
    if ((int)colsum_cb.size() < cw + 2) { colsum_cb.resize(cw + 2); colsum_cr.resize(cw + 2); }
    if ((int)up_cb.size() < 2 * cw + 32) { up_cb.resize(2 * cw + 32); up_cr.resize(2 * cw + 32); }
    for (int y = 0; y < height; ++y) {
        const uint8_t* cbrow;
        const uint8_t* crrow;
        if (h2) {
            const int r = v2 ? y >> 1 : y;
            int far = r;
            if (v2) far = (y & 1) ? (r + 1 < ch ? r + 1 : ch - 1) : (r > 0 ? r - 1 : 0);
            for (int pl = 0; pl < 2; ++pl) {
                const Component& C = pl ? R : B;
                const uint8_t* near_row = C.plane.data() + (size_t)r * C.stride;
                const uint8_t* far_row = C.plane.data() + (size_t)far * C.stride;
                int16_t* cs = (pl ? colsum_cr : colsum_cb).data() + 1;
                uint8_t* up = (pl ? up_cr : up_cb).data();
                if (v2) {
                    for (int i = 0; i < cw; ++i) cs[i] = (int16_t)(3 * near_row[i] + far_row[i]);
                    cs[-1] = cs[0];
                    cs[cw] = cs[cw - 1];
                    for (int i = 0; i < cw; ++i) {
                        up[2 * i] = (uint8_t)((3 * cs[i] + cs[i - 1] + 8) >> 4);
                        up[2 * i + 1] = (uint8_t)((3 * cs[i] + cs[i + 1] + 7) >> 4);
                    }
                } else {
                    for (int i = 0; i < cw; ++i) cs[i] = near_row[i];
                    cs[-1] = cs[0];
                    cs[cw] = cs[cw - 1];
                    for (int i = 0; i < cw; ++i) {
                        up[2 * i] = (uint8_t)((3 * cs[i] + cs[i - 1] + 1) >> 2);
                        up[2 * i + 1] = (uint8_t)((3 * cs[i] + cs[i + 1] + 2) >> 2);
                    }
                    up[0] = near_row[0];                     // the two edge samples are copies, not filtered
                    up[2 * cw - 1] = near_row[cw - 1];
                }
            }
            cbrow = up_cb.data();
            crrow = up_cr.data();
        } else {
            cbrow = B.plane.data() + (size_t)y * B.stride;
            crrow = R.plane.data() + (size_t)y * R.stride;
        }
        const uint8_t* yrow = Y.plane.data() + (size_t)y * Y.stride;
        uint8_t* d = out + (size_t)y * width * channels;
        if (channels == 1) {
            for (int x = 0; x < width; ++x) {
                const int cb = cbrow[x] - 128, cr = crrow[x] - 128;
                int g = yrow[x] + ((-22554 * cb - 46802 * cr + 32768) >> 16);
                d[x] = (uint8_t)(g < 0 ? 0 : g > 255 ? 255 : g);
            }
        } else {
            for (int x = 0; x < width; ++x) {
                const int cb = cbrow[x] - 128, cr = crrow[x] - 128, yy = yrow[x];
                int r = yy + ((91881 * cr + 32768) >> 16);
                int g = yy + ((-22554 * cb - 46802 * cr + 32768) >> 16);
                int b = yy + ((116130 * cb + 32768) >> 16);
                d[3 * x] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
                d[3 * x + 1] = (uint8_t)(g < 0 ? 0 : g > 255 ? 255 : g);
                d[3 * x + 2] = (uint8_t)(b < 0 ? 0 : b > 255 ? 255 : b);
            }
        }
    }
}

bool Decoder::decode(const uint8_t* data, size_t size, int want_h, int want_w, int channels, uint8_t* out) {
    for (int i = 0; i < 4; ++i) { dc[i].present = ac[i].present = false; quant_present[i] = false; }
    restart_interval = 0;
    ncomp = 0;
    bool adobe_rgb = false, have_frame = false;
    const uint8_t* p = data;
    const uint8_t* end = data + size;
    if (size < 4 || p[0] != 0xFF || p[1] != 0xD8) return fail("no SOI marker");
    p += 2;
    while (p + 4 <= end) {
        if (*p != 0xFF) { ++p; continue; }
        while (p < end && *p == 0xFF) ++p;                // fill bytes
        if (p >= end) break;
        const int m = *p++;
        if (m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
        if (m == 0xD9) break;
        if (p + 2 > end) break;
        const int len = (p[0] << 8) | p[1];
        if (len < 2 || p + len > end) return fail("segment runs past the end of the frame");
        const uint8_t* s = p + 2;
        const uint8_t* se = p + len;
        if (m == 0xDB) {
            while (s < se) {
                const int pq = *s >> 4, tq = *s & 15;
                ++s;
                if (tq > 3 || pq > 1 || s + 64 * (pq + 1) > se) return fail("bad DQT segment");
                // 16-bit tables (Pq = 1) belong to 12-bit JPEG; with 8-bit samples libjpeg accepts them, but entries up to
                // 65 535 overflow the int32 arithmetic of fill_block / idct_block here (dc * quant * 4 + 16: undefined
                // behaviour, and the vector products wrap differently from libjpeg's): hand such frames to the fallback
                if (pq == 1) return fail("unsupported: 16-bit quantisation table");
                for (int i = 0; i < 64; ++i) {
                    quant[tq][ZIGZAG[i]] = pq ? (uint16_t)((s[0] << 8) | s[1]) : s[0];
                    s += pq + 1;
                }
                quant_present[tq] = true;
            }
        } else if (m == 0xC4) {
            while (s + 17 <= se) {
                const int tc = *s >> 4, th = *s & 15;
                int total = 0;
                for (int i = 0; i < 16; ++i) total += s[1 + i];
                if (tc > 1 || th > 3 || s + 17 + total > se) return fail("bad DHT segment");
                if (!(tc ? ac : dc)[th].build(s + 1, s + 17, total)) return fail("inconsistent Huffman table");
                s += 17 + total;
            }
        } else if (m == 0xC0 || m == 0xC1) {
            if (se - s < 6) return fail("bad SOF segment");
            if (s[0] != 8) return fail("unsupported: sample precision is not 8 bits");
            height = (s[1] << 8) | s[2];
            width = (s[3] << 8) | s[4];
            ncomp = s[5];
            if (ncomp != 1 && ncomp != 3) return fail("unsupported: neither grey nor three components");
            if (se - s < 6 + 3 * ncomp) return fail("bad SOF segment");
            for (int c = 0; c < ncomp; ++c) {
                comp[c].id = s[6 + 3 * c];
                comp[c].h = s[7 + 3 * c] >> 4;
                comp[c].v = s[7 + 3 * c] & 15;
                comp[c].tq = s[8 + 3 * c];
                if (comp[c].tq > 3) return fail("bad SOF segment");
            }
            if (ncomp == 1) comp[0].h = comp[0].v = 1;
            else {
                const bool ok = comp[1].h == 1 && comp[1].v == 1 && comp[2].h == 1 && comp[2].v == 1 &&
                                ((comp[0].h == 1 && comp[0].v == 1) || (comp[0].h == 2 && comp[0].v == 1) || (comp[0].h == 2 && comp[0].v == 2));
                if (!ok) return fail("unsupported: chroma sampling other than 4:4:4 / 4:2:2 / 4:2:0");
                if (comp[0].id == 'R' && comp[1].id == 'G' && comp[2].id == 'B') adobe_rgb = true;
            }
            have_frame = true;
        } else if (m == 0xC2 || (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {
            return fail("unsupported: not a baseline / extended-sequential Huffman JPEG");
        } else if (m == 0xDD) {
            if (se - s < 2) return fail("bad DRI segment");
            restart_interval = (s[0] << 8) | s[1];
        } else if (m == 0xEE) {
            if (se - s >= 12 && !memcmp(s, "Adobe", 5) && ncomp != 1 && s[11] == 0) adobe_rgb = true;
            if (se - s >= 12 && !memcmp(s, "Adobe", 5) && s[11] == 2) return fail("unsupported: YCCK");
        } else if (m == 0xDA) {
            if (!have_frame) return fail("SOS before SOF");
            if (adobe_rgb) return fail("unsupported: RGB-coded JPEG");
            if (width != want_w || height != want_h) {
                err = "frame is " + std::to_string(height) + "x" + std::to_string(width) + ", the stream header says " +
                      std::to_string(want_h) + "x" + std::to_string(want_w);
                return false;
            }
            if (se - s < 1 || s[0] != ncomp || se - s < 1 + 2 * ncomp + 3) return fail("unsupported: non-interleaved scan");
            for (int c = 0; c < ncomp; ++c) {
                if (s[1 + 2 * c] != comp[c].id) return fail("unsupported: scan components out of frame order");
                comp[c].td = s[2 + 2 * c] >> 4;
                comp[c].ta = s[2 + 2 * c] & 15;
                if (comp[c].td > 3 || comp[c].ta > 3) return fail("bad SOS segment");
            }
            const uint8_t* t = s + 1 + 2 * ncomp;
            if (t[0] != 0 || t[1] != 63 || t[2] != 0) return fail("unsupported: spectral selection / successive approximation");
            install_std_tables();
            if (!scan(se, end)) return false;
            emit(channels, out);
            return true;
        }
        p += len;
    }
    return fail("no scan in the frame");
}

}  // namespace

extern "C" int cbas_mjpeg_decode(const uint8_t* data, const uint64_t* offsets, const uint32_t* sizes, int32_t n_frames,
                                 int32_t height, int32_t width, int32_t channels, uint8_t* out, int32_t n_threads,
                                 int32_t* bad_frame) {
    if (bad_frame) *bad_frame = -1;
    if (n_frames < 0 || height <= 0 || width <= 0 || (channels != 1 && channels != 3))
        return cbas_fail(CBAS_EINVAL, "cbas_mjpeg_decode: bad shape");
    if (n_frames == 0) return CBAS_OK;
    if (!data || !offsets || !sizes || !out) return cbas_fail(CBAS_EINVAL, "cbas_mjpeg_decode: null argument");
    if (!__builtin_cpu_supports("avx2")) return cbas_fail(CBAS_EINVAL, "cbas_mjpeg_decode: unsupported: this build needs AVX2");
    const int nt = n_threads < 1 ? 1 : (n_threads > n_frames ? n_frames : n_threads);
    std::atomic<int> next{0};
    std::atomic<int> first_bad{n_frames};
    std::vector<std::string> errs(nt);
    std::vector<int> bad_of(nt, -1);
    const size_t frame_bytes = (size_t)height * width * channels;
    auto work = [&](int t) {
        std::unique_ptr<Decoder> d(new (std::nothrow) Decoder);
        if (!d) {
            errs[t] = "out of host memory";
            bad_of[t] = 0;
            first_bad.store(0);
            return;
        }
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n_frames || i > first_bad.load()) return;
            bool ok;
            try {
                ok = d->decode(data + offsets[i], sizes[i], height, width, channels, out + (size_t)i * frame_bytes);
            } catch (const std::bad_alloc&) {
                ok = false;
                d->err = "out of host memory";
            }
            if (!ok) {
                int cur = first_bad.load();
                while (i < cur && !first_bad.compare_exchange_weak(cur, i)) {}
                errs[t] = d->err;
                bad_of[t] = i;
                return;
            }
        }
    };
    if (nt == 1) work(0);
    else {
        std::vector<std::thread> th;
        try {
            for (int t = 1; t < nt; ++t) th.emplace_back(work, t);
        } catch (...) {}
        work(0);
        for (auto& x : th) x.join();
    }
    const int bad = first_bad.load();
    if (bad < n_frames) {
        if (bad_frame) *bad_frame = bad;
        std::string why;
        for (int t = 0; t < nt; ++t) if (bad_of[t] == bad) why = errs[t];
        return cbas_fail(CBAS_EINVAL, "cbas_mjpeg_decode: frame %d: %s", bad, why.c_str());
    }
    return CBAS_OK;
}

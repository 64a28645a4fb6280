// Host-side text emitter of the C ABI: `<video>_<model>_outputs.csv`.
//
// The reference writes it with pd.DataFrame(np.array(all_probs), columns=behaviors).to_csv(path, index=False)
// (backend/cbas.py:565).  For a float32 frame pandas turns every value into numpy's str(np.float32): the shortest
// decimal string that round-trips in float32, positional for 1e-4 <= |x| < 1e16 (compared in double), scientific
// with a sign and at least two exponent digits otherwise, "1.0" / "1e-05" trimming, NaN as the empty field.
// pandas takes 0.5 s per 18 000 x 9 clip for that; this does it in a few milliseconds, byte for byte
// (tests/test_host_properties.py, tests/test_csv_native.py).
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

#include "api_common.h"

namespace {

constexpr int kMaxField = 24;      // "-1.2345679e-38" is 14 characters; positional forms stay below 20

inline char* fmt_f32(float v, char* p) {
    if (v != v) return p;                                   // NaN -> na_rep ''
    if (std::signbit(v)) { *p++ = '-'; v = -v; }
    if (std::isinf(v)) { std::memcpy(p, "inf", 3); return p + 3; }
    if (v == 0.0f) { std::memcpy(p, "0.0", 3); return p + 3; }
    char tmp[32];
    const auto r = std::to_chars(tmp, tmp + sizeof(tmp), v, std::chars_format::scientific);   // d[.ddd]e[+-]XX, shortest
    const double a = (double)v;
    if (!(a >= 1e-4 && a < 1e16)) {                         // numpy: scientific, '.'-trimmed, >= 2 exponent digits
        const size_t n = (size_t)(r.ptr - tmp);
        std::memcpy(p, tmp, n);
        return p + n;
    }
    char dig[16];
    int nd = 0;
    const char* q = tmp;
    dig[nd++] = *q++;
    if (*q == '.') { ++q; while (*q != 'e') dig[nd++] = *q++; }
    ++q;                                                    // 'e'
    const bool neg = *q == '-';
    ++q;
    int e = 0;
    while (q < r.ptr) e = e * 10 + (*q++ - '0');
    if (neg) e = -e;
    if (e < 0) {                                            // 0.000ddd
        *p++ = '0'; *p++ = '.';
        for (int i = 0; i < -e - 1; ++i) *p++ = '0';
        std::memcpy(p, dig, (size_t)nd);
        return p + nd;
    }
    if (e >= nd - 1) {                                      // ddd000.0
        std::memcpy(p, dig, (size_t)nd); p += nd;
        for (int i = 0; i < e - (nd - 1); ++i) *p++ = '0';
        *p++ = '.'; *p++ = '0';
        return p;
    }
    std::memcpy(p, dig, (size_t)e + 1); p += e + 1;         // dd.ddd
    *p++ = '.';
    std::memcpy(p, dig + e + 1, (size_t)(nd - e - 1));
    return p + (nd - e - 1);
}

char* fmt_rows(const float* v, int64_t rows, int cols, char* p) {
    for (int64_t r = 0; r < rows; ++r) {
        for (int c = 0; c < cols; ++c) {
            if (c) *p++ = ',';
            p = fmt_f32(v[r * cols + c], p);
        }
        *p++ = '\n';
    }
    return p;
}

}  // namespace

extern "C" int64_t cbas_csv_format_f32(const float* values_host, int64_t n_rows, int32_t n_cols, char* out, int64_t cap) {
    if (n_rows < 0 || n_cols <= 0) return cbas_fail(CBAS_EINVAL, "cbas_csv_format_f32: bad shape");
    const int64_t need = n_rows * ((int64_t)n_cols * (kMaxField + 1) + 1);
    if (!out) return need;
    if (n_rows && !values_host) return cbas_fail(CBAS_EINVAL, "cbas_csv_format_f32: null values");
    if (cap < need) return cbas_fail(CBAS_EINVAL, "cbas_csv_format_f32: buffer of %lld bytes, %lld needed", (long long)cap, (long long)need);
    return fmt_rows(values_host, n_rows, n_cols, out) - out;
}

extern "C" int cbas_csv_write_f32(const char* path, const char* header_line, const float* values_host, int64_t n_rows,
                                  int32_t n_cols, int32_t n_threads) {
    if (!path || n_rows < 0 || n_cols <= 0 || (n_rows && !values_host)) return cbas_fail(CBAS_EINVAL, "cbas_csv_write_f32: bad argument");
    int nt = n_threads < 1 ? 1 : (n_threads > 64 ? 64 : n_threads);
    if (n_rows < 4096 * (int64_t)nt) nt = n_rows >= 8192 ? (int)(n_rows / 4096) : 1;
    const int64_t per = (n_rows + nt - 1) / nt;
    const int64_t row_cap = (int64_t)n_cols * (kMaxField + 1) + 1;
    std::vector<char*> bufs((size_t)nt, nullptr);
    std::vector<int64_t> lens((size_t)nt, 0);
    bool oom = false;
    for (int t = 0; t < nt; ++t) {
        const int64_t a = t * per, b = a + per < n_rows ? a + per : n_rows;
        if (b <= a) continue;
        bufs[t] = new (std::nothrow) char[(size_t)((b - a) * row_cap)];
        oom |= !bufs[t];
    }
    if (!oom) {
        auto work = [&](int t) {
            const int64_t a = t * per, b = a + per < n_rows ? a + per : n_rows;
            if (b > a) lens[t] = fmt_rows(values_host + a * n_cols, b - a, n_cols, bufs[t]) - bufs[t];
        };
        std::vector<std::thread> th;
        for (int t = 1; t < nt; ++t) th.emplace_back(work, t);
        work(0);
        for (auto& x : th) x.join();
    }
    int rc = CBAS_OK;
    if (oom) rc = cbas_fail(CBAS_ENOMEM, "cbas_csv_write_f32: out of host memory");
    if (!rc) {
        FILE* f = std::fopen(path, "wb");
        if (!f) rc = cbas_fail(CBAS_EINVAL, "cbas_csv_write_f32: cannot open %s", path);
        else {
            bool ok = true;
            if (header_line) ok = std::fputs(header_line, f) >= 0;
            for (int t = 0; t < nt && ok; ++t)
                if (lens[t]) ok = std::fwrite(bufs[t], 1, (size_t)lens[t], f) == (size_t)lens[t];
            ok = (std::fclose(f) == 0) && ok;
            if (!ok) rc = cbas_fail(CBAS_EINVAL, "cbas_csv_write_f32: write to %s failed", path);
        }
    }
    for (char* b : bufs) delete[] b;
    return rc;
}

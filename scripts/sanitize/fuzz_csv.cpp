#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <random>
#include <vector>
extern "C" int64_t cbas_csv_format_f32(const float*, int64_t, int32_t, char*, int64_t);
thread_local char g_cbas_err[512];
int main() {
    std::mt19937 rng(3);
    long total = 0;
    for (int it = 0; it < 20000; ++it) {
        int rows = 1 + rng() % 40, cols = 1 + rng() % 12;
        float* v = (float*)malloc(sizeof(float) * rows * cols);
        for (int i = 0; i < rows * cols; ++i) { uint32_t b = (it % 2) ? rng() : (0x3f800000u - (rng() % 0x10000000u)); memcpy(&v[i], &b, 4); }
        int64_t cap = cbas_csv_format_f32(nullptr, rows, cols, nullptr, 0);
        char* out = (char*)malloc(cap);
        int64_t n = cbas_csv_format_f32(v, rows, cols, out, cap);
        if (n < 0 || n > cap) { printf("bad n %lld cap %lld\n", (long long)n, (long long)cap); return 1; }
        // exact-size second pass: the formatter must accept a buffer of exactly n bytes? (it requires cap >= bound) - only check the bound holds
        total += n; free(out); free(v);
    }
    printf("ok %ld bytes\n", total);
}

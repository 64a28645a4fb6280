#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <vector>
#include <random>
extern "C" int cbas_mjpeg_decode(const uint8_t*, const uint64_t*, const uint32_t*, int32_t, int32_t, int32_t, int32_t, uint8_t*, int32_t, int32_t*);
thread_local char g_cbas_err[512];
int main(int argc, char** argv) {
    std::vector<std::vector<uint8_t>> base;
    for (int i = 1; i < argc; ++i) { FILE* f = fopen(argv[i], "rb"); std::vector<uint8_t> b(1 << 20); size_t n = fread(b.data(), 1, b.size(), f); fclose(f); b.resize(n); base.push_back(b); }
    std::mt19937 rng(1);
    std::vector<uint8_t> out(48 * 64 * 3);
    long ok = 0, err = 0;
    for (int it = 0; it < 60000; ++it) {
        std::vector<uint8_t> s = base[it % base.size()];
        int k = 1 + rng() % 7;
        for (int j = 0; j < k; ++j) {
            size_t pos = 2 + rng() % (s.size() - 2);
            int mode = it % 3;
            if (mode == 0) s[pos] = rng() & 255; else if (mode == 1) s[pos] ^= 1 << (rng() & 7); else if (pos + 4 < s.size()) s.erase(s.begin() + pos, s.begin() + pos + 1 + rng() % 3);
        }
        // exact-size heap copy: ASan sees any read past the end
        uint8_t* d = (uint8_t*)malloc(s.size()); memcpy(d, s.data(), s.size());
        uint64_t off = 0; uint32_t sz = (uint32_t)s.size(); int32_t bad;
        int rc = cbas_mjpeg_decode(d, &off, &sz, 1, 48, 64, 1 + 2 * (it & 1), out.data(), 1, &bad);
        free(d);
        (rc == 0 ? ok : err)++;
    }
    printf("ok %ld err %ld\n", ok, err);
    return 0;
}

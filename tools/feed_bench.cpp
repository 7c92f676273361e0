// feed_bench.cpp -- host range-coder pack pair fed from 16-bit rows and from 13-bit groups (no GPU).
//   g++ -O2 -std=c++17 tools/feed_bench.cpp nblic-image-compression_amd/csrc/build/range_coder_x8.o -o /tmp/feed_bench
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "../nblic-image-compression_amd/csrc/range_coder.h"
using namespace nblic;
int main(int argc, char **argv) {
    const size_t n = argc > 1 ? size_t(atol(argv[1])) : size_t(1) << 22;      // bins per lane
    std::mt19937 rng(1);
    std::vector<std::vector<uint16_t>> s(24, std::vector<uint16_t>(n));
    for (auto &v : s) for (auto &e : v) { const uint32_t p = 1 + rng() % 4095; e = uint16_t(p | ((rng() % 4096 < p) ? 0x8000u : 0u)); }
    if (argc > 2) {                                          // real records: <prefix>K.u16 for K = 0.. (as many as exist, reused in turn), n of each
        int have = 0;
        for (int l = 0; l < 24; l++) {
            char path[512];
            snprintf(path, sizeof path, "%s%d.u16", argv[2], l);
            FILE *f = fopen(path, "rb");
            if (f) { if (fread(s[l].data(), 2, n, f) != n) { fprintf(stderr, "%s is shorter than %zu records\n", path, n); return 1; } fclose(f); have = l + 1; }
            else if (have) s[l] = s[l % have];
            else { fprintf(stderr, "no %s\n", path); return 1; }
        }
        printf("records from %s* (%d files)\n", argv[2], have);
    }
    uint64_t *r16 = (uint64_t *)aligned_alloc(64, n / 4 * 16 * 8), *r13 = (uint64_t *)aligned_alloc(64, group_words(n) * 8);
    memset(r16, 0, n / 4 * 16 * 8); memset(r13, 0, group_words(n) * 8);
    size_t len[16];
    for (int l = 0; l < 16; l++) {
        len[l] = n;
        for (size_t i = 0; i < n; i++) r16[16 * (i >> 2) + l] |= uint64_t(s[l][i]) << (16 * (i & 3));
        pack_groups_host(r13, l, s[l].data(), n);
    }
    uint64_t *r24 = (uint64_t *)aligned_alloc(64, group_words(n, 24) * 8);
    memset(r24, 0, group_words(n, 24) * 8);
    size_t len24[24];
    for (int l = 0; l < 24; l++) { len24[l] = n; pack_groups_host(r24, l, s[l].data(), n, 24); }
    std::vector<std::vector<uint8_t>> out(24, std::vector<uint8_t>(2 * n + 64));
    uint8_t *outs[24]; size_t caps[24], la[24], lb[24], lc[24];
    for (int l = 0; l < 24; l++) { outs[l] = out[l].data(); caps[l] = out[l].size(); }
    for (int rep = 0; rep < 3; rep++) {
        for (int form = 0; form < 2; form++) {
            RangeX8 a, b;
            a.begin(8, outs, caps); b.begin(8, outs + 8, caps + 8);
            auto t0 = std::chrono::steady_clock::now();
            if (form == 0) feed_pair_rows(a, b, r16, len); else feed_pair_groups(a, b, r13, len);
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            a.end(form ? lb : la); b.end((form ? lb : la) + 8);
            printf("%s: %.0f Mbins/s\n", form ? "13-bit groups" : "16-bit rows  ", 16.0 * n / dt / 1e6);
        }
        {
            RangeX8 a, b, c;
            a.begin(8, outs, caps); b.begin(8, outs + 8, caps + 8); c.begin(8, outs + 16, caps + 16);
            auto t0 = std::chrono::steady_clock::now();
            feed_triple_groups(a, b, c, r24, len24);
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            a.end(lc); b.end(lc + 8); c.end(lc + 16);
            printf("13-bit groups, three packs: %.0f Mbins/s\n", 24.0 * n / dt / 1e6);
        }
        printf("same lengths: %d %d\n", !memcmp(la, lb, 16 * sizeof(size_t)), !memcmp(la, lc, 16 * sizeof(size_t)));
    }
    return 0;
}

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
    std::vector<std::vector<uint16_t>> s(16, std::vector<uint16_t>(n));
    for (auto &v : s) for (auto &e : v) { const uint32_t p = 1 + rng() % 4095; e = uint16_t(p | ((rng() % 4096 < p) ? 0x8000u : 0u)); }
    uint64_t *r16 = (uint64_t *)aligned_alloc(64, n / 4 * 16 * 8), *r13 = (uint64_t *)aligned_alloc(64, group_words(n) * 8);
    memset(r16, 0, n / 4 * 16 * 8); memset(r13, 0, group_words(n) * 8);
    size_t len[16];
    for (int l = 0; l < 16; l++) {
        len[l] = n;
        for (size_t i = 0; i < n; i++) r16[16 * (i >> 2) + l] |= uint64_t(s[l][i]) << (16 * (i & 3));
        pack_groups_host(r13, l, s[l].data(), n);
    }
    std::vector<std::vector<uint8_t>> out(16, std::vector<uint8_t>(2 * n + 64));
    uint8_t *outs[16]; size_t caps[16], la[16], lb[16];
    for (int l = 0; l < 16; l++) { outs[l] = out[l].data(); caps[l] = out[l].size(); }
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
        printf("same lengths: %d\n", !memcmp(la, lb, sizeof la));
    }
    return 0;
}

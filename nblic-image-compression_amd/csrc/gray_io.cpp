// gray_io.cpp -- PGM (P5) and 8-bit gray BMP files, byte-compatible with what the reference's
// front end accepts and produces (src/FileIO.c:81-287).  Whole-file buffers and explicit little-endian
// field access instead of the reference's byte-at-a-time stdio calls; same accept / reject rules.
#include "gray_io.h"

#include <cstdio>
#include <cstring>

namespace nblic {

namespace {

struct File {
    FILE *f;
    explicit File(const std::string &path, const char *mode) : f(fopen(path.c_str(), mode)) {}
    ~File() { if (f) fclose(f); }
    explicit operator bool() const { return f != nullptr; }
};

constexpr int kMaxSide = 65535;                       // NBLIC.h:29-30
constexpr size_t kBmpHeader = 14 + 40 + 1024;         // file header + BITMAPINFOHEADER + palette = 0x436

uint32_t le(const uint8_t *p, int n) { uint32_t v = 0; for (int k = n - 1; k >= 0; k--) v = (v << 8) | p[k]; return v; }
void put_le(uint8_t *p, uint32_t v, int n) { for (int k = 0; k < n; k++) { p[k] = uint8_t(v); v >>= 8; } }

// what fscanf(" %d") accepts: white space, an optional sign, decimal digits
bool scan_int(const std::vector<uint8_t> &b, size_t &at, long &out) {
    while (at < b.size() && (b[at] == ' ' || (b[at] >= '\t' && b[at] <= '\r'))) at++;
    bool neg = false;
    if (at < b.size() && (b[at] == '+' || b[at] == '-')) { neg = b[at] == '-'; at++; }
    if (at >= b.size() || b[at] < '0' || b[at] > '9') return false;
    long v = 0;
    while (at < b.size() && b[at] >= '0' && b[at] <= '9') { v = v * 10 + (b[at] - '0'); if (v > (1L << 40)) return false; at++; }
    out = neg ? -v : v;
    return true;
}

}  // namespace

bool read_file(const std::string &path, std::vector<uint8_t> &bytes, size_t limit) {
    File fp(path, "rb");
    if (!fp) return false;
    if (fseek(fp.f, 0, SEEK_END) != 0) return false;
    const long n = ftell(fp.f);
    if (n < 0 || size_t(n) > limit) return false;
    rewind(fp.f);
    bytes.resize(size_t(n));
    return n == 0 || fread(bytes.data(), 1, size_t(n), fp.f) == size_t(n);
}

bool write_file(const std::string &path, const uint8_t *bytes, size_t n) {
    File fp(path, "wb");
    return fp && fwrite(bytes, 1, n, fp.f) == n && fflush(fp.f) == 0;
}

bool read_pgm(const std::string &path, GrayImage &img) {
    std::vector<uint8_t> b;
    if (!read_file(path, b, size_t(kMaxSide) * kMaxSide + 4096) || b.size() < 2 || b[0] != 'P' || b[1] != '5') return false;
    size_t at = 2;
    long w = 0, h = 0, maxval = 0;
    if (!scan_int(b, at, w) || !scan_int(b, at, h) || !scan_int(b, at, maxval)) return false;
    if (maxval < 1 || maxval > 255 || w < 1 || h < 1 || w > kMaxSide || h > kMaxSide) return false;
    at++;                                              // exactly one separator byte after maxval
    const size_t n = size_t(w) * size_t(h);
    if (at > b.size() || b.size() - at < n) return false;
    img.h = int(h); img.w = int(w);
    img.px.assign(b.begin() + long(at), b.begin() + long(at + n));
    return true;
}

bool read_bmp8(const std::string &path, GrayImage &img) {
    std::vector<uint8_t> b;
    if (!read_file(path, b, size_t(kMaxSide) * (kMaxSide + 3) + (1u << 20)) || b.size() < 34) return false;
    const uint32_t magic = le(&b[0], 2), offset = le(&b[10], 4);
    const int32_t w = int32_t(le(&b[18], 4)), h = int32_t(le(&b[22], 4));
    const uint32_t planes = le(&b[26], 2), bpp = le(&b[28], 2), compression = le(&b[30], 4);
    if (magic != 0x4D42 || planes != 1 || bpp != 8 || compression != 0 || w < 1 || h < 1 || w > kMaxSide || h > kMaxSide) return false;
    if (offset < 34 || offset > b.size()) return false;
    const size_t stride = (size_t(w) + 3) & ~size_t(3);
    // every row must be complete; the padding after the LAST stored row (the top one) may be missing
    if (b.size() - offset < stride * size_t(h - 1) + size_t(w)) return false;
    img.h = h; img.w = w;
    img.px.resize(size_t(w) * size_t(h));
    for (int i = 0; i < h; i++)                        // stored bottom-up
        memcpy(&img.px[size_t(h - 1 - i) * size_t(w)], &b[offset + stride * size_t(i)], size_t(w));
    return true;
}

GrayFormat read_gray(const std::string &path, GrayImage &img) {
    if (read_pgm(path, img)) return GrayFormat::kPgm;
    if (read_bmp8(path, img)) return GrayFormat::kBmp;
    return GrayFormat::kNone;
}

bool write_pgm(const std::string &path, const uint8_t *px, int h, int w) {
    if (w < 1 || h < 1) return false;
    File fp(path, "wb");
    if (!fp) return false;
    fprintf(fp.f, "P5\n%d %d\n255\n", w, h);
    const size_t n = size_t(w) * size_t(h);
    return fwrite(px, 1, n, fp.f) == n && fflush(fp.f) == 0;
}

bool write_bmp8(const std::string &path, const uint8_t *px, int h, int w) {
    if (w < 1 || h < 1) return false;
    const size_t stride = (size_t(w) + 3) & ~size_t(3);
    std::vector<uint8_t> out(kBmpHeader + stride * size_t(h), 0);
    uint8_t *p = out.data();
    put_le(p + 0, 0x4D42, 2);                          // "BM"
    put_le(p + 2, uint32_t(out.size()), 4);
    put_le(p + 10, uint32_t(kBmpHeader), 4);           // 0x436: where the pixels start
    put_le(p + 14, 40, 4);
    put_le(p + 18, uint32_t(w), 4);
    put_le(p + 22, uint32_t(h), 4);
    put_le(p + 26, 1, 2);
    put_le(p + 28, 8, 2);
    put_le(p + 38, 0xEC4, 4);                          // 3780 px/m both ways
    put_le(p + 42, 0xEC4, 4);
    put_le(p + 46, 256, 4);                            // palette entries
    for (int k = 0; k < 256; k++) { uint8_t *e = p + 54 + 4 * k; e[0] = e[1] = e[2] = uint8_t(k); e[3] = 0xFF; }
    for (int i = 0; i < h; i++) memcpy(p + kBmpHeader + stride * size_t(i), px + size_t(h - 1 - i) * size_t(w), size_t(w));
    return write_file(path, out.data(), out.size());
}

}  // namespace nblic

// q_entropy.cpp -- host half of the QNBLIC (effort 0) path: everything after the per-pixel model.
//
// The GPU hands over, per image, one (level, symbol) pair per pixel and the twelve 256-bin symbol
// histograms.  What is left (QNBLIC.c:625-655) is small or serial: scale every histogram to a sum
// of 2^15, write them in the reference's compact 16-bit code, then push the symbols through one
// 32-bit rANS coder LAST PIXEL FIRST and reverse the emitted words.  The decoder side builds the
// per-level slot tables for the device engine.
//
// The histogram scaling is the only floating point on any NBLIC path and it decides bits of the
// stream, so this file is compiled by the host compiler with contraction off (csrc/Makefile):
// IEEE double, multiply then add, truncating conversion -- what gcc -O3 does for the reference.
#include <stddef.h>
#include <stdint.h>
#include <string.h>

namespace nblic {

constexpr int kQLevels = 12, kQSyms = 256, kQNormBits = 15, kQAnsBits = 16;
constexpr uint32_t kQNormSum = 1u << kQNormBits;

// QNBLIC.c:308-358
static void q_normalise(uint32_t *h) {
    uint32_t total = 0, used = 0, last = 0;
    for (uint32_t s = 0; s < kQSyms; s++) if (h[s]) { total += h[s]; used++; last = s; }
    if (used == 0) { h[0] = kQNormSum - 1; h[1] = 1; return; }
    if (used == 1) { h[last] = kQNormSum - 1; h[(last + 1) & 255] = 1; return; }
    const double scale = (1.0 * kQNormSum) / total;
    total = 0;
    for (uint32_t s = 0; s < kQSyms; s++) {
        if (!h[s]) continue;
        uint32_t v = (uint32_t)(0.49 + scale * h[s]);
        h[s] = v ? v : 1;
        total += h[s];
    }
    for (uint32_t s = 0; total > kQNormSum; s = (s + 1) & 255) if (h[s] > 1) { h[s]--; total--; }
    for (uint32_t s = 0; total < kQNormSum; s = (s + 1) & 255) if (h[s] > 0) { h[s]++; total++; }
}

// QNBLIC.c:362-459: five code shapes -- 15-bit value, 2 x 7 bit, 3 x 4 bit, 4 x 3 bit, run of 0/1 (+ one nibble)
static uint16_t *q_write_hist(uint16_t *p, uint16_t *end, const uint32_t *h) {
    uint32_t i = 0, covered = 0;
    while (i < kQSyms && covered < kQNormSum) {
        if (p >= end) return nullptr;
        const uint32_t first = h[i] & 0xFFFF;
        uint32_t j = i + 1, next = 0xFFFF;
        while (j < kQSyms) { next = h[j] & 0xFFFF; if (next != first) break; j++; }
        const uint32_t run = j - i;
        uint32_t code;
        if (first <= 1 && run >= 4) {
            if (j < kQSyms && next <= 15) j++; else next = first;
            code = 0xE000u | (first << 12) | (next << 8) | (run - 4);
        } else {
            auto at = [&](uint32_t k) { return k < kQSyms ? (h[k] & 0xFFFF) : 0xFFFFu; };
            const uint32_t b = at(i + 1), c = at(i + 2), d = at(i + 3);
            if (first <= 7 && b <= 7 && c <= 7 && d <= 7)   { code = 0xD000u | (first << 9) | (b << 6) | (c << 3) | d; j = i + 4; }
            else if (first <= 15 && b <= 15 && c <= 15)     { code = 0xC000u | (first << 8) | (b << 4) | c;             j = i + 3; }
            else if (first <= 127 && b <= 127)              { code = 0x8000u | (first << 7) | b;                        j = i + 2; }
            else                                            { code = first;                                             j = i + 1; }
        }
        *p++ = (uint16_t)code;
        for (; i < j; i++) covered += h[i];
    }
    return p;
}

static const uint16_t *q_read_hist(const uint16_t *p, const uint16_t *end, uint32_t *h) {
    memset(h, 0, sizeof(uint32_t) * kQSyms);
    uint32_t i = 0, covered = 0;
    auto put = [&](uint32_t v) { if (i < kQSyms) { h[i++] = v; covered += v; } };
    while (i < kQSyms && covered < kQNormSum) {
        if (p >= end) return nullptr;
        const uint32_t code = *p++;
        if (!(code & 0x8000u))              put(code);
        else if ((code >> 14) == 2)       { put((code >> 7) & 0x7F); put(code & 0x7F); }
        else if ((code >> 12) == 12)      { put((code >> 8) & 15); put((code >> 4) & 15); put(code & 15); }
        else if ((code >> 12) == 13)      { put((code >> 9) & 7); put((code >> 6) & 7); put((code >> 3) & 7); put(code & 7); }
        else {
            const uint32_t v = (code >> 12) & 1, tail = (code >> 8) & 15;
            for (uint32_t r = (code & 0xFF) + 4; r > 0; r--) put(v);
            if (tail != v) put(tail);
        }
    }
    return p;
}

// Entropy stage of QNBLICcompress.  qy[t] = level | symbol << 8; hist = 12 x 256 raw counts.
// Returns the stream length in 16-bit words (header included) or -1 if cap_words is too small.
long q_entropy_encode(uint16_t *out, size_t cap_words, int h, int w, const uint16_t *qy, const uint32_t *hist_in) {
    uint32_t freq[kQLevels][kQSyms], start[kQLevels][kQSyms];
    const size_t n = (size_t)h * (size_t)w;
    if (cap_words < 6) return -1;
    uint16_t *p = out, *const end = out + cap_words;
    *p++ = (uint16_t)('Q' | ('0' << 8)); *p++ = (uint16_t)('.' | ('2' << 8));            // QNBLIC.c:463-473, host byte order
    *p++ = (uint16_t)h; *p++ = (uint16_t)w;
    memcpy(freq, hist_in, sizeof freq);
    for (int k = 0; k < kQLevels; k++) {
        q_normalise(freq[k]);
        uint32_t acc = 0;
        for (int s = 0; s < kQSyms; s++) { start[k][s] = acc; acc += freq[k][s]; }
        p = q_write_hist(p, end, freq[k]);
        if (!p) return -1;
    }
    uint16_t *const body = p;
    uint32_t x = 1u << kQAnsBits;                                                       // QNBLIC.c:238-253
    for (size_t t = n; t-- > 0;) {
        const uint32_t e = qy[t], f = freq[e & 0xFF][e >> 8], s0 = start[e & 0xFF][e >> 8];
        uint32_t q = x / f;
        if (q > (1u << (2 * kQAnsBits - kQNormBits)) - 1) {
            if (p >= end) return -1;
            *p++ = (uint16_t)x; x >>= kQAnsBits; q = x / f;
        }
        x = (x - q * f) + (q << kQNormBits) + s0;
    }
    if (p + 2 > end) return -1;
    *p++ = (uint16_t)x; *p++ = (uint16_t)(x >> kQAnsBits);
    for (uint16_t *a = body, *b = p - 1; a < b; a++, b--) { uint16_t t = *a; *a = *b; *b = t; }   // decoder reads forwards
    return (long)(p - out);
}

// Decoder front matter (QNBLIC.c:505-518): parses the header and the twelve histograms, builds
// freq / start (12 x 256 each) and, when `slot` is given, the slot -> symbol tables (12 x 32768 bytes).  Returns the word index where the rANS payload starts, or -1.
long q_decode_tables(const uint16_t *in, size_t n_words, int *h, int *w, uint32_t *freq, uint32_t *start, uint8_t *slot) {
    if (n_words < 6 || in[0] != (uint16_t)('Q' | ('0' << 8)) || in[1] != (uint16_t)('.' | ('2' << 8))) return -1;
    *h = in[2]; *w = in[3];
    const uint16_t *p = in + 4, *const end = in + n_words;
    for (int k = 0; k < kQLevels; k++) {
        uint32_t *f = freq + k * kQSyms, *s0 = start + k * kQSyms;
        p = q_read_hist(p, end, f);
        if (!p) return -1;
        uint32_t acc = 0;
        for (int s = 0; s < kQSyms; s++) { s0[s] = acc; acc += f[s]; }
        if (!slot) continue;                 // the device engine searches the cumulative table itself
        uint8_t *tab = slot + (size_t)k * kQNormSum;
        for (uint32_t s = 0; s + 1 < kQSyms; s++)
            for (uint32_t i = s0[s]; i < s0[s + 1] && i < kQNormSum; i++) tab[i] = (uint8_t)s;
        for (uint32_t i = s0[kQSyms - 1]; i < kQNormSum; i++) tab[i] = (uint8_t)(kQSyms - 1);
    }
    return (long)(p - in);
}

}  // namespace nblic

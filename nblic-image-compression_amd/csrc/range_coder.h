// range_coder.h -- host side of stage S6: the 32-bit carry-less binary range coder (NBLIC.c:527-586)
// in resumable form.  The coder threads stream an image's bins from HBM chunk by chunk, so both
// coders keep their state between calls: begin() once, feed() per chunk, end() once.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace nblic {

// One stream.  coded[r] = prob (12 bit, P(bin==1)) | bin << 15.  Bin 1 takes the lower part of [lo, hi].
struct RangeScalar {
    uint32_t lo = 0, hi = 0xFFFFFFFFu;
    uint8_t *out = nullptr, *p = nullptr, *end = nullptr;
    bool overflow = false;
    void begin(uint8_t *out_, size_t cap) {
        lo = 0; hi = 0xFFFFFFFFu; out = p = out_; overflow = cap < 4;
        end = out_ + (cap < 4 ? 0 : cap - 4);                  // keep room for the flush
    }
    void feed(const uint16_t *coded, size_t n);
    size_t finish();                                           // bytes written, or SIZE_MAX if the output did not fit
};

// Eight streams in the eight 64-bit lanes of one AVX-512 register (range_coder_x8.cpp).
struct RangeX8 {
    struct State;
    State *st;
    RangeX8();
    ~RangeX8();
    RangeX8(const RangeX8 &) = delete;
    RangeX8 &operator=(const RangeX8 &) = delete;
    void begin(int count, uint8_t *const *outs, const size_t *caps);
    void feed(const uint16_t *const *src, const size_t *len);  // lane k codes src[k][0 .. len[k]); len 0 = idle
    void end(size_t *lens);                                    // per lane: bytes written or SIZE_MAX
};

// two packs (sixteen streams) advanced in lock-step by one thread
void feed_pair(RangeX8 &a, const uint16_t *const *src_a, const size_t *len_a, RangeX8 &b, const uint16_t *const *src_b, const size_t *len_b);

// the same from row-interleaved bins: rows[16 * i + lane] = bins 4i..4i+3 of lane `lane` as one 64-bit word
// (64-byte aligned; lanes 0-7 pack a, 8-15 pack b); len[lane] = the lane's bins in this chunk
void feed_pair_rows(RangeX8 &a, RangeX8 &b, const uint64_t *rows, const size_t *len);

// the form in which a pack pair's bins cross PCIe: 64 records of a lane as 64 x 13 bits = thirteen 64-bit words,
// rows[(13 * g + j) * 16 + lane] = word j of the lane's group g (range_coder_x8.cpp, k_pack_groups in pipeline.hip)
constexpr size_t kGroupBins = 64, kGroupWords = 13;
constexpr uint32_t code13(uint32_t rec) { return (rec & 0xFFFu) | ((rec >> 3) & 0x1000u); }           // prob | bin << 12
constexpr size_t group_words(size_t bins, size_t lanes = 16) { return (bins + kGroupBins - 1) / kGroupBins * kGroupWords * lanes; }   // words of `bins` bins x `lanes` lanes
void feed_pair_groups(RangeX8 &a, RangeX8 &b, const uint64_t *rows, const size_t *len);                    // 16 lanes
void feed_triple_groups(RangeX8 &a, RangeX8 &b, RangeX8 &c, const uint64_t *rows, const size_t *len);      // 24 lanes: rows[(13 g + j) * 24 + lane]
void pack_groups_host(uint64_t *rows, int lane, const uint16_t *coded, size_t len, int lanes = 16);   // ORs one lane's records into zeroed rows (tests, the chunked self-check)

bool have_avx512();
size_t range_code(const uint16_t *coded, size_t n, uint8_t *out, size_t cap);
void range_code_x8(const uint16_t *const *coded, const size_t *n, int count, uint8_t *const *outs, const size_t *caps, size_t *lens);

}  // namespace nblic

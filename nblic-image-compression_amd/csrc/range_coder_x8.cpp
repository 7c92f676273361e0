// range_coder_x8.cpp -- the serial range-coder stage (NBLIC.c:527-586) for EIGHT images at once.
//
// One bin of one stream is a ~9-cycle dependent chain (subtract, 32x32->64 multiply, shift, add,
// select) that no core can shorten; what a core can do is run eight independent chains in the
// eight 64-bit lanes of one AVX-512 register.  Every lane is one image's coder: interval
// [lo, hi] in the low 32 bits of its lane, its own output pointer, its own bin count.  Byte
// emission is data dependent per lane, so emitted bytes are collected in a per-lane 64-bit
// accumulator and flushed eight at a time with a scalar store.
//
// Bit-exactness: per lane this is instruction for instruction the scalar coder in pipeline.hip
// (same floor((hi-lo)*prob/4096), same renormalisation loop, same 4-byte flush).
#include <immintrin.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <new>

#include "range_coder.h"

namespace nblic {

bool have_avx512() {
    return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512dq") &&
           __builtin_cpu_supports("avx512vl") && __builtin_cpu_supports("avx512vbmi2");
}

#define NB_TARGET __attribute__((target("avx512f,avx512bw,avx512dq,avx512vl,avx512vbmi2,bmi,bmi2,lzcnt")))

namespace {

#define NB_INLINE inline __attribute__((always_inline))
constexpr size_t kPrefetchAhead = 1024;                       // bytes; sixteen gathered streams defeat the hardware prefetcher

// Per lane the interval is kept as (lo, span = hi - lo): the multiply then depends on one value.
// The four vectors live in REGISTERS for the whole of a feed() call: they are copied out of the
// persistent state at its start and back at its end, and every helper below is force-inlined --
// passed around by reference to memory, each step's interval update went through a 64-byte store
// and reload, and that round trip, not the arithmetic, set the pace.
struct Regs { __m512i lo, span, acc, cnt; };
struct Outs { uint8_t *outp[8], *oend[8]; bool overflow[8]; };
struct Lanes { Regs r; Outs o; };

NB_TARGET NB_INLINE void flush_full(Regs &R, Outs &O, __mmask8 kf) {
    alignas(64) uint64_t a[8];
    _mm512_store_si512((void *)a, R.acc);
    unsigned m = kf;
    while (m) {
        int k = __builtin_ctz(m);
        m &= m - 1;
        if (O.outp[k] + 8 <= O.oend[k]) {
            uint64_t be = __builtin_bswap64(a[k]);
            __builtin_memcpy(O.outp[k], &be, 8);
        } else {
            O.overflow[k] = true;
        }
        O.outp[k] += 8;
    }
    R.cnt = _mm512_mask_mov_epi64(R.cnt, kf, _mm512_setzero_si512());
}

// lanes in k shift one byte out (NBLIC.c:563-572); returns the lanes that must shift again
NB_TARGET NB_INLINE __mmask8 renorm_once(Regs &R, Outs &O, __mmask8 k) {
    const __m512i m32 = _mm512_set1_epi64(0xFFFFFFFFll), top = _mm512_set1_epi64(0xFF000000ll);
    const __m512i hi = _mm512_add_epi64(R.lo, R.span);
    k = _mm512_mask_testn_epi64_mask(k, _mm512_xor_si512(R.lo, hi), top);                // top bytes agree
    R.acc = _mm512_mask_or_epi64(R.acc, k, _mm512_slli_epi64(R.acc, 8), _mm512_srli_epi64(hi, 24));
    R.cnt = _mm512_mask_add_epi64(R.cnt, k, R.cnt, _mm512_set1_epi64(1));
    R.lo = _mm512_mask_and_epi64(R.lo, k, _mm512_slli_epi64(R.lo, 8), m32);
    R.span = _mm512_mask_or_epi64(R.span, k, _mm512_slli_epi64(R.span, 8), _mm512_set1_epi64(0xFF));   // span < 2^24 here
    const __mmask8 kf = _mm512_mask_cmpeq_epi64_mask(k, R.cnt, _mm512_set1_epi64(8));
    if (__builtin_expect(kf != 0, 0)) flush_full(R, O, kf);
    return k;
}

// one bin per active lane; ev holds prob in bits 0-11 and the bin at BIN (bit 15 of a raw 16-bit record, bit 12
// of a 13-bit code); whatever else the lane holds is ignored
template <long long BIN = 0x8000>
NB_TARGET NB_INLINE void step(Regs &R, Outs &O, __m512i ev, __mmask8 kact) {
    const __m512i prob = _mm512_and_si512(ev, _mm512_set1_epi64(0xFFF));
    const __m512i t1 = _mm512_add_epi64(_mm512_srli_epi64(_mm512_mul_epu32(R.span, prob), 12), _mm512_set1_epi64(1));
    const __mmask8 kone = _mm512_test_epi64_mask(ev, _mm512_set1_epi64(BIN));
    const __mmask8 k1 = kone & kact, k0 = (__mmask8)(~kone) & kact;
    // bin 1 keeps [lo, cut]: span = t; bin 0 keeps [cut + 1, hi]: lo += t + 1, span -= t + 1
    R.lo = _mm512_mask_add_epi64(R.lo, k0, R.lo, t1);
    R.span = _mm512_mask_sub_epi64(R.span, k0, R.span, t1);
    R.span = _mm512_mask_sub_epi64(R.span, k1, t1, _mm512_set1_epi64(1));
    __mmask8 k = renorm_once(R, O, kact);                    // first byte: executed unconditionally, usually a no-op
    // a second byte in the same step is rare: test before doing the masked work again
    const __m512i top = _mm512_set1_epi64(0xFF000000ll);
    k = _mm512_mask_testn_epi64_mask(k, _mm512_xor_si512(R.lo, _mm512_add_epi64(R.lo, R.span)), top);
    while (__builtin_expect(k != 0, 0)) {
        renorm_once(R, O, k);
        k = _mm512_mask_testn_epi64_mask(k, _mm512_xor_si512(R.lo, _mm512_add_epi64(R.lo, R.span)), top);
    }
}

// The same step when EVERY lane of the register may be updated (all streams of the pack alive;
// lanes beyond the pack's count hold nothing that is kept) -- the steady state, so it is written
// for instruction count: no activity mask, the bin-0 mask straight from a test-not, the interval
// update as two masked operations, and ONE rarely-taken branch for everything unusual (a full byte
// accumulator, a second byte in the same step).
// (prob: the 12-bit probability alone; k0: the lanes whose bin is 0)
NB_TARGET NB_INLINE void step_core(Regs &R, Outs &O, __m512i prob, __mmask8 k0) {
    const __m512i tm = _mm512_srli_epi64(_mm512_mul_epu32(R.span, prob), 12);
    const __m512i t1 = _mm512_add_epi64(tm, _mm512_set1_epi64(1));
    R.lo = _mm512_mask_add_epi64(R.lo, k0, R.lo, t1);                                        // bin 0: lo += t + 1
    R.span = _mm512_mask_sub_epi64(tm, k0, R.span, t1);                                      // bin 0: span -= t + 1; bin 1: span = t
    const __m512i hi = _mm512_add_epi64(R.lo, R.span);
    const __m512i x = _mm512_xor_si512(R.lo, hi);
    const __mmask8 k = _mm512_testn_epi64_mask(x, _mm512_set1_epi64(0xFF000000ll));         // top bytes agree: one byte out
    const __mmask8 k2 = _mm512_testn_epi64_mask(x, _mm512_set1_epi64(0xFFFF0000ll));        // top two bytes agree (rare)
    R.acc = _mm512_mask_shldi_epi64(R.acc, k, R.acc, _mm512_slli_epi64(hi, 32), 8);          // acc = acc << 8 | hi >> 24
    R.cnt = _mm512_mask_add_epi64(R.cnt, k, R.cnt, _mm512_set1_epi64(1));
    R.lo = _mm512_mask_and_epi64(R.lo, k, _mm512_slli_epi64(R.lo, 8), _mm512_set1_epi64(0xFFFFFFFFll));
    R.span = _mm512_mask_or_epi64(R.span, k, _mm512_slli_epi64(R.span, 8), _mm512_set1_epi64(0xFF));
    const __mmask8 kf = _mm512_cmpeq_epi64_mask(R.cnt, _mm512_set1_epi64(8));
    if (__builtin_expect((kf | k2) != 0, 0)) {
        if (kf) flush_full(R, O, kf);
        __mmask8 kk = k2;
        while (kk) {                                          // further bytes of the same step, one at a time
            renorm_once(R, O, kk);
            kk = _mm512_mask_testn_epi64_mask(kk, _mm512_xor_si512(R.lo, _mm512_add_epi64(R.lo, R.span)), _mm512_set1_epi64(0xFF000000ll));
        }
    }
}
template <long long BIN = 0x8000>
NB_TARGET NB_INLINE void step_all(Regs &R, Outs &O, __m512i ev) {
    step_core(R, O, _mm512_and_si512(ev, _mm512_set1_epi64(0xFFF)), _mm512_testn_epi64_mask(ev, _mm512_set1_epi64(BIN)));
}

}  // namespace

// Resumable form: the coder threads stream each image's bins from HBM in chunks, so the eight
// lanes are fed piecewise.  begin() once, feed() per chunk, end() once.
struct RangeX8::State { Lanes L; uint8_t *out0[8]; int count; };

RangeX8::RangeX8() : st(nullptr) {}
RangeX8::~RangeX8() { if (st) { st->~State(); free(st); } }

NB_TARGET void RangeX8::begin(int count, uint8_t *const *outs, const size_t *caps) {
    if (!st) { st = (State *)aligned_alloc(64, (sizeof(State) + 63) & ~size_t(63)); new (st) State; }
    Lanes &L = st->L;
    st->count = count;
    for (int k = 0; k < 8; k++) {
        L.o.outp[k] = k < count ? outs[k] : nullptr;
        L.o.oend[k] = k < count ? outs[k] + caps[k] : nullptr;
        L.o.overflow[k] = false;
        st->out0[k] = L.o.outp[k];
    }
    L.r.lo = _mm512_setzero_si512();
    L.r.span = _mm512_set1_epi64(0xFFFFFFFFll);
    L.r.acc = _mm512_setzero_si512();
    L.r.cnt = _mm512_setzero_si512();
}

// Lane k codes src[k][0 .. len[k]) (len 0 = lane idle in this chunk).  While every active lane has
// four bins left, one 8-byte gather per lane brings the next four bins of every stream; lanes drop
// out of the mask as their chunk ends.
NB_TARGET void RangeX8::feed(const uint16_t *const *src, const size_t *len) {
    Regs R = st->L.r;
    Outs &O = st->L.o;
    const int count = st->count;
    alignas(64) uint64_t base[8];
    for (int k = 0; k < 8; k++) base[k] = (k < count && len[k]) ? (uint64_t)(uintptr_t)src[k] : 0;
    const __m512i vbase = _mm512_load_si512((const void *)base);
    size_t pos = 0;
    for (;;) {
        unsigned act = 0;
        size_t m = SIZE_MAX;
        for (int k = 0; k < count; k++)
            if (len[k] > pos) { act |= 1u << k; if (len[k] < m) m = len[k]; }
        if (!act) break;
        const __mmask8 ka = (__mmask8)act;
        if (act == (1u << count) - 1u) {                      // every stream of the pack alive
            for (; pos + 4 <= m; pos += 4) {
                const __m512i addr = _mm512_add_epi64(vbase, _mm512_set1_epi64((long long)(2 * pos)));
                const __m512i g = _mm512_mask_i64gather_epi64(_mm512_setzero_si512(), ka, addr, (const void *)0, 1);
                step_all(R, O, g);
                step_all(R, O, _mm512_srli_epi64(g, 16));
                step_all(R, O, _mm512_srli_epi64(g, 32));
                step_all(R, O, _mm512_srli_epi64(g, 48));
            }
        }
        for (; pos + 4 <= m; pos += 4) {
            const __m512i addr = _mm512_add_epi64(vbase, _mm512_set1_epi64((long long)(2 * pos)));
            const __m512i g = _mm512_mask_i64gather_epi64(_mm512_setzero_si512(), ka, addr, (const void *)0, 1);
            step(R, O, g, ka);
            step(R, O, _mm512_srli_epi64(g, 16), ka);
            step(R, O, _mm512_srli_epi64(g, 32), ka);
            step(R, O, _mm512_srli_epi64(g, 48), ka);
        }
        for (; pos < m; pos++) {                              // at most three bins: the shortest lane's end
            alignas(64) uint64_t e[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int k = 0; k < count; k++) if (act >> k & 1u) e[k] = src[k][pos];
            step(R, O, _mm512_load_si512((const void *)e), ka);
        }
    }
    st->L.r = R;
}

// Two packs advanced in lock-step.  One pack's step is a ~25-cycle dependent chain of ~35
// instructions, so a core that interleaves two independent packs (sixteen images) nearly doubles
// its throughput.  The joint loop runs while every active lane of both packs has four bins left;
// the few bins after that go through the single-pack feed.
NB_TARGET void feed_pair(RangeX8 &A, const uint16_t *const *src_a, const size_t *len_a,
                         RangeX8 &B, const uint16_t *const *src_b, const size_t *len_b) {
    Regs RA = A.st->L.r, RB = B.st->L.r;
    Outs &OA = A.st->L.o, &OB = B.st->L.o;
    alignas(64) uint64_t base_a[8], base_b[8];
    unsigned act_a = 0, act_b = 0;
    size_t m = SIZE_MAX;
    for (int k = 0; k < 8; k++) {
        const bool a = k < A.st->count && len_a[k], b = k < B.st->count && len_b[k];
        base_a[k] = a ? (uint64_t)(uintptr_t)src_a[k] : 0;
        base_b[k] = b ? (uint64_t)(uintptr_t)src_b[k] : 0;
        if (a) { act_a |= 1u << k; if (len_a[k] < m) m = len_a[k]; }
        if (b) { act_b |= 1u << k; if (len_b[k] < m) m = len_b[k]; }
    }
    size_t pos = 0;
    if (act_a == (1u << A.st->count) - 1u && act_b == (1u << B.st->count) - 1u) {
        const __m512i va = _mm512_load_si512((const void *)base_a), vb = _mm512_load_si512((const void *)base_b);
        const __mmask8 ka = (__mmask8)act_a, kb = (__mmask8)act_b;
        for (; pos + 4 <= m; pos += 4) {
            if ((pos & 31) == 0) {                            // one 64-byte line per stream per 32 bins: ask for it early
                for (int k = 0; k < 8; k++) {
                    _mm_prefetch((const char *)base_a[k] + 2 * pos + kPrefetchAhead, _MM_HINT_T0);
                    _mm_prefetch((const char *)base_b[k] + 2 * pos + kPrefetchAhead, _MM_HINT_T0);
                }
            }
            const __m512i off = _mm512_set1_epi64((long long)(2 * pos));
            const __m512i ga = _mm512_mask_i64gather_epi64(_mm512_setzero_si512(), ka, _mm512_add_epi64(va, off), (const void *)0, 1);
            const __m512i gb = _mm512_mask_i64gather_epi64(_mm512_setzero_si512(), kb, _mm512_add_epi64(vb, off), (const void *)0, 1);
            step_all(RA, OA, ga);                         step_all(RB, OB, gb);
            step_all(RA, OA, _mm512_srli_epi64(ga, 16));  step_all(RB, OB, _mm512_srli_epi64(gb, 16));
            step_all(RA, OA, _mm512_srli_epi64(ga, 32));  step_all(RB, OB, _mm512_srli_epi64(gb, 32));
            step_all(RA, OA, _mm512_srli_epi64(ga, 48));  step_all(RB, OB, _mm512_srli_epi64(gb, 48));
        }
    }
    A.st->L.r = RA;
    B.st->L.r = RB;
    const uint16_t *ra[8], *rb[8]; size_t la[8], lb[8];
    for (int k = 0; k < 8; k++) {
        ra[k] = src_a[k] ? src_a[k] + pos : nullptr; la[k] = (act_a >> k & 1u) ? len_a[k] - pos : 0;
        rb[k] = src_b[k] ? src_b[k] + pos : nullptr; lb[k] = (act_b >> k & 1u) ? len_b[k] - pos : 0;
    }
    A.feed(ra, la);
    B.feed(rb, lb);
}

// The same pair fed from ROW-INTERLEAVED bins: rows[16 * i + lane] holds, as one 64-bit word, bins
// 4i .. 4i+3 of lane `lane` (lanes 0-7 = pack A, 8-15 = pack B; zero where a lane has no bin).  That is
// exactly what the gathers above assemble, so with this layout -- which a GPU kernel produces for the
// coder threads before the bins leave HBM (k_interleave16) -- a pack's next four steps are ONE
// aligned 64-byte load, and a chunk of sixteen images is one contiguous copy.  len[lane] = bins of
// the lane in this chunk (0 = idle).
NB_TARGET void feed_pair_rows(RangeX8 &A, RangeX8 &B, const uint64_t *rows, const size_t *len) {
    Regs RA = A.st->L.r, RB = B.st->L.r;
    Outs &OA = A.st->L.o, &OB = B.st->L.o;
    const int ca = A.st->count, cb = B.st->count;
    unsigned act_a = 0, act_b = 0;
    size_t m = SIZE_MAX, longest = 0;
    alignas(64) uint64_t lv[16];
    for (int k = 0; k < 16; k++) {
        const bool on = (k < 8 ? k < ca : k - 8 < cb) && len[k];
        lv[k] = on ? len[k] : 0;
        if (on) { (k < 8 ? act_a : act_b) |= 1u << (k & 7); if (len[k] < m) m = len[k]; if (len[k] > longest) longest = len[k]; }
    }
    size_t pos = 0;
    if (act_a == (1u << ca) - 1u && act_b == (1u << cb) - 1u && (act_a | act_b)) {
        for (; pos + 4 <= m; pos += 4) {
            const uint64_t *row = rows + 4 * pos;                // 16 words per four bins
            _mm_prefetch((const char *)(row + 16 * 16), _MM_HINT_T0);
            const __m512i ga = _mm512_load_si512((const void *)row), gb = _mm512_load_si512((const void *)(row + 8));
            step_all(RA, OA, ga);                         step_all(RB, OB, gb);
            step_all(RA, OA, _mm512_srli_epi64(ga, 16));  step_all(RB, OB, _mm512_srli_epi64(gb, 16));
            step_all(RA, OA, _mm512_srli_epi64(ga, 32));  step_all(RB, OB, _mm512_srli_epi64(gb, 32));
            step_all(RA, OA, _mm512_srli_epi64(ga, 48));  step_all(RB, OB, _mm512_srli_epi64(gb, 48));
        }
    }
    // whatever is left (lanes of different length, a pack with idle lanes): one bin at a time, lanes masked by their length
    const __m512i la = _mm512_load_si512((const void *)lv), lb = _mm512_load_si512((const void *)(lv + 8));
    for (; pos < longest; pos++) {
        const uint64_t *row = rows + 16 * (pos >> 2);
        const __m512i sh = _mm512_set1_epi64((long long)(16 * (pos & 3))), p = _mm512_set1_epi64((long long)pos);
        const __mmask8 ka = _mm512_cmplt_epu64_mask(p, la), kb = _mm512_cmplt_epu64_mask(p, lb);
        if (ka) step(RA, OA, _mm512_srlv_epi64(_mm512_load_si512((const void *)row), sh), ka);
        if (kb) step(RB, OB, _mm512_srlv_epi64(_mm512_load_si512((const void *)(row + 8)), sh), kb);
    }
    A.st->L.r = RA;
    B.st->L.r = RB;
}

// The pair fed from 13-BIT GROUPS -- the form in which bins cross PCIe.  A record is 12 bits of probability and the
// bin: 13 bits as code13(), and 64 of them are exactly thirteen 64-bit words.  rows[(13 * g + j) * 16 + lane] is word j
// of lane `lane`'s group g (bins 64g .. 64g+63; zero where the lane has no bin):
//     bits  0..51 of word j            codes 4j .. 4j+3 of the group, 13 bits each            (52 codes in 13 words)
//     bits 52..63 of word j, j < 12    the probability of code 52+j
//     bits 52..63 of word 12           bit e = the bin of code 52+e
// so that no code straddles two words and the walk is two SHORT loops with constant shifts -- thirteen times four steps
// per pack exactly as from 16-bit rows, then twelve steps from the top fields.  (First attempt: codes back to back,
// 13k .. 13k+12 of the 832 bits, the 64 x 2 steps fully unrolled since every one has its own shift -- the compiler kept
// the coder state on the stack there, and the walk ran 9-15 % slower than from 16-bit rows.)  A pack's word is still ONE
// aligned 64-byte load, and the link -- which bounds the pipeline -- carries 18.75 % fewer
// bytes.  k_pack_groups (pipeline.hip) writes the layout; pack_groups_host is the same on the host, for tests.
void pack_groups_host(uint64_t *rows, int lane, const uint16_t *coded, size_t len, int lanes) {
    const size_t L = size_t(lanes);
    for (size_t i = 0; i < len; i++) {
        uint64_t *grp = rows + (i >> 6) * kGroupWords * L + size_t(lane);
        const uint64_t c = code13(coded[i]);
        const size_t k = i & 63;
        if (k < 52) grp[L * (k >> 2)] |= c << (13 * (k & 3));
        else { grp[L * (k - 52)] |= (c & 0xFFF) << 52; grp[L * 12] |= (c >> 12) << (52 + (k - 52)); }
    }
}

// NP packs (8 * NP lanes, a word-row is NP aligned 64-byte loads) advanced in lock-step: two are a pack pair, three
// ride a third pack along on whatever the core's ports have left (EPYC 9575F, one thread alone: 2150 -> 2380 Mbins/s).
template <int NP>
NB_TARGET NB_INLINE void feed_groups_np(RangeX8 *const *P, const uint64_t *rows, const size_t *len) {
    constexpr size_t L = 8 * NP;
    Regs R[NP];
    Outs *O[NP];
    unsigned act[NP], full[NP];
    bool all_on = true, any = false;
    size_t m = SIZE_MAX, longest = 0;
    alignas(64) uint64_t lv[L];
    for (int p = 0; p < NP; p++) {
        R[p] = P[p]->st->L.r; O[p] = &P[p]->st->L.o;
        act[p] = 0; full[p] = (1u << P[p]->st->count) - 1u;
        for (int k = 0; k < 8; k++) {
            const size_t n = len[8 * p + k];
            const bool on = k < P[p]->st->count && n;
            lv[8 * p + k] = on ? n : 0;
            if (on) { act[p] |= 1u << k; if (n < m) m = n; if (n > longest) longest = n; }
        }
        all_on = all_on && act[p] == full[p];
        any = any || act[p];
    }
    size_t pos = 0;
    static const size_t ahead_groups = getenv("NBLIC_AMD_PREFETCH_GROUPS") ? size_t(atoi(getenv("NBLIC_AMD_PREFETCH_GROUPS"))) : 2;
    const size_t ahead = ahead_groups * kGroupWords * L;          // words; 2 groups = 3.3 KB per pack pair
    if (all_on && any) {
        for (; pos + kGroupBins <= m; pos += kGroupBins) {
            const uint64_t *grp = rows + (pos >> 6) * (kGroupWords * L);
            for (int j = 0; j < int(kGroupWords); j++) {
                const uint64_t *row = grp + L * j;
                __m512i g[NP];
#pragma GCC unroll 3
                for (int p = 0; p < NP; p++) {
                    _mm_prefetch((const char *)(row + ahead + 8 * p), _MM_HINT_T0);
                    g[p] = _mm512_load_si512((const void *)(row + 8 * p));
                }
#pragma GCC unroll 3
                for (int p = 0; p < NP; p++) step_all<0x1000>(R[p], *O[p], g[p]);
#pragma GCC unroll 3
                for (int p = 0; p < NP; p++) step_all<0x1000>(R[p], *O[p], _mm512_srli_epi64(g[p], 13));
#pragma GCC unroll 3
                for (int p = 0; p < NP; p++) step_all<0x1000>(R[p], *O[p], _mm512_srli_epi64(g[p], 26));
#pragma GCC unroll 3
                for (int p = 0; p < NP; p++) step_all<0x1000>(R[p], *O[p], _mm512_srli_epi64(g[p], 39));
            }
            __m512i bins[NP];
#pragma GCC unroll 3
            for (int p = 0; p < NP; p++) bins[p] = _mm512_load_si512((const void *)(grp + L * 12 + 8 * p));
            __m512i bit = _mm512_set1_epi64(1ll << 52);
            for (int e = 0; e < 12; e++) {
                const uint64_t *row = grp + L * e;
#pragma GCC unroll 3
                for (int p = 0; p < NP; p++)
                    step_core(R[p], *O[p], _mm512_srli_epi64(_mm512_load_si512((const void *)(row + 8 * p)), 52), _mm512_testn_epi64_mask(bins[p], bit));
                bit = _mm512_slli_epi64(bit, 1);
            }
        }
    }
    // whatever is left (lanes of different length, a pack with idle lanes): one bin at a time, lanes masked by their length
    for (; pos < longest; pos++) {
        const uint64_t *grp = rows + (pos >> 6) * (kGroupWords * L);
        const size_t k = pos & 63;
        const __m512i at = _mm512_set1_epi64((long long)pos);
        for (int p = 0; p < NP; p++) {
            const __mmask8 kk = _mm512_cmplt_epu64_mask(at, _mm512_load_si512((const void *)(lv + 8 * p)));
            if (!kk) continue;
            __m512i ev;
            if (k < 52) {
                ev = _mm512_srl_epi64(_mm512_load_si512((const void *)(grp + L * (k >> 2) + 8 * p)), _mm_cvtsi64_si128((long long)(13 * (k & 3))));
            } else {
                const __m512i bin = _mm512_srl_epi64(_mm512_load_si512((const void *)(grp + L * 12 + 8 * p)), _mm_cvtsi64_si128((long long)(52 + (k - 52))));
                ev = _mm512_or_si512(_mm512_srli_epi64(_mm512_load_si512((const void *)(grp + L * (k - 52) + 8 * p)), 52),
                                     _mm512_and_si512(_mm512_slli_epi64(bin, 12), _mm512_set1_epi64(0x1000)));
            }
            step<0x1000>(R[p], *O[p], ev, kk);
        }
    }
    for (int p = 0; p < NP; p++) P[p]->st->L.r = R[p];
}

NB_TARGET void feed_pair_groups(RangeX8 &A, RangeX8 &B, const uint64_t *rows, const size_t *len) {
    RangeX8 *const P[2] = {&A, &B};
    feed_groups_np<2>(P, rows, len);
}
NB_TARGET void feed_triple_groups(RangeX8 &A, RangeX8 &B, RangeX8 &C, const uint64_t *rows, const size_t *len) {
    RangeX8 *const P[3] = {&A, &B, &C};
    feed_groups_np<3>(P, rows, len);
}

// leftovers of the byte accumulators, then the 4-byte flush of lo (NBLIC.c:576-586)
NB_TARGET void RangeX8::end(size_t *lens) {
    Lanes &L = st->L;
    alignas(64) uint64_t a[8], c[8], lo[8];
    _mm512_store_si512((void *)a, L.r.acc);
    _mm512_store_si512((void *)c, L.r.cnt);
    _mm512_store_si512((void *)lo, L.r.lo);
    for (int k = 0; k < st->count; k++) {
        uint8_t *p = L.o.outp[k];
        const int left = (int)c[k];
        if (L.o.overflow[k] || p + left + 4 > L.o.oend[k]) { lens[k] = SIZE_MAX; continue; }
        for (int i = left - 1; i >= 0; i--) *p++ = (uint8_t)(a[k] >> (8 * i));
        uint32_t v = (uint32_t)lo[k];
        for (int i = 0; i < 4; i++) { *p++ = (uint8_t)(v >> 24); v <<= 8; }
        lens[k] = (size_t)(p - st->out0[k]);
    }
}

// Codes `count` (<= 8) whole streams.  coded[k][0..n[k]) are u16 bins (prob | bin << 15); outs[k] has
// caps[k] bytes.  lens[k] = bytes written (coder bytes + 4 flush bytes) or SIZE_MAX if it did not fit.
void range_code_x8(const uint16_t *const *coded, const size_t *n, int count, uint8_t *const *outs,
                   const size_t *caps, size_t *lens) {
    if (count <= 0) return;
    if (count > 8) {                                          // up to sixteen: two packs in lock-step
        RangeX8 a, b;
        a.begin(8, outs, caps);
        b.begin(count - 8 < 8 ? count - 8 : 8, outs + 8, caps + 8);
        const uint16_t *sb[8] = {nullptr}; size_t nb[8] = {0};
        for (int k = 8; k < count && k < 16; k++) { sb[k - 8] = coded[k]; nb[k - 8] = n[k]; }
        feed_pair(a, coded, n, b, sb, nb);
        a.end(lens);
        b.end(lens + 8);
        return;
    }
    RangeX8 x;
    x.begin(count, outs, caps);
    x.feed(coded, n);
    x.end(lens);
}

// Measured (EPYC 9575F, one thread): one pack 1280 Mbins/s (27 cycles per step: the latency of its
// dependent chain), two packs in lock-step 2000 (17.6 cycles per pack-step), three packs 2220
// (15.8) -- beyond two the core is bound by throughput (the four mask-producing compares and two
// mask moves per step), so a third pack buys 11 % for 50 % more images in flight per thread.
// Two it is.
//
// Measured and rejected (EPYC 9575F): 2-4 streams interleaved in general-purpose registers.
// With the data-dependent branches kept, two interleaved streams reach 633 Mbins/s against 515 for
// one; fully branch-free they top out at ~550 Mbins/s for ANY stream count -- the scalar coder is
// bound by the core's instruction throughput (~30 instructions per bin), not by its dependency
// chain, so only the vector form above buys anything.

}  // namespace nblic

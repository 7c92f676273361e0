// model.h -- the NBLIC v0.3 per-pixel model as host/device inline functions.
//
// Everything here is integer arithmetic that defines the .nblic bitstream; the
// constants are frozen (reference: src/NBLIC.c:45-90).  The functions are written
// for the GPU first (branch-light, no pointers into image memory: callers hand in
// the twelve causal taps), and are shared by the staged -e1 kernels, the serial
// engine and the small amount of host code that writes headers.
//
// Reference lines each block reproduces are cited next to it (paths relative to
// the reference's src/).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define NB_HD __host__ __device__ __forceinline__
#else
#define NB_HD inline
#endif

namespace nblic {

constexpr int kMaxVal     = 255;
constexpr int kMid        = 128;
constexpr int kMaxNear    = 9;      // 255 / 26
constexpr int kMinKStep   = 3;
constexpr int kLevels     = 16;     // N_QD: activity levels == counter trees
constexpr int kContexts   = 2048;   // (N_QD/2) * 256
constexpr int kCtxCoef    = 7;
constexpr int kCtxScale   = 8;
constexpr int kWeightOne  = 32;     // N_QW
constexpr int kMapSyms    = 20;     // N_MAPPER
constexpr int kCountLimit = 32 * 256;  // N_QW * MAX_COUNTER
constexpr int kProbOne    = 4096;
constexpr int kTreeNodes  = 256;
constexpr int kHeaderBytes = 16;
constexpr long kMaxPixels = 100000000L;   // NBLIC.h:31

NB_HD int iabs(int v) { return v < 0 ? -v : v; }
NB_HD int iclip(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Twelve causal taps around pixel x (NBLIC.c:287-304):
//      s h f g r        row i-2
//      q c b d t        row i-1
//      e a x            row i
struct Taps { int a, b, c, d, e, f, g, h, q, r, s, t; };

// Fetches the taps with the reference's chained fall-backs.  `Pix` is any callable
// (int row, int col) -> int that is only invoked for in-image coordinates.
template <class Pix>
NB_HD Taps sample_taps(Pix pix, int w, int i, int j) {
    Taps n;
    const bool up1 = i >= 1, up2 = i >= 2, l1 = j >= 1, l2 = j >= 2, r1 = j + 1 < w, r2 = j + 2 < w;
    int a = l1 ? pix(i, j - 1) : kMid;
    int b = up1 ? pix(i - 1, j) : kMid;
    if (!up1) b = a; else if (!l1) a = b;
    n.a = a; n.b = b;
    n.e = l2 ? pix(i, j - 2) : a;
    n.c = (up1 && l1) ? pix(i - 1, j - 1) : b;
    n.d = (up1 && r1) ? pix(i - 1, j + 1) : b;
    n.f = up2 ? pix(i - 2, j) : b;
    n.g = (up2 && r1) ? pix(i - 2, j + 1) : n.f;
    n.h = (up2 && l1) ? pix(i - 2, j - 1) : n.f;
    n.q = (up1 && l2) ? pix(i - 1, j - 2) : n.c;
    n.r = (up2 && r2) ? pix(i - 2, j + 2) : n.g;
    n.s = (up2 && l2) ? pix(i - 2, j - 2) : n.h;
    n.t = (up1 && r2) ? pix(i - 1, j + 2) : n.d;
    return n;
}

// Seven-direction blended predictor (NBLIC.c:307-370).  A direction's cost is the error it
// would have made at the four coded probes a, c, b, d, each seen with its own west /
// north-west / north / north-east neighbours.  Candidates are tested in the reference's
// order with a strict '<' so the first minimum wins.
NB_HD void keep_min(int cost, int cand, int &best, int &ang) { if (cost < best) { best = cost; ang = cand; } }

NB_HD int predict(const Taps &n) {
    const int a = n.a, b = n.b, c = n.c, d = n.d, e = n.e, f = n.f, g = n.g, h = n.h, q = n.q, r = n.r, s = n.s;
    int lin = iclip(9 * (a + b) + 2 * (d - c) - e - f, 0, 16 * kMaxVal);
    int cw  = 2 * (iabs(a - e) + iabs(c - q) + iabs(b - c) + iabs(d - b));
    int cn  = 2 * (iabs(a - c) + iabs(c - h) + iabs(b - f) + iabs(d - g));
    int cnw = 2 * (iabs(a - q) + iabs(c - s) + iabs(b - h) + iabs(d - f));
    int cne = 2 * (iabs(a - b) + iabs(c - f) + iabs(b - g) + iabs(d - r));
    int c1  = iabs(2 * a - e - q) + iabs(2 * c - q - s) + iabs(2 * b - c - h) + iabs(2 * d - b - f);
    int c2  = iabs(2 * a - q - c) + iabs(2 * c - s - h) + iabs(2 * b - h - f) + iabs(2 * d - f - g);
    int c3  = iabs(2 * a - c - b) + iabs(2 * c - h - f) + iabs(2 * b - f - g) + iabs(2 * d - g - r);
    int best = 0xFFFFFF, ang = 0;
    keep_min(cw, 2 * a, best, ang);  keep_min(cn, 2 * b, best, ang);
    keep_min(cnw, 2 * c, best, ang); keep_min(cne, 2 * d, best, ang);
    keep_min(c1, a + c, best, ang);  keep_min(c2, c + b, best, ang);  keep_min(c3, b + d, best, ang);
    int spread = cw + cn + cnw + cne + c1 + c2 + c3 - 7 * best;
    int wt = (spread >= 31) + (spread >= 93) + (spread >= 279) + (spread >= 620) +
             (spread >= 1550) + (spread >= 3410) + (spread >= 9300) + (spread >= 24800);
    return (8 * wt * ang + (8 - wt) * lin + 64) >> 7;
}

// Local activity (NBLIC.c:376) and its soft quantisation onto two adjacent levels
// (NBLIC.c:373-395): level `qu` carries weight 32-qw, level `qv` weight qw, qw in [0,16].
NB_HD int activity(const Taps &n, int err_prev) {
    return iabs(n.a - n.e) + iabs(n.b - n.c) + iabs(n.b - n.d) + iabs(n.a - n.c) +
           iabs(n.b - n.f) + iabs(n.d - n.g) + 2 * iabs(err_prev);
}

// ---- QNBLIC (effort 0) variants, reference: src/QNBLIC.c ------------------------------------
// Its neighbourhood is a running window (QNBLIC.c:48-79) that differs from direct sampling on
// rows 0-1 and at column 1; written here in closed form so that a pixel
// can be evaluated on its own.  `Pix` is only called for in-image coordinates.  Tap t is unused.
template <class Pix>
NB_HD Taps sample_taps_q(Pix pix, int w, int i, int j) {
    Taps n{};
    auto cl = [w](int c) { return c < 0 ? 0 : (c >= w ? w - 1 : c); };
    if (i == 0) {
        auto x0 = [&](int k) { return k >= 0 ? pix(0, k) : kMid; };
        n.a = x0(j - 1); n.e = x0(j - 2); n.d = n.a; n.b = n.e; n.c = x0(j - 3); n.q = x0(j - 4);
        n.r = n.a; n.g = n.e; n.f = n.c; n.h = n.q; n.s = x0(j - 5);
        return n;
    }
    const int u0 = pix(i - 1, 0);
    n.b = pix(i - 1, cl(j)); n.c = pix(i - 1, cl(j - 1)); n.q = pix(i - 1, cl(j - 2)); n.d = pix(i - 1, cl(j + 1));
    n.a = j >= 1 ? pix(i, j - 1) : u0;
    n.e = j >= 2 ? pix(i, j - 2) : u0;               // also row i-1's first pixel at j == 1 (the window hands on a(0))
    if (i == 1) {
        n.r = j >= 1 ? pix(0, cl(j + 1)) : u0;
        n.g = j >= 2 ? pix(0, cl(j)) : u0;
        n.f = j >= 3 ? pix(0, j - 1) : u0;
        n.h = j >= 4 ? pix(0, j - 2) : u0;
        n.s = j >= 5 ? pix(0, j - 3) : u0;
    } else {
        n.f = pix(i - 2, cl(j)); n.h = pix(i - 2, cl(j - 1)); n.s = pix(i - 2, cl(j - 2));
        n.g = pix(i - 2, cl(j + 1)); n.r = pix(i - 2, cl(j + 2));
    }
    return n;
}

// QNBLIC.c:94-149: the same seven directions, blend weight 0..7 from (spread >> 3)
NB_HD int predict_q(const Taps &n) {
    const int a = n.a, b = n.b, c = n.c, d = n.d, e = n.e, f = n.f, g = n.g, h = n.h, q = n.q, r = n.r, s = n.s;
    int lin = iclip(9 * (a + b) + 2 * (d - c) - e - f, 0, 16 * kMaxVal);
    int cw  = 2 * (iabs(a - e) + iabs(c - q) + iabs(b - c) + iabs(d - b));
    int cn  = 2 * (iabs(a - c) + iabs(c - h) + iabs(b - f) + iabs(d - g));
    int cnw = 2 * (iabs(a - q) + iabs(c - s) + iabs(b - h) + iabs(d - f));
    int cne = 2 * (iabs(a - b) + iabs(c - f) + iabs(b - g) + iabs(d - r));
    int c1  = iabs(2 * a - e - q) + iabs(2 * c - q - s) + iabs(2 * b - c - h) + iabs(2 * d - b - f);
    int c2  = iabs(2 * a - q - c) + iabs(2 * c - s - h) + iabs(2 * b - h - f) + iabs(2 * d - f - g);
    int c3  = iabs(2 * a - c - b) + iabs(2 * c - h - f) + iabs(2 * b - f - g) + iabs(2 * d - g - r);
    int best = cw, ang = 2 * a;
    keep_min(cn, 2 * b, best, ang);  keep_min(cnw, 2 * c, best, ang); keep_min(cne, 2 * d, best, ang);
    keep_min(c1, a + c, best, ang);  keep_min(c2, c + b, best, ang);  keep_min(c3, b + d, best, ang);
    int v = (cw + cn + cnw + cne + c1 + c2 + c3 - 7 * best) >> 3;
    int wt = (v >= 5) + (v >= 12) + (v >= 34) + (v >= 78) + (v >= 194) + (v >= 431) + (v >= 601);
    return (8 * wt * ang + (8 - wt) * lin + 64) >> 7;
}
// QNBLIC.c:152-161, :599-601: twelve hard levels; err_prev is the UNCLIPPED x - px0 of the left pixel
NB_HD int level_q(const Taps &n, int err_prev) {
    int v = activity(n, err_prev);
    return (v >= 1) + (v >= 2) + (v >= 4) + (v >= 6) + (v >= 9) + (v >= 15) + (v >= 25) + (v >= 39) + (v >= 63) + (v >= 101) + (v >= 151);
}
// QNBLIC.c:164-173: level in the high bits, comparisons MSB first
NB_HD int context_address_q(const Taps &n, int qd, int px0) {
    return (qd << 8) | (int(px0 > n.a) << 7) | (int(px0 > n.b) << 6) | (int(px0 > n.c) << 5) | (int(px0 > n.d) << 4) |
           (int(px0 > n.e) << 3) | (int(px0 > n.f) << 2) | (int(px0 > 2 * n.a - n.e) << 1) | int(px0 > 2 * n.b - n.f);
}

struct Level { int qu, qv, qw; };

NB_HD int level_centre(int k) {
    // {0,2,4,7,10,14,20,26,34,42,52,64,78,95,135,200} packed as a switch so it stays in registers
    switch (k) {
        case 0: return 0;  case 1: return 2;  case 2: return 4;  case 3: return 7;
        case 4: return 10; case 5: return 14; case 6: return 20; case 7: return 26;
        case 8: return 34; case 9: return 42; case 10: return 52; case 11: return 64;
        case 12: return 78; case 13: return 95; case 14: return 135; default: return 200;
    }
}

NB_HD Level quantise(int delta) {
    int qd = (delta > 0) + (delta > 2) + (delta > 4) + (delta > 7) + (delta > 10) + (delta > 14) +
             (delta > 20) + (delta > 26) + (delta > 34) + (delta > 42) + (delta > 52) + (delta > 64) +
             (delta > 78) + (delta > 95) + (delta > 135);
    Level L{qd, qd, 0};
    int hi = level_centre(qd);
    if (delta < hi) {
        int lo = level_centre(qd - 1);
        int w = kWeightOne * (delta - lo) / (hi - lo);
        if (w < kWeightOne / 2) { L.qu = qd - 1; L.qw = w; }
        else                    { L.qv = qd - 1; L.qw = kWeightOne - w; }
    }
    return L;
}

// Context address (NBLIC.c:398-410): coarse activity plus eight texture comparisons.
NB_HD int context_address(const Taps &n, int qu, int px0) {
    return ((qu >> 1) << 8) | int(px0 > n.a) | (int(px0 > n.b) << 1) | (int(px0 > n.c) << 2) |
           (int(px0 > n.d) << 3) | (int(px0 > n.e) << 4) | (int(px0 > n.f) << 5) |
           (int(px0 > 2 * n.a - n.e) << 6) | (int(px0 > 2 * n.b - n.f) << 7);
}

// Context bias state v ~ 256 x running mean of the prediction error (NBLIC.c:413-428).
// Right shifts of negative ints are arithmetic on every target we build for (gfx950, x86-64).
NB_HD int bias_sign(int v) { return (v >> (kCtxScale - 1)) & 1; }
NB_HD int bias_apply(int v, int px0) { return iclip(px0 + (v >> kCtxScale) + bias_sign(v), 0, kMaxVal); }
NB_HD int bias_update(int v, int err) {
    // 127*v written as (v << 7) - v: a 32-bit integer multiply is a quarter-rate op on the chain's critical path
    return (v * (1 << kCtxCoef) - v + err * (1 << kCtxScale) + (1 << (kCtxCoef - 1))) >> kCtxCoef;
}
NB_HD int clip_err(int x, int px0) { return iclip(x - px0, -(kMaxVal - kMid), kMaxVal - kMid); }

// Residual folding (NBLIC.c:431-466).
NB_HD int fold_limit(int px, int near) {
    int m = px < kMaxVal - px ? px : kMaxVal - px;
    return (m + near) / (2 * near + 1);
}
NB_HD int residual_to_symbol(int x, int px, int sign, int near) {
    int ty = fold_limit(px, near);
    int y = (iabs(x - px) + near) / (2 * near + 1);
    if (y <= 0) return 0;
    if (y <= ty) return 2 * y - (int(x >= px) ^ sign);
    return y + ty;
}
NB_HD int symbol_to_pixel(int y, int px, int sign, int near) {
    int ty = fold_limit(px, near), mag, up;
    if (y <= 0)           { mag = 0;           up = 0; }
    else if (y <= 2 * ty) { mag = (y + 1) >> 1; up = (y & 1) ^ sign; }
    else                  { mag = y - ty;      up = px < kMid; }
    mag *= 2 * near + 1;
    return iclip(up ? px + mag : px - mag, 0, kMaxVal);
}

// Adaptive binary counter (NBLIC.c:589-637).
struct Counter { int c0, c1; };
NB_HD int counter_p1(int c0, int c1) { return kProbOne * c1 / (c0 + c1); }
NB_HD void counter_add(Counter &c, int bin, int weight) {
    if (bin) c.c1 += weight; else c.c0 += weight;
    if (c.c0 + c.c1 > kCountLimit) { c.c0 = (c.c0 + 1) >> 1; c.c1 = (c.c1 + 1) >> 1; }
}
NB_HD int mix_prob(int pu, int pv, int qw) {
    return iclip((pu * (kWeightOne - qw) + pv * qw + kWeightOne / 2) >> 5, 1, kProbOne - 1);
}

NB_HD int k_step_for_near(int near) { return iclip(kMinKStep + 2 * near, kMinKStep, kLevels); }

// Binarisation walk (NBLIC.c:640-679).  Visits (tree_u, tree_v, node) in coding order and
// calls step(qu, qv, node, bin) -> bin.  Encoding passes the known bin; decoding passes -1
// and continues with whatever `step` returns.  Returns the symbol.
template <class Step>
NB_HD int walk_symbol(int k_step, int qu, int qv, int z_in, Step step) {
    const int k_max = (kLevels - 1) / k_step;
    const bool decoding = z_in < 0;
    int node = 0, k, bin, z;
    if (qv / k_step != qu / k_step) qv = qu;
    for (;;) {
        k = qu / k_step;
        bin = decoding ? -1 : int((node >> k_max) < (z_in >> k));
        bin = step(qu, qv, node, bin);
        if (!bin) break;
        node += 1 << k_max;
        if (node >= kTreeNodes) { node >>= 1; qu = qv = (k + 1) * k_step; }
    }
    z = decoding ? ((node >> k_max) << k) : z_in;
    node++;
    for (k--; k >= 0; k--) {
        bin = decoding ? -1 : ((z_in >> k) & 1);
        bin = step(qu, qv, node, bin);
        if (decoding && bin) z += 1 << k;
        node += bin ? (1 << k) : 1;
    }
    return z;
}

// ---- near-lossless arithmetic without integer divides (the serial kernels run one pixel at a time:
// a divide by the run-time step 2*near+1 or by k_step would cost ~40 instructions each) ----------
// floor(num / (2*near+1)) for 0 <= num < 2048 by reciprocal multiply (checked exhaustively in tests/)
struct NearParams { int near, width, recip; };
NB_HD NearParams near_params(int near) { return NearParams{near, 2 * near + 1, 65536 / (2 * near + 1) + 1}; }
NB_HD int div_width(int num, const NearParams &p) { return (num * p.recip) >> 16; }
NB_HD int residual_to_symbol(int x, int px, int sign, const NearParams &p) {
    int m = px < kMaxVal - px ? px : kMaxVal - px;
    int ty = div_width(m + p.near, p);
    int y = div_width(iabs(x - px) + p.near, p);
    if (y <= 0) return 0;
    if (y <= ty) return 2 * y - (int(x >= px) ^ sign);
    return y + ty;
}
NB_HD int symbol_to_pixel(int y, int px, int sign, const NearParams &p) {
    int m = px < kMaxVal - px ? px : kMaxVal - px;
    int ty = div_width(m + p.near, p), mag, up;
    if (y <= 0)           { mag = 0;           up = 0; }
    else if (y <= 2 * ty) { mag = (y + 1) >> 1; up = (y & 1) ^ sign; }
    else                  { mag = y - ty;      up = px < kMid; }
    mag *= p.width;
    return iclip(up ? px + mag : px - mag, 0, kMaxVal);
}

// What the decoder will reconstruct for pixel x predicted as px (NBLIC.c:431-466): symbol_to_pixel(residual_to_symbol(x))
// without the symbol.  The quantised magnitude q survives the folding unchanged and the direction is x >= px in both of
// its branches (beyond the fold limit only the side with room is left, which is where x lies); checked exhaustively
// against the round trip in tests/host_harness.cpp.  sign does not matter.
NB_HD int reconstruct_pixel(int x, int px, const NearParams &p) {
    const int d = x - px, mag = div_width((d < 0 ? -d : d) + p.near, p) * p.width;
    return iclip(d >= 0 ? px + mag : px - mag, 0, kMaxVal);
}

// level / k_step for the sixteen levels, four bits each (bits 60..63 = 15 / k_step = k_max)
NB_HD uint64_t level_shift_table(int k_step) {
    uint64_t t = 0;
    for (int q = 0; q < kLevels; q++) t |= uint64_t(q / k_step) << (4 * q);
    return t;
}
// walk_symbol with the divisions looked up in level_shift_table(k_step)
template <class Step>
NB_HD int walk_symbol_t(int k_step, uint64_t ktab, int qu, int qv, int z_in, Step step) {
    const int k_max = int(ktab >> 60);
    const bool decoding = z_in < 0;
    int node = 0, bin, z;
    int k = int(ktab >> (4 * qu)) & 15;
    if ((int(ktab >> (4 * qv)) & 15) != k) qv = qu;
    for (;;) {
        bin = decoding ? -1 : int((node >> k_max) < (z_in >> k));
        bin = step(qu, qv, node, bin);
        if (!bin) break;
        node += 1 << k_max;
        if (node >= kTreeNodes) { node >>= 1; k++; qu = qv = k * k_step; }
    }
    z = decoding ? ((node >> k_max) << k) : z_in;
    node++;
    for (k--; k >= 0; k--) {
        bin = decoding ? -1 : ((z_in >> k) & 1);
        bin = step(qu, qv, node, bin);
        if (decoding && bin) z += 1 << k;
        node += bin ? (1 << k) : 1;
    }
    return z;
}

// The suffix bits of a symbol walk down a binary tree below the node the prefix stopped at: a one at depth d moves on
// by 2^(k-1-d) nodes, a zero by one (walk_symbol's second loop).  The decoders hold that tree in heap order, one node
// per lane (serial_engine.hip decode_symbol): lane l is the node at depth d = floor(log2(l + 1)) reached by the bits
// prefix = l + 1 - 2^d (first bit in the most significant place), and its distance from the tree's root is
NB_HD int suffix_lane_offset(int k, int d, int prefix) {
    return d - __builtin_popcount(unsigned(prefix)) + (prefix << (k > d ? k - d : 0));
}

// ---- record packing shared by the staged -e1 kernels -------------------------------------
// S1 record, one u32 per pixel:  px0[0:8) | adr[8:19) | qw[19:24) | qu_lsb[24] | qv_rel[25:27)
// qv_rel: 0 -> qv == qu, 1 -> qv == qu + 1, 2 -> qv == qu - 1.  qu = ((adr >> 8) << 1) | qu_lsb.
NB_HD uint32_t pack_s1(int px0, int adr, const Level &L) {
    uint32_t rel = L.qv == L.qu ? 0u : (L.qv > L.qu ? 1u : 2u);
    return uint32_t(px0) | (uint32_t(adr) << 8) | (uint32_t(L.qw) << 19) | (uint32_t(L.qu & 1) << 24) | (rel << 25);
}
NB_HD int s1_px0(uint32_t r) { return int(r & 0xFF); }
NB_HD int s1_adr(uint32_t r) { return int((r >> 8) & 0x7FF); }
NB_HD Level s1_level(uint32_t r) {
    Level L;
    L.qu = int(((r >> 16) & 7) << 1) | int((r >> 24) & 1);
    uint32_t rel = (r >> 25) & 3;
    L.qv = rel == 0 ? L.qu : (rel == 1 ? L.qu + 1 : L.qu - 1);
    L.qw = int((r >> 19) & 31);
    return L;
}

// Bin event, one u32:  tree_u[0:4) | tree_v[4:8) | node[8:16) | qw[16:21) | bin[21]
NB_HD uint32_t pack_event(int qu, int qv, int node, int qw, int bin) {
    return uint32_t(qu) | (uint32_t(qv) << 4) | (uint32_t(node) << 8) | (uint32_t(qw) << 16) | (uint32_t(bin) << 21);
}
NB_HD int ev_qu(uint32_t e) { return int(e & 15); }
NB_HD int ev_qv(uint32_t e) { return int((e >> 4) & 15); }
NB_HD int ev_node(uint32_t e) { return int((e >> 8) & 255); }
NB_HD int ev_qw(uint32_t e) { return int((e >> 16) & 31); }
NB_HD int ev_bin(uint32_t e) { return int((e >> 21) & 1); }

// Coded bin handed to the host range coder, one u16: prob[0:12) | bin[15]
NB_HD uint16_t pack_coded(int prob, int bin) { return uint16_t(prob | (bin << 15)); }

}  // namespace nblic

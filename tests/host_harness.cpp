// tests/host_harness.cpp -- TEST HARNESS (host only): a scalar, fused NBLIC encoder put together
// from the PRODUCT's host-compilable headers (csrc/model.h, csrc/lsq_f64.h), so that the integer
// model functions and the double-carried least-squares arithmetic the GPU kernels use can be
// checked against the oracle without a GPU (tests/test_host_logic.py).  It emits the coded-bin
// stream (prob | bin << 15 per bin); the test turns that into bytes with the product's host range
// coder (nblic_amd_range_code).  Not part of the product library; never a fallback for it.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "../nblic-image-compression_amd/csrc/lane_table.h"
#include "../nblic-image-compression_amd/csrc/lsq_f64.h"
#include "../nblic-image-compression_amd/csrc/model.h"

using namespace nblic;
typedef long long i64;
typedef unsigned long long u64;

namespace {

i64 mulw(i64 a, i64 b) { return i64(u64(a) * u64(b)); }
i64 abs64(i64 v) { return v < 0 ? -v : v; }

// plain-integer solve of the augmented system M[n][n+1] (NBLIC.c:112-161, :233-236): the fallback
bool solve_int(int n, const double Md[lsq::kMaxN][lsq::kMaxN + 1], const int *vn, i64 *px) {
    i64 M[lsq::kMaxN][lsq::kMaxN + 1];
    for (int i = 0; i < n; i++) for (int j = 0; j <= n; j++) M[i][j] = i64(Md[i][j]);
    for (int k = 0; k + 1 < n; k++) {
        int piv = k;
        for (int i = k + 1; i < n; i++) if (abs64(M[i][k]) > abs64(M[piv][k])) piv = i;
        if (piv != k) for (int j = 0; j <= n; j++) { i64 t = M[k][j]; M[k][j] = M[piv][j]; M[piv][j] = t; }
        const i64 d = M[k][k];
        if (d == 0) return false;
        for (int i = k + 1; i < n; i++) {
            const i64 l = M[i][k];
            for (int j = k + 1; j <= n; j++) M[i][j] -= mulw(M[k][j], l) / d;
        }
    }
    for (int k = n - 1; k > 0; k--) {
        const i64 d = M[k][k];
        if (d == 0) return false;
        for (int i = 0; i < k; i++) M[i][n] -= mulw(M[k][n], M[i][k]) / d;
    }
    i64 p = i64(kMid) << lsq::kFb1;
    for (int k = 0; k < n; k++) { const i64 d = M[k][k]; p += (mulw(mulw(M[k][n], vn[k]), 1 << lsq::kFb2) + (d >> 1)) / d; }
    *px = p;
    return true;
}

// the same solve in doubles, organised the way the GPU does it: rows stay where they are and carry
// a position; the pivot is the candidate with the largest |entry|, ties to the smallest position
bool solve_f64(int n, double M[lsq::kMaxN][lsq::kMaxN + 1], const int *vn, lsq::Guard &g, double *px) {
    int pos[lsq::kMaxN], at[lsq::kMaxN];
    double diag[lsq::kMaxN];
    for (int i = 0; i < n; i++) { pos[i] = i; at[i] = i; for (int j = 0; j <= n; j++) g.see_entry(M[i][j]); }
    for (int k = 0; k + 1 < n; k++) {
        int c = -1; double best = -1.0;
        for (int r = 0; r < n; r++) {
            if (pos[r] < k) continue;
            const double key = fabs(M[r][k]) * 256.0 + double((15 - pos[r]) * 16 + r);
            if (key > best) { best = key; c = r; }
        }
        const int a = at[k];                                   // row that sat at position k
        at[pos[c]] = a; pos[a] = pos[c]; pos[c] = k; at[k] = c;
        const double d = M[c][k];
        diag[c] = d;
        if (d == 0.0) return false;
        const lsq::Recip rd = lsq::recip_of(d);
        g.see_pivot(d);
        for (int r = 0; r < n; r++) {
            if (pos[r] <= k) continue;
            const double l = M[r][k];
            for (int j = k + 1; j <= n; j++) {
                M[r][j] -= lsq::muldiv_trunc(M[c][j], l, d, rd, g);
                g.see_entry(M[r][j]);
            }
        }
    }
    diag[at[n - 1]] = M[at[n - 1]][n - 1];
    for (int k = n - 1; k > 0; k--) {
        const int c = at[k];
        const double d = diag[c];
        if (d == 0.0) return false;
        const lsq::Recip rd = lsq::recip_of(d);
        const double bk = M[c][n];
        g.see_pivot(d);
        for (int r = 0; r < n; r++) {
            if (pos[r] >= k) continue;
            M[r][n] -= lsq::muldiv_trunc(bk, M[r][k], d, rd, g);
            g.see_entry(M[r][n]);
        }
    }
    double p = double(kMid << lsq::kFb1);
    for (int r = 0; r < n; r++) { g.see_pivot(diag[r]); p += lsq::term(M[r][n], vn[pos[r]], diag[r], lsq::recip_of(diag[r]), g); }
    *px = p;
    return true;
}

}  // namespace

extern "C" long hh_model_encode(const uint8_t *img_in, uint8_t *recon, int h, int w, int near, int effort, uint16_t *coded, long cap,
                                long *fallbacks) {
    const int n = lsq::order_of(effort), m = lsq::vec_len(n);
    const int k_step = k_step_for_near(near);
    std::vector<int> ctx(kContexts, 0);
    std::vector<Counter> tree(size_t(kLevels) * kTreeNodes, Counter{kWeightOne, kWeightOne});
    struct Mapper { int count[kMapSyms]; uint8_t rank_of[kMapSyms], sym_at[kMapSyms]; };
    std::vector<Mapper> maps(512);
    for (auto &mp : maps) for (int s = 0; s < kMapSyms; s++) { mp.count[s] = 2 * (kMapSyms - 1 - s); mp.rank_of[s] = mp.sym_at[s] = uint8_t(s); }
    std::vector<double> B(size_t(n ? w : 0) * m, 0.0), F(size_t(n ? w : 0) * m, 0.0), E(m, 0.0);
    memcpy(recon, img_in, size_t(h) * w);
    long n_bins = 0, n_fallback = 0;
    int bias = lsq::kBiasInit;
    auto pix = [&](int r, int c) { return int(recon[size_t(r) * w + c]); };
    for (int i = 0; i < h; i++) {
        int err = 0;
        if (n) {
            for (int k = 0; k < m; k++) E[k] = 0.0;
            for (int k = 0; k < m; k++) {                     // NBLIC.c:186-204
                double carry = 0.0;
                for (int j = w - 1; j >= 0; j--) { const double f = carry + B[size_t(j) * m + k]; F[size_t(j) * m + k] = f; carry = lsq::decay_k(f, k); }
            }
        }
        for (int j = 0; j < w; j++) {
            const Taps t = sample_taps(pix, w, i, j);
            int vn[lsq::kMaxN] = {t.a - kMid, t.b - kMid, t.c - kMid, t.d - kMid, t.e - kMid, t.f - kMid, t.t - kMid, t.h - kMid, t.q - kMid, t.g - kMid};
            int b1 = 0, b2 = 0, px0;
            i64 p1 = 0, p2 = 0;
            bool ok1 = false, ok2 = false;
            double *Bj = n ? &B[size_t(j) * m] : nullptr, *Fj = n ? &F[size_t(j) * m] : nullptr;
            if (n) {
                lsq::bias_pair(bias, b1, b2);
                for (int s = 0; s < 2; s++) {
                    const int bs = s ? b2 : b1;
                    double M[lsq::kMaxN][lsq::kMaxN + 1], Mi[lsq::kMaxN][lsq::kMaxN + 1];
                    for (int r = 0; r < n; r++) {
                        for (int c = 0; c < n; c++) M[r][c] = E[1 + n + r * n + c] + Fj[1 + n + r * n + c] + (r == c ? double(bs * n) : 0.0);
                        M[r][n] = E[1 + r] + Fj[1 + r] + double(bs) * double(1 << lsq::kFb3);
                    }
                    memcpy(Mi, M, sizeof M);
                    lsq::Guard g;
                    double pd = 0.0;
                    bool ok = solve_f64(n, M, vn, g, &pd);
                    i64 p = i64(pd);
                    if (!g.ok()) { n_fallback++; ok = solve_int(n, Mi, vn, &p); }
                    p = p < 0 ? 0 : (p > (i64(kMaxVal) << lsq::kFb1) ? (i64(kMaxVal) << lsq::kFb1) : p);
                    if (s) { ok2 = ok; p2 = p; } else { ok1 = ok; p1 = p; }
                }
            }
            if (ok1) px0 = int((p1 + (1 << (lsq::kFb1 - 1))) >> lsq::kFb1);
            else { px0 = predict(t); p1 = i64(px0) << lsq::kFb1; }
            const Level L = quantise(activity(t, err));
            const int adr = context_address(t, L.qu, px0);
            const int v = ctx[adr];
            const int sign = bias_sign(v), px = bias_apply(v, px0);
            Mapper &mp = maps[size_t(px * 2 + sign)];
            const int y = residual_to_symbol(pix(i, j), px, sign, near);
            walk_symbol(k_step, L.qu, L.qv, y < kMapSyms ? int(mp.rank_of[y]) : y, [&](int qu, int qv, int node, int bin) {
                Counter &cu = tree[size_t(qu) * kTreeNodes + node], &cv = tree[size_t(qv) * kTreeNodes + node];
                const int prob = mix_prob(counter_p1(cu.c0, cu.c1), counter_p1(cv.c0, cv.c1), L.qw);
                if (n_bins < cap) coded[n_bins] = pack_coded(prob, bin);
                n_bins++;
                counter_add(cu, bin, kWeightOne - L.qw);
                counter_add(cv, bin, L.qw);
                return bin;
            });
            if (y < kMapSyms) {                               // NBLIC.c:497-523
                const int z = mp.rank_of[y];
                mp.count[z]++;
                if (z > 0 && mp.count[z - 1] < mp.count[z]) {
                    const int other = mp.sym_at[z - 1], c = mp.count[z];
                    mp.count[z] = mp.count[z - 1]; mp.count[z - 1] = c;
                    mp.sym_at[z] = uint8_t(other); mp.sym_at[z - 1] = uint8_t(y);
                    mp.rank_of[y] = uint8_t(z - 1); mp.rank_of[other] = uint8_t(z);
                }
            }
            const int xr = symbol_to_pixel(y, px, sign, near);
            recon[size_t(i) * w + j] = uint8_t(xr);
            err = clip_err(xr, px0);
            ctx[adr] = bias_update(v, err);
            if (n) {                                          // NBLIC.c:882-893, :242-283
                const i64 xq = i64(xr) << lsq::kFb1;
                const double s_curr = double(abs64(p1 - xq));
                const double s_sum = (E[0] + Fj[0]) + floor(s_curr * double(lsq::kDecayS) / double(lsq::kDecayS - 1));
                const double s = lsq::sample_weight(s_sum), rs = lsq::recip_raw(s);
                const int xc = xr - kMid;
                for (int k = 0; k < m; k++) {
                    double sample;
                    if (k == 0) sample = s_curr;
                    else if (k <= n) sample = lsq::sample_entry(xc * vn[k - 1], lsq::kScaleB, s, rs);
                    else { const int r = (k - 1 - n) / n, c = (k - 1 - n) - r * n; sample = lsq::sample_entry(vn[r] * vn[c], lsq::kScaleA, s, rs); }
                    const double b = lsq::decay_k(Bj[k], k) + sample;
                    Bj[k] = b;
                    E[k] = lsq::decay_k(E[k], k) + b;
                }
                if (ok1 && ok2) bias = (abs64(p1 - xq) > abs64(p2 - xq)) ? b2 : b1;
            }
        }
    }
    if (fallbacks) *fallbacks = n_fallback;
    return n_bins;
}

// Exhaustive comparison of the divide-free helpers the serial kernels use (model.h NearParams,
// level_shift_table / walk_symbol_t) with the plain formulations.  Returns the number of mismatches.
extern "C" long hh_check_divide_free(void) {
    long bad = 0;
    for (int near = 0; near <= kMaxNear; near++) {
        const NearParams np = near_params(near);
        for (int num = 0; num < 2048; num++) bad += div_width(num, np) != num / (2 * near + 1);
        for (int px = 0; px < 256; px++)
            for (int sign = 0; sign < 2; sign++) {
                for (int x = 0; x < 256; x++) bad += residual_to_symbol(x, px, sign, np) != residual_to_symbol(x, px, sign, near);
                for (int y = 0; y < 300; y++) bad += symbol_to_pixel(y, px, sign, np) != symbol_to_pixel(y, px, sign, near);
                for (int x = 0; x < 256; x++) bad += reconstruct_pixel(x, px, np) != symbol_to_pixel(residual_to_symbol(x, px, sign, near), px, sign, near);
            }
        const int k_step = k_step_for_near(near);
        const uint64_t ktab = level_shift_table(k_step);
        for (int qu = 0; qu < kLevels; qu++)
            for (int dq = -1; dq <= 1; dq++) {
                const int qv = qu + dq;
                if (qv < 0 || qv >= kLevels) continue;
                for (int z = 0; z < 256; z++) {
                    long sig_a = 0, sig_b = 0;
                    auto rec = [](long &sig) { return [&sig](int a, int b, int node, int bin) { sig = sig * 1000003 + ((a * 16 + b) * 256 + node) * 2 + bin; return bin; }; };
                    const int za = walk_symbol(k_step, qu, qv, z, rec(sig_a));
                    const int zb = walk_symbol_t(k_step, ktab, qu, qv, z, rec(sig_b));
                    bad += za != zb || sig_a != sig_b;
                }
            }
    }
    for (int d = 0; d < 2000; d++) {                              // the activity table of the serial kernels clips at 200
        const Level a = quantise(d), b = quantise(d < 200 ? d : 200);
        bad += a.qu != b.qu || a.qv != b.qv || a.qw != b.qw;
    }
    return bad;
}

// The lane-parallel pixel front (csrc/lane_table.h, serial_engine.hip LaneFront / the QNBLIC decoder's row loop) walked
// on the CPU: for rows >= 2 of random and extreme planes, every lane's term 2X - Y - Z from the table, the quad sums, the
// first-minimum key, the thresholds counted, the extrapolation picked by direction, the comparison mask, the regressor
// bytes -- against model.h's predict / activity / context_address (NBLIC) or predict_q / level_q / context_address_q
// (QNBLIC) on the taps sample_taps / sample_taps_q deliver.  Returns the number of mismatches.
extern "C" long hh_check_lane_front(int qnblic, int seed) {
    static const QLaneTable tab_n = make_lanes(false), tab_q = make_lanes(true);
    const QLaneTable &tab = qnblic ? tab_q : tab_n;
    long bad = 0;
    uint32_t rng = uint32_t(seed) * 2654435761u + 12345u;
    auto rnd = [&]() { rng = rng * 1664525u + 1013904223u; return rng >> 8; };
    const int widths[] = {1, 2, 3, 4, 5, 7, 16, 33, 64};
    for (int wi = 0; wi < 9; wi++) {
        const int w = widths[wi];
        for (int rep = 0; rep < 40; rep++) {
            std::vector<uint8_t> img(size_t(4) * w);
            const int kind = rep % 4;                                    // noise, extremes, smooth, flat
            for (size_t k = 0; k < img.size(); k++)
                img[k] = kind == 0 ? uint8_t(rnd()) : kind == 1 ? uint8_t((rnd() & 1) ? 255 : 0) : kind == 2 ? uint8_t(100 + (rnd() % 9)) : uint8_t(200);
            auto pix = [&](int r, int c) { return int(img[size_t(r) * w + c]); };
            for (int i = 2; i < 4; i++) {
                const uint8_t *r0 = &img[size_t(i) * w], *r1 = &img[size_t(i - 1) * w], *r2 = &img[size_t(i - 2) * w];
                auto cl = [w](int c) { return c < 0 ? 0 : (c >= w ? w - 1 : c); };
                int a = r1[0], e = a;                                    // the front's registers at column 0 (either codec)
                for (int j = 0; j < w; j++) {
                    const Taps t = qnblic ? sample_taps_q(pix, w, i, j) : sample_taps(pix, w, i, j);
                    const int err = int(rnd() % 511) - 255;
                    int V[64], T[64], Q[64];
                    for (int k = 0; k < 64; k++) {
                        const QLaneConst &c = tab.l[k];
                        int op[3];
                        for (int o = 0; o < 3; o++) op[o] = c.sel[o] == 0 ? 0 : (c.sel[o] == 1 ? r1 : r2)[cl(j + c.dx[o])];
                        V[k] = 2 * op[0] + a * c.a2 - (op[1] + op[2] + e * c.ce);
                        T[k] = V[k] < 0 ? -V[k] : V[k];
                    }
                    for (int k = 0; k < 64; k++) { const int q = k & ~3; Q[k] = T[q] + T[q + 1] + T[q + 2] + T[q + 3]; }
                    // activity
                    const int act = ((Q[28] + Q[32]) >> 1) + 2 * (err < 0 ? -err : err);
                    bad += act != activity(t, err);
                    // predictor
                    unsigned key = 0xFFFFFFFFu; int total = 0;
                    for (int k = 0; k < 64; k++) {
                        const unsigned kk = unsigned((Q[k] << 3) | tab.l[k].key_or);
                        if (kk < key) key = kk;
                        if ((k & 3) == 0) total += Q[k] & tab.l[k].sum_and;
                    }
                    const int best = int(key >> 3), dir = int(key & 7);
                    const int spread = qnblic ? (total - 7 * best) >> 3 : total - 7 * best;
                    int wt = 0;
                    for (int k = 0; k < 64; k++) wt += tab.l[k].thr_weight <= spread;
                    const QLaneConst &cd = tab.l[dir];
                    const int ang = a * cd.ca + t.b * cd.cb + t.c * cd.cc + t.d * cd.cd;
                    const int lin = iclip(9 * (a + t.b) + 2 * (t.d - t.c) - e - t.f, 0, 16 * kMaxVal);
                    const int px0 = (8 * wt * ang + (8 - wt) * lin + 64) >> 7;
                    bad += px0 != (qnblic ? predict_q(t) : predict(t));
                    bad += a != t.a || e != t.e;
                    // context address: the comparison mask of lanes 36..43
                    int mask = 0;
                    for (int k = 36; k < 44; k++) mask |= int((px0 << tab.l[k].sh) > V[k]) << (k - 36);
                    if (qnblic) {
                        int qd = 0;
                        for (int k = 0; k < 64; k++) qd += tab.l[k].thr_level <= act;
                        bad += qd != level_q(t, err);
                        bad += ((qd << 8) | mask) != context_address_q(t, qd, px0);
                    } else {
                        const int qu = int(rnd() % kLevels);
                        bad += ((((qu >> 1) << 8)) | mask) != context_address(t, qu, px0);
                        const int want[10] = {t.a, t.b, t.c, t.d, t.e, t.f, t.t, t.h, t.q, t.g};      // NBLIC.c:164-183
                        for (int k = 44; k < 54; k++) bad += tab.l[k].dst != k - 44 || (V[k] >> 1) != want[k - 44];
                    }
                    // the row being coded moves on (LaneFront::advance / the QNBLIC loop's e = a; a = pixel)
                    const int x = r0[j];
                    e = (qnblic || j >= 1) ? a : x;
                    a = x;
                }
            }
        }
    }
    return bad;
}

// The decoders' lane layout of a symbol's bins (serial_engine.hip decode_symbol) against the walk itself: for every
// k_step, level pair and symbol, the nodes walk_symbol visits must be node t << k_max for the t-th prefix bin (while the
// prefix stays inside the level's tree and the lanes: afterwards the kernel walks bin by bin like the reference) and, for
// the suffix, root + suffix_lane_offset(k, d, prefix) with the lane moving 2 l + 1 + bin in heap order.
extern "C" long hh_check_symbol_lanes(void) {
    long bad = 0, checked = 0;
    for (int near = 0; near <= kMaxNear; near++) {
        const int k_step = k_step_for_near(near), k_max = (kLevels - 1) / k_step;
        const int reach = (kTreeNodes >> k_max) < 64 ? (kTreeNodes >> k_max) : 64;
        for (int qu = 0; qu < kLevels; qu++)
            for (int dq = -1; dq <= 1; dq++) {
                const int qv = qu + dq;
                if (qv < 0 || qv >= kLevels) continue;
                for (int z = 0; z < 400; z++) {
                    std::vector<int> nodes, bins;
                    walk_symbol(k_step, qu, qv, z, [&](int, int, int node, int bin) { nodes.push_back(node); bins.push_back(bin); return bin; });
                    // prefix: bins up to and including the first zero
                    size_t n_prefix = 0;
                    while (bins[n_prefix]) n_prefix++;
                    n_prefix++;
                    for (size_t t = 0; t < n_prefix && int(t) < reach; t++) { bad += nodes[t] != int(t) << k_max; checked++; }
                    // suffix: heap order below the node after the prefix's last
                    const int k = int(nodes.size() - n_prefix), root = nodes[n_prefix - 1] + 1;
                    int lane = 0;
                    for (int s = 0; s < k; s++) {
                        int d = 0;
                        while ((2 << d) <= lane + 1) d++;
                        const int prefix = lane + 1 - (1 << d);
                        bad += d != s || lane >= 64 || nodes[n_prefix + s] != root + suffix_lane_offset(k, d, prefix);
                        checked++;
                        lane = 2 * lane + 1 + bins[n_prefix + s];
                    }
                }
            }
    }
    return checked > 100000 ? bad : -1;
}

/*
 * oracle/qnblic_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of QNBLIC, the effort-0 lossless codec of NBLIC v0.3
 * (/root/reference/src/QNBLIC.c): same predictor family as NBLIC but with a
 * shift-register neighbourhood, 12 hard activity levels, 3072 contexts, and a
 * two-pass entropy stage -- per-level symbol histograms normalised to 2^15,
 * a compact histogram code, and one 32-bit rANS stream coded last pixel first.
 *
 * Parity status: PINNED against oracle/_ref/libnblic_ref.so and the q_* streams in
 * tests/golden/ (tests/test_oracle_q.py).
 *
 * Two formulations of the neighbourhood live here on purpose:
 *   qn_window   the reference's running 11-tap window (QNBLIC.c:48-79), used by the fused codec;
 *   orc_q_taps  a closed form per pixel, which is what a stateless GPU
 *               kernel needs; tests check the two against each other on every small shape.
 */
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

enum { Q_MAXVAL = 255, Q_MID = 128, Q_LEVELS = 12, Q_NCTX = Q_LEVELS * 256, Q_CTX_COEF = 7, Q_CTX_SCALE = 11,
       Q_NORM_BITS = 15, Q_NORM_SUM = 1 << Q_NORM_BITS, Q_ANS_BITS = 16 };

static inline int q_iabs(int v) { return v < 0 ? -v : v; }
static inline int q_clip(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int q_min(int a, int b) { return a < b ? a : b; }
static inline int q_asr(int v, int s) { return v >= 0 ? (v >> s) : -(((-v) + (1 << s) - 1) >> s); }

typedef struct { int a, b, c, d, e, f, g, h, q, r, s; } qn_taps;

/* ---- neighbourhood, window form (QNBLIC.c:48-79) ------------------------ */
static int qpix(const uint8_t *img, int w, int i, int j, int dflt) {
    return (i >= 0 && j >= 0 && j < w) ? img[(size_t)i * w + j] : dflt;
}
static void qn_window_start(const uint8_t *img, int w, int i, qn_taps *n) {       /* column 0 of row i */
    int a = qpix(img, w, i, -1, Q_MID), b = qpix(img, w, i - 1, 0, Q_MID);
    if (i == 0) b = a; else a = b;                                                  /* j == 0 here */
    n->a = a; n->b = b;
    n->e = qpix(img, w, i, -2, a);
    n->c = qpix(img, w, i - 1, -1, b);
    n->d = qpix(img, w, i - 1, 1, b);
    n->f = qpix(img, w, i - 2, 0, b);
    n->g = qpix(img, w, i - 2, 1, n->f);
    n->h = qpix(img, w, i - 2, -1, n->f);
    n->q = qpix(img, w, i - 1, -2, n->c);
    n->r = qpix(img, w, i - 2, 2, n->g);
    n->s = qpix(img, w, i - 2, -2, n->h);
}
static void qn_window_advance(const uint8_t *img, int w, int i, int j, int x, qn_taps *n) {   /* after pixel (i,j) */
    n->e = n->a; n->a = x;
    n->q = n->c; n->c = n->b; n->b = n->d;
    n->s = n->h; n->h = n->f; n->f = n->g; n->g = n->r;
    if (i <= 0) n->d = n->a; else if (j + 2 < w) n->d = img[(size_t)(i - 1) * w + j + 2];
    if (i <= 1) n->r = n->d; else if (j + 3 < w) n->r = img[(size_t)(i - 2) * w + j + 3];
}

/* ---- neighbourhood, closed form ------------------- */
static int clampc(int j, int w) { return j < 0 ? 0 : (j >= w ? w - 1 : j); }
void orc_q_taps(const uint8_t *img, int w, int i, int j, int *out11) {
    qn_taps n;
    if (i == 0) {
#define X0(k) ((k) >= 0 ? (int)img[(k)] : Q_MID)
        n.a = X0(j - 1); n.e = X0(j - 2); n.d = X0(j - 1); n.b = X0(j - 2); n.c = X0(j - 3); n.q = X0(j - 4);
        n.r = X0(j - 1); n.g = X0(j - 2); n.f = X0(j - 3); n.h = X0(j - 4); n.s = X0(j - 5);
#undef X0
    } else {
        const uint8_t *U = img + (size_t)(i - 1) * w, *X = img + (size_t)i * w;
        n.b = U[clampc(j, w)]; n.c = U[clampc(j - 1, w)]; n.q = U[clampc(j - 2, w)]; n.d = U[clampc(j + 1, w)];
        n.a = j >= 1 ? X[j - 1] : U[0];
        n.e = j >= 2 ? X[j - 2] : U[0];                    /* also U(0) at j == 1: the window hands on a(0) */
        if (i == 1) {
            n.r = j >= 1 ? U[clampc(j + 1, w)] : U[0];
            n.g = j >= 2 ? U[clampc(j, w)] : U[0];
            n.f = j >= 3 ? U[j - 1] : U[0];
            n.h = j >= 4 ? U[j - 2] : U[0];
            n.s = j >= 5 ? U[j - 3] : U[0];
        } else {
            const uint8_t *V = img + (size_t)(i - 2) * w;
            n.f = V[clampc(j, w)]; n.h = V[clampc(j - 1, w)]; n.s = V[clampc(j - 2, w)];
            n.g = V[clampc(j + 1, w)]; n.r = V[clampc(j + 2, w)];
        }
    }
    out11[0] = n.a; out11[1] = n.b; out11[2] = n.c; out11[3] = n.d; out11[4] = n.e; out11[5] = n.f;
    out11[6] = n.g; out11[7] = n.h; out11[8] = n.q; out11[9] = n.r; out11[10] = n.s;
}
/* the window's taps at every pixel, for comparing the two formulations */
void orc_q_taps_window(const uint8_t *img, int h, int w, int *out /* h*w*11 */) {
    for (int i = 0; i < h; i++) {
        qn_taps n; qn_window_start(img, w, i, &n);
        for (int j = 0; j < w; j++) {
            int *o = out + ((size_t)i * w + j) * 11;
            o[0] = n.a; o[1] = n.b; o[2] = n.c; o[3] = n.d; o[4] = n.e; o[5] = n.f; o[6] = n.g; o[7] = n.h; o[8] = n.q; o[9] = n.r; o[10] = n.s;
            qn_window_advance(img, w, i, j, img[(size_t)i * w + j], &n);
        }
    }
}

/* ---- predictor (QNBLIC.c:94-149): NBLIC's seven directions, blend weight 0..7 -------------- */
static int q_predict(const qn_taps *n) {
    const int P[4][5] = { { n->a, n->e, n->q, n->c, n->b }, { n->c, n->q, n->s, n->h, n->f },
                          { n->b, n->c, n->h, n->f, n->g }, { n->d, n->b, n->f, n->g, n->r } };
    const int cur[4] = { n->a, n->c, n->b, n->d };
    static const int du[7] = { 1, 3, 2, 4, 1, 2, 3 }, dv[7] = { 1, 3, 2, 4, 2, 3, 4 };
    static const int thr[7] = { 5, 12, 34, 78, 194, 431, 601 };
    int lin = q_clip(9 * n->a + 9 * n->b + 2 * n->d - 2 * n->c - n->e - n->f, 0, 16 * Q_MAXVAL);
    int best = 0, ang = 0, sum = 0;
    for (int k = 0; k < 7; k++) {
        int cost = 0;
        for (int p = 0; p < 4; p++) cost += q_iabs(2 * P[p][0] - P[p][du[k]] - P[p][dv[k]]);
        sum += cost;
        if (k == 0 || cost < best) { best = cost; ang = cur[du[k] - 1] + cur[dv[k] - 1]; }
    }
    int v = q_min((sum - 7 * best) >> 3, 607), wt = 0;
    while (wt < 7 && thr[wt] <= v) wt++;
    return (8 * wt * ang + (8 - wt) * lin + 64) >> 7;
}

/* ---- hard activity level (QNBLIC.c:152-161, :599-601): err is the UNCLIPPED x - px0 of the left pixel */
static int q_level(const qn_taps *n, int err_prev) {
    static const int thr[Q_LEVELS - 1] = { 1, 2, 4, 6, 9, 15, 25, 39, 63, 101, 151 };
    int delta = q_iabs(n->a - n->e) + q_iabs(n->b - n->c) + q_iabs(n->b - n->d) + q_iabs(n->a - n->c) +
                q_iabs(n->b - n->f) + q_iabs(n->d - n->g) + 2 * q_iabs(err_prev);
    int v = q_min(delta, 151), qd = 0;
    while (qd < Q_LEVELS - 1 && thr[qd] <= v) qd++;
    return qd;
}

/* ---- context address (QNBLIC.c:164-173): level in the high bits, comparisons MSB-first */
static int q_ctx_addr(const qn_taps *n, int qd, int px0) {
    return (qd << 8) | ((px0 > n->a) << 7) | ((px0 > n->b) << 6) | ((px0 > n->c) << 5) | ((px0 > n->d) << 4) |
           ((px0 > n->e) << 3) | ((px0 > n->f) << 2) | ((px0 > 2 * n->a - n->e) << 1) | (px0 > 2 * n->b - n->f);
}
static int q_ctx_correct(int v, int px0, int *sign) {                    /* QNBLIC.c:176-180 */
    *sign = q_asr(v, Q_CTX_SCALE - 1) & 1;
    return q_clip(px0 + q_asr(v, Q_CTX_SCALE) + *sign, 0, Q_MAXVAL);
}
static int q_ctx_update(int v, int err) {                                /* QNBLIC.c:183-188: rounding constant 63 */
    return q_asr(v * 127 + err * (1 << Q_CTX_SCALE) + 63, Q_CTX_COEF);
}
static int q_x_to_y(int x, int px, int sign) {                           /* QNBLIC.c:191-202 */
    int ty = q_min(px, Q_MAXVAL - px), y = q_iabs(x - px);
    if (y <= 0) return 0;
    if (y <= ty) return 2 * y - ((x >= px) ^ sign);
    return y + ty;
}
static int q_y_to_x(int y, int px, int sign) {                           /* QNBLIC.c:205-217 */
    int ty = q_min(px, Q_MAXVAL - px);
    if (y <= 0) return px;
    if (y <= 2 * ty) { int m = (y + 1) >> 1; return ((y & 1) ^ sign) ? px + m : px - m; }
    return px < Q_MID ? px + (y - ty) : px - (y - ty);
}

/* ---- histogram normalisation to sum 2^15 (QNBLIC.c:308-358).  The only floating point on the
 * path: IEEE double, no contraction (oracle/Makefile passes -ffp-contract=off), truncating cast. */
void orc_q_norm_hist(uint32_t hist[256]) {
    uint32_t sum = 0, nz = 0, last = 0;
    for (uint32_t i = 0; i < 256; i++) if (hist[i] > 0) { sum += hist[i]; nz++; last = i; }
    if (nz == 0) { hist[0] = Q_NORM_SUM - 1; hist[1] = 1; return; }
    if (nz == 1) { hist[last] = Q_NORM_SUM - 1; hist[(last + 1) % 256] = 1; return; }
    double scale = (1.0 * Q_NORM_SUM) / sum;
    sum = 0;
    for (uint32_t i = 0; i < 256; i++) if (hist[i] > 0) {
        uint32_t v = (uint32_t)(0.49 + scale * hist[i]);
        hist[i] = v > 1 ? v : 1;
        sum += hist[i];
    }
    for (uint32_t i = 0; sum > Q_NORM_SUM; i = (i + 1) % 256) if (hist[i] > 1) { hist[i]--; sum--; }
    for (uint32_t i = 0; sum < Q_NORM_SUM; i = (i + 1) % 256) if (hist[i] > 0) { hist[i]++; sum++; }
}

/* ---- histogram code (QNBLIC.c:362-459) ---------------------------------- */
static uint16_t *q_put_hist(uint16_t *p, const uint32_t hist[256]) {
    uint32_t i = 0, sum = 0;
    while (i < 256 && sum < Q_NORM_SUM) {
        uint32_t h0 = hist[i], j = i + 1, he = 0xFFFF, code;
        for (; j < 256; j++) { he = hist[j] & 0xFFFF; if (he != (h0 & 0xFFFF)) break; }
        uint32_t run = j - i;
        if (h0 <= 1 && run >= 4) {                            /* run of 0s or 1s, optionally followed by one small value */
            if (j < 256 && he <= 15) j++; else he = h0;
            code = (7u << 13) | (h0 << 12) | (he << 8) | (run - 4);
        } else {
            uint32_t h1 = i + 1 < 256 ? (hist[i + 1] & 0xFFFF) : 0xFFFF, h2 = i + 2 < 256 ? (hist[i + 2] & 0xFFFF) : 0xFFFF,
                     h3 = i + 3 < 256 ? (hist[i + 3] & 0xFFFF) : 0xFFFF;
            h0 &= 0xFFFF;
            if (h0 <= 7 && h1 <= 7 && h2 <= 7 && h3 <= 7)  { code = (13u << 12) | (h0 << 9) | (h1 << 6) | (h2 << 3) | h3; j = i + 4; }
            else if (h0 <= 15 && h1 <= 15 && h2 <= 15)     { code = (12u << 12) | (h0 << 8) | (h1 << 4) | h2;            j = i + 3; }
            else if (h0 <= 127 && h1 <= 127)               { code = (2u << 14) | (h0 << 7) | h1;                         j = i + 2; }
            else                                           { code = h0;                                                   j = i + 1; }
        }
        *p++ = (uint16_t)code;
        for (; i < j; i++) sum += hist[i];
    }
    return p;
}
static const uint16_t *q_get_hist(const uint16_t *p, uint32_t hist[256]) {
    uint32_t i = 0, sum = 0;
    memset(hist, 0, 256 * sizeof(uint32_t));
    while (i < 256 && sum < Q_NORM_SUM) {
        uint32_t code = *p++;
        if ((code >> 15) == 0)       { sum += (hist[i++] = code); }
        else if ((code >> 14) == 2)  { sum += (hist[i++] = (code >> 7) & 0x7F); sum += (hist[i++] = code & 0x7F); }
        else if ((code >> 12) == 12) { sum += (hist[i++] = (code >> 8) & 15); sum += (hist[i++] = (code >> 4) & 15); sum += (hist[i++] = code & 15); }
        else if ((code >> 12) == 13) { sum += (hist[i++] = (code >> 9) & 7); sum += (hist[i++] = (code >> 6) & 7); sum += (hist[i++] = (code >> 3) & 7); sum += (hist[i++] = code & 7); }
        else {
            uint32_t run = (code & 0xFF) + 4, he = (code >> 8) & 15, h0 = (code >> 12) & 1;
            for (; run > 0; run--) sum += (hist[i++] = h0);
            if (he != h0) sum += (hist[i++] = he);
        }
    }
    return p;
}

/* ---- rANS, 32-bit state, 16-bit renormalisation (QNBLIC.c:238-274) ------ */
static inline uint16_t *q_ans_put(uint32_t *state, uint16_t *p, uint32_t freq, uint32_t start) {
    uint32_t q = *state / freq;
    if (q > (1u << (2 * Q_ANS_BITS - Q_NORM_BITS)) - 1) { *p++ = (uint16_t)*state; *state >>= Q_ANS_BITS; q = *state / freq; }
    *state = (*state % freq) + (q << Q_NORM_BITS) + start;
    return p;
}
/* exported on its own: the serial tail of the encoder over precomputed (level, symbol) pairs */
long orc_q_entropy_stage(uint16_t *out, int h, int w, const uint8_t *qd, const uint8_t *y, const uint32_t hist_in[Q_LEVELS][256]) {
    static const char title[4] = { 'Q', '0', '.', '2' };
    uint32_t hist[Q_LEVELS][256], acc[Q_LEVELS][256];
    uint16_t *p = out;
    *p++ = (uint16_t)(title[0] | (title[1] << 8)); *p++ = (uint16_t)(title[2] | (title[3] << 8));      /* QNBLIC.c:463-473 */
    *p++ = (uint16_t)h; *p++ = (uint16_t)w;
    memcpy(hist, hist_in, sizeof hist);
    for (int k = 0; k < Q_LEVELS; k++) {
        orc_q_norm_hist(hist[k]);
        acc[k][0] = 0;
        for (int s = 1; s < 256; s++) acc[k][s] = acc[k][s - 1] + hist[k][s - 1];
        p = q_put_hist(p, hist[k]);
    }
    uint16_t *body = p;
    uint32_t state = 1u << Q_ANS_BITS;
    for (size_t t = (size_t)h * w; t-- > 0;) p = q_ans_put(&state, p, hist[qd[t]][y[t]], acc[qd[t]][y[t]]);
    *p++ = (uint16_t)state; *p++ = (uint16_t)(state >> Q_ANS_BITS);
    for (uint16_t *lo = body, *hi = p - 1; lo < hi; lo++, hi--) { uint16_t t = *lo; *lo = *hi; *hi = t; }
    return (long)(p - out);
}

static int q_size_ok(int h, int w, long max_px) { return h > 0 && w > 0 && h <= 65535 && w <= 65535 && (long)h * w <= max_px; }

/* stage 1 for the GPU parity tests: everything per pixel that is computed before the entropy stage */
void orc_q_model(const uint8_t *img, int h, int w, uint8_t *px0_out, uint16_t *adr_out, uint8_t *qd_out, uint8_t *y_out,
                 uint32_t hist[Q_LEVELS][256]) {
    int *ctx = (int *)calloc(Q_NCTX, sizeof(int));
    if (hist) memset(hist, 0, sizeof(uint32_t) * Q_LEVELS * 256);
    for (int i = 0; i < h; i++) {
        qn_taps n; qn_window_start(img, w, i, &n);
        int err = 0;
        for (int j = 0; j < w; j++) {
            size_t t = (size_t)i * w + j;
            int x = img[t], px0 = q_predict(&n), qd = q_level(&n, err), sign;
            err = x - px0;
            int adr = q_ctx_addr(&n, qd, px0);
            int px = q_ctx_correct(ctx[adr], px0, &sign);
            int y = q_x_to_y(x, px, sign);
            ctx[adr] = q_ctx_update(ctx[adr], err);
            if (px0_out) px0_out[t] = (uint8_t)px0;
            if (adr_out) adr_out[t] = (uint16_t)adr;
            if (qd_out) qd_out[t] = (uint8_t)qd;
            if (y_out) y_out[t] = (uint8_t)y;
            if (hist) hist[qd][y]++;
            qn_window_advance(img, w, i, j, x, &n);
        }
    }
    free(ctx);
}

/* QNBLICcompress (QNBLIC.c:562-655): returns the length in 16-bit words, or -1 */
long orc_qnblic_encode(uint16_t *out, const uint8_t *img, int h, int w, long max_px) {
    if (max_px <= 0) max_px = 100000000L;
    if (!q_size_ok(h, w, max_px)) return -1;
    size_t n = (size_t)h * w;
    uint8_t *qd = (uint8_t *)malloc(n), *y = (uint8_t *)malloc(n);
    uint32_t (*hist)[256] = (uint32_t (*)[256])malloc(sizeof(uint32_t) * Q_LEVELS * 256);
    if (!qd || !y || !hist) { free(qd); free(y); free(hist); return -1; }
    orc_q_model(img, h, w, NULL, NULL, qd, y, hist);
    long words = orc_q_entropy_stage(out, h, w, qd, y, (const uint32_t (*)[256])hist);
    free(qd); free(y); free(hist);
    return words;
}

/* QNBLICdecompress (QNBLIC.c:493-555): 0 / -1 */
long orc_qnblic_decode(const uint16_t *in, uint8_t *img, int *h, int *w, long max_px) {
    if (max_px <= 0) max_px = 100000000L;
    if (in[0] != (uint16_t)('Q' | ('0' << 8)) || in[1] != (uint16_t)('.' | ('2' << 8))) return -1;
    *h = in[2]; *w = in[3];
    if (!q_size_ok(*h, *w, max_px)) return -1;
    const uint16_t *p = in + 4;
    uint32_t (*hist)[256] = (uint32_t (*)[256])malloc(sizeof(uint32_t) * Q_LEVELS * 256);
    uint32_t (*acc)[256] = (uint32_t (*)[256])malloc(sizeof(uint32_t) * Q_LEVELS * 256);
    uint8_t *slot = (uint8_t *)malloc((size_t)Q_LEVELS * Q_NORM_SUM);                /* slot -> symbol, per level */
    int *ctx = (int *)calloc(Q_NCTX, sizeof(int));
    for (int k = 0; k < Q_LEVELS; k++) {
        p = q_get_hist(p, hist[k]);
        acc[k][0] = 0;
        for (int s = 1; s < 256; s++) acc[k][s] = acc[k][s - 1] + hist[k][s - 1];
        uint8_t *tab = slot + (size_t)k * Q_NORM_SUM;
        for (uint32_t s = 0; s < 255; s++) for (uint32_t i = acc[k][s]; i < acc[k][s + 1] && i < Q_NORM_SUM; i++) tab[i] = (uint8_t)s;
        for (uint32_t i = acc[k][255]; i < Q_NORM_SUM; i++) tab[i] = 255;
    }
    uint32_t state = (uint32_t)(*p++) << Q_ANS_BITS; state |= *p++;
    for (int i = 0; i < *h; i++) {
        qn_taps n; qn_window_start(img, *w, i, &n);
        int err = 0;
        for (int j = 0; j < *w; j++) {
            int px0 = q_predict(&n), qd = q_level(&n, err), sign;
            int adr = q_ctx_addr(&n, qd, px0);
            int px = q_ctx_correct(ctx[adr], px0, &sign);
            uint32_t low = state & (Q_NORM_SUM - 1);
            int y = slot[(size_t)qd * Q_NORM_SUM + low];
            state = (state >> Q_NORM_BITS) * hist[qd][y] + low - acc[qd][y];
            if (state < (1u << Q_ANS_BITS)) { state = (state << Q_ANS_BITS) | *p++; }
            int x = q_y_to_x(y, px, sign);
            img[(size_t)i * *w + j] = (uint8_t)x;
            err = x - px0;
            ctx[adr] = q_ctx_update(ctx[adr], err);
            qn_window_advance(img, *w, i, j, x, &n);
        }
    }
    free(hist); free(acc); free(slot); free(ctx);
    return 0;
}

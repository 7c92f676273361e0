/*
 * oracle/nblic_model.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the NBLIC v0.3 per-pixel model (predictor, activity
 * quantiser, context address, context bias, residual mapping, adaptive
 * re-mapper, adaptive binary counters, Golomb-like binarisation and the
 * 32-bit binary range coder).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may use anything under oracle/.
 *
 * Parity status: PINNED.  Every function here is exercised through
 * oracle/nblic_oracle.c, whose byte output is compared against the compiled
 * reference (oracle/_ref/libnblic_ref.so, built by oracle/Makefile from
 * /root/reference/src) and against the committed fixtures in tests/golden/.
 *
 * Each helper cites the reference lines (relative to /root/reference/src/)
 * whose integers it must reproduce.  The formulation is our own: the seven
 * directional costs are written over a (W, NW, N, NE) neighbourhood table,
 * state lives in explicit structs, and all arithmetic on possibly-negative
 * values is spelled with well-defined operations.
 */
#ifndef NBLIC_ORACLE_MODEL_H
#define NBLIC_ORACLE_MODEL_H

#include <stdint.h>
#include <stddef.h>

/* ---- bitstream-defining constants (NBLIC.c:45-90) ---------------------- */
enum {
    NB_MAXVAL      = 255,
    NB_MID         = 128,
    NB_MAX_NEAR    = 9,        /* 255/26 */
    NB_MIN_KSTEP   = 3,
    NB_NQD         = 16,       /* activity levels = counter trees */
    NB_NCTX        = 2048,     /* (NQD/2)*256 */
    NB_CTX_COEF    = 7,
    NB_CTX_SCALE   = 8,
    NB_NQW         = 32,       /* interpolation weight denominator */
    NB_NMAP        = 20,       /* symbols handled by the re-mapper */
    NB_MAXCOUNT    = 256,      /* counters halve above NQW*MAXCOUNT */
    NB_PROB_ONE    = 4096,
    NB_TREE        = 256       /* nodes per counter tree */
};

static inline int nb_iabs(int v)            { return v < 0 ? -v : v; }
static inline int nb_clip(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
/* arithmetic shift right of a possibly negative int == floor division */
static inline int nb_floor_shr(int v, int s) {
    return v >= 0 ? (v >> s) : -(((-v) + (1 << s) - 1) >> s);
}

/* ---- causal neighbourhood (NBLIC.c:287-304) ---------------------------- */
typedef struct { int a, b, c, d, e, f, g, h, q, r, s, t; } nb_taps;

static inline int nb_pix_or(const uint8_t *img, int w, int i, int j, int dflt) {
    return (i >= 0 && j >= 0 && j < w) ? (int)img[(size_t)i * (size_t)w + (size_t)j] : dflt;
}

/* Chained fall-backs: every missing tap inherits from a nearer one. */
static inline void nb_sample(const uint8_t *img, int w, int i, int j, nb_taps *n) {
    int a = nb_pix_or(img, w, i, j - 1, NB_MID);
    int b = nb_pix_or(img, w, i - 1, j, NB_MID);
    if (i == 0)      b = a;
    else if (j == 0) a = b;
    n->a = a; n->b = b;
    n->e = nb_pix_or(img, w, i,     j - 2, a);
    n->c = nb_pix_or(img, w, i - 1, j - 1, b);
    n->d = nb_pix_or(img, w, i - 1, j + 1, b);
    n->f = nb_pix_or(img, w, i - 2, j,     b);
    n->g = nb_pix_or(img, w, i - 2, j + 1, n->f);
    n->h = nb_pix_or(img, w, i - 2, j - 1, n->f);
    n->q = nb_pix_or(img, w, i - 1, j - 2, n->c);
    n->r = nb_pix_or(img, w, i - 2, j + 2, n->g);
    n->s = nb_pix_or(img, w, i - 2, j - 2, n->h);
    n->t = nb_pix_or(img, w, i - 1, j + 2, n->d);
}

/* ---- 7-direction blended predictor (NBLIC.c:307-370) -------------------
 * Four already-coded "probe" pixels P in {a, c, b, d} each have a west,
 * north-west, north and north-east neighbour.  A direction's cost is the
 * summed prediction error it would have made at the four probes; the
 * direction order (W, N, NW, NE, W+NW, NW+N, N+NE) and the strict '<' keep
 * the reference's first-minimum tie rule.  `thr` lets QNBLIC reuse this with
 * its own blend thresholds.                                                */
static inline int nb_predict_dirs(const nb_taps *n, int *csum_out, int *pxang_out) {
    /* rows: probe, W, NW, N, NE */
    const int P[4][5] = {
        { n->a, n->e, n->q, n->c, n->b },
        { n->c, n->q, n->s, n->h, n->f },
        { n->b, n->c, n->h, n->f, n->g },
        { n->d, n->b, n->f, n->g, n->r },
    };
    /* prediction candidates for the current pixel: its W,NW,N,NE = a,c,b,d */
    const int cur[4] = { n->a, n->c, n->b, n->d };     /* W, NW, N, NE */
    /* direction k uses neighbour columns (u,v): single dirs have u==v */
    static const int du[7] = { 1, 3, 2, 4, 1, 2, 3 };
    static const int dv[7] = { 1, 3, 2, 4, 2, 3, 4 };
    int best = 0xFFFFFF, ang = 0, sum = 0;
    for (int k = 0; k < 7; k++) {
        int cost = 0;
        for (int p = 0; p < 4; p++)
            cost += nb_iabs(2 * P[p][0] - P[p][du[k]] - P[p][dv[k]]);
        sum += cost;
        if (cost < best) { best = cost; ang = cur[du[k] - 1] + cur[dv[k] - 1]; }
    }
    *csum_out  = sum - 7 * best;
    *pxang_out = ang;                                  /* 2x scale */
    return 0;
}

static inline int nb_predict(const nb_taps *n) {
    static const int thr[8] = { 31, 93, 279, 620, 1550, 3410, 9300, 24800 };
    int lin = nb_clip(9 * n->a + 9 * n->b + 2 * n->d - 2 * n->c - n->e - n->f, 0, 16 * NB_MAXVAL);
    int csum, ang, wt = 0;
    nb_predict_dirs(n, &csum, &ang);
    while (wt < 8 && thr[wt] <= csum) wt++;
    return (8 * wt * ang + (8 - wt) * lin + 64) >> 7;
}

/* ---- activity -> two adjacent levels + weight (NBLIC.c:373-395) -------- */
static inline int nb_delta(const nb_taps *n, int err_prev) {
    return nb_iabs(n->a - n->e) + nb_iabs(n->b - n->c) + nb_iabs(n->b - n->d) +
           nb_iabs(n->a - n->c) + nb_iabs(n->b - n->f) + nb_iabs(n->d - n->g) +
           2 * nb_iabs(err_prev);
}

static inline void nb_quantise(int delta, int *qu, int *qv, int *qw) {
    static const int mid[NB_NQD] = { 0, 2, 4, 7, 10, 14, 20, 26, 34, 42, 52, 64, 78, 95, 135, 200 };
    int qd = 0;
    while (qd < NB_NQD - 1 && delta > mid[qd]) qd++;
    *qu = *qv = qd; *qw = 0;
    if (delta < mid[qd]) {                 /* strictly between two centres */
        int w = NB_NQW * (delta - mid[qd - 1]) / (mid[qd] - mid[qd - 1]);
        if (w < NB_NQW / 2) { *qu = qd - 1; *qw = w; }
        else                { *qv = qd - 1; *qw = NB_NQW - w; }
    }
}

/* ---- context address (NBLIC.c:398-410) --------------------------------- */
static inline int nb_ctx_addr(const nb_taps *n, int qu, int px0) {
    return ((qu >> 1) << 8)
         | (px0 > n->a)               | ((px0 > n->b) << 1)
         | ((px0 > n->c) << 2)        | ((px0 > n->d) << 3)
         | ((px0 > n->e) << 4)        | ((px0 > n->f) << 5)
         | ((px0 > 2 * n->a - n->e) << 6) | ((px0 > 2 * n->b - n->f) << 7);
}

/* ---- context bias state (NBLIC.c:413-428) ------------------------------ */
static inline int nb_ctx_correct(int v, int px0, int *sign) {
    *sign = nb_floor_shr(v, NB_CTX_SCALE - 1) & 1;
    return nb_clip(px0 + nb_floor_shr(v, NB_CTX_SCALE) + *sign, 0, NB_MAXVAL);
}
static inline int nb_ctx_update(int v, int err) {
    return nb_floor_shr(v * ((1 << NB_CTX_COEF) - 1) + err * (1 << NB_CTX_SCALE) + (1 << (NB_CTX_COEF - 1)),
                        NB_CTX_COEF);
}

/* ---- residual <-> symbol (NBLIC.c:431-466) ----------------------------- */
static inline int nb_fold_limit(int px, int near) {
    int m = px < NB_MAXVAL - px ? px : NB_MAXVAL - px;     /* CLIP(px,0,255-px) for px in 0..255 */
    if (m < 0) m = 0;
    return (m + near) / (2 * near + 1);
}
static inline int nb_x_to_y(int x, int px, int sign, int near) {
    int ty = nb_fold_limit(px, near);
    int up = x >= px;
    int y  = (nb_iabs(x - px) + near) / (2 * near + 1);
    if (y <= 0)  return 0;
    if (y <= ty) return 2 * y - (up ^ sign);
    return y + ty;
}
static inline int nb_y_to_x(int y, int px, int sign, int near) {
    int ty = nb_fold_limit(px, near);
    int mag, up;
    if (y <= 0)            { mag = 0;           up = 0; }
    else if (y <= 2 * ty)  { mag = (y + 1) / 2; up = (y & 1) ^ sign; }
    else                   { mag = y - ty;      up = px < NB_MID; }
    mag *= 2 * near + 1;
    return nb_clip(up ? px + mag : px - mag, 0, NB_MAXVAL);
}

/* ---- adaptive symbol re-mapper (NBLIC.c:470-523) -----------------------
 * A rank list of the 20 smallest symbols ordered by frequency: coding symbol
 * y bumps its count and lets it overtake the rank just above it.            */
typedef struct {
    uint8_t rank_of[NB_NMAP];   /* y -> z */
    uint8_t sym_at[NB_NMAP];    /* z -> y */
    int     count[NB_NMAP];     /* indexed by rank */
} nb_mapper;

static inline void nb_mapper_init(nb_mapper *m) {
    for (int i = 0; i < NB_NMAP; i++) {
        m->rank_of[i] = (uint8_t)i; m->sym_at[i] = (uint8_t)i;
        m->count[i] = 2 * (NB_NMAP - 1 - i);
    }
}
static inline int nb_mapper_y2z(const nb_mapper *m, int y) { return y < NB_NMAP ? m->rank_of[y] : y; }
static inline int nb_mapper_z2y(const nb_mapper *m, int z) { return z < NB_NMAP ? m->sym_at[z] : z; }
static inline void nb_mapper_observe(nb_mapper *m, int y) {
    if (y >= NB_NMAP) return;
    int z = m->rank_of[y];
    m->count[z]++;
    if (z > 0 && m->count[z - 1] < m->count[z]) {
        int other = m->sym_at[z - 1];
        int tmp = m->count[z]; m->count[z] = m->count[z - 1]; m->count[z - 1] = tmp;
        m->sym_at[z] = (uint8_t)other;   m->sym_at[z - 1] = (uint8_t)y;
        m->rank_of[y] = (uint8_t)(z - 1); m->rank_of[other] = (uint8_t)z;
    }
}

/* ---- adaptive binary counters (NBLIC.c:589-637) ------------------------ */
typedef struct { int c0, c1; } nb_counter;

static inline int nb_counter_p1(const nb_counter *c) { return NB_PROB_ONE * c->c1 / (c->c0 + c->c1); }
static inline void nb_counter_add(nb_counter *c, int bin, int weight) {
    if (bin) c->c1 += weight; else c->c0 += weight;
    if (c->c0 + c->c1 > NB_NQW * NB_MAXCOUNT) { c->c0 = (c->c0 + 1) >> 1; c->c1 = (c->c1 + 1) >> 1; }
}
static inline int nb_mix_prob(int pu, int pv, int qw) {
    return nb_clip((pu * (NB_NQW - qw) + pv * qw + NB_NQW / 2) / NB_NQW, 1, NB_PROB_ONE - 1);
}

/* ---- 32-bit carry-less binary range coder (NBLIC.c:527-586) ------------ */
typedef struct {
    uint8_t *p;          /* next byte to write / read */
    uint32_t lo, hi;     /* inclusive interval */
    uint32_t window;     /* decoder: last four stream bytes */
    int      decoding;
} nb_rc;

static inline void nb_rc_start(nb_rc *rc, uint8_t *p, int decoding) {
    rc->p = p; rc->lo = 0; rc->hi = 0xFFFFFFFFu; rc->window = 0; rc->decoding = decoding;
    if (decoding) for (int k = 0; k < 4; k++) rc->window = (rc->window << 8) | *rc->p++;
}
/* prob = P(bin==1) in 1/4096; bin 1 takes the lower sub-interval. */
static inline int nb_rc_bin(nb_rc *rc, int bin, uint32_t prob) {
    uint32_t span = rc->hi - rc->lo;
    uint32_t cut  = rc->lo + (uint32_t)(((uint64_t)span * prob) >> 12);   /* == the two-term form at :553 */
    if (rc->decoding) bin = rc->window <= cut;
    if (bin) rc->hi = cut; else rc->lo = cut + 1;
    while (((rc->lo ^ rc->hi) & 0xFF000000u) == 0) {
        if (rc->decoding) rc->window = (rc->window << 8) | *rc->p++;
        else              *rc->p++ = (uint8_t)(rc->hi >> 24);
        rc->lo <<= 8; rc->hi = (rc->hi << 8) | 0xFF;
    }
    return bin;
}
static inline void nb_rc_finish(nb_rc *rc) {
    if (rc->decoding) return;
    for (int k = 0; k < 4; k++) { *rc->p++ = (uint8_t)(rc->lo >> 24); rc->lo <<= 8; }
}

/* ---- binarisation walk (NBLIC.c:640-679) -------------------------------
 * Visits the (tree-u, tree-v, node) triples of one symbol in coding order.
 * `step(ctx, qu, qv, node, qw, bin)` codes one bin and returns it; when
 * z_in < 0 (decoding) the walk passes bin = -1 and uses the returned value.
 * Returns the symbol.                                                      */
typedef int (*nb_bin_fn)(void *ctx, int qu, int qv, int node, int qw, int bin);

static inline int nb_walk_symbol(int k_step, int qu, int qv, int qw, int z_in, nb_bin_fn step, void *ctx) {
    const int k_max = (NB_NQD - 1) / k_step;
    const int decoding = z_in < 0;
    int node = 0, k, bin, z;
    if (qv / k_step != qu / k_step) qv = qu;
    for (;;) {
        k = qu / k_step;
        bin = decoding ? -1 : ((node >> k_max) < (z_in >> k));
        bin = step(ctx, qu, qv, node, qw, bin);
        if (!bin) break;
        node += 1 << k_max;
        if (node >= NB_TREE) { node >>= 1; qu = qv = (k + 1) * k_step; }
    }
    z = decoding ? ((node >> k_max) << k) : z_in;
    node++;
    for (k--; k >= 0; k--) {
        bin = decoding ? -1 : ((z_in >> k) & 1);
        bin = step(ctx, qu, qv, node, qw, bin);
        if (decoding && bin) z += 1 << k;
        node += bin ? (1 << k) : 1;
    }
    return z;
}

/* ---- stream header (NBLIC.c:682-712) ----------------------------------- */
#define NB_MAGIC "NBLIC0.3"
enum { NB_HEADER_BYTES = 16 };

#endif

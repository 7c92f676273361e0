/*
 * oracle/nblic_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Fused, single-threaded CPU restatement of the NBLIC v0.3 codec loop
 * (/root/reference/src/NBLIC.c:749-908): every effort (1..3), every near
 * (0..9), encoder and decoder.  It is the checker for the HIP path and the
 * "port" CPU baseline that may travel to the GPU box; the product library
 * never links or calls it.
 *
 * Parity status: PINNED against oracle/_ref/libnblic_ref.so (the unmodified
 * reference compiled by oracle/Makefile) and tests/golden/ fixtures; see
 * tests/test_oracle.py (golden fixtures, live compares with the compiled reference, Kodak).
 *
 * Layout of this file
 *   lsq_*      integer weighted-least-squares predictor ("AVP", NBLIC.c:112-283)
 *   engine     shared adaptive state + one code_pixel() used by both directions
 *   orc_*      exported entry points (plain C ABI for ctypes)
 */
#include <stdlib.h>
#include <string.h>
#include "nblic_model.h"

typedef int64_t i64;
typedef uint64_t u64;

/* wrapping signed multiply (the reference relies on two's-complement wrap, NBLIC.c:139) */
static inline i64 mulw(i64 a, i64 b) { return (i64)((u64)a * (u64)b); }
static inline i64 labs64(i64 v)      { return v < 0 ? -v : v; }
static inline i64 clip64(i64 v, i64 lo, i64 hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline i64 floor_half(i64 v)  { return v >= 0 ? v / 2 : -((-v + 1) / 2); }   /* v >> 1 */

/* ------------------------------------------------------------------------
 * Least-squares predictor.  Statistics vectors have m = 1 + n + n*n entries
 * laid out as [s | b(n) | A(n x n)] (NBLIC.c:213-215, 246-248).
 * ---------------------------------------------------------------------- */
enum { LSQ_MAX_N = 10, LSQ_MAX_M = 1 + LSQ_MAX_N + LSQ_MAX_N * LSQ_MAX_N };
enum { FB1 = 12, FB2 = 2, FB3 = 10, DECAY_S = 3, DECAY_V = 5, BIAS_INIT = 8, BIAS_MAX = 4096, BIAS_COEF = 21 };

static inline int lsq_order(int effort) { return effort == 2 ? 6 : (effort == 3 ? 10 : 0); }   /* N_LIST, :88 */

/* exponential decay with round-half-up on the magnitude-truncating divide (:199, :273-279) */
static inline i64 lsq_decay(i64 v, int ab) { return (mulw(v, ab - 1) + ab / 2) / ab; }

/* regressor = (a,b,c,d,e,f,t,h,q,g)[0..n) - 128   (NBLIC.c:164-183) */
static void lsq_regressor(i64 *vn, int n, const nb_taps *t) {
    const int order[LSQ_MAX_N] = { t->a, t->b, t->c, t->d, t->e, t->f, t->t, t->h, t->q, t->g };
    for (int k = 0; k < n; k++) vn[k] = order[k] - NB_MID;
}

/* once per row, right-to-left horizontal accumulation of the row-above stats (NBLIC.c:186-204) */
static void lsq_row_prepare(int m, i64 *F, const i64 *B, int w) {
    for (int j = w - 1; j >= 0; j--)
        for (int k = 0; k < m; k++) {
            i64 carry = (j == w - 1) ? 0 : lsq_decay(F[(size_t)(j + 1) * m + k], k ? DECAY_V : DECAY_S);
            F[(size_t)j * m + k] = carry + B[(size_t)j * m + k];
        }
}

/* in-place integer Gaussian elimination, partial pivoting, then back-substitution on b only
 * (NBLIC.c:112-161).  Returns 0 when a pivot is zero.                                        */
static int lsq_solve(int n, i64 *A, i64 *b) {
    for (int k = 0; k + 1 < n; k++) {
        int piv = k;
        for (int i = k + 1; i < n; i++)
            if (labs64(A[i * n + k]) > labs64(A[piv * n + k])) piv = i;
        if (piv != k) {
            i64 t = b[k]; b[k] = b[piv]; b[piv] = t;
            for (int j = k; j < n; j++) { t = A[k * n + j]; A[k * n + j] = A[piv * n + j]; A[piv * n + j] = t; }
        }
        i64 d = A[k * n + k];
        if (d == 0) return 0;
        for (int i = k + 1; i < n; i++) {
            i64 l = A[i * n + k];
            A[i * n + k] = 0;
            if (l == 0) continue;
            for (int j = k + 1; j < n; j++) A[i * n + j] -= mulw(A[k * n + j], l) / d;
            b[i] -= mulw(b[k], l) / d;
        }
    }
    for (int k = n - 1; k > 0; k--) {
        i64 d = A[k * n + k];
        if (d == 0) return 0;
        for (int i = 0; i < k; i++) {
            i64 l = A[i * n + k];
            A[i * n + k] = 0;
            if (l != 0) b[i] -= mulw(b[k], l) / d;
        }
    }
    return 1;
}

/* Q12 prediction from the regularised normal equations (NBLIC.c:210-239) */
static int lsq_predict(int n, int m, const i64 *E, const i64 *F, const i64 *vn, i64 bias, i64 *px_q12) {
    i64 sys[LSQ_MAX_M];
    i64 *b = sys + 1, *A = sys + 1 + n;
    for (int k = 1; k < m; k++) sys[k] = E[k] + F[k];
    for (int k = 0; k < n; k++) { b[k] += bias * (1 << FB3); A[k * n + k] += bias * n; }
    if (!lsq_solve(n, A, b)) return 0;
    i64 px = (i64)NB_MID << FB1;
    for (int k = 0; k < n; k++) {
        i64 d = A[k * n + k];
        px += (mulw(mulw(b[k], vn[k]), 1 << FB2) + floor_half(d)) / d;
    }
    *px_q12 = clip64(px, 0, (i64)NB_MAXVAL << FB1);
    return 1;
}

/* fold the newly coded pixel into the running statistics (NBLIC.c:242-283) */
static void lsq_update(int n, int m, i64 *E, i64 *B, const i64 *vn, int x, i64 s_curr, i64 s_sum) {
    i64 sample[LSQ_MAX_M];
    i64 *b = sample + 1, *A = sample + 1 + n;
    i64 xc = x - NB_MID;
    s_sum = clip64(s_sum + (1 << FB1), 1 << FB1, 16 << FB1);
    i64 half = s_sum >> 1;
    sample[0] = s_curr;
    for (int k = 0; k < n; k++)
        b[k] = (mulw(xc * vn[k], (i64)1 << (4 + FB1 + FB1)) + half) / s_sum;
    for (int j = 0; j < n; j++)
        for (int k = 0; k < n; k++)
            A[j * n + k] = (mulw(vn[j] * vn[k], (i64)1 << (4 + FB2 + FB1)) + half) / s_sum;
    for (int k = 0; k < m; k++) {
        int ab = k ? DECAY_V : DECAY_S;
        B[k] = lsq_decay(B[k], ab) + sample[k];
        E[k] = lsq_decay(E[k], ab) + B[k];
    }
}

/* ------------------------------------------------------------------------
 * Engine: all adaptive state of one image stream.
 * ---------------------------------------------------------------------- */
typedef struct {
    int        near, k_step, effort, n, m, w, h;
    int        ctx[NB_NCTX];
    nb_counter tree[NB_NQD][NB_TREE];
    nb_mapper  map[256][2];
    nb_rc      rc;
    /* least-squares state */
    i64       *Brow, *Frow;          /* w*m each */
    i64        E[LSQ_MAX_M];
    i64        bias;
    /* statistics, filled when non-NULL */
    long       n_bins;
} engine;

static int engine_bin(void *vp, int qu, int qv, int node, int qw, int bin) {
    engine *en = (engine *)vp;
    nb_counter *cu = &en->tree[qu][node], *cv = &en->tree[qv][node];
    int prob = nb_mix_prob(nb_counter_p1(cu), nb_counter_p1(cv), qw);
    bin = nb_rc_bin(&en->rc, bin, (uint32_t)prob);
    nb_counter_add(cu, bin, NB_NQW - qw);
    nb_counter_add(cv, bin, qw);
    en->n_bins++;
    return bin;
}

static int engine_init(engine *en, int h, int w, int near, int k_step, int effort, uint8_t *stream, int decoding) {
    memset(en, 0, sizeof *en);
    en->h = h; en->w = w; en->near = near; en->k_step = k_step; en->effort = effort;
    en->n = lsq_order(effort); en->m = 1 + en->n + en->n * en->n;
    en->bias = BIAS_INIT;
    for (int q = 0; q < NB_NQD; q++)
        for (int i = 0; i < NB_TREE; i++) en->tree[q][i].c0 = en->tree[q][i].c1 = NB_NQW;
    for (int p = 0; p < 256; p++) { nb_mapper_init(&en->map[p][0]); nb_mapper_init(&en->map[p][1]); }
    if (en->n > 0) {
        en->Brow = (i64 *)calloc((size_t)w * en->m * 2, sizeof(i64));
        if (!en->Brow) return -1;
        en->Frow = en->Brow + (size_t)w * en->m;
    }
    nb_rc_start(&en->rc, stream, decoding);
    return 0;
}

/* Codes pixel (i,j).  `recon` holds reconstructed pixels for everything already coded
 * (and, when encoding, the original value at (i,j) itself -- NBLIC.c:862 reads it before :876
 * overwrites it).  *err_prev is the in-row carried error (:808, :878).                    */
static void engine_pixel(engine *en, uint8_t *recon, int i, int j, int decoding, int *err_prev) {
    const int w = en->w, n = en->n, m = en->m;
    nb_taps t;
    i64 vn[LSQ_MAX_N], b1 = 0, b2 = 0, p1 = 0, p2 = 0;
    int ok1 = 0, ok2 = 0, px0;
    i64 *B = NULL, *F = NULL;

    nb_sample(recon, w, i, j, &t);

    if (n > 0) {                                             /* NBLIC.c:831-846 */
        lsq_regressor(vn, n, &t);
        B = en->Brow + (size_t)j * m; F = en->Frow + (size_t)j * m;
        b1 = en->bias * BIAS_COEF / (BIAS_COEF + 1);
        b2 = en->bias * (BIAS_COEF + 1) / BIAS_COEF;
        b1 = clip64(clip64(b1, -1, en->bias - 1), 0, BIAS_MAX);
        b2 = clip64(clip64(b2, en->bias + 1, BIAS_MAX + 1), 0, BIAS_MAX);
        ok1 = lsq_predict(n, m, en->E, F, vn, b1, &p1);
        ok2 = lsq_predict(n, m, en->E, F, vn, b2, &p2);
    }
    if (ok1) px0 = (int)((p1 + (1 << (FB1 - 1))) >> FB1);
    else   { px0 = nb_predict(&t); p1 = (i64)px0 << FB1; }

    int qu, qv, qw, sign;
    nb_quantise(nb_delta(&t, *err_prev), &qu, &qv, &qw);
    int adr = nb_ctx_addr(&t, qu, px0);
    int px  = nb_ctx_correct(en->ctx[adr], px0, &sign);
    nb_mapper *mp = &en->map[px][sign];

    int y;
    if (!decoding) {
        int x = recon[(size_t)i * w + j];
        y = nb_x_to_y(x, px, sign, en->near);
        nb_walk_symbol(en->k_step, qu, qv, qw, nb_mapper_y2z(mp, y), engine_bin, en);
    } else {
        y = nb_mapper_z2y(mp, nb_walk_symbol(en->k_step, qu, qv, qw, -1, engine_bin, en));
    }
    nb_mapper_observe(mp, y);

    int xr = nb_y_to_x(y, px, sign, en->near);
    recon[(size_t)i * w + j] = (uint8_t)xr;
    *err_prev = nb_clip(xr - px0, -(NB_MAXVAL - NB_MID), NB_MAXVAL - NB_MID);
    en->ctx[adr] = nb_ctx_update(en->ctx[adr], *err_prev);

    if (n > 0) {                                             /* NBLIC.c:882-893 */
        i64 xq = (i64)xr << FB1;
        i64 s_curr = labs64(p1 - xq);
        i64 s_sum  = (en->E[0] + F[0]) + s_curr * DECAY_S / (DECAY_S - 1);
        lsq_update(n, m, en->E, B, vn, xr, s_curr, s_sum);
        if (ok1 && ok2) en->bias = (labs64(p1 - xq) > labs64(p2 - xq)) ? b2 : b1;
    }
}

static void engine_run(engine *en, uint8_t *recon, int decoding) {
    for (int i = 0; i < en->h; i++) {
        int err = 0;
        if (en->n > 0) {
            memset(en->E, 0, sizeof(i64) * (size_t)en->m);
            lsq_row_prepare(en->m, en->Frow, en->Brow, en->w);
        }
        for (int j = 0; j < en->w; j++) engine_pixel(en, recon, i, j, decoding, &err);
    }
    nb_rc_finish(&en->rc);
    free(en->Brow);
}

static int size_ok(int h, int w, long max_px) {          /* NBLIC.c:717-729 */
    return h > 0 && w > 0 && h <= 65535 && w <= 65535 && (long)h * (long)w <= max_px;
}

/* ------------------------------------------------------------------------
 * Exported entry points
 * ---------------------------------------------------------------------- */

/* Encode.  `img` is overwritten with the reconstruction exactly like the reference
 * (NBLIC.c:876).  *near / *effort are clamped and written back (:768-770).
 * max_px <= 0 selects the reference limit of 100,000,000 pixels.  Returns stream bytes or -1. */
long orc_nblic_encode(uint8_t *out, uint8_t *img, int h, int w, int *near, int *effort, long max_px, long *n_bins) {
    if (max_px <= 0) max_px = 100000000L;
    *near   = nb_clip(*near, 0, NB_MAX_NEAR);
    *effort = nb_clip(*effort, 1, 3);
    int k_step = nb_clip(NB_MIN_KSTEP + 2 * *near, NB_MIN_KSTEP, NB_NQD);
    uint8_t *p = out;
    memcpy(p, NB_MAGIC, 8); p += 8;
    *p++ = 1;
    *p++ = (uint8_t)(h >> 8); *p++ = (uint8_t)h;
    *p++ = (uint8_t)(w >> 8); *p++ = (uint8_t)w;
    *p++ = (uint8_t)*near; *p++ = (uint8_t)k_step; *p++ = (uint8_t)*effort;
    if (!size_ok(h, w, max_px)) return -1;
    engine *en = (engine *)malloc(sizeof(engine));
    if (!en || engine_init(en, h, w, *near, k_step, *effort, p, 0)) { free(en); return -1; }
    engine_run(en, img, 0);
    long len = (long)(en->rc.p - out);
    if (n_bins) *n_bins = en->n_bins;
    free(en);
    return len;
}

/* Decode.  Returns 0 / -1; all four geometry/parameter outputs come from the header. */
int orc_nblic_decode(const uint8_t *in, uint8_t *img, int *h, int *w, int *near, int *effort, long max_px) {
    if (max_px <= 0) max_px = 100000000L;
    if (memcmp(in, NB_MAGIC, 8) != 0) return -1;
    int n_channel = in[8];
    *h = (in[9] << 8) | in[10]; *w = (in[11] << 8) | in[12];
    *near = in[13]; int k_step = in[14]; *effort = in[15];
    if (!size_ok(*h, *w, max_px) || n_channel > 1 || *near > NB_MAX_NEAR ||
        k_step < NB_MIN_KSTEP || k_step > NB_NQD || *effort < 1 || *effort > 3) return -1;
    engine *en = (engine *)malloc(sizeof(engine));
    if (!en || engine_init(en, *h, *w, *near, k_step, *effort, (uint8_t *)in + NB_HEADER_BYTES, 1)) { free(en); return -1; }
    engine_run(en, img, 1);
    free(en);
    return 0;
}

/* SYN-1 deterministic test frame. */
void orc_syn1(uint8_t *img, int h, int w, uint32_t seed) {
    uint32_t xs = seed;
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
            xs ^= xs << 13; xs ^= xs >> 17; xs ^= xs << 5;
            int t = ((i + 2 * j) >> 3) & 511;
            int base = nb_iabs(t - 256); if (base > 255) base = 255;
            int tex = (i ^ j) & 15;
            int noise = (int)(xs & 7) + (int)((xs >> 3) & 7) - 7;
            img[(size_t)i * w + j] = (uint8_t)nb_clip(((base * 3) >> 2) + 32 + tex + noise, 0, 255);
        }
}

/*
 * oracle/nblic_staged.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * The -e1 lossless NBLIC encoder restated as six array-to-array stages
 * (SURVEY.md section 7.3).  Its purpose is (1) to prove on the CPU that the
 * raster-ordered adaptive state can be replayed one key at a time without
 * changing a single output byte, and (2) to give every HIP kernel a
 * stage-level expected output.
 *
 *   S1  stateless per pixel : px0, err, (qu,qv,qw), context address
 *   S2  2048 chains by adr  : bias-corrected px and sign          (NBLIC.c:413-428)
 *   S3  512 chains by px|sign: residual symbol y -> adaptive rank z (NBLIC.c:431-523)
 *   S4  stateless per pixel : bin events (tree-u, tree-v, node, qw, bin) (NBLIC.c:640-679)
 *   S5  4096 chains by counter: probability of every event          (NBLIC.c:589-637)
 *   S6  one serial chain    : range coder                           (NBLIC.c:552-586)
 *
 * S2, S3 and S5 deliberately run "all of key 0, then all of key 1, ..." over a
 * stable partition, never in raster order, so a passing byte-compare against
 * the fused engine / the reference is a proof of the decomposition.
 *
 * Parity status: PINNED (tests/test_oracle_staged.py compares against
 * nblic_oracle.c, the compiled reference and tests/golden/).
 */
#include <stdlib.h>
#include <string.h>
#include "nblic_model.h"

enum { KSTEP_LOSSLESS = 3 };

/* ---- S1 ---------------------------------------------------------------- */
void orc_s1(const uint8_t *img, int h, int w,
            uint8_t *px0, int8_t *err, uint8_t *qu, uint8_t *qv, uint8_t *qw, uint16_t *adr) {
    for (int i = 0; i < h; i++) {
        int e_prev = 0;
        for (int j = 0; j < w; j++) {
            size_t t = (size_t)i * w + j;
            nb_taps n; int u, v, k;
            nb_sample(img, w, i, j, &n);
            int p = nb_predict(&n);
            nb_quantise(nb_delta(&n, e_prev), &u, &v, &k);
            e_prev = nb_clip((int)img[t] - p, -127, 127);
            px0[t] = (uint8_t)p; err[t] = (int8_t)e_prev;
            qu[t] = (uint8_t)u; qv[t] = (uint8_t)v; qw[t] = (uint8_t)k;
            adr[t] = (uint16_t)nb_ctx_addr(&n, u, p);
        }
    }
}

/* stable counting partition: order[] lists item indices grouped by key, raster order kept */
static size_t *partition(const uint32_t *key, size_t n, int n_keys, size_t *start /* n_keys+1 */) {
    size_t *order = (size_t *)malloc(sizeof(size_t) * (n ? n : 1));
    memset(start, 0, sizeof(size_t) * (size_t)(n_keys + 1));
    for (size_t t = 0; t < n; t++) start[key[t] + 1]++;
    for (int k = 0; k < n_keys; k++) start[k + 1] += start[k];
    size_t *cur = (size_t *)malloc(sizeof(size_t) * (size_t)n_keys);
    memcpy(cur, start, sizeof(size_t) * (size_t)n_keys);
    for (size_t t = 0; t < n; t++) order[cur[key[t]]++] = t;
    free(cur);
    return order;
}

/* ---- S2 ---------------------------------------------------------------- */
void orc_s2(size_t n, const uint16_t *adr, const uint8_t *px0, const int8_t *err,
            uint8_t *px, uint8_t *sign) {
    uint32_t *key = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    size_t start[NB_NCTX + 1];
    for (size_t t = 0; t < n; t++) key[t] = adr[t];
    size_t *order = partition(key, n, NB_NCTX, start);
    for (int k = 0; k < NB_NCTX; k++) {
        int v = 0;
        for (size_t r = start[k]; r < start[k + 1]; r++) {
            size_t t = order[r]; int s;
            px[t] = (uint8_t)nb_ctx_correct(v, px0[t], &s);
            sign[t] = (uint8_t)s;
            v = nb_ctx_update(v, err[t]);
        }
    }
    free(order); free(key);
}

/* ---- S3 ---------------------------------------------------------------- */
void orc_s3(size_t n, const uint8_t *img, const uint8_t *px, const uint8_t *sign, uint8_t *y_out, uint8_t *z_out) {
    uint32_t *key = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    size_t start[512 + 1];
    for (size_t t = 0; t < n; t++) key[t] = (uint32_t)px[t] * 2 + sign[t];
    size_t *order = partition(key, n, 512, start);
    for (int k = 0; k < 512; k++) {
        nb_mapper m; nb_mapper_init(&m);
        for (size_t r = start[k]; r < start[k + 1]; r++) {
            size_t t = order[r];
            int y = nb_x_to_y(img[t], px[t], sign[t], 0);
            y_out[t] = (uint8_t)y;
            z_out[t] = (uint8_t)nb_mapper_y2z(&m, y);
            nb_mapper_observe(&m, y);
        }
    }
    free(order); free(key);
}

/* ---- S4 ---------------------------------------------------------------- */
typedef struct { uint16_t *cu, *cv; uint8_t *qw, *bin; size_t n; int count_only; } ev_sink;

static int ev_emit(void *vp, int qu, int qv, int node, int qw, int bin) {
    ev_sink *s = (ev_sink *)vp;
    if (!s->count_only) {
        s->cu[s->n] = (uint16_t)(qu * NB_TREE + node);
        s->cv[s->n] = (uint16_t)(qv * NB_TREE + node);
        s->qw[s->n] = (uint8_t)qw; s->bin[s->n] = (uint8_t)bin;
    }
    s->n++;
    return bin;
}

/* first call with cu==NULL to size, then again to fill.  ev_count[t] (optional) = events of pixel t */
size_t orc_s4(size_t n, const uint8_t *qu, const uint8_t *qv, const uint8_t *qw, const uint8_t *z,
              uint16_t *cu, uint16_t *cv, uint8_t *ev_qw, uint8_t *ev_bin, uint8_t *ev_count) {
    ev_sink s = { cu, cv, ev_qw, ev_bin, 0, cu == NULL };
    for (size_t t = 0; t < n; t++) {
        size_t before = s.n;
        nb_walk_symbol(KSTEP_LOSSLESS, qu[t], qv[t], qw[t], z[t], ev_emit, &s);
        if (ev_count) ev_count[t] = (uint8_t)(s.n - before);
    }
    return s.n;
}

/* ---- S5 ----------------------------------------------------------------
 * Touch list: event r contributes touch 2r (counter cu, weight 32-qw) and, when cv != cu,
 * touch 2r+1 (counter cv, weight qw).  When cu == cv the single counter receives both
 * weights back to back and both probabilities are read before the first (NBLIC.c:629-636). */
void orc_s5(size_t n_ev, const uint16_t *cu, const uint16_t *cv, const uint8_t *qw, const uint8_t *bin,
            uint16_t *prob) {
    size_t n_touch = 0;
    uint32_t *key = (uint32_t *)malloc(sizeof(uint32_t) * (2 * n_ev + 1));
    size_t   *ref = (size_t *)malloc(sizeof(size_t) * (2 * n_ev + 1));     /* 2*event + slot */
    for (size_t r = 0; r < n_ev; r++) {
        key[n_touch] = cu[r]; ref[n_touch++] = 2 * r;
        if (cv[r] != cu[r]) { key[n_touch] = cv[r]; ref[n_touch++] = 2 * r + 1; }
    }
    size_t *start = (size_t *)malloc(sizeof(size_t) * (NB_NQD * NB_TREE + 1));
    size_t *order = partition(key, n_touch, NB_NQD * NB_TREE, start);
    uint16_t *p_uv = (uint16_t *)calloc(2 * n_ev + 1, sizeof(uint16_t));
    for (int k = 0; k < NB_NQD * NB_TREE; k++) {
        nb_counter c = { NB_NQW, NB_NQW };
        for (size_t q = start[k]; q < start[k + 1]; q++) {
            size_t tr = ref[order[q]], r = tr >> 1; int slot = (int)(tr & 1);
            p_uv[tr] = (uint16_t)nb_counter_p1(&c);
            if (cu[r] == cv[r]) {
                p_uv[tr + 1] = p_uv[tr];
                nb_counter_add(&c, bin[r], NB_NQW - qw[r]);
                nb_counter_add(&c, bin[r], qw[r]);
            } else {
                nb_counter_add(&c, bin[r], slot ? qw[r] : NB_NQW - qw[r]);
            }
        }
    }
    for (size_t r = 0; r < n_ev; r++) prob[r] = (uint16_t)nb_mix_prob(p_uv[2 * r], p_uv[2 * r + 1], qw[r]);
    free(p_uv); free(order); free(start); free(ref); free(key);
}

/* ---- S6 ---------------------------------------------------------------- */
size_t orc_s6(size_t n_ev, const uint16_t *prob, const uint8_t *bin, uint8_t *out) {
    nb_rc rc; nb_rc_start(&rc, out, 0);
    for (size_t r = 0; r < n_ev; r++) nb_rc_bin(&rc, bin[r], prob[r]);
    nb_rc_finish(&rc);
    return (size_t)(rc.p - out);
}

/* ---- whole -e1 lossless encode through the stages ---------------------- */
long orc_nblic_encode_staged(uint8_t *out, const uint8_t *img, int h, int w, long *n_events) {
    size_t n = (size_t)h * w;
    uint8_t *p = out;
    memcpy(p, NB_MAGIC, 8); p += 8; *p++ = 1;
    *p++ = (uint8_t)(h >> 8); *p++ = (uint8_t)h; *p++ = (uint8_t)(w >> 8); *p++ = (uint8_t)w;
    *p++ = 0; *p++ = KSTEP_LOSSLESS; *p++ = 1;
    uint8_t *px0 = malloc(n), *qu = malloc(n), *qv = malloc(n), *qw = malloc(n);
    uint8_t *px = malloc(n), *sg = malloc(n), *y = malloc(n), *z = malloc(n);
    int8_t *err = malloc(n); uint16_t *adr = malloc(2 * n);
    orc_s1(img, h, w, px0, err, qu, qv, qw, adr);
    orc_s2(n, adr, px0, err, px, sg);
    orc_s3(n, img, px, sg, y, z);
    size_t ne = orc_s4(n, qu, qv, qw, z, NULL, NULL, NULL, NULL, NULL);
    uint16_t *cu = malloc(2 * ne + 2), *cv = malloc(2 * ne + 2), *prob = malloc(2 * ne + 2);
    uint8_t *eq = malloc(ne + 1), *eb = malloc(ne + 1);
    orc_s4(n, qu, qv, qw, z, cu, cv, eq, eb, NULL);
    orc_s5(ne, cu, cv, eq, eb, prob);
    size_t body = orc_s6(ne, prob, eb, p);
    if (n_events) *n_events = (long)ne;
    free(px0); free(qu); free(qv); free(qw); free(px); free(sg); free(y); free(z); free(err); free(adr);
    free(cu); free(cv); free(prob); free(eq); free(eb);
    return (long)(NB_HEADER_BYTES + body);
}

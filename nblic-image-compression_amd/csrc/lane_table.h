// lane_table.h -- the per-lane constants of the lane-parallel pixel front (serial_engine.hip LaneFront, the QNBLIC
// decoder's row loop): which term of the model a lane computes, as a constexpr table.  Host-compilable, so that
// tests/host_harness.cpp can walk the 64 lanes on the CPU and compare with model.h (tests/test_host_logic.py).
//
// Every cost of the seven-direction predictor, every term of the activity and every comparison of the context address
// has the shape  2 X - Y - Z  over the taps; lane k holds ONE such term:
//   lanes  0..27  the 7 x 4 cost terms, direction d in quad d (NBLIC.c:307-370 / QNBLIC.c:94-149)
//   lanes 28..33  the six activity terms, twice their value (NBLIC.c:376 / QNBLIC.c:152-161)
//   lanes 36..43  the eight comparison values of the context address, bit 0 first (NBLIC.c:398-410 / QNBLIC.c:164-173)
//   lanes 44..53  (NBLIC) the ten least-squares regressors, twice their value (NBLIC.c:164-183)
#pragma once
#include <stdint.h>

namespace nblic {

enum QTap : int8_t { qZ, qA, qE, qB, qC, qD, qQ, qT, qF, qG, qH, qR, qS };
struct QLaneConst {
    int8_t sel[3], dx[3];                    // X, Y, Z: 0 the zero byte / 1 row i-1 / 2 row i-2, and the column offset
    int8_t a2, ce, sh;                       // + a2 * a into 2X; + ce * e into Y + Z; the prediction is shifted left by sh for the comparison
    int8_t ca, cb, cc, cd;                   // lanes 0..6: the direction's extrapolation  ca a + cb b + cc c + cd d  (twice the neighbour)
    int8_t dst;                              // lanes 44..53 (NBLIC): the regressor the lane holds (V / 2), -1 elsewhere
    int16_t thr_level, thr_weight;           // lanes 0..10 / 0..7 (0x7FFF elsewhere)
    int32_t key_or, sum_and;                 // lanes 0..27: direction / all ones; elsewhere 0x7FFFFFFF / 0
};
struct QLaneTable { QLaneConst l[64]; };
constexpr int kRowPad = 8;                   // a row's margins in LDS: 2 columns left (behind 4 bytes, the first of them zero, in front of row 0), >= 4 right
constexpr QLaneTable make_lanes(bool q) {    // q: QNBLIC (effort 0), else NBLIC
    constexpr QTap terms[36][3] = {
        {qA, qE, qE}, {qC, qQ, qQ}, {qB, qC, qC}, {qD, qB, qB},          // west        2 (|a-e| + |c-q| + |b-c| + |d-b|)
        {qA, qC, qC}, {qC, qH, qH}, {qB, qF, qF}, {qD, qG, qG},          // north
        {qA, qQ, qQ}, {qC, qS, qS}, {qB, qH, qH}, {qD, qF, qF},          // north-west
        {qA, qB, qB}, {qC, qF, qF}, {qB, qG, qG}, {qD, qR, qR},          // north-east
        {qA, qE, qQ}, {qC, qQ, qS}, {qB, qC, qH}, {qD, qB, qF},          // between west and north-west
        {qA, qQ, qC}, {qC, qS, qH}, {qB, qH, qF}, {qD, qF, qG},          // between north-west and north
        {qA, qC, qB}, {qC, qH, qF}, {qB, qF, qG}, {qD, qG, qR},          // between north and north-east
        {qA, qE, qE}, {qB, qC, qC}, {qB, qD, qD}, {qA, qC, qC},          // activity
        {qB, qF, qF}, {qD, qG, qG}, {qZ, qZ, qZ}, {qZ, qZ, qZ}};
    // comparison values, bit 0 first; a and e alone are compared as 2a / 2e with twice the prediction (qA, qZ, qZ / ce = -2)
    constexpr QTap cmp_q[8][3] = {{qB, qF, qZ}, {qA, qE, qZ}, {qF, qF, qZ}, {qE, qZ, qZ}, {qD, qD, qZ}, {qC, qC, qZ}, {qB, qB, qZ}, {qA, qZ, qZ}};
    constexpr QTap cmp_n[8][3] = {{qA, qZ, qZ}, {qB, qB, qZ}, {qC, qC, qZ}, {qD, qD, qZ}, {qE, qZ, qZ}, {qF, qF, qZ}, {qA, qE, qZ}, {qB, qF, qZ}};
    constexpr QTap regress[10] = {qA, qB, qC, qD, qE, qF, qT, qH, qQ, qG};
    constexpr int8_t row_of[13] = {0, 0, 0, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2};   // Z A E B C D Q T F G H R S
    constexpr int8_t col_of[13] = {0, 0, 0, 0, -1, 1, -2, 2, 0, 1, -1, 2, -2};
    constexpr int8_t ang[7][4] = {{2, 0, 0, 0}, {0, 2, 0, 0}, {0, 0, 2, 0}, {0, 0, 0, 2}, {1, 0, 1, 0}, {0, 1, 1, 0}, {0, 1, 0, 1}};
    constexpr int16_t levels[11] = {1, 2, 4, 6, 9, 15, 25, 39, 63, 101, 151};
    constexpr int16_t weights_q[8] = {5, 12, 34, 78, 194, 431, 601, 0x7FFF};
    constexpr int16_t weights_n[8] = {31, 93, 279, 620, 1550, 3410, 9300, 24800};
    QLaneTable t{};
    for (int k = 0; k < 64; k++) {
        QLaneConst &c = t.l[k];
        QTap x[3] = {qZ, qZ, qZ};
        if (k < 36) for (int o = 0; o < 3; o++) x[o] = terms[k][o];
        else if (k < 44) for (int o = 0; o < 3; o++) x[o] = q ? cmp_q[k - 36][o] : cmp_n[k - 36][o];
        else if (k < 54 && !q) x[0] = regress[k - 44];
        c.dst = (k >= 44 && k < 54 && !q) ? int8_t(k - 44) : int8_t(-1);
        const bool doubled = k >= 36 && x[1] == qZ;                      // the lane's value is 2 * tap
        for (int o = 0; o < 3; o++) {
            QTap tap = x[o];
            if (o == 0 && tap == qA) { c.a2 = 2; tap = qZ; }
            if (o == 0 && tap == qE) { c.ce = -2; tap = qZ; }            // 2e = 0 - (-2e)
            if (o > 0 && tap == qE) { c.ce += 1; tap = qZ; }
            c.sel[o] = row_of[tap]; c.dx[o] = col_of[tap];
        }
        c.sh = (doubled && k < 44) ? 1 : 0;
        if (k < 7) { c.ca = ang[k][0]; c.cb = ang[k][1]; c.cc = ang[k][2]; c.cd = ang[k][3]; }
        c.thr_level = k < 11 ? levels[k] : int16_t(0x7FFF);
        c.thr_weight = k < 8 ? (q ? weights_q[k] : weights_n[k]) : int16_t(0x7FFF);
        c.key_or = k < 28 ? k >> 2 : 0x7FFFFFFF;
        c.sum_and = k < 28 ? -1 : 0;
    }
    return t;
}

}  // namespace nblic

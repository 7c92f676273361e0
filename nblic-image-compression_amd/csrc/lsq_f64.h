// lsq_f64.h -- exact integer arithmetic of the least-squares predictor ("AVP", NBLIC.c:112-283)
// carried in IEEE doubles.
//
// The reference keeps its weighted normal equations in int64 and divides with C's truncating '/'
// (NBLIC.c:139-155, :230, :258-279).  A 64-bit signed divide costs ~180 instructions on gfx950
// (there is no hardware integer divider), and the predictor needs ~120 of them per pixel.  Every
// quantity involved is an INTEGER far below 2^53 (bounds below), so it can live in a double without
// loss, and a truncating integer quotient is then: one approximate quotient (multiply by a
// reciprocal, truncate) plus an EXACT remainder by fused multiply-add, which tells whether the
// estimate has to move by one.  Products that exceed 53 bits (right-hand side x matrix entry) are
// carried as an unevaluated pair p + e with e = fma(a, b, -p), which is exact for |a b| < 2^106.
//
// Bounds (x in [0,255], regressors in [-128,127], weights s in [2^12, 2^16], decays 2/3 and 4/5):
//   matrix samples < 2^21, right-hand-side samples < 2^31; column / row / pre-pass sums each gain at
//   most a factor 5 (3 for the weight channel): statistics < 2^37, the assembled system < 2^38.
//   Elimination can double an entry per step: < 2^48 after nine steps.  None of this is relied on
//   blindly: the solver tracks the largest product and the largest entry it produced (Guard) and
//   the caller re-does the pixel with plain 64-bit integers when a limit is crossed, so the result
//   is the reference's integer in every case.
//
// Shared by the device code (serial_engine.hip) and by the host-side harness of tests/ that
// checks this arithmetic against the CPU checker without a GPU.  Compile with -ffp-contract=off.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define LSQ_HD __host__ __device__ __forceinline__
#else
#define LSQ_HD inline
#endif

namespace nblic {
namespace lsq {

constexpr int kMaxN = 10;                               // N_LIST, NBLIC.c:88: effort 2 -> 6, effort 3 -> 10
constexpr int kFb1 = 12, kFb2 = 2, kFb3 = 10;           // fixed-point positions, NBLIC.c:66-68
constexpr int kDecayS = 3, kDecayV = 5;                 // ALPHA, BETA analogues: decay (ab-1)/ab of the weight / value channels
constexpr int kBiasInit = 8, kBiasMax = 4096, kBiasCoef = 21;

LSQ_HD int order_of(int effort) { return effort == 2 ? 6 : (effort == 3 ? kMaxN : 0); }
LSQ_HD int vec_len(int n) { return 1 + n + n * n; }     // [s | b(n) | A(n x n)], NBLIC.c:213-215

// Sticky record of the largest magnitudes one solve has seen: product / entry / quotient = max |v| so far, one
// instruction per value on the device (v_max_f64 with |.| on both operands; the library fmax would wrap each operand
// in a canonicalising maximum of its own).  A maximum drops NaNs, which is sound here: a NaN can only arise after a
// zero pivot (0 * inf), the solve then reports "no solution" exactly as the reference does (NBLIC.c:118), and an
// infinity still reads as over the limit.  ok() false => redo the pixel with integers.
LSQ_HD double max_abs(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r;
    asm("v_max_f64 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return fmax(fabs(a), fabs(b));
#endif
}
struct Guard {
    double product = 0.0, entry = 0.0, quotient = 0.0, pivot = 0.0;
    LSQ_HD void see_product(double v) { product = max_abs(product, v); }
    LSQ_HD void see_entry(double v) { entry = max_abs(entry, v); }
    LSQ_HD void see_quotient(double v) { quotient = max_abs(quotient, v); }
    LSQ_HD void see_pivot(double v) { pivot = max_abs(pivot, v); }
    // 2^62: beyond it the reference's int64 product may wrap; 2^44: entries stay exact and the pivot key (|v| * 256 + tag)
    // fits 53 bits; 2^46: an estimate this large may be off by more than one; 2^38: a divisor this large defeats the
    // one-multiply correction below
    LSQ_HD bool ok() const {
        return product < 4611686018427387904.0 && entry < 17592186044416.0 && quotient < 70368744177664.0 && pivot < 274877906944.0;
    }
};

// ---- truncating division by estimate + exact remainder -------------------------------------------
// The estimate q0 = trunc(n * rs) is made with a reciprocal that is deliberately SHORT by a factor (1 - 2^-48): then
// |q0| is never above |trunc(n / d)| and at most one below it while the quotient is below 2^46.  The remainder
// r = n - q0 d is exact (a fused multiply-add), has the sign of n and |r| <= 2|d| - 1; what is left to decide is
// whether |r| >= |d|, and with which sign the quotient moves.  Both at once: trunc(r * rl) with a reciprocal that is
// LONG by (1 + 2^-40).  For |r| >= |d| the product is >= 1 + 2^-41 in magnitude and, because |r / d| <= 2 - 1/|d|,
// below 2 while |d| < 2^38.9; for |r| < |d| it is at most (1 - 1/|d|)(1 + 2^-40 + ...) < 1 while |d| < 2^39.9; its sign is
// that of r * d = that of n * d.  So the correction is ONE multiply and ONE truncation -- no compare, no select, no sign
// logic (that was five operations per quotient, forty-four quotients per pixel) -- for divisors below 2^38, which the
// caller guarantees (Guard::see_pivot; the pivots of these systems are ~2^25).  The raw reciprocal needs ~2^-49
// relative accuracy (hardware seed + two Newton steps on the device; checked there by nblic_amd_serial_selftest).
constexpr double kShort = 1.0 - 3.5527136788005009e-15;               // 1 - 2^-48
constexpr double kLong = 1.0 + 9.0949470177292824e-13;                // 1 + 2^-40

LSQ_HD double recip_raw(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;
#else
    return 1.0 / d;
#endif
}
struct Recip { double s, l; };                                         // the short and the long reciprocal of one divisor
LSQ_HD Recip recip_of_raw(double raw) { return Recip{raw * kShort, raw * kLong}; }
LSQ_HD Recip recip_of(double d) { return recip_of_raw(recip_raw(d)); }

// trunc(n / d) for an integer-valued n with |n| < 2^53 and |d| < 2^38.  Exact while |n / d| < 2^46.
LSQ_HD double div_trunc(double n, double d, const Recip &rc) {
    const double q0 = trunc(n * rc.s);
    const double r = fma(-q0, d, n);                     // exact; same sign as n, |r| < 2|d|
    return q0 + trunc(r * rc.l);
}

// trunc(a * b / d) with the product carried exactly as p + e (|a b| may exceed 2^53); |d| < 2^38.
LSQ_HD double muldiv_trunc(double a, double b, double d, const Recip &rc, Guard &g) {
    const double p = a * b;
    const double e = fma(a, b, -p);                      // a*b == p + e exactly
    const double q0 = trunc(p * rc.s);
    const double r = fma(-q0, d, p) + e;                 // exact remainder of the estimate
    g.see_product(p);
    return q0 + trunc(r * rc.l);
}

// ---- truncating division of a statistic by a small divisor: one multiply ---------------------------
// trunc(n / c) for an integer n and an integer divisor c >= 1 with |n / c| * 2^-49 < 0.5 / c:  n = k c + r (0 <= r < c,
// magnitudes) gives (|n| + 0.5) / c = k + (r + 0.5) / c, whose fractional part lies in [0.5 / c, 1 - 0.5 / c]: the
// product with a reciprocal of relative error e lands on the right side of both integers as long as (k + 1) e < 0.5 / c.
//   decays (c = 3, 5; |n| < 2^47; exactly rounded 1 / c):   2^47 * 2^-52 = 2^-5 < 0.1
//   samples (c = s <= 2^16; |n / s| <= 2^30; device reciprocal, e < 2^-49):   2^30 * 2^-49 = 2^-19 < 2^-17
LSQ_HD double half_toward(double n) {                     // n + 0.5 with n's sign (n == 0 counts as positive)
#if defined(__HIP_DEVICE_COMPILE__)
    return n + __hiloint2double((__double2hiint(n) & int(0x80000000u)) | 0x3FE00000, 0);
#else
    return n + copysign(0.5, n);
#endif
}
// (v * (ab-1) + ab/2) / ab, truncating (NBLIC.c:199, :273-279); |v| < 2^44.  The numerator has v's sign (it is ab/2 > 0 for v = 0).
template <int AB>
LSQ_HD double decay(double v) {
    const double n = fma(v, double(AB - 1), double(AB / 2));
    return trunc(half_toward(n) * (1.0 / double(AB)));
}
LSQ_HD double decay_k(double v, int k) { return k ? decay<kDecayV>(v) : decay<kDecayS>(v); }

// weight of a new sample (NBLIC.c:249-251): clip(s_sum + 4096, 4096, 65536)
LSQ_HD double sample_weight(double s_sum) {
    const double s = s_sum + double(1 << kFb1);
    return s < double(1 << kFb1) ? double(1 << kFb1) : (s > double(16 << kFb1) ? double(16 << kFb1) : s);
}

// one entry of the new sample (NBLIC.c:253-267): (prod << shift + s/2) / s with prod = (x-128)*vn_k
// (shift 28) or vn_j*vn_k (shift 18); |prod| <= 2^14, s in [2^12, 2^16]; r = recip_raw(s)
LSQ_HD double sample_entry(int prod, double scale, double s, double r) {
    const double n = fma(double(prod), scale, floor(s * 0.5));
    return trunc(half_toward(n) * r);
}
constexpr double kScaleB = 268435456.0;                  // 1 << (4 + FB1 + FB1)
constexpr double kScaleA = 262144.0;                     // 1 << (4 + FB2 + FB1)

// contribution of coefficient k to the Q12 prediction (NBLIC.c:233-236): (b*vn*4 + (d >> 1)) / d
LSQ_HD double term(double b, int vn, double d, const Recip &rc, Guard &g) {
    const double n = fma(b, double(vn * (1 << kFb2)), floor(d * 0.5));
    g.see_quotient(n * rc.s);
    g.see_entry(b);
    return div_trunc(n, d, rc);
}

// the two candidate regularisation strengths around `bias` (NBLIC.c:837-842)
LSQ_HD void bias_pair(int bias, int &b1, int &b2) {
    b1 = bias * kBiasCoef / (kBiasCoef + 1);
    b2 = bias * (kBiasCoef + 1) / kBiasCoef;
    b1 = b1 < -1 ? -1 : (b1 > bias - 1 ? bias - 1 : b1);
    b1 = b1 < 0 ? 0 : (b1 > kBiasMax ? kBiasMax : b1);
    b2 = b2 < bias + 1 ? bias + 1 : (b2 > kBiasMax + 1 ? kBiasMax + 1 : b2);
    b2 = b2 < 0 ? 0 : (b2 > kBiasMax ? kBiasMax : b2);
}

}  // namespace lsq
}  // namespace nblic

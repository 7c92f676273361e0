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
    double product = 0.0, entry = 0.0, quotient = 0.0;
    LSQ_HD void see_product(double v) { product = max_abs(product, v); }
    LSQ_HD void see_entry(double v) { entry = max_abs(entry, v); }
    LSQ_HD void see_quotient(double v) { quotient = max_abs(quotient, v); }
    // 2^62: beyond it the reference's int64 product may wrap; 2^44: entries stay exact and the pivot key (|v| * 256 + tag)
    // fits 53 bits; 2^46: an estimate this large may be off by more than one
    LSQ_HD bool ok() const { return product < 4611686018427387904.0 && entry < 17592186044416.0 && quotient < 70368744177664.0; }
};

// ---- truncating division by estimate + exact remainder -------------------------------------------
// The estimate is made with a reciprocal that is deliberately SHORT by a factor (1 - 2^-48): then
// |trunc(estimate)| is never above |trunc(true quotient)| and at most one below it while the quotient
// is below 2^46, so a single one-sided correction (is the remainder still >= |d| ?) finishes it, and
// the reciprocal itself needs no more than ~2^-50 relative accuracy (hardware seed + two Newton
// steps on the device; checked there by nblic_amd_serial_selftest).
constexpr double kShort = 1.0 - 3.5527136788005009e-15;               // 1 - 2^-48

LSQ_HD double recip_short(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r * kShort;
#else
    return (1.0 / d) * kShort;
#endif
}

// The one-sided correction of an estimated quotient: +-1.0 (the sign of n * d) when the remainder r of the estimate
// is still as large as the divisor, else 0.  On the device the select is spelled out (a compare into vcc and ONE
// conditional move of a float unit, widened afterwards): left to itself the compiler turns the rare correction into
// an exec-masked block around three integer operations -- two scalar instructions and a pipeline bubble per quotient,
// forty-four times per pixel.
LSQ_HD double fix_up(double r, double d, double n) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int unit = ((__double2hiint(n) ^ __double2hiint(d)) & int(0x80000000u)) | 0x3F800000;      // +-1.0f
    float c;
    asm("v_cmp_ge_f64 vcc, |%1|, |%2|\n\tv_cndmask_b32 %0, 0, %3, vcc" : "=v"(c) : "v"(r), "v"(d), "v"(unit) : "vcc");
    return double(c);
#else
    return fabs(r) >= fabs(d) ? copysign(1.0, n) * copysign(1.0, d) : 0.0;
#endif
}

// trunc(n / d) for an integer-valued n with |n| < 2^53; rs = recip_short(d).  Exact while |n / d| < 2^46.
LSQ_HD double div_trunc(double n, double d, double rs) {
    const double q0 = trunc(n * rs);
    const double r = fma(-q0, d, n);                     // exact; same sign as n, |r| < 2|d|
    return q0 + fix_up(r, d, n);
}

// trunc(a * b / d) with the product carried exactly as p + e (|a b| may exceed 2^53).
LSQ_HD double muldiv_trunc(double a, double b, double d, double rs, Guard &g) {
    const double p = a * b;
    const double e = fma(a, b, -p);                      // a*b == p + e exactly
    const double q0 = trunc(p * rs);
    const double r = fma(-q0, d, p) + e;                 // exact remainder of the estimate
    g.see_product(p);
    return q0 + fix_up(r, d, p);
}

// (v * (ab-1) + ab/2) / ab, truncating (NBLIC.c:199, :273-279); |v| < 2^44
template <int AB>
LSQ_HD double decay(double v) {
    const double n = fma(v, double(AB - 1), double(AB / 2));
    return div_trunc(n, double(AB), kShort / double(AB));
}
LSQ_HD double decay_k(double v, int k) { return k ? decay<kDecayV>(v) : decay<kDecayS>(v); }

// weight of a new sample (NBLIC.c:249-251): clip(s_sum + 4096, 4096, 65536)
LSQ_HD double sample_weight(double s_sum) {
    const double s = s_sum + double(1 << kFb1);
    return s < double(1 << kFb1) ? double(1 << kFb1) : (s > double(16 << kFb1) ? double(16 << kFb1) : s);
}

// one entry of the new sample (NBLIC.c:253-267): (prod << shift + s/2) / s with prod = (x-128)*vn_k
// (shift 28) or vn_j*vn_k (shift 18); |prod| <= 2^14
LSQ_HD double sample_entry(int prod, double scale, double s, double rs) {     // rs = recip_short(s)
    const double n = fma(double(prod), scale, floor(s * 0.5));
    return div_trunc(n, s, rs);
}
constexpr double kScaleB = 268435456.0;                  // 1 << (4 + FB1 + FB1)
constexpr double kScaleA = 262144.0;                     // 1 << (4 + FB2 + FB1)

// contribution of coefficient k to the Q12 prediction (NBLIC.c:233-236): (b*vn*4 + (d >> 1)) / d
LSQ_HD double term(double b, int vn, double d, double rs, Guard &g) {                  // rs = recip_short(d)
    const double n = fma(b, double(vn * (1 << kFb2)), floor(d * 0.5));
    g.see_quotient(n * rs);
    g.see_entry(b);
    return div_trunc(n, d, rs);
}
LSQ_HD double term(double b, int vn, double d, Guard &g) { return term(b, vn, d, recip_short(d), g); }

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

// serial_engine.hip -- the raster-serial chains of NBLIC, one wave per image (see serial_engine.h).
//
// A lone wave issues one instruction every ~4 cycles and pays ~50 cycles for every dependent LDS
// round trip, so these kernels are written for INSTRUCTION COUNT on the chain:
//   * the model of a pixel (taps, predictor, activity, context comparisons, regressors) is spread over
//     the lanes, one term each (lane_table.h, LaneFront); what is left of the scalar part runs as
//     scalar code (values every lane read alike are declared uniform), with every integer divide
//     replaced by a table or a reciprocal multiply (model.h NearParams / level_shift_table);
//   * the decoders compute a symbol's bin probabilities on the lanes too (decode_symbol);
//   * the least-squares predictor of efforts 2/3 (NBLIC.c:112-283) keeps its statistics in doubles
//     that hold the reference's integers exactly (lsq_f64.h).  The two regularised systems of a
//     pixel are solved side by side: lane r of the 16-lane row g holds row r of system g in
//     registers; the pivot is found with DPP row rotations, the pivot row travels by ds_bpermute,
//     and each elimination step is one double-carried multiply-divide per column for all rows at
//     once.  Rows never move: each carries its position, so "first maximum wins" is a key compare.
//     A pixel whose magnitudes leave the exact range (lsq::Guard) is redone with 64-bit integers;
//   * the running statistics are updated one or two entries per lane ([s | b | A] order, coalesced
//     in HBM, the next pixel's columns prefetched a pixel ahead) and handed to the row layout
//     through 1 KB of LDS.
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "lsq_f64.h"
#include "lane_table.h"
#include "model.h"
#include "serial_engine.h"

namespace nblic {

typedef long long i64;
typedef unsigned long long u64;

// One wave per block: "synchronising" means that this wave's LDS operations have completed (the lanes run in lock-step)
// and that the compiler moves no memory access across the point.  A __syncthreads() would also wait for every global
// load and store in flight (vmcnt(0)) -- the statistics prefetched for the next pixel, the stores of this one -- four
// times per pixel.
__device__ __forceinline__ void wave_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// Pointers read out of the job record are generic to the compiler; telling it they are global memory
// turns flat_load / flat_store into global_load / global_store and keeps LDS out of their waits.
#define NB_GLOBAL __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ NB_GLOBAL T *gp(T *p) { return (NB_GLOBAL T *)p; }

// ---- cross-lane moves of doubles ------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);               // every lane is written: no `old` value to carry (it would cost a copy per word)
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
constexpr int kRor1 = 0x121, kRor2 = 0x122, kRor4 = 0x124, kRor8 = 0x128;   // rotate right inside each 16-lane row
// one v_max_f64 (the library fmax wraps each operand in a canonicalising maximum of its own; the keys are never NaN)
__device__ __forceinline__ double max_f64(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double row_max(double v) {
    v = max_f64(v, dpp_f64<kRor1>(v)); v = max_f64(v, dpp_f64<kRor2>(v));
    v = max_f64(v, dpp_f64<kRor4>(v)); v = max_f64(v, dpp_f64<kRor8>(v));
    return v;
}
__device__ __forceinline__ double row_sum(double v) {
    v += dpp_f64<kRor1>(v); v += dpp_f64<kRor2>(v); v += dpp_f64<kRor4>(v); v += dpp_f64<kRor8>(v);
    return v;
}
// The value a lane of half HALF (0: lanes 0..31, 1: lanes 32..63) holds, in both halves of the wave, lane for lane:
// v_permlane32_swap_b32 swaps lanes 32..63 of its first operand with lanes 0..31 of its second, so with the same value
// in both operands the first result is the lower half's value everywhere and the second the upper half's (gfx950).
template <int HALF>
__device__ __forceinline__ double bcast_half(double v) {
    const unsigned lo = unsigned(__double2loint(v)), hi = unsigned(__double2hiint(v));
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double(int(b[HALF]), int(a[HALF]));
}
// The same among the FOUR 16-lane rows of the wave (v_permlane16_swap_b32 swaps the odd rows of its first operand with
// the even rows of its second: with the same value in both, the first result is rows [0 0 2 2], the second [1 1 3 3]).
template <int ROW>
__device__ __forceinline__ double bcast_row(double v) {
    const unsigned lo = unsigned(__double2loint(v)), hi = unsigned(__double2hiint(v));
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const unsigned tl = a[ROW & 1], th = b[ROW & 1];
    const auto c = __builtin_amdgcn_permlane32_swap(tl, tl, false, false);
    const auto d = __builtin_amdgcn_permlane32_swap(th, th, false, false);
    return __hiloint2double(int(d[ROW >> 1]), int(c[ROW >> 1]));
}
// maximum / sum over the EIGHT lanes of a half-row: xor 1, xor 2 inside the quads, then the mirror image of the half-row
constexpr int kQuadX1 = 0xB1, kQuadX2 = 0x4E, kHalfMirror = 0x141;
__device__ __forceinline__ double half_row_max(double v) {
    v = max_f64(v, dpp_f64<kQuadX1>(v)); v = max_f64(v, dpp_f64<kQuadX2>(v)); v = max_f64(v, dpp_f64<kHalfMirror>(v));
    return v;
}
__device__ __forceinline__ double half_row_sum(double v) {
    v += dpp_f64<kQuadX1>(v); v += dpp_f64<kQuadX2>(v); v += dpp_f64<kHalfMirror>(v);
    return v;
}
// value of lane (byte_addr / 4)
__device__ __forceinline__ double fetch_f64(double v, int byte_addr) {
    const int lo = __builtin_amdgcn_ds_bpermute(byte_addr, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(byte_addr, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

// P(bin==1) = floor(4096 * c1 / (c0 + c1)) (NBLIC.c:621-625): both operands are < 2^14, so the float
// reciprocal estimate is off by less than one and a single remainder check makes it exact.
__device__ __forceinline__ int prob_one(int c0, int c1) {
    const int sum = c0 + c1, n = c1 << 12;
    int q = int(float(n) * __builtin_amdgcn_rcpf(float(sum)));
    const int r = n - q * sum;
    q += (r >= sum) - (r < 0);
    return q;
}

// ---- LDS images --------------------------------------------------------------------------------
struct LsqLds {
    double D[128];                           // statistics of the pixel about to be predicted: [s | b | A] (E + F)
    int8_t vn8[16];                          // regressors 0..9 (tap - 128); [14] = 0
    int xch[3], bias_pub;                    // two-wave solve: the second wave's prediction / verdict / "redo with integers"; the regularisation strength for it
    i64 Mi[lsq::kMaxN][lsq::kMaxN + 1];      // integer redo of a pixel (rare): augmented system, terms
    i64 termi[lsq::kMaxN];
    int8_t dump[64];                         // lane-parallel front: where the lanes without a regressor store theirs
};
struct ModelLds {
    int ctx[kContexts];
    uint16_t qlut[208];                      // activity (clipped to 200) -> qu | qv << 4 | qw << 8
    uint32_t rec_ring[64];
    uint16_t pxs_ring[64];
    LsqLds q;
};
struct DecodeLds {
    // what a resumed launch reloads, in the order of the state record (serial_engine.h kDecodeStateBytes)
    int ctx[kContexts];
    uint32_t cnt[kLevels][kTreeNodes];       // c0 | c1 << 16 (both <= 8224)
    int count[512][kMapSyms];
    uint8_t rank_of[512][kMapSyms], sym_at[512][kMapSyms];
    // rebuilt by every launch
    uint16_t qlut[208];
    uint32_t sbuf[256];                      // 1 KB window of the stream, two halves
    LsqLds q;
};
static_assert(offsetof(DecodeLds, qlut) == kDecodeStateBytes - sizeof(SerialState), "state record = the leading tables");
// Many images side by side: 89 KB of LDS per image is ONE wave per CU.  The lean image keeps the re-mappers' hit counts
// (40 KB, touched twice per pixel and off the coder's chain) in the image's state record in memory -- where a resumed
// launch keeps them anyway -- and drops the symbol -> rank table the decoder only ever writes: 38 KB, four waves per CU.
struct DecodeLdsLean {
    int ctx[kContexts];
    uint32_t cnt[kLevels][kTreeNodes];
    uint8_t sym_at[512][kMapSyms];
    uint16_t qlut[208];
    uint32_t sbuf[256];
    LsqLds q;
};
static_assert(sizeof(DecodeLdsLean) <= 38 * 1024, "four lean decoder images (+ their row rings) per 160 KB of LDS");
constexpr int kRecCount = kContexts + kLevels * kTreeNodes;              // word offsets of the tables in the state record (after SerialState)
constexpr int kRecRank = kRecCount + 512 * kMapSyms, kRecSym = kRecRank + 512 * kMapSyms / 4;

// activity -> (qu, qv, qw) (model.h quantise) as a table: the interpolation divides by a level gap
__device__ void fill_qlut(uint16_t *qlut) {
    for (int d = int(threadIdx.x); d < 208; d += 64) {
        const Level L = quantise(d < 200 ? d : 200);
        qlut[d] = uint16_t(L.qu | (L.qv << 4) | (L.qw << 8));
    }
}
__device__ __forceinline__ Level level_from(uint16_t e) { return Level{e & 15, (e >> 4) & 15, e >> 8}; }

// ---- least squares, integer redo (NBLIC.c:112-161, :210-239): all 64 lanes, one system ------------
__device__ __forceinline__ i64 mulw(i64 a, i64 b) { return i64(u64(a) * u64(b)); }
__device__ __forceinline__ i64 abs64(i64 v) { return v < 0 ? -v : v; }

__device__ __noinline__ int lsq_solve_int(LsqLds &S, int n, i64 bias, i64 *px_q12) {
    const int lane = int(threadIdx.x), cols = n + 1;
    wave_sync();
    for (int e = lane; e < n * cols; e += 64) {
        const int i = e / cols, j = e - i * cols;
        S.Mi[i][j] = j == n ? i64(S.D[1 + i]) + bias * (1 << lsq::kFb3) : i64(S.D[1 + n + i * n + j]) + (i == j ? bias * n : 0);
    }
    wave_sync();
    for (int k = 0; k + 1 < n; k++) {
        int piv = k;
        i64 best = abs64(S.Mi[k][k]);
        for (int i = k + 1; i < n; i++) { const i64 v = abs64(S.Mi[i][k]); if (v > best) { best = v; piv = i; } }
        wave_sync();
        if (piv != k) {
            for (int j = lane; j < cols; j += 64) { const i64 t = S.Mi[k][j]; S.Mi[k][j] = S.Mi[piv][j]; S.Mi[piv][j] = t; }
            wave_sync();
        }
        const i64 d = S.Mi[k][k];
        if (d == 0) return 0;
        i64 upd[2]; int cnt = 0;
        for (int e = lane; e < n * cols; e += 64) {
            const int i = e / cols, j = e - i * cols;
            upd[cnt++] = (i > k && j > k) ? S.Mi[i][j] - mulw(S.Mi[k][j], S.Mi[i][k]) / d : 0;
        }
        wave_sync();
        cnt = 0;
        for (int e = lane; e < n * cols; e += 64) {
            const int i = e / cols, j = e - i * cols;
            if (i > k && j > k) S.Mi[i][j] = upd[cnt]; else if (i > k && j == k) S.Mi[i][j] = 0;
            cnt++;
        }
        wave_sync();
    }
    for (int k = n - 1; k > 0; k--) {
        const i64 d = S.Mi[k][k];
        if (d == 0) return 0;
        wave_sync();
        for (int i = lane; i < k; i += 64) { S.Mi[i][n] -= mulw(S.Mi[k][n], S.Mi[i][k]) / d; S.Mi[i][k] = 0; }
        wave_sync();
    }
    for (int k = lane; k < n; k += 64) {
        const i64 d = S.Mi[k][k];
        S.termi[k] = (mulw(mulw(S.Mi[k][n], i64(S.vn8[k])), 1 << lsq::kFb2) + (d >> 1)) / d;
    }
    wave_sync();
    i64 px = i64(kMid) << lsq::kFb1;
    for (int k = 0; k < n; k++) px += S.termi[k];
    wave_sync();
    *px_q12 = px;
    return 1;
}

// ---- least squares in registers: the pixel's two systems side by side, their columns split over the wave ----
// Lanes: R per (system, column group) -- R = 16 for N = 10 (two column groups: the half-waves), R = 8 for N = 6 (four
// column groups: the 16-lane rows) --, row = lane & (R - 1), system = (lane / R) & 1, column group cg = lane / (2 R).
// Slot s of M holds column G s + cg of the augmented row [A | b] (column N = the right-hand side; columns beyond it
// do not exist and their slots stay 0), so an elimination step costs ceil((N - k) / G) multiply-divides instead of
// N - k.  Per step the pivot column is handed to the other column groups with v_permlane32_swap / v_permlane16_swap
// (no LDS), every group then finds the pivot for itself (DPP) and fetches the pivot row's entries of ITS columns
// (ds_bpermute inside the group).  Rows never move: each carries its position.  The broadcast columns are kept: once
// a row has been placed its entries no longer change, so column k as seen at step k is what the back substitution
// needs above the diagonal.
// Returns the Q12 prediction of the lane's system in every lane of its group; ok = 0: a pivot was zero (NBLIC.c:118).
// WAVES = 2 (effort 3): the pixel's two systems are solved by the two WAVES of the image's workgroup, one system each
// (model_body); a wave then has all four 16-lane rows for its system's columns.
template <int N_, int WAVES_>
struct SplitLayout {
    static constexpr int N = N_, WAVES = WAVES_;
    static constexpr int SYS = WAVES == 2 ? 1 : 2;                      // systems a wave solves side by side
    static constexpr int R = (WAVES == 1 && N <= 8) ? 8 : 16;           // lanes per (system, column group)
    static constexpr int G = 64 / (SYS * R);                            // column groups
    static constexpr int kS = (N + G) / G;                              // column slots per lane: ceil((N + 1) / G)
};
template <class L, int CG>
__device__ __forceinline__ double bcast_group(double v) {
    if constexpr (L::G == 2) return bcast_half<CG>(v);
    else return bcast_row<CG>(v);
}
template <class L>
__device__ __forceinline__ double group_max(double v) { if constexpr (L::R == 16) return row_max(v); else return half_row_max(v); }
template <class L>
__device__ __forceinline__ double group_sum(double v) { if constexpr (L::R == 16) return row_sum(v); else return half_row_sum(v); }
// column k of the system, from the group that owns it, in every group (k is a constant after unrolling)
template <class L, int K>
__device__ __forceinline__ double column_everywhere(const double (&M)[L::kS]) {
    return bcast_group<L, K % L::G>(M[K / L::G]);
}

template <class L, int K>
__device__ __forceinline__ void eliminate_step(double (&M)[L::kS], double (&col)[L::N], int (&at)[L::N], int &at_sum, int &pos, double &diag, int &ok,
                                               const int live, const int row, const int cg, const int group_base4, lsq::Guard &g) {
    constexpr int k = K;
    const double ck = column_everywhere<L, K>(M);
    col[k] = ck;
    // every lane prepares the reciprocal of ITS candidate while the pivot search runs: the winner's is fetched with
    // its value, and the reciprocal's dependent chain is off the step's critical path
    const double raw_c = lsq::recip_raw(ck);
    // pivot: largest |entry| of column k among the rows at positions >= k, first position wins (NBLIC.c:121-127)
    const double key = group_max<L>((live & (pos >= k)) ? fma(fabs(ck), 256.0, double((15 - pos) * 16 + row)) : -1.0);
    const int tag = int(fma(-256.0, floor(key * (1.0 / 256.0)), key));
    const int c = tag & 15, pc = 15 - (tag >> 4);
    const int src = group_base4 | (c << 2);
    constexpr int s0 = (k + 1) / L::G;                                   // first slot with a column beyond k in some group
    const double d = fetch_f64(ck, src);
    const lsq::Recip rc = lsq::recip_of_raw(fetch_f64(raw_c, src));
    double prow[L::kS];                                                  // the pivot row's entries of this lane's columns: all requests go out together
#pragma unroll
    for (int s = s0; s < L::kS; s++) prow[s] = fetch_f64(M[s], src);
    at[k] = c; at_sum += c;
    pos = pos == k ? pc : pos;                                           // the row that sat at k takes the pivot's place
    pos = (live & (row == c)) ? k : pos;
    diag = pos == k ? d : diag;
    ok &= int(d != 0.0);
    const double l = (live & (pos > k)) ? ck : 0.0;                      // rows already placed take no part: their quotient is 0
#pragma unroll
    for (int s = s0; s < L::kS; s++) {
        // slot s0 may still hold columns <= k in the lower groups: they are done with (the reference never reads them again)
        const double ls = (s == s0 && L::G * s0 <= k) ? ((L::G * s0 + cg > k) ? l : 0.0) : l;
        M[s] -= lsq::muldiv_trunc(prow[s], ls, d, rc, g);
        g.see_entry(M[s]);
    }
}
template <class L, int K>
__device__ __forceinline__ void eliminate_from(double (&M)[L::kS], double (&col)[L::N], int (&at)[L::N], int &at_sum, int &pos, double &diag, int &ok,
                                               const int live, const int row, const int cg, const int group_base4, lsq::Guard &g) {
    if constexpr (K + 1 < L::N) {
        eliminate_step<L, K>(M, col, at, at_sum, pos, diag, ok, live, row, cg, group_base4, g);
        eliminate_from<L, K + 1>(M, col, at, at_sum, pos, diag, ok, live, row, cg, group_base4, g);
    }
}

template <class L>
__device__ __forceinline__ double lsq_solve_split(double (&M)[L::kS], const int row, const int cg, const int group_base4,
                                                  const int8_t *vn8, lsq::Guard &g, int &ok) {
    constexpr int N = L::N;
    // (flags are ints in vector registers: as booleans they would each pin a scalar register pair for the whole solve)
    const int live = row < N;
    int pos = row, at_sum = 0;
    int at[N];
    double col[N];
    double diag = 1.0;
    ok = 1;
    eliminate_from<L, 0>(M, col, at, at_sum, pos, diag, ok, live, row, cg, group_base4, g);
    at[N - 1] = N * (N - 1) / 2 - at_sum;
    col[N - 1] = column_everywhere<L, N - 1>(M);
    diag = (live & (pos == N - 1)) ? col[N - 1] : diag;
    double rhs = column_everywhere<L, N>(M);                             // every group finishes the solve alike
    g.see_pivot(diag);                                                   // every divisor of the solve is some row's diagonal entry
    const double raw_own = lsq::recip_raw(diag);                         // every row's own reciprocal at once, fetched below
#pragma unroll
    for (int k = N - 1; k > 0; k--) {                                     // back substitution on the right-hand side (NBLIC.c:148-158)
        const int src = group_base4 | (at[k] << 2);
        const double d = fetch_f64(diag, src), bk = fetch_f64(rhs, src);
        const lsq::Recip rcd = lsq::recip_of_raw(fetch_f64(raw_own, src));
        ok &= int(d != 0.0);
        const double l = (live & (pos < k)) ? col[k] : 0.0;
        rhs -= lsq::muldiv_trunc(bk, l, d, rcd, g);
        g.see_entry(rhs);
    }
    const int v = vn8[live ? pos : 14];
    const double t = lsq::term(live ? rhs : 0.0, v, diag, lsq::recip_of_raw(raw_own), g);     // NBLIC.c:233-236
    return double(kMid << lsq::kFb1) + group_sum<L>(live ? t : 0.0);
}

// Per-lane description of the one or two statistics entries a lane maintains ([s | b(n) | A(n x n)] order).
template <int N>
struct LsqEntries {
    static constexpr int kM = 1 + N + N * N, kSlots = kM > 64 ? 2 : 1, kStride = kM > 64 ? 128 : 64;
    int ia[kSlots], ib[kSlots];              // vn8 indices whose product is the entry's sample (b: x' x vn, A: vn x vn)
    double scale[kSlots], abm1[kSlots], abh[kSlots], rab[kSlots];
    bool active[kSlots];
    __device__ void init(int lane) {
#pragma unroll
        for (int s = 0; s < kSlots; s++) {
            const int k = lane + 64 * s;
            active[s] = k < kM;
            const int kk = active[s] ? k : 0;
            if (kk == 0) { ia[s] = 14; ib[s] = 14; }
            else if (kk <= N) { ia[s] = 15; ib[s] = kk - 1; }
            else { ia[s] = (kk - 1 - N) / N; ib[s] = (kk - 1 - N) - ia[s] * N; }
            scale[s] = kk <= N ? lsq::kScaleB : lsq::kScaleA;
            const int a = kk ? lsq::kDecayV : lsq::kDecayS;
            abm1[s] = double(a - 1); abh[s] = double(a / 2); rab[s] = 1.0 / double(a);
        }
    }
    __device__ __forceinline__ double decay(double v, int s) const { return trunc(lsq::half_toward(fma(v, abm1[s], abh[s])) * rab[s]); }   // lsq::decay with the lane's constants
};

// once per row: right-to-left accumulation of the column sums (NBLIC.c:186-204); a lane owns its entries
template <int N>
__device__ void lsq_row_prepare(const LsqEntries<N> &en, NB_GLOBAL double *F, NB_GLOBAL const double *B, int w, int lane) {
    using T = LsqEntries<N>;
#pragma unroll
    for (int s = 0; s < T::kSlots; s++) {
        if (!en.active[s]) continue;
        const size_t k = size_t(lane + 64 * s);
        double carry = 0.0;
        int j = w - 1;
        for (; j >= 3; j -= 4) {                                          // four loads in flight per step of the chain
            const double b0 = B[size_t(j) * T::kStride + k], b1 = B[size_t(j - 1) * T::kStride + k];
            const double b2 = B[size_t(j - 2) * T::kStride + k], b3 = B[size_t(j - 3) * T::kStride + k];
            double f = carry + b0; F[size_t(j) * T::kStride + k] = f; carry = en.decay(f, s);
            f = carry + b1; F[size_t(j - 1) * T::kStride + k] = f; carry = en.decay(f, s);
            f = carry + b2; F[size_t(j - 2) * T::kStride + k] = f; carry = en.decay(f, s);
            f = carry + b3; F[size_t(j - 3) * T::kStride + k] = f; carry = en.decay(f, s);
        }
        for (; j >= 0; j--) { const double f = carry + B[size_t(j) * T::kStride + k]; F[size_t(j) * T::kStride + k] = f; carry = en.decay(f, s); }
    }
}

// The least-squares state of one image walk, shared by the encoder's model kernel and the decoder.
// WAVES = 1: one wave does everything, its two systems side by side (predict).  WAVES = 2: the image's workgroup has
// two waves; wave 0 ("main") owns the statistics, the model and system 0, wave 1 solves system 1 and nothing else
// (solve_one); the caller's barriers order the hand-overs through LsqLds (D, vn8, bias_pub in; xch out).
template <int N, int WAVES = 1>
struct LsqWalk {
    using T = LsqEntries<N>;
    using L = SplitLayout<N, WAVES>;
    static constexpr int kS = L::kS;                                     // column slots per lane (lsq_solve_split)
    LsqEntries<N> en;
    double E[T::kSlots], Bj[T::kSlots], Fj[T::kSlots], Bn[T::kSlots], Fn[T::kSlots];
    int ra[T::kSlots], rb[T::kSlots];                                    // the regressors the lane's entries multiply (read while the systems are solved)
    NB_GLOBAL double *Bst, *Fst;
    int bias, b1, b2, lane, row, cg, sys, group_base4, w;
    int d_base, d_last, diag_slot, last_kind;                            // where the lane's slots sit in S.D; which slot (if any) is on the diagonal; last slot: 0 matrix column, 1 right-hand side, 2 nothing
    int p1, p2;                                                          // Q12 predictions (<= 255 << 12)
    bool ok1, ok2;

    __device__ void init(double *stats, int w_, int lane_, int wave_, int bias_) {
        lane = lane_; w = w_; row = lane & (L::R - 1);
        sys = L::SYS == 2 ? (lane / L::R) & 1 : wave_;
        cg = lane / (L::SYS * L::R); group_base4 = (lane & ~(L::R - 1)) << 2;
        Bst = gp(stats); Fst = gp(stats) + size_t(w) * T::kStride;
        bias = bias_;
        en.init(lane);
        const int r = row < N ? row : N - 1;                             // the idle lanes of a row mirror its last system row (they never take part)
        d_base = 1 + N + r * N + cg;                                       // slot s: column G s + cg
        const int last_col = L::G * (kS - 1) + cg;
        last_kind = last_col < N ? 0 : (last_col == N ? 1 : 2);
        d_last = last_kind == 0 ? d_base + L::G * (kS - 1) : 1 + r;       // (an address inside S.D in every case)
        diag_slot = (row < N && row >= cg && (row - cg) % L::G == 0) ? (row - cg) / L::G : -1;
    }
    __device__ __forceinline__ void load_cols(int j, double (&b)[T::kSlots], double (&f)[T::kSlots]) const {
        const int jj = j < w ? j : w - 1;                                 // the prefetch past the row end re-reads the last column
#pragma unroll
        for (int s = 0; s < T::kSlots; s++) {
            const size_t at = size_t(jj) * T::kStride + size_t(en.active[s] ? lane + 64 * s : 0);
            b[s] = Bst[at]; f[s] = Fst[at];
        }
    }
    __device__ void row_begin(LsqLds &S) {                               // (the main wave only)
        lsq_row_prepare<N>(en, Fst, Bst, w, lane);
        __threadfence_block();
        wave_sync();
        load_cols(0, Bj, Fj);
#pragma unroll
        for (int s = 0; s < T::kSlots; s++) { E[s] = 0.0; if (en.active[s]) S.D[lane + 64 * s] = Fj[s]; }
        S.bias_pub = bias;
        wave_sync();
    }
    // the lane's slots of the system regularised with strength bs, from S.D; then the solve
    __device__ __forceinline__ double solve_with(LsqLds &S, int bs, lsq::Guard &g, int &ok) const {
        const double reg = double(bs * N);
        double M[kS];
#pragma unroll
        for (int s = 0; s + 1 < kS; s++) {                                // slots 0 .. kS-2 are matrix columns in every column group
            M[s] = S.D[d_base + L::G * s];
            M[s] = diag_slot == s ? M[s] + reg : M[s];
        }
        {   // the last slot: a matrix column, the right-hand side, or nothing, depending on the lane's column group
            const double v = S.D[d_last];
            const double add = last_kind == 1 ? double(bs << lsq::kFb3) : (diag_slot == kS - 1 ? reg : 0.0);
            M[kS - 1] = last_kind == 2 ? 0.0 : v + add;
        }
        const double p = lsq_solve_split<L>(M, row, cg, group_base4, S.vn8, g, ok);
        return p < 0.0 ? 0.0 : (p > double(kMaxVal << lsq::kFb1) ? double(kMaxVal << lsq::kFb1) : p);
    }
    // the pixel's regressors stand in S.vn8: the lane's factors, long before update() needs them (x' itself comes in a register)
    __device__ __forceinline__ void fetch_factors(const LsqLds &S) {
#pragma unroll
        for (int s = 0; s < T::kSlots; s++) { ra[s] = S.vn8[en.ia[s] == 15 ? 14 : en.ia[s]]; rb[s] = S.vn8[en.ib[s]]; }
    }
    __device__ __forceinline__ int clamp_q12(i64 q) const { const i64 top = i64(kMaxVal) << lsq::kFb1; return int(q < 0 ? 0 : (q > top ? top : q)); }
    // WAVES = 1: both predictions of pixel j from S.D and S.vn8 (complete and synchronised)
    __device__ __forceinline__ void predict(LsqLds &S, int j) {
        load_cols(j + 1, Bn, Fn);                                         // next pixel's columns: a whole pixel ahead of their use
        fetch_factors(S);
        lsq::bias_pair(bias, b1, b2);
        lsq::Guard g;
        int ok;
        const int pi = int(solve_with(S, sys ? b2 : b1, g, ok));
        const u64 bad = __ballot(!g.ok());
        p1 = __builtin_amdgcn_readlane(pi, 0); p2 = __builtin_amdgcn_readlane(pi, L::R);      // system 0 / system 1 of column group 0
        const u64 okm = __ballot(ok != 0);
        ok1 = (okm & 1ull) != 0; ok2 = ((okm >> L::R) & 1ull) != 0;
        if (bad) {                                                       // magnitudes left the exact range: integers decide (rare)
            i64 q1 = 0, q2 = 0;                                          // (the results are the same on every lane: say so, or everything downstream is compiled as divergent)
            ok1 = __builtin_amdgcn_readfirstlane(lsq_solve_int(S, N, b1, &q1)) != 0;
            ok2 = __builtin_amdgcn_readfirstlane(lsq_solve_int(S, N, b2, &q2)) != 0;
            p1 = __builtin_amdgcn_readfirstlane(clamp_q12(q1)); p2 = __builtin_amdgcn_readfirstlane(clamp_q12(q2));
        }
    }
    // WAVES = 2: this wave's ONE system.  The main wave keeps p1 / ok1 (and redoes its system with integers at once if
    // it has to: the prediction itself depends on it); the other wave leaves prediction, verdict and "redo me" in S.xch
    __device__ __forceinline__ void solve_one(LsqLds &S, int j, bool main) {
        if (main) { load_cols(j + 1, Bn, Fn); fetch_factors(S); }
        lsq::bias_pair(main ? bias : S.bias_pub, b1, b2);
        lsq::Guard g;
        int ok;
        const int pi = int(solve_with(S, main ? b1 : b2, g, ok));
        const bool bad = __ballot(!g.ok()) != 0;
        const int mine = __builtin_amdgcn_readlane(pi, 0);
        const bool ok_mine = (__ballot(ok != 0) & 1ull) != 0;
        if (main) {
            p1 = mine; ok1 = ok_mine;
            if (bad) { i64 q1 = 0; ok1 = __builtin_amdgcn_readfirstlane(lsq_solve_int(S, N, b1, &q1)) != 0; p1 = __builtin_amdgcn_readfirstlane(clamp_q12(q1)); }
        } else {
            S.xch[0] = mine; S.xch[1] = int(ok_mine); S.xch[2] = int(bad);
        }
    }
    // WAVES = 2, main wave, after the barrier that follows solve_one: the other system's result
    __device__ __forceinline__ void take_other(LsqLds &S) {
        const int x0 = S.xch[0], x1 = S.xch[1], x2 = S.xch[2];           // one round trip for the three
        p2 = __builtin_amdgcn_readfirstlane(x0); ok2 = __builtin_amdgcn_readfirstlane(x1) != 0;
        if (__builtin_amdgcn_readfirstlane(x2)) { i64 q2 = 0; ok2 = __builtin_amdgcn_readfirstlane(lsq_solve_int(S, N, b2, &q2)) != 0; p2 = __builtin_amdgcn_readfirstlane(clamp_q12(q2)); }
    }
    // fold the coded pixel in (NBLIC.c:242-283, :882-893) and publish the next pixel's statistics
    __device__ __forceinline__ void update(LsqLds &S, int j, int xr, int p1_used) {
        const int xq = xr << lsq::kFb1;
        const double s_curr = double(iabs(p1_used - xq));
        const double e0 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(E[0]), 0), __builtin_amdgcn_readlane(__double2loint(E[0]), 0));
        const double f0 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(Fj[0]), 0), __builtin_amdgcn_readlane(__double2loint(Fj[0]), 0));
        const double s_sum = (e0 + f0) + floor(s_curr * 1.5);
        const double sw = lsq::sample_weight(s_sum), rs = lsq::recip_raw(sw);
#pragma unroll
        for (int s = 0; s < T::kSlots; s++) {
            const int prod = (en.ia[s] == 15 ? xr - kMid : ra[s]) * rb[s];
            double sample = lsq::sample_entry(prod, en.scale[s], sw, rs);
            if (s == 0) sample = lane == 0 ? s_curr : sample;
            const double b = en.decay(Bj[s], s) + sample;
            if (en.active[s]) Bst[size_t(j) * T::kStride + size_t(lane + 64 * s)] = b;
            E[s] = en.decay(E[s], s) + b;
        }
        if (ok1 && ok2) bias = (iabs(p1 - xq) > iabs(p2 - xq)) ? b2 : b1;
        if (WAVES == 2) S.bias_pub = bias;
#pragma unroll
        for (int s = 0; s < T::kSlots; s++) {
            if (en.active[s]) S.D[lane + 64 * s] = E[s] + Fn[s];
            Bj[s] = Bn[s]; Fj[s] = Fn[s];
        }
    }
};

// pack the ten regressors (a,b,c,d,e,f,t,h,q,g, NBLIC.c:164-183) as bytes; every lane stores the same three words
// (one LDS write each, no divergence)
__device__ __forceinline__ void store_regressors(int8_t *vn8, const Taps &t) {
    auto b = [](int v) { return uint32_t(v - kMid) & 0xFFu; };
    uint32_t *w = reinterpret_cast<uint32_t *>(vn8);
    w[0] = b(t.a) | (b(t.b) << 8) | (b(t.c) << 16) | (b(t.d) << 24);
    w[1] = b(t.e) | (b(t.f) << 8) | (b(t.t) << 16) | (b(t.h) << 24);
    w[2] = b(t.q) | (b(t.g) << 8);
}

// ---- causal taps from the LDS row ring, as a sliding window --------------------------------------
// The twelve RAW neighbours (columns clamped into the row, so every address is valid; rows above the
// image read whatever the ring holds) live in registers and shift by one column per pixel; only the two
// right-most ones (rows i-1 and i-2, column j+2) are new, and they are requested one pixel AHEAD, so a
// pixel never waits for its taps.  The reference's fall-back chain (NBLIC.c:287-304) is applied to the raw
// values with selects.
struct TapWindow {
    int A, E, B, C, D, Q, T, F, G, H, R, S;          // raw: a e / b c d q t / f g h r s
    int Tn, Rn;                                      // column j+3 of rows i-1 / i-2, in flight
    __device__ __forceinline__ void row_start(const uint8_t *r0, const uint8_t *r1, const uint8_t *r2, int w) {
        auto cl = [w](int c) { return c < 0 ? 0 : (c >= w ? w - 1 : c); };
        A = r0[0]; E = r0[0];
        B = r1[0]; C = r1[0]; D = r1[cl(1)]; Q = r1[0]; T = r1[cl(2)];
        F = r2[0]; H = r2[0]; G = r2[cl(1)]; S = r2[0]; R = r2[cl(2)];
        Tn = r1[cl(3)]; Rn = r2[cl(3)];
    }
    // after pixel j has been reconstructed as xr: the window of pixel j+1; requests column j+4
    __device__ __forceinline__ void advance(const uint8_t *r1, const uint8_t *r2, int w, int j, int xr) {
        E = j >= 1 ? A : xr; A = xr;                                   // at j == 0 the raw A was never a coded pixel
        Q = C; C = B; B = D; D = T; T = Tn;
        S = H; H = F; F = G; G = R; R = Rn;
        const int c = j + 4 < w ? j + 4 : w - 1;
        Tn = r1[c]; Rn = r2[c];
    }
    __device__ __forceinline__ Taps taps(int w, int i, int j) const {
        const bool up1 = i >= 1, up2 = i >= 2, l1 = j >= 1, l2 = j >= 2, r1_ = j + 1 < w, r2_ = j + 2 < w;
        Taps n;
        int a = l1 ? A : kMid, b = up1 ? B : kMid;
        if (!up1) b = a; else if (!l1) a = b;
        n.a = a; n.b = b;
        n.e = l2 ? E : a;
        n.c = (up1 && l1) ? C : b;
        n.d = (up1 && r1_) ? D : b;
        n.f = up2 ? F : b;
        n.g = (up2 && r1_) ? G : n.f;
        n.h = (up2 && l1) ? H : n.f;
        n.q = (up1 && l2) ? Q : n.c;
        n.r = (up2 && r2_) ? R : n.g;
        n.s = (up2 && l2) ? S : n.h;
        n.t = (up1 && r2_) ? T : n.d;
        return n;
    }
};

// ---- the pixel's model spread over the lanes (rows >= 2 of an image whose rows fit in LDS) ---------
// Every cost of the seven-direction predictor, every term of the activity and every comparison of the context address
// has the shape  2 X - Y - Z  over the taps (|a - e| is |2a - e - e| / 2; tap f is 2f - f - 0), so a lane takes ONE term:
// its three operands come from the rows above in LDS at the lane's own offsets (read a pixel ahead; the rows carry two
// columns of margin with the edge pixel repeated, which IS the reference's fall-back chain for rows >= 2, NBLIC.c:287-304
// / QNBLIC.c:48-79), the two taps of the row being coded (a, e) are added in from registers, and what took ~250 scalar
// instructions per pixel is
//   lanes  0..27  the 7 x 4 cost terms (direction d in quad d): one absolute difference, two quad adds, a minimum
//                 over keys (cost << 3 | direction: the first minimum wins, NBLIC.c:347-353) and a sum over the rows;
//   lanes 28..33  the six terms of the activity (NBLIC.c:376), twice their value, in quads 7 and 8;
//   lanes 36..43  the eight comparison values of the context address (NBLIC.c:398-410), bit 0 first, compared with the
//                 prediction in one instruction whose lane mask IS the address byte;
//   lanes 44..53  (NBLIC) the ten regressors of the least squares (NBLIC.c:164-183): each lane writes its own byte;
// the thresholds of QNBLIC's level (11) and of the blend weight (7 / 8) are compared lane-wise too and counted with s_bcnt1.
__device__ const QLaneTable kQLanes = make_lanes(true);
__device__ const QLaneTable kNLanes = make_lanes(false);

template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true); }
__device__ __forceinline__ uint32_t mul_u24(uint32_t a, uint32_t b) {          // full rate; v_mul_lo_u32 is a quarter of it
    uint32_t r;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ int min_u(int a, int b) { return int(unsigned(a) < unsigned(b) ? unsigned(a) : unsigned(b)); }

// The NBLIC front of a pixel on the lanes: taps, regressors, activity, predictor, context address.
struct LaneFront {
    QLaneConst lc;
    const uint8_t *pX, *pY, *pZ;
    int8_t *pR;                              // where the lane's regressor byte goes (a dump byte for the other lanes)
    int stX, stY, stZ;
    int X, Y, Z, Xn, Yn, Zn;
    int a, e;                                // the two taps of the row being coded (uniform)
    int V, Q;                                // the lane's 2X - Y - Z; its quad's sum of |V|
    __device__ __forceinline__ void init(int lane, LsqLds *q) {
        lc = kNLanes.l[lane];
        pR = q ? (lc.dst >= 0 ? &q->vn8[lc.dst] : &q->dump[lane]) : nullptr;
    }
    // rows: the ring's base (pixel 0 of slot 0); r1 / r2: rows i-1 / i-2
    __device__ __forceinline__ void row_start(uint8_t *rows, uint8_t *r1, uint8_t *r2, int w, int lane) {
        if (lane < 8) {                                                  // the edge pixels, two columns out
            uint8_t *r = lane < 4 ? r1 : r2;
            const int k = lane & 3;
            r[k < 2 ? -1 - k : w + k - 2] = r[k < 2 ? 0 : w - 1];
        }
        wave_sync();
        const uint8_t *zero = rows - 4;
        pX = lc.sel[0] == 0 ? zero : (lc.sel[0] == 1 ? r1 : r2) + lc.dx[0];
        pY = lc.sel[1] == 0 ? zero : (lc.sel[1] == 1 ? r1 : r2) + lc.dx[1];
        pZ = lc.sel[2] == 0 ? zero : (lc.sel[2] == 1 ? r1 : r2) + lc.dx[2];
        stX = lc.sel[0] != 0; stY = lc.sel[1] != 0; stZ = lc.sel[2] != 0;
        a = r1[0]; e = a;                                                // NBLIC.c:287-304 at column 0: a = b, e = a
        Xn = *pX; Yn = *pY; Zn = *pZ;
    }
    __device__ __forceinline__ void begin(int) {
        X = Xn; Y = Yn; Z = Zn;
        pX += stX; pY += stY; pZ += stZ;
        Xn = *pX; Yn = *pY; Zn = *pZ;                                    // the next pixel's operands
        const int s_yz = __mul24(e, int(lc.ce)) + (Y + Z);
        V = (X << 1) + __mul24(a, int(lc.a2)) - s_yz;
        Q = V < 0 ? -V : V;
        Q += dpp_i32<kQuadX1>(Q); Q += dpp_i32<kQuadX2>(Q);
    }
    __device__ __forceinline__ void regressors() const { *pR = int8_t((V >> 1) - kMid); }
    __device__ __forceinline__ int activity(int err) const {
        return ((__builtin_amdgcn_readlane(Q, 28) + __builtin_amdgcn_readlane(Q, 32)) >> 1) + 2 * iabs(err);
    }
    __device__ __forceinline__ int predict() const {                     // model.h predict
        const int b = __builtin_amdgcn_readlane(X, 2), c = __builtin_amdgcn_readlane(X, 1), d = __builtin_amdgcn_readlane(X, 3);
        const int f = __builtin_amdgcn_readlane(Y, 6);
        int key = (Q << 3) | lc.key_or, sum = Q & lc.sum_and;
        key = min_u(key, dpp_i32<kRor4>(key)); key = min_u(key, dpp_i32<kRor8>(key));
        sum += dpp_i32<kRor4>(sum); sum += dpp_i32<kRor8>(sum);
        const int k_min = min_u(__builtin_amdgcn_readlane(key, 0), __builtin_amdgcn_readlane(key, 16));
        const int total = __builtin_amdgcn_readlane(sum, 0) + __builtin_amdgcn_readlane(sum, 16);
        const int best = k_min >> 3, dir = k_min & 7;
        const int spread = total - 7 * best;
        const int wt = __builtin_popcountll(__ballot(int(lc.thr_weight) <= spread));
        const int angv = __mul24(a, int(lc.ca)) + __mul24(b, int(lc.cb)) + __mul24(c, int(lc.cc)) + __mul24(d, int(lc.cd));
        const int ang = __builtin_amdgcn_readlane(angv, dir);
        const int lin = __builtin_amdgcn_readfirstlane(iclip(9 * (a + b) + 2 * (d - c) - e - f, 0, 16 * kMaxVal));
        return (8 * wt * ang + (8 - wt) * lin + 64) >> 7;
    }
    __device__ __forceinline__ int context(int qu, int px0) const {      // model.h context_address
        return ((qu >> 1) << 8) | int((__ballot((px0 << int(lc.sh)) > V) >> 36) & 0xFFull);
    }
    __device__ __forceinline__ void advance(int j, int xr) { e = j >= 1 ? a : xr; a = xr; }
};

// The same interface on one lane's registers (rows 0 and 1; images whose rows do not fit in LDS)
template <bool CACHED, class Pix>
struct ScalarFront {
    Pix pix;
    const uint8_t *r1, *r2;
    int w, i;
    TapWindow tw;
    Taps t;
    int8_t *vn8;
    __device__ __forceinline__ void begin(int j) { t = CACHED ? tw.taps(w, i, j) : sample_taps(pix, w, i, j); }
    __device__ __forceinline__ void regressors() const { store_regressors(vn8, t); }
    __device__ __forceinline__ int activity(int err) const { return nblic::activity(t, err); }
    __device__ __forceinline__ int predict() const { return nblic::predict(t); }
    __device__ __forceinline__ int context(int qu, int px0) const { return context_address(t, qu, px0); }
    __device__ __forceinline__ void advance(int j, int xr) { if (CACHED) tw.advance(r1, r2, w, j, xr); }
};

// ---- encoder: the serial model stage (prediction, context bias, quantisation) ------------------
// CACHED: the three rows the taps can touch live in LDS (a ring, row r at r % 3), so a pixel's twelve
// taps are twelve LDS reads issued together; otherwise (rows wider than the LDS left over) they come
// from the reconstruction in memory.  The two variants are separate code: one generic accessor would
// turn every tap into a flat load with a branch and a full wait of its own.
// A launch works on rows [i0, i1) of the image (serial_engine.h: resumable launches).
// WAVES = 2 (effort 3): the image's workgroup has a second wave whose only job is the pixel's SECOND system -- two
// 10 x 10 eliminations are two thirds of a pixel, they are independent, and the second one's result is not needed
// before the regularisation strength of the NEXT pixel is chosen.  With a wave to itself a system has four column
// groups instead of two (19 + 9 multiply-divides per pixel on the main wave's path instead of 34 + 9).  Per pixel two
// workgroup barriers: one when the statistics and regressors of the pixel stand in LDS (the second wave is waiting
// there), one before the main wave picks up the second wave's prediction (it has long been left there: the main wave
// still had the whole context model to do).
__device__ __forceinline__ void block_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int N, bool CACHED, int WAVES>
__device__ __forceinline__ void model_body(ModelLds &S, uint8_t *rows, const SerialJob &J, const int rs, const int i0, const int i1, int &bias_io) {
    // (the wave number is the same on every lane: read it as a scalar, or `main` and everything under it is compiled as divergent)
    const int w = J.w, lane = int(threadIdx.x) & 63, wave = WAVES == 1 ? 0 : __builtin_amdgcn_readfirstlane(int(threadIdx.x)) >> 6;
    const bool main = wave == 0;
    // One wave per image: what every lane has read alike is declared uniform, and the rest of the pixel compiles to scalar
    // code (-3 % per pixel at effort 2).  With a second wave sharing the LDS the same change measured +1..2 %: left as it was.
    auto uni = [](int v) { return WAVES == 1 ? __builtin_amdgcn_readfirstlane(v) : v; };
    const NearParams np = near_params(J.near);
    const auto img = gp(J.img);
    const auto recon = gp(J.recon);
    const auto rec1 = gp(J.rec1);
    const auto pxs = gp(J.pxs);
    LsqWalk<N, WAVES> lw;
    if constexpr (N > 0) lw.init(J.stats, w, lane, wave, bias_io);
    LaneFront lf;
    lf.init(lane, &S.q);
    if (main && CACHED && i0 > 0) {                                      // resuming: the two rows above come back from the reconstruction
        const auto prev = J.recon ? gp(const_cast<const uint8_t *>(J.recon)) : img;      // lossless: the reconstruction IS the input
        for (int r = i0 > 1 ? i0 - 2 : i0 - 1; r < i0; r++) {
            uint8_t *dst = rows + (r % 3) * rs;
            for (int c = lane; c < w; c += 64) dst[c] = prev[size_t(r) * size_t(w) + c];
        }
        wave_sync();
    }

    for (int i = i0; i < i1; i++) {
        uint8_t *r0 = rows + (i % 3) * rs, *r1 = rows + ((i + 2) % 3) * rs, *r2 = rows + ((i + 1) % 3) * rs;
        const size_t row_at = size_t(i) * size_t(w), out_at = size_t(i - J.out_row0) * size_t(w);
        if (main && CACHED) {                                            // the row's ORIGINAL pixels; each is replaced by its reconstruction once coded
            for (int c = lane; c < w; c += 64) r0[c] = img[row_at + c];
            wave_sync();
        }
        auto pix = [&](int r, int c) {
            if (CACHED) return int((r == i ? r0 : (r == i - 1 ? r1 : r2))[c]);
            return int(recon[size_t(r) * size_t(w) + size_t(c)]);
        };
        if constexpr (N > 0) { if (main) lw.row_begin(S.q); }
        int err = 0;
        int x_next = 0;
        if (main && CACHED) x_next = r0[0];
        auto run_row = [&](auto &front) {
        for (int j = 0; j < w; j++) {
            int x = 0;
            if (main) {
                front.begin(j);
                x = uni(CACHED ? x_next : int(img[row_at + j]));
                if (CACHED) x_next = r0[j + 1 < w ? j + 1 : j];            // the next original pixel: requested a pixel ahead
                if constexpr (N > 0) front.regressors();
            }
            int px0 = 0, p1_used = 0;
            if constexpr (N > 0) {
                if constexpr (WAVES == 2) { block_sync(); lw.solve_one(S.q, j, main); }
                else { wave_sync(); lw.predict(S.q, j); }
            }
            int xr = 0;
            if (main) {
                if constexpr (N > 0) {
                    if (lw.ok1) { px0 = (lw.p1 + (1 << (lsq::kFb1 - 1))) >> lsq::kFb1; p1_used = lw.p1; }
                    else { px0 = front.predict(); p1_used = px0 << lsq::kFb1; }
                } else {
                    px0 = front.predict();
                }
                const int delta = front.activity(err);
                const Level L = level_from(uint16_t(uni(S.qlut[delta < 200 ? delta : 200])));
                const int adr = front.context(L.qu, px0);
                const int v = uni(S.ctx[adr]);                            // every lane reads the same word: the rest of the pixel is scalar code
                const int sign = bias_sign(v), px = bias_apply(v, px0);
                // what the decoder will reconstruct, without the symbol (model.h reconstruct_pixel; the entropy stages recompute
                // the symbol from the image and px | sign): lossless it is the pixel itself
                xr = np.near == 0 ? x : __builtin_amdgcn_readfirstlane(reconstruct_pixel(x, px, np));
                err = clip_err(xr, px0);
                S.ctx[adr] = bias_update(v, err);
                if (CACHED) r0[j] = uint8_t(xr); else { recon[row_at + j] = uint8_t(xr); __threadfence_block(); }
                front.advance(j, xr);
                S.rec_ring[j & 63] = pack_s1(px0, adr, L);
                S.pxs_ring[j & 63] = uint16_t(px | (sign << 8));
                if ((j & 63) == 63 || j == w - 1) {                      // a lane per record: coalesced stores
                    const int base = j & ~63;
                    if (base + lane <= j) { rec1[out_at + base + lane] = S.rec_ring[lane]; pxs[out_at + base + lane] = S.pxs_ring[lane]; }
                }
            }
            if constexpr (N > 0) {
                if constexpr (WAVES == 2) block_sync();                   // the other wave's verdict stands in S.q.xch
                if (main) {
                    if constexpr (WAVES == 2) lw.take_other(S.q);
                    lw.update(S.q, j, xr, p1_used);
                    wave_sync();
                }
            }
        }
        };
        if (CACHED && i >= 2) {
            if (main) lf.row_start(rows, r1, r2, w, lane);
            run_row(lf);
        } else {
            ScalarFront<CACHED, decltype(pix)> sf{pix, r1, r2, w, i, TapWindow{}, Taps{}, S.q.vn8};
            if (main && CACHED) sf.tw.row_start(r0, r1, r2, w);
            run_row(sf);
        }
        if (main && CACHED && J.recon) {
            wave_sync();
            for (int c = lane; c < w; c += 64) recon[row_at + c] = r0[c];
        }
    }
    if constexpr (N > 0) bias_io = lw.bias;
}

template <int N, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) k_serial_model(const SerialJob *__restrict__ jobs, int dyn_bytes) {
    __shared__ ModelLds S;
    extern __shared__ __align__(16) uint8_t rows_raw[];
    uint8_t *rows = rows_raw + 4;                                        // margins for the lane-parallel front (kRowPad)
    const SerialJob &J = jobs[blockIdx.x];
    const auto st = gp(J.state);
    const auto st_ctx = gp(reinterpret_cast<int *>(J.state + 1));
    const int tid = int(threadIdx.x);
    if (st->status != kRunning) return;                                  // finished in an earlier launch (every wave of the workgroup sees the same)
    const int i0 = st->next_row, i1 = i0 + J.rows < J.h ? i0 + J.rows : J.h;
    int bias = i0 ? st->bias : lsq::kBiasInit;
    if (WAVES == 2) block_sync();                                        // every wave has read the record before wave 0 may write it again (a one-row image)
    for (int k = tid; k < kContexts; k += 64 * WAVES) S.ctx[k] = i0 ? st_ctx[k] : 0;
    if (tid < 64) fill_qlut(S.qlut);
    if (tid < 16) S.q.vn8[tid] = 0;
    if (tid < 4) rows_raw[tid] = 0;
    if (WAVES == 2) block_sync(); else wave_sync();
    const int rs = (J.w + kRowPad + 15) & ~15;
    if (3 * rs + 4 <= dyn_bytes) model_body<N, true, WAVES>(S, rows, J, rs, i0, i1, bias);
    else model_body<N, false, WAVES>(S, rows, J, rs, i0, i1, bias);
    if (WAVES == 2) block_sync(); else wave_sync();
    if (i1 < J.h) for (int k = tid; k < kContexts; k += 64 * WAVES) st_ctx[k] = S.ctx[k];
    if (tid == 0) { st->next_row = i1; st->bias = bias; st->status = i1 < J.h ? kRunning : kDone; }
}

// ---- decoder: the whole NBLIC loop (NBLIC.c:749-908 with decode = 1) ----------------------------
// The stream is staged through LDS 512 bytes at a time so that a renormalisation byte costs an LDS
// read, not a trip to HBM in the middle of the chain.  `len` is what is PRESENT of the stream (the device
// copy is padded so that whole 512-byte blocks can be fetched); consuming a byte at or beyond it raises
// `dry` and returns zeros without moving on, and the caller winds the image up at once.
struct StreamWindow {
    const uint8_t *base; size_t len, pos;      // pos = next byte to consume
    uint32_t *sbuf;
    bool dry;
    __device__ void fill_half(size_t from) {   // bytes [from, from + 512) -> sbuf half (from / 512) & 1; from is a multiple of 512
        const auto src = gp(reinterpret_cast<const uint32_t *>(base));       // device copies are 16-byte aligned and padded by 2 KB
        const size_t word = from / 4 + threadIdx.x * 2;
        sbuf[((from >> 9) & 1) * 128 + threadIdx.x * 2] = src[word];
        sbuf[((from >> 9) & 1) * 128 + threadIdx.x * 2 + 1] = src[word + 1];
    }
    __device__ void start(const uint8_t *b, size_t n, size_t at, uint32_t *buf) {
        base = b; len = n; pos = at; sbuf = buf; dry = false;
        fill_half(at & ~size_t(511)); fill_half((at & ~size_t(511)) + 512);
        wave_sync();
    }
    __device__ __forceinline__ uint32_t next() {
        if (pos >= len) { dry = true; return 0u; }
        const uint32_t byte = (sbuf[(pos & 1023) >> 2] >> (8 * (pos & 3))) & 0xFFu;
        pos++;
        if ((pos & 511) == 0) {                 // a half has been consumed: refill it with the bytes 1024 ahead of its start
            wave_sync();
            fill_half(pos + 512);
            wave_sync();
        }
        return byte;
    }
};

// One symbol of the NBLIC decoder (NBLIC.c:640-679, :628-637, :552-573) with the PROBABILITIES on the lanes.  A symbol's
// bins are a unary prefix over the nodes 0, S, 2S, ... (S = 1 << k_max) and then k suffix bits down a binary tree below
// the node the prefix stopped at.  Which bins are coded depends on the stream; which nodes CAN come does not, and a
// symbol never visits a counter twice -- so lane t reads the counters of prefix node t and mixes their probability
// while lane 0 ... all at once, the serial part of a bin shrinks to the coder's interval arithmetic (one v_readlane,
// a 32 x 12-bit product, a compare, the renormalisation), and the counters of the bins that were coded are updated by
// their lanes together.  The suffix likewise: lane l holds node l of the tree in heap order (2^k - 1 nodes, k <= 5).
// Prefixes longer than the lanes (64 nodes, or the end of the tree: the reference moves on to the next level's tree)
// continue bin by bin as the reference does.
template <class Lds>
__device__ __forceinline__ int decode_symbol(Lds &S, const Level &L, const int k_step, const uint64_t ktab, uint32_t &lo, uint32_t &hi,
                                             uint32_t &window, StreamWindow &sw, bool &damaged, const int lane, const int sfx_d, const int sfx_prefix) {
    const int k_max = int(ktab >> 60), qw = L.qw;
    int qu = L.qu, qv = L.qv;
    int k = int(ktab >> (4 * qu)) & 15;
    if ((int(ktab >> (4 * qv)) & 15) != k) qv = qu;
    auto prob_of = [&](uint32_t cu, uint32_t cv) {
        return mix_prob(prob_one(int(cu & 0xFFFFu), int(cu >> 16)), prob_one(int(cv & 0xFFFFu), int(cv >> 16)), qw);
    };
    auto code = [&](int prob, bool stop) {                               // one bin off the stream (uniform)
        const uint32_t cut = lo + uint32_t((u64(hi - lo) * uint32_t(prob)) >> 12);
        const int bin = (sw.dry | stop) ? 0 : int(window <= cut);
        if (bin) hi = cut; else lo = cut + 1;
        while (((lo ^ hi) >> 24) == 0 && !sw.dry) { window = (window << 8) | sw.next(); lo <<= 8; hi = (hi << 8) | 0xFFu; }
        return bin;
    };
    auto settle = [&](uint32_t cu, uint32_t cv, int node, int bin) {     // the bin into its two counters (one, if the levels coincide)
        Counter a{int(cu & 0xFFFFu), int(cu >> 16)};
        counter_add(a, bin, kWeightOne - qw);
        if (qu == qv) counter_add(a, bin, qw);
        S.cnt[qu][node] = uint32_t(a.c0) | (uint32_t(a.c1) << 16);
        if (qu != qv) {
            Counter b{int(cv & 0xFFFFu), int(cv >> 16)};
            counter_add(b, bin, qw);
            S.cnt[qv][node] = uint32_t(b.c0) | (uint32_t(b.c1) << 16);
        }
    };
    // prefix
    const int reach = (kTreeNodes >> k_max) < 64 ? (kTreeNodes >> k_max) : 64;
    int node = lane < reach ? lane << k_max : 0;
    uint32_t cu = S.cnt[qu][node], cv = S.cnt[qv][node];
    int P = prob_of(cu, cv);
    int t = 0, bin;
    for (;;) {
        bin = code(__builtin_amdgcn_readlane(P, t), false);
        if (!bin || ++t == reach) break;
    }
    if (lane < t + 1 - bin) settle(cu, cv, node, int(lane < t));         // lanes 0 .. t-1 saw a one, lane t (if there) the zero
    node = t << k_max;
    if (bin) {                                                           // beyond the lanes
        int bins = reach;
        for (;;) {
            if (node >= kTreeNodes) { node >>= 1; k++; qu = qv = k * k_step; }
            cu = S.cnt[qu][node]; cv = S.cnt[qv][node];
            // a symbol of a valid stream has well under a hundred bins; a damaged one (all ones) could walk for ever:
            // from the 512th bin of a pixel on, and once the stream has run dry, every symbol ends at once
            if (++bins > 512) damaged = true;
            bin = code(prob_of(cu, cv), damaged);
            settle(cu, cv, node, bin);
            if (!bin) break;
            node += 1 << k_max;
        }
    }
    int z = (node >> k_max) << k;
    if (k > 0) {                                                         // suffix: node + 1 is the root; a one at depth d moves on by 2^(k-d), a zero by 1
        const int off = suffix_lane_offset(k, sfx_d, sfx_prefix);
        int n2 = node + 1 + (sfx_d < k ? off : 0);
        n2 = n2 < kTreeNodes ? n2 : kTreeNodes - 1;
        cu = S.cnt[qu][n2]; cv = S.cnt[qv][n2];
        P = prob_of(cu, cv);
        int at = 0;
        u64 seen = 0, ones = 0;
        for (int kk = k - 1; kk >= 0; kk--) {
            bin = code(__builtin_amdgcn_readlane(P, at), damaged);
            seen |= 1ull << at;
            if (bin) { ones |= 1ull << at; z += 1 << kk; }
            at = 2 * at + 1 + bin;
        }
        if ((seen >> lane) & 1ull) settle(cu, cv, n2, int((ones >> lane) & 1ull));
    }
    return z;
}

// rows [i0, i1) of the image; coder state in / out through `cs` (lo, hi, window); returns the row it stopped in front of
// and sets `stop` to kRunning (ran its rows), kStarved, kStarvedMidRow or kFailed
template <int N, bool CACHED, class Lds>
__device__ __forceinline__ int decode_body(Lds &S, uint8_t *rows, const SerialJob &J, const int rs, const int i0, const int i1,
                                           StreamWindow &sw, uint32_t (&cs)[3], int &bias_io, const bool final_, int &stop) {
    constexpr bool kLean = std::is_same<Lds, DecodeLdsLean>::value;
    const int w = J.w, lane = int(threadIdx.x), k_step = J.k_step;
    // the re-mappers' hit counts: LDS, or the image's state record
    auto hits = [&] {
        if constexpr (kLean) return gp(reinterpret_cast<int *>(J.state + 1) + kRecCount);
        else return &S.count[0][0];
    }();
    const NearParams np = near_params(J.near);
    const uint64_t ktab = level_shift_table(k_step);
    const auto out = gp(J.recon);
    LsqWalk<N> lw;
    if constexpr (N > 0) lw.init(J.stats, w, lane, 0, bias_io);
    LaneFront lf;
    lf.init(lane, &S.q);
    const int sfx_d = 31 - __clz(lane + 1), sfx_prefix = lane + 1 - (1 << sfx_d);      // decode_symbol: the lane's place in the suffix tree
    uint32_t lo = cs[0], hi = cs[1], window = cs[2];
    if (CACHED && i0 > 0) {                                              // resuming: the two rows above come back from the decoded plane
        for (int r = i0 > 1 ? i0 - 2 : i0 - 1; r < i0; r++) {
            uint8_t *dst = rows + (r % 3) * rs;
            for (int c = lane; c < w; c += 64) dst[c] = out[size_t(r) * size_t(w) + c];
        }
        wave_sync();
    }
    bool damaged = false;
    int i = i0;
    stop = kRunning;
    for (; i < i1; i++) {
        // a stream that is still being fed: stop in front of a row rather than inside it (serial_engine.h starve_margin)
        if (!final_ && sw.len - sw.pos < starve_margin(w)) { stop = kStarved; break; }
        uint8_t *r0 = rows + (i % 3) * rs, *r1 = rows + ((i + 2) % 3) * rs, *r2 = rows + ((i + 1) % 3) * rs;
        auto pix = [&](int r, int c) {
            if (CACHED) return int((r == i ? r0 : (r == i - 1 ? r1 : r2))[c]);
            return int(out[size_t(r) * size_t(w) + size_t(c)]);
        };
        if constexpr (N > 0) lw.row_begin(S.q);
        int err = 0;
        const size_t row_at = size_t(i) * size_t(w);
        auto run_row = [&](auto &front) {
        for (int j = 0; j < w; j++) {
            front.begin(j);
            int px0, p1_used = 0;
            if constexpr (N > 0) {
                front.regressors();
                wave_sync();
                lw.predict(S.q, j);
                if (lw.ok1) { px0 = (lw.p1 + (1 << (lsq::kFb1 - 1))) >> lsq::kFb1; p1_used = lw.p1; }
                else { px0 = front.predict(); p1_used = px0 << lsq::kFb1; }
            } else {
                px0 = front.predict();
            }
            const int delta = front.activity(err);
            const Level L = level_from(uint16_t(__builtin_amdgcn_readfirstlane(S.qlut[delta < 200 ? delta : 200])));
            const int adr = front.context(L.qu, px0);
            const int v = __builtin_amdgcn_readfirstlane(S.ctx[adr]);
            const int sign = bias_sign(v), px = bias_apply(v, px0);
            const int mk = px * 2 + sign;
            const int z = decode_symbol(S, L, k_step, ktab, lo, hi, window, sw, damaged, lane, sfx_d, sfx_prefix);
            if (sw.dry | damaged) return;                                // the image cannot be finished from here: no pixel is written for this symbol
            const int y = z < kMapSyms ? __builtin_amdgcn_readfirstlane(int(S.sym_at[mk][z])) : z;
            if (y < kMapSyms) {                                          // NBLIC.c:497-523 (z is y's rank)
                const int at = mk * kMapSyms + z;
                const int c = hits[at] + 1;
                const int c_up = z > 0 ? hits[at - 1] : 0x7FFFFFFF;
                if (c_up < c) {
                    const int other = S.sym_at[mk][z - 1];
                    hits[at] = c_up; hits[at - 1] = c;
                    S.sym_at[mk][z] = uint8_t(other); S.sym_at[mk][z - 1] = uint8_t(y);
                    if constexpr (!kLean) { S.rank_of[mk][y] = uint8_t(z - 1); S.rank_of[mk][other] = uint8_t(z); }
                } else {
                    hits[at] = c;
                }
            }
            const int xr = __builtin_amdgcn_readfirstlane(symbol_to_pixel(y, px, sign, np));
            err = clip_err(xr, px0);
            S.ctx[adr] = bias_update(v, err);
            if (CACHED) r0[j] = uint8_t(xr); else { out[row_at + j] = uint8_t(xr); __threadfence_block(); }
            front.advance(j, xr);
            if constexpr (N > 0) {
                lw.update(S.q, j, xr, p1_used);
                wave_sync();
            }
        }
        };
        if (CACHED && i >= 2) {
            lf.row_start(rows, r1, r2, w, lane);
            run_row(lf);
        } else {
            ScalarFront<CACHED, decltype(pix)> sf{pix, r1, r2, w, i, TapWindow{}, Taps{}, S.q.vn8};
            if (CACHED) sf.tw.row_start(r0, r1, r2, w);
            run_row(sf);
        }
        if (sw.dry | damaged) { stop = (damaged || final_) ? kFailed : kStarvedMidRow; break; }
        if (CACHED) {
            wave_sync();
            for (int c = lane; c < w; c += 64) out[row_at + c] = r0[c];
        }
    }
    cs[0] = lo; cs[1] = hi; cs[2] = window;
    if constexpr (N > 0) bias_io = lw.bias;
    return i;
}

template <int N, bool LEAN>
__global__ void __launch_bounds__(64) k_serial_decode(const SerialJob *__restrict__ jobs, int dyn_bytes) {
    using Lds = typename std::conditional<LEAN, DecodeLdsLean, DecodeLds>::type;
    __shared__ Lds S;
    extern __shared__ __align__(16) uint8_t rows_raw[];
    uint8_t *rows = rows_raw + 4;                                        // serial_engine.h row_stride: margins for the lane-parallel front
    const SerialJob &J = jobs[blockIdx.x];
    const auto st = gp(J.state);
    const auto st_tab = gp(reinterpret_cast<uint32_t *>(J.state + 1));
    constexpr int kTabWords = int((kDecodeStateBytes - sizeof(SerialState)) / 4);
    uint32_t *lds_tab = reinterpret_cast<uint32_t *>(&S);
    uint32_t *lds_sym = reinterpret_cast<uint32_t *>(&S.sym_at[0][0]);
    const int lane = int(threadIdx.x);
    // the tables between the image's state record and LDS (the lean image: contexts and counters, then the rank -> symbol table)
    auto tables = [&](auto move) {
        if constexpr (LEAN) {
            for (int k = lane; k < kRecCount; k += 64) move(lds_tab[k], st_tab[k]);
            for (int k = lane; k < 512 * kMapSyms / 4; k += 64) move(lds_sym[k], st_tab[kRecSym + k]);
        } else {
            for (int k = lane; k < kTabWords; k += 64) move(lds_tab[k], st_tab[k]);
        }
    };
    if (st->status != kRunning) return;                                  // finished, failed, or waiting for the host to feed the stream
    const int i0 = st->next_row, i1 = i0 + J.rows < J.h ? i0 + J.rows : J.h;
    const bool final_ = st->final_ != 0;
    const size_t avail = size_t(st->avail);
    if (!final_ && avail < size_t(kHeaderBytes) + 4 + starve_margin(J.w) && i0 == 0) {       // not even the start of the stream is there yet
        if (lane == 0) st->status = kStarved;
        return;
    }
    if (i0 == 0) {
        for (int k = lane; k < kContexts; k += 64) S.ctx[k] = 0;
        for (int k = lane; k < kLevels * kTreeNodes; k += 64) (&S.cnt[0][0])[k] = uint32_t(kWeightOne) | (uint32_t(kWeightOne) << 16);
        for (int k = lane; k < 512 * kMapSyms; k += 64) {
            const int s = k % kMapSyms;
            (&S.sym_at[0][0])[k] = uint8_t(s);
            if constexpr (LEAN) st_tab[kRecCount + k] = uint32_t(2 * (kMapSyms - 1 - s));
            else { (&S.count[0][0])[k] = 2 * (kMapSyms - 1 - s); (&S.rank_of[0][0])[k] = uint8_t(s); }
        }
        if constexpr (LEAN) __threadfence_block();
    } else {
        tables([](uint32_t &lds, NB_GLOBAL uint32_t &rec) { lds = rec; });
    }
    fill_qlut(S.qlut);
    if (lane < 16) S.q.vn8[lane] = 0;
    wave_sync();
    StreamWindow sw;
    uint32_t cs[3] = {0u, 0xFFFFFFFFu, 0u};                              // NBLIC.c:536-549
    int bias = lsq::kBiasInit;
    if (i0 == 0) {
        sw.start(J.stream, avail, kHeaderBytes, S.sbuf);
        for (int k = 0; k < 4; k++) cs[2] = (cs[2] << 8) | sw.next();
    } else {
        sw.start(J.stream, avail, size_t(st->pos), S.sbuf);
        cs[0] = st->lo; cs[1] = st->hi; cs[2] = st->window; bias = st->bias;
    }
    const int rs = (J.w + kRowPad + 15) & ~15;
    if (lane < 4) rows_raw[lane] = 0;
    int stop = kRunning, at;
    if (sw.dry) { stop = kFailed; at = i0; }                             // a final stream shorter than its own start
    else if (3 * rs + 4 <= dyn_bytes) at = decode_body<N, true>(S, rows, J, rs, i0, i1, sw, cs, bias, final_, stop);
    else at = decode_body<N, false>(S, rows, J, rs, i0, i1, sw, cs, bias, final_, stop);
    wave_sync();
    if (stop == kFailed || stop == kStarvedMidRow) { if (lane == 0) st->status = stop; return; }
    if (at < J.h) tables([](uint32_t &lds, NB_GLOBAL uint32_t &rec) { rec = lds; });
    if (lane == 0) {
        st->next_row = at; st->pos = sw.pos; st->lo = cs[0]; st->hi = cs[1]; st->window = cs[2]; st->bias = bias;
        st->status = at >= J.h ? kDone : stop;                           // kRunning (more rows to go) or kStarved (feed me)
    }
}

// ---- QNBLIC decoder (QNBLIC.c:493-555): one image per wave ----------------------------------------------
// Context table, the twelve cumulative tables (16-bit: the histograms are normalised to 2^15) and a coarse
// symbol index -- built by the kernel itself from the cumulative table -- live in LDS: a symbol is found from the
// rANS state's low 15 bits by one coarse lookup (the symbol of slot low & ~127) plus a short walk along the
// cumulative table; no slot -> symbol table exists anywhere.  Rows >= 2 (rows that fit in LDS): the pixel's model
// on the lanes (lane_table.h; QNBLIC's window neighbourhood equals clamped sampling there except a / e at the row
// start); rows 0 and 1 use the closed form of model.h sample_taps_q on every lane alike.
// Resumable like the others; a pixel consumes at most one 16-bit word, so a row never needs more than 2 w bytes
// and a stream that is still being fed is only ever left in front of a row (kStarved).
struct QDecodeLds {
    int ctx[3072];
    uint32_t span[12 * 256 + 2];             // per (level, symbol): first slot | first slot of the next symbol << 16 (the histograms sum to 2^15)
    uint8_t coarse[12][256];
    uint32_t sbuf[256];
};

// What a pixel costs here is the chain  rANS state -> symbol -> pixel -> error -> (next pixel's level) -> ...  with a
// dependent LDS round trip (~64 cycles) at every table lookup, and the ~200 instructions of the predictor next to it.
// So: the level (cheap, needs only the taps and the last error) is computed FIRST and the symbol search started from it
// -- one coarse lookup, then ONE 8-byte read that brings the boundaries of three consecutive symbols (the frequency is
// their difference: no second table) -- while the seven-direction predictor runs; the context bias is looked up when the
// prediction is there, and only the pixel itself needs both chains.  The row loops are straight-line code: taps from the
// register window for rows >= 2 of an image whose rows fit in LDS (the usual case), a generic accessor otherwise.
struct QRans {
    uint32_t x;
    __device__ __forceinline__ int symbol(const QDecodeLds &S, int qd, StreamWindow &sw) {
        const uint32_t low = x & 32767u;
        int y = S.coarse[qd][low >> 7];
        uint32_t s0, s1;
        for (;;) {                                                       // symbols with no slots are stepped over; cumulative starts are <= 2^15
            const uint32_t p0 = S.span[qd * 256 + y], p1 = S.span[qd * 256 + y + 1];
            const uint32_t e0 = p0 >> 16, e1 = p1 >> 16;
            if (low < e0 || y >= 255) { s0 = p0 & 0xFFFFu; s1 = e0; break; }
            if (low < e1 || y >= 254) { y += 1; s0 = e0; s1 = e1; break; }
            y += 2;
        }
        x = (x >> 15) * (s1 - s0) + low - s0;
        if (x < 65536u) { const uint32_t lo = sw.next(); x = (x << 16) | lo | (sw.next() << 8); }
        return y;
    }
};

template <bool CACHED>
__device__ __forceinline__ int qdecode_rows(QDecodeLds &S, uint8_t *rows, const SerialJob &J, const int rs, const int i0, const int i1,
                                            StreamWindow &sw, QRans &rans, const bool final_, const size_t row_need, int &stop) {
    const int lane = int(threadIdx.x), w = J.w;
    const auto out = gp(J.recon);
    const QLaneConst lc = kQLanes.l[lane];
    if (CACHED && i0 > 0) {
        for (int r = i0 > 1 ? i0 - 2 : i0 - 1; r < i0; r++) {
            uint8_t *dst = rows + (r % 3) * rs;
            for (int c = lane; c < w; c += 64) dst[c] = out[size_t(r) * size_t(w) + c];
        }
        wave_sync();
    }
    int i = i0;
    stop = kRunning;
    for (; i < i1; i++) {
        if (!final_ && sw.len - sw.pos < row_need) { stop = kStarved; break; }
        uint8_t *r0 = rows + (i % 3) * rs, *r1 = rows + ((i + 2) % 3) * rs, *r2 = rows + ((i + 1) % 3) * rs;
        const size_t row_at = size_t(i) * size_t(w);
        int err = 0;
        auto pixel = [&](const Taps &n, int j) {                         // returns the decoded pixel
            const int qd = level_q(n, err);
            const int y = rans.symbol(S, qd, sw);                        // independent of the prediction: its lookups overlap the predictor
            const int px0 = predict_q(n);
            const int adr = context_address_q(n, qd, px0);
            const int v = S.ctx[adr];
            const int sign = (v >> 10) & 1;
            const int px = iclip(px0 + (v >> 11) + sign, 0, kMaxVal);
            const int px_out = symbol_to_pixel(y, px, sign, 0);
            err = px_out - px0;
            S.ctx[adr] = (v * 128 - v + err * 2048 + 63) >> 7;
            (void)j;
            return px_out;
        };
        if (CACHED && i >= 2) {
            // the rows above, with their edge pixels repeated two columns out (QNBLIC.c:48-79 clamps the columns)
            if (lane < 8) {
                uint8_t *r = lane < 4 ? r1 : r2;
                const int k = lane & 3;
                r[k < 2 ? -1 - k : w + k - 2] = r[k < 2 ? 0 : w - 1];
            }
            wave_sync();
            const uint8_t *zero = rows - 4;
            const uint8_t *pX = lc.sel[0] == 0 ? zero : (lc.sel[0] == 1 ? r1 : r2) + lc.dx[0];
            const uint8_t *pY = lc.sel[1] == 0 ? zero : (lc.sel[1] == 1 ? r1 : r2) + lc.dx[1];
            const uint8_t *pZ = lc.sel[2] == 0 ? zero : (lc.sel[2] == 1 ? r1 : r2) + lc.dx[2];
            const int stX = lc.sel[0] != 0, stY = lc.sel[1] != 0, stZ = lc.sel[2] != 0;
            int a = r1[0], e = a;                                         // the window hands on row i-1's first pixel
            // What of a pixel does not wait for the pixel on its left (whose value enters as `a` only) is computed one pixel
            // AHEAD, in the shadow of the previous pixel's context and symbol lookups: the a-free part of every term, the
            // rows-above part of the seven extrapolations and of the linear predictor.
            struct Early { int s_yz, x2, ang_bcd, lin_part; };
            auto early = [&](int X, int Y, int Z, int e_) {
                Early E;
                E.s_yz = __mul24(e_, int(lc.ce)) + (Y + Z);
                E.x2 = X << 1;
                const int b = __builtin_amdgcn_readlane(X, 2), c = __builtin_amdgcn_readlane(X, 1), d = __builtin_amdgcn_readlane(X, 3);
                const int f = __builtin_amdgcn_readlane(Y, 6);
                E.ang_bcd = __mul24(b, int(lc.cb)) + __mul24(c, int(lc.cc)) + __mul24(d, int(lc.cd));
                E.lin_part = 9 * b + 2 * (d - c) - e_ - f;
                return E;
            };
            Early E = early(int(*pX), int(*pY), int(*pZ), e);
            pX += stX; pY += stY; pZ += stZ;
            int Xn = *pX, Yn = *pY, Zn = *pZ;                                         // pixel 1's operands
            for (int j = 0; j < w; j++) {
                // the lane's term
                const int X2 = E.x2 + __mul24(a, int(lc.a2));
                const int V = X2 - E.s_yz;
                int Q = V < 0 ? -V : V;
                Q += dpp_i32<kQuadX1>(Q); Q += dpp_i32<kQuadX2>(Q);                   // the quad's four terms
                // level -> first lookup of the symbol search (its round trips to LDS run under the prediction)
                const int act = ((__builtin_amdgcn_readlane(Q, 28) + __builtin_amdgcn_readlane(Q, 32)) >> 1) + 2 * iabs(err);
                const int qd = __builtin_popcountll(__ballot(int(lc.thr_level) <= act));
                const uint32_t low = rans.x & 32767u;
                const int yc = __builtin_amdgcn_readfirstlane(S.coarse[qd][low >> 7]);
                // prediction, first half
                int key = (Q << 3) | lc.key_or, sum = Q & lc.sum_and;
                key = min_u(key, dpp_i32<kRor4>(key)); key = min_u(key, dpp_i32<kRor8>(key));
                sum += dpp_i32<kRor4>(sum); sum += dpp_i32<kRor8>(sum);
                const int k_min = min_u(__builtin_amdgcn_readlane(key, 0), __builtin_amdgcn_readlane(key, 16));
                const int total = __builtin_amdgcn_readlane(sum, 0) + __builtin_amdgcn_readlane(sum, 16);
                const int best = k_min >> 3, dir = k_min & 7;
                const int spread = (total - 7 * best) >> 3;
                // second lookup: the slot boundaries of symbols yc, yc + 1 (and the start of yc + 2)
                const uint32_t p0 = S.span[qd * 256 + yc], p1 = S.span[qd * 256 + yc + 1];
                // prediction, second half
                const int wt = __builtin_popcountll(__ballot(int(lc.thr_weight) <= spread));
                const int ang = __builtin_amdgcn_readlane(E.ang_bcd + __mul24(a, int(lc.ca)), dir);
                const int lin = iclip(E.lin_part + 9 * a, 0, 16 * kMaxVal);
                const int px0 = (8 * wt * ang + (8 - wt) * lin + 64) >> 7;
                const int adr = (qd << 8) | int((__ballot((px0 << int(lc.sh)) > V) >> 36) & 0xFFull);
                const int v_raw = S.ctx[adr];
                // in the shadow of those lookups: the next pixel's early part (its e is this pixel's a), the operands after it
                const Early En = early(Xn, Yn, Zn, a);
                pX += stX; pY += stY; pZ += stZ;
                const int Xnn = *pX, Ynn = *pY, Znn = *pZ;
                // the symbol: nearly always one of the two just read
                const uint32_t e0 = p0 >> 16, e1 = p1 >> 16;
                int y;
                uint32_t s0, s1;
                if (low < e0 || yc >= 255) { y = yc; s0 = p0 & 0xFFFFu; s1 = e0; }
                else if (low < e1 || yc >= 254) { y = yc + 1; s0 = e0; s1 = e1; }
                else {
                    y = yc + 2;
                    for (;;) {                                                       // symbols with no slots are stepped over
                        const uint32_t q0 = S.span[qd * 256 + y];
                        if (low < (q0 >> 16) || y >= 255) { s0 = q0 & 0xFFFFu; s1 = q0 >> 16; break; }
                        y++;
                    }
                }
                rans.x = mul_u24(rans.x >> 15, s1 - s0) + low - s0;               // < 2^17 times <= 2^15
                if (rans.x < 65536u) { const uint32_t lo = sw.next(); rans.x = (rans.x << 16) | lo | (sw.next() << 8); }
                // the pixel
                const int v = __builtin_amdgcn_readfirstlane(v_raw);
                const int sign = (v >> 10) & 1;
                const int px = iclip(px0 + (v >> 11) + sign, 0, kMaxVal);
                const int px_out = __builtin_amdgcn_readfirstlane(symbol_to_pixel(y, px, sign, 0));
                err = px_out - px0;
                S.ctx[adr] = (__mul24(v, 127) + err * 2048 + 63) >> 7;
                r0[j] = uint8_t(px_out);
                e = a; a = px_out;
                E = En; Xn = Xnn; Yn = Ynn; Zn = Znn;
                if (sw.dry) break;
            }
        } else {
            auto pix = [&](int r, int c) {
                if (CACHED) return int((r == i ? r0 : (r == i - 1 ? r1 : r2))[c]);
                return int(out[size_t(r) * size_t(w) + size_t(c)]);
            };
            for (int j = 0; j < w; j++) {
                const int px_out = pixel(sample_taps_q(pix, w, i, j), j);
                if (CACHED) r0[j] = uint8_t(px_out); else { out[row_at + j] = uint8_t(px_out); __threadfence_block(); }
                if (sw.dry) break;
            }
        }
        if (sw.dry) { stop = kFailed; break; }                           // only a final stream can run dry inside a row (a pixel takes one word at most)
        if (CACHED) {
            wave_sync();
            for (int c = lane; c < w; c += 64) out[row_at + c] = r0[c];
        }
    }
    return i;
}

__global__ void __launch_bounds__(64) k_serial_qdecode(const SerialJob *__restrict__ jobs, int dyn_bytes) {
    __shared__ QDecodeLds S;
    extern __shared__ __align__(16) uint8_t rows_raw[];
    const SerialJob &J = jobs[blockIdx.x];
    const auto st = gp(J.state);
    const auto st_ctx = gp(reinterpret_cast<int *>(J.state + 1));
    const int lane = int(threadIdx.x), w = J.w, h = J.h;
    if (st->status != kRunning) return;
    const int i0 = st->next_row, i1 = i0 + J.rows < h ? i0 + J.rows : h;
    const bool final_ = st->final_ != 0;
    const size_t avail = size_t(st->avail) & ~size_t(1), row_need = size_t(2) * size_t(w) + 8;
    if (!final_ && (avail < size_t(st->pos) || avail - size_t(st->pos) < row_need)) { if (lane == 0) st->status = kStarved; return; }
    for (int k = lane; k < 3072; k += 64) {
        const uint32_t s0 = gp(J.q_start)[k], f = gp(J.q_freq)[k];
        S.ctx[k] = i0 ? st_ctx[k] : 0;
        S.span[k] = (s0 & 0xFFFFu) | (((s0 + f) & 0xFFFFu) << 16);
    }
    if (lane < 2) S.span[3072 + lane] = 0;
    wave_sync();
    for (int k = lane; k < 3072; k += 64) {                              // the symbol that holds slot 128 (k & 255) of level k >> 8: the last one starting at or below it
        const uint32_t slot = uint32_t(k & 255) * 128u, *sp = S.span + (k & ~255);
        int lo = 0, hi = 255;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if ((sp[mid] & 0xFFFFu) <= slot) lo = mid; else hi = mid - 1;
        }
        (&S.coarse[0][0])[k] = uint8_t(lo);
    }
    wave_sync();
    // a row in LDS: two columns of margin on the left (after the four bytes in front of row 0, which hold zero for the
    // lanes without an operand) and at least four on the right -- the lane-parallel model reads the clamped columns there
    const int rs = (w + kRowPad + 15) & ~15;
    uint8_t *rows = rows_raw + 4;
    if (lane < 4) rows_raw[lane] = 0;
    StreamWindow sw;
    sw.start(J.stream, avail, size_t(st->pos), S.sbuf);                   // a fresh image: the host has set pos to the first word after the tables
    auto next_word = [&]() { const uint32_t lo = sw.next(); return lo | (sw.next() << 8); };
    QRans rans;
    if (i0 == 0) { rans.x = next_word() << 16; rans.x |= next_word(); }
    else rans.x = st->lo;
    int stop = kRunning, at;
    if (3 * rs + 4 <= dyn_bytes) at = qdecode_rows<true>(S, rows, J, rs, i0, i1, sw, rans, final_, row_need, stop);
    else at = qdecode_rows<false>(S, rows, J, rs, i0, i1, sw, rans, final_, row_need, stop);
    wave_sync();
    if (stop == kFailed || sw.dry) { if (lane == 0) st->status = kFailed; return; }
    if (at < h) for (int k = lane; k < 3072; k += 64) st_ctx[k] = S.ctx[k];
    if (lane == 0) { st->next_row = at; st->pos = sw.pos; st->lo = rans.x; st->status = at >= h ? kDone : stop; }
}

// ---- self-test: the double-carried divisions against 64-bit integers ------------------------------
__global__ void k_selftest_div(const i64 *a, const i64 *b, const i64 *d, int n, uint32_t *bad) {
    const int t = int(blockIdx.x * blockDim.x + threadIdx.x);
    if (t >= n) return;
    lsq::Guard g;
    const double dd = double(d[t]);
    const i64 want = mulw(a[t], b[t]) / d[t];
    const double got = lsq::muldiv_trunc(double(a[t]), double(b[t]), dd, lsq::recip_of(dd), g);
    const double raw = lsq::recip_raw(dd);                                              // the device reciprocal: seed + two Newton steps
    const double rel = fabs(fma(raw, dd, -1.0));                                         // |raw * d - 1| exactly rounded once
    if (!(rel < 1.5e-15)) atomicAdd(bad, 1u);                                            // 2^-49 = 1.78e-15
    if (i64(got) != want) atomicAdd(bad, 1u);
    const i64 v = a[t] >> 4;
    if (i64(lsq::decay<5>(double(v))) != (v * 4 + 2) / 5 || i64(lsq::decay<3>(double(v))) != (v * 2 + 1) / 3) atomicAdd(bad, 1u);
}

// the half-wave exchange the split solve relies on: lane L must see lane (L & 31) + 32 * HALF's value
__global__ void k_selftest_swap(uint32_t *bad) {
    const int lane = int(threadIdx.x);
    const double v = 1000.0 * double(blockIdx.x + 1) + double(lane) + 0.25;
    const double lo = bcast_half<0>(v), hi = bcast_half<1>(v);
    const double base = 1000.0 * double(blockIdx.x + 1) + 0.25;
    if (lo != base + double(lane & 31) || hi != base + double((lane & 31) + 32)) atomicAdd(bad, 1u);
    const double r0 = bcast_row<0>(v), r1 = bcast_row<1>(v), r2 = bcast_row<2>(v), r3 = bcast_row<3>(v);
    const double l15 = base + double(lane & 15);
    if (r0 != l15 || r1 != l15 + 16.0 || r2 != l15 + 32.0 || r3 != l15 + 48.0) atomicAdd(bad, 1u);
    // the eight-lane reductions: every lane of a half-row ends with the half-row's maximum / sum
    const double hm = half_row_max(double(lane * 7 % 13)), hs = half_row_sum(double(lane));
    double want_m = 0.0, want_s = 0.0;
    for (int k = 0; k < 8; k++) { const int l = (lane & ~7) + k; want_m = want_m > double(l * 7 % 13) ? want_m : double(l * 7 % 13); want_s += double(l); }
    if (hm != want_m || hs != want_s) atomicAdd(bad, 1u);
}

int serial_selftest(hipStream_t s) {
    constexpr int n = 1 << 16;
    i64 *h = static_cast<i64 *>(malloc(3 * n * sizeof(i64))), *dv = nullptr;
    uint32_t *d_bad = nullptr, bad = 1;
    if (!h) return -1;
    u64 x = 88172645463325252ull;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    for (int k = 0; k < n; k++) {
        const int sa = 1 + int(rnd() % 44), sb = 1 + int(rnd() % 18), sd = 1 + int(rnd() % 30);       // |a b| < 2^62, |a b / d| < 2^48 is NOT guaranteed:
        i64 a = i64(rnd() >> (64 - sa)), b = i64(rnd() >> (64 - sb)), d = i64(rnd() >> (64 - sd)) + 1; // keep the quotient inside the estimate's range
        while (double(a) * double(b) / double(d) >= 6.0e13) d *= 2;
        if (rnd() & 1) a = -a;
        if (rnd() & 1) b = -b;
        if (rnd() & 1) d = -d;
        if (k < 64) { a = (k & 1) ? -(k / 2) : k / 2; b = 3 - (k % 7); d = (k % 5) - 2; if (d == 0) d = 1; }   // exact multiples, zeros, sign mixes
        h[k] = a; h[n + k] = b; h[2 * n + k] = d;
    }
    if (hipMalloc((void **)&dv, 3 * n * sizeof(i64)) != hipSuccess || hipMalloc((void **)&d_bad, 4) != hipSuccess) { free(h); return -1; }
    hipMemcpyAsync(dv, h, 3 * n * sizeof(i64), hipMemcpyHostToDevice, s);
    hipMemsetAsync(d_bad, 0, 4, s);
    hipLaunchKernelGGL(k_selftest_div, dim3(n / 256), dim3(256), 0, s, dv, dv + n, dv + 2 * n, n, d_bad);
    hipLaunchKernelGGL(k_selftest_swap, dim3(4), dim3(64), 0, s, d_bad);
    hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, s);
    hipStreamSynchronize(s);
    hipFree(dv); hipFree(d_bad); free(h);
    return int(bad);
}

// ---- launchers ----------------------------------------------------------------------------------
constexpr int kLdsBudget = 160 * 1024;
constexpr int kLeanImages = 256;                  // decode launches of more images than this keep the hit counts in memory (four waves per CU)
constexpr int kTwoWaveImages = 64;               // effort-3 launches of at most this many images give every image a second wave
constexpr int lds_room(size_t static_lds) { return int(kLdsBudget - static_lds - 256) & ~15; }

bool serial_model_rows_fit(int w) { return 3 * ((w + kRowPad + 15) & ~15) + 4 <= lds_room(sizeof(ModelLds)); }

template <class K>
static bool launch_rows(K kernel, size_t static_lds, const SerialJob *d_jobs, const SerialJob *h_jobs, int n, hipStream_t s, int threads = 64,
                        int row_pad = 0) {
    int max_w = 1;
    for (int k = 0; k < n; k++) max_w = h_jobs[k].w > max_w ? h_jobs[k].w : max_w;
    const int room = lds_room(static_lds);
    int dyn = 3 * ((max_w + row_pad + 15) & ~15) + (row_pad ? 4 : 0);   // the kernels' row ring: rows_raw + 4, stride (w + kRowPad + 15) & ~15
    if (dyn > room) dyn = room;                  // wider images fall back to taps from memory (the kernel compares per job)
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, dyn) != hipSuccess) return false;
    hipLaunchKernelGGL(kernel, dim3(unsigned(n)), dim3(unsigned(threads)), size_t(dyn), s, d_jobs, dyn);
    return hipGetLastError() == hipSuccess;
}

bool serial_model_launch(const SerialJob *d_jobs, const SerialJob *h_jobs, int n, hipStream_t s) {
    if (n <= 0) return true;
    switch (h_jobs[0].effort) {
        case 1: return launch_rows(k_serial_model<0, 1>, sizeof(ModelLds), d_jobs, h_jobs, n, s, 64, kRowPad);
        case 2: return launch_rows(k_serial_model<6, 1>, sizeof(ModelLds), d_jobs, h_jobs, n, s, 64, kRowPad);
        default:
            // Effort 3, few images: two waves per image, a system each -- 4.39 instead of 4.5 us per pixel per image (the
            // two barriers per pixel eat two thirds of what the shorter elimination saves).  Many images: the second wave
            // would take a SIMD slot from another image for a 3 % shorter chain (2048 images: 156 against 261 Mpx/s), so
            // a batch that can fill the GPU with single waves keeps them.
            if (n <= kTwoWaveImages) return launch_rows(k_serial_model<10, 2>, sizeof(ModelLds), d_jobs, h_jobs, n, s, 128, kRowPad);
            return launch_rows(k_serial_model<10, 1>, sizeof(ModelLds), d_jobs, h_jobs, n, s, 64, kRowPad);
    }
}

bool serial_decode_launch(const SerialJob *d_jobs, const SerialJob *h_jobs, int n, hipStream_t s, bool whole_streams) {
    if (n <= 0) return true;
    // More images than CUs: the lean LDS image (four waves per CU instead of one; a pixel pays two reads of its hit counts
    // in L2 instead of LDS).  Only for streams that are there in full: a launch that runs dry inside a row is REDONE from
    // the state record, and the lean image has already changed the counts in it.
    if (whole_streams && n > kLeanImages) {
        switch (h_jobs[0].effort) {
            case 1: return launch_rows(k_serial_decode<0, true>, sizeof(DecodeLdsLean), d_jobs, h_jobs, n, s, 64, kRowPad);
            case 2: return launch_rows(k_serial_decode<6, true>, sizeof(DecodeLdsLean), d_jobs, h_jobs, n, s, 64, kRowPad);
            default: return launch_rows(k_serial_decode<10, true>, sizeof(DecodeLdsLean), d_jobs, h_jobs, n, s, 64, kRowPad);
        }
    }
    switch (h_jobs[0].effort) {
        case 1: return launch_rows(k_serial_decode<0, false>, sizeof(DecodeLds), d_jobs, h_jobs, n, s, 64, kRowPad);
        case 2: return launch_rows(k_serial_decode<6, false>, sizeof(DecodeLds), d_jobs, h_jobs, n, s, 64, kRowPad);
        default: return launch_rows(k_serial_decode<10, false>, sizeof(DecodeLds), d_jobs, h_jobs, n, s, 64, kRowPad);
    }
}

bool serial_qdecode_launch(const SerialJob *d_jobs, const SerialJob *h_jobs, int n, hipStream_t s) {
    if (n <= 0) return true;
    return launch_rows(k_serial_qdecode, sizeof(QDecodeLds), d_jobs, h_jobs, n, s, 64, kRowPad);
}

}  // namespace nblic

// serial_engine.hip -- raster-serial NBLIC engine: one wave per image, model state in LDS.
//
// Used for everything that cannot be replayed per key: NBLICdecompress at any
// setting, near-lossless encode, and the least-squares efforts 2/3.  Mirrors the reference's
// fused loop (NBLIC.c:749-908) on the device: context table (8 KB), counter trees (32 KB) and
// re-mappers (60 KB) sit in the CU's 160 KB LDS; the image, the stream and the least-squares
// row statistics stay in HBM.
//
// The whole wave walks the pixel loop with UNIFORM control flow: the coding of a pixel is a
// dependent chain, so all 64 lanes carry the same scalars (LDS reads broadcast; global stores are
// issued by lane 0).  What is not a chain is spread over the lanes: the 43/111-element
// least-squares statistics (update, per-row pre-pass, system assembly) are one or two elements per
// lane, and the n x n integer elimination (NBLIC.c:112-161) runs one matrix entry per lane -- within
// an elimination step every entry only reads the pivot row and the pivot column of the previous
// step, so the lane-parallel order gives the same integers as the reference's nested loops.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include "model.h"
#include "serial_engine.h"

namespace nblic {

typedef long long i64;
typedef unsigned long long u64;

constexpr int kLsqMaxN = 10, kLsqMaxM = 1 + kLsqMaxN + kLsqMaxN * kLsqMaxN;
constexpr int kFb1 = 12, kFb2 = 2, kFb3 = 10, kDecayS = 3, kDecayV = 5;
constexpr i64 kBiasInit = 8, kBiasMax = 4096, kBiasCoef = 21;
constexpr int kRowCache = 16384;                 // widest image whose three tap rows fit next to the model in LDS

__device__ __forceinline__ i64 mulw(i64 a, i64 b) { return i64(u64(a) * u64(b)); }    // wrapping, NBLIC.c:139
__device__ __forceinline__ i64 abs64(i64 v) { return v < 0 ? -v : v; }
__device__ __forceinline__ i64 clip64(i64 v, i64 lo, i64 hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ i64 decay(i64 v, int ab) { return (mulw(v, ab - 1) + ab / 2) / ab; }   // NBLIC.c:199,273-279

struct SerialArgs {
    uint8_t *img; uint8_t *stream; size_t stream_cap;
    int h, w, near, k_step, effort, decode;
    i64 *stats; long *len_out;
};

// ---- range coder on the device (NBLIC.c:527-586) -------------------------------------------
struct DevCoder {
    uint8_t *p, *end;
    uint32_t lo, hi, window;
    bool overflow;
    bool store;                 // only one lane of the wave writes the stream
};

template <bool DEC>
__device__ __forceinline__ int coder_bin(DevCoder &rc, int bin, uint32_t prob) {
    uint32_t cut = rc.lo + uint32_t((u64(rc.hi - rc.lo) * prob) >> 12);
    if (DEC) bin = rc.window <= cut;
    if (bin) rc.hi = cut; else rc.lo = cut + 1;
    while (((rc.lo ^ rc.hi) >> 24) == 0) {
        if (rc.p < rc.end) {
            if (DEC) rc.window = (rc.window << 8) | *rc.p; else if (rc.store) *rc.p = uint8_t(rc.hi >> 24);
        } else { rc.overflow = true; if (DEC) rc.window <<= 8; }
        rc.p++;
        rc.lo <<= 8; rc.hi = (rc.hi << 8) | 0xFFu;
    }
    return bin;
}

// P(bin==1) = floor(4096 * c1 / (c0 + c1)) (NBLIC.c:621-625) without the integer-divide expansion:
// both operands are < 2^14, so the float reciprocal estimate is off by less than one and a single
// remainder check makes it exact.
__device__ __forceinline__ int prob_one(int c0, int c1) {
    const int sum = c0 + c1, n = c1 << 12;
    int q = int(float(n) * __builtin_amdgcn_rcpf(float(sum)));
    const int r = n - q * sum;
    q += (r >= sum) - (r < 0);
    return q;
}

// ---- the engine's LDS image ------------------------------------------------------------------
struct Lds {
    int     ctx[kContexts];
    int     c0[kLevels][kTreeNodes], c1[kLevels][kTreeNodes];
    int     count[512][kMapSyms];
    uint8_t rank_of[512][kMapSyms], sym_at[512][kMapSyms];
    // least squares (efforts 2/3): vectors are [s | b(n) | A(n x n)] (NBLIC.c:213-215)
    i64     E[kLsqMaxM + 1];                     // statistics of the current row so far
    i64     D[kLsqMaxM + 1];                     // E + F at the current pixel
    i64     M[kLsqMaxN][kLsqMaxN + 1];           // augmented system [A | b]
    i64     vn[kLsqMaxN];
    i64     term[kLsqMaxN];
    // the three image rows the causal taps can touch, as a ring (row r lives at r % 3): taps come
    // from LDS instead of a store -> fence -> load round trip through L2 for every pixel
    uint8_t rows[3][kRowCache];
};

__device__ __forceinline__ void wave_sync() { __syncthreads(); }     // one wave per block: orders LDS traffic between lanes

// ---- least-squares predictor, lane-parallel ----------------------------------------------------
// Q12 prediction from the regularised normal equations (NBLIC.c:210-239 with the solve of :112-161).
// Uniform result (every lane returns the same values).
__device__ int lsq_predict_wave(Lds &S, int n, int m, const i64 *F, i64 bias, i64 *px_q12) {
    const int lane = int(threadIdx.x), cols = n + 1;
    for (int k = lane; k < m; k += 64) S.D[k] = k ? S.E[k] + F[k] : 0;
    wave_sync();
    for (int e = lane; e < n * cols; e += 64) {
        const int i = e / cols, j = e - i * cols;
        S.M[i][j] = j == n ? S.D[1 + i] + bias * (1 << kFb3) : S.D[1 + n + i * n + j] + (i == j ? bias * n : 0);
    }
    wave_sync();
    for (int k = 0; k + 1 < n; k++) {                                   // forward elimination, partial pivoting
        int piv = k;
        i64 best = abs64(S.M[k][k]);
        for (int i = k + 1; i < n; i++) { const i64 v = abs64(S.M[i][k]); if (v > best) { best = v; piv = i; } }   // strict: first maximum wins
        if (piv != k) {
            for (int j = lane; j < cols; j += 64) { const i64 t = S.M[k][j]; S.M[k][j] = S.M[piv][j]; S.M[piv][j] = t; }
            wave_sync();
        }
        const i64 d = S.M[k][k];
        if (d == 0) return 0;
        i64 upd[2]; int cnt = 0;
        for (int e = lane; e < n * cols; e += 64) {
            const int i = e / cols, j = e - i * cols;
            upd[cnt++] = (i > k && j > k) ? S.M[i][j] - mulw(S.M[k][j], S.M[i][k]) / d : 0;
        }
        wave_sync();                                                   // every lane has read column k before it is cleared
        cnt = 0;
        for (int e = lane; e < n * cols; e += 64) {
            const int i = e / cols, j = e - i * cols;
            if (i > k && j > k) S.M[i][j] = upd[cnt]; else if (i > k && j == k) S.M[i][j] = 0;
            cnt++;
        }
        wave_sync();
    }
    for (int k = n - 1; k > 0; k--) {                                   // back substitution on b only
        const i64 d = S.M[k][k];
        if (d == 0) return 0;
        for (int i = lane; i < k; i += 64) { S.M[i][n] -= mulw(S.M[k][n], S.M[i][k]) / d; S.M[i][k] = 0; }
        wave_sync();
    }
    for (int k = lane; k < n; k += 64) {
        const i64 d = S.M[k][k];
        S.term[k] = (mulw(mulw(S.M[k][n], S.vn[k]), 1 << kFb2) + (d >> 1)) / d;
    }
    wave_sync();
    i64 px = i64(kMid) << kFb1;
    for (int k = 0; k < n; k++) px += S.term[k];
    wave_sync();
    *px_q12 = clip64(px, 0, i64(kMaxVal) << kFb1);
    return 1;
}

// fold the newly coded pixel into the running statistics (NBLIC.c:242-283), one or two entries per lane
__device__ void lsq_update_wave(Lds &S, int n, int m, i64 *B, int x, i64 s_curr, i64 s_sum) {
    const i64 xc = x - kMid;
    s_sum = clip64(s_sum + (1 << kFb1), 1 << kFb1, 16 << kFb1);
    const i64 half = s_sum >> 1;
    for (int k = int(threadIdx.x); k < m; k += 64) {
        i64 sample;
        if (k == 0) sample = s_curr;
        else if (k <= n) sample = (mulw(xc * S.vn[k - 1], i64(1) << (4 + kFb1 + kFb1)) + half) / s_sum;
        else { const int r = (k - 1 - n) / n, c = (k - 1 - n) - r * n; sample = (mulw(S.vn[r] * S.vn[c], i64(1) << (4 + kFb2 + kFb1)) + half) / s_sum; }
        const int ab = k ? kDecayV : kDecayS;
        const i64 b = decay(B[k], ab) + sample;
        B[k] = b;
        S.E[k] = decay(S.E[k], ab) + b;
    }
    wave_sync();
}

// once per row: right-to-left accumulation of the row-above statistics (NBLIC.c:186-204); a lane owns a channel
__device__ void lsq_row_prepare_wave(int m, i64 *Frow, const i64 *Brow, int w) {
    for (int k = int(threadIdx.x); k < m; k += 64) {
        const int ab = k ? kDecayV : kDecayS;
        i64 carry = 0;
        for (int j = w - 1; j >= 0; j--) {
            const i64 f = carry + Brow[size_t(j) * m + k];
            Frow[size_t(j) * m + k] = f;
            carry = decay(f, ab);
        }
    }
    __threadfence_block();
    wave_sync();
}

// ---- the engine --------------------------------------------------------------------------------
template <bool DEC>
__device__ void run_engine(const SerialArgs &a, Lds &S) {
    const int w = a.w, h = a.h, near = a.near, k_step = a.k_step;
    const int n = a.effort == 2 ? 6 : (a.effort == 3 ? 10 : 0);                    // N_LIST, NBLIC.c:88
    const int m = 1 + n + n * n;
    const bool lane0 = threadIdx.x == 0;
    uint8_t *img = a.img;
    i64 *Brow = a.stats, *Frow = a.stats + size_t(w) * m;
    i64 bias = kBiasInit;

    DevCoder rc{a.stream + kHeaderBytes, a.stream + a.stream_cap, 0u, 0xFFFFFFFFu, 0u, false};
    rc.store = lane0;
    if (DEC) for (int k = 0; k < 4; k++) rc.window = (rc.window << 8) | *rc.p++;

    const bool cached = w <= kRowCache;
    auto pix = [&](int r, int c) {
        return cached ? int(S.rows[r % 3][c]) : int(img[size_t(r) * size_t(w) + size_t(c)]);
    };

    for (int i = 0; i < h; i++) {
        int err = 0;
        if (cached && !DEC) {                                                      // encoder: the row's ORIGINAL pixels; each is replaced by its reconstruction once coded
            for (int c = int(threadIdx.x); c < w; c += 64) S.rows[i % 3][c] = img[size_t(i) * size_t(w) + size_t(c)];
            wave_sync();
        }
        if (n > 0) {                                                               // NBLIC.c:817-820
            for (int k = int(threadIdx.x); k < m; k += 64) S.E[k] = 0;
            lsq_row_prepare_wave(m, Frow, Brow, w);
        }
        for (int j = 0; j < w; j++) {
            Taps t = sample_taps(pix, w, i, j);
            i64 b1 = 0, b2 = 0, p1 = 0, p2 = 0;
            int ok1 = 0, ok2 = 0, px0;
            i64 *B = nullptr, *F = nullptr;
            if (n > 0) {                                                           // NBLIC.c:831-846
                if (threadIdx.x < unsigned(n)) {
                    const int order[kLsqMaxN] = {t.a, t.b, t.c, t.d, t.e, t.f, t.t, t.h, t.q, t.g};
                    int v = order[0];
                    for (int k = 1; k < kLsqMaxN; k++) if (int(threadIdx.x) == k) v = order[k];
                    S.vn[threadIdx.x] = v - kMid;
                }
                wave_sync();
                B = Brow + size_t(j) * m; F = Frow + size_t(j) * m;
                b1 = bias * kBiasCoef / (kBiasCoef + 1);
                b2 = bias * (kBiasCoef + 1) / kBiasCoef;
                b1 = clip64(clip64(b1, -1, bias - 1), 0, kBiasMax);
                b2 = clip64(clip64(b2, bias + 1, kBiasMax + 1), 0, kBiasMax);
                ok1 = lsq_predict_wave(S, n, m, F, b1, &p1);
                ok2 = lsq_predict_wave(S, n, m, F, b2, &p2);
            }
            if (ok1) px0 = int((p1 + (1 << (kFb1 - 1))) >> kFb1);
            else { px0 = predict(t); p1 = i64(px0) << kFb1; }

            Level L = quantise(activity(t, err));
            int adr = context_address(t, L.qu, px0);
            int v = S.ctx[adr];
            int sign = bias_sign(v), px = bias_apply(v, px0);
            int mk = px * 2 + sign;

            auto step = [&](int qu, int qv, int node, int bin) {                   // NBLIC.c:628-637
                int u0 = S.c0[qu][node], u1 = S.c1[qu][node], v0 = S.c0[qv][node], v1 = S.c1[qv][node];
                int prob = mix_prob(prob_one(u0, u1), prob_one(v0, v1), L.qw);
                bin = coder_bin<DEC>(rc, bin, uint32_t(prob));
                Counter cu{u0, u1};
                counter_add(cu, bin, kWeightOne - L.qw);
                if (qu == qv) counter_add(cu, bin, L.qw);                          // same counter takes both weights
                S.c0[qu][node] = cu.c0; S.c1[qu][node] = cu.c1;
                if (qu != qv) {
                    Counter cv{v0, v1};
                    counter_add(cv, bin, L.qw);
                    S.c0[qv][node] = cv.c0; S.c1[qv][node] = cv.c1;
                }
                return bin;
            };

            int y;
            if (!DEC) {
                y = residual_to_symbol(pix(i, j), px, sign, near);
                walk_symbol(k_step, L.qu, L.qv, y < kMapSyms ? int(S.rank_of[mk][y]) : y, step);
            } else {
                int z = walk_symbol(k_step, L.qu, L.qv, -1, step);
                y = z < kMapSyms ? int(S.sym_at[mk][z]) : z;
            }
            if (y < kMapSyms) {                                                    // NBLIC.c:497-523
                int z = S.rank_of[mk][y];
                int c = S.count[mk][z] + 1;
                int c_up = z > 0 ? S.count[mk][z - 1] : 0x7FFFFFFF;
                int other = z > 0 ? int(S.sym_at[mk][z - 1]) : 0;
                wave_sync();                                                       // every lane has read before any lane writes
                if (c_up < c) {
                    S.count[mk][z] = c_up; S.count[mk][z - 1] = c;
                    S.sym_at[mk][z] = uint8_t(other); S.sym_at[mk][z - 1] = uint8_t(y);
                    S.rank_of[mk][y] = uint8_t(z - 1); S.rank_of[mk][other] = uint8_t(z);
                } else {
                    S.count[mk][z] = c;
                }
            }
            int xr = symbol_to_pixel(y, px, sign, near);
            if (lane0) img[size_t(i) * size_t(w) + size_t(j)] = uint8_t(xr);
            err = clip_err(xr, px0);
            S.ctx[adr] = bias_update(v, err);
            if (cached) S.rows[i % 3][j] = uint8_t(xr);
            else __threadfence_block();
            wave_sync();                                                           // the pixel is visible to every lane's next taps

            if (n > 0) {                                                           // NBLIC.c:882-893
                i64 xq = i64(xr) << kFb1;
                i64 s_curr = abs64(p1 - xq);
                i64 s_sum = (S.E[0] + F[0]) + s_curr * kDecayS / (kDecayS - 1);
                wave_sync();
                lsq_update_wave(S, n, m, B, xr, s_curr, s_sum);
                if (ok1 && ok2) bias = (abs64(p1 - xq) > abs64(p2 - xq)) ? b2 : b1;
            }
        }
    }
    if (!DEC) for (int k = 0; k < 4; k++) {                                        // NBLIC.c:576-586
        if (rc.p < rc.end) { if (lane0) *rc.p = uint8_t(rc.lo >> 24); } else rc.overflow = true;
        rc.p++; rc.lo <<= 8;
    }
    if (lane0) *a.len_out = rc.overflow ? -1L : long(rc.p - a.stream);
}

__global__ void __launch_bounds__(64) k_serial_codec(SerialArgs a) {
    __shared__ Lds S;
    for (int k = int(threadIdx.x); k < kContexts; k += 64) S.ctx[k] = 0;
    for (int k = int(threadIdx.x); k < kLevels * kTreeNodes; k += 64) { (&S.c0[0][0])[k] = kWeightOne; (&S.c1[0][0])[k] = kWeightOne; }
    for (int k = int(threadIdx.x); k < 512 * kMapSyms; k += 64) {
        int s = k % kMapSyms;
        (&S.count[0][0])[k] = 2 * (kMapSyms - 1 - s); (&S.rank_of[0][0])[k] = uint8_t(s); (&S.sym_at[0][0])[k] = uint8_t(s);
    }
    __syncthreads();
    if (a.decode) run_engine<true>(a, S); else run_engine<false>(a, S);
}

// ---- QNBLIC decoder (QNBLIC.c:493-555): one lane, context table + frequency tables in LDS ----
struct QDecArgs {
    uint8_t *img; const uint16_t *words; size_t n_words; size_t pos;
    int h, w;
    const uint32_t *freq, *start; const uint8_t *slot;
    int *status;
};

__global__ void __launch_bounds__(64) k_serial_qdecode(QDecArgs a) {
    __shared__ int ctx[3072];
    __shared__ uint32_t freq[12 * 256], start[12 * 256];
    for (int k = int(threadIdx.x); k < 3072; k += 64) { ctx[k] = 0; freq[k] = a.freq[k]; start[k] = a.start[k]; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    const int w = a.w;
    uint8_t *img = a.img;
    auto pix = [&](int r, int c) { return int(img[size_t(r) * size_t(w) + size_t(c)]); };
    size_t pos = a.pos;
    bool bad = pos + 2 > a.n_words;
    uint32_t x = bad ? 0u : ((uint32_t(a.words[pos]) << 16) | a.words[pos + 1]);
    pos += 2;
    for (int i = 0; i < a.h && !bad; i++) {
        int err = 0;
        for (int j = 0; j < w; j++) {
            const Taps n = sample_taps_q(pix, w, i, j);
            const int px0 = predict_q(n), qd = level_q(n, err);
            const int adr = context_address_q(n, qd, px0);
            const int v = ctx[adr];
            const int sign = (v >> 10) & 1;
            const int px = iclip(px0 + (v >> 11) + sign, 0, kMaxVal);
            const uint32_t low = x & 32767u;
            const int y = a.slot[size_t(qd) * 32768 + low];
            x = (x >> 15) * freq[qd * 256 + y] + low - start[qd * 256 + y];
            if (x < 65536u) { if (pos >= a.n_words) { bad = true; break; } x = (x << 16) | a.words[pos++]; }
            const int px_out = symbol_to_pixel(y, px, sign, 0);
            img[size_t(i) * size_t(w) + size_t(j)] = uint8_t(px_out);
            err = px_out - px0;
            ctx[adr] = (v * 128 - v + err * 2048 + 63) >> 7;
        }
    }
    *a.status = bad ? -1 : 0;
}

// ---- host side -------------------------------------------------------------------------------
#define SE_OK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    fprintf(stderr, "[nblic_amd] %s failed: %s\n", #call, hipGetErrorString(e_)); return -1; } } while (0)

bool SerialEngine::init() {
    if (hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) return false;
    return hipMalloc((void **)&d_len, sizeof(long)) == hipSuccess && hipMalloc((void **)&d_status, sizeof(int)) == hipSuccess;
}

void SerialEngine::destroy() {
    hipFree(d_img); hipFree(d_stream); hipFree(d_stats); hipFree(d_len); hipFree(d_qtab); hipFree(d_status);
    d_qtab = nullptr; d_status = nullptr;
    if (stream) hipStreamDestroy(stream);
    d_img = d_stream = nullptr; d_stats = nullptr; d_len = nullptr; stream = nullptr;
}

static long run_serial(SerialEngine &e, uint8_t *host_img, const uint8_t *host_in, size_t in_len, uint8_t *host_out,
                       int h, int w, int near, int k_step, int effort, bool decode, int device) {
    SE_OK(hipSetDevice(device));
    size_t n = size_t(h) * size_t(w);
    size_t cap = decode ? in_len + 8 : 2 * n + 4096;
    int lsq_n = effort == 2 ? 6 : (effort == 3 ? 10 : 0);
    size_t stats = lsq_n ? 2 * size_t(w) * size_t(1 + lsq_n + lsq_n * lsq_n) : 0;
    if (n > e.img_cap) { hipFree(e.d_img); e.d_img = nullptr; SE_OK(hipMalloc((void **)&e.d_img, n)); e.img_cap = n; }
    if (cap > e.stream_cap) { hipFree(e.d_stream); e.d_stream = nullptr; SE_OK(hipMalloc((void **)&e.d_stream, cap)); e.stream_cap = cap; }
    if (stats > e.stats_cap) { hipFree(e.d_stats); e.d_stats = nullptr; SE_OK(hipMalloc((void **)&e.d_stats, stats * sizeof(int64_t))); e.stats_cap = stats; }
    if (stats) SE_OK(hipMemsetAsync(e.d_stats, 0, stats * sizeof(int64_t), e.stream));        // NBLIC.c:789
    if (decode) {
        SE_OK(hipMemsetAsync(e.d_stream, 0, cap, e.stream));
        SE_OK(hipMemcpyAsync(e.d_stream, host_in, in_len, hipMemcpyHostToDevice, e.stream));
    } else {
        SE_OK(hipMemcpyAsync(e.d_img, host_img, n, hipMemcpyHostToDevice, e.stream));
    }
    SerialArgs a{e.d_img, e.d_stream, cap, h, w, near, k_step, effort, decode ? 1 : 0, (long long *)e.d_stats, e.d_len};
    hipLaunchKernelGGL(k_serial_codec, dim3(1), dim3(64), 0, e.stream, a);
    long len = -1;
    SE_OK(hipMemcpyAsync(&len, e.d_len, sizeof(long), hipMemcpyDeviceToHost, e.stream));
    SE_OK(hipStreamSynchronize(e.stream));
    if (len < 0) { fprintf(stderr, "[nblic_amd] serial engine: stream buffer exhausted\n"); return -1; }
    SE_OK(hipMemcpy(host_img, e.d_img, n, hipMemcpyDeviceToHost));                           // reconstruction / decoded image
    if (!decode) SE_OK(hipMemcpy(host_out + kHeaderBytes, e.d_stream + kHeaderBytes, size_t(len) - kHeaderBytes, hipMemcpyDeviceToHost));
    return len;
}

long SerialEngine::encode(uint8_t *out, uint8_t *img, int h, int w, int near, int k_step, int effort, int device) {
    return run_serial(*this, img, nullptr, 0, out, h, w, near, k_step, effort, false, device);
}

// How many bytes of the caller's stream may be read.  The reference's decoder reads exactly the
// bytes the encoder wrote and its ABI carries no length (NBLIC.h:72), so the copy to the device
// is bounded by the end of the readable mapping that holds `p` and by the CLI's 2 B/px provision.
static size_t readable_span(const uint8_t *p, size_t want) {
    FILE *f = fopen("/proc/self/maps", "r");
    if (!f) return want;
    unsigned long lo, hi, addr = (unsigned long)p, end = 0;
    char perms[8], line[512];
    while (fgets(line, sizeof line, f)) {
        if (sscanf(line, "%lx-%lx %7s", &lo, &hi, perms) != 3 || perms[0] != 'r') { if (end) break; continue; }
        if (!end) { if (addr >= lo && addr < hi) end = hi; }
        else if (lo == end) end = hi;                                   // contiguous readable mapping
        else break;
    }
    fclose(f);
    if (!end) return want;
    size_t avail = size_t(end - addr);
    return avail < want ? avail : want;
}

int SerialEngine::decode(const uint8_t *in, uint8_t *img, int h, int w, int near, int k_step, int effort, int device) {
    size_t want = 2 * size_t(h) * size_t(w) + 4096;
    size_t len = readable_span(in, want);
    long r = run_serial(*this, img, in, len, nullptr, h, w, near, k_step, effort, true, device);
    return r < 0 ? -1 : 0;
}

long q_decode_tables(const uint16_t *in, size_t n_words, int *h, int *w, uint32_t *freq, uint32_t *start, uint8_t *slot);

int SerialEngine::qdecode(const uint16_t *in, uint8_t *img, int *h, int *w, long max_px, int device) {
    SE_OK(hipSetDevice(device));
    if (readable_span((const uint8_t *)in, 8) < 8) return -1;
    if (in[0] != (uint16_t)('Q' | ('0' << 8)) || in[1] != (uint16_t)('.' | ('2' << 8))) return -1;      // QNBLIC.c:475-486
    *h = in[2]; *w = in[3];
    if (*h <= 0 || *w <= 0 || long(*h) * long(*w) > max_px) return -1;
    const size_t n = size_t(*h) * size_t(*w);
    const size_t n_words = readable_span((const uint8_t *)in, 2 * n + 16384) / 2;
    const size_t tab_bytes = 2 * 12 * 256 * sizeof(uint32_t) + size_t(12) * 32768;
    uint8_t *tabs = (uint8_t *)malloc(tab_bytes);
    if (!tabs) return -1;
    uint32_t *freq = (uint32_t *)tabs, *start = freq + 12 * 256;
    uint8_t *slot = tabs + 2 * 12 * 256 * sizeof(uint32_t);
    long pos = q_decode_tables(in, n_words, h, w, freq, start, slot);
    if (pos < 0) { free(tabs); return -1; }
    const size_t stream_bytes = n_words * 2;
    if (n > img_cap) { hipFree(d_img); d_img = nullptr; SE_OK(hipMalloc((void **)&d_img, n)); img_cap = n; }
    if (stream_bytes > stream_cap) { hipFree(d_stream); d_stream = nullptr; SE_OK(hipMalloc((void **)&d_stream, stream_bytes)); stream_cap = stream_bytes; }
    if (!d_qtab) SE_OK(hipMalloc((void **)&d_qtab, tab_bytes));
    SE_OK(hipMemcpyAsync(d_qtab, tabs, tab_bytes, hipMemcpyHostToDevice, stream));
    SE_OK(hipMemcpyAsync(d_stream, in, stream_bytes, hipMemcpyHostToDevice, stream));
    QDecArgs a{d_img, (const uint16_t *)d_stream, n_words, size_t(pos), *h, *w, (const uint32_t *)d_qtab,
               (const uint32_t *)d_qtab + 12 * 256, d_qtab + 2 * 12 * 256 * sizeof(uint32_t), d_status};
    hipLaunchKernelGGL(k_serial_qdecode, dim3(1), dim3(64), 0, stream, a);
    int status = -1;
    SE_OK(hipMemcpyAsync(&status, d_status, sizeof(int), hipMemcpyDeviceToHost, stream));
    SE_OK(hipStreamSynchronize(stream));
    free(tabs);
    if (status != 0) return -1;
    SE_OK(hipMemcpy(img, d_img, n, hipMemcpyDeviceToHost));
    return 0;
}

}  // namespace nblic

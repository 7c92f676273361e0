// kernels_e1.hip -- hand-written gfx950 kernels for the -e1 lossless NBLIC encoder.
//
// The reference codes pixels in raster order through four pieces of adaptive state
// (NBLIC.c:752-756: context-bias table, symbol re-mappers, binary counter trees, coder
// interval).  Each table ENTRY is an independent chain, so the state is replayed one key
// at a time over a stable (raster-order-preserving) partition of the work items:
//
//   k_predict            S1  stateless, one lane per pixel        -> rec1[t]
//   partition by adr     (count -> scan -> scatter)               -> s2rec[]  grouped by context
//   k_bias_chains        S2  one LANE per context chain           -> pxs[t] = px | sign<<8
//   partition by px|sign                                          -> s3rec[]  grouped by re-mapper
//   k_mapper_chains      S3  one LANE per re-mapper chain, state in LDS -> z[t]
//   k_count_bins/k_emit_bins  S4 stateless                        -> events[r]
//   partition by counter (even / odd trees)                       -> touch[]  grouped by counter
//   k_counter_chains     S5  one WAVE per counter chain (scan within a halving epoch)
//   k_mix                probability mix + pack                   -> coded[r] (u16) for the host coder
//
// Wave64 throughout; no MFMA (nothing here is a contraction).  All kernels take one image;
// the host pipeline runs several images concurrently on separate HIP streams.
#include <hip/hip_runtime.h>
#include "model.h"
#include "kernels_e1.h"

namespace nblic {

// ------------------------------------------------------------------------------------------
// wave helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return int(threadIdx.x & 63); }
__device__ __forceinline__ uint64_t lanes_below() { return (1ull << lane_id()) - 1ull; }

// Mask of the valid lanes holding the same BITS-bit key as this lane.
template <int BITS>
__device__ __forceinline__ uint64_t match_lanes(uint32_t key, bool valid) {
    uint64_t m = __ballot(valid);
#pragma unroll
    for (int b = 0; b < BITS; b++) {
        bool bit = (key >> b) & 1u;
        uint64_t set = __ballot(valid && bit);
        m &= bit ? set : ~set;
    }
    return valid ? m : 0ull;
}

// inclusive wave prefix sum (64 lanes)
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(v, d, 64);
        if (lane_id() >= d) v += o;
    }
    return v;
}

// Same prefix sum on the DPP crossbar (no LDS round trips): Kogge-Stone inside each 16-lane
// row with row_shr, then row_bcast:15 into rows 1/3 and row_bcast:31 into rows 2/3.
__device__ __forceinline__ uint32_t wave_scan_incl_dpp(uint32_t v) {
    v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x111, 0xf, 0xf, true));    // row_shr:1
    v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x112, 0xf, 0xf, true));    // row_shr:2
    v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x114, 0xf, 0xf, true));    // row_shr:4
    v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x118, 0xf, 0xf, true));    // row_shr:8
    v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x142, 0xa, 0xf, false));   // row_bcast:15 -> rows 1,3
    v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x143, 0xc, 0xf, false));   // row_bcast:31 -> rows 2,3
    return v;
}

__device__ __forceinline__ uint32_t read_lane(uint32_t v, int l) { return uint32_t(__builtin_amdgcn_readlane(int(v), l)); }

// Stages, for each of the wave's 64 private streams, the next <= 64 records into LDS with
// COALESCED global reads: for source lane l the whole wave reads stream l's next run
// (one contiguous <= 512 B / 256 B request), then every lane consumes its own row.  Rows are
// padded to 65 records so the column reads of the consume loop are bank-conflict free.
constexpr int kStageRow = 65;

template <class Rec>
__device__ __forceinline__ void stage_streams(const Rec *__restrict__ src, Rec *stage, uint32_t r, uint32_t rem, uint64_t active) {
    const int lane = lane_id();
    for (int l0 = 0; l0 < 64; l0 += 8) {
        if (((active >> l0) & 0xFFull) == 0ull) continue;                       // wave-uniform
        Rec tmp[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            uint32_t base = read_lane(r, l0 + u), cnt = min(read_lane(rem, l0 + u), 64u);
            if (uint32_t(lane) < cnt) tmp[u] = src[base + lane];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) stage[(l0 + u) * kStageRow + lane] = tmp[u];
    }
}

// ------------------------------------------------------------------------------------------
// S1: predictor, activity level, context address.  NBLIC.c:287-410.
// grid = (ceil(w/256), h); one lane per pixel.  err_prev (the clipped error of the pixel to
// the left, NBLIC.c:808/:878) is recomputed from the input image -- in lossless mode the
// reconstruction IS the input, so S1 carries no state at all.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_predict(const uint8_t *__restrict__ img, int h, int w, int row0, int rows,
                                                 uint32_t *__restrict__ rec1) {
    int j = int(blockIdx.x) * 256 + int(threadIdx.x);
    int i = row0 + int(blockIdx.y);
    if (j >= w) return;
    auto pix = [&](int r, int c) { return int(img[size_t(r) * size_t(w) + size_t(c)]); };
    Taps n = sample_taps(pix, w, i, j);
    int px0 = predict(n);
    int err_prev = 0;
    if (j > 0) {
        Taps m = sample_taps(pix, w, i, j - 1);
        err_prev = clip_err(n.a, predict(m));
    }
    Level L = quantise(activity(n, err_prev));
    rec1[size_t(blockIdx.y) * size_t(w) + size_t(j)] = pack_s1(px0, context_address(n, L.qu, px0), L);
}

// ------------------------------------------------------------------------------------------
// Stable partition = per-segment histogram -> exclusive scan of table[key][segment] ->
// ranked scatter.  One wave owns one segment (a contiguous run of items in raster order)
// and keeps its histogram / running offsets in its own LDS slice, so no barriers are needed.
// ------------------------------------------------------------------------------------------
template <int NKEYS>
__device__ __forceinline__ void lds_fill(uint32_t *slice, uint32_t v) {
    for (int k = lane_id(); k < NKEYS; k += 64) slice[k] = v;
}

// ---- partition 1: pixels by context address (2048 keys) ----------------------------------
__global__ void __launch_bounds__(256) k_adr_count(const uint32_t *__restrict__ rec1, uint32_t n, SegPlan plan,
                                                   uint32_t *__restrict__ table) {
    __shared__ uint32_t lds[4][kContexts];
    int seg = int(blockIdx.x) * 4 + int(threadIdx.x >> 6);
    if (seg >= plan.nseg) return;
    uint32_t *hist = lds[threadIdx.x >> 6];
    lds_fill<kContexts>(hist, 0);
    uint32_t lo = uint32_t(seg) * plan.seg_len, hi = min(n, lo + plan.seg_len);
    for (uint32_t base = lo; base < hi; base += 64) {
        uint32_t t = base + lane_id();
        if (t < hi) atomicAdd(&hist[s1_adr(rec1[t])], 1u);
    }
    for (int k = lane_id(); k < kContexts; k += 64) table[size_t(k) * plan.nseg + seg] = hist[k];
}

__global__ void __launch_bounds__(256) k_adr_scatter(const uint32_t *__restrict__ rec1, const uint8_t *__restrict__ x,
                                                     uint32_t n, SegPlan plan, const uint32_t *__restrict__ table,
                                                     uint2 *__restrict__ s2rec) {
    __shared__ uint32_t lds[4][kContexts];
    int seg = int(blockIdx.x) * 4 + int(threadIdx.x >> 6);
    if (seg >= plan.nseg) return;
    uint32_t *off = lds[threadIdx.x >> 6];
    for (int k = lane_id(); k < kContexts; k += 64) off[k] = table[size_t(k) * plan.nseg + seg];
    uint32_t lo = uint32_t(seg) * plan.seg_len, hi = min(n, lo + plan.seg_len);
    for (uint32_t base = lo; base < hi; base += 64) {
        uint32_t t = base + lane_id();
        bool valid = t < hi;
        uint32_t r = valid ? rec1[t] : 0u;
        uint32_t key = uint32_t(s1_adr(r));
        uint64_t same = match_lanes<11>(key, valid);
        if (valid) {
            uint32_t rank = __popcll(same & lanes_below());
            uint32_t pos = off[key] + rank;
            int px0 = s1_px0(r);
            int err = clip_err(int(x[t]), px0);
            s2rec[pos] = make_uint2(t, uint32_t(px0) | (uint32_t(err & 0xFF) << 8));
            if (rank == 0) off[key] += uint32_t(__popcll(same));
        }
    }
}

// ---- S2: context-bias chains, one lane per context (NBLIC.c:413-428) ---------------------
__global__ void __launch_bounds__(64) k_bias_chains(const uint2 *__restrict__ s2rec, const uint32_t *__restrict__ table,
                                                    SegPlan plan, uint32_t n, int *__restrict__ ctx_state,
                                                    uint16_t *__restrict__ pxs) {
    __shared__ uint2 stage[64 * kStageRow];
    const int lane = int(threadIdx.x);
    const int key = int(blockIdx.x) * 64 + lane;
    uint32_t r = table[size_t(key) * plan.nseg];
    const uint32_t end = key + 1 < kContexts ? table[size_t(key + 1) * plan.nseg] : n;
    int v = ctx_state[key];
    for (;;) {
        uint32_t rem = end - r;
        uint64_t active = __ballot(rem > 0u);
        if (active == 0ull) break;
        stage_streams(s2rec, stage, r, rem, active);
        __syncthreads();
        const int mine = int(min(rem, 64u));
        for (int i = 0; i < 64; i++) {
            if (__ballot(i < mine) == 0ull) break;
            if (i < mine) {
                uint2 cur = stage[lane * kStageRow + i];
                int px0 = int(cur.y & 0xFF);
                int err = int(int8_t(cur.y >> 8));
                pxs[cur.x] = uint16_t(bias_apply(v, px0) | (bias_sign(v) << 8));
                v = bias_update(v, err);
            }
        }
        r += uint32_t(mine);
        __syncthreads();
    }
    ctx_state[key] = v;
}

// ---- partition 2: pixels by (px, sign) (512 keys); symbols >= 20 bypass the re-mapper -----
__device__ __forceinline__ bool mapper_item(const uint8_t *x, const uint16_t *pxs, uint32_t t, uint32_t &key, int &y) {
    uint32_t ps = pxs[t];
    int px = int(ps & 0xFF), sign = int(ps >> 8);
    y = residual_to_symbol(int(x[t]), px, sign, 0);
    key = uint32_t(px) * 2u + uint32_t(sign);
    return y < kMapSyms;
}

__global__ void __launch_bounds__(256) k_map_count(const uint8_t *__restrict__ x, const uint16_t *__restrict__ pxs, uint32_t n,
                                                   SegPlan plan, uint32_t *__restrict__ table) {
    __shared__ uint32_t lds[4][512];
    int seg = int(blockIdx.x) * 4 + int(threadIdx.x >> 6);
    if (seg >= plan.nseg) return;
    uint32_t *hist = lds[threadIdx.x >> 6];
    lds_fill<512>(hist, 0);
    uint32_t lo = uint32_t(seg) * plan.seg_len, hi = min(n, lo + plan.seg_len);
    for (uint32_t base = lo; base < hi; base += 64) {
        uint32_t t = base + lane_id(), key; int y;
        if (t < hi && mapper_item(x, pxs, t, key, y)) atomicAdd(&hist[key], 1u);
    }
    for (int k = lane_id(); k < 512; k += 64) table[size_t(k) * plan.nseg + seg] = hist[k];
}

__global__ void __launch_bounds__(256) k_map_scatter(const uint8_t *__restrict__ x, const uint16_t *__restrict__ pxs, uint32_t n,
                                                     SegPlan plan, const uint32_t *__restrict__ table,
                                                     uint32_t *__restrict__ s3rec, uint8_t *__restrict__ z) {
    __shared__ uint32_t lds[4][512];
    int seg = int(blockIdx.x) * 4 + int(threadIdx.x >> 6);
    if (seg >= plan.nseg) return;
    uint32_t *off = lds[threadIdx.x >> 6];
    for (int k = lane_id(); k < 512; k += 64) off[k] = table[size_t(k) * plan.nseg + seg];
    uint32_t lo = uint32_t(seg) * plan.seg_len, hi = min(n, lo + plan.seg_len);
    for (uint32_t base = lo; base < hi; base += 64) {
        uint32_t t = base + lane_id(), key = 0; int y = 0;
        bool in = t < hi;
        bool valid = in && mapper_item(x, pxs, t, key, y);
        if (in && !valid) z[t] = uint8_t(y);                      // y >= 20 codes as itself (NBLIC.c:488)
        uint64_t same = match_lanes<9>(key, valid);
        if (valid) {
            uint32_t rank = __popcll(same & lanes_below());
            s3rec[off[key] + rank] = t | (uint32_t(y) << 27);
            if (rank == 0) off[key] += uint32_t(__popcll(same));
        }
    }
}

// ---- S3: re-mapper chains, one lane per (px, sign) (NBLIC.c:470-523) ----------------------
// Per-lane state (20 ranks, 20 symbols, 20 counts) lives in LDS as [entry][lane] so that the
// 64 lanes of a wave hit 64 different banks whatever entry each of them indexes.
__global__ void __launch_bounds__(64) k_mapper_chains(const uint32_t *__restrict__ s3rec, const uint32_t *__restrict__ table,
                                                      SegPlan plan, uint32_t n_items_total_dummy, const uint32_t *__restrict__ total,
                                                      int *__restrict__ map_state, uint8_t *__restrict__ z) {
    __shared__ int rank_of[kMapSyms][64], sym_at[kMapSyms][64], count[kMapSyms][64];
    __shared__ uint32_t stage[64 * kStageRow];
    const int lane = int(threadIdx.x);
    const int key = int(blockIdx.x) * 64 + lane;
    int *st = map_state + size_t(key) * (3 * kMapSyms);
    for (int k = 0; k < kMapSyms; k++) {
        rank_of[k][lane] = st[k]; sym_at[k][lane] = st[kMapSyms + k]; count[k][lane] = st[2 * kMapSyms + k];
    }
    uint32_t r = table[size_t(key) * plan.nseg];
    const uint32_t end = key + 1 < 512 ? table[size_t(key + 1) * plan.nseg] : *total;
    (void)n_items_total_dummy;
    for (;;) {
        uint32_t rem = end - r;
        uint64_t active = __ballot(rem > 0u);
        if (active == 0ull) break;
        stage_streams(s3rec, stage, r, rem, active);
        __syncthreads();
        const int mine = int(min(rem, 64u));
        for (int i = 0; i < 64; i++) {
            if (__ballot(i < mine) == 0ull) break;
            if (i < mine) {
                uint32_t cur = stage[lane * kStageRow + i];
                int y = int(cur >> 27);
                int zz = rank_of[y][lane];
                z[cur & 0x7FFFFFFu] = uint8_t(zz);
                int c = count[zz][lane] + 1;
                count[zz][lane] = c;
                if (zz > 0) {
                    int c_up = count[zz - 1][lane];
                    if (c_up < c) {                               // overtake the rank above
                        int other = sym_at[zz - 1][lane];
                        count[zz][lane] = c_up;  count[zz - 1][lane] = c;
                        sym_at[zz][lane] = other; sym_at[zz - 1][lane] = y;
                        rank_of[y][lane] = zz - 1; rank_of[other][lane] = zz;
                    }
                }
            }
        }
        r += uint32_t(mine);
        __syncthreads();
    }
    for (int k = 0; k < kMapSyms; k++) {
        st[k] = rank_of[k][lane]; st[kMapSyms + k] = sym_at[k][lane]; st[2 * kMapSyms + k] = count[k][lane];
    }
}

// ---- S4: binarisation (NBLIC.c:640-679); path depends on (qu,qv,qw,z) only ----------------
__global__ void __launch_bounds__(256) k_count_bins(const uint32_t *__restrict__ rec1, const uint8_t *__restrict__ z, uint32_t n,
                                                    uint8_t *__restrict__ cnt) {
    uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n) return;
    Level L = s1_level(rec1[t]);
    int c = 0;
    walk_symbol(kMinKStep, L.qu, L.qv, int(z[t]), [&](int, int, int, int bin) { c++; return bin; });
    cnt[t] = uint8_t(c);
}

__global__ void __launch_bounds__(256) k_emit_bins(const uint32_t *__restrict__ rec1, const uint8_t *__restrict__ z, uint32_t n,
                                                   const uint32_t *__restrict__ ev_off, uint32_t *__restrict__ events) {
    uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n) return;
    Level L = s1_level(rec1[t]);
    uint32_t *out = events + ev_off[t];
    walk_symbol(kMinKStep, L.qu, L.qv, int(z[t]), [&](int qu, int qv, int node, int bin) {
        *out++ = pack_event(qu, qv, node, L.qw, bin);
        return bin;
    });
}

// ---- partition 3: counter touches (4096 keys = parity | tree/2 | node) --------------------
// Adjacent levels differ by one, so of an event's two trees exactly one is even and one is
// odd: per parity an event contributes AT MOST one touch, which keeps the ranking a plain
// one-key match.  An event whose two trees coincide touches that one counter twice
// (weights 32-qw then qw, NBLIC.c:635-636) and is carried as a single "double" item.
struct Touch { bool valid; uint32_t key; uint32_t payload; };

__device__ __forceinline__ Touch touch_of(uint32_t e, int parity) {
    int qu = ev_qu(e), qv = ev_qv(e), node = ev_node(e), qw = ev_qw(e), bin = ev_bin(e);
    Touch t{false, 0u, 0u};
    int tree, w1, w2 = 0, slot = 0, dbl = 0;
    if (qu == qv) {
        if ((qu & 1) != parity) return t;
        tree = qu; w1 = kWeightOne - qw; w2 = qw; dbl = 1;
    } else if ((qu & 1) == parity) {
        tree = qu; w1 = kWeightOne - qw;
    } else {
        if (qw == 0) return t;                   // weight-0 touch: no state change, its P is multiplied by 0
        tree = qv; w1 = qw; slot = 1;
    }
    t.valid = true;
    t.key = uint32_t(parity) * 2048u + uint32_t(tree >> 1) * 256u + uint32_t(node);
    t.payload = uint32_t(w1) | (uint32_t(w2) << 6) | (uint32_t(bin) << 12) | (uint32_t(slot) << 13) | (uint32_t(dbl) << 14);
    return t;
}

__global__ void __launch_bounds__(256) k_touch_count(const uint32_t *__restrict__ events, uint32_t n_ev, SegPlan plan,
                                                     uint32_t *__restrict__ table) {
    __shared__ uint32_t lds[4][4096];
    int seg = int(blockIdx.x) * 4 + int(threadIdx.x >> 6);
    if (seg >= plan.nseg) return;
    uint32_t *hist = lds[threadIdx.x >> 6];
    lds_fill<4096>(hist, 0);
    uint32_t lo = uint32_t(seg) * plan.seg_len, hi = min(n_ev, lo + plan.seg_len);
    for (uint32_t base = lo; base < hi; base += 64) {
        uint32_t r = base + lane_id();
        if (r < hi) {
            uint32_t e = events[r];
            Touch a = touch_of(e, 0), b = touch_of(e, 1);
            if (a.valid) atomicAdd(&hist[a.key], 1u);
            if (b.valid) atomicAdd(&hist[b.key], 1u);
        }
    }
    for (int k = lane_id(); k < 4096; k += 64) table[size_t(k) * plan.nseg + seg] = hist[k];
}

__global__ void __launch_bounds__(256) k_touch_scatter(const uint32_t *__restrict__ events, uint32_t n_ev, SegPlan plan,
                                                       const uint32_t *__restrict__ table, uint2 *__restrict__ touch) {
    __shared__ uint32_t lds[4][4096];
    int seg = int(blockIdx.x) * 4 + int(threadIdx.x >> 6);
    if (seg >= plan.nseg) return;
    uint32_t *off = lds[threadIdx.x >> 6];
    for (int k = lane_id(); k < 4096; k += 64) off[k] = table[size_t(k) * plan.nseg + seg];
    uint32_t lo = uint32_t(seg) * plan.seg_len, hi = min(n_ev, lo + plan.seg_len);
    for (uint32_t base = lo; base < hi; base += 64) {
        uint32_t r = base + lane_id();
        uint32_t e = r < hi ? events[r] : 0u;
#pragma unroll
        for (int parity = 0; parity < 2; parity++) {
            Touch t = touch_of(e, parity);
            bool valid = r < hi && t.valid;
            uint64_t same = match_lanes<11>(t.key, valid);       // parity bit is common to the pass
            if (valid) {
                uint32_t rank = __popcll(same & lanes_below());
                touch[off[t.key] + rank] = make_uint2(r, t.payload);
                if (rank == 0) off[t.key] += uint32_t(__popcll(same));
            }
        }
    }
}

// ---- S5: counter chains, one wave per counter (NBLIC.c:589-637) ---------------------------
// Between two halvings a counter is a pure running sum, and a halving needs the sum to climb
// from <= 4129 past 8192 in steps <= 32, i.e. >= 127 touches.  Each iteration takes 256 touches
// (4 consecutive ones per lane, read as two 16-byte loads, next chunk prefetched), prefix-sums
// the weights on the DPP crossbar, and then resolves the (at most three) halvings that fall in
// the chunk one epoch at a time; every other touch is plain arithmetic on its prefix.
struct TouchQuad { uint32_t ev[4]; uint32_t pay[4]; };

__device__ __forceinline__ TouchQuad load_quad(const uint2 *__restrict__ touch, uint32_t first, uint32_t end) {
    TouchQuad q;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint2 t = (first + k < end) ? touch[first + k] : make_uint2(0u, 0u);
        q.ev[k] = t.x; q.pay[k] = t.y;
    }
    return q;
}

__global__ void __launch_bounds__(64) k_counter_chains(const uint2 *__restrict__ touch, const uint32_t *__restrict__ table,
                                                       SegPlan plan, const uint32_t *__restrict__ total,
                                                       int2 *__restrict__ cnt_state, uint16_t *__restrict__ puv) {
    const int key = int(blockIdx.x);
    const int lane = int(threadIdx.x);
    const uint32_t start = table[size_t(key) * plan.nseg];
    const uint32_t end = key + 1 < 4096 ? table[size_t(key + 1) * plan.nseg] : *total;
    if (start >= end) return;
    int2 st = cnt_state[key];
    int base_s = st.x + st.y, base_1 = st.y;                         // wave-uniform running state
    TouchQuad nxt = load_quad(touch, start + uint32_t(lane) * 4u, end);
    for (uint32_t chunk = start; chunk < end; chunk += 256u) {
        const TouchQuad q = nxt;
        const uint32_t first = chunk + uint32_t(lane) * 4u;
        if (chunk + 256u < end) nxt = load_quad(touch, first + 256u, end);
        int tot[4], one[4], tex[4], oex[4];
        int lt = 0, lo = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int w = int(q.pay[k] & 63) + int((q.pay[k] >> 6) & 63);
            tot[k] = (first + k < end) ? w : 0;
            one[k] = ((q.pay[k] >> 12) & 1) ? tot[k] : 0;
            tex[k] = lt; oex[k] = lo; lt += tot[k]; lo += one[k];
        }
        uint32_t incl = wave_scan_incl_dpp((uint32_t(lt) << 16) | uint32_t(lo));
        const int lane_t = int(incl >> 16) - lt, lane_o = int(incl & 0xFFFF) - lo;    // exclusive over lanes
        const int chunk_t = int(read_lane(incl, 63) >> 16), chunk_o = int(read_lane(incl, 63) & 0xFFFF);
#pragma unroll
        for (int k = 0; k < 4; k++) { tex[k] += lane_t; oex[k] += lane_o; }
        // virtual base: state before touch j of the current epoch = vb + exclusive prefix(j)
        int vb_s = base_s, vb_1 = base_1;
        int from = 0;                                               // first touch (chunk-relative) of the epoch
        int s_pre[4], c1_pre[4];
        for (;;) {
            int trig = 4;
#pragma unroll
            for (int k = 3; k >= 0; k--) {
                int j = lane * 4 + k;
                if (j >= from) {
                    s_pre[k] = vb_s + tex[k]; c1_pre[k] = vb_1 + oex[k];
                    if (s_pre[k] + tot[k] > kCountLimit) trig = k;
                }
            }
            uint64_t over = __ballot(trig < 4);
            if (over == 0ull) break;
            const int H = __ffsll((unsigned long long)over) - 1;    // first lane with a halving, uniform
            const int hk = int(read_lane(uint32_t(trig), H));
            int sel_s = s_pre[0], sel_1 = c1_pre[0], sel_tex = tex[0], sel_oex = oex[0], sel_tot = tot[0], sel_one = one[0];
            uint32_t sel_pay = q.pay[0];
#pragma unroll
            for (int k = 1; k < 4; k++)
                if (hk == k) { sel_s = s_pre[k]; sel_1 = c1_pre[k]; sel_tex = tex[k]; sel_oex = oex[k]; sel_tot = tot[k]; sel_one = one[k]; sel_pay = q.pay[k]; }
            const int hs = int(read_lane(uint32_t(sel_s), H)), h1 = int(read_lane(uint32_t(sel_1), H));
            const uint32_t hp = read_lane(sel_pay, H);
            const int h_tin = int(read_lane(uint32_t(sel_tex + sel_tot), H)), h_oin = int(read_lane(uint32_t(sel_oex + sel_one), H));
            Counter c{hs - h1, h1};
            const int hb = int((hp >> 12) & 1), hw1 = int(hp & 63), hw2 = int((hp >> 6) & 63);
            counter_add(c, hb, hw1);
            if (hw2) counter_add(c, hb, hw2);                       // state after the triggering touch
            vb_s = c.c0 + c.c1 - h_tin; vb_1 = c.c1 - h_oin;
            from = H * 4 + hk + 1;
        }
        base_s = vb_s + chunk_t; base_1 = vb_1 + chunk_o;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (first + k < end) {
                uint16_t p = uint16_t(counter_p1(s_pre[k] - c1_pre[k], c1_pre[k]));
                size_t o = size_t(q.ev[k]) * 2;
                if ((q.pay[k] >> 14) & 1) { puv[o] = p; puv[o + 1] = p; } else puv[o + ((q.pay[k] >> 13) & 1)] = p;
            }
        }
    }
    if (lane == 0) cnt_state[key] = make_int2(base_s - base_1, base_1);
}

// ---- mix the two trees' probabilities and pack for the host coder (NBLIC.c:629-633) -------
__global__ void __launch_bounds__(256) k_mix(const uint32_t *__restrict__ events, const uint16_t *__restrict__ puv, uint32_t n_ev,
                                             uint16_t *__restrict__ coded) {
    uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= n_ev) return;
    uint32_t e = events[r];
    int qw = ev_qw(e);
    int pu = puv[size_t(r) * 2], pv = qw ? int(puv[size_t(r) * 2 + 1]) : 0;
    coded[r] = pack_coded(mix_prob(pu, pv, qw), ev_bin(e));
}

// ---- model state init (NBLIC.c:797-804) ---------------------------------------------------
__global__ void k_init_state(int *ctx_state, int *map_state, int2 *cnt_state) {
    int g = int(blockIdx.x) * 256 + int(threadIdx.x);
    if (g < kContexts) ctx_state[g] = 0;
    if (g < 4096) cnt_state[g] = make_int2(kWeightOne, kWeightOne);
    if (g < 512)
        for (int k = 0; k < kMapSyms; k++) {
            map_state[g * 60 + k] = k; map_state[g * 60 + 20 + k] = k; map_state[g * 60 + 40 + k] = 2 * (kMapSyms - 1 - k);
        }
}

// ------------------------------------------------------------------------------------------
// exclusive scan (u32 out) over u32 or u8 input: reduce -> scan block sums -> apply
// ------------------------------------------------------------------------------------------
constexpr int kScanThreads = 256, kScanPerThread = 16, kScanTile = kScanThreads * kScanPerThread;

template <class T>
__global__ void __launch_bounds__(kScanThreads) k_scan_reduce(const T *__restrict__ in, uint32_t n, uint32_t *__restrict__ sums) {
    __shared__ uint32_t part[kScanThreads / 64];
    uint32_t base = blockIdx.x * uint32_t(kScanTile) + threadIdx.x * kScanPerThread, s = 0;
    for (int k = 0; k < kScanPerThread; k++) if (base + k < n) s += uint32_t(in[base + k]);
    s = wave_scan_incl(s);
    if (lane_id() == 63) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

__global__ void __launch_bounds__(1024) k_scan_sums(uint32_t *sums, uint32_t nblocks, uint32_t *total) {
    __shared__ uint32_t part[16];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nblocks; base += 1024) {
        uint32_t i = base + threadIdx.x;
        uint32_t v = i < nblocks ? sums[i] : 0u;
        uint32_t incl = wave_scan_incl(v);
        if (lane_id() == 63) part[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint32_t pre = carry;
        for (int wv = 0; wv < int(threadIdx.x >> 6); wv++) pre += part[wv];
        if (i < nblocks) sums[i] = pre + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = pre + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}

template <class T>
__global__ void __launch_bounds__(kScanThreads) k_scan_apply(const T *in, uint32_t n, const uint32_t *__restrict__ sums,
                                                             uint32_t *out) {   // in may alias out (each thread re-writes its own items)
    __shared__ uint32_t part[kScanThreads / 64];
    uint32_t base = blockIdx.x * uint32_t(kScanTile) + threadIdx.x * kScanPerThread;
    uint32_t v[kScanPerThread], s = 0;
    for (int k = 0; k < kScanPerThread; k++) { v[k] = base + k < n ? uint32_t(in[base + k]) : 0u; s += v[k]; }
    uint32_t incl = wave_scan_incl(s);
    if (lane_id() == 63) part[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t pre = sums[blockIdx.x] + incl - s;
    for (int wv = 0; wv < int(threadIdx.x >> 6); wv++) pre += part[wv];
    for (int k = 0; k < kScanPerThread; k++) { if (base + k < n) out[base + k] = pre; pre += v[k]; }
}

// self-test: the DPP scan must equal the shuffle scan on arbitrary data
__global__ void k_selftest_scan(const uint32_t *in, uint32_t *bad) {
    uint32_t v = in[blockIdx.x * 64 + threadIdx.x];
    if (wave_scan_incl(v) != wave_scan_incl_dpp(v)) atomicAdd(bad, 1u);
}

int e1_selftest(hipStream_t s) {
    uint32_t host[64 * 64], *d_in = nullptr, *d_bad = nullptr, bad = 1;
    uint32_t x = 12345u;
    for (auto &h : host) { x = x * 1664525u + 1013904223u; h = x >> 8; }
    if (hipMalloc((void **)&d_in, sizeof host) != hipSuccess || hipMalloc((void **)&d_bad, 4) != hipSuccess) return -1;
    hipMemcpyAsync(d_in, host, sizeof host, hipMemcpyHostToDevice, s);
    hipMemsetAsync(d_bad, 0, 4, s);
    hipLaunchKernelGGL(k_selftest_scan, dim3(64), dim3(64), 0, s, d_in, d_bad);
    hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, s);
    hipStreamSynchronize(s);
    hipFree(d_in); hipFree(d_bad);
    return int(bad);
}

// ------------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------------
static inline unsigned cdiv(size_t a, size_t b) { return unsigned((a + b - 1) / b); }

struct Marker {                       // records one event in front of every launch
    E1Timers *tm; hipStream_t s; int k;
    void operator()() { if (tm) hipEventRecord(tm->ev[k], s); k++; }
};

template <class T>
static void scan_exclusive(const T *in, uint32_t n, uint32_t *out, uint32_t *sums, uint32_t *total, hipStream_t s, Marker &mark) {
    unsigned nb = cdiv(n, kScanTile);
    if (nb == 0) { mark(); mark(); mark(); hipMemsetAsync(total, 0, 4, s); return; }
    mark(); hipLaunchKernelGGL(k_scan_reduce<T>, dim3(nb), dim3(kScanThreads), 0, s, in, n, sums);
    mark(); hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, s, sums, nb, total);
    mark(); hipLaunchKernelGGL(k_scan_apply<T>, dim3(nb), dim3(kScanThreads), 0, s, in, n, sums, out);
}

SegPlan make_plan(uint32_t n_items) {
    SegPlan p;
    uint32_t want = (n_items + kMaxSegments - 1) / kMaxSegments;          // items per segment
    if (want < 1024) want = 1024;
    p.seg_len = (want + 63u) & ~63u;
    p.nseg = int((n_items + p.seg_len - 1) / p.seg_len);
    if (p.nseg < 1) p.nseg = 1;
    return p;
}

void e1_init_state(const E1Buffers &b, hipStream_t s) {
    hipLaunchKernelGGL(k_init_state, dim3(16), dim3(256), 0, s, b.ctx_state, b.map_state, (int2 *)b.cnt_state);
}

// Stage group A: everything up to the per-pixel bin counts and their scan (the event total
// is needed on the host before the event buffers can be sized).  17 launches.
void e1_launch_front(const E1Buffers &b, int h, int w, hipStream_t s, E1Timers *tm) {
    const uint32_t n = uint32_t(size_t(h) * size_t(w));
    SegPlan pp = make_plan(n);
    Marker mark{tm, s, 0};
    mark(); hipLaunchKernelGGL(k_predict, dim3(cdiv(w, 256), h), dim3(256), 0, s, b.img, h, w, 0, h, b.rec1);
    mark(); hipLaunchKernelGGL(k_adr_count, dim3(cdiv(pp.nseg, 4)), dim3(256), 0, s, b.rec1, n, pp, b.table);
    scan_exclusive<uint32_t>(b.table, uint32_t(kContexts) * pp.nseg, b.table, b.scan_sums, b.totals + 0, s, mark);
    mark(); hipLaunchKernelGGL(k_adr_scatter, dim3(cdiv(pp.nseg, 4)), dim3(256), 0, s, b.rec1, b.img, n, pp, b.table, (uint2 *)b.s2rec);
    mark(); hipLaunchKernelGGL(k_bias_chains, dim3(kContexts / 64), dim3(64), 0, s, (const uint2 *)b.s2rec, b.table, pp, n, b.ctx_state, b.pxs);
    mark(); hipLaunchKernelGGL(k_map_count, dim3(cdiv(pp.nseg, 4)), dim3(256), 0, s, b.img, b.pxs, n, pp, b.table);
    scan_exclusive<uint32_t>(b.table, 512u * pp.nseg, b.table, b.scan_sums, b.totals + 1, s, mark);
    mark(); hipLaunchKernelGGL(k_map_scatter, dim3(cdiv(pp.nseg, 4)), dim3(256), 0, s, b.img, b.pxs, n, pp, b.table, b.s3rec, b.z);
    mark(); hipLaunchKernelGGL(k_mapper_chains, dim3(512 / 64), dim3(64), 0, s, b.s3rec, b.table, pp, 0u, b.totals + 1, b.map_state, b.z);
    mark(); hipLaunchKernelGGL(k_count_bins, dim3(cdiv(n, 256)), dim3(256), 0, s, b.rec1, b.z, n, b.cnt);
    scan_exclusive<uint8_t>(b.cnt, n, b.ev_off, b.scan_sums, b.totals + 2, s, mark);
    mark();                                                     // start of the host gap (index 17)
}

// Stage group B: needs n_ev (read back from totals[2]) and event-sized buffers.  8 launches.
void e1_launch_back(const E1Buffers &b, int h, int w, uint32_t n_ev, hipStream_t s, E1Timers *tm) {
    const uint32_t n = uint32_t(size_t(h) * size_t(w));
    SegPlan pe = make_plan(n_ev);
    Marker mark{tm, s, 18};
    mark(); hipLaunchKernelGGL(k_emit_bins, dim3(cdiv(n, 256)), dim3(256), 0, s, b.rec1, b.z, n, b.ev_off, b.events);
    mark(); hipLaunchKernelGGL(k_touch_count, dim3(cdiv(pe.nseg, 4)), dim3(256), 0, s, b.events, n_ev, pe, b.table);
    scan_exclusive<uint32_t>(b.table, 4096u * pe.nseg, b.table, b.scan_sums, b.totals + 3, s, mark);
    mark(); hipLaunchKernelGGL(k_touch_scatter, dim3(cdiv(pe.nseg, 4)), dim3(256), 0, s, b.events, n_ev, pe, b.table, (uint2 *)b.touch);
    mark(); hipLaunchKernelGGL(k_counter_chains, dim3(4096), dim3(64), 0, s, (const uint2 *)b.touch, b.table, pe, b.totals + 3,
                               (int2 *)b.cnt_state, b.puv);
    mark(); hipLaunchKernelGGL(k_mix, dim3(cdiv(n_ev, 256)), dim3(256), 0, s, b.events, b.puv, n_ev, b.coded);
    mark();                                                     // index 26: end
}

}  // namespace nblic

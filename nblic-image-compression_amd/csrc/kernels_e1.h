// kernels_e1.h -- launch interface of the staged -e1 lossless kernels (kernels_e1.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nblic {

constexpr uint32_t kMaxSegments = 1024;   // waves per partition pass (pixel partitions)
// The touch partition keeps 4096 output runs open per wave; with 1024 waves per image that is
// 4 M partially written lines (512 MB) -- more than the 256 MB Infinity Cache, and rocprof showed
// 9x write amplification.  256 waves per image keep the open lines of a launch resident.
constexpr uint32_t kTouchSegments = 256;
constexpr int kTotalsStride = 8, kWideTouchFlag = 4;      // words per image in the totals array; index of the wide-position flag

struct SegPlan { int nseg; uint32_t seg_len; };
SegPlan make_plan(uint32_t n_items, uint32_t max_segments = kMaxSegments);

// Device buffers of one image in flight.  Pixel-sized arrays hold n = h*w entries,
// event-sized arrays hold ev_cap entries.
constexpr size_t kStreamPad = 512;         // u16 records of slack after every key-sorted stream

struct E1Buffers {
    const uint8_t *img;      // n      input plane (lossless: also the reconstruction)
    uint32_t *rec1;          // n      S1 record (model.h pack_s1)
    uint16_t *s2in;          // n+pad  px0 | err<<8, grouped by context
    uint32_t *pos2;          // n      where pixel t sits in s2in / s2out
    uint16_t *s2out;         // n+pad  px | sign<<8, same order as s2in
    uint16_t *pxs;           // n      px | sign<<8 back in raster order
    uint16_t *s3in;          // n+pad  y (< 20), grouped by re-mapper
    uint32_t *pos3;          // n      where pixel t sits in s3in / s3out, or 0x80000000 | y (bypass)
    uint16_t *s3out;         // n+pad  z, same order as s3in
    uint8_t  *z;             // n      coded symbol, raster order
    uint8_t  *cnt;           // n      bins per pixel
    uint32_t *ev_off;        // n      exclusive scan of cnt
    uint32_t *table;         // 4096 * kMaxSegments   partition histogram / offsets
    uint32_t *scan_sums;     // scan scratch
    uint32_t *totals;        // [kTotalsStride] item totals: adr, mapper, events, touches; [kWideTouchFlag] 1 = this image needs 32-bit touch positions
    int      *ctx_state;     // 2048
    int      *map_state;     // 512 * 60
    int      *cnt_state;     // 4096 * 2
    uint32_t *events;        // ev_cap        model.h pack_event
    uint16_t *tin;           // 2*ev_cap+pad  touch payloads grouped by counter
    uint64_t *tpos;          // ev_cap + 64   two u32 arrays (even trees, odd trees): position of the event's touch in tin/tout (28 bits, all ones = none) with qw / bin / parity in the top four bits (kernels_e1.hip pack_pos); plain 32-bit positions, ~0 = none, for an image with >= 2^28 - 1 touches
    uint16_t *tout;          // 2*ev_cap+pad  P(bin==1) before each touch, same order as tin
    uint32_t *blk_base;      // 2049          first block of every context chain (+ total)
    int      *blk_end;       // n/4096+2048   context state at the end of each block
    uint8_t  *blk_ok;        // n/4096+2048   1 = the block's warm-up copies met (its output is exact)
    uint32_t *qhist;         // 12 * 256      QNBLIC symbol histograms per activity level
    unsigned long long *dbg_out;   // 4096 words of in-kernel cycle stamps, written only when E1Job::dbg & 8
    uint32_t *win_base;      // 4097 + 4096   first window record of every counter chain (+ total); chain keys, longest first
    uint32_t *win_recs;      // 24 words per 512-touch window: entry state + halving epochs (kernels_e1.hip WinRec)
    uint16_t *coded;         // ev_cap        prob | bin<<15 for the host range coder
};

// One image of a group.  The array of jobs lives in device memory; every kernel picks its job
// from the last grid dimension.
struct E1Job {
    E1Buffers b;
    int h, w;
    uint32_t n;          // h * w
    SegPlan pp;          // partition plan over pixels
    uint32_t n_ev;       // bins (known after the front half)
    SegPlan pe;          // partition plan over bins
    int dbg;             // timing experiments only (NBLIC_AMD_DBG); 0 in normal operation
    int near, k_step;    // serial modes only (the staged -e1 kernels use the lossless constants 0 and 3)
    uint64_t ktab;       // model.h level_shift_table(k_step)
};

// One HIP event before every kernel launch (and one after the last): interval k is exactly
// launch k of the sequence below, measured on the stream it runs on.  A launch covers the
// whole group of images.
constexpr int kE1Kernels = 31;
constexpr int kE1Marks = kE1Kernels + 1;
struct E1Timers { hipEvent_t ev[kE1Marks]; uint64_t mask = ~0ull; };     // mask: bit k = stage k is timed (events k and k + 1 are recorded)
constexpr uint64_t kRooflineStages = (1ull << 1) | (1ull << 26);            // k_predict (S1) and k_touch_scatter: what the bench line's rooflines need
static const char *const kE1StageNames[kE1Kernels] = {
    "k_init_state", "k_predict",
    "k_adr_count", "scan_reduce.adr", "scan_sums.adr", "scan_apply.adr", "k_adr_scatter",
    "k_plan_blocks", "k_bias_blocks", "k_bias_fixup",
    "k_map_count", "scan_reduce.map", "scan_sums.map", "scan_apply.map", "k_map_scatter",
    "k_mapper_chains",
    "k_count_bins", "scan_reduce.bins", "scan_sums.bins", "scan_apply.bins",
    "host_gap",
    "k_emit_bins",
    "k_touch_count", "scan_reduce.touch", "scan_sums.touch", "scan_apply.touch", "k_touch_scatter",
    "k_plan_windows", "k_counter_epochs", "k_counter_probs",
    "k_mix"};

int e1_selftest(hipStream_t s);     // 0 = DPP wave scan agrees with the shuffle scan
// d_jobs: device copy of h_jobs[0..n_jobs).  The host copy is only read to size the grids.
void e1_launch_front(const E1Job *d_jobs, const E1Job *h_jobs, int n_jobs, hipStream_t s, E1Timers *tm);
void e1_launch_back(const E1Job *d_jobs, const E1Job *h_jobs, int n_jobs, hipStream_t s, E1Timers *tm, bool general = false);
// serial modes (near > 0, efforts 2/3): model state init, then -- after the caller has run the serial
// model stage (serial_engine.h) that leaves rec1 and px | sign per pixel -- the re-mapper partition,
// the re-mapper chains and the bin counts; e1_launch_back(..., general = true) finishes the job
void e1_launch_init(const E1Job *d_jobs, int n_jobs, hipStream_t s);
void e1_launch_front_pre(const E1Job *d_jobs, const E1Job *h_jobs, int n_jobs, hipStream_t s);
// QNBLIC (effort 0) model stage for a group: leaves level | symbol << 8 per pixel in `pxs` and the
// 12 x 256 histograms in `qhist`; the entropy stage (normalise, histogram code, rANS) is host work.
void q_launch_model(const E1Job *d_jobs, const E1Job *h_jobs, int n_jobs, hipStream_t s);

}  // namespace nblic

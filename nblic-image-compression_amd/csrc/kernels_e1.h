// kernels_e1.h -- launch interface of the staged -e1 lossless kernels (kernels_e1.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nblic {

constexpr uint32_t kMaxSegments = 1024;   // waves per partition pass

struct SegPlan { int nseg; uint32_t seg_len; };
SegPlan make_plan(uint32_t n_items);

// Device buffers of one image in flight.  Pixel-sized arrays hold n = h*w entries,
// event-sized arrays hold ev_cap entries.
struct E1Buffers {
    const uint8_t *img;      // n      input plane (lossless: also the reconstruction)
    uint32_t *rec1;          // n      S1 record (model.h pack_s1)
    uint64_t *s2rec;         // n      {t, px0|err<<8} grouped by context
    uint16_t *pxs;           // n      px | sign<<8
    uint32_t *s3rec;         // n      t | y<<27 grouped by re-mapper
    uint8_t  *z;             // n      coded symbol
    uint8_t  *cnt;           // n      bins per pixel
    uint32_t *ev_off;        // n      exclusive scan of cnt
    uint32_t *table;         // 4096 * kMaxSegments   partition histogram / offsets
    uint32_t *scan_sums;     // scan scratch
    uint32_t *totals;        // [4] item totals: adr, mapper, events, touches
    int      *ctx_state;     // 2048
    int      *map_state;     // 512 * 60
    int      *cnt_state;     // 4096 * 2
    uint32_t *events;        // ev_cap
    uint64_t *touch;         // 2 * ev_cap
    uint16_t *puv;           // 2 * ev_cap
    uint16_t *coded;         // ev_cap   prob | bin<<15 for the host range coder
};

// One HIP event before every kernel launch (and one after the last): interval k is exactly
// kernel k of the launch sequence below, measured on the stream it runs on.
constexpr int kE1Kernels = 26;
constexpr int kE1Marks = kE1Kernels + 1;
struct E1Timers { hipEvent_t ev[kE1Marks]; };
static const char *const kE1StageNames[kE1Kernels] = {
    "k_predict",
    "k_adr_count", "scan_reduce.adr", "scan_sums.adr", "scan_apply.adr", "k_adr_scatter",
    "k_bias_chains",
    "k_map_count", "scan_reduce.map", "scan_sums.map", "scan_apply.map", "k_map_scatter",
    "k_mapper_chains",
    "k_count_bins", "scan_reduce.bins", "scan_sums.bins", "scan_apply.bins",
    "host_gap",
    "k_emit_bins",
    "k_touch_count", "scan_reduce.touch", "scan_sums.touch", "scan_apply.touch", "k_touch_scatter",
    "k_counter_chains",
    "k_mix"};

int e1_selftest(hipStream_t s);     // 0 = DPP wave scan agrees with the shuffle scan
void e1_init_state(const E1Buffers &b, hipStream_t s);
void e1_launch_front(const E1Buffers &b, int h, int w, hipStream_t s, E1Timers *tm);
void e1_launch_back(const E1Buffers &b, int h, int w, uint32_t n_ev, hipStream_t s, E1Timers *tm);

}  // namespace nblic

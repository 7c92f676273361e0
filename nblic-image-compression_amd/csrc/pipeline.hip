// pipeline.hip -- host side of libnblic_amd.so: workspaces, streams, the serial range-coder
// stage (S6) on host threads, and the C ABI declared in include/nblic_amd.h.
//
// Images in flight are split into GROUPS that share every kernel launch (a driver thread and a HIP
// stream per group).  A finished image's coded bins (u16 per bin) stay in an HBM buffer of a pool
// until a coder thread streams them to the host chunk by chunk (packed to 13 bits per bin and laid
// out for its AVX-512 lanes by k_pack_groups, up to 24 images at a time) through its own pinned ring
// and turns them into the byte-exact range-coder stream (NBLIC.c:552-586), while the GPU is already
// working on the next groups.  The rank's CPU share is what bounds the pipeline, so nothing here spins:
// drivers sleep on a condition variable, coder threads poll for a chunk with 100 us sleeps.  Three kinds of group: staged -n0 -e1 encode, QNBLIC (effort 0) encode,
// and the serial modes (near > 0, efforts 2/3), whose front half is the one-wave-per-image model
// stage of serial_engine.hip and whose entropy stages are the same parallel kernels.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include <pthread.h>
#include <sched.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include "../../include/nblic_amd.h"
#include "device_coder.h"
#include "kernels_e1.h"
#include "model.h"
#include "range_coder.h"
#include "serial_engine.h"
#include "sha256.h"

struct nblic_amd_ctx;

namespace nblic {

#define HIP_OK(call)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            fprintf(stderr, "[nblic_amd] %s failed: %s (%s:%d)\n", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return false;                                                                     \
        }                                                                                     \
    } while (0)

// ---- S6: 32-bit carry-less binary range coder (NBLIC.c:527-586), encoder side ------------
// (range_coder.h: resumable, because the coder threads stream the bins from HBM in chunks)
void RangeScalar::feed(const uint16_t *coded, size_t n) {
    if (overflow) return;
    uint32_t l = lo, h = hi;
    uint8_t *q = p;
    for (size_t r = 0; r < n; r++) {
        uint32_t e = coded[r];
        uint32_t cut = l + uint32_t((uint64_t(h - l) * (e & 0xFFFu)) >> 12);
        bool one = (e >> 15) != 0;
        h = one ? cut : h;
        l = one ? l : cut + 1;
        while (((l ^ h) >> 24) == 0) {
            if (q == end) { overflow = true; return; }
            *q++ = uint8_t(h >> 24);
            l <<= 8;
            h = (h << 8) | 0xFFu;
        }
    }
    lo = l; hi = h; p = q;
}

size_t RangeScalar::finish() {
    if (overflow) return SIZE_MAX;
    uint32_t l = lo;
    for (int k = 0; k < 4; k++) { *p++ = uint8_t(l >> 24); l <<= 8; }
    return size_t(p - out);
}

// Returns the number of bytes written, or SIZE_MAX if `cap` bytes were not enough.
size_t range_code(const uint16_t *coded, size_t n, uint8_t *out, size_t cap) {
    if (cap < 4) return SIZE_MAX;
    RangeScalar r;
    r.begin(out, cap);
    r.feed(coded, n);
    return r.finish();
}

long q_entropy_encode(uint16_t *out, size_t cap_words, int h, int w, const uint16_t *qy, const uint32_t *hist_in);

void write_header(uint8_t *p, int h, int w, int near, int k_step, int effort) {   // NBLIC.c:682-694
    memcpy(p, "NBLIC0.3", 8);
    p[8] = 1;
    p[9] = uint8_t(h >> 8); p[10] = uint8_t(h);
    p[11] = uint8_t(w >> 8); p[12] = uint8_t(w);
    p[13] = uint8_t(near); p[14] = uint8_t(k_step); p[15] = uint8_t(effort);
}

bool size_ok(int h, int w, long max_px) {                                          // NBLIC.c:717-729
    return h > 0 && w > 0 && h <= NBLIC_MAX_HEIGHT && w <= NBLIC_MAX_WIDTH && long(h) * long(w) <= max_px;
}

// ---- an image between the GPU and the coder threads ----------------------------------------
// The backlog lives in HBM: a finished image's coded bins stay in a device buffer until a coder
// thread streams them to the host chunk by chunk (its own small pinned ring), so the pinned host
// memory is per THREAD, not per image, and the GPU never waits for host buffers.
// Host memory the coder threads READ at full speed: ordinary pages, first touched by the thread
// that will read them (so they sit on its NUMA node), then page-locked in place.  hipHostMalloc'ed
// memory reads 14-18 % slower from these threads (measured: 1650 vs 1950 Mbins/s through the
// sixteen-lane coder, tools/pinned_coder_bench.py).
static std::mutex g_locked_m;
static std::vector<void *> g_from_runtime;                          // the few buffers that had to come from hipHostMalloc instead

static uint16_t *locked_alloc(size_t words) {
    const size_t bytes = (words * sizeof(uint16_t) + (size_t(2) << 20) - 1) & ~((size_t(2) << 20) - 1);
    if (!getenv("NBLIC_AMD_HOSTMALLOC")) {
        if (void *p = aligned_alloc(size_t(2) << 20, bytes)) {
            madvise(p, bytes, MADV_HUGEPAGE);
            memset(p, 0, bytes);
            if (hipHostRegister(p, bytes, hipHostRegisterDefault) == hipSuccess) return static_cast<uint16_t *>(p);
            free(p);
        }
    }
    void *q = nullptr;                                              // registration refused (or disabled): the runtime's own pinned memory
    if (hipHostMalloc(&q, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    { std::lock_guard<std::mutex> l(g_locked_m); g_from_runtime.push_back(q); }
    return static_cast<uint16_t *>(q);
}
static void locked_free(uint16_t *p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> l(g_locked_m);
        auto it = std::find(g_from_runtime.begin(), g_from_runtime.end(), static_cast<void *>(p));
        if (it != g_from_runtime.end()) { g_from_runtime.erase(it); hipHostFree(p); return; }
    }
    hipHostUnregister(p);
    free(p);
}

struct CodedBuf { uint16_t *p = nullptr; size_t cap = 0; };        // device; one image's coded bins (QNBLIC: pairs + histograms)
// Bins per lane per chunk of the host ring.  Every chunk costs the GPU two dispatches per thread (the
// interleave kernel and the copy, which the runtime performs with a blit kernel) that have to find
// room between the encoder's own kernels: with 1 Mbin chunks the coder threads waited for their
// next chunk 15-20 % of the time (4.7 Gpx/s), with 4 Mbin chunks 3 % (5.2 Gpx/s).
constexpr size_t kChunkBins = size_t(1) << 22;
constexpr int kCopyStreams = 8;
constexpr int kRingDepth = 3;                                      // ring slots per coder thread: the chunk being coded + the next two on their way (two slots: 6.14-6.30 Gpx/s, three: 6.35-6.38)
constexpr int kMaxTake = 24;                                       // images one coder thread codes together (three AVX-512 packs; NBLIC_AMD_MAX_TAKE=16: two)

}  // namespace nblic
// One submitted batch (nblic_amd_encode_batch_begin .. _end); `remaining` is guarded by ctx->fm.
struct nblic_amd_batch { int remaining = 0; int n_images = 0; long *lens = nullptr; bool ok = true; bool submitted = false; };   // submitted: every image has been handed to a group (guarded by ctx->fm)
namespace nblic {

struct ReadyImage {                                                  // everything a coder thread needs
    int cb, job, h, w;
    uint32_t n_ev;
    unsigned char *const *outs; const size_t *caps; long *lens;      // -e1: byte streams; effort 0: uint16_t streams, caps/lens in words
    int kind;                                                        // 0 / 2 = NBLIC range coder, 1 = QNBLIC entropy stage
    ::nblic_amd_batch *batch;                                        // whose completion this image counts towards
    int near, k_step, effort;                                        // header fields (NBLIC.c:682-694)
};

// ---- one image in flight -------------------------------------------------------------------
struct Slot {
    E1Buffers b{};
    size_t px_cap = 0, ev_cap = 0, img_cap = 0;
    uint8_t *d_img = nullptr;         // device copy when the caller hands a host image
    int cb = -1;                      // coded-bin buffer (HBM) this image's back half writes to
    int job = -1, h = 0, w = 0;       // current image
    int near = 0, effort = 1;         // its mode (kind 2 groups; 0 / 1 otherwise)
    uint32_t n_ev = 0;
    // serial modes: reconstruction (near > 0) and least-squares statistics (efforts 2/3)
    uint8_t *d_recon = nullptr; size_t recon_cap = 0;
    double *d_stats = nullptr; size_t stats_cap = 0;
    SerialState *d_state = nullptr;   // what the model stage carries from launch to launch (serial_engine.h)
};

// ---- a group of images that shares every kernel launch ---------------------------------------
// What a driver thread sleeps on while its group's front half runs (a host function queued behind the front half
// sets `ready`).  hipEventSynchronize is NOT a sleep here, whatever the event's flags say: measured, a driver burnt
// 35 ms of CPU per 55 ms wait, 1.1 of the rank's 16 CPUs between the six of them -- quota the coder threads need.
struct GroupWait { std::mutex m; std::condition_variable cv; bool ready = false; };

struct Group {
    int id = 0;
    std::unique_ptr<GroupWait> front = std::make_unique<GroupWait>();
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    E1Timers tm{};
    std::vector<Slot> slots;
    E1Job *h_jobs = nullptr, *d_jobs = nullptr;        // pinned host / device job records
    SerialJob *h_sjobs = nullptr, *d_sjobs = nullptr;  // the same images for the serial model stage (kind 2)
    unsigned char *const *recons = nullptr;            // kind 2: where each image's reconstruction goes (host; entries may be null)
    uint32_t *h_totals = nullptr, *d_totals = nullptr; // kTotalsStride words per slot
    int n_jobs = 0;
    bool tm_pending = false;                           // timer events recorded, not yet read
    ::nblic_amd_ctx *ctx = nullptr;
    // the batch this group currently serves (valid from launch_back until its coders finish)
    unsigned char *const *outs = nullptr; const size_t *caps = nullptr; long *lens = nullptr;
    int kind = 0;
    ::nblic_amd_batch *batch = nullptr;
    // hand-over to the group's driver thread (guarded by ctx->dm)
    const uint8_t *const *imgs = nullptr; bool on_device = false; bool has_work = false;
};

template <class T> static bool dev_alloc(T *&p, size_t count) {
    if (p) hipFree(p);
    p = nullptr;
    HIP_OK(hipMalloc((void **)&p, count * sizeof(T)));
    return true;
}

}  // namespace nblic

using namespace nblic;

struct nblic_amd_ctx {
    int device = 0;
    long max_px = kMaxPixels;
    bool timing = false;
    uint64_t timing_mask = ~0ull;            // stages to time (kernels_e1.h E1Timers::mask)
    bool simd = false;                    // AVX-512 host: up to sixteen images per coder thread (two packs in lock-step)
    std::vector<Group> groups;
    std::mutex api;                       // one batch at a time per context
    std::mutex fm;                        // free groups / free coded-bin buffers / outstanding work
    std::condition_variable fcv;
    std::deque<int> free_groups;
    bool trace = false;                      // NBLIC_AMD_DBG & 64: timeline of groups and coder takes on stderr
    std::chrono::steady_clock::time_point t_batch;
    double now() const { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_batch).count(); }
    int coders_wanted = 0;                   // coder threads of this context (set before they start: pinning needs it)
    int max_take = kMaxTake;                 // images a coder thread takes together: 24 = three AVX-512 packs (NBLIC_AMD_MAX_TAKE=16: two, for A/B runs)
    std::vector<hipStream_t> copy_streams;   // shared by the coder threads (device -> host chunk copies)
    size_t chunk_bins = kChunkBins;          // bins per lane per chunk (NBLIC_AMD_CHUNK_BINS shrinks it, for tests of the chunk boundaries)
    std::vector<CodedBuf> cbufs;
    std::deque<int> free_cbufs;
    int coding = 0;                       // images handed to the GPU whose streams are not finished yet
    // coder threads
    std::vector<std::thread> coders;
    std::mutex rm;
    std::condition_variable rcv;
    std::deque<ReadyImage> ready;
    int idle_coders = 0;
    int batch_to_come = 0;                // images of the running batch that have not reached `ready` yet (guarded by rm)
    bool stop = false;
    // One driver thread per group: a group's launch sequence has a host round trip in the middle
    // (the event count sizes the back half) and may wait for a coded-bin buffer; with a thread each,
    // one group waiting never keeps the others from being launched.
    std::vector<std::thread> drivers;
    std::mutex dm;
    std::condition_variable dcv;
    bool stop_drivers = false;
    std::atomic<bool> broken{false};         // a thread of the context could not set itself up (sticky); a failure of one image is recorded in ITS batch
    // reporting
    double stage_ms[kE1Kernels] = {0};
    long stage_launches = 0;
    double total_bins = 0, coder_s = 0;
    double pack_bins = 0, pack_s = 0;     // the part of the above coded in packs (2..16 images per thread)
    double wait_s = 0, issue_s = 0;       // of coder_s: waiting for bins to arrive from HBM / queueing the next chunk
    double driver_cpu_s = 0, driver_front_cpu_s = 0, driver_wait_cpu_s = 0; long driver_launches = 0;   // CPU time of the driver threads (reporting)
    long takes[kMaxTake + 1] = {0};       // how many times a thread took k images together
    std::mutex stat_m;
    // Submission is asynchronous: _begin only queues the batch; the submitter thread hands its images to the groups
    // (which blocks while every group is busy), so a caller can keep several batches ahead of the pipeline.
    struct SubmitItem {
        ::nblic_amd_batch *b; int n; const uint8_t *const *imgs; bool on_device; const int *hs, *ws;
        uint8_t *const *outs; const size_t *caps; long *lens; const int *nears, *efforts; unsigned char *const *recons;
    };
    std::thread submitter;
    std::mutex sm;
    std::condition_variable scv;
    std::deque<SubmitItem> sq;
    bool stop_submit = false;
    int queued_images = 0;                // images of batches still waiting in sq (guarded by rm, like batch_to_come)
    // device coder (device_coder.hip): pack threads that hand 64 queued images at a time to one wave each
    std::vector<std::thread> dev_coders;
    int dev_min_outstanding = 0;          // a pack is taken only while at least this many images of the submitted batches are unfinished
    double dev_bins = 0; long dev_packs = 0, dev_images = 0;
    // decode batches (nblic_amd_decode_batch): a stream of their own and grow-only device / pinned arenas
    hipStream_t dec_stream = nullptr, dec_stream2 = nullptr;           // decode_batch alternates its chunks between the two
    uint8_t *dec_arena = nullptr; size_t dec_arena_cap = 0;
    SerialJob *dec_jobs = nullptr; int dec_jobs_cap = 0;
    int serial_rows = 0;                  // rows per launch of the serial kernels; 0 = sized for a few seconds per launch (nblic_amd_set_serial_rows)
    long serial_launch_count = 0;         // launches of the serial model / decode kernels since the context was created (reporting, tests)
    size_t feed_chunk = size_t(1) << 20;  // bytes per step in which the drop-in decoders fetch a stream of unknown length (nblic_amd_set_feed_chunk)
    long fed_bytes = 0;                   // bytes the last drop-in decode read from the caller's stream
    int feed_pipe[2] = {-1, -1};          // safe_copy: the kernel does the reading
};

namespace nblic {

static bool group_init(Group &g, int id, int n_slots, nblic_amd_ctx *c) {
    g.id = id; g.ctx = c;
    g.slots.resize(size_t(n_slots));
    HIP_OK(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
    HIP_OK(hipEventCreateWithFlags(&g.done, hipEventDisableTiming | hipEventBlockingSync));
    for (auto &e : g.tm.ev) HIP_OK(hipEventCreate(&e));
    HIP_OK(hipHostMalloc((void **)&g.h_jobs, size_t(n_slots) * sizeof(E1Job), hipHostMallocDefault));
    HIP_OK(hipMalloc((void **)&g.d_jobs, size_t(n_slots) * sizeof(E1Job)));
    HIP_OK(hipHostMalloc((void **)&g.h_sjobs, size_t(n_slots) * sizeof(SerialJob), hipHostMallocDefault));
    HIP_OK(hipMalloc((void **)&g.d_sjobs, size_t(n_slots) * sizeof(SerialJob)));
    HIP_OK(hipHostMalloc((void **)&g.h_totals, size_t(n_slots) * kTotalsStride * sizeof(uint32_t), hipHostMallocDefault));
    HIP_OK(hipMalloc((void **)&g.d_totals, size_t(n_slots) * kTotalsStride * sizeof(uint32_t)));
    for (int k = 0; k < n_slots; k++) {
        Slot &s = g.slots[size_t(k)];
        HIP_OK(hipMalloc((void **)&s.b.table, size_t(4096) * kMaxSegments * sizeof(uint32_t)));
        HIP_OK(hipMalloc((void **)&s.b.scan_sums, size_t(1) << 20));
        s.b.totals = g.d_totals + size_t(k) * kTotalsStride;
        HIP_OK(hipMalloc((void **)&s.b.ctx_state, 4096 * sizeof(int)));           // 2048 (NBLIC) or 3072 (QNBLIC) contexts
        HIP_OK(hipMalloc((void **)&s.b.qhist, 12 * 256 * sizeof(uint32_t)));
        HIP_OK(hipMalloc((void **)&s.b.map_state, 512 * 60 * sizeof(int)));
        HIP_OK(hipMalloc((void **)&s.b.cnt_state, 4096 * 2 * sizeof(int)));
        HIP_OK(hipMalloc((void **)&s.b.win_base, (4097 + 4096) * sizeof(uint32_t)));
        HIP_OK(hipMalloc((void **)&s.b.blk_base, 4097 * sizeof(uint32_t)));
        HIP_OK(hipMalloc((void **)&s.b.dbg_out, 4096 * sizeof(unsigned long long)));
        HIP_OK(hipMemset(s.b.dbg_out, 0, 4096 * sizeof(unsigned long long)));
        HIP_OK(hipMalloc((void **)&s.d_state, kModelStateBytes));
    }
    return true;
}

static void group_free(Group &g) {
    for (auto &s : g.slots) {
        hipFree(s.b.rec1); hipFree(s.b.s2in); hipFree(s.b.pos2); hipFree(s.b.s2out); hipFree(s.b.pxs); hipFree(s.b.s3in);
        hipFree(s.b.pos3); hipFree(s.b.s3out); hipFree(s.b.z); hipFree(s.b.cnt); hipFree(s.b.ev_off); hipFree(s.b.table);
        hipFree(s.b.scan_sums); hipFree(s.b.ctx_state); hipFree(s.b.map_state); hipFree(s.b.cnt_state); hipFree(s.b.events);
        hipFree(s.b.tin); hipFree(s.b.tpos); hipFree(s.b.tout); hipFree(s.b.win_base); hipFree(s.b.win_recs); hipFree(s.b.blk_base); hipFree(s.b.dbg_out); hipFree(s.b.qhist); hipFree(s.b.blk_end); hipFree(s.b.blk_ok); hipFree(s.d_img);
        hipFree(s.d_recon); hipFree(s.d_stats); hipFree(s.d_state);
    }
    hipFree(g.d_jobs); hipFree(g.d_totals); hipFree(g.d_sjobs);
    if (g.h_sjobs) hipHostFree(g.h_sjobs);
    if (g.h_jobs) hipHostFree(g.h_jobs);
    if (g.h_totals) hipHostFree(g.h_totals);
    for (auto &e : g.tm.ev) if (e) hipEventDestroy(e);
    if (g.done) hipEventDestroy(g.done);
    if (g.stream) hipStreamDestroy(g.stream);
}

static bool ensure_events(Slot &s, size_t n_ev) {
    if (n_ev <= s.ev_cap) return true;
    size_t cap = n_ev + n_ev / 8 + 1024;
    if (!dev_alloc(s.b.events, cap) || !dev_alloc(s.b.tin, 2 * cap + kStreamPad) || !dev_alloc(s.b.tpos, cap + 64) ||
        !dev_alloc(s.b.tout, 2 * cap + kStreamPad) ||
        !dev_alloc(s.b.win_recs, (2 * cap / 512 + 4096 + 8) * 24)) return false;
    s.ev_cap = cap;
    return true;
}

static bool ensure_pixels(Slot &s, size_t n, bool with_events = true) {
    if (n > s.px_cap) {
        size_t cap = n;
        if (!dev_alloc(s.b.rec1, cap) || !dev_alloc(s.b.s2in, cap + kStreamPad) || !dev_alloc(s.b.pos2, cap) ||
            !dev_alloc(s.b.s2out, cap + kStreamPad) || !dev_alloc(s.b.pxs, cap) || !dev_alloc(s.b.s3in, cap + kStreamPad) ||
            !dev_alloc(s.b.pos3, cap) || !dev_alloc(s.b.s3out, cap + kStreamPad) || !dev_alloc(s.b.z, cap) ||
            !dev_alloc(s.b.cnt, cap) || !dev_alloc(s.b.ev_off, cap) || !dev_alloc(s.b.blk_end, cap / 4096 + 4096 + 64) ||
            !dev_alloc(s.b.blk_ok, cap / 4096 + 4096 + 64)) return false;
        s.px_cap = cap;
    }
    return with_events ? ensure_events(s, 5 * n) : true;   // typical images need 4.3-4.5 bins/px (ensure_events adds 1/8); grown on demand
}

// Front half for the images assigned to group g (slots 0..n_jobs-1 already carry job/h/w).
static bool launch_front(nblic_amd_ctx *c, Group &g, const uint8_t *const *imgs, bool on_device) {
    for (int k = 0; k < g.n_jobs; k++) {
        Slot &s = g.slots[size_t(k)];
        size_t n = size_t(s.h) * size_t(s.w);
        if (!ensure_pixels(s, n)) return false;
        if (on_device) {
            s.b.img = imgs[s.job];
        } else {
            if (n > s.img_cap) { if (!dev_alloc(s.d_img, n)) return false; s.img_cap = n; }
            HIP_OK(hipMemcpyAsync(s.d_img, imgs[s.job], n, hipMemcpyHostToDevice, g.stream));
            s.b.img = s.d_img;
        }
        E1Job &J = g.h_jobs[k];
        J.b = s.b; J.h = s.h; J.w = s.w; J.n = uint32_t(n); J.pp = make_plan(J.n); J.n_ev = 0; J.pe = make_plan(0, kTouchSegments);
        { static const int dbg = getenv("NBLIC_AMD_DBG") ? atoi(getenv("NBLIC_AMD_DBG")) : 0; J.dbg = dbg; }
    }
    HIP_OK(hipMemcpyAsync(g.d_jobs, g.h_jobs, size_t(g.n_jobs) * sizeof(E1Job), hipMemcpyHostToDevice, g.stream));
    g.tm.mask = c->timing_mask;
    e1_launch_front(g.d_jobs, g.h_jobs, g.n_jobs, g.stream, c->timing ? &g.tm : nullptr);
    HIP_OK(hipMemcpyAsync(g.h_totals, g.d_totals, size_t(g.n_jobs) * kTotalsStride * sizeof(uint32_t), hipMemcpyDeviceToHost, g.stream));
    return true;
}

// Front half of a serial-mode group (near > 0 and / or efforts 2, 3): model state init, the serial
// model stage (one wave per image, all images of the group side by side; the slots are ordered by
// effort, one launch per effort present), the reconstructions on their way back to the host, then
// the re-mapper partition and chains and the bin counts.
static bool launch_front_serial(nblic_amd_ctx *c, Group &g, const uint8_t *const *imgs, bool on_device) {
    for (int k = 0; k < g.n_jobs; k++) {
        Slot &s = g.slots[size_t(k)];
        const size_t n = size_t(s.h) * size_t(s.w);
        if (!ensure_pixels(s, n)) return false;
        if (on_device) {
            s.b.img = imgs[s.job];
        } else {
            if (n > s.img_cap) { if (!dev_alloc(s.d_img, n)) return false; s.img_cap = n; }
            HIP_OK(hipMemcpyAsync(s.d_img, imgs[s.job], n, hipMemcpyHostToDevice, g.stream));
            s.b.img = s.d_img;
        }
        const bool wide = !serial_model_rows_fit(s.w);                   // rows do not fit in LDS: taps come from the reconstruction in memory
        const bool want_recon = s.near > 0 || wide;
        if (want_recon && n > s.recon_cap) { if (!dev_alloc(s.d_recon, n)) return false; s.recon_cap = n; }
        const size_t st = stats_doubles(s.effort, s.w);
        if (st > s.stats_cap) { if (!dev_alloc(s.d_stats, st)) return false; s.stats_cap = st; }
        if (st) HIP_OK(hipMemsetAsync(s.d_stats, 0, st * sizeof(double), g.stream));         // NBLIC.c:789
        E1Job &J = g.h_jobs[k];
        J.b = s.b; J.h = s.h; J.w = s.w; J.n = uint32_t(n); J.pp = make_plan(J.n); J.n_ev = 0; J.pe = make_plan(0, kTouchSegments); J.dbg = 0;
        J.near = s.near; J.k_step = k_step_for_near(s.near); J.ktab = level_shift_table(J.k_step);
        SerialJob &Q = g.h_sjobs[k];
        Q = SerialJob{};
        Q.img = s.b.img; Q.recon = want_recon ? s.d_recon : nullptr; Q.rec1 = s.b.rec1; Q.pxs = s.b.pxs; Q.stats = s.d_stats;
        Q.h = s.h; Q.w = s.w; Q.near = s.near; Q.k_step = J.k_step; Q.effort = s.effort;
        Q.state = s.d_state; Q.rows = serial_rows_per_launch(s.h, s.w, s.effort, c->serial_rows);
        HIP_OK(hipMemsetAsync(s.d_state, 0, sizeof(SerialState), g.stream));               // a fresh image: row 0, running
    }
    HIP_OK(hipMemcpyAsync(g.d_jobs, g.h_jobs, size_t(g.n_jobs) * sizeof(E1Job), hipMemcpyHostToDevice, g.stream));
    HIP_OK(hipMemcpyAsync(g.d_sjobs, g.h_sjobs, size_t(g.n_jobs) * sizeof(SerialJob), hipMemcpyHostToDevice, g.stream));
    e1_launch_init(g.d_jobs, g.n_jobs, g.stream);
    for (int k0 = 0; k0 < g.n_jobs;) {
        int k1 = k0 + 1;
        while (k1 < g.n_jobs && g.slots[size_t(k1)].effort == g.slots[size_t(k0)].effort) k1++;
        // an image is worked through `rows` rows per launch (its state record carries it from one to the next), so no
        // kernel runs longer than a few seconds however large the image; images that are done return at once
        int launches = 1;
        for (int k = k0; k < k1; k++) launches = std::max(launches, serial_launches(g.h_sjobs[k].h, g.h_sjobs[k].rows));
        for (int l = 0; l < launches; l++)
            if (!serial_model_launch(g.d_sjobs + k0, g.h_sjobs + k0, k1 - k0, g.stream)) return false;
        { std::lock_guard<std::mutex> sl(c->stat_m); c->serial_launch_count += launches; }
        k0 = k1;
    }
    for (int k = 0; k < g.n_jobs; k++) {                                 // the encoder leaves the reconstruction in the caller's plane (NBLIC.c:876)
        Slot &s = g.slots[size_t(k)];
        unsigned char *dst = g.recons ? g.recons[s.job] : nullptr;
        if (!dst) continue;
        const size_t n = size_t(s.h) * size_t(s.w);
        if (s.near > 0) HIP_OK(hipMemcpyAsync(dst, s.d_recon, n, hipMemcpyDeviceToHost, g.stream));
        else if (!on_device && dst != imgs[s.job]) memcpy(dst, imgs[s.job], n);
        else if (on_device) HIP_OK(hipMemcpyAsync(dst, imgs[s.job], n, hipMemcpyDeviceToHost, g.stream));
    }
    e1_launch_front_pre(g.d_jobs, g.h_jobs, g.n_jobs, g.stream);
    HIP_OK(hipMemcpyAsync(g.h_totals, g.d_totals, size_t(g.n_jobs) * kTotalsStride * sizeof(uint32_t), hipMemcpyDeviceToHost, g.stream));
    return true;
}

// ---- bins leave HBM in the layout the host coder wants ---------------------------------------
// A chunk of up to sixteen images becomes ONE contiguous device->host copy of 13-bit groups (range_coder.h,
// layout in range_coder_x8.cpp): rows[(13 * g + j) * 16 + lane] = word j of the thirteen 64-bit words that hold
// bins 64g .. 64g+63 of lane `lane` (zero past the lane's end).  On the host a pack's word is one aligned
// 64-byte load, and the link -- which bounds the pipeline -- carries 13 bits per bin instead of 16.
// One thread produces one WORD: it reads the four records the word's low 52 bits hold (8 bytes) and the record
// whose probability rides in its top field (2 bytes; word 12 reads the twelve records whose bins it collects) -- all
// thirteen threads of a group read inside the same 128-byte line, the sixteen lanes of a word sit side by side, and a
// wave stores four words x sixteen lanes = 512 contiguous bytes.  No LDS and a handful of registers ON PURPOSE: this
// kernel runs on the coder threads' streams underneath the encoder's own kernels, and what it costs is the time its
// workgroups wait for a slot, not its memory efficiency.  (Measured in the pipeline, per 4 Mbin chunk: a thread per
// group fetching its own 128-byte line 3.5 ms; the same with the lines staged through 33 KB of LDS by coalesced
// loads 5.9 ms -- the big workgroups find a CU late; the 16-bit interleave this replaces 2.8 ms.)
// (Measured and rejected: letting this kernel store straight into the mapped host ring.  The
// PCIe-bound waves crowd the encoder's own kernels off the GPU: 4.6 -> 2.4 Gpx/s.)
struct InterleaveArgs { const uint16_t *src[kMaxTake]; uint32_t len[kMaxTake]; };
__global__ void __launch_bounds__(256) k_pack_groups(InterleaveArgs a, uint64_t *__restrict__ rows, uint32_t n_words, uint32_t lanes) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    const uint32_t word = t / lanes, lane = t - word * lanes;   // word = 13 * group + j; lanes = 16 (a pack pair) or 24 (three packs)
    if (word >= n_words) return;
    const uint32_t g = word / uint32_t(kGroupWords), j = word - g * uint32_t(kGroupWords);
    const uint32_t pos = g * uint32_t(kGroupBins), len = a.len[lane];
    const uint16_t *src = a.src[lane] + pos;                     // chunk starts are multiples of 64 bins in 256-byte aligned buffers
    auto rec = [&](uint32_t k) { return pos + k < len ? uint64_t(code13(src[k])) : uint64_t(0); };
    uint64_t v;
    if (pos + uint32_t(kGroupBins) <= len) {
        const uint64_t q = *reinterpret_cast<const uint64_t *>(src + 4u * j);
        v = uint64_t(code13(uint32_t(q) & 0xFFFFu)) | uint64_t(code13(uint32_t(q >> 16) & 0xFFFFu)) << 13 |
            uint64_t(code13(uint32_t(q >> 32) & 0xFFFFu)) << 26 | uint64_t(code13(uint32_t(q >> 48))) << 39;
        if (j < 12u) {
            v |= uint64_t(src[52u + j] & 0xFFFu) << 52;
        } else {
            const uint64_t *tail = reinterpret_cast<const uint64_t *>(src + 52);
            const uint64_t m = 0x8000800080008000ull;
            // the four bins of each 8-byte load sit at bits 15, 31, 47, 63: gather them to bits 0..3
            auto bins4 = [&](uint64_t x) { x &= m; return ((x >> 15) | (x >> 30) | (x >> 45) | (x >> 60)) & 0xFull; };
            v |= (bins4(tail[0]) | bins4(tail[1]) << 4 | bins4(tail[2]) << 8) << 52;
        }
    } else {                                                     // the lane's last group of the chunk (or nothing at all)
        v = rec(4u * j) | rec(4u * j + 1u) << 13 | rec(4u * j + 2u) << 26 | rec(4u * j + 3u) << 39;
        if (j < 12u) v |= (rec(52u + j) & 0xFFFu) << 52;
        else for (uint32_t e = 0; e < 12u; e++) v |= (rec(52u + e) >> 12) << (52u + e);
    }
    rows[t] = v;
}

// What a coder thread owns: a pinned ring of two half-buffers x sixteen lanes x kChunkBins, so chunk
// c+1 lands while chunk c is coded.  Its device->host copies go through one of the context's few
// copy streams: a stream per thread would outnumber the hardware queues, and streams that share a
// hardware queue with a group's kernels have their copies stuck behind those kernels.
struct CoderThread {
    hipStream_t stream = nullptr;
    hipEvent_t ev[kRingDepth] = {};
    uint16_t *ring = nullptr;
    uint64_t *d_rows = nullptr;                          // device: two halves of 13-bit groups (k_pack_groups' output)
    uint16_t *whole = nullptr; size_t whole_cap = 0;     // pinned; one whole QNBLIC image (its rANS runs last pixel first)
    RangeX8 x8, x8b, x8c;
    RangeScalar x1;
    double wait_s = 0, issue_s = 0;                      // time spent waiting for chunks / inside the runtime calls that queue a chunk (reporting)
    bool init(int device, hipStream_t copy_stream) {
        HIP_OK(hipSetDevice(device));
        stream = copy_stream;
        for (auto &e : ev) HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventBlockingSync));
        return true;
    }
    void destroy() {
        locked_free(ring);
        locked_free(whole);
        if (d_rows) hipFree(d_rows);
        for (auto &e : ev) if (e) hipEventDestroy(e);
    }
    // The ring holds kRingDepth slots of ring_lanes x ring_chunk bins; it is sized by what the thread has actually been
    // asked to code (one lane for an image coded alone, sixteen for a pack pair; the chunk no longer than the longest
    // image) and only grows: a context that codes one small image through the drop-in entry points pins kilobytes,
    // the bench's threads end up at 3 x 24 x 4 Mbin x 1.625 B = 491 MB each.
    size_t ring_lanes = 0, ring_chunk = 0, rows_cap = 0;
    // 16-bit words per ring slot: a lone image's chunk as it is, or the 13-bit groups of ring_lanes lanes
    size_t slot_words() const { return ring_lanes > 1 ? group_words(ring_chunk, ring_lanes) * 4 : ring_chunk; }
    bool ensure_ring(size_t lanes, size_t chunk, bool need_rows) {
        chunk = (chunk + 4095) & ~size_t(4095);
        if (lanes > ring_lanes || chunk > ring_chunk) {
            const size_t nl = lanes > ring_lanes ? lanes : ring_lanes, nc = chunk > ring_chunk ? chunk : ring_chunk;
            locked_free(ring);
            ring_lanes = nl; ring_chunk = nc;
            ring = locked_alloc(kRingDepth * slot_words());
            if (!ring) { ring_lanes = ring_chunk = 0; fprintf(stderr, "[nblic_amd] cannot allocate the coder thread's ring\n"); return false; }
        }
        const size_t want_rows = need_rows ? kRingDepth * group_words(ring_chunk, ring_lanes) : 0;
        if (want_rows > rows_cap) {
            if (d_rows) hipFree(d_rows);
            d_rows = nullptr; rows_cap = 0;
            HIP_OK(hipMalloc((void **)&d_rows, want_rows * sizeof(uint64_t)));
            rows_cap = want_rows;
        }
        return true;
    }
    uint16_t *slot(size_t chunk) { return ring + size_t(chunk % kRingDepth) * slot_words(); }            // a lone image's chunk
    uint64_t *rows(size_t chunk) { return reinterpret_cast<uint64_t *>(slot(chunk)); }
    uint64_t *dev_rows(size_t chunk) { return d_rows + size_t(chunk % kRingDepth) * group_words(ring_chunk, ring_lanes); }
};

// Streams `take` images' bins from HBM and codes them: one image with the scalar coder, up to
// eight in the lanes of the AVX-512 coder, up to sixteen as two packs in lock-step.
// lens[k] = coder bytes or SIZE_MAX.
static bool code_streamed(CoderThread &t, const uint16_t *const *dev, const size_t *n, int take, uint8_t *const *dst,
                          const size_t *caps, size_t *lens, size_t chunk_bins) {
    size_t n_max = 0;
    for (int k = 0; k < take; k++) n_max = n[k] > n_max ? n[k] : n_max;
    // More than one image: AVX-512 packs in lock-step -- a lone pack is bound by the latency of its own dependent
    // chain, a second one rides along almost for free, a third on what the core's ports have left (+20 % bins per
    // CPU-second on the records of real frames, and the rank's CPU quota is what bounds the pipeline).  Up to sixteen
    // images make two packs (16 lanes per word-row), more make three (24 lanes); the images are dealt to the packs
    // in order, as evenly as they go: pack p owns lanes 8p .. 8p + count_p - 1.
    const int n_packs = take > 16 ? 3 : (take > 1 ? 2 : 0);
    const size_t lanes = size_t(8 * n_packs);
    int pack_n[3] = {0, 0, 0}, pack_first[3] = {0, 0, 0};
    for (int p = 0, at = 0; p < n_packs; p++) { pack_n[p] = take / n_packs + (p < take % n_packs ? 1 : 0); pack_first[p] = at; at += pack_n[p]; }
    if (!t.ensure_ring(take > 1 ? lanes : 1, n_max < chunk_bins ? n_max + 4 : chunk_bins, take > 1)) return false;
    const size_t chunks = (n_max + chunk_bins - 1) / chunk_bins;                     // chunk_bins <= kChunkBins, the ring's slot size
    auto chunk_len = [&](size_t c, int k) { const size_t off = c * chunk_bins; return off >= n[k] ? size_t(0) : (n[k] - off < chunk_bins ? n[k] - off : chunk_bins); };
    auto lane_of = [&](int k) { int p = 0; while (p + 1 < n_packs && k >= pack_first[p + 1]) p++; return 8 * p + (k - pack_first[p]); };
    auto issue = [&](size_t c) -> bool {
        if (take == 1) {                                      // one image: its bins as they are, for the scalar coder
            HIP_OK(hipMemcpyAsync(t.slot(c), dev[0] + c * chunk_bins, chunk_len(c, 0) * sizeof(uint16_t), hipMemcpyDeviceToHost, t.stream));
        } else {                                              // a pack pair: interleaved on the GPU, one copy
            InterleaveArgs a{};
            size_t longest = 0;
            for (int k = 0; k < take; k++) {
                const size_t len = chunk_len(c, k);
                a.src[lane_of(k)] = dev[k] + c * chunk_bins; a.len[lane_of(k)] = uint32_t(len);
                longest = len > longest ? len : longest;
            }
            const uint32_t n_groups = uint32_t((longest + kGroupBins - 1) / kGroupBins);
            // (Measured and rejected, twice: letting this kernel store straight into the mapped host ring.  Round 1, chip-wide
            // grid: 4.6 -> 2.4 Gpx/s.  Round 2, small grids so that few CUs wait on the link: 3.34 / 2.68 / 2.46 Gpx/s with
            // 16 / 48 / 128 workgroups per chunk against 5.5 with the staging pass + runtime copy.)
            uint64_t *d = t.dev_rows(c);
            static const int feed_dbg = getenv("NBLIC_AMD_DBG") ? atoi(getenv("NBLIC_AMD_DBG")) : 0;      // measurement aids (with & 128): & 512 no pack kernel, & 1024 no copy
            if (n_groups) {
                if (!(feed_dbg & 512)) hipLaunchKernelGGL(k_pack_groups, dim3((n_groups * uint32_t(kGroupWords) * uint32_t(lanes) + 255u) / 256u), dim3(256), 0, t.stream, a, d, n_groups * uint32_t(kGroupWords), uint32_t(lanes));
                HIP_OK(hipGetLastError());
                // (In this pipeline the runtime performs the copy with its blit kernel -- four 32 MB dispatches per 128 MB chunk --
                // whatever was tried: ring from hipHostMalloc instead of hipHostRegister, 2 / 4 / 8 copy streams, the copy cut
                // into 8 or 16 MB pieces; the same copy from a bare test program goes through SDMA.  DESIGN.md section 4.)
                if (!(feed_dbg & 1024)) HIP_OK(hipMemcpyAsync(t.rows(c), d, group_words(longest, lanes) * sizeof(uint64_t), hipMemcpyDeviceToHost, t.stream));
            }
        }
        // (Measured and rejected: sleeping on a condition variable woken by a host function behind the copy, as the
        // driver threads do.  A host function holds its stream until it has run, two threads share a copy stream,
        // and the chunks arrived so much later that the threads fell back to packs of eight: 6.1 -> 4.0 Gpx/s.
        // The threads wait for chunks for < 10 % of their time, so what the event wait burns is small.)
        HIP_OK(hipEventRecord(t.ev[c % kRingDepth], t.stream));
        return true;
    };
    RangeX8 *const packs[3] = {&t.x8, &t.x8b, &t.x8c};
    if (take > 1) { for (int p = 0; p < n_packs; p++) packs[p]->begin(pack_n[p], dst + pack_first[p], caps + pack_first[p]); }
    else t.x1.begin(dst[0], caps[0]);
    for (size_t c = 0; c + 1 < size_t(kRingDepth) && c < chunks; c++) if (!issue(c)) return false;
    for (size_t c = 0; c < chunks; c++) {
        auto i0 = std::chrono::steady_clock::now();
        if (c + kRingDepth - 1 < chunks && !issue(c + kRingDepth - 1)) return false;      // its ring slot was consumed one chunk ago
        auto w0 = std::chrono::steady_clock::now();
        t.issue_s += std::chrono::duration<double>(w0 - i0).count();
        // a sleeping poll: hipEventSynchronize spins through the wait (see GroupWait), and CPU time is what the rank is
        // short of; a chunk is ~30 ms of coding, so 100 us of extra latency on its arrival is nothing
        for (;;) {
            const hipError_t q = hipEventQuery(t.ev[c % kRingDepth]);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) { fprintf(stderr, "[nblic_amd] HIP error: %s\n", hipGetErrorString(q)); return false; }
            std::this_thread::sleep_for(std::chrono::microseconds(100));
        }
        t.wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
        if (take > 1) {
            size_t len[kMaxTake] = {0};
            for (int k = 0; k < take; k++) len[lane_of(k)] = chunk_len(c, k);
            static const bool feed_only = getenv("NBLIC_AMD_DBG") && (atoi(getenv("NBLIC_AMD_DBG")) & 128);   // measurement aid: bins reach the host but are not coded
            if (feed_only) {}
            else if (n_packs == 3) feed_triple_groups(t.x8, t.x8b, t.x8c, t.rows(c), len);
            else feed_pair_groups(t.x8, t.x8b, t.rows(c), len);
        } else {
            t.x1.feed(t.slot(c), chunk_len(c, 0));
        }
    }
    if (take > 1) { for (int p = 0; p < n_packs; p++) packs[p]->end(lens + pack_first[p]); }
    else lens[0] = t.x1.finish();
    return true;
}

// Coder thread.  Measured on the GPU box (EPYC 9575F), per thread: one stream alone 400-510
// Mbins/s; packs always run as two AVX-512 registers in lock-step: 2 x 4 images 1300 Mbins/s,
// 2 x 8 images 1950 Mbins/s -- at 2.5x / 3.4x the latency of a stream coded alone.  The host's
// CPU share (16 cores, enforced as a quota), not the GPU, bounds the pipeline, so what counts is
// bins per CPU-second: mid-batch, once every other thread is busy, a thread waits the ~30 ms it
// takes for sixteen images to be queued rather than start a smaller pack (while others are idle --
// the start of a batch -- it takes what is there, so all threads are at work within 0.4 s);
// towards the end it takes whatever is there; and only
// when at most two images per thread are left -- a short batch, or the very tail of a long one --
// does each image go to a thread of its own.  (Alternatives ranked with a
// discrete-event model of arrivals and coder speeds, tools/coder_policy_sim.py, then in situ.)
static int coder_take(const nblic_amd_ctx *c) {                  // call with c->rm held; 0 = nothing to take
    const size_t q = c->ready.size();
    if (q == 0) return 0;
    if (c->ready.front().kind == 1 || !c->simd) return 1;
    const size_t left = q + size_t(c->batch_to_come), threads = c->coders.size();
    if (left <= 2 * threads) return 1;                           // two rounds of singles beat one small pack
    const size_t full = size_t(c->max_take);
    if (c->batch_to_come > 0 && left >= 4 * threads) {
        // mid-batch: with every other thread busy wait (~30-45 ms) for a full set; with others idle too -- the start of a
        // batch, or the GPU side not keeping up -- at least for two full packs: packs of four lanes cost twice the CPU
        // time per bin, and CPU time is what the rank is short of
        const size_t want = c->idle_coders <= 1 ? full : (full < 16 ? full : size_t(16));
        if (q < want) return 0;
    }
    return int(q < full ? q : full);
}

// One logical CPU per physical core of the process's affinity mask (the lowest-numbered sibling that is allowed).
// NBLIC_AMD_PIN=1 pins coder thread i to core i of the mask (when there are enough cores), so that two coders --
// each a dependent chain per AVX-512 lane that keeps its core's vector unit busy by itself -- never share the SMT
// siblings of one core.  Off by default: measured on the GPU box, alternating runs, 5712 / 5356 Mpx/s pinned against
// 5642 / 5642 left to the scheduler -- a pinned thread cannot step aside when a driver or runtime thread is put on
// its CPU, and under the box's CPU quota that costs as much as the pinning saves.
static std::vector<int> primary_cpus() {
    std::vector<int> out;
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) != 0) return out;
    std::vector<int> cores;
    for (int cpu = 0; cpu < CPU_SETSIZE; cpu++) {
        if (!CPU_ISSET(cpu, &set)) continue;
        char path[96];
        snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", cpu);
        int first = cpu;
        if (FILE *f = fopen(path, "r")) { if (fscanf(f, "%d", &first) != 1) first = cpu; fclose(f); }
        if (std::find(cores.begin(), cores.end(), first) == cores.end()) { cores.push_back(first); out.push_back(cpu); }
    }
    return out;
}

static void coder_main(nblic_amd_ctx *c, int index) {
    pthread_setname_np(pthread_self(), "nblic-coder");         // (thread names: who uses the rank's CPU share, tools/thread_cpu.py)
    {
        static const std::vector<int> cpus = primary_cpus();
        static const bool pin = getenv("NBLIC_AMD_PIN") && atoi(getenv("NBLIC_AMD_PIN")) != 0;
        static const int pin_raw = getenv("NBLIC_AMD_PIN") ? atoi(getenv("NBLIC_AMD_PIN")) : 0;
        if (pin_raw >= 100) {                                     // experiment: coder i on logical CPU (pin_raw - 100) + i, whatever the process's mask
            cpu_set_t one;
            CPU_ZERO(&one);
            CPU_SET(pin_raw - 100 + index, &one);
            pthread_setaffinity_np(pthread_self(), sizeof one, &one);
        } else if (pin && c->coders_wanted > 1 && cpus.size() >= size_t(c->coders_wanted)) {
            cpu_set_t one;
            CPU_ZERO(&one);
            static std::atomic<unsigned> next_core{0};            // across contexts: a second context's threads take the next cores
            CPU_SET(cpus[size_t(next_core++ % cpus.size())], &one);
            pthread_setaffinity_np(pthread_self(), sizeof one, &one);
        }
    }
    CoderThread t;
    if (!t.init(c->device, c->copy_streams[size_t(index) % c->copy_streams.size()])) c->broken = true;
    for (;;) {
        ReadyImage im[kMaxTake];
        int take = 0;
        {
            std::unique_lock<std::mutex> l(c->rm);
            c->idle_coders++;
            c->rcv.wait(l, [c] { return c->stop || coder_take(c) > 0; });
            if (c->ready.empty()) break;
            take = coder_take(c);
            if (take == 0) break;                                // shutdown while waiting for a pack to fill
            c->idle_coders--;
            for (int k = 0; k < take; k++) {
                if (k > 0 && c->ready.front().kind == 1) { take = k; break; }
                im[k] = c->ready.front(); c->ready.pop_front();
            }
        }
        if (im[0].kind == 1) {                               // QNBLIC: histogram normalisation + rANS, one image per thread
            const ReadyImage &q = im[0];
            const size_t n = size_t(q.h) * size_t(q.w), n_pad = (n + 1) & ~size_t(1), words = n_pad + 2 * 12 * 256;
            bool ok = true;
            if (t.whole_cap < words) {
                locked_free(t.whole);
                t.whole = nullptr; t.whole_cap = 0;
                if ((t.whole = locked_alloc(words + 1024)) != nullptr) t.whole_cap = words + 1024;
                else ok = false;
            }
            ok = ok && hipMemcpyAsync(t.whole, c->cbufs[size_t(q.cb)].p, words * sizeof(uint16_t), hipMemcpyDeviceToHost, t.stream) == hipSuccess &&
                 hipEventRecord(t.ev[0], t.stream) == hipSuccess && hipEventSynchronize(t.ev[0]) == hipSuccess;
            long words_out = -1;
            if (ok) {
                const uint32_t *hist = reinterpret_cast<const uint32_t *>(t.whole + n_pad);
                words_out = q_entropy_encode(reinterpret_cast<uint16_t *>(q.outs[q.job]), q.caps[q.job], q.h, q.w, t.whole, hist);
                if (words_out < 0) fprintf(stderr, "[nblic_amd] image %d: output buffer of %zu words is too small\n", q.job, q.caps[q.job]);
            }
            q.lens[q.job] = words_out;                           // -1: the batch this image belongs to reports the failure
            { std::lock_guard<std::mutex> l(c->fm); c->free_cbufs.push_back(q.cb); c->coding -= 1; if (q.batch) q.batch->remaining -= 1; }
            c->fcv.notify_all();
            continue;
        }
        auto t0 = std::chrono::steady_clock::now();
        if (c->trace) fprintf(stderr, "[trace] %.3f coder %d takes %d\n", c->now(), index, take);
        const uint16_t *src[kMaxTake]; size_t n[kMaxTake], caps[kMaxTake], lens[kMaxTake]; uint8_t *dst[kMaxTake];
        double bins = 0;
        for (int k = 0; k < take; k++) {
            src[k] = c->cbufs[size_t(im[k].cb)].p; n[k] = im[k].n_ev; bins += double(im[k].n_ev);
            const size_t cap = im[k].caps[im[k].job] < (size_t(1) << 46) ? im[k].caps[im[k].job] : (size_t(1) << 46);   // SIZE_MAX = "no limit"
            dst[k] = im[k].outs[im[k].job] + kHeaderBytes;
            caps[k] = cap >= size_t(kHeaderBytes) ? cap - kHeaderBytes : 0;
            if (cap >= size_t(kHeaderBytes)) write_header(im[k].outs[im[k].job], im[k].h, im[k].w, im[k].near, im[k].k_step, im[k].effort);
        }
        static const bool skip_coding = getenv("NBLIC_AMD_DBG") && (atoi(getenv("NBLIC_AMD_DBG")) & 16);   // measurement aid: device side alone
        if (skip_coding) { for (int k = 0; k < take; k++) lens[k] = 0; }
        else if (!code_streamed(t, src, n, take, dst, caps, lens, c->chunk_bins)) {
            hipDeviceSynchronize();
            for (int k = 0; k < take; k++) lens[k] = SIZE_MAX;
        }
        for (int k = 0; k < take; k++) {
            if (lens[k] == SIZE_MAX) fprintf(stderr, "[nblic_amd] image %d: output buffer of %zu bytes is too small\n", im[k].job, im[k].caps[im[k].job]);
            im[k].lens[im[k].job] = lens[k] == SIZE_MAX ? -1 : long(kHeaderBytes + lens[k]);
        }
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (c->trace) fprintf(stderr, "[trace] %.3f coder %d finished %d in %.3f s\n", c->now(), index, take, dt);
        { std::lock_guard<std::mutex> l(c->stat_m); c->total_bins += bins; c->coder_s += dt; if (take > 1) { c->pack_bins += bins; c->pack_s += dt; } c->wait_s += t.wait_s; t.wait_s = 0; c->issue_s += t.issue_s; t.issue_s = 0; c->takes[take]++; }
        {
            std::lock_guard<std::mutex> l(c->fm);
            for (int k = 0; k < take; k++) { c->free_cbufs.push_back(im[k].cb); if (im[k].batch) im[k].batch->remaining -= 1; }
            c->coding -= take;
        }
        c->fcv.notify_all();
    }
    t.destroy();
}

// ---- device coder pack threads -----------------------------------------------------------------
// A pack thread waits until the queue holds a full pack BEYOND what the host coder threads can take
// at once (they keep priority: a host core codes an image forty times faster than a lane) and until
// enough work is outstanding that the pack's latency (seconds) cannot become the tail of the batch;
// then it hands up to 64 images to one wave, sleeps in a blocking stream wait, copies the coder bytes
// to the callers' buffers and completes the images exactly as a host coder thread does.
constexpr int kDevPack = 64;

static int dev_take(const nblic_amd_ctx *c) {                  // call with c->rm held
    const size_t q = c->ready.size();
    const size_t reserve = c->simd ? c->coders.size() * size_t(kMaxTake) / 2 : c->coders.size();
    if (q < size_t(kDevPack) + reserve) return 0;
    if (int(q) + c->batch_to_come + c->queued_images < c->dev_min_outstanding) return 0;
    for (size_t k = 0; k < size_t(kDevPack); k++) if (c->ready[q - 1 - k].kind == 1) return 0;   // QNBLIC images are host work
    return kDevPack;
}

static void dev_coder_main(nblic_amd_ctx *c, int index) {
    pthread_setname_np(pthread_self(), "nblic-devcoder");
    (void)index;
    hipStream_t st = nullptr;
    hipEvent_t done = nullptr;
    RcJob *h_jobs = nullptr, *d_jobs = nullptr;
    uint32_t *h_lens = nullptr, *d_lens = nullptr;
    uint8_t *d_out = nullptr; size_t out_cap = 0;
    bool ok = hipSetDevice(c->device) == hipSuccess && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&done, hipEventDisableTiming | hipEventBlockingSync) == hipSuccess &&
              hipHostMalloc((void **)&h_jobs, kDevPack * sizeof(RcJob), hipHostMallocDefault) == hipSuccess &&
              hipMalloc((void **)&d_jobs, kDevPack * sizeof(RcJob)) == hipSuccess &&
              hipHostMalloc((void **)&h_lens, kDevPack * sizeof(uint32_t), hipHostMallocDefault) == hipSuccess &&
              hipMalloc((void **)&d_lens, kDevPack * sizeof(uint32_t)) == hipSuccess;
    if (!ok) c->broken = true;
    for (;;) {
        ReadyImage im[kDevPack];
        int take = 0;
        {
            std::unique_lock<std::mutex> l(c->rm);
            c->rcv.wait(l, [c] { return c->stop || dev_take(c) > 0; });
            if (c->stop) break;
            take = dev_take(c);
            for (int k = 0; k < take; k++) { im[k] = c->ready.back(); c->ready.pop_back(); }   // the NEWEST images: the oldest are the host threads' next packs
        }
        if (take == 0) continue;
        // output slots on the device: worst case seen is 1.0025 B/px + 20
        size_t need = 0, off[kDevPack];
        for (int k = 0; k < take; k++) { off[k] = need; need += (size_t(im[k].h) * size_t(im[k].w) * 9 / 8 + 4096 + 255) & ~size_t(255); }
        bool good = ok;
        if (good && need > out_cap) { hipFree(d_out); d_out = nullptr; out_cap = 0; good = hipMalloc((void **)&d_out, need) == hipSuccess; if (good) out_cap = need; }
        double bins = 0;
        for (int k = 0; k < take && good; k++) {
            const size_t cap_user = im[k].caps[im[k].job] < (size_t(1) << 40) ? im[k].caps[im[k].job] : (size_t(1) << 40);
            const size_t cap_dev = (k + 1 < take ? off[k + 1] : need) - off[k];
            const size_t cap = cap_user > size_t(kHeaderBytes) ? (cap_user - kHeaderBytes < cap_dev ? cap_user - kHeaderBytes : cap_dev) : 0;
            h_jobs[k] = RcJob{c->cbufs[size_t(im[k].cb)].p, d_out + off[k], d_lens + k, im[k].n_ev, uint32_t(cap < 0xFFFFFFF0u ? cap : 0xFFFFFFF0u)};
            bins += double(im[k].n_ev);
        }
        good = good && hipMemcpyAsync(d_jobs, h_jobs, size_t(take) * sizeof(RcJob), hipMemcpyHostToDevice, st) == hipSuccess &&
               device_range_code(d_jobs, take, st) &&
               hipMemcpyAsync(h_lens, d_lens, size_t(take) * sizeof(uint32_t), hipMemcpyDeviceToHost, st) == hipSuccess &&
               hipEventRecord(done, st) == hipSuccess && hipEventSynchronize(done) == hipSuccess;
        for (int k = 0; k < take; k++) {
            long len = -1;
            if (good && h_lens[k] != 0xFFFFFFFFu) {
                unsigned char *dst = im[k].outs[im[k].job];
                write_header(dst, im[k].h, im[k].w, im[k].near, im[k].k_step, im[k].effort);
                if (hipMemcpyAsync(dst + kHeaderBytes, d_out + off[k], h_lens[k], hipMemcpyDeviceToHost, st) == hipSuccess) len = long(kHeaderBytes) + long(h_lens[k]);
            } else if (good) {
                fprintf(stderr, "[nblic_amd] image %d: output buffer of %zu bytes is too small\n", im[k].job, im[k].caps[im[k].job]);
            }
            im[k].lens[im[k].job] = len;
        }
        if (hipStreamSynchronize(st) != hipSuccess) good = false;
        if (!good) { for (int k = 0; k < take; k++) im[k].lens[im[k].job] = -1; }
        { std::lock_guard<std::mutex> l(c->stat_m); c->dev_bins += bins; c->dev_packs++; c->dev_images += take; }
        {
            std::lock_guard<std::mutex> l(c->fm);
            for (int k = 0; k < take; k++) { c->free_cbufs.push_back(im[k].cb); if (im[k].batch) im[k].batch->remaining -= 1; }
            c->coding -= take;
        }
        c->fcv.notify_all();
    }
    hipFree(d_out); hipFree(d_jobs); hipFree(d_lens);
    if (h_jobs) hipHostFree(h_jobs);
    if (h_lens) hipHostFree(h_lens);
    if (done) hipEventDestroy(done);
    if (st) hipStreamDestroy(st);
}

// Takes a coded-bin buffer of at least `words` for slot s (waits for one if the coder threads are
// behind: that is the pipeline's back-pressure).
static bool acquire_coded(nblic_amd_ctx *c, Slot &s, size_t words) {
    {
        std::unique_lock<std::mutex> l(c->fm);
        c->fcv.wait(l, [c] { return !c->free_cbufs.empty(); });
        s.cb = c->free_cbufs.front(); c->free_cbufs.pop_front();
    }
    CodedBuf &cb = c->cbufs[size_t(s.cb)];
    if (cb.cap < words) {
        if (cb.p) hipFree(cb.p);
        cb.p = nullptr; cb.cap = 0;
        const size_t cap = words + words / 8 + 1024;
        HIP_OK(hipMalloc((void **)&cb.p, cap * sizeof(uint16_t)));
        cb.cap = cap;
    }
    return true;
}

// Runs on a HIP runtime thread when the group's kernels have finished: queues the images for the
// coder threads and hands the device workspace back.  (No HIP calls in here.)
static void on_group_done(void *vp) {
    Group *gp = static_cast<Group *>(vp);
    nblic_amd_ctx *c = gp->ctx;
    if (c->trace) fprintf(stderr, "[trace] %.3f group %d done (%d images)\n", c->now(), gp->id, gp->n_jobs);
    {
        std::lock_guard<std::mutex> l(c->rm);
        for (int k = 0; k < gp->n_jobs; k++) {
            const Slot &s = gp->slots[size_t(k)];
            c->ready.push_back(ReadyImage{s.cb, s.job, s.h, s.w, s.n_ev, gp->outs, gp->caps, gp->lens, gp->kind, gp->batch, s.near, k_step_for_near(s.near), s.effort});
        }
        c->batch_to_come -= gp->n_jobs;
    }
    c->rcv.notify_all();
    { std::lock_guard<std::mutex> l(c->fm); c->free_groups.push_back(gp->id); }
    c->fcv.notify_all();
}

static double thread_cpu_s();
static bool launch_back(nblic_amd_ctx *c, Group &g, bool with_coders, bool general = false) {
    // a BLOCKING wait: a spinning one per driver thread would take cores from the coder threads
    const double w0 = thread_cpu_s();
    { std::lock_guard<std::mutex> l(g.front->m); g.front->ready = false; }
    HIP_OK(hipLaunchHostFunc(g.stream, [](void *p) {
        GroupWait *w = static_cast<GroupWait *>(p);
        { std::lock_guard<std::mutex> l(w->m); w->ready = true; }
        w->cv.notify_one();
    }, g.front.get()));
    {   // a sleep, but not an unconditional one: if the stream has faulted the host function may never run
        std::unique_lock<std::mutex> l(g.front->m);
        while (!g.front->cv.wait_for(l, std::chrono::milliseconds(50), [&] { return g.front->ready; })) {
            l.unlock();
            const hipError_t q = hipStreamQuery(g.stream);
            l.lock();
            if (q != hipSuccess && q != hipErrorNotReady) { fprintf(stderr, "[nblic_amd] group %d: %s while waiting for the front half\n", g.id, hipGetErrorString(q)); return false; }
        }
    }
    { std::lock_guard<std::mutex> l(c->stat_m); c->driver_wait_cpu_s += thread_cpu_s() - w0; }
    for (int k = 0; k < g.n_jobs; k++) {
        Slot &s = g.slots[size_t(k)];
        s.n_ev = g.h_totals[size_t(k) * kTotalsStride + 2];
        if (s.n_ev >= 0x7FFFFFFFu) { fprintf(stderr, "[nblic_amd] event count overflow\n"); return false; }
        if (!ensure_events(s, s.n_ev)) return false;
        // the coded bins go straight into a pool buffer that outlives this group's turn on the slot
        if (!acquire_coded(c, s, size_t(s.n_ev) + 8)) return false;
        s.b.coded = c->cbufs[size_t(s.cb)].p;
        E1Job &J = g.h_jobs[k];
        J.b = s.b; J.n_ev = s.n_ev; J.pe = make_plan(s.n_ev, kTouchSegments);
    }
    HIP_OK(hipMemcpyAsync(g.d_jobs, g.h_jobs, size_t(g.n_jobs) * sizeof(E1Job), hipMemcpyHostToDevice, g.stream));
    e1_launch_back(g.d_jobs, g.h_jobs, g.n_jobs, g.stream, (c->timing && !general) ? &g.tm : nullptr, general);
    g.tm_pending = c->timing && !general;
    if (with_coders) HIP_OK(hipLaunchHostFunc(g.stream, on_group_done, &g));       // the caller has counted the images in ctx->coding
    return true;
}

static void collect_timing(nblic_amd_ctx *c, Group &g) {
    if (!g.tm_pending) return;
    g.tm_pending = false;
    int last = -1;
    for (int k = 0; k < kE1Kernels; k++) if ((g.tm.mask >> k) & 1ull) last = k;
    if (last < 0 || hipEventSynchronize(g.tm.ev[last + 1]) != hipSuccess) return;
    for (int k = 0; k < kE1Kernels; k++) {
        float ms = 0.f;
        if (((g.tm.mask >> k) & 1ull) && hipEventElapsedTime(&ms, g.tm.ev[k], g.tm.ev[k + 1]) == hipSuccess) c->stage_ms[k] += ms;
    }
    c->stage_launches++;
}

static void release_group(nblic_amd_ctx *c, int id) {
    { std::lock_guard<std::mutex> g(c->fm); c->free_groups.push_back(id); }
    c->fcv.notify_all();
}

static bool launch_q(nblic_amd_ctx *c, Group &g, const uint8_t *const *imgs, bool on_device);

static double thread_cpu_s() {
    timespec ts;
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
    return double(ts.tv_sec) + 1e-9 * double(ts.tv_nsec);
}

static void driver_main(nblic_amd_ctx *c, int id) {
    pthread_setname_np(pthread_self(), "nblic-driver");
    Group &g = c->groups[size_t(id)];
    if (hipSetDevice(c->device) != hipSuccess) fprintf(stderr, "[nblic_amd] driver thread: cannot select device %d\n", c->device);
    for (;;) {
        {
            std::unique_lock<std::mutex> l(c->dm);
            c->dcv.wait(l, [&] { return c->stop_drivers || g.has_work; });
            if (!g.has_work) return;
            g.has_work = false;
        }
        const double cpu0 = thread_cpu_s();
        const bool ok = g.kind == 0 ? (launch_front(c, g, g.imgs, g.on_device) && launch_back(c, g, true))
                      : g.kind == 2 ? (launch_front_serial(c, g, g.imgs, g.on_device) && launch_back(c, g, true, true))
                                    : launch_q(c, g, g.imgs, g.on_device);
        { std::lock_guard<std::mutex> l(c->stat_m); c->driver_cpu_s += thread_cpu_s() - cpu0; c->driver_launches++; }
        if (!ok) {
            hipStreamSynchronize(g.stream);
            {   // coded-bin buffers the failed launch had already taken go back to the pool
                std::lock_guard<std::mutex> l(c->fm);
                for (int k = 0; k < g.n_jobs; k++) {
                    Slot &s = g.slots[size_t(k)];
                    if (s.cb >= 0) { c->free_cbufs.push_back(s.cb); s.cb = -1; }
                }
            }
            { std::lock_guard<std::mutex> l(c->rm); c->batch_to_come -= g.n_jobs; }
            c->rcv.notify_all();                                 // a pack may be waiting for images that will not come
            { std::lock_guard<std::mutex> l(c->fm); c->coding -= g.n_jobs; if (g.batch) { g.batch->remaining -= g.n_jobs; g.batch->ok = false; } }
            release_group(c, id);
        }
    }
}

// The images are counted as outstanding BEFORE the driver thread is woken, so the batch's final
// wait cannot slip through between the hand-over and the launch.
static void start_group(nblic_amd_ctx *c, Group &g, const uint8_t *const *imgs, bool on_device) {
    { std::lock_guard<std::mutex> l(c->fm); c->coding += g.n_jobs; if (g.batch) g.batch->remaining += g.n_jobs; }
    { std::lock_guard<std::mutex> l(c->dm); g.imgs = imgs; g.on_device = on_device; g.has_work = true; }
    c->dcv.notify_all();
}

// Groups are started one after the other and run concurrently on the GPU (a stream each); the
// host codes finished groups while the GPU is busy with the following ones.
// Submission half of a batch: hands the images to the groups (blocks only while all groups are
// busy, i.e. until the GPU is down to its last few groups of this batch).  The coder threads and the
// groups still in flight finish on their own; encode_wait() collects.  Several batches may be
// outstanding: the next one fills the pipeline while this one drains.
static void encode_submit(nblic_amd_ctx *c, nblic_amd_batch *b, int n_images, const uint8_t *const *imgs, bool on_device,
                          const int *hs, const int *ws, uint8_t *const *outs, const size_t *caps, long *lens,
                          const int *nears = nullptr, const int *efforts = nullptr, unsigned char *const *recons = nullptr) {
    bool idle;
    { std::lock_guard<std::mutex> l(c->fm); idle = c->coding == 0; }
    if (idle) {                                                   // nothing outstanding: start the reporting afresh
        for (auto &gr : c->groups) gr.tm_pending = false;         // (timer events of earlier batches that nobody collected belong to THEIR figures, not to this batch's)
        for (auto &v : c->stage_ms) v = 0;
        c->stage_launches = 0;
        c->total_bins = 0; c->coder_s = 0; c->pack_bins = 0; c->pack_s = 0; c->wait_s = 0; c->issue_s = 0; c->driver_cpu_s = 0; c->driver_wait_cpu_s = 0; c->driver_launches = 0; for (auto &v : c->takes) v = 0;
        c->dev_bins = 0; c->dev_packs = 0; c->dev_images = 0;
        c->t_batch = std::chrono::steady_clock::now();
        c->trace = getenv("NBLIC_AMD_DBG") && (atoi(getenv("NBLIC_AMD_DBG")) & 64);
    }
    b->n_images = n_images; b->lens = lens;
    for (int k = 0; k < n_images; k++) lens[k] = -1;
    // modes: (near, effort) clamped as the reference clamps them (NBLIC.c:768-770); -n0 -e1 images take the
    // staged pipeline (kind 0), everything else the serial model stage (kind 2).  Images are handed out kind by
    // kind and, inside kind 2, effort by effort, so a group's launches are homogeneous.
    auto near_of = [&](int k) { return nears ? iclip(nears[k], 0, kMaxNear) : 0; };
    auto effort_of = [&](int k) { return efforts ? iclip(efforts[k], 1, 3) : 1; };
    auto class_of = [&](int k) { return (near_of(k) == 0 && effort_of(k) == 1) ? 0 : effort_of(k); };   // 0 staged; 1..3 serial by effort
    std::vector<int> order;
    order.reserve(size_t(n_images));
    for (int cls = 0; cls <= 3; cls++)
        for (int k = 0; k < n_images; k++) {
            if (class_of(k) != cls) continue;
            if (!size_ok(hs[k], ws[k], c->max_px)) { b->ok = false; continue; }
            order.push_back(k);
        }
    { std::lock_guard<std::mutex> l(c->rm); c->batch_to_come += int(order.size()); }
    size_t next = 0;
    while (next < order.size()) {
        int id;
        {
            std::unique_lock<std::mutex> l(c->fm);
            c->fcv.wait(l, [c] { return !c->free_groups.empty(); });
            id = c->free_groups.front(); c->free_groups.pop_front();
        }
        Group &g = c->groups[size_t(id)];
        collect_timing(c, g);                               // events of its previous use are complete by now
        const int kind = class_of(order[next]) == 0 ? 0 : 2;
        g.outs = outs; g.caps = caps; g.lens = lens; g.kind = kind; g.batch = b; g.recons = recons;
        g.n_jobs = 0;
        while (next < order.size() && g.n_jobs < int(g.slots.size()) && (class_of(order[next]) == 0 ? 0 : 2) == kind) {
            const int k = order[next++];
            Slot &s = g.slots[size_t(g.n_jobs++)];
            s.job = k; s.h = hs[k]; s.w = ws[k]; s.cb = -1; s.near = near_of(k); s.effort = effort_of(k);
            if (kind == 0 && recons && recons[k]) {                       // -n0 -e1: the reconstruction IS the input (NBLIC.c:876 rewrites the same bytes)
                const size_t n = size_t(hs[k]) * size_t(ws[k]);
                if (on_device) { if (hipMemcpy(recons[k], imgs[k], n, hipMemcpyDeviceToHost) != hipSuccess) b->ok = false; }
                else if (recons[k] != imgs[k]) memcpy(recons[k], imgs[k], n);
            }
        }
        start_group(c, g, imgs, on_device);
    }
}

static void submitter_main(nblic_amd_ctx *c) {
    pthread_setname_np(pthread_self(), "nblic-submit");
    if (hipSetDevice(c->device) != hipSuccess) c->broken = true;
    for (;;) {
        nblic_amd_ctx::SubmitItem it;
        {
            std::unique_lock<std::mutex> l(c->sm);
            c->scv.wait(l, [c] { return c->stop_submit || !c->sq.empty(); });
            if (c->sq.empty()) return;
            it = c->sq.front(); c->sq.pop_front();
        }
        {
            std::lock_guard<std::mutex> g(c->api);
            { std::lock_guard<std::mutex> l(c->rm); c->queued_images -= it.n; }         // from here on they are counted in batch_to_come
            encode_submit(c, it.b, it.n, it.imgs, it.on_device, it.hs, it.ws, it.outs, it.caps, it.lens, it.nears, it.efforts, it.recons);
        }
        { std::lock_guard<std::mutex> l(c->fm); it.b->submitted = true; }
        c->fcv.notify_all();
    }
}

static void queue_batch(nblic_amd_ctx *c, const nblic_amd_ctx::SubmitItem &it) {
    for (int k = 0; k < it.n; k++) it.lens[k] = -1;
    { std::lock_guard<std::mutex> l(c->rm); c->queued_images += it.n; }
    { std::lock_guard<std::mutex> l(c->sm); c->sq.push_back(it); }
    c->scv.notify_all();
}

static void report_coders(nblic_amd_ctx *c) {                        // NBLIC_AMD_DBG & 32, when nothing is outstanding
    bool idle;
    { std::lock_guard<std::mutex> l(c->fm); idle = c->coding == 0; }
    if (!idle || !(getenv("NBLIC_AMD_DBG") && (atoi(getenv("NBLIC_AMD_DBG")) & 32))) return;
    fprintf(stderr, "[nblic_amd] coder: singles %.0f Mbins in %.2f thread-s (%.0f Mbins/s), packs %.0f Mbins in %.2f thread-s (%.0f Mbins/s)\n",
            (c->total_bins - c->pack_bins) / 1e6, c->coder_s - c->pack_s, (c->total_bins - c->pack_bins) / 1e6 / (c->coder_s - c->pack_s + 1e-9),
            c->pack_bins / 1e6, c->pack_s, c->pack_bins / 1e6 / (c->pack_s + 1e-9));
    fprintf(stderr, "[nblic_amd] coder: %.2f thread-s of that queueing chunks (runtime calls)\n", c->issue_s);
    fprintf(stderr, "[nblic_amd] drivers: %.2f CPU-s in %ld group launches (%.1f ms each), %.2f CPU-s of that inside the wait for the front half\n",
            c->driver_cpu_s, c->driver_launches, 1e3 * c->driver_cpu_s / double(c->driver_launches ? c->driver_launches : 1), c->driver_wait_cpu_s);
    long t2 = 0, t9 = 0, t17 = 0;
    for (int k = 2; k <= 7; k++) t2 += c->takes[k];
    for (int k = 9; k <= 15; k++) t9 += c->takes[k];
    for (int k = 17; k < kMaxTake; k++) t17 += c->takes[k];
    fprintf(stderr, "[nblic_amd] coder: %.2f thread-s of that waiting for chunks; takes of 1/2-7/8/9-15/16/17-23/24 images: %ld/%ld/%ld/%ld/%ld/%ld/%ld\n", c->wait_s,
            c->takes[1], t2, c->takes[8], t9, c->takes[16], t17, c->takes[kMaxTake]);
}

static bool encode_wait(nblic_amd_ctx *c, nblic_amd_batch *b) {
    {   // wait for the coder threads (and with them every group's GPU work) of THIS batch
        std::unique_lock<std::mutex> l(c->fm);
        c->fcv.wait(l, [b] { return b->submitted && b->remaining == 0; });
    }
    bool ok = b->ok;
    for (int k = 0; k < b->n_images; k++) if (b->lens[k] < 0) ok = false;
    return ok;
}

static bool encode_batch(nblic_amd_ctx *c, int n_images, const uint8_t *const *imgs, bool on_device, const int *hs,
                         const int *ws, uint8_t *const *outs, const size_t *caps, long *lens) {
    if (hipSetDevice(c->device) != hipSuccess) return false;
    nblic_amd_batch b;
    queue_batch(c, nblic_amd_ctx::SubmitItem{&b, n_images, imgs, on_device, hs, ws, outs, caps, lens, nullptr, nullptr, nullptr});
    bool ok = encode_wait(c, &b);
    bool idle;
    { std::lock_guard<std::mutex> l(c->fm); idle = c->coding == 0; }
    if (idle) for (auto &g : c->groups) collect_timing(c, g);
    report_coders(c);
    return ok && !c->broken;
}

// ---- QNBLIC (effort 0): model on the GPU, entropy stage on a coder thread ---------------------
static bool launch_q(nblic_amd_ctx *c, Group &g, const uint8_t *const *imgs, bool on_device) {
    for (int k = 0; k < g.n_jobs; k++) {
        Slot &s = g.slots[size_t(k)];
        size_t n = size_t(s.h) * size_t(s.w);
        if (!ensure_pixels(s, n, false)) return false;
        if (on_device) {
            s.b.img = imgs[s.job];
        } else {
            if (n > s.img_cap) { if (!dev_alloc(s.d_img, n)) return false; s.img_cap = n; }
            HIP_OK(hipMemcpyAsync(s.d_img, imgs[s.job], n, hipMemcpyHostToDevice, g.stream));
            s.b.img = s.d_img;
        }
        E1Job &J = g.h_jobs[k];
        J.b = s.b; J.h = s.h; J.w = s.w; J.n = uint32_t(n); J.pp = make_plan(J.n); J.n_ev = 0; J.pe = make_plan(0, kTouchSegments); J.dbg = 0;
        s.n_ev = 0;
    }
    HIP_OK(hipMemcpyAsync(g.d_jobs, g.h_jobs, size_t(g.n_jobs) * sizeof(E1Job), hipMemcpyHostToDevice, g.stream));
    q_launch_model(g.d_jobs, g.h_jobs, g.n_jobs, g.stream);
    g.tm_pending = false;
    for (int k = 0; k < g.n_jobs; k++) {
        Slot &s = g.slots[size_t(k)];
        const size_t n = size_t(s.h) * size_t(s.w), n_pad = (n + 1) & ~size_t(1), need = n_pad + 2 * 12 * 256;
        if (!acquire_coded(c, s, need)) return false;
        uint16_t *dst = c->cbufs[size_t(s.cb)].p;
        HIP_OK(hipMemcpyAsync(dst, s.b.pxs, n * sizeof(uint16_t), hipMemcpyDeviceToDevice, g.stream));
        HIP_OK(hipMemcpyAsync(dst + n_pad, s.b.qhist, 12 * 256 * sizeof(uint32_t), hipMemcpyDeviceToDevice, g.stream));
    }
    HIP_OK(hipLaunchHostFunc(g.stream, on_group_done, &g));
    return true;
}

static bool encode_q_batch(nblic_amd_ctx *c, int n_images, const uint8_t *const *imgs, bool on_device, const int *hs,
                           const int *ws, uint16_t *const *outs, const size_t *caps_words, long *len_words) {
    if (hipSetDevice(c->device) != hipSuccess) return false;
    bool ok = true;
    for (int k = 0; k < n_images; k++) len_words[k] = -1;
    { std::lock_guard<std::mutex> l(c->rm); c->batch_to_come += n_images; }
    int next = 0;
    while (next < n_images) {
        int id;
        {
            std::unique_lock<std::mutex> l(c->fm);
            c->fcv.wait(l, [c] { return !c->free_groups.empty(); });
            id = c->free_groups.front(); c->free_groups.pop_front();
        }
        Group &g = c->groups[size_t(id)];
        collect_timing(c, g);
        g.outs = reinterpret_cast<unsigned char *const *>(outs); g.caps = caps_words; g.lens = len_words; g.kind = 1; g.batch = nullptr;
        g.n_jobs = 0;
        while (next < n_images && g.n_jobs < int(g.slots.size())) {
            int k = next++;
            if (!size_ok(hs[k], ws[k], c->max_px)) { ok = false; std::lock_guard<std::mutex> l(c->rm); c->batch_to_come--; continue; }
            Slot &s = g.slots[size_t(g.n_jobs++)];
            s.job = k; s.h = hs[k]; s.w = ws[k]; s.cb = -1; s.near = 0; s.effort = 1;
        }
        if (g.n_jobs == 0) { release_group(c, id); continue; }
        start_group(c, g, imgs, on_device);
    }
    {
        std::unique_lock<std::mutex> l(c->fm);
        c->fcv.wait(l, [c] { return c->coding == 0; });
    }
    for (int k = 0; k < n_images; k++) if (len_words[k] < 0) ok = false;
    return ok && !c->broken;
}

// ---- decoders: every stream of a batch side by side, one wave per image (serial_engine.hip) -----
long q_decode_tables(const uint16_t *in, size_t n_words, int *h, int *w, uint32_t *freq, uint32_t *start, uint8_t *slot);

struct DecodeItem { int k, h, w, near, k_step, effort, kind; size_t len; long q_pos; int qtab; };      // kind 0 NBLIC, 1 QNBLIC; q_pos: first rANS word; qtab: which parsed table set

constexpr int kDecodeChunk = 1024;                                                         // images per chunk of decode_batch (above the lean decoder's threshold)
constexpr size_t kQTab = 2 * 12 * 256 * sizeof(uint32_t);                                   // QNBLIC: frequencies, cumulative starts (the kernel derives its symbol index from them)
static size_t up256(size_t v) { return (v + 255) & ~size_t(255); }

// Header of a stream of which `len` bytes are in hand (NBLIC.c:698-745, QNBLIC.c:475-486).  0 = not a stream this
// library decodes (or refused: size, parameters), 1 = fields filled in.
static int parse_stream_header(const unsigned char *p, size_t len, long max_px, DecodeItem &it) {
    if (len >= size_t(kHeaderBytes) && memcmp(p, "NBLIC0.3", 8) == 0) {
        const int n_channel = p[8];
        it.kind = 0; it.h = (p[9] << 8) | p[10]; it.w = (p[11] << 8) | p[12]; it.near = p[13]; it.k_step = p[14]; it.effort = p[15];
        return size_ok(it.h, it.w, max_px) && n_channel <= 1 && it.near <= kMaxNear && it.k_step >= kMinKStep && it.k_step <= kLevels &&
               it.effort >= 1 && it.effort <= 3;
    }
    if (len >= 8 && p[0] == 'Q' && p[1] == '0' && p[2] == '.' && p[3] == '2') {
        uint16_t q[4];
        memcpy(q, p, 8);
        it.kind = 1; it.h = q[2]; it.w = q[3]; it.near = it.effort = 0; it.k_step = kMinKStep;
        return size_ok(it.h, it.w, max_px);
    }
    return 0;
}

static size_t decode_state_bytes(const DecodeItem &it) { return up256(it.kind ? kQDecodeStateBytes : kDecodeStateBytes); }

// One launch round of a (codec, effort) class: every job advances by its `rows`.
static bool decode_launch(const DecodeItem &first, const SerialJob *d_jobs, const SerialJob *h_jobs, int n, hipStream_t st, bool whole_streams) {
    return first.kind == 1 ? serial_qdecode_launch(d_jobs, h_jobs, n, st) : serial_decode_launch(d_jobs, h_jobs, n, st, whole_streams);
}

static bool ensure_decode_space(nblic_amd_ctx *c, size_t arena, int m) {
    if (arena > c->dec_arena_cap) { hipFree(c->dec_arena); c->dec_arena = nullptr; c->dec_arena_cap = 0; HIP_OK(hipMalloc((void **)&c->dec_arena, arena)); c->dec_arena_cap = arena; }
    if (m > c->dec_jobs_cap) {
        hipFree(c->dec_jobs); c->dec_jobs = nullptr; c->dec_jobs_cap = 0;
        HIP_OK(hipMalloc((void **)&c->dec_jobs, size_t(m) * sizeof(SerialJob)));
        c->dec_jobs_cap = m;
    }
    return true;
}

// Parses and validates the headers, uploads the streams (their lengths are known here: running dry is an error),
// works every (codec, effort) class present through its launches -- `rows` rows of every image per launch, the
// state records carry the images from one launch to the next -- and copies the planes back.  status[k] = 0 / -1.
static bool decode_batch(nblic_amd_ctx *c, int n, const unsigned char *const *streams, const size_t *lens,
                         unsigned char *const *imgs, const size_t *img_caps, int *hs, int *ws, int *nears, int *efforts, int *status) {
    if (hipSetDevice(c->device) != hipSuccess) return false;
    std::vector<DecodeItem> items;
    std::vector<std::vector<uint8_t>> qtabs;                            // per QNBLIC item, alive until the copies have been made
    size_t arena = 0;
    for (int k = 0; k < n; k++) {
        status[k] = -1; hs[k] = ws[k] = 0; nears[k] = efforts[k] = 0;
        DecodeItem it{k, 0, 0, 0, 0, 0, 0, lens[k], -1, -1};
        if (!parse_stream_header(streams[k], lens[k], c->max_px, it)) continue;
        if (it.kind == 0 && lens[k] < size_t(kHeaderBytes) + 4) continue;
        hs[k] = it.h; ws[k] = it.w; nears[k] = it.near; efforts[k] = it.effort;
        if (size_t(it.h) * size_t(it.w) > img_caps[k]) continue;
        if (it.kind == 1) {                                              // QNBLIC: histogram tables parsed on the host; a stream whose tables
            std::vector<uint8_t> tab(kQTab);                             // do not parse is refused here and never reaches the GPU
            uint32_t *freq = reinterpret_cast<uint32_t *>(tab.data()), *start = freq + 12 * 256;
            int hh = 0, ww = 0;
            it.q_pos = q_decode_tables(reinterpret_cast<const uint16_t *>(streams[k]), lens[k] / 2, &hh, &ww, freq, start, nullptr);
            if (it.q_pos < 0 || size_t(it.q_pos) * 2 + 4 > lens[k]) continue;
            it.qtab = int(qtabs.size());
            qtabs.push_back(std::move(tab));
        }
        items.push_back(it);
        arena += up256(lens[k] + 2048) + up256(size_t(it.h) * size_t(it.w)) + up256(stats_doubles(it.effort, it.w) * sizeof(double)) +
                 decode_state_bytes(it) + (it.kind ? up256(kQTab) : 0);
    }
    if (items.empty()) return true;
    std::stable_sort(items.begin(), items.end(), [](const DecodeItem &a, const DecodeItem &b) { return a.kind * 4 + a.effort < b.kind * 4 + b.effort; });
    const int m = int(items.size());
    if (!ensure_decode_space(c, arena, m)) return false;
    std::vector<SerialJob> jobs(static_cast<size_t>(m));
    std::vector<SerialState> heads(static_cast<size_t>(m));
    std::vector<uint8_t *> d_streams(static_cast<size_t>(m)), d_tabs(static_cast<size_t>(m), nullptr);
    size_t off = 0;
    for (int i = 0; i < m; i++) {                                        // the arena's layout
        const DecodeItem &it = items[size_t(i)];
        SerialJob &J = jobs[size_t(i)];
        J = SerialJob{};
        d_streams[size_t(i)] = c->dec_arena + off; off += up256(it.len + 2048);
        J.recon = c->dec_arena + off; off += up256(size_t(it.h) * size_t(it.w));
        const size_t sb = stats_doubles(it.effort, it.w) * sizeof(double);
        if (sb) { J.stats = reinterpret_cast<double *>(c->dec_arena + off); off += up256(sb); }
        J.state = reinterpret_cast<SerialState *>(c->dec_arena + off); off += decode_state_bytes(it);
        J.stream = d_streams[size_t(i)];
        J.h = it.h; J.w = it.w; J.near = it.near; J.k_step = it.k_step; J.effort = it.effort;
        J.rows = serial_rows_per_launch(it.h, it.w, it.kind ? 1 : it.effort, c->serial_rows);
        SerialState &H = heads[size_t(i)];
        H = SerialState{};
        H.pos = it.kind ? (unsigned long long)(it.q_pos) * 2ull : (unsigned long long)(kHeaderBytes);
        H.avail = it.len; H.final_ = 1;
        if (it.kind == 1) {
            d_tabs[size_t(i)] = c->dec_arena + off; off += up256(kQTab);
            J.q_freq = reinterpret_cast<const uint32_t *>(d_tabs[size_t(i)]); J.q_start = J.q_freq + 12 * 256; J.q_slot = nullptr;
        }
    }
    // Chunks of one (codec, effort) class, at most kDecodeChunk images each, alternate between two streams: a chunk's uploads,
    // its launches (`rows` rows of every image per launch) and its copies back are all on ITS stream, and the host issues
    // upload + launches of chunk n before it waits for the planes of chunk n - 1 -- so one chunk computes while the other's
    // bytes cross the bus (the caller's memory is pageable: those copies hold the host thread, not the other stream).
    struct Chunk { int i0, i1; hipStream_t st; };
    std::vector<Chunk> chunks;
    for (int i0 = 0; i0 < m;) {
        int i1 = i0 + 1;
        while (i1 < m && i1 - i0 < kDecodeChunk && items[size_t(i1)].kind == items[size_t(i0)].kind && items[size_t(i1)].effort == items[size_t(i0)].effort) i1++;
        chunks.push_back(Chunk{i0, i1, (chunks.size() & 1) ? c->dec_stream2 : c->dec_stream});
        i0 = i1;
    }
    auto fail = [&](const char *what) {
        fprintf(stderr, "[nblic_amd] decode: %s failed\n", what);
        hipStreamSynchronize(c->dec_stream); hipStreamSynchronize(c->dec_stream2);
        return false;
    };
    auto upload_and_launch = [&](const Chunk &ch) {
        hipStream_t st = ch.st;
        for (int i = ch.i0; i < ch.i1; i++) {
            const DecodeItem &it = items[size_t(i)];
            const SerialJob &J = jobs[size_t(i)];
            const size_t sb = stats_doubles(it.effort, it.w) * sizeof(double);
            if (sb && hipMemsetAsync(J.stats, 0, sb, st) != hipSuccess) return false;
            if (hipMemsetAsync(d_streams[size_t(i)] + (it.len & ~size_t(3)), 0, 2048, st) != hipSuccess) return false;      // the window reads whole 512-byte blocks past the end
            if (hipMemcpyAsync(d_streams[size_t(i)], streams[it.k], it.len, hipMemcpyHostToDevice, st) != hipSuccess) return false;
            if (hipMemcpyAsync(J.state, &heads[size_t(i)], sizeof(SerialState), hipMemcpyHostToDevice, st) != hipSuccess) return false;
            if (it.kind == 1 && hipMemcpyAsync(d_tabs[size_t(i)], qtabs[size_t(it.qtab)].data(), kQTab, hipMemcpyHostToDevice, st) != hipSuccess) return false;
        }
        if (hipMemcpyAsync(c->dec_jobs + ch.i0, jobs.data() + ch.i0, size_t(ch.i1 - ch.i0) * sizeof(SerialJob), hipMemcpyHostToDevice, st) != hipSuccess) return false;
        int launches = 1;
        for (int i = ch.i0; i < ch.i1; i++) launches = std::max(launches, serial_launches(jobs[size_t(i)].h, jobs[size_t(i)].rows));
        for (int l = 0; l < launches; l++)
            if (!decode_launch(items[size_t(ch.i0)], c->dec_jobs + ch.i0, jobs.data() + ch.i0, ch.i1 - ch.i0, st, true)) return false;
        c->serial_launch_count += launches;
        return true;
    };
    auto download = [&](const Chunk &ch) {
        for (int i = ch.i0; i < ch.i1; i++) {
            const DecodeItem &it = items[size_t(i)];
            if (hipMemcpyAsync(&heads[size_t(i)], jobs[size_t(i)].state, sizeof(SerialState), hipMemcpyDeviceToHost, ch.st) != hipSuccess) return false;
            if (hipMemcpyAsync(imgs[it.k], jobs[size_t(i)].recon, size_t(it.h) * size_t(it.w), hipMemcpyDeviceToHost, ch.st) != hipSuccess) return false;
        }
        return true;
    };
    for (size_t n = 0; n < chunks.size(); n++) {
        if (n >= 2 && hipStreamSynchronize(chunks[n].st) != hipSuccess) return fail("a chunk");      // heads[] of chunk n - 2 have landed before its stream is reused (its H2D reads heads of chunk n)
        if (!upload_and_launch(chunks[n])) return fail("a launch");
        if (n >= 1 && !download(chunks[n - 1])) return fail("a copy");
    }
    if (!download(chunks.back())) return fail("a copy");
    if (hipStreamSynchronize(c->dec_stream) != hipSuccess || hipStreamSynchronize(c->dec_stream2) != hipSuccess) return fail("the last chunk");
    for (int i = 0; i < m; i++) status[items[size_t(i)].k] = heads[size_t(i)].status == kDone ? 0 : -1;
    return true;
}

// ---- the drop-in decoders: a stream whose length nobody tells us ------------------------------------
// The reference's decoders take no length (NBLIC.h:72, QNBLIC.h:16): they read what the encoder wrote, byte by byte.
// The shims fetch the stream in steps of `feed_chunk` bytes ON DEMAND -- the decoder stops in front of a row when it is
// about to run short (SerialState kStarved), the next step is copied in, it goes on -- so that no byte beyond what the
// decoder consumes plus one step is read from the caller's buffer.  Every step is copied by the KERNEL (write(2) into a
// pipe, read back): where the caller's memory ends (the next page unmapped or protected, a file mapping past its end)
// the copy comes back short instead of raising a signal, and a stream that sits right at the end of a mapping is read
// exactly to its last byte.
static size_t safe_copy(nblic_amd_ctx *c, void *dst, const void *src, size_t n) {
    if (c->feed_pipe[0] < 0) {
        if (pipe(c->feed_pipe) != 0) { c->feed_pipe[0] = c->feed_pipe[1] = -1; return 0; }
        // never block: nobody else reads this pipe, so a write that does not fit would wait for ever (a process over its
        // pipe quota gets single-page pipes)
        fcntl(c->feed_pipe[1], F_SETFL, fcntl(c->feed_pipe[1], F_GETFL) | O_NONBLOCK);
        fcntl(c->feed_pipe[0], F_SETFD, FD_CLOEXEC); fcntl(c->feed_pipe[1], F_SETFD, FD_CLOEXEC);
    }
    const size_t page = size_t(sysconf(_SC_PAGESIZE));
    const long pipe_cap = fcntl(c->feed_pipe[1], F_GETPIPE_SZ);
    const size_t burst = pipe_cap >= long(page) ? (size_t(pipe_cap) < size_t(65536) ? size_t(pipe_cap) & ~(page - 1) : size_t(65536)) : page;
    size_t done = 0;
    while (done < n) {
        // The pipe takes a write in page-sized pieces and DROPS a piece it could only copy in part, so the pieces have to
        // coincide with the source's pages: first the bytes up to the next page boundary, then whole pages (64 KB at a time:
        // what a fresh pipe holds, so the write never blocks).  A write that comes back short then ends exactly where the
        // readable memory ends.
        const size_t addr = size_t(reinterpret_cast<uintptr_t>(src)) + done;
        const size_t to_boundary = page - (addr & (page - 1));
        size_t want = (addr & (page - 1)) ? to_boundary : burst;               // what the pipe holds: a short return then means "memory ends", never "pipe full"
        if (want > n - done) want = n - done;
        const ssize_t k = write(c->feed_pipe[1], static_cast<const char *>(src) + done, want);
        if (k <= 0) break;                                                          // EFAULT: not one more byte can be read
        size_t got = 0;
        while (got < size_t(k)) {
            const ssize_t r = read(c->feed_pipe[0], static_cast<char *>(dst) + done + got, size_t(k) - got);
            if (r <= 0) return done + got;
            got += size_t(r);
        }
        done += size_t(k);
        if (size_t(k) < want) break;                                                // stopped at the end of the readable memory
    }
    return done;
}

// Decodes ONE stream that starts at p; *ph .. *peffort receive the header fields.  0 / -1.
static int decode_fed(nblic_amd_ctx *c, const unsigned char *p, bool qnblic, unsigned char *img, int *ph, int *pw, int *pnear, int *peffort) {
    if (hipSetDevice(c->device) != hipSuccess) return -1;
    c->fed_bytes = 0;
    unsigned char head[kHeaderBytes];
    const size_t head_want = qnblic ? 8 : size_t(kHeaderBytes);
    if (safe_copy(c, head, p, head_want) < head_want) return -1;
    DecodeItem it{0, 0, 0, 0, 0, 0, 0, head_want, -1, -1};
    if (!parse_stream_header(head, head_want, c->max_px, it) || (it.kind == 1) != qnblic) return -1;
    *ph = it.h; *pw = it.w;
    if (pnear) *pnear = it.near;
    if (peffort) *peffort = it.effort;
    const size_t npx = size_t(it.h) * size_t(it.w);
    // no valid stream of this geometry is longer (worst case seen: 1.0025 B/px + 20; QNBLIC: a word per pixel + tables)
    const size_t bound = qnblic ? 2 * npx + 32768 + 16 : npx + npx / 8 + 4096;
    const size_t sb = stats_doubles(it.effort, it.w) * sizeof(double);
    const size_t arena = up256(bound + 2048) + up256(npx) + up256(sb) + decode_state_bytes(it) + (qnblic ? up256(kQTab) : 0);
    if (!ensure_decode_space(c, arena, 1)) return -1;
    hipStream_t st = c->dec_stream;
    size_t off = 0;
    uint8_t *d_stream = c->dec_arena + off; off += up256(bound + 2048);
    SerialJob J{};
    J.recon = c->dec_arena + off; off += up256(npx);
    if (sb) { J.stats = reinterpret_cast<double *>(c->dec_arena + off); off += up256(sb); }
    J.state = reinterpret_cast<SerialState *>(c->dec_arena + off); off += decode_state_bytes(it);
    uint8_t *d_tab = qnblic ? c->dec_arena + off : nullptr;
    J.stream = d_stream;
    J.h = it.h; J.w = it.w; J.near = it.near; J.k_step = it.k_step; J.effort = it.effort;
    J.rows = serial_rows_per_launch(it.h, it.w, qnblic ? 1 : it.effort, c->serial_rows);
    if (qnblic) { J.q_freq = reinterpret_cast<const uint32_t *>(d_tab); J.q_start = J.q_freq + 12 * 256; J.q_slot = nullptr; }
    if (hipMemcpyAsync(c->dec_jobs, &J, sizeof J, hipMemcpyHostToDevice, st) != hipSuccess) return -1;

    std::vector<uint8_t> host;                                           // the stream as far as it has been fetched
    bool final_ = false;
    auto feed = [&](size_t want_total) -> bool {                         // extends `host` (and the device copy) to want_total bytes, or to where the memory ends
        if (want_total > bound) want_total = bound;
        const size_t have = host.size();
        if (want_total <= have) { if (have >= bound) final_ = true; return true; }
        host.resize(want_total);
        const size_t got = safe_copy(c, host.data() + have, p + have, want_total - have);
        host.resize(have + got);
        if (got < want_total - have || host.size() >= bound) final_ = true;
        c->fed_bytes = long(host.size());
        if (got == 0) return true;
        // (the kernel fetches whole 512-byte blocks beyond what is there; it never CONSUMES a byte at or beyond `avail`)
        return hipMemcpyAsync(d_stream + have, host.data() + have, got, hipMemcpyHostToDevice, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
    };
    const size_t chunk = c->feed_chunk < 4096 ? 4096 : c->feed_chunk;
    if (!feed(chunk)) return -1;
    SerialState H{};
    if (qnblic) {                                                        // the histogram tables sit in front of the rANS words: at most 12 x 256 codes
        std::vector<uint8_t> tab(kQTab);
        uint32_t *freq = reinterpret_cast<uint32_t *>(tab.data()), *start = freq + 12 * 256;
        int hh = 0, ww = 0;
        long pos = q_decode_tables(reinterpret_cast<const uint16_t *>(host.data()), host.size() / 2, &hh, &ww, freq, start, nullptr);
        if (pos < 0 && !final_ && host.size() < 65536) {                 // the tables may simply not be in hand yet
            if (!feed(65536)) return -1;
            pos = q_decode_tables(reinterpret_cast<const uint16_t *>(host.data()), host.size() / 2, &hh, &ww, freq, start, nullptr);
        }
        if (pos < 0) return -1;
        if (hipMemcpyAsync(d_tab, tab.data(), kQTab, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return -1;
        H.pos = (unsigned long long)(pos) * 2ull;
    } else {
        H.pos = kHeaderBytes;
    }
    const int launches = serial_launches(it.h, J.rows);
    for (int attempt = 0; attempt < 2; attempt++) {                      // the second attempt (whole stream in hand) only after kStarvedMidRow
        if (sb && hipMemsetAsync(J.stats, 0, sb, st) != hipSuccess) return -1;
        SerialState S = H;
        for (;;) {
            S.avail = host.size(); S.final_ = final_ ? 1 : 0; S.status = kRunning;
            // header fields the host owns are rewritten; on a resumed image the kernel's own fields come back unchanged
            if (hipMemcpyAsync(J.state, &S, sizeof S, hipMemcpyHostToDevice, st) != hipSuccess) return -1;
            for (int l = 0; l < launches; l++)                           // launches after a stop return at once
                if (!decode_launch(it, c->dec_jobs, &J, 1, st, false)) return -1;
            c->serial_launch_count += launches;
            if (hipMemcpyAsync(&S, J.state, sizeof S, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return -1;
            if (S.status == kStarved && !final_) { if (!feed(host.size() + chunk)) return -1; continue; }
            break;
        }
        if (S.status == kDone) {
            if (hipMemcpyAsync(img, J.recon, npx, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return -1;
            return 0;
        }
        if (S.status != kStarvedMidRow || final_) return -1;
        if (!feed(bound)) return -1;                                      // a row dearer than the margin allows: everything there is, from the top
        final_ = true;
    }
    return -1;
}

// ---- one image in ROW BANDS: bounded workspace, bounded launches, suspend and resume ----------------
// The serial modes' model stage is resumable row by row (serial_engine.h), and the entropy stages carry their
// adaptive state -- the 512 re-mappers, the 4096 counters -- in small per-image tables from launch to launch
// (kernels_e1.hip k_mapper_chains / k_counter_epochs read and write map_state / cnt_state).  So an image of any
// size can be worked through band by band: model stage for the band's rows -> re-mapper partition and chains,
// binarisation, counter partition, epochs, probabilities, mix for THOSE pixels -> the band's coded bins to the
// host -> the range coder, which is resumable too, carries on.  The device workspace is that of one band
// (config 5 of BASELINE.json, 268 Mpixel at effort 3, would need 32 GB in one piece), no kernel runs longer than
// a band, and between bands EVERYTHING the encoder carries is small enough to be written down: a checkpoint
// (model state record, the least-squares column statistics, the two tables, the coder interval, a running SHA-256
// of the bytes emitted so far) from which another call -- another process -- carries on.
struct BandCheckpoint {                 // followed by: model state record | B statistics | map_state | cnt_state | two reconstruction rows (if kept)
    char magic[8];                      // "NBLCKPT1"
    int h, w, near, effort, band_rows, next_row;
    uint32_t lo, hi;                    // coder interval (NBLIC.c:527-533)
    unsigned long long bytes_total;     // stream bytes emitted so far, header included
    Sha256 sha;                         // of exactly those bytes
    unsigned long long stats_bytes, recon_bytes;
};

}  // namespace nblic

struct nblic_amd_stream {
    nblic_amd_ctx *c = nullptr;
    int gid = -1;
    int h = 0, w = 0, near = 0, effort = 1, k_step = 3, band_rows = 1, next_row = 0;
    int first_row = 0;                                                  // the first row THIS object coded (> 0 after a resume)
    const uint8_t *d_img = nullptr; uint8_t *own_img = nullptr;        // the whole plane on the device
    uint8_t *d_recon = nullptr;                                         // whole reconstruction (near > 0 or rows too wide for LDS)
    double *d_stats = nullptr; size_t stats_bytes = 0;                  // [B | F], efforts 2 / 3
    uint16_t *d_coded = nullptr; size_t coded_cap = 0;                  // one band's coded bins
    uint16_t *h_coded = nullptr; size_t h_coded_cap = 0;                // the same, page-locked host memory
    uint32_t lo = 0, hi = 0xFFFFFFFFu;
    unsigned long long bytes_total = 0;
    nblic::Sha256 sha;
    bool finished = false, failed = false;
    long bands = 0;
    double model_ms = 0;
};

namespace nblic {

static void stream_free(nblic_amd_stream *s) {
    if (!s) return;
    if (s->c && hipSetDevice(s->c->device) == hipSuccess) {
        hipFree(s->own_img); hipFree(s->d_recon); hipFree(s->d_stats); hipFree(s->d_coded);
        locked_free(s->h_coded);
    }
    if (s->c && s->gid >= 0) release_group(s->c, s->gid);
    delete s;
}

static nblic_amd_stream *stream_open(nblic_amd_ctx *c, const unsigned char *img, bool on_device, int h, int w, int near, int effort, int band_rows) {
    if (!c || !img || !size_ok(h, w, c->max_px) || hipSetDevice(c->device) != hipSuccess) return nullptr;
    auto *s = new nblic_amd_stream;
    s->c = c; s->h = h; s->w = w; s->near = iclip(near, 0, kMaxNear); s->effort = iclip(effort, 1, 3); s->k_step = k_step_for_near(s->near);
    s->band_rows = band_rows > 0 ? (band_rows < h ? band_rows : h) : serial_rows_per_launch(h, w, s->effort, 0);
    {   // a group of the context for as long as the stream lives: its first slot's band workspace, its stream, its pinned job records
        std::unique_lock<std::mutex> l(c->fm);
        c->fcv.wait(l, [c] { return !c->free_groups.empty(); });
        s->gid = c->free_groups.front(); c->free_groups.pop_front();
    }
    Group &g = c->groups[size_t(s->gid)];
    const size_t n = size_t(h) * size_t(w);
    bool ok = true;
    if (on_device) s->d_img = img;
    else ok = hipMalloc((void **)&s->own_img, n) == hipSuccess && hipMemcpyAsync(s->own_img, img, n, hipMemcpyHostToDevice, g.stream) == hipSuccess && (s->d_img = s->own_img, true);
    if (ok && (s->near > 0 || !serial_model_rows_fit(w))) ok = hipMalloc((void **)&s->d_recon, n) == hipSuccess;
    s->stats_bytes = stats_doubles(s->effort, w) * sizeof(double);
    if (ok && s->stats_bytes) ok = hipMalloc((void **)&s->d_stats, s->stats_bytes) == hipSuccess && hipMemsetAsync(s->d_stats, 0, s->stats_bytes, g.stream) == hipSuccess;   // NBLIC.c:789
    Slot &sl = g.slots[0];
    ok = ok && ensure_pixels(sl, size_t(s->band_rows) * size_t(w));
    if (!ok) { fprintf(stderr, "[nblic_amd] stream: cannot set up the band workspace\n"); stream_free(s); return nullptr; }
    return s;
}

// the job records of the band that starts at row i0
static void stream_band_jobs(nblic_amd_stream *s, Group &g, int i0, int rows, uint32_t n_ev) {
    Slot &sl = g.slots[0];
    E1Job &J = g.h_jobs[0];
    J = E1Job{};
    J.b = sl.b; J.b.img = s->d_img + size_t(i0) * size_t(s->w); J.b.coded = s->d_coded;
    J.h = rows; J.w = s->w; J.n = uint32_t(size_t(rows) * size_t(s->w)); J.pp = make_plan(J.n);
    J.n_ev = n_ev; J.pe = make_plan(n_ev, kTouchSegments); J.dbg = 0;
    J.near = s->near; J.k_step = s->k_step; J.ktab = level_shift_table(s->k_step);
    SerialJob &Q = g.h_sjobs[0];
    Q = SerialJob{};
    Q.img = s->d_img; Q.recon = s->d_recon; Q.rec1 = sl.b.rec1; Q.pxs = sl.b.pxs; Q.stats = s->d_stats; Q.state = sl.d_state;
    Q.h = s->h; Q.w = s->w; Q.near = s->near; Q.k_step = s->k_step; Q.effort = s->effort; Q.rows = rows; Q.out_row0 = i0;
}

// Runs bands until the image is finished or the budget is spent.  1 finished, 0 suspended between two bands, -1 error.
static int stream_run(nblic_amd_stream *s, double budget_s, unsigned char *out, size_t cap, size_t *out_len) {
    *out_len = 0;
    if (!s || s->failed) return -1;
    if (s->finished) return 1;
    nblic_amd_ctx *c = s->c;
    if (hipSetDevice(c->device) != hipSuccess) return -1;
    Group &g = c->groups[size_t(s->gid)];
    Slot &sl = g.slots[0];
    const auto t0 = std::chrono::steady_clock::now();
    auto fail = [&](const char *what) { fprintf(stderr, "[nblic_amd] stream: %s\n", what); s->failed = true; hipStreamSynchronize(g.stream); return -1; };
    uint8_t *p = out;
    if (s->bytes_total == 0) {                                          // a fresh image: header, tables, state record
        if (cap < size_t(kHeaderBytes) + 4) return fail("output buffer too small");
        write_header(p, s->h, s->w, s->near, s->k_step, s->effort);
        p += kHeaderBytes;
        stream_band_jobs(s, g, 0, 1, 0);
        if (hipMemcpyAsync(g.d_jobs, g.h_jobs, sizeof(E1Job), hipMemcpyHostToDevice, g.stream) != hipSuccess) return fail("upload");
        e1_launch_init(g.d_jobs, 1, g.stream);
        if (hipMemsetAsync(sl.d_state, 0, sizeof(SerialState), g.stream) != hipSuccess) return fail("state");
    }
    RangeScalar rc;
    rc.begin(p, cap - size_t(p - out));
    rc.lo = s->lo; rc.hi = s->hi;
    while (s->next_row < s->h) {
        const int i0 = s->next_row, rows = std::min(s->band_rows, s->h - i0);
        stream_band_jobs(s, g, i0, rows, 0);
        hipEvent_t e0 = g.tm.ev[0], e1 = g.tm.ev[1];
        if (hipMemcpyAsync(g.d_jobs, g.h_jobs, sizeof(E1Job), hipMemcpyHostToDevice, g.stream) != hipSuccess ||
            hipMemcpyAsync(g.d_sjobs, g.h_sjobs, sizeof(SerialJob), hipMemcpyHostToDevice, g.stream) != hipSuccess) return fail("upload");
        hipEventRecord(e0, g.stream);
        if (!serial_model_launch(g.d_sjobs, g.h_sjobs, 1, g.stream)) return fail("model launch");
        hipEventRecord(e1, g.stream);
        e1_launch_front_pre(g.d_jobs, g.h_jobs, 1, g.stream);
        if (hipMemcpyAsync(g.h_totals, g.d_totals, kTotalsStride * sizeof(uint32_t), hipMemcpyDeviceToHost, g.stream) != hipSuccess ||
            hipStreamSynchronize(g.stream) != hipSuccess) return fail("front half");
        { float ms = 0.f; if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess) s->model_ms += ms; }
        const uint32_t n_ev = g.h_totals[2];
        if (n_ev >= 0x7FFFFFFFu || !ensure_events(sl, n_ev)) return fail("bin count");
        if (size_t(n_ev) + 8 > s->coded_cap) {
            hipFree(s->d_coded); s->d_coded = nullptr; locked_free(s->h_coded); s->h_coded = nullptr;
            s->coded_cap = size_t(n_ev) + size_t(n_ev) / 4 + 4096;
            if (hipMalloc((void **)&s->d_coded, s->coded_cap * sizeof(uint16_t)) != hipSuccess) return fail("coded bins");
            s->h_coded = locked_alloc(s->coded_cap);
            if (!s->h_coded) return fail("pinned bins");
        }
        stream_band_jobs(s, g, i0, rows, n_ev);
        if (hipMemcpyAsync(g.d_jobs, g.h_jobs, sizeof(E1Job), hipMemcpyHostToDevice, g.stream) != hipSuccess) return fail("upload");
        e1_launch_back(g.d_jobs, g.h_jobs, 1, g.stream, nullptr, true);
        if (n_ev && hipMemcpyAsync(s->h_coded, s->d_coded, size_t(n_ev) * sizeof(uint16_t), hipMemcpyDeviceToHost, g.stream) != hipSuccess) return fail("bins to the host");
        if (hipStreamSynchronize(g.stream) != hipSuccess) return fail("back half");
        rc.feed(s->h_coded, n_ev);
        if (rc.overflow) return fail("output buffer too small");
        s->next_row = i0 + rows; s->bands++;
        { std::lock_guard<std::mutex> l(c->stat_m); c->serial_launch_count++; }
        if (budget_s > 0 && s->next_row < s->h && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() >= budget_s) break;
    }
    s->lo = rc.lo; s->hi = rc.hi;
    size_t n = size_t(rc.p - out);
    if (s->next_row >= s->h) {
        const size_t body = rc.finish();                                 // the four flush bytes (NBLIC.c:576-586)
        if (body == SIZE_MAX) return fail("output buffer too small");
        n = size_t(rc.p - out);
        s->finished = true;
    }
    s->sha.update(out, n);
    s->bytes_total += n;
    *out_len = n;
    return s->finished ? 1 : 0;
}

static size_t stream_checkpoint_bytes(const nblic_amd_stream *s) {
    return sizeof(BandCheckpoint) + kModelStateBytes + s->stats_bytes / 2 + 512 * 60 * sizeof(int) + 4096 * 2 * sizeof(int) + (s->d_recon ? 2 * size_t(s->w) : 0);
}

static size_t stream_checkpoint(nblic_amd_stream *s, void *buf, size_t cap) {
    const size_t need = stream_checkpoint_bytes(s);
    if (!buf || cap < need) return need;
    if (s->failed || hipSetDevice(s->c->device) != hipSuccess) return 0;
    Group &g = s->c->groups[size_t(s->gid)];
    Slot &sl = g.slots[0];
    BandCheckpoint H{};
    memcpy(H.magic, "NBLCKPT1", 8);
    H.h = s->h; H.w = s->w; H.near = s->near; H.effort = s->effort; H.band_rows = s->band_rows; H.next_row = s->next_row;
    H.lo = s->lo; H.hi = s->hi; H.bytes_total = s->bytes_total; H.sha = s->sha;
    H.stats_bytes = s->stats_bytes / 2; H.recon_bytes = s->d_recon ? 2 * size_t(s->w) : 0;
    uint8_t *p = static_cast<uint8_t *>(buf);
    memcpy(p, &H, sizeof H); p += sizeof H;
    bool ok = hipMemcpy(p, sl.d_state, kModelStateBytes, hipMemcpyDeviceToHost) == hipSuccess; p += kModelStateBytes;
    if (H.stats_bytes) ok = ok && hipMemcpy(p, s->d_stats, H.stats_bytes, hipMemcpyDeviceToHost) == hipSuccess;       // the column statistics B; the row pre-pass F is recomputed
    p += H.stats_bytes;
    ok = ok && hipMemcpy(p, sl.b.map_state, 512 * 60 * sizeof(int), hipMemcpyDeviceToHost) == hipSuccess; p += 512 * 60 * sizeof(int);
    ok = ok && hipMemcpy(p, sl.b.cnt_state, 4096 * 2 * sizeof(int), hipMemcpyDeviceToHost) == hipSuccess; p += 4096 * 2 * sizeof(int);
    if (H.recon_bytes) {                                                  // the two rows above the next one (fewer at the top of the image: zeros)
        memset(p, 0, H.recon_bytes);
        const int r0 = s->next_row >= 2 ? s->next_row - 2 : 0, nr = s->next_row - r0;
        if (nr > 0) ok = ok && hipMemcpy(p + size_t(2 - nr) * size_t(s->w), s->d_recon + size_t(r0) * size_t(s->w), size_t(nr) * size_t(s->w), hipMemcpyDeviceToHost) == hipSuccess;
    }
    return ok ? need : 0;
}

static nblic_amd_stream *stream_resume(nblic_amd_ctx *c, const unsigned char *img, bool on_device, const void *ck, size_t ck_len) {
    if (!ck || ck_len < sizeof(BandCheckpoint)) return nullptr;
    BandCheckpoint H;
    memcpy(&H, ck, sizeof H);
    if (memcmp(H.magic, "NBLCKPT1", 8) != 0) return nullptr;
    nblic_amd_stream *s = stream_open(c, img, on_device, H.h, H.w, H.near, H.effort, H.band_rows);
    if (!s) return nullptr;
    if (ck_len != stream_checkpoint_bytes(s) || H.stats_bytes != s->stats_bytes / 2 || H.next_row < 0 || H.next_row > H.h) { stream_free(s); return nullptr; }
    Group &g = c->groups[size_t(s->gid)];
    Slot &sl = g.slots[0];
    s->next_row = s->first_row = H.next_row; s->lo = H.lo; s->hi = H.hi; s->bytes_total = H.bytes_total; s->sha = H.sha;
    s->finished = false;
    const uint8_t *p = static_cast<const uint8_t *>(ck) + sizeof H;
    bool ok = hipStreamSynchronize(g.stream) == hipSuccess;
    ok = ok && hipMemcpy(sl.d_state, p, kModelStateBytes, hipMemcpyHostToDevice) == hipSuccess; p += kModelStateBytes;
    if (H.stats_bytes) ok = ok && hipMemcpy(s->d_stats, p, H.stats_bytes, hipMemcpyHostToDevice) == hipSuccess;
    p += H.stats_bytes;
    ok = ok && hipMemcpy(sl.b.map_state, p, 512 * 60 * sizeof(int), hipMemcpyHostToDevice) == hipSuccess; p += 512 * 60 * sizeof(int);
    ok = ok && hipMemcpy(sl.b.cnt_state, p, 4096 * 2 * sizeof(int), hipMemcpyHostToDevice) == hipSuccess; p += 4096 * 2 * sizeof(int);
    if (H.recon_bytes && s->d_recon) {
        const int r0 = s->next_row >= 2 ? s->next_row - 2 : 0, nr = s->next_row - r0;
        if (nr > 0) ok = ok && hipMemcpy(s->d_recon + size_t(r0) * size_t(s->w), p + size_t(2 - nr) * size_t(s->w), size_t(nr) * size_t(s->w), hipMemcpyHostToDevice) == hipSuccess;
    }
    if (!ok || s->bytes_total == 0) { stream_free(s); return nullptr; }
    return s;
}

// ---- default context behind the drop-in entry points ---------------------------------------
static nblic_amd_ctx *g_default = nullptr;
static std::mutex g_default_m;

static nblic_amd_ctx *default_ctx() {
    std::lock_guard<std::mutex> g(g_default_m);
    if (!g_default) {
        const char *dev = getenv("NBLIC_AMD_DEVICE");
        g_default = nblic_amd_create(dev ? atoi(dev) : 0, 2, 2);
    }
    return g_default;
}

}  // namespace nblic

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

size_t nblic_amd_range_code(const uint16_t *coded, size_t n, unsigned char *out, size_t cap) {
    return range_code(coded, n, out, cap);
}

int nblic_amd_range_code_multi(const uint16_t *const *coded, const size_t *n, int count, unsigned char *const *outs,
                               const size_t *caps, size_t *lens) {
    if (count < 0) return -1;
    if (have_avx512()) {
        for (int k = 0; k < count; k += 16) range_code_x8(coded + k, n + k, count - k < 16 ? count - k : 16, outs + k, caps + k, lens + k);
        return 1;
    }
    for (int k = 0; k < count; k++) lens[k] = range_code(coded[k], n[k], outs[k], caps[k]);
    return 0;
}

int nblic_amd_range_code_chunked(const uint16_t *const *coded, const size_t *n, int count, unsigned char *const *outs,
                                 const size_t *caps, size_t *lens, size_t chunk) {
    if (count < 1 || count > kMaxTake || chunk == 0) return -1;
    size_t n_max = 0;
    for (int k = 0; k < count; k++) n_max = n[k] > n_max ? n[k] : n_max;
    const bool packs = count > 1 && have_avx512();
    if (!packs) {                                              // one after the other through the scalar coder
        for (int k = 0; k < count; k++) {
            RangeScalar r;
            r.begin(outs[k], caps[k]);
            for (size_t off = 0; off < n[k]; off += chunk) r.feed(coded[k] + off, n[k] - off < chunk ? n[k] - off : chunk);
            lens[k] = r.finish();
        }
        return 0;
    }
    // as in the coder threads: two packs up to sixteen streams, three beyond (dealt in order, as evenly as they go;
    // pack p owns lanes 8p ..), each chunk laid out as 13-bit groups (here on the host, in the pipeline by
    // k_pack_groups on the GPU) and fed through feed_pair_groups / feed_triple_groups
    RangeX8 x[3];
    const int n_packs = count > 16 ? 3 : 2;
    const size_t lanes = size_t(8 * n_packs);
    int pack_n[3] = {0, 0, 0}, pack_first[3] = {0, 0, 0};
    for (int p = 0, at = 0; p < n_packs; p++) { pack_n[p] = count / n_packs + (p < count % n_packs ? 1 : 0); pack_first[p] = at; at += pack_n[p]; }
    for (int p = 0; p < n_packs; p++) x[p].begin(pack_n[p], outs + pack_first[p], caps + pack_first[p]);
    const size_t rows_cap = group_words(chunk, lanes);
    uint64_t *rows = static_cast<uint64_t *>(aligned_alloc(64, (rows_cap * sizeof(uint64_t) + 63) & ~size_t(63)));
    if (!rows) return -1;
    for (size_t off = 0; off < n_max; off += chunk) {
        size_t len[kMaxTake] = {0};
        memset(rows, 0, rows_cap * sizeof(uint64_t));
        for (int k = 0; k < count; k++) {
            int p = 0;
            while (p + 1 < n_packs && k >= pack_first[p + 1]) p++;
            const int lane = 8 * p + (k - pack_first[p]);
            len[lane] = off >= n[k] ? 0 : (n[k] - off < chunk ? n[k] - off : chunk);
            pack_groups_host(rows, lane, coded[k] + off, len[lane], int(lanes));
        }
        if (n_packs == 3) feed_triple_groups(x[0], x[1], x[2], rows, len);
        else feed_pair_groups(x[0], x[1], rows, len);
    }
    free(rows);
    for (int p = 0; p < n_packs; p++) x[p].end(lens + pack_first[p]);
    return 0;
}

int nblic_amd_selftest(nblic_amd_ctx *c) {
    if (!c || hipSetDevice(c->device) != hipSuccess) return -1;
    return e1_selftest(c->groups[0].stream);
}

void nblic_amd_syn1(unsigned char *img, int h, int w, uint32_t seed) {
    uint32_t xs = seed;
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
            xs ^= xs << 13; xs ^= xs >> 17; xs ^= xs << 5;
            int t = ((i + 2 * j) >> 3) & 511;
            int base = iabs(t - 256); if (base > 255) base = 255;
            int v = ((base * 3) >> 2) + 32 + ((i ^ j) & 15) + int(xs & 7) + int((xs >> 3) & 7) - 7;
            img[size_t(i) * size_t(w) + size_t(j)] = uint8_t(iclip(v, 0, 255));
        }
}

const char *nblic_amd_version(void) { return "nblic_amd 0.2 (NBLIC v0.3 bitstream, gfx950)"; }

nblic_amd_ctx *nblic_amd_create_ex(int device, int n_groups, int group_size, int n_coders, int n_host_buffers) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        fprintf(stderr, "[nblic_amd] no HIP device available -- this library has no CPU fallback\n");
        return nullptr;
    }
    if (device < 0 || device >= count || hipSetDevice(device) != hipSuccess) {
        fprintf(stderr, "[nblic_amd] cannot select HIP device %d of %d\n", device, count);
        return nullptr;
    }
    if (n_groups < 1) n_groups = 1;
    if (group_size < 1) group_size = 1;
    if (n_coders < 1) n_coders = 1;
    if (n_host_buffers < n_groups * group_size + kMaxTake) n_host_buffers = n_groups * group_size + kMaxTake;   // groups in flight + one pack filling
    auto *c = new nblic_amd_ctx;
    c->device = device;
    c->simd = have_avx512() && !getenv("NBLIC_AMD_NO_SIMD");
    c->groups.resize(size_t(n_groups));
    for (int i = 0; i < n_groups; i++) {
        if (!group_init(c->groups[size_t(i)], i, group_size, c)) { nblic_amd_destroy(c); return nullptr; }
        c->free_groups.push_back(i);
    }
    c->cbufs.resize(size_t(n_host_buffers));
    for (int i = 0; i < n_host_buffers; i++) c->free_cbufs.push_back(i);
    if (hipStreamCreateWithFlags(&c->dec_stream, hipStreamNonBlocking) != hipSuccess) { c->dec_stream = nullptr; nblic_amd_destroy(c); return nullptr; }
    if (hipStreamCreateWithFlags(&c->dec_stream2, hipStreamNonBlocking) != hipSuccess) { c->dec_stream2 = nullptr; nblic_amd_destroy(c); return nullptr; }
    // (Measured and rejected: creating the copy streams with the highest stream priority, so that the
    // coder threads' short interleave kernels and copies overtake the encoder's long kernels -- the
    // pipeline drops from 4.9 to 3.1 Gpx/s.)
    if (const char *cb = getenv("NBLIC_AMD_CHUNK_BINS")) {
        const size_t v = size_t(atol(cb));
        if (v >= 4096 && v <= kChunkBins) c->chunk_bins = v & ~(kGroupBins - 1);     // chunks start on a group boundary
    }
    int n_copy = n_coders < kCopyStreams ? n_coders : kCopyStreams;
    if (const char *cs = getenv("NBLIC_AMD_COPY_STREAMS")) { const int v = atoi(cs); if (v >= 1 && v <= 32) n_copy = v; }   // experiments with the copy engines
    c->copy_streams.resize(size_t(n_copy));
    for (auto &cs : c->copy_streams)
        if (hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess) { cs = nullptr; nblic_amd_destroy(c); return nullptr; }
    c->coders_wanted = n_coders;
    if (const char *mt = getenv("NBLIC_AMD_MAX_TAKE")) { const int v = atoi(mt); if (v >= 2 && v <= kMaxTake) c->max_take = v; }
    for (int i = 0; i < n_coders; i++) c->coders.emplace_back(coder_main, c, i);
    for (int i = 0; i < n_groups; i++) c->drivers.emplace_back(driver_main, c, i);
    c->submitter = std::thread(submitter_main, c);
    return c;
}

nblic_amd_ctx *nblic_amd_create(int device, int n_slots, int n_coders) {
    // images in flight are split into groups that share kernel launches: two groups, so the GPU
    // works on one while the host codes the other
    if (n_slots < 1) n_slots = 1;
    int n_groups = n_slots >= 2 ? 2 : 1;
    return nblic_amd_create_ex(device, n_groups, (n_slots + n_groups - 1) / n_groups, n_coders, 2 * n_slots);
}

void nblic_amd_destroy(nblic_amd_ctx *c) {
    if (!c) return;
    hipSetDevice(c->device);
    { std::lock_guard<std::mutex> l(c->sm); c->stop_submit = true; }
    c->scv.notify_all();
    if (c->submitter.joinable()) c->submitter.join();
    { std::lock_guard<std::mutex> l(c->rm); c->stop = true; }
    c->rcv.notify_all();
    for (auto &t : c->coders) t.join();
    for (auto &t : c->dev_coders) t.join();
    { std::lock_guard<std::mutex> l(c->dm); c->stop_drivers = true; }
    c->dcv.notify_all();
    for (auto &t : c->drivers) t.join();
    for (auto &g : c->groups) group_free(g);
    for (auto &cb : c->cbufs) if (cb.p) hipFree(cb.p);
    for (auto &cs : c->copy_streams) if (cs) hipStreamDestroy(cs);
    hipFree(c->dec_arena); hipFree(c->dec_jobs);
    if (c->feed_pipe[0] >= 0) { close(c->feed_pipe[0]); close(c->feed_pipe[1]); }
    if (c->dec_stream) hipStreamDestroy(c->dec_stream);
    if (c->dec_stream2) hipStreamDestroy(c->dec_stream2);
    delete c;
}

void nblic_amd_set_max_pixels(nblic_amd_ctx *c, long max_pixels) {
    if (!c) c = default_ctx();                                               // NULL: the context behind the drop-in entry points
    if (c) c->max_px = max_pixels > 0 ? max_pixels : kMaxPixels;
}
void nblic_amd_set_serial_rows(nblic_amd_ctx *c, int rows) {
    if (!c) c = default_ctx();
    if (c) { std::lock_guard<std::mutex> g(c->api); c->serial_rows = rows > 0 ? rows : 0; }
}
long nblic_amd_serial_launches(nblic_amd_ctx *c) {
    if (!c) c = default_ctx();
    if (!c) return -1;
    std::lock_guard<std::mutex> l(c->stat_m);
    return c->serial_launch_count;
}
void nblic_amd_set_feed_chunk(nblic_amd_ctx *c, size_t bytes) {
    if (!c) c = default_ctx();
    if (c) { std::lock_guard<std::mutex> g(c->api); c->feed_chunk = bytes ? bytes : (size_t(1) << 20); }
}
long nblic_amd_last_fed_bytes(nblic_amd_ctx *c) {
    if (!c) c = default_ctx();
    return c ? c->fed_bytes : -1;
}
void nblic_amd_enable_timing(nblic_amd_ctx *c, int on) { c->timing = on != 0; c->timing_mask = on == 2 ? kRooflineStages : ~0ull; }

int nblic_amd_stage_times(nblic_amd_ctx *c, double *ms, const char **names, int cap) {
    int n = kE1Kernels;
    for (int k = 0; k < n && k < cap; k++) { ms[k] = c->stage_ms[k]; if (names) names[k] = kE1StageNames[k]; }
    return n < cap ? n : cap;
}

long nblic_amd_last_launches(nblic_amd_ctx *c) { return c->stage_launches; }

void nblic_amd_last_stats(nblic_amd_ctx *c, double *total_bins, double *coder_seconds_sum) {
    if (total_bins) *total_bins = c->total_bins;
    if (coder_seconds_sum) *coder_seconds_sum = c->coder_s;
}

int nblic_amd_encode_batch(nblic_amd_ctx *c, int n_images, const unsigned char *const *imgs, int imgs_on_device,
                           const int *heights, const int *widths, unsigned char *const *outs, const size_t *out_caps,
                           long *out_lens) {
    if (!c || n_images < 0) return -1;
    return encode_batch(c, n_images, imgs, imgs_on_device != 0, heights, widths, outs, out_caps, out_lens) ? 0 : -1;
}

nblic_amd_batch *nblic_amd_encode_batch_begin(nblic_amd_ctx *c, int n_images, const unsigned char *const *imgs, int imgs_on_device,
                                              const int *heights, const int *widths, unsigned char *const *outs,
                                              const size_t *out_caps, long *out_lens) {
    if (!c || n_images < 0 || hipSetDevice(c->device) != hipSuccess) return nullptr;
    auto *b = new nblic_amd_batch;
    queue_batch(c, nblic_amd_ctx::SubmitItem{b, n_images, imgs, imgs_on_device != 0, heights, widths, outs, out_caps, out_lens, nullptr, nullptr, nullptr});
    return b;
}

int nblic_amd_encode_batch_end(nblic_amd_ctx *c, nblic_amd_batch *b) {
    if (!c || !b) return -1;
    const bool ok = encode_wait(c, b) && !c->broken;         // b->ok and b->lens: this batch's images only
    report_coders(c);
    delete b;
    return ok ? 0 : -1;
}

long nblic_amd_debug_stage(nblic_amd_ctx *c, const unsigned char *img, int h, int w, int which, void *out, size_t out_bytes) {
    if (!c || !size_ok(h, w, c->max_px)) return -1;
    std::lock_guard<std::mutex> g(c->api);
    if (hipSetDevice(c->device) != hipSuccess) return -1;
    int id;
    { std::unique_lock<std::mutex> l(c->fm); c->fcv.wait(l, [c] { return !c->free_groups.empty(); }); id = c->free_groups.front(); c->free_groups.pop_front(); }
    Group &grp = c->groups[size_t(id)];
    Slot &s = grp.slots[0];
    grp.n_jobs = 1; s.job = 0; s.h = h; s.w = w; s.cb = -1; s.near = 0; s.effort = 1;
    const uint8_t *imgs[1] = {img};
    long count = -1;
    size_t n = size_t(h) * size_t(w);
    if (launch_front(c, grp, imgs, false) && launch_back(c, grp, false) && hipStreamSynchronize(grp.stream) == hipSuccess) {
        grp.tm_pending = false;
        const void *src = nullptr; size_t esz = 0, cnt = 0;
        switch (which) {
            case 0: src = s.b.rec1; esz = 4; cnt = n; break;
            case 1: src = s.b.pxs; esz = 2; cnt = n; break;
            case 2: src = s.b.z; esz = 1; cnt = n; break;
            case 3: src = s.b.cnt; esz = 1; cnt = n; break;
            case 4: src = s.b.events; esz = 4; cnt = s.n_ev; break;
            case 5: src = s.b.coded; esz = 2; cnt = s.n_ev; break;
            case 6: src = s.b.dbg_out; esz = 8; cnt = 4096; break;
            case 7: src = s.b.totals; esz = 4; cnt = size_t(kTotalsStride); break;       // [kWideTouchFlag]: 32-bit touch positions were needed
            default: break;
        }
        if (src && cnt * esz <= out_bytes && hipMemcpy(out, src, cnt * esz, hipMemcpyDeviceToHost) == hipSuccess) count = long(cnt);
    }
    if (s.cb >= 0) { std::lock_guard<std::mutex> l(c->fm); c->free_cbufs.push_back(s.cb); s.cb = -1; }
    release_group(c, id);
    return count;
}

int nblic_amd_encode_batch_modes(nblic_amd_ctx *c, int n_images, const unsigned char *const *imgs, int imgs_on_device,
                                 const int *heights, const int *widths, const int *nears, const int *efforts,
                                 unsigned char *const *outs, const size_t *out_caps, long *out_lens, unsigned char *const *recons) {
    if (!c || n_images < 0 || hipSetDevice(c->device) != hipSuccess) return -1;
    nblic_amd_batch b;
    queue_batch(c, nblic_amd_ctx::SubmitItem{&b, n_images, imgs, imgs_on_device != 0, heights, widths, outs, out_caps, out_lens, nears, efforts, recons});
    const bool ok = encode_wait(c, &b);
    return ok && !c->broken ? 0 : -1;
}

int nblic_amd_decode_batch(nblic_amd_ctx *c, int n_images, const unsigned char *const *streams, const size_t *stream_lens,
                           unsigned char *const *imgs, const size_t *img_caps, int *heights, int *widths, int *nears, int *efforts,
                           int *status) {
    if (!c || n_images < 0) return -1;
    std::lock_guard<std::mutex> g(c->api);
    if (!decode_batch(c, n_images, streams, stream_lens, imgs, img_caps, heights, widths, nears, efforts, status)) return -1;
    for (int k = 0; k < n_images; k++) if (status[k] != 0) return -1;
    return 0;
}

nblic_amd_stream *nblic_amd_stream_begin(nblic_amd_ctx *c, const unsigned char *img, int img_on_device, int height, int width, int near, int effort, int band_rows) {
    return stream_open(c, img, img_on_device != 0, height, width, near, effort, band_rows);
}
nblic_amd_stream *nblic_amd_stream_resume(nblic_amd_ctx *c, const unsigned char *img, int img_on_device, const void *checkpoint, size_t checkpoint_bytes) {
    return stream_resume(c, img, img_on_device != 0, checkpoint, checkpoint_bytes);
}
int nblic_amd_stream_run(nblic_amd_stream *s, double budget_seconds, unsigned char *out, size_t out_cap, size_t *out_len) {
    size_t n = 0;
    const int rc = stream_run(s, budget_seconds, out, out_cap, &n);
    if (out_len) *out_len = n;
    return rc;
}
size_t nblic_amd_stream_checkpoint(nblic_amd_stream *s, void *buf, size_t cap) { return s ? stream_checkpoint(s, buf, cap) : 0; }
int nblic_amd_stream_progress(nblic_amd_stream *s, int *rows_done, unsigned long long *bytes_total, unsigned char sha256[32], double *model_ms) {
    if (!s) return -1;
    if (rows_done) *rows_done = s->next_row;
    if (bytes_total) *bytes_total = s->bytes_total;
    if (sha256) s->sha.digest(sha256);
    if (model_ms) *model_ms = s->model_ms;
    return s->failed ? -1 : (s->finished ? 1 : 0);
}
int nblic_amd_stream_recon(nblic_amd_stream *s, unsigned char *plane, int *first_row, int *end_row) {
    if (!s || !plane || s->failed || hipSetDevice(s->c->device) != hipSuccess) return -1;
    if (first_row) *first_row = s->first_row;
    if (end_row) *end_row = s->next_row;
    const size_t at = size_t(s->first_row) * size_t(s->w), n = size_t(s->next_row - s->first_row) * size_t(s->w);
    if (n == 0) return 0;
    // lossless: the reconstruction IS the input (NBLIC.c:876 rewrites the same bytes)
    return hipMemcpy(plane + at, (s->near > 0 ? s->d_recon : s->d_img) + at, n, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
void nblic_amd_stream_end(nblic_amd_stream *s) { stream_free(s); }

int nblic_amd_set_device_coder(nblic_amd_ctx *c, int n_packs, int min_outstanding) {
    if (!c || n_packs < 0 || n_packs > 64) return -1;
    std::lock_guard<std::mutex> g(c->api);
    { std::lock_guard<std::mutex> l(c->rm); c->dev_min_outstanding = min_outstanding > 0 ? min_outstanding : 0; }
    while (int(c->dev_coders.size()) < n_packs) c->dev_coders.emplace_back(dev_coder_main, c, int(c->dev_coders.size()));
    return int(c->dev_coders.size());
}

void nblic_amd_device_coder_stats(nblic_amd_ctx *c, double *bins, long *packs, long *images) {
    std::lock_guard<std::mutex> l(c->stat_m);
    if (bins) *bins = c->dev_bins;
    if (packs) *packs = c->dev_packs;
    if (images) *images = c->dev_images;
}

int nblic_amd_serial_selftest(nblic_amd_ctx *c) {
    if (!c || hipSetDevice(c->device) != hipSuccess) return -1;
    return serial_selftest(c->dec_stream);
}

// ---- drop-in entry points --------------------------------------------------------------------
int NBLICcompress(int verbose, unsigned char *p_buf, unsigned char *p_img, int height, int width, int *p_near, int *p_effort) {
    (void)verbose;
    *p_near = iclip(*p_near, 0, kMaxNear);                                   // NBLIC.c:768
    *p_effort = iclip(*p_effort, 1, 3);                                      // NBLIC.c:770
    int k_step = k_step_for_near(*p_near);
    write_header(p_buf, height, width, *p_near, k_step, *p_effort);          // the reference writes it before validating
    nblic_amd_ctx *c = default_ctx();
    if (!c) return -1;
    if (!size_ok(height, width, c->max_px)) return -1;                       // NBLIC.h:31 unless nblic_amd_set_max_pixels(NULL, ...) raised it
    const bool serial_mode = !(*p_near == 0 && *p_effort == 1);
    if (serial_mode && size_t(height) * size_t(width) > (size_t(1) << 23)) {
        // a large image of a raster-serial mode: row bands (one band's workspace instead of 120 bytes per pixel of the
        // whole image, one model launch per band) -- the same bytes
        nblic_amd_stream *st = stream_open(c, p_img, false, height, width, *p_near, *p_effort, 0);
        if (!st) return -1;
        size_t n = 0;
        int rc = stream_run(st, 0.0, p_buf, size_t(1) << 46, &n);            // the reference ABI carries no capacity
        if (rc == 1 && *p_near > 0) { int r0 = 0, r1 = 0; rc = nblic_amd_stream_recon(st, p_img, &r0, &r1) == 0 ? 1 : -1; }   // NBLIC.c:876
        stream_free(st);
        return rc == 1 && n < (size_t(1) << 31) ? int(n) : -1;
    }
    const unsigned char *imgs[1] = {p_img};
    unsigned char *outs[1] = {p_buf}, *recons[1] = {p_img};
    size_t caps[1] = {SIZE_MAX};
    long lens[1] = {-1};
    if (nblic_amd_encode_batch_modes(c, 1, imgs, 0, &height, &width, p_near, p_effort, outs, caps, lens, recons) != 0) return -1;
    return int(lens[0]);
}

int NBLICdecompress(int verbose, unsigned char *p_buf, unsigned char *p_img, int *p_height, int *p_width, int *p_near, int *p_effort) {
    (void)verbose;
    nblic_amd_ctx *c = default_ctx();
    if (!c) return -1;
    std::lock_guard<std::mutex> g(c->api);
    return decode_fed(c, p_buf, false, p_img, p_height, p_width, p_near, p_effort);          // NBLIC.c:698-712, :924-926
}

int nblic_amd_qencode_batch(nblic_amd_ctx *c, int n_images, const unsigned char *const *imgs, int imgs_on_device,
                            const int *heights, const int *widths, uint16_t *const *outs, const size_t *out_caps_words,
                            long *out_len_words) {
    if (!c || n_images < 0) return -1;
    std::lock_guard<std::mutex> g(c->api);
    return encode_q_batch(c, n_images, imgs, imgs_on_device != 0, heights, widths, outs, out_caps_words, out_len_words) ? 0 : -1;
}

int QNBLICcompress(uint16_t *p_buf, unsigned char *p_img, int height, int width) {
    nblic_amd_ctx *c = default_ctx();
    if (!c) return -1;
    if (!size_ok(height, width, c->max_px)) return -1;                       // QNBLIC.c:575
    const unsigned char *imgs[1] = {p_img};
    uint16_t *outs[1] = {p_buf};
    size_t caps[1] = {SIZE_MAX / 4};                                          // the reference ABI carries no capacity
    long lens[1] = {-1};
    if (nblic_amd_qencode_batch(c, 1, imgs, 0, &height, &width, outs, caps, lens) != 0) return -1;
    return int(lens[0]);
}
int QNBLICdecompress(uint16_t *p_buf, unsigned char *p_img, int *p_height, int *p_width) {
    nblic_amd_ctx *c = default_ctx();
    if (!c) return -1;
    std::lock_guard<std::mutex> g(c->api);
    return decode_fed(c, reinterpret_cast<const unsigned char *>(p_buf), true, p_img, p_height, p_width, nullptr, nullptr);   // QNBLIC.c:475-555
}
int QNBLICcompressMultiThread(uint16_t *p_buf, unsigned char *p_img, int height, int width) {
    return QNBLICcompress(p_buf, p_img, height, width);                      // QNBLIC.c:872-883: same stream either way
}

}  // extern "C"

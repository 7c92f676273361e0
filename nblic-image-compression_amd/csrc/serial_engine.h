// serial_engine.h -- the raster-serial part of NBLIC on the GPU (serial_engine.hip).
//
// Every mode other than -n0 -e1 encode has a chain that runs pixel by pixel through the whole image
// (SURVEY.md section 0.4): near-lossless encode predicts from RECONSTRUCTED neighbours, efforts 2/3
// carry least-squares statistics and a global regularisation strength from pixel to pixel, and every
// decoder needs the previous pixel before it can decode the next.  What is serial differs, though:
//
//   encode (any near, any effort)   only prediction, context bias and quantisation are a chain; the
//       adaptive re-mappers, the binarisation, the counters and the range coder never feed back into
//       a pixel value.  k_serial_model runs that chain -- ONE WAVE PER IMAGE, hundreds of images side
//       by side -- and leaves per pixel the same records the staged -e1 front half leaves
//       (rec1, px | sign); the entropy stages then run on the key-partitioned kernels of
//       kernels_e1.hip and the host range coder, exactly as for -n0 -e1.
//   decode (NBLIC, any mode)        the whole loop is one chain: k_serial_decode, one wave per image,
//       model state (contexts, counters, re-mappers) in LDS.
//   decode (QNBLIC)                 k_serial_qdecode, one wave per image.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nblic {

// ---- resumable launches ---------------------------------------------------------------------------
// A serial kernel never has to run an image in one piece: everything its chain carries across a ROW boundary lives
// in a small per-image state record in device memory, a launch works on at most `rows` rows from the row the record
// names, and the next launch picks up there.  What crosses a row boundary:
//   model stage (encode)   the 2048 context biases; the least-squares regularisation strength `bias` (NBLIC.c:762 --
//       never reset); the column statistics B are in memory anyway (SerialJob::stats), the running row statistics E
//       are reset per row (NBLIC.c:818), the row pre-pass F is recomputed per row; the two rows above come back from
//       the reconstruction (or, lossless, from the input plane);
//   NBLIC decoder          the same plus the 4096 counters, the 512 re-mappers, the coder interval and its 4-byte
//       window, and the position in the stream;
//   QNBLIC decoder         the 3072 contexts, the rANS state and the position in the stream.
// Header of the record (SerialState), then the tables.  The host zeroes the header before the first launch.
struct SerialState {
    int next_row;              // first row the next launch works on (0: fresh image -- the launch initialises the tables itself)
    int status;                // kRunning / kDone / kFailed / kStarved / kStarvedMidRow
    unsigned long long pos;    // decoders: next stream byte to consume
    uint32_t lo, hi, window;   // decoders: coder interval and window (QNBLIC: lo = rANS state)
    int bias;                  // efforts 2/3
    unsigned long long avail;  // decoders, written by the HOST before a launch: stream bytes present in device memory
    int final_;                // decoders, written by the host: 1 = `avail` is the whole stream (running dry is an error), 0 = more may follow
    int pad[5];
};
static_assert(sizeof(SerialState) == 64, "header is sixteen words");
enum : int { kRunning = 0, kDone = 1, kFailed = -1, kStarved = 2, kStarvedMidRow = 3 };
// kStarved: a decoder of a stream that is still being fed stopped cleanly in front of row next_row because fewer than
// starve_margin(w) bytes were left -- feed more, set status back to kRunning, launch again.  kStarvedMidRow: it ran dry
// inside a row although the margin was there (a row that costs more than four bytes per pixel: never seen, possible
// for a damaged stream); the record is then NOT resumable and the image has to be decoded again with the whole stream.
constexpr size_t starve_margin(int w) { return size_t(4) * size_t(w) + 1024; }
constexpr size_t kModelStateBytes = sizeof(SerialState) + 2048 * sizeof(int);
constexpr size_t kDecodeStateBytes = sizeof(SerialState) + (2048 + 4096 + 512 * 20) * sizeof(int) + 2 * 512 * 20;
constexpr size_t kQDecodeStateBytes = sizeof(SerialState) + 3072 * sizeof(int);

// One image of a serial launch (array in device memory, job = blockIdx.x).
struct SerialJob {
    const uint8_t *img;        // encode: the plane to code (never written)
    uint8_t *recon;            // encode: reconstruction (may be null when near == 0 and the rows fit in LDS); decode: the decoded plane
    uint32_t *rec1;            // encode out: S1 record per pixel (model.h pack_s1)
    uint16_t *pxs;             // encode out: px | sign << 8 per pixel
    const uint8_t *stream;     // decode in: the .nblic stream (header included); how much of it is there is SerialState::avail
    double *stats;             // efforts 2/3: 2 * w * stats_stride(effort) doubles, zeroed (column sums, then the row pre-pass)
    SerialState *state;        // resumable state (above): kModelStateBytes / kDecodeStateBytes / kQDecodeStateBytes
    int h, w, near, k_step, effort;
    int rows;                  // rows per launch (>= 1)
    int out_row0;              // encode: the row whose records sit at index 0 of rec1 / pxs (0, or the first row of the band they hold)
    // QNBLIC decode only
    const uint32_t *q_freq, *q_start; const uint8_t *q_slot;      // 12 x 256 frequencies and cumulative starts; q_slot: unused (the kernel searches q_start)
};

constexpr int stats_stride(int effort) { return effort == 3 ? 128 : (effort == 2 ? 64 : 0); }   // doubles per pixel column per array
inline size_t stats_doubles(int effort, int w) { return size_t(2) * size_t(w) * size_t(stats_stride(effort)); }

// Rows one launch should cover so that it lasts a few seconds at most (a pixel of effort 2 / 3 costs about four / eight
// times a pixel of effort 1); `override_rows` > 0 (nblic_amd_set_serial_rows) wins.
inline int serial_rows_per_launch(int h, int w, int effort, int override_rows) {
    if (override_rows > 0) return override_rows < h ? override_rows : h;
    const long budget = (long(1) << 22) / (effort == 3 ? 8 : (effort == 2 ? 4 : 1));
    long rows = budget / (w > 0 ? w : 1);
    if (rows < 1) rows = 1;
    return int(rows < h ? rows : h);
}
inline int serial_launches(int h, int rows) { return (h + rows - 1) / rows; }

// true when the three rows a pixel's taps can touch fit in the LDS the model kernel has left, i.e. the kernel will keep
// them there; otherwise it reads its taps from SerialJob::recon, which then has to be there even for lossless jobs
bool serial_model_rows_fit(int w);

// d_jobs[0..n): all of one effort (1, 2 or 3); h_jobs: host copy, read to size the launch.  ONE launch: every job
// advances by its `rows`; call serial_launches(h, rows) times (maximum over the jobs) to finish them.
bool serial_model_launch(const SerialJob *d_jobs, const SerialJob *h_jobs, int n, hipStream_t s);
bool serial_decode_launch(const SerialJob *d_jobs, const SerialJob *h_jobs, int n, hipStream_t s, bool whole_streams);   // whole_streams: every job's stream is final (SerialState::final_)
bool serial_qdecode_launch(const SerialJob *d_jobs, const SerialJob *h_jobs, int n, hipStream_t s);
int serial_selftest(hipStream_t s);                    // device check of the double-carried divisions against 64-bit integers and of the half-wave exchange; 0 = pass

}  // namespace nblic

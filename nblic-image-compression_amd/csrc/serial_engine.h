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

// One image of a serial launch (array in device memory, job = blockIdx.x).
struct SerialJob {
    const uint8_t *img;        // encode: the plane to code (never written)
    uint8_t *recon;            // encode: reconstruction (may be null when near == 0 and the rows fit in LDS); decode: the decoded plane
    uint32_t *rec1;            // encode out: S1 record per pixel (model.h pack_s1)
    uint16_t *pxs;             // encode out: px | sign << 8 per pixel
    const uint8_t *stream;     // decode in: the .nblic stream (header included)
    size_t stream_len;
    double *stats;             // efforts 2/3: 2 * w * stats_stride(effort) doubles, zeroed (column sums, then the row pre-pass)
    int *status;               // decode out: 0 / -1 (stream exhausted)
    int h, w, near, k_step, effort;
    // QNBLIC decode only
    const uint32_t *q_freq, *q_start; const uint8_t *q_slot; size_t q_pos, q_words;
};

constexpr int stats_stride(int effort) { return effort == 3 ? 128 : (effort == 2 ? 64 : 0); }   // doubles per pixel column per array
inline size_t stats_doubles(int effort, int w) { return size_t(2) * size_t(w) * size_t(stats_stride(effort)); }

// d_jobs[0..n): all of one effort (1, 2 or 3); h_jobs: host copy, read to size the launch
bool serial_model_launch(const SerialJob *d_jobs, const SerialJob *h_jobs, int n, hipStream_t s);
bool serial_decode_launch(const SerialJob *d_jobs, const SerialJob *h_jobs, int n, hipStream_t s);
bool serial_qdecode_launch(const SerialJob *d_jobs, const SerialJob *h_jobs, int n, hipStream_t s);
int serial_selftest(hipStream_t s);                    // device check of the double-carried divisions against 64-bit integers; 0 = pass

}  // namespace nblic

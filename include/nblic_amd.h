/*
 * nblic_amd.h -- C ABI of libnblic_amd.so, the MI355X-native drop-in for the NBLIC v0.3
 * codec entry points.  Plain pointers and sizes only; no torch / HIP types.
 *
 * Section 1 re-declares the reference's own API with identical names, argument meaning and
 * return conventions, so a caller of the reference library (its only caller is main(),
 * src/NBLIC_main.c:184-188,223-226) links against this library unchanged.
 * Section 2 is additive: a context + batch interface that keeps several images in flight per
 * GPU (the reference has no equivalent; it is what bench.py and the Python host layer use).
 *
 * The compute path is HIP only.  If no gfx950 device is usable every entry point that has to
 * compute returns -1 after printing a diagnostic to stderr -- there is no CPU fallback.
 */
#ifndef NBLIC_AMD_H
#define NBLIC_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* limits, same values as the reference (src/NBLIC.h:29-31, src/QNBLIC.h:9-11) */
#define NBLIC_MAX_HEIGHT     65535
#define NBLIC_MAX_WIDTH      65535
#define NBLIC_MAX_IMG_SIZE   100000000
#define QNBLIC_MAX_HEIGHT    65535
#define QNBLIC_MAX_WIDTH     65535
#define QNBLIC_MAX_IMG_SIZE  100000000

/* ---- 1. drop-in entry points -------------------------------------------------------- */

/* Replaces NBLICcompress (src/NBLIC.h:54, src/NBLIC.c:915).
 * verbose  : accepted, ignored on the GPU path (the reference prints row progress).
 * p_buf    : caller-owned output, no capacity argument (reference contract); worst case seen
 *            is ~1.0025 B/px + 20, the reference CLI provides 2 B/px.
 * p_img    : 8-bit gray, row-major, stride == width.  WRITTEN: receives the reconstruction
 *            (identical bytes when *p_near == 0), as the reference does at NBLIC.c:876.
 * p_near   : clamped to [0,9] and written back.   p_effort : clamped to [1,3], written back.
 * returns  : stream length in BYTES, or -1.                                              */
int NBLICcompress(int verbose, unsigned char *p_buf, unsigned char *p_img, int height, int width,
                  int *p_near, int *p_effort);

/* Replaces NBLICdecompress (src/NBLIC.h:72, src/NBLIC.c:924).  All four int outputs are
 * parsed from the 16-byte header.  returns 0 or -1.                                       */
int NBLICdecompress(int verbose, unsigned char *p_buf, unsigned char *p_img, int *p_height, int *p_width,
                    int *p_near, int *p_effort);

/* Replace QNBLICcompress / QNBLICdecompress / QNBLICcompressMultiThread (src/QNBLIC.h:14-18,
 * src/QNBLIC.c:562,493,872).  Effort 0, lossless only.  compress returns the length in
 * 16-bit WORDS (the reference's caller doubles it, NBLIC_main.c:184-186) or -1.           */
int QNBLICcompress(uint16_t *p_buf, unsigned char *p_img, int height, int width);
int QNBLICdecompress(uint16_t *p_buf, unsigned char *p_img, int *p_height, int *p_width);
int QNBLICcompressMultiThread(uint16_t *p_buf, unsigned char *p_img, int height, int width);

/* ---- 2. additive context / batch API ------------------------------------------------- */

typedef struct nblic_amd_ctx nblic_amd_ctx;

/* device   : HIP device ordinal.
 * n_slots  : images kept in flight on the GPU (each owns a stream and a workspace); >= 1.
 * n_coders : host threads running the serial range-coder stage; >= 1.
 * returns NULL (and prints why) when the device cannot be used.                            */
nblic_amd_ctx *nblic_amd_create(int device, int n_slots, int n_coders);

/* Same, with the split of the images in flight spelled out: n_groups groups of group_size
 * images.  A group shares every kernel launch (its serial chains run side by side); while the
 * host codes one group the GPU works on the next.  n_host_buffers (raised to at least
 * n_groups * group_size + 16) buffers IN HBM hold coded-bin streams waiting for a coder thread,
 * so the device workspace of an image is free again as soon as its kernels have finished; the
 * coder threads stream the bins to the host through a small pinned ring of their own.  (The
 * parameter keeps its name from when these buffers were pinned host memory.)
 * nblic_amd_create uses two groups and 2 * n_slots buffers.                                   */
nblic_amd_ctx *nblic_amd_create_ex(int device, int n_groups, int group_size, int n_coders, int n_host_buffers);
void nblic_amd_destroy(nblic_amd_ctx *ctx);

/* Encode n_images gray planes at -n0 -e1 (lossless) into byte-exact .nblic streams.
 * imgs[k]        : plane k, heights[k] x widths[k], stride == width.
 * imgs_on_device : 0 = host pointers, 1 = device (HBM) pointers on the context's device.
 * outs[k]        : host buffer for stream k, out_caps[k] bytes (checked; -1 if too small).
 * out_lens[k]    : receives the stream length in bytes, or -1 for that image.
 * returns 0 when every image succeeded, -1 otherwise.                                      */
int nblic_amd_encode_batch(nblic_amd_ctx *ctx, int n_images, const unsigned char *const *imgs, int imgs_on_device,
                           const int *heights, const int *widths, unsigned char *const *outs,
                           const size_t *out_caps, long *out_lens);

/* The same batch in two halves, so that several batches can be in flight: _begin only QUEUES the
 * batch and returns at once -- a submitter thread of the context hands its images to the GPU
 * pipeline later (waiting while every group is busy), so the argument arrays are read AFTER _begin
 * has returned; _end waits until every stream of THAT batch has been written and returns 0 / -1
 * like nblic_amd_encode_batch: -1 when one of THIS batch's images failed (a failure in another
 * batch in flight does not show here) or when the context itself is unusable.  While batch k
 * drains through the host coder threads (~0.8 s for the last packs) batch k+1 is already filling
 * the GPU: a continuous feed never sees the pipeline's fill and drain.  All argument arrays and
 * buffers of a batch must stay valid, and its outputs untouched, until its _end; batches may be
 * ended in any order.                                                                          */
typedef struct nblic_amd_batch nblic_amd_batch;
nblic_amd_batch *nblic_amd_encode_batch_begin(nblic_amd_ctx *ctx, int n_images, const unsigned char *const *imgs,
                                              int imgs_on_device, const int *heights, const int *widths,
                                              unsigned char *const *outs, const size_t *out_caps, long *out_lens);
int nblic_amd_encode_batch_end(nblic_amd_ctx *ctx, nblic_amd_batch *batch);

/* Same for effort 0 (QNBLIC): the per-pixel model runs on the GPU, the entropy stage (histogram
 * normalisation, histogram code, rANS) on a coder thread.  outs[k] are uint16_t buffers; capacities
 * and lengths are in 16-bit WORDS, like QNBLICcompress's return value.                         */
int nblic_amd_qencode_batch(nblic_amd_ctx *ctx, int n_images, const unsigned char *const *imgs, int imgs_on_device,
                            const int *heights, const int *widths, uint16_t *const *outs, const size_t *out_caps_words,
                            long *out_len_words);

/* Any mode, many images: the batch form of NBLICcompress (src/NBLIC.h:54, src/NBLIC.c:749-908, :915).
 * nears[k] / efforts[k] : image k's -n / -e, clamped to [0,9] / [1,3] exactly as the reference clamps them
 *                  (either array may be NULL: 0 / 1).  -n0 -e1 images take the staged pipeline; every other
 *                  mode is raster-serial in its prediction (reconstructed neighbours, least-squares
 *                  statistics), so its model stage runs ONE WAVE PER IMAGE with all images of the batch side
 *                  by side -- size the context (n_groups x group_size) for the number of images you want in
 *                  flight -- and its entropy stages on the same parallel kernels and host coder threads.
 * recons        : NULL, or per image NULL / a host buffer of h*w bytes that receives the reconstruction the
 *                  reference leaves in p_img (NBLIC.c:876); may be the (host) input plane itself.
 * Everything else as nblic_amd_encode_batch.  returns 0 / -1.                                        */
int nblic_amd_encode_batch_modes(nblic_amd_ctx *ctx, int n_images, const unsigned char *const *imgs, int imgs_on_device,
                                 const int *heights, const int *widths, const int *nears, const int *efforts,
                                 unsigned char *const *outs, const size_t *out_caps, long *out_lens,
                                 unsigned char *const *recons);

/* The batch form of NBLICdecompress / QNBLICdecompress (src/NBLIC.h:72, src/QNBLIC.h:16; the codec is told
 * from the magic like src/NBLIC_main.c:223-226 does): n_images streams decoded side by side, one wave each.
 * streams[k] / stream_lens[k] : the stream and its length in BYTES (unlike the reference ABI, which takes none).
 * imgs[k] / img_caps[k]       : host buffer for the plane and its size in bytes.
 * heights / widths / nears / efforts : outputs parsed from the headers (QNBLIC: near = effort = 0).
 * status[k]                   : 0, or -1 for that stream (bad header, plane too large, stream exhausted).
 * returns 0 when every stream decoded, -1 otherwise.                                                 */
int nblic_amd_decode_batch(nblic_amd_ctx *ctx, int n_images, const unsigned char *const *streams, const size_t *stream_lens,
                           unsigned char *const *imgs, const size_t *img_caps, int *heights, int *widths, int *nears,
                           int *efforts, int *status);

/* Opt-in: the range-coder stage (src/NBLIC.c:552-586) on the GPU as a SUPPLEMENT to the host coder threads.
 * n_packs pack threads are started (they sleep unless there is work); each hands 64 queued images at a time to
 * one wave of the GPU, ONE LANE PER IMAGE (~17 Mbins/s per image: a 4096^2 frame takes seconds, but 64 of them
 * take the same seconds and none of their bins cross PCIe).  A pack is taken only when the queue of finished
 * images holds 64 more than the host threads can take at once, and only while at least min_outstanding images
 * of the submitted batches are unfinished -- set it to (pack latency x the host threads' image rate) so that a
 * pack can never become the tail of the work.  The streams are byte-identical either way.  Returns the number
 * of pack threads, or -1.  Size n_host_buffers (nblic_amd_create_ex) 64 x n_packs larger than without.       */
int nblic_amd_set_device_coder(nblic_amd_ctx *ctx, int n_packs, int min_outstanding);
/* What the device coder did since the context was last idle: bins coded, packs launched, images coded. */
void nblic_amd_device_coder_stats(nblic_amd_ctx *ctx, double *bins, long *packs, long *images);

/* Opt-in: raise the pixel-count limit above NBLIC_MAX_IMG_SIZE for this context (config 5 of
 * BASELINE.json exceeds the reference's own limit, src/NBLIC.h:31).  0 restores the reference limit.
 * ctx == NULL addresses the context behind the drop-in entry points of section 1.                   */
void nblic_amd_set_max_pixels(nblic_amd_ctx *ctx, long max_pixels);

/* The raster-serial kernels (the model stage of near > 0 / efforts 2, 3 encodes, every decoder) are RESUMABLE:
 * whatever their chain carries across a row boundary lives in a per-image state record on the device, a launch works
 * on at most `rows` rows of every image, and the next launch picks up where it stopped -- so no kernel runs longer
 * than a few seconds however large the image (config 5 of BASELINE.json is a 268 Mpixel frame at effort 3).
 * rows > 0 fixes the rows per launch (tests use it to force many resumptions); 0 = sized automatically.
 * ctx == NULL addresses the context behind the drop-in entry points.  nblic_amd_serial_launches: launches of the
 * serial kernels since the context was created.                                                              */
void nblic_amd_set_serial_rows(nblic_amd_ctx *ctx, int rows);
long nblic_amd_serial_launches(nblic_amd_ctx *ctx);

/* ONE image of any mode, worked through in ROW BANDS (src/NBLIC.c:749-908 is one loop over the rows; every piece of
 * state it carries from row to row is small).  Per band: the model stage for the band's rows, the entropy stages for
 * those pixels (their adaptive tables carried from band to band), the band's bins through the range coder.  The device
 * workspace is one band's whatever the image size, no kernel runs longer than a band, and between two bands the
 * encoder can be SUSPENDED: nblic_amd_stream_checkpoint writes down everything it carries (a few hundred KB plus
 * 8 * width * (1 + n + n^2) bytes of least-squares statistics at efforts 2 / 3), nblic_amd_stream_resume -- in
 * another call, another process, on another GPU -- carries on from there.  The stream's bytes are identical to
 * NBLICcompress's.  A running SHA-256 of the bytes emitted travels with the checkpoint, so a run that never holds the
 * whole stream in one place can still be checked against a golden hash.
 *   _begin    img: the whole plane (host or device; it must stay valid until _end).  band_rows <= 0: sized automatically.
 *             Takes one group of the context until _end.  NULL on failure.
 *   _run      codes bands until the image is finished (returns 1) or budget_seconds (> 0) have passed (returns 0); the
 *             stream bytes produced by THIS call are written to out (out_cap is checked, -1 if too small) and counted
 *             in *out_len: concatenate the pieces of successive calls.
 *   _progress rows finished, bytes emitted so far, their SHA-256, milliseconds spent in the model kernel; returns 1 / 0 / -1.
 *   _checkpoint  writes the checkpoint into buf (cap bytes) and returns its size; with buf == NULL or cap too small it
 *             only returns the size needed.  Valid between two _run calls.
 *   _recon    the reconstruction the reference leaves in p_img (NBLIC.c:876), as far as THIS object has produced it:
 *             rows [*first_row, *end_row) are written at their place in `plane` (a whole h x w plane); an object resumed
 *             from a checkpoint starts at the checkpoint's row, the rows before it came out of the earlier objects.   */
typedef struct nblic_amd_stream nblic_amd_stream;
nblic_amd_stream *nblic_amd_stream_begin(nblic_amd_ctx *ctx, const unsigned char *img, int img_on_device, int height, int width,
                                         int near, int effort, int band_rows);
nblic_amd_stream *nblic_amd_stream_resume(nblic_amd_ctx *ctx, const unsigned char *img, int img_on_device, const void *checkpoint,
                                          size_t checkpoint_bytes);
int nblic_amd_stream_run(nblic_amd_stream *s, double budget_seconds, unsigned char *out, size_t out_cap, size_t *out_len);
size_t nblic_amd_stream_checkpoint(nblic_amd_stream *s, void *buf, size_t cap);
int nblic_amd_stream_progress(nblic_amd_stream *s, int *rows_done, unsigned long long *bytes_total, unsigned char sha256[32], double *model_ms);
int nblic_amd_stream_recon(nblic_amd_stream *s, unsigned char *plane, int *first_row, int *end_row);
void nblic_amd_stream_end(nblic_amd_stream *s);

/* The reference's decoders take no stream length (src/NBLIC.h:72, src/QNBLIC.h:16).  NBLICdecompress / QNBLICdecompress
 * therefore fetch the caller's stream ON DEMAND in steps of `bytes` (default 1 MiB, at least 4096): the decoder stops in
 * front of a row when it is about to run short, the next step is copied in, it resumes.  No byte beyond the last one the
 * decoder consumes plus one step is read, and each step is copied by the kernel (a pipe write), so a stream that ends
 * right in front of an unmapped page is read exactly to its end instead of faulting.  ctx == NULL: the drop-in context.
 * nblic_amd_last_fed_bytes: how many bytes the last drop-in decode read from the caller's buffer.                      */
void nblic_amd_set_feed_chunk(nblic_amd_ctx *ctx, size_t bytes);
long nblic_amd_last_fed_bytes(nblic_amd_ctx *ctx);

/* Per-kernel device times of the LAST nblic_amd_encode_batch: one HIP event in front of every
 * launch, on the stream the kernel runs on, summed over the batch's group launches (divide by
 * nblic_amd_last_launches() for the average launch duration).  Writes up to `cap` entries
 * of milliseconds into ms[] and matching static strings into names[]; returns the count.
 * Timing is recorded only after nblic_amd_enable_timing(ctx, 1) (every stage) or (ctx, 2): only
 * k_predict and k_touch_scatter, the two stages the bench line reports against a roof -- thirty-two
 * events per group launch cost the pipeline 2 % of its throughput, four do not; untimed stages read 0. */
void nblic_amd_enable_timing(nblic_amd_ctx *ctx, int on);
int nblic_amd_stage_times(nblic_amd_ctx *ctx, double *ms, const char **names, int cap);

/* How many group launches the last batch took (each kernel of the sequence is launched once
 * per group of images); the entries of nblic_amd_stage_times() are sums over these launches. */
long nblic_amd_last_launches(nblic_amd_ctx *ctx);

/* Bins coded / host range-coder seconds summed over the last batch (for reporting). */
void nblic_amd_last_stats(nblic_amd_ctx *ctx, double *total_bins, double *coder_seconds_sum);

/* Stage-level debug hook used by the parity tests: runs the staged -e1 pipeline on ONE host
 * image and copies the named intermediate array back.  which: 0 rec1(u32) 1 pxs(u16) 2 z(u8)
 * 3 cnt(u8) 4 events(u32) 5 coded(u16).  Returns the element count, or -1.                 */
long nblic_amd_debug_stage(nblic_amd_ctx *ctx, const unsigned char *img, int height, int width, int which,
                           void *out, size_t out_bytes);

/* Device self-test of the wave primitives the chain kernels rely on (DPP prefix sum against the
 * shuffle formulation).  Returns the number of mismatching lanes (0 = pass) or -1.           */
int nblic_amd_selftest(nblic_amd_ctx *ctx);

/* The host half of the path on its own: the serial range-coder stage (src/NBLIC.c:552-586) over
 * n coded bins (u16 each: probability of a 1 in 1/4096 in bits 0-11, the bin in bit 15).
 * Writes at most cap bytes (coder bytes + 4 flush bytes, no header); returns the byte count or
 * (size_t)-1 when cap is too small.  Needs no GPU.                                         */
size_t nblic_amd_range_code(const uint16_t *coded, size_t n, unsigned char *out, size_t cap);

/* Synthetic benchmark frame "SYN-1" (SURVEY.md 8d): xorshift32 noise on a triangular ramp with a
 * 16-level texture; deterministic, integer only.  Host function, needs no GPU.              */
void nblic_amd_syn1(unsigned char *img, int height, int width, uint32_t seed);

/* Same stage for `count` independent streams.  On hosts with AVX-512 eight streams are coded at
 * once in the eight 64-bit lanes of a vector register (returns 1), otherwise one after the other
 * (returns 0); the bytes are identical either way.  lens[k] = byte count or (size_t)-1.       */
int nblic_amd_range_code_multi(const uint16_t *const *coded, const size_t *n, int count, unsigned char *const *outs,
                               const size_t *caps, size_t *lens);

/* The same streams fed `chunk` bins at a time through the RESUMABLE coders, exactly as the coder
 * threads do when they stream an image's bins from HBM (one stream: scalar coder; more: two
 * AVX-512 packs in lock-step, up to 16 streams).  Host function; exists so that the chunked path
 * can be checked without a GPU.  Returns 0, or -1 for count outside 1..16 or chunk == 0.        */
int nblic_amd_range_code_chunked(const uint16_t *const *coded, const size_t *n, int count, unsigned char *const *outs,
                                 const size_t *caps, size_t *lens, size_t chunk);

/* ---- 3. front end: image files and the reference tool's command line (host only) ----------------- */

/* The reference tool's main() (src/NBLIC_main.c:139-254) on this library: same switch grammar (groups
 * such as "-cn2e2V"), same PGM-then-BMP probing of the input (:168-169), -n0 -e0 -> QNBLIC with the word
 * count doubled (:182-188), QNBLIC-then-NBLIC decoding (:223-226), ".bmp" suffix -> BMP output (:233).
 * Returns 0, or -1 after printing an "***Error" line.  The `nblic_codec_amd` executable is this call.  */
int nblic_amd_cli_main(int argc, char **argv);

/* The parsed command line (tests of the grammar): fields[0..7] = decompress, near, effort, verbose,
 * multithread, large-image opt-in (-L), device (-g<N>, -1 = unset), have_src | have_dst << 1.            */
void nblic_amd_cli_parse(int argc, char **argv, int *fields, char *src, char *dst, size_t cap);

/* Reads an 8-bit gray image the way the reference's front end does (src/FileIO.c:81-131 binary PGM, then
 * :170-225 8-bit BMP: bottom-up rows, 4-byte row padding, pixel = palette index).  Writes h*w bytes (row
 * major, top-down) to px.  Returns 1 = PGM, 2 = BMP, 0 = neither, -1 = cap too small (*h, *w still set). */
int nblic_amd_read_gray(const char *path, unsigned char *px, size_t cap, int *h, int *w);

/* Writes "P5\n<w> <h>\n255\n" + pixels (src/FileIO.c:141-159), or with as_bmp != 0 the 1078-byte header
 * (identity gray palette) + bottom-up padded rows (src/FileIO.c:229-287).  Returns 0 / -1.               */
int nblic_amd_write_gray(const char *path, const unsigned char *px, int h, int w, int as_bmp);

/* Device self-test of the serial kernels' arithmetic: the double-carried truncating divisions of the
 * least-squares predictor against 64-bit integer division on 65536 operand triples.  0 = pass.   */
int nblic_amd_serial_selftest(nblic_amd_ctx *ctx);

const char *nblic_amd_version(void);

#ifdef __cplusplus
}
#endif
#endif /* NBLIC_AMD_H */

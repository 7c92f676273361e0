// serial_engine.h -- raster-serial NBLIC engine on the GPU (serial_engine.hip).
//
// Every mode other than -n0 -e1 encode is strictly raster-serial: the
// decoders, near-lossless encode (neighbours are reconstructed values) and the least-squares
// efforts 2/3.  This engine runs one image per workgroup with the whole adaptive model in LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nblic {

struct SerialEngine {
    uint8_t *d_img = nullptr;      size_t img_cap = 0;
    uint8_t *d_stream = nullptr;   size_t stream_cap = 0;
    int64_t *d_stats = nullptr;    size_t stats_cap = 0;     // least-squares row statistics, 2*w*m
    long    *d_len = nullptr;
    hipStream_t stream = nullptr;

    bool init();
    void destroy();
    // returns stream length in bytes (header included) or -1; img receives the reconstruction
    long encode(uint8_t *out, uint8_t *img, int h, int w, int near, int k_step, int effort, int device);
    // returns 0 / -1
    int decode(const uint8_t *in, uint8_t *img, int h, int w, int near, int k_step, int effort, int device);
    // QNBLIC (effort 0) decoder: header + histogram tables are parsed on the host, the pixel loop
    // (model + rANS) runs on the device.  returns 0 / -1
    int qdecode(const uint16_t *in, uint8_t *img, int *h, int *w, long max_px, int device);
    uint8_t *d_qtab = nullptr;     // freq[12][256] u32 | start[12][256] u32 | slot[12][32768] u8
    int *d_status = nullptr;
};

}  // namespace nblic

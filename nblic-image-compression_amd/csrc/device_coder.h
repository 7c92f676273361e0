// device_coder.h -- the range-coder stage on the GPU, one lane per image (device_coder.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nblic {

// One image of a device-coder launch (array in device memory; lane = index & 63 of wave index / 64).
struct RcJob {
    const uint16_t *coded;     // prob | bin << 15 per bin; 256-byte aligned, readable to the end of its last 512-byte window
    uint8_t *out;              // device buffer for the coder bytes (no header)
    uint32_t *len_out;         // bytes written (flush included), or 0xFFFFFFFF when cap was too small
    uint32_t n, cap;
};

bool device_range_code(const RcJob *d_jobs, int n_jobs, hipStream_t s);

}  // namespace nblic

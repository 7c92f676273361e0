// device_coder.hip -- stage S6 (the 32-bit carry-less binary range coder, NBLIC.c:527-586) on the GPU,
// one LANE per image.
//
// The coder is one dependent chain per image (one interval update per bin, ~74 M bins for a 4096^2
// frame), so a lane codes its image no faster than ~17 Mbins/s -- forty times slower than a host
// core.  What the GPU offers instead is width: a wave codes 64 images at once, the wave needs one
// SIMD of the 1024 on the chip, and its input never leaves HBM (2 bytes per bin do not cross PCIe,
// only the 0.12 bytes per bin it produces do).  The pipeline (pipeline.hip) therefore uses it as a
// high-latency SUPPLEMENT to the host coder threads: when the backlog of finished images is deep
// enough that the host would need longer than a pack's latency to work through it, 64 images are
// handed to a wave.  Same bytes either way (tests/test_gpu_parity.py::test_device_coder_*).
//
// Memory traffic is staged exactly like the chain kernels' (kernels_e1.hip run_lane_streams): per round
// the wave fetches, for each of its 64 streams, the aligned 512-byte window holding that stream's next
// 256 bins with ONE coalesced request, parks the windows in LDS (rows padded to 65 words: the per-lane
// walk is bank-conflict free), and every lane walks its own row; the next round's 64 requests are in
// flight while the current one is walked.  Emitted bytes go straight to the lane's output (consecutive
// bytes of a lane share a line; the L2 merges them).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_coder.h"

namespace nblic {

#define NB_GLOBAL __attribute__((address_space(1)))
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int kRowWords = 65;

__global__ void __launch_bounds__(64) k_range_code_lanes(const RcJob *__restrict__ jobs, int n_jobs) {
    __shared__ u32x2 stage[64 * kRowWords];
    const int lane = int(threadIdx.x);
    const int id = int(blockIdx.x) * 64 + lane;
    const bool have = id < n_jobs;
    const RcJob J = jobs[have ? id : n_jobs - 1];
    const uint32_t n = have ? J.n : 0u;
    const auto in = (NB_GLOBAL const u32x2 *)J.coded;              // 256-byte aligned, padded by a window
    const auto out = (NB_GLOBAL uint8_t *)J.out;
    const uint32_t cap = J.cap >= 4u ? J.cap - 4u : 0u;           // room for the flush is kept back
    uint32_t lo = 0u, hi = 0xFFFFFFFFu, cnt = 0u;
    bool overflow = J.cap < 4u;

    uint32_t n_max = n;                                           // rounds are wave-uniform: the longest stream sets their number
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) n_max = max(n_max, uint32_t(__shfl_xor(int(n_max), d, 64)));
    const uint32_t rounds = (n_max + 255u) >> 8;
    const uint32_t my_rounds = (n + 255u) >> 8;

    // the pointers of the 64 streams, fetched per stream with readlane (wave-uniform addresses, coalesced 8-byte loads)
    const uint32_t p_lo = uint32_t(uintptr_t(J.coded)), p_hi = uint32_t(uintptr_t(J.coded) >> 32);
    auto window = [&](int l, uint32_t round) {
        const uint64_t base = (uint64_t(uint32_t(__builtin_amdgcn_readlane(int(p_hi), l))) << 32) | uint32_t(__builtin_amdgcn_readlane(int(p_lo), l));
        const uint32_t r_l = uint32_t(__builtin_amdgcn_readlane(int(my_rounds), l));
        const uint32_t rr = r_l == 0u ? 0u : (round < r_l ? round : r_l - 1u);                 // finished streams re-read their last window
        return ((NB_GLOBAL const u32x2 *)base)[size_t(rr) * 64u + uint32_t(lane)];
    };
    (void)in;
    u32x2 regs[64];
    if (rounds) {
#pragma unroll
        for (int l = 0; l < 64; l++) regs[l] = window(l, 0u);
    }
    for (uint32_t round = 0; round < rounds; round++) {
#pragma unroll
        for (int l = 0; l < 64; l++) stage[l * kRowWords + lane] = regs[l];
        __syncthreads();
        if (round + 1 < rounds) {                                 // next round's windows: issued now, consumed after the walk
#pragma unroll
            for (int l = 0; l < 64; l++) regs[l] = window(l, round + 1u);
        }
        const uint32_t base = round << 8;
        const uint32_t mine = base < n ? min(256u, n - base) : 0u;   // bins of this lane in this round
        for (uint32_t wi = 0; wi < 64u; wi++) {
            if (__ballot(wi * 4u < mine) == 0ull) break;
            if (wi * 4u < mine) {
                const u32x2 wv = stage[lane * kRowWords + int(wi)];
                const uint32_t rec[4] = {wv.x & 0xFFFFu, wv.x >> 16, wv.y & 0xFFFFu, wv.y >> 16};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (wi * 4u + uint32_t(k) < mine) {
                        const uint32_t prob = rec[k] & 0xFFFu, range = hi - lo;
                        // floor(range * prob / 4096) from two 24-bit multiplies (a 32-bit multiply is a quarter-rate op)
                        const uint32_t cut = lo + ((__umul24(range >> 16, prob) << 4) + (__umul24(range & 0xFFFFu, prob) >> 12));
                        const bool one = (rec[k] >> 15) != 0u;
                        hi = one ? cut : hi;
                        lo = one ? lo : cut + 1u;
                        while (((lo ^ hi) >> 24) == 0u) {
                            if (cnt < cap) out[cnt] = uint8_t(hi >> 24); else overflow = true;
                            cnt++;
                            lo <<= 8;
                            hi = (hi << 8) | 0xFFu;
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
    if (have) {
        if (!overflow) {
#pragma unroll
            for (int k = 0; k < 4; k++) { out[cnt + uint32_t(k)] = uint8_t(lo >> 24); lo <<= 8; }      // NBLIC.c:576-586
        }
        ((NB_GLOBAL uint32_t *)J.len_out)[0] = overflow ? 0xFFFFFFFFu : cnt + 4u;
    }
}

bool device_range_code(const RcJob *d_jobs, int n_jobs, hipStream_t s) {
    if (n_jobs <= 0) return true;
    hipLaunchKernelGGL(k_range_code_lanes, dim3(unsigned((n_jobs + 63) / 64)), dim3(64), 0, s, d_jobs, n_jobs);
    return hipGetLastError() == hipSuccess;
}

}  // namespace nblic

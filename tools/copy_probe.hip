// copy_probe.hip -- measurement aid: how does the runtime move a large device->host copy?
// Copies 128 MB chunks from HBM to (a) hipHostMalloc'ed and (b) hipHostRegister'ed memory and prints
// GB/s; run it under `rocprofv3 --kernel-trace --memory-copy-trace --stats` to see whether the copies
// show up as __amd_rocclr_copyBuffer dispatches (blit kernel on the CUs) or as SDMA transfers.
//   hipcc --offload-arch=gfx950 -O2 -o tools/copy_probe tools/copy_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/mman.h>

__global__ void k_touch(unsigned *p, size_t n) {
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) p[i] += 1u;
}

#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv) {
    const size_t bytes = size_t(128) << 20;
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    void *dev = nullptr, *pinned = nullptr;
    OK(hipMalloc(&dev, bytes));
    OK(hipMemset(dev, 1, bytes));
    OK(hipHostMalloc(&pinned, bytes, hipHostMallocDefault));
    void *reg = aligned_alloc(size_t(2) << 20, bytes);
    madvise(reg, bytes, MADV_HUGEPAGE);
    memset(reg, 0, bytes);
    OK(hipHostRegister(reg, bytes, hipHostRegisterDefault));
    hipStream_t s;
    OK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    // does a kernel in front of the copy change the engine?  (a) same stream, (b) kernel on s, copy on s2 behind an event
    {
        hipStream_t s2; hipEvent_t ev;
        OK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
        OK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        for (int mode = 0; mode < 2; mode++) {
            auto t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < reps; r++) {
                hipLaunchKernelGGL(k_touch, dim3(1024), dim3(256), 0, s, (unsigned *)dev, bytes / 4);
                if (mode == 0) {
                    OK(hipMemcpyAsync(reg, dev, bytes, hipMemcpyDeviceToHost, s));
                } else {
                    OK(hipEventRecord(ev, s));
                    OK(hipStreamWaitEvent(s2, ev, 0));
                    OK(hipMemcpyAsync(reg, dev, bytes, hipMemcpyDeviceToHost, s2));
                    OK(hipStreamSynchronize(s2));
                }
            }
            OK(hipStreamSynchronize(s)); OK(hipStreamSynchronize(s2));
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            printf("kernel then copy, %s: %.1f GB/s\n", mode == 0 ? "same stream" : "copy on a second stream behind an event", double(bytes) * reps / dt / 1e9);
        }
    }
    const char *names[2] = {"hipHostMalloc", "hipHostRegister"};
    void *dst[2] = {pinned, reg};
    for (int which = 0; which < 2; which++) {
        OK(hipMemcpyAsync(dst[which], dev, bytes, hipMemcpyDeviceToHost, s));
        OK(hipStreamSynchronize(s));
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; r++) OK(hipMemcpyAsync(dst[which], dev, bytes, hipMemcpyDeviceToHost, s));
        OK(hipStreamSynchronize(s));
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("%-16s D2H %zu MB x %d: %.1f GB/s\n", names[which], bytes >> 20, reps, double(bytes) * reps / dt / 1e9);
    }
    return 0;
}

// tools/microbench.hip -- calibration of the serial-chain regime on MI355X (not product code).
// How fast does ONE wave retire dependent / independent VALU, SALU and LDS operations?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void k_dep_valu(int n, int *out, unsigned long long *cyc) {
    int v = threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) v = v * 3 + 1;            // mad: dependent chain
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x + blockIdx.x * blockDim.x] = v;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_indep_valu(int n, int *out, unsigned long long *cyc) {
    int a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 4; k++) { a = a * 3 + 1; b = b * 5 + 1; c = c * 7 + 1; d = d * 9 + 1; }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x + blockIdx.x * blockDim.x] = a + b + c + d;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_dep_add(int n, int *out, unsigned long long *cyc) {
    int v = threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) v = (v ^ (v >> 3)) + k;     // 3 cheap dependent ops
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x + blockIdx.x * blockDim.x] = v;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_dep_lds(int n, int *out, unsigned long long *cyc) {
    __shared__ int tab[64 * 65];
    for (int i = threadIdx.x; i < 64 * 65; i += 64) tab[i] = (i * 7 + 3) % 64;
    __syncthreads();
    int v = threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) v = tab[v * 65 + threadIdx.x % 64 * 0 + (v & 63)] ;   // dependent LDS reads
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x + blockIdx.x * blockDim.x] = v;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <class K>
static void run(const char *name, K kern, int blocks, int threads, int n, int ops_per_iter) {
    int *out; unsigned long long *cyc, h[4096];
    hipMalloc(&out, sizeof(int) * blocks * threads); hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, n / 10, out, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, n, out, cyc); hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    double ops = double(n) * ops_per_iter;
    printf("%-14s blocks=%4d thr=%4d  %.3f ms  %.2f ns/op  memtime ticks/op %.2f (100MHz ticks => x%.1f cycles @2.4GHz)\n", name, blocks, threads, ms,
           ms * 1e6 / ops, double(h[0]) / ops, 24.0);
    hipFree(out); hipFree(cyc);
}

int main() {
    int n = 200000;
    run("dep_mad", k_dep_valu, 1, 64, n, 16);
    run("dep_mad x256CU", k_dep_valu, 256, 64, n, 16);
    run("dep_mad 4w/CU", k_dep_valu, 256, 256, n, 16);
    run("dep_mad 8w/SIMD", k_dep_valu, 256 * 8, 256, n, 16);
    run("indep_mad", k_indep_valu, 1, 64, n, 16);
    run("dep_xor_add", k_dep_add, 1, 64, n, 48);
    run("dep_lds", k_dep_lds, 1, 64, n, 16);
    return 0;
}

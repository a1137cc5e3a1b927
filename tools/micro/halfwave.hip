// Microbenchmark (GPU box): does a wave64 with only its low 32 lanes active issue
// VALU work faster than a full wave on gfx950's SIMD-32, and how many waves per
// SIMD does it take to reach the 2-cycle issue rate?   hipcc --offload-arch=gfx950 -O3 halfwave.hip -o /tmp/halfwave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int ILP>
__global__ void chain(float* out, int iters, int active_lanes) {
    const int lane = threadIdx.x & 63;
    float a[ILP];
#pragma unroll
    for (int k = 0; k < ILP; k++) a[k] = 1.0f + 1e-3f * (threadIdx.x + k);
    if (lane < active_lanes) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int k = 0; k < ILP; k++) a[k] = __builtin_fmaf(a[k], 1.0000001f, 1e-7f);
        }
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < ILP; k++) s += a[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ILP>
double run(int waves_per_simd, int active, int iters) {
    const int blocks = 256 * 4 * waves_per_simd;  // 64-thread blocks: one wave each
    float* out;
    hipMalloc(&out, sizeof(float) * blocks * 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    chain<ILP><<<blocks, 64>>>(out, iters, active);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    chain<ILP><<<blocks, 64>>>(out, iters, active);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipFree(out);
    return ms * 1e3;
}

int main() {
    const int iters = 20000;
    printf("ILP waves/SIMD active  us   cycles_per_fma_per_wave(@2.4GHz)\n");
    for (int w : {1, 2, 4}) for (int act : {64, 32}) {
        double us8 = run<8>(w, act, iters);
        printf("8   %d          %2d   %8.1f  %.2f\n", w, act, us8, us8 * 2400.0 / (double(iters) * 8 * w));
    }
    for (int w : {1, 2}) for (int act : {64, 32}) {
        double us1 = run<1>(w, act, iters);
        printf("1   %d          %2d   %8.1f  %.2f\n", w, act, us1, us1 * 2400.0 / (double(iters) * 1 * w));
    }
    return 0;
}

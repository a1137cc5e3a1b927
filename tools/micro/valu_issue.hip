// Microbenchmark (GPU box): VALU issue rate of v_fma_f32 / v_pk_fma_f32 / v_mul_lo_u32 on gfx950
// in shader cycles (s_memtime), for 1..8 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize valu_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void k(unsigned long long* cyc, float* sink, int iters) {
    float a[16];
    f2 p[16];
    unsigned u[16];
#pragma unroll
    for (int j = 0; j < 16; j++) { a[j] = 1.0f + 1e-3f * (threadIdx.x + j); p[j] = f2{a[j], a[j] + 1.f}; u[j] = threadIdx.x * 7u + j; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int j = 0; j < 16; j++) {
                if (KIND == 0) a[j] = __builtin_fmaf(a[j], 1.0000001f, 1e-7f);
                if (KIND == 1) p[j] = __builtin_elementwise_fma(p[j], f2{1.0000001f, 0.9999999f}, f2{1e-7f, 2e-7f});
                if (KIND == 2) u[j] = u[j] * 0x7feb352du + 1u;
                if (KIND == 3) a[j] = a[j] * 1.0000001f;
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) s += a[j] + p[j].x + p[j].y + (float)u[j];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char* name) {
    const int iters = 2000;
    for (int w : {1, 2, 4, 8}) {
        const int blocks = 256 * 4 * w;
        unsigned long long* cyc; float* sink;
        hipMalloc(&cyc, 8 * blocks); hipMalloc(&sink, 4 * blocks * 64);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<KIND><<<blocks, 64>>>(cyc, sink, iters);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<KIND><<<blocks, 64>>>(cyc, sink, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long* h = new unsigned long long[blocks];
        hipMemcpy(h, cyc, 8 * blocks, hipMemcpyDeviceToHost);
        double sum = 0; for (int i = 0; i < blocks; i++) sum += (double)h[i];
        const double per_wave_inst = sum / blocks / (double(iters) * 64);
        const double ns_per_simd_inst = ms * 1e6 / (double(iters) * 64 * w);   // wall time per instruction issued on one SIMD
        const double tick_ghz = (sum / blocks) / (ms * 1e6);                      // s_memtime ticks per ns (kernel ~= one wave's span)
        printf("%-16s waves/SIMD %d: %.2f ticks/inst/wave = %.2f ticks/inst/SIMD; wall %.3f ns/inst/SIMD (%.2f cycles @2.4GHz); ticks/ns %.2f\n",
               name, w, per_wave_inst, per_wave_inst / w, ns_per_simd_inst, ns_per_simd_inst * 2.4, tick_ghz);
        delete[] h; hipFree(cyc); hipFree(sink);
    }
}

int main() {
    run<0>("v_fma_f32");
    run<3>("v_mul_f32");
    run<1>("v_pk_fma_f32");
    run<2>("v_mul_lo_u32+add");
    return 0;
}

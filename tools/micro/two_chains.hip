// Microbenchmark (GPU box) for VERDICT r3 item 6: "two drones' independent RK4 chains interleaved per lane at
// <= 65 536 envs (32 768 lanes x 2 drones) -> the 4-cycle single-wave issue cost is hidden".
// The env step at 65 536 envs is one wave per SIMD issuing ~450 DEPENDENT f32 VALU instructions. The question the
// proposal rests on: does a lone wave issue faster when it holds two independent dependency chains instead of one?
// Shapes, all with the same total work (65 536 chains of N dependent v_fma_f32):
//   A  1024 waves x 1 chain  per lane   (today: one wave on every SIMD)
//   B   512 waves x 2 chains per lane   (the proposal: half the waves, two interleaved chains each)
//   C  1024 waves x 2 chains per lane   (131 072 chains: what two chains cost a wave that keeps its SIMD to itself)
//   D  2048 waves x 1 chain  per lane   (131 072 chains as two waves per SIMD: what a second WAVE buys instead)
// Reported: shader cycles per wave from s_memtime around the chain (per instruction issued), and the kernel's wall time
// from HIP events over 200 back-to-back launches.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o two_chains two_chains.hip && ./two_chains
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int N = 448;  // dependent instructions per chain (the env step: 390-450)

template <int CHAINS>
__global__ __launch_bounds__(256) void k(unsigned long long* cyc, float* sink, float seed) {
    float a[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) a[c] = seed + 1e-3f * (float)(threadIdx.x + 64 * c);
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < N; i++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) a[c] = __builtin_fmaf(a[c], 1.0000001f, 1e-7f);  // chains alternate instruction by instruction
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::"v"(a[0]), "v"(a[CHAINS - 1]));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) s += a[c];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;
}

template <int CHAINS>
void run(const char* name, int waves) {
    const int blocks = waves / 4;  // 256 threads = 4 waves per workgroup, like the env kernels
    unsigned long long* cyc;
    float* sink;
    hipMalloc(&cyc, 8 * waves);
    hipMalloc(&sink, 4 * waves * 64);
    for (int i = 0; i < 20; i++) k<CHAINS><<<blocks, 256>>>(cyc, sink, 1.0f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int reps = 200;
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++) k<CHAINS><<<blocks, 256>>>(cyc, sink, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(waves);
    hipMemcpy(h.data(), cyc, 8 * waves, hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto x : h) sum += (double)x;
    const double ticks = sum / waves;
    printf("%-34s waves %5d  chains/lane %d  instr/wave %4d : %7.1f s_memtime ticks per wave = %.2f per instruction; launch-to-launch %.2f us\n", name, waves, CHAINS,
           N * CHAINS, ticks, ticks / (N * CHAINS), ms * 1e3 / reps);
    hipFree(cyc);
    hipFree(sink);
}

int main() {
    run<1>("A 65536 chains, 1 wave/SIMD x1", 1024);
    run<2>("B 65536 chains, 512 waves x2", 512);
    run<2>("C 131072 chains, 1 wave/SIMD x2", 1024);
    run<1>("D 131072 chains, 2 waves/SIMD x1", 2048);
    run<4>("E 131072 chains, 512 waves x4", 512);
    return 0;
}

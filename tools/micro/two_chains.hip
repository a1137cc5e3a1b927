// Microbenchmark (GPU box) for VERDICT r3 item 6: "two drones' independent RK4 chains interleaved per lane at
// <= 65 536 envs (32 768 lanes x 2 drones) -> the 4-cycle single-wave issue cost is hidden".
// The env step at 65 536 envs is one wave per SIMD issuing ~450 DEPENDENT f32 VALU instructions. The question the
// proposal rests on: does a lone wave issue faster when it holds two independent dependency chains instead of one?
// Shapes, all with the same total work (65 536 chains of N dependent v_fma_f32):
//   A  1024 waves x 1 chain  per lane   (today: one wave on every SIMD)
//   B   512 waves x 2 chains per lane   (the proposal: half the waves, two interleaved chains each)
//   C  1024 waves x 2 chains per lane   (131 072 chains: what two chains cost a wave that keeps its SIMD to itself)
//   D  2048 waves x 1 chain  per lane   (131 072 chains as two waves per SIMD: what a second WAVE buys instead)
// Reported, from HIP events over 200 back-to-back launches: launch-to-launch time with one "step" (N instructions per
// chain) per launch, and the marginal time of a step inside a launch (64 vs 16 steps) = the wave's own issue time.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o two_chains two_chains.hip && ./two_chains
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int N = 448;  // dependent instructions per chain and "step" (the env step: 390-450)

template <int CHAINS>
__global__ __launch_bounds__(256) void k(float* sink, float seed, int steps) {
    float a[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) a[c] = seed + 1e-3f * (float)(threadIdx.x + 64 * c);
    for (int s = 0; s < steps; s++) {  // run-time trip count: `steps` env steps' worth of arithmetic per launch
#pragma unroll
        for (int i = 0; i < N; i++) {
#pragma unroll
            for (int c = 0; c < CHAINS; c++) a[c] = __builtin_fmaf(a[c], 1.0000001f, 1e-7f);  // chains alternate instruction by instruction
        }
    }
    float r = 0.f;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) r += a[c];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int CHAINS>
double launch_us(int waves, int steps, float* sink) {
    const int blocks = waves / 4;  // 256 threads = 4 waves per workgroup, like the env kernels
    for (int i = 0; i < 20; i++) k<CHAINS><<<blocks, 256>>>(sink, 1.0f, steps);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int reps = 200;
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++) k<CHAINS><<<blocks, 256>>>(sink, 1.0f, steps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return ms * 1e3 / reps;
}

template <int CHAINS>
void run(const char* name, int waves, float* sink) {
    // one step per launch (launch boundary included, like drone_vec_step), and the marginal cost of a step inside a
    // launch (64 vs 16 steps: like drone_vec_step_many / the fused rollout), which is the wave's own issue time
    const double one = launch_us<CHAINS>(waves, 1, sink), t16 = launch_us<CHAINS>(waves, 16, sink), t64 = launch_us<CHAINS>(waves, 64, sink);
    const double per_step = (t64 - t16) / 48.0;
    printf("%-36s waves %5d x %d chains/lane: 1 step per launch %.2f us launch-to-launch; in-launch %.3f us per step = %.2f ns per instruction per wave (%.2f cycles at 2.4 GHz)\n",
           name, waves, CHAINS, one, per_step, per_step * 1e3 / (N * CHAINS), per_step * 1e3 / (N * CHAINS) * 2.4);
}

int main() {
    float* sink;
    hipMalloc(&sink, 4 * 4096 * 64);
    run<1>("A  65536 chains: 1 wave/SIMD x 1", 1024, sink);
    run<2>("B  65536 chains: 512 waves x 2 (the proposal)", 512, sink);
    run<2>("C 131072 chains: 1 wave/SIMD x 2", 1024, sink);
    run<1>("D 131072 chains: 2 waves/SIMD x 1", 2048, sink);
    run<4>("E 131072 chains: 512 waves x 4", 512, sink);
    hipFree(sink);
    return 0;
}

// Microbenchmark (GPU box): what a plain float4 streaming kernel reaches on this chip, for the
// read:write mix of the per-step kernel (112 B read : 166 B written per env) and for a 1:1 copy.
//   hipcc --offload-arch=gfx950 -O3 copy_bw.hip -o /tmp/copy_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f4 __attribute__((ext_vector_type(4)));

// each thread: R float4 loads from R read streams, W float4 stores to W write streams
template <int R, int W, bool NT>
__global__ __launch_bounds__(256) void stream(const f4* __restrict__ in, f4* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    f4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < R; r++) acc += in[(size_t)r * n + i];
#pragma unroll
    for (int w = 0; w < W; w++) {
        f4 v = acc + (float)w;
        if (NT) __builtin_nontemporal_store(v, &out[(size_t)w * n + i]);
        else out[(size_t)w * n + i] = v;
    }
}

// same bytes, but the R (W) float4 of the 64 elements of a wave sit together: [wave tile][stream][lane]
// -> one fat read stream and one fat write stream instead of R + W thin ones
template <int R, int W, bool NT>
__global__ __launch_bounds__(256) void stream_tiled(const f4* __restrict__ in, f4* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const size_t tile = i >> 6, lane = i & 63;
    f4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < R; r++) acc += in[(tile * R + r) * 64 + lane];
#pragma unroll
    for (int w = 0; w < W; w++) {
        f4 v = acc + (float)w;
        if (NT) __builtin_nontemporal_store(v, &out[(tile * W + w) * 64 + lane]);
        else out[(tile * W + w) * 64 + lane] = v;
    }
}

template <int R, int W, bool NT, bool TILED = false>
void run(const char* name, size_t n) {
    f4 *in, *out;
    hipMalloc(&in, sizeof(f4) * n * R);
    hipMalloc(&out, sizeof(f4) * n * W);
    hipMemset(in, 0, sizeof(f4) * n * R);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int reps = 200;
    auto launch = [&]() {
        if (TILED) stream_tiled<R, W, NT><<<(n + 255) / 256, 256>>>(in, out, n);
        else stream<R, W, NT><<<(n + 255) / 256, 256>>>(in, out, n);
    };
    for (int k = 0; k < 20; k++) launch();
    hipEventRecord(e0);
    for (int k = 0; k < reps; k++) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 16.0 * n * (R + W);
    printf("%-34s %2d read + %2d write streams, %6.1f MB/launch: %7.2f us  %6.2f TB/s\n", name, R, W, bytes / 1e6, ms * 1e3 / reps, bytes * reps / (ms * 1e-3) / 1e12);
    hipFree(in);
    hipFree(out);
}

// an empty launch of the same grid: the launch-to-launch floor of a dependent stream of kernels
__global__ __launch_bounds__(256) void empty_kernel(const f4* in, f4* out, size_t n) {}

void run_empty(size_t n) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int reps = 2000;
    for (int k = 0; k < 20; k++) empty_kernel<<<(n + 255) / 256, 256>>>(nullptr, nullptr, n);
    hipEventRecord(e0);
    for (int k = 0; k < reps; k++) empty_kernel<<<(n + 255) / 256, 256>>>(nullptr, nullptr, n);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s grid %zu: %7.2f us launch-to-launch\n", "empty kernel", (n + 255) / 256, ms * 1e3 / reps);
}

// usage: copy_bw [n ...]   (elements per stream = envs; default 2^20)
int main(int argc, char** argv) {
    size_t sizes[16];
    int ns = 0;
    for (int k = 1; k < argc && ns < 16; k++) sizes[ns++] = (size_t)strtoull(argv[k], nullptr, 10);
    if (ns == 0) sizes[ns++] = 1 << 20;
    for (int k = 0; k < ns; k++) {
        const size_t n = sizes[k];  // one float4 per env per stream, as in the env kernels
        printf("---- n = %zu elements per stream ----\n", n);
        run_empty(n);
        run<1, 1, false>("copy 1:1", n * 8);
        run<7, 10, false>("env-like 7:10 (112 B : 160 B)", n);
        run<7, 10, true>("env-like 7:10, nt stores", n);
        run<7, 10, false, true>("env-like 7:10, wave-tiled layout", n);
        run<7, 10, true, true>("env-like 7:10, wave-tiled, nt", n);
        run<9, 9, false>("9:9", n);
        run<0, 15, false>("write only (reset-like)", n);
        run<15, 1, false>("read mostly 15:1", n);
    }
    return 0;
}

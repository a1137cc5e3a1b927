// Microbenchmark (GPU box): the memory pattern of the per-step env kernel with NO arithmetic, to find
// the layout / launch shape that streams its byte mix fastest where the Infinity Cache cannot help
// (2^22 envs: 1.1 GB per launch), and the compute-free floor at small shards.
//   per env: read 6 state float4 + 1 action float4; write 5 state float4 IN PLACE + 5 obs float4 (AoS rows,
//   stored flat per wave as the real kernel does after its LDS transpose) + 1 reward float = 276 B.
// State layouts: planes [P][n] (round 1) or tiles [n/T][P][T] for T = 64, 256, 1024 envs.
//   hipcc --offload-arch=gfx950 -O3 stream_mix.hip -o /tmp/stream_mix && /tmp/stream_mix 4194304 1048576 131072 65536
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int P = 6;   // state planes read
constexpr int PW = 5;  // state planes written back

template <int T>
__device__ __forceinline__ size_t state_index(size_t n, size_t i, int p) {
    if (T == 0) return (size_t)p * n + i;                  // planes
    return ((i / T) * P + p) * T + (i % T);                // tiles of T envs
}

template <int T, bool NT, int BLOCK, bool XCD>
__global__ __launch_bounds__(BLOCK) void env_like(f4* __restrict__ state, const f4* __restrict__ act, f4* __restrict__ obs, float* __restrict__ rew, size_t n) {
    unsigned chunk = blockIdx.x;
    if (XCD) {  // workgroups of one XCD (blockIdx % 8) take one contiguous eighth
        const unsigned nwg = gridDim.x, x = blockIdx.x & 7u, q = nwg >> 3, r = nwg & 7u;
        chunk = (x < r ? x * (q + 1u) : r * (q + 1u) + (x - r) * q) + (blockIdx.x >> 3);
    }
    const size_t i = (size_t)chunk * BLOCK + threadIdx.x;
    if (i >= n) return;
    f4 s[P];
#pragma unroll
    for (int p = 0; p < P; p++) s[p] = state[state_index<T>(n, i, p)];
    const f4 a = act[i];
    f4 acc = a;
#pragma unroll
    for (int p = 0; p < P; p++) acc += s[p];
#pragma unroll
    for (int p = 0; p < PW; p++) {
        const f4 v = s[p] + acc;
        if (NT) __builtin_nontemporal_store(v, &state[state_index<T>(n, i, p)]);
        else state[state_index<T>(n, i, p)] = v;
    }
    const size_t lane = threadIdx.x & 63, wave_base = i - lane;
    f4* dst = obs + wave_base * 5;
#pragma unroll
    for (int k = 0; k < 5; k++) __builtin_nontemporal_store(acc + (float)k, &dst[k * 64 + lane]);
    __builtin_nontemporal_store(acc.x, &rew[i]);
}

// the same traffic with the real kernel's other two properties dialled in: VALU work between the loads and the
// stores (FMAS dependent v_fma_f32 per lane, in 4 chains) and a cap on waves per SIMD (the real kernel holds 5)
template <int FMAS, int WAVES, bool EARLY_STATE_STORE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, WAVES))) void env_like_work(f4* __restrict__ state, const f4* __restrict__ act, f4* __restrict__ obs, float* __restrict__ rew, size_t n, float k) {
    const unsigned nwg = gridDim.x, x = blockIdx.x & 7u, q = nwg >> 3, r = nwg & 7u;
    const unsigned chunk = (x < r ? x * (q + 1u) : r * (q + 1u) + (x - r) * q) + (blockIdx.x >> 3);
    const size_t i = (size_t)chunk * 256 + threadIdx.x;
    if (i >= n) return;
    f4 s[P];
#pragma unroll
    for (int p = 0; p < P; p++) s[p] = state[state_index<0>(n, i, p)];
    const f4 a = act[i];
    f4 acc = a;
#pragma unroll
    for (int p = 0; p < P; p++) acc += s[p];
#pragma unroll 8
    for (int j = 0; j < FMAS / 4; j++) {
        acc.x = __builtin_fmaf(acc.x, k, 1.0f);
        acc.y = __builtin_fmaf(acc.y, k, 1.0f);
        acc.z = __builtin_fmaf(acc.z, k, 1.0f);
        acc.w = __builtin_fmaf(acc.w, k, 1.0f);
    }
#pragma unroll
    for (int p = 0; p < PW; p++) __builtin_nontemporal_store(s[p] + acc, &state[state_index<0>(n, i, p)]);
    if (EARLY_STATE_STORE) {
#pragma unroll 8
        for (int j = 0; j < 64 / 4; j++) {
            acc.x = __builtin_fmaf(acc.x, k, 1.0f);
            acc.y = __builtin_fmaf(acc.y, k, 1.0f);
            acc.z = __builtin_fmaf(acc.z, k, 1.0f);
            acc.w = __builtin_fmaf(acc.w, k, 1.0f);
        }
    }
    const size_t lane = threadIdx.x & 63, wave_base = i - lane;
    f4* dst = obs + wave_base * 5;
#pragma unroll
    for (int kk = 0; kk < 5; kk++) __builtin_nontemporal_store(acc + (float)kk, &dst[kk * 64 + lane]);
    __builtin_nontemporal_store(acc.x, &rew[i]);
}

template <int FMAS, int WAVES, bool EARLY = false>
void run_work(const char* name, f4* state, f4* act, f4* obs, float* rew, size_t n, int reps);

// persistent form: grid = a few workgroups per CU, each walks tiles with a stride
template <int T, bool NT, int BLOCK>
__global__ __launch_bounds__(BLOCK) void env_like_persistent(f4* __restrict__ state, const f4* __restrict__ act, f4* __restrict__ obs, float* __restrict__ rew, size_t n) {
    for (size_t base = (size_t)blockIdx.x * BLOCK; base < n; base += (size_t)gridDim.x * BLOCK) {
        const size_t i = base + threadIdx.x;
        f4 s[P];
#pragma unroll
        for (int p = 0; p < P; p++) s[p] = state[state_index<T>(n, i, p)];
        const f4 a = act[i];
        f4 acc = a;
#pragma unroll
        for (int p = 0; p < P; p++) acc += s[p];
#pragma unroll
        for (int p = 0; p < PW; p++) {
            const f4 v = s[p] + acc;
            if (NT) __builtin_nontemporal_store(v, &state[state_index<T>(n, i, p)]);
            else state[state_index<T>(n, i, p)] = v;
        }
        const size_t lane = threadIdx.x & 63, wave_base = i - lane;
        f4* dst = obs + wave_base * 5;
#pragma unroll
        for (int k = 0; k < 5; k++) __builtin_nontemporal_store(acc + (float)k, &dst[k * 64 + lane]);
        __builtin_nontemporal_store(acc.x, &rew[i]);
    }
}

// yardsticks on one fat stream
__global__ __launch_bounds__(256) void copy_inplace(f4* __restrict__ a, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) a[i] = a[i] + 1.0f;
}
__global__ __launch_bounds__(256) void copy_oop(const f4* __restrict__ a, f4* __restrict__ b, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) b[i] = a[i] + 1.0f;
}
__global__ __launch_bounds__(256) void read_only(const f4* __restrict__ a, float* __restrict__ sink, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const f4 v = a[i];
        if (v.x == 123.456f) sink[0] = v.y;  // never true: keeps the load
    }
}
__global__ __launch_bounds__(256) void write_only(f4* __restrict__ a, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const f4 v = {1, 2, 3, 4};
    if (i < n) a[i] = v;
}

struct Bufs {
    f4 *state, *act, *obs;
    float* rew;
};

template <class F>
double time_us(F launch, int reps) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int k = 0; k < 10; k++) launch();
    (void)hipEventRecord(e0);
    for (int k = 0; k < reps; k++) launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return ms * 1e3 / reps;
}

void report(const char* name, size_t n, double us, double bytes_per_env) {
    const double b = bytes_per_env * (double)n;
    printf("  %-52s %9.2f us  %6.2f TB/s  %5.3f of 8 TB/s\n", name, us, b / us / 1e6, b / us / 1e6 / 8.0);
}

template <int T, bool NT, int BLOCK, bool XCD>
void run_env(const char* name, const Bufs& B, size_t n, int reps) {
    const unsigned grid = (unsigned)((n + BLOCK - 1) / BLOCK);
    report(name, n, time_us([&]() { env_like<T, NT, BLOCK, XCD><<<grid, BLOCK>>>(B.state, B.act, B.obs, B.rew, n); }, reps), 276.0);
}
template <int T, bool NT, int BLOCK>
void run_env_p(const char* name, const Bufs& B, size_t n, int reps, int wg_per_cu) {
    unsigned grid = 256u * wg_per_cu;
    const unsigned full = (unsigned)(n / BLOCK);
    if (grid > full) grid = full;
    report(name, n, time_us([&]() { env_like_persistent<T, NT, BLOCK><<<grid, BLOCK>>>(B.state, B.act, B.obs, B.rew, n); }, reps), 276.0);
}

template <int FMAS, int WAVES, bool EARLY>
void run_work(const char* name, f4* state, f4* act, f4* obs, float* rew, size_t n, int reps) {
    const unsigned grid = (unsigned)((n + 255) / 256);
    report(name, n, time_us([&]() { env_like_work<FMAS, WAVES, EARLY><<<grid, 256>>>(state, act, obs, rew, n, 0.999f); }, reps), 276.0);
}

int main(int argc, char** argv) {
    size_t sizes[16];
    int ns = 0;
    for (int k = 1; k < argc && ns < 16; k++) sizes[ns++] = (size_t)strtoull(argv[k], nullptr, 10);
    if (ns == 0) sizes[ns++] = 1 << 22;
    for (int k = 0; k < ns; k++) {
        const size_t n = sizes[k];  // multiple of 1024
        const int reps = n >= (1u << 21) ? 50 : 300;
        Bufs B;
        (void)hipMalloc(&B.state, sizeof(f4) * n * P);
        (void)hipMalloc(&B.act, sizeof(f4) * n);
        (void)hipMalloc(&B.obs, sizeof(f4) * n * 5);
        (void)hipMalloc(&B.rew, sizeof(float) * n);
        (void)hipMemset(B.state, 0, sizeof(f4) * n * P);
        (void)hipMemset(B.act, 0, sizeof(f4) * n);
        printf("---- n = %zu envs: %.1f MB per launch ----\n", n, 276.0 * n / 1e6);
        // yardsticks: one fat stream over the state buffer (6n float4 = 96 B/env)
        {
            const size_t m = n * P;
            const unsigned g = (unsigned)((m + 255) / 256);
            report("yardstick: in-place RMW a[i] += 1 (96 r + 96 w)", n, time_us([&]() { copy_inplace<<<g, 256>>>(B.state, m); }, reps), 192.0);
            report("yardstick: copy state -> obs (80 r + 80 w)", n, time_us([&]() { copy_oop<<<(unsigned)((n * 5 + 255) / 256), 256>>>(B.state, B.obs, n * 5); }, reps), 160.0);
            report("yardstick: read only (96 r)", n, time_us([&]() { read_only<<<g, 256>>>(B.state, B.rew, m); }, reps), 96.0);
            report("yardstick: write only (96 w)", n, time_us([&]() { write_only<<<g, 256>>>(B.state, m); }, reps), 96.0);
        }
        run_env<0, false, 256, true>("planes, wg256, xcd-chunked (round-1 shape)", B, n, reps);
        run_env<0, true, 256, true>("planes, wg256, xcd, nt state stores (round-1 final)", B, n, reps);
        run_env<0, true, 256, false>("planes, wg256, round-robin, nt", B, n, reps);
        run_env<64, false, 256, true>("tiles of 64, wg256, xcd", B, n, reps);
        run_env<64, true, 256, true>("tiles of 64, wg256, xcd, nt", B, n, reps);
        run_env<256, false, 256, true>("tiles of 256, wg256, xcd", B, n, reps);
        run_env<256, true, 256, true>("tiles of 256, wg256, xcd, nt", B, n, reps);
        run_env<256, true, 256, false>("tiles of 256, wg256, round-robin, nt", B, n, reps);
        run_env<1024, true, 256, true>("tiles of 1024, wg256, xcd, nt", B, n, reps);
        run_env<1024, true, 1024, true>("tiles of 1024, wg1024, xcd, nt", B, n, reps);
        run_env<64, true, 64, true>("tiles of 64, wg64, xcd, nt", B, n, reps);
        run_env<0, true, 64, true>("planes, wg64, xcd, nt", B, n, reps);
        run_work<0, 8>("planes + 0 fma, <=8 waves/SIMD", B.state, B.act, B.obs, B.rew, n, reps);
        run_work<0, 5>("planes + 0 fma, <=5 waves/SIMD", B.state, B.act, B.obs, B.rew, n, reps);
        run_work<0, 4>("planes + 0 fma, <=4 waves/SIMD", B.state, B.act, B.obs, B.rew, n, reps);
        run_work<0, 2>("planes + 0 fma, <=2 waves/SIMD", B.state, B.act, B.obs, B.rew, n, reps);
        run_work<200, 8>("planes + 200 fma, <=8 waves/SIMD", B.state, B.act, B.obs, B.rew, n, reps);
        run_work<200, 5>("planes + 200 fma, <=5 waves/SIMD", B.state, B.act, B.obs, B.rew, n, reps);
        run_work<448, 8>("planes + 448 fma, <=8 waves/SIMD", B.state, B.act, B.obs, B.rew, n, reps);
        run_work<448, 6>("planes + 448 fma, <=6 waves/SIMD", B.state, B.act, B.obs, B.rew, n, reps);
        run_work<448, 5>("planes + 448 fma, <=5 waves/SIMD", B.state, B.act, B.obs, B.rew, n, reps);
        run_work<448, 4>("planes + 448 fma, <=4 waves/SIMD", B.state, B.act, B.obs, B.rew, n, reps);
        run_work<448, 5, true>("planes + 448 fma, <=5 waves, +64 fma before obs", B.state, B.act, B.obs, B.rew, n, reps);
        run_work<640, 5>("planes + 640 fma, <=5 waves/SIMD", B.state, B.act, B.obs, B.rew, n, reps);
        run_env_p<256, true, 256>("tiles of 256, persistent 4 wg/CU, nt", B, n, reps, 4);
        run_env_p<256, true, 256>("tiles of 256, persistent 8 wg/CU, nt", B, n, reps, 8);
        run_env_p<0, true, 256>("planes, persistent 8 wg/CU, nt", B, n, reps, 8);
        (void)hipFree(B.state);
        (void)hipFree(B.act);
        (void)hipFree(B.obs);
        (void)hipFree(B.rew);
    }
    return 0;
}

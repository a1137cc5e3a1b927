// Microbenchmark (GPU box): the host-visible round trip of ONE small launch — what a host-buffer vec-env step can cost at
// the least — by how the host learns that the kernel has finished.
//   hipcc --offload-arch=gfx950 -O3 host_roundtrip.hip -o /tmp/host_roundtrip && /tmp/host_roundtrip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>

__global__ void empty_kernel() {}

// the last workgroup of the launch writes the flag itself (system-scope release), no extra packet on the queue
__global__ void flag_kernel(volatile uint32_t* flag, uint32_t* arrive, uint32_t seq) {
    __threadfence_system();
    if (threadIdx.x == 0 && atomicAdd(arrive, 1u) == gridDim.x - 1u) {
        *arrive = 0u;
        __hip_atomic_store(const_cast<uint32_t*>(flag), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// 16 bytes per thread read from and written to mapped host memory: the PCIe legs of a zero-copy step
__global__ void touch_kernel(const float4* in, float4* out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i];
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void spin(volatile uint32_t* f, uint32_t seq) {
    while (__atomic_load_n(f, __ATOMIC_ACQUIRE) != seq) __builtin_ia32_pause();
}

int main() {
    hipStream_t s;
    (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    volatile uint32_t* hflag;
    void* dflag;
    (void)hipHostMalloc((void**)&hflag, 64, hipHostMallocMapped);
    (void)hipHostGetDevicePointer(&dflag, (void*)hflag, 0);
    *hflag = 0;
    uint32_t* arrive;
    (void)hipMalloc(&arrive, 4);
    (void)hipMemset(arrive, 0, 4);
    float4 *hin, *hout, *din, *dout;
    const uint32_t n = 1024 * 8;  // 128 KiB each way: a 1 024-env step's order of magnitude
    (void)hipHostMalloc((void**)&hin, n * 16, hipHostMallocMapped);
    (void)hipHostMalloc((void**)&hout, n * 16, hipHostMallocMapped);
    (void)hipHostGetDevicePointer((void**)&din, hin, 0);
    (void)hipHostGetDevicePointer((void**)&dout, hout, 0);
    const int reps = 5000;
    uint32_t seq = 0;
    for (int grid : {1, 4, 64}) {
        for (int k = 0; k < 200; k++) { empty_kernel<<<grid, 256, 0, s>>>(); (void)hipStreamSynchronize(s); }
        double t0 = now_us();
        for (int k = 0; k < reps; k++) { empty_kernel<<<grid, 256, 0, s>>>(); (void)hipStreamSynchronize(s); }
        const double a = (now_us() - t0) / reps;
        t0 = now_us();
        for (int k = 0; k < reps; k++) { empty_kernel<<<grid, 256, 0, s>>>(); (void)hipStreamWriteValue32(s, dflag, ++seq, 0); spin(hflag, seq); }
        const double b = (now_us() - t0) / reps;
        t0 = now_us();
        for (int k = 0; k < reps; k++) { flag_kernel<<<grid, 256, 0, s>>>((volatile uint32_t*)dflag, arrive, ++seq); spin(hflag, seq); }
        const double c = (now_us() - t0) / reps;
        printf("grid %3d x 256: launch + hipStreamSynchronize %6.2f us | + hipStreamWriteValue32, host polls %6.2f us | kernel writes the flag, host polls %6.2f us\n", grid, a, b, c);
    }
    (void)hipStreamSynchronize(s);
    for (int k = 0; k < 200; k++) { touch_kernel<<<n / 256, 256, 0, s>>>(din, dout, n); (void)hipStreamSynchronize(s); }
    double t0 = now_us();
    for (int k = 0; k < reps; k++) { touch_kernel<<<n / 256, 256, 0, s>>>(din, dout, n); (void)hipStreamWriteValue32(s, dflag, ++seq, 0); spin(hflag, seq); }
    printf("128 KiB read from + 128 KiB written to mapped host memory, write-value + poll: %6.2f us\n", (now_us() - t0) / reps);
    return 0;
}

/* tests/test_peer_store_gpu.py::test_export_of_a_virtual_memory_allocation_names_the_way_out — VERDICT r5 item 5.
 * The root's global batch in a HIP virtual-memory mapping (hipMemCreate + hipMemAddressReserve + hipMemMap: what an
 * expandable-segments allocator hands out) has no IPC handle: drone_vec_gather_peer_export must fail with a message that
 * names drone_device_malloc. The same export from drone_device_malloc buffers is the positive control.
 *   gcc -O1 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tests/c/vmm_export_probe.c -Ldrone_amd -l:libdrone_hip.so
 *       -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/drone_amd -Wl,-rpath,/opt/rocm/lib */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <string.h>

#include "drone_vec.h"

int main(void) {
    const int n = 4096, dev = 0;
    DroneConfig cfg;
    drone_config_default(&cfg, DRONE_TASK_HOVER);
    cfg.buffer_kind = DRONE_BUFFERS_DEVICE;
    cfg.device = dev;
    DroneVec* v = drone_vec_init(NULL, NULL, NULL, NULL, NULL, n, 1, &cfg);
    if (!v) { printf("INIT_FAILED %s\n", drone_last_error()); return 2; }
    unsigned char token[DRONE_PEER_TOKEN_BYTES];

    /* positive control: plain allocations */
    float* obs = (float*)drone_device_malloc(dev, (size_t)n * DRONE_OBS_DIM * 4);
    float* rew = (float*)drone_device_malloc(dev, (size_t)n * 4);
    unsigned char* term = (unsigned char*)drone_device_malloc(dev, n);
    unsigned char* trunc = (unsigned char*)drone_device_malloc(dev, n);
    const int ok = drone_vec_gather_peer_export(v, obs, rew, term, trunc, token);
    printf("PLAIN rc=%d %s\n", ok, ok ? drone_last_error() : "");

    /* the same four buffers inside ONE virtual-memory mapping */
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    if (hipSetDevice(dev) != hipSuccess || hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum) != hipSuccess || gran == 0) {
        printf("VMM_UNAVAILABLE granularity: %s\n", hipGetErrorString(hipGetLastError()));
        return 0;
    }
    size_t want = (size_t)n * (DRONE_OBS_DIM * 4 + 4 + 2) + 4096, size = (want + gran - 1) / gran * gran;
    hipMemGenericAllocationHandle_t h;
    void* base = NULL;
    hipError_t e;
    if ((e = hipMemCreate(&h, size, &prop, 0)) != hipSuccess) { printf("VMM_UNAVAILABLE hipMemCreate: %s\n", hipGetErrorString(e)); return 0; }
    if ((e = hipMemAddressReserve(&base, size, gran, NULL, 0)) != hipSuccess) { printf("VMM_UNAVAILABLE hipMemAddressReserve: %s\n", hipGetErrorString(e)); return 0; }
    if ((e = hipMemMap(base, size, 0, h, 0)) != hipSuccess) { printf("VMM_UNAVAILABLE hipMemMap: %s\n", hipGetErrorString(e)); return 0; }
    hipMemAccessDesc acc;
    memset(&acc, 0, sizeof(acc));
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = dev;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if ((e = hipMemSetAccess(base, size, &acc, 1)) != hipSuccess) { printf("VMM_UNAVAILABLE hipMemSetAccess: %s\n", hipGetErrorString(e)); return 0; }
    char* p = (char*)base;
    float* vobs = (float*)p;
    float* vrew = (float*)(p + (size_t)n * DRONE_OBS_DIM * 4);
    unsigned char* vterm = (unsigned char*)(p + (size_t)n * (DRONE_OBS_DIM * 4 + 4));
    unsigned char* vtrunc = vterm + n;
    const int rc = drone_vec_gather_peer_export(v, vobs, vrew, vterm, vtrunc, token);
    printf("VMM rc=%d %s\n", rc, rc ? drone_last_error() : "(exported: this runtime gives a virtual-memory mapping an IPC handle)");
    drone_vec_close(v);
    return 0;
}

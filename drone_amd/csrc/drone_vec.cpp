// drone_vec.cpp — host side of the C-ABI in include/drone_vec.h.
//
// Owns the device state planes, the KParams block, the stream and (in
// host-buffer mode) the device mirrors of the caller's buffers. Every entry
// point ends in a HIP launch from drone_kernels.hip: there is no CPU
// implementation of the env in this library, and it fails loudly (NULL /
// non-zero + drone_last_error()) when HIP or the device is unavailable.
//
// Replaces, on the PufferLib side, the binding's vec_init / vec_reset /
// vec_step / vec_log / vec_close loop over per-env c_step; the
// reference file:line cannot be cited — no source in /root/reference
// (.gitmodules:1-3).
#include <hip/hip_runtime_api.h>
#include <dlfcn.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include <rccl/rccl.h>  // types only: the library is dlopen'ed on first use (drone_vec_gather_init), never linked

#include "drone_host_copy.hpp"
#include "drone_kernels.h"

using namespace drone;

namespace {

thread_local char g_err[512] = "";

void set_err(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

#define HIP_TRY(expr, onfail)                                                        \
    do {                                                                             \
        hipError_t e_ = (expr);                                                      \
        if (e_ != hipSuccess) {                                                      \
            set_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            onfail;                                                                  \
        }                                                                            \
    } while (0)

constexpr int kLogMaxGrid = 1024;

// DRONE_DEBUG_REG=1: trace every host-memory registration the library makes or drops (stderr), to match a GPU memory
// fault's address against what was mapped when
bool debug_reg() {
    static const bool on = [] { const char* e = getenv("DRONE_DEBUG_REG"); return e && *e && *e != '0'; }();
    return on;
}
hipError_t host_register(void* p, size_t bytes, const void* who, const char* what) {
    const hipError_t e = hipHostRegister(p, bytes, hipHostRegisterDefault);
    if (debug_reg()) fprintf(stderr, "[drone reg] + %p..%p %s of %p -> %s\n", p, (char*)p + bytes, what, who, e == hipSuccess ? "ok" : hipGetErrorString(e));
    return e;
}
void host_unregister(void* p, const void* who, const char* what) {
    const hipError_t e = hipHostUnregister(p);
    if (debug_reg()) fprintf(stderr, "[drone reg] - %p %s of %p -> %s\n", p, what, who, e == hipSuccess ? "ok" : hipGetErrorString(e));
    if (e != hipSuccess) (void)hipGetLastError();
}

}  // namespace

struct DroneVec {
    DroneConfig cfg;
    KParams kp;
    uint64_t seed;
    uint32_t gstep;
    int n;
    uint32_t n_pad;
    uint32_t stride;
    int device;
    bool host_buffers;
    bool zero_copy;      // host buffers mapped into the device address space: kernels read / write them over PCIe directly
    // caller buffers (host or device, per cfg.buffer_kind)
    float* u_obs;
    float* u_act;
    float* u_rew;
    unsigned char* u_term;
    unsigned char* u_trunc;
    bool registered[5];
    void* registered_ptr[5];
    // device side
    DeviceView dv;
    uint32_t* d_kp;
    double* d_partials;
    double* h_partials;  // pinned
    // completion flag of the zero-copy host transport: written in stream order after the kernel (hipStreamWriteValue32),
    // polled by the host instead of a hipStreamSynchronize (see wait_zero_copy)
    volatile uint32_t* h_flag;  // pinned + mapped
    void* d_flag;               // its device address
    uint32_t flag_seq;
    bool flag_tried;            // ensure_flag ran (the flag is allocated on first need)
    bool flag_posted;           // the current flag_seq is already on the stream (drone_vec_step_send)
    bool pending;               // a step was sent and not yet received
    float* d_obs;        // host-buffer handles: device mirrors; device handles: the library-owned buffers, if any
    float* d_act;
    float* d_rew;
    unsigned char* d_term;
    unsigned char* d_trunc;
    // zero_copy: pinned + mapped stand-ins, owned here, for the caller buffers that could not be pinned themselves
    // (slot order: observations, actions, rewards, terminals, truncations; null = the caller's buffer is mapped directly).
    // The kernel reads / writes the stand-in over PCIe; the host copies between it and the caller's buffer around the step.
    void* bounce[5];
    size_t bounce_bytes[5];
    // zero_copy with stand-ins too large for one memcpy around the step (round 5, drone_vec_host_transport 3): the host copy
    // pool moves them — the action rows in as parallel slices, the outputs out WHILE the step kernel runs, chunk by chunk as
    // its workgroups raise their words in h_wg_done (LaunchSig::wg_done)
    bool threaded;
    uint32_t* h_wg_done;  // pinned + mapped: one word per 256-drone chunk, env order
    uint32_t* d_wg_done;  // its device address
    uint32_t n_wg;
    uint32_t wg_seq;      // what a chunk's word reads once the CURRENT step's rows of that chunk have landed
    bool copy_started;    // the pool is delivering this handle's outputs (from step_send until step_recv / the end of step)
    int stream_idle;      // set (atomically) by the calling thread once the stream is known to have drained: nothing is left to wait for, copy the rest
    int copy_abort;       // ... or to have failed: stop
    float* m_obs;        // device-visible addresses of the caller's registered host buffers or of their stand-ins (zero_copy)
    float* m_act;
    float* m_rew;
    unsigned char* m_term;
    unsigned char* m_trunc;
    hipStream_t stream;
    bool own_stream;
    hipEvent_t ev0, ev1;
    // done-id list (compact_done): the counter slot is keyed on the number of STEP launches, not on gstep, so a
    // fused rollout (which advances gstep but builds no list) cannot desynchronise the ping-pong
    uint32_t step_launches;
    bool list_valid;     // the last path call was drone_vec_step
    // drone_vec_step_many: per-step done-id lists [many_cap][n] + counts [many_cap] (compact_done), device staging of the
    // K-major blocks (host-buffer handles), all grown on demand
    uint32_t* many_ids;
    uint32_t* many_count;
    int many_cap;        // steps the list storage holds
    int many_k;          // k_steps of the last drone_vec_step_many, 0 if the last path call was something else
    float* s_act; float* s_obs; float* s_rew; unsigned char* s_term; unsigned char* s_trunc;
    int stage_cap;       // steps the staging blocks hold
    // blocks drone_vec_host_pin registered on this handle (and only those: host_unpin drops nothing else)
    void* pinned_blocks[64];
    int n_pinned_blocks;
    char variant[448];   // drone_vec_variant
    // DRONE_AUTOTUNE=1: the sweep order / load hints of an HBM-bound handle are measured on the box it runs on, under the
    // workload it runs, during its first few hundred real steps (struct SweepTune); null otherwise — the footprint table of
    // drone_vec_init decides (round 6: opt-in, the measurement re-derived the table in 12 of 12 logged cases)
    struct SweepTune* tune;
    size_t touched_mib;  // MiB one step touches (the footprint the table is indexed by)
    // sticky status: the first failure of any call on this handle (drone_vec_status)
    int status;
    char status_msg[512];
    struct Gather* gather;  // host-boundary exchange (RCCL: drone_vec_gather_init[_root]; peer stores: drone_vec_gather_init_peer), or null
    // peer-store exchange: the global buffers this (root) handle exported with drone_vec_gather_peer_export
    float* px_obs; float* px_rew; unsigned char* px_term; unsigned char* px_trunc;
};

namespace {

// Every entry point that takes a handle opens with one of these: clears the
// calling thread's error text, switches to the handle's device and puts the
// caller's device back on the way out (a process that drives several GPUs, or
// torch with another current device, must not find its device changed by a
// step()); a failure anywhere inside the call sticks to the handle
// (drone_vec_status) because the path calls themselves return void.
// puts the caller's current device back when the scope ends (init, which has no handle yet)
struct DeviceRestore {
    int prev = -1;
    DeviceRestore() { if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; } }
    ~DeviceRestore() { if (prev >= 0) (void)hipSetDevice(prev); }
};

struct Entry {
    DroneVec* v;
    int prev = -1;
    bool ok = false;
    explicit Entry(const DroneVec* cv) : v(const_cast<DroneVec*>(cv)) {
        g_err[0] = 0;
        if (!v) { set_err("handle is NULL"); return; }
        if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; }
        if (prev != v->device) {
            hipError_t e = hipSetDevice(v->device);
            if (e != hipSuccess) { set_err("hipSetDevice(%d) failed: %s", v->device, hipGetErrorString(e)); return; }
        }
        ok = true;
    }
    ~Entry() {
        if (v && g_err[0] && v->status == 0) {
            v->status = 1;
            snprintf(v->status_msg, sizeof(v->status_msg), "%s", g_err);
        }
        if (v && prev >= 0 && prev != v->device) (void)hipSetDevice(prev);
    }
    explicit operator bool() const { return ok; }
};

// Plane stride padding, in float4 elements (DRONE_PLANE_PAD, tuning experiments
// only). With a power-of-two env count the planes sit exactly 2^k bytes apart;
// the HBM address hash copes: pads of 16...65552 elements measure within +-0.7 %
// of no pad at equal placement (profiles/r01_ab/ab14_pad.txt), so the default is 0.
uint32_t plane_pad_elems() {
    const char* e = getenv("DRONE_PLANE_PAD");
    if (e && *e) return (uint32_t)strtoul(e, nullptr, 10);
    return 0;
}

// graph-safe stepping: the device copies of the counters ({gstep, step launches, arrivals}); the host fields mirror them
bool push_counters(DroneVec* v) {
    if (!v->dv.ctr) return true;
    const uint32_t c[3] = {v->gstep, v->step_launches, 0u};
    HIP_TRY(hipMemcpyAsync(v->dv.ctr, c, sizeof(c), hipMemcpyHostToDevice, v->stream), return false);
    HIP_TRY(hipStreamSynchronize(v->stream), return false);  // `c` is on the stack
    return true;
}
bool pull_counters(DroneVec* v) {
    if (!v->dv.ctr) return true;
    uint32_t c[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(c, v->dv.ctr, sizeof(c), hipMemcpyDeviceToHost, v->stream), return false);
    HIP_TRY(hipStreamSynchronize(v->stream), return false);
    if (c[1] != v->step_launches) v->list_valid = true;  // steps ran (e.g. graph replays) since the host last looked
    v->gstep = c[0];
    v->step_launches = c[1];
    return true;
}

bool upload_params(DroneVec* v) {
    derive_kparams(v->cfg, v->seed, v->kp);
    HIP_TRY(hipMemcpyAsync(v->d_kp, &v->kp, sizeof(KParams), hipMemcpyHostToDevice, v->stream), return false);
    // the source is the handle's own field; make the copy complete before it can change again
    HIP_TRY(hipStreamSynchronize(v->stream), return false);
    return true;
}

bool validate(const DroneConfig* c, int num_envs) {
    if (!c) { set_err("config is NULL"); return false; }
    if (c->struct_size != sizeof(DroneConfig)) { set_err("DroneConfig.struct_size %u != %zu", c->struct_size, sizeof(DroneConfig)); return false; }
    if (num_envs <= 0) { set_err("num_envs must be positive"); return false; }
    if (c->task != DRONE_TASK_HOVER && c->task != DRONE_TASK_WAYPOINT && c->task != DRONE_TASK_SWARM && c->task != DRONE_TASK_RACE) { set_err("unknown task %d", c->task); return false; }
    if (c->task == DRONE_TASK_SWARM) {
        const int A = c->agents_per_env;
        if (A < 1 || A > 64 || (A & (A - 1))) { set_err("agents_per_env must be a power of two in [1, 64], got %d", A); return false; }
        if (num_envs % A || c->env_offset % (uint32_t)A) { set_err("num_envs and env_offset must be multiples of agents_per_env (%d)", A); return false; }
        if (!(c->proximity_radius > 0.0f)) { set_err("proximity_radius must be positive"); return false; }
    }
    {   // state addressing in the kernels is 32-bit: the hot region's n_pad x planes-per-tile float4 elements must fit
        const uint64_t n_pad = ((uint64_t)num_envs + kBlock - 1) / kBlock * kBlock;
        if (n_pad * 7u > 0xFFFFFFFFull || 2u * (n_pad + plane_pad_elems()) > 0xFFFFFFFFull) {
            set_err("num_envs %d too large: the state region must fit 32-bit element indices (max about %llu envs per handle; shard further)",
                    num_envs, (unsigned long long)(0xFFFFFFFFull / 7u - kBlock));
            return false;
        }
    }
    if (c->state_layout != DRONE_LAYOUT_AUTO && c->state_layout != DRONE_LAYOUT_TARGET_PLANE && c->state_layout != DRONE_LAYOUT_DERIVED_TARGET) { set_err("unknown state_layout %d", c->state_layout); return false; }
    if (c->state_layout == DRONE_LAYOUT_DERIVED_TARGET && !(task_has_derived_target(c->task) && c->horizon <= 65535)) {
        set_err("state_layout = DRONE_LAYOUT_DERIVED_TARGET needs the hover or swarm task and horizon <= 65535 (task %d, horizon %d)", c->task, c->horizon);
        return false;
    }
    if (c->buffer_kind != DRONE_BUFFERS_HOST && c->buffer_kind != DRONE_BUFFERS_DEVICE) { set_err("unknown buffer_kind %d", c->buffer_kind); return false; }
    if (c->substeps < 1 || c->horizon < 1) { set_err("substeps and horizon must be >= 1"); return false; }
    if (!(c->dt > 0.0f) || !(c->mass > 0.0f) || !(c->ixx > 0.0f) || !(c->iyy > 0.0f) || !(c->izz > 0.0f) || !(c->motor_tau > 0.0f) ||
        !(c->max_rpm > 0.0f) || !(c->max_vel > 0.0f) || !(c->max_omega > 0.0f) || !(c->bound > 0.0f) || !(c->k_thrust > 0.0f)) {
        set_err("physical constants must be positive");
        return false;
    }
    if (c->task == DRONE_TASK_WAYPOINT && !(c->wind_max > 0.0f)) { set_err("wind_max must be positive (SPEC.md §4: clamp bounds are never zero)"); return false; }
    return true;
}

// ---- the host copy pool's jobs (transport 3) ----
// a slice of `bytes` for part `part` of `parts`, cut at 4 KiB so that no two threads share a page
void slice_of(size_t bytes, int part, int parts, size_t& begin, size_t& end) {
    const size_t per = ((bytes + (size_t)parts - 1) / (size_t)parts + 4095u) & ~(size_t)4095u;
    begin = per * (size_t)part < bytes ? per * (size_t)part : bytes;
    end = begin + per < bytes ? begin + per : bytes;
}
void copy_actions_part(void* ctx, int part, int parts) {
    DroneVec* v = static_cast<DroneVec*>(ctx);
    size_t b, e;
    slice_of(v->bounce_bytes[1], part, parts, b, e);
    if (b < e) memcpy(static_cast<char*>(v->bounce[1]) + b, reinterpret_cast<const char*>(v->u_act) + b, e - b);
}
// every stand-in of an output buffer, whole (reset, rollout: launches that raise no per-chunk words)
void copy_all_outputs_part(void* ctx, int part, int parts) {
    DroneVec* v = static_cast<DroneVec*>(ctx);
    void* const dst[5] = {v->u_obs, nullptr, v->u_rew, v->u_term, v->u_trunc};
    for (int k : {0, 2, 3, 4}) {
        if (!v->bounce[k]) continue;
        size_t b, e;
        slice_of(v->bounce_bytes[k], part, parts, b, e);
        if (b < e) memcpy(static_cast<char*>(dst[k]) + b, static_cast<const char*>(v->bounce[k]) + b, e - b);
    }
}
// rows of the chunks [c0, c1) of every output stand-in
void copy_chunks(DroneVec* v, uint32_t c0, uint32_t c1) {
    const size_t n = (size_t)v->n, od = (size_t)drone_obs_dim(v->cfg.task) * sizeof(float);
    const size_t r0 = (size_t)c0 * kBlock, r1 = (size_t)c1 * kBlock < n ? (size_t)c1 * kBlock : n;
    if (r0 >= r1) return;
    if (v->bounce[0]) memcpy(reinterpret_cast<char*>(v->u_obs) + r0 * od, static_cast<const char*>(v->bounce[0]) + r0 * od, (r1 - r0) * od);
    if (v->bounce[2]) memcpy(v->u_rew + r0, static_cast<const float*>(v->bounce[2]) + r0, (r1 - r0) * sizeof(float));
    if (v->bounce[3]) memcpy(v->u_term + r0, static_cast<const unsigned char*>(v->bounce[3]) + r0, r1 - r0);
    if (v->bounce[4]) memcpy(v->u_trunc + r0, static_cast<const unsigned char*>(v->bounce[4]) + r0, r1 - r0);
}
#ifndef DRONE_HOST_STAMPS
#define DRONE_HOST_STAMPS 0  // diagnostic build: where a transport-3 step's microseconds go on the host (tools/host_timeline.py)
#endif
#if DRONE_HOST_STAMPS
struct HostStamps {
    enum { kEnter, kActionsIn, kLaunched, kPoolStarted, kFirstChunkSeen, kLastChunkCopied, kPoolFinished, kFlagSeen, kCount };
    double sum[kCount] = {};
    uint64_t steps = 0;
    double t0 = 0, first_seen[64], last_done[64];
    static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; }
    void enter() { t0 = now(); for (int k = 0; k < 64; k++) { first_seen[k] = 1e300; last_done[k] = 0; } }
    void at(int what) { sum[what] += now() - t0; }
    void fold() {
        double f = 1e300, l = 0;
        for (int k = 0; k < 64; k++) { if (first_seen[k] < f) f = first_seen[k]; if (last_done[k] > l) l = last_done[k]; }
        if (l > 0) { sum[kFirstChunkSeen] += f - t0; sum[kLastChunkCopied] += l - t0; }
        steps++;
    }
    ~HostStamps() {
        if (!steps) return;
        static const char* names[kCount] = {"enter", "actions_in", "launched", "pool_started", "first_chunk_seen", "last_chunk_copied", "pool_finished", "flag_seen"};
        fprintf(stderr, "[drone host stamps] %llu steps, us from entry:", (unsigned long long)steps);
        for (int k = 1; k < kCount; k++) fprintf(stderr, " %s=%.2f", names[k], sum[k] / (double)steps);
        fprintf(stderr, "\n");
    }
};
HostStamps g_stamps;
#define HOST_STAMP(what) g_stamps.at(HostStamps::what)
#else
#define HOST_STAMP(what) ((void)0)
#endif

// what the thread that called the step knows about the stream while the outputs are being delivered: drained = everything has
// landed, nobody needs to look at the words any more; failed = everybody must stop
void poll_stream(void* ctx) {
    DroneVec* h = static_cast<DroneVec*>(ctx);
    const hipError_t q = hipStreamQuery(h->stream);
    if (q == hipSuccess) __atomic_store_n(&h->stream_idle, 1, __ATOMIC_RELEASE);
    else if (q != hipErrorNotReady) { (void)hipGetLastError(); __atomic_store_n(&h->copy_abort, 1, __ATOMIC_RELEASE); }
}

// The step's outputs, while the kernel runs: this thread owns a contiguous share of the chunks and copies every run of
// chunks whose words have turned to the step's sequence number. The words are an accelerator, not the contract: once the
// stream is known to have drained (stream_idle) everything has landed and the rest is copied without looking. Part 0 runs on
// the CALLING thread (CopyPool::finish) ahead of the pool's own watch: while it is stalled it polls the stream itself, so a
// chunk word that never comes (a failed launch, a fault) ends in an error instead of a spin.
void copy_outputs_part(void* ctx, int part, int parts) {
    DroneVec* v = static_cast<DroneVec*>(ctx);
    const uint32_t c0 = (uint32_t)((uint64_t)v->n_wg * (uint32_t)part / (uint32_t)parts), c1 = (uint32_t)((uint64_t)v->n_wg * (uint32_t)(part + 1) / (uint32_t)parts);
    const uint32_t seq = v->wg_seq;
    bool idle = false;
    uint32_t stalls = 0;
    for (uint32_t c = c0; c < c1;) {
        uint32_t e = c;
        while (e < c1 && (idle || __atomic_load_n(v->h_wg_done + e, __ATOMIC_ACQUIRE) == seq)) e++;
        if (e == c) {
            if (__atomic_load_n(&v->copy_abort, __ATOMIC_ACQUIRE)) return;
            if (__atomic_load_n(&v->stream_idle, __ATOMIC_ACQUIRE)) idle = true;
            else {
                if (part == 0 && (++stalls & 255u) == 0) poll_stream(v);
                CopyPool::cpu_relax();
            }
            continue;
        }
#if DRONE_HOST_STAMPS
        if (g_stamps.first_seen[part & 63] > 1e299) g_stamps.first_seen[part & 63] = HostStamps::now();
#endif
        copy_chunks(v, c, e);
        c = e;
    }
#if DRONE_HOST_STAMPS
    g_stamps.last_done[part & 63] = HostStamps::now();
#endif
}

bool host_to_device_actions(DroneVec* v) {
    if (v->zero_copy) {  // the kernel reads the caller's action buffer itself, or its pinned stand-in
        if (v->bounce[1]) {
            if (v->threaded) CopyPool::get().run(copy_actions_part, v);
            else memcpy(v->bounce[1], v->u_act, v->bounce_bytes[1]);
        }
        return true;
    }
    HIP_TRY(hipMemcpyAsync(v->d_act, v->u_act, (size_t)v->n * DRONE_ACT_DIM * sizeof(float), hipMemcpyHostToDevice, v->stream), return false);
    return true;
}

// Zero-copy transport: the kernel has written the caller's buffers itself, so all that is left is to learn that it has
// finished. A 32-bit sequence number written to pinned host memory in stream order right behind the kernel, and polled
// here, tells the host as soon as the write lands; hipStreamSynchronize goes through the runtime's signal wait instead.
// Polling is bounded: a kernel that takes longer than the spin budget (large shards — where the wait's latency no longer
// matters — or a fault, which only the runtime can report) falls back to hipStreamSynchronize. DRONE_HOST_SPIN=0 turns
// the flag off. Measured with host/drone_host --fill 0 on one box: 15.4 -> 13.2 us per step at 256 envs, 16.7 -> 15.1 at
// 1 024, 22.8 -> 19.6 at 4 096, 43.3 -> 41.1 at 16 384, no difference from 65 536 on (the step is PCIe-bound there).
//
// ensure_flag allocates the flag (best effort: without it the waits are hipStreamSynchronize);
// post_flag puts the next sequence number on the stream, behind everything enqueued so far; wait_zero_copy polls for it.
void ensure_flag(DroneVec* v) {
    if (v->h_flag || v->flag_tried) return;
    v->flag_tried = true;
    const char* sp = getenv("DRONE_HOST_SPIN");
    if (sp && *sp && atoi(sp) == 0) return;
    void* hf = nullptr;
    if (hipHostMalloc(&hf, 64, hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(&v->d_flag, hf, 0) == hipSuccess) {
        v->h_flag = static_cast<volatile uint32_t*>(hf);
        *v->h_flag = 0u;
    } else {
        (void)hipGetLastError();
        if (hf) (void)hipHostFree(hf);
        v->d_flag = nullptr;
    }
}

void post_flag(DroneVec* v) {
    v->flag_posted = false;
    if (!v->h_flag) return;
    const uint32_t seq = v->flag_seq + 1u;
    if (hipStreamWriteValue32(v->stream, v->d_flag, seq, 0) == hipSuccess) {
        v->flag_seq = seq;
        v->flag_posted = true;
    } else {
        (void)hipGetLastError();
    }
}

bool wait_zero_copy(DroneVec* v) {
    if (!v->flag_posted) post_flag(v);
    if (v->flag_posted) {
        v->flag_posted = false;
        const uint32_t seq = v->flag_seq;
        for (uint32_t spins = 0; spins < (1u << 16); spins++) {
            if (__atomic_load_n(v->h_flag, __ATOMIC_ACQUIRE) == seq) return true;
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#endif
        }
    }
    HIP_TRY(hipStreamSynchronize(v->stream), return false);
    return true;
}

// The two halves of handing a host caller its outputs: what can be put on the stream right behind the kernel (the
// completion flag, or the mirror transport's D2H copies), and the wait plus whatever the host has to copy itself.
// drone_vec_step_send / drone_vec_step_recv run them apart; every synchronous path call runs them back to back.
bool enqueue_host_outputs(DroneVec* v) {
    if (v->zero_copy) { post_flag(v); return true; }
    const size_t n = (size_t)v->n;
    HIP_TRY(hipMemcpyAsync(v->u_obs, v->d_obs, n * (size_t)drone_obs_dim(v->cfg.task) * sizeof(float), hipMemcpyDeviceToHost, v->stream), return false);
    HIP_TRY(hipMemcpyAsync(v->u_rew, v->d_rew, n * sizeof(float), hipMemcpyDeviceToHost, v->stream), return false);
    HIP_TRY(hipMemcpyAsync(v->u_term, v->d_term, n, hipMemcpyDeviceToHost, v->stream), return false);
    HIP_TRY(hipMemcpyAsync(v->u_trunc, v->d_trunc, n, hipMemcpyDeviceToHost, v->stream), return false);
    return true;
}

// the calling thread's part of a threaded copy-out, then the wait for the helpers — during which it keeps an eye on the stream:
// drained = everything has landed, the helpers need not look at the words any more; failed = they must stop
bool finish_threaded_copy(DroneVec* v) {
    CopyPool::get().finish(poll_stream, v);
    v->copy_started = false;
    if (__atomic_load_n(&v->copy_abort, __ATOMIC_ACQUIRE)) { set_err("the stream failed while the step's outputs were being delivered"); return false; }
    return true;
}

bool finish_host_outputs(DroneVec* v) {
    if (v->zero_copy) {  // outputs already landed in the caller's memory (or its stand-ins): just wait for the kernel
        if (v->copy_started) {  // transport 3, a step: the pool has been copying chunks out since the launch
            const bool ok = finish_threaded_copy(v);
            HOST_STAMP(kPoolFinished);
            const bool landed = wait_zero_copy(v);
            HOST_STAMP(kFlagSeen);
#if DRONE_HOST_STAMPS
            g_stamps.fold();
#endif
            return landed && ok;
        }
        if (!wait_zero_copy(v)) return false;
        if (v->threaded) {
            CopyPool::get().run(copy_all_outputs_part, v);
            return true;
        }
        if (v->bounce[0]) memcpy(v->u_obs, v->bounce[0], v->bounce_bytes[0]);
        if (v->bounce[2]) memcpy(v->u_rew, v->bounce[2], v->bounce_bytes[2]);
        if (v->bounce[3]) memcpy(v->u_term, v->bounce[3], v->bounce_bytes[3]);
        if (v->bounce[4]) memcpy(v->u_trunc, v->bounce[4], v->bounce_bytes[4]);
        return true;
    }
    HIP_TRY(hipStreamSynchronize(v->stream), return false);
    return true;
}

bool device_to_host_outputs(DroneVec* v) { return enqueue_host_outputs(v) && finish_host_outputs(v); }

// every path / plumbing call except drone_vec_step_recv and close: not while a sent step is in flight
bool idle(DroneVec* v, const char* what) {
    if (!v->pending) return true;
    set_err("%s: a step sent with drone_vec_step_send has not been received (drone_vec_step_recv)", what);
    return false;
}

// Host-buffer mode has two transports. Mirror: actions H2D, kernel on device
// mirrors, four D2H copies. Zero-copy: host memory is mapped into the device
// address space and the kernel loads the actions and stores its outputs through
// PCIe itself — no copy commands at all, which is what small vec-envs
// (launch / copy-latency bound) want. What gets mapped is, per buffer, the
// caller's own memory where it may be pinned (pin_caller_buffer) and a pinned
// stand-in owned by the handle where it may not (DroneVec::bounce: copied to /
// from the caller's memory on the host around the step, so only while that is
// cheaper than the mirror's DMA copies). Chosen at init: zero-copy when every
// buffer is mapped one way or the other and 16-B aligned and the shard is at
// most DRONE_ZERO_COPY_MAX_ENVS envs; DRONE_HOST_ZEROCOPY=0/1 forces it.
constexpr int kZeroCopyMaxEnvsDefault = 1 << 30;

bool want_zero_copy(int num_envs) {
    const char* e = getenv("DRONE_HOST_ZEROCOPY");
    if (e && *e) return atoi(e) != 0;
    const char* m = getenv("DRONE_ZERO_COPY_MAX_ENVS");
    const long cap = (m && *m) ? atol(m) : (long)kZeroCopyMaxEnvsDefault;
    return num_envs <= cap;
}

void* mapped_ptr(void* host) {
    void* d = nullptr;
    if (hipHostGetDevicePointer(&d, host, 0) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return d;
}

// A caller that rebinds away from a buffer we pinned at init may free it right away; a pin left on freed pages makes
// any later copy that touches those addresses again fail ("invalid argument": a range that is only partly pinned).
void unpin_if_rebound(DroneVec* v, int slot, const void* now) {
    if (v->registered[slot] && v->registered_ptr[slot] != now) {
        host_unregister(v->registered_ptr[slot], v, "rebound buffer");
        v->registered[slot] = false;
    }
}

void drop_bounce(DroneVec* v) {
    v->threaded = false;
    for (int k = 0; k < 5; k++) {
        if (v->bounce[k]) (void)hipHostFree(v->bounce[k]);
        v->bounce[k] = nullptr;
        v->bounce_bytes[k] = 0;
    }
}

void leave_zero_copy(DroneVec* v) {
    v->zero_copy = false;
    drop_bounce(v);  // (the caller syncs the stream before anything reuses the mirrors: Entry-guarded calls only)
    v->dv.obs = v->d_obs; v->dv.act = v->d_act; v->dv.rew = v->d_rew; v->dv.term = v->d_term; v->dv.trunc = v->d_trunc;
}

// Which caller host buffers may be pinned. hipHostRegister works at page granularity, and on ROCm 7 registering (and
// later unregistering) a range that shares a page with OTHER heap memory breaks the runtime's own on-the-fly pinning of
// pageable copy destinations on that page: a later hipMemcpy / torch .cpu() into a neighbouring allocation dies with
// "Memory access fault by GPU ... on address <heap address>" (tools/debug/pageable_copy_stress.py reproduces it with
// plain HIP calls; ~1 in 12 runs of this repo's GPU test suite hit it before this rule). Round 5 found the rule of rounds
// 3-4 — "starts on a page boundary and spans whole pages" — still too generous: such a block INSIDE the malloc heap (a
// numpy array that happens to start on a page boundary, a posix_memalign block) owns its pages but not its mapping, and
// when the heap around it is trimmed or reused while the GPU writes the registered pages the same fault appears ("Write
// access to a read-only page": tools/debug/heap_interior_registration_stress.py, library-free; two of eight soak runs died
// of it once heap-buffer handles took the zero-copy transports more often). A mapping of its own (mmap, POSIX shm) under
// the same stress never faults. The library cannot tell the two apart, so alignment alone no longer suffices: a buffer is
// registered only when the CALLER vouches for it (cfg.host_pages_exclusive: every buffer is a mapping of its own, page-
// aligned, nothing else in its pages) or has pinned it itself (hipHostMalloc, hipHostRegister). Everything else is left
// alone and goes through stand-ins or plain pageable copies.
constexpr uintptr_t kPage = 4096;

// Pinned by its owner (hipHostMalloc / hipHostRegister) over ALL of [p, p + bytes): the first and the last byte are both
// pinned host memory and map to device addresses exactly bytes - 1 apart, i.e. one mapping covers the block. (ADVICE r3:
// looking at the first byte only accepted a slice that starts inside someone's registration and ends outside it; the
// kernel then faulted on the tail instead of the call falling back to staging.)
bool already_pinned(const void* p, size_t bytes) {
    if (!p || !bytes) return false;
    hipPointerAttribute_t a0, a1;
    if (hipPointerGetAttributes(&a0, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (a0.type != hipMemoryTypeHost) return false;
    if (bytes == 1) return true;
    const char* last = static_cast<const char*>(p) + (bytes - 1);
    if (hipPointerGetAttributes(&a1, last) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (a1.type != hipMemoryTypeHost) return false;
    if (a0.devicePointer && a1.devicePointer)
        return static_cast<const char*>(a1.devicePointer) - static_cast<const char*>(a0.devicePointer) == (ptrdiff_t)(bytes - 1);
    return true;
}

// returns true if the buffer ends up pinned (by us: v->registered[slot]; or by its owner)
bool pin_caller_buffer(DroneVec* v, int slot, void* p, size_t bytes) {
    v->registered[slot] = false;
    v->registered_ptr[slot] = p;
    if (already_pinned(p, bytes)) return true;
    const bool aligned = (reinterpret_cast<uintptr_t>(p) % kPage) == 0;
    if (!aligned || !v->cfg.host_pages_exclusive) return false;
    const size_t span = (bytes + kPage - 1) / kPage * kPage;
    v->registered[slot] = (host_register(p, span, v, "caller buffer") == hipSuccess);
    if (!v->registered[slot]) (void)hipGetLastError();
    return v->registered[slot];
}

// ---------------------------------------------------------------------------
// Host-boundary all-gather over RCCL (BASELINE.json north_star:
// "RCCL gather of obs/rewards over xGMI only at the host boundary"). The env
// path itself has no collective; this is the one exchange step, for a consumer
// that wants every rank's observations / rewards / flags in one buffer.
// librccl is dlopen'ed on first use so that single-GPU users never load it; in
// a process where torch already mapped its librccl.so.1 the same copy is reused.
// ---------------------------------------------------------------------------
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

// Loaded once per process, whichever thread gets there first (callers may drive one handle per host thread): the
// table is filled under std::call_once and is read-only afterwards; a failed load is remembered with its reason.
Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    static char why[256] = "";
    std::call_once(once, [] {
        // DRONE_RCCL_LIB: another library with the same entry points (tests/rccl_stub: lets several ranks share one
        // GPU, which RCCL itself refuses)
        const char* alt = getenv("DRONE_RCCL_LIB");
        const char* names[] = {alt && *alt ? alt : "librccl.so.1", "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        void* lib = nullptr;
        for (const char* n : names) {
            lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (lib || (alt && *alt)) break;  // an explicit choice is not silently replaced
        }
        if (!lib) { const char* e = dlerror(); snprintf(why, sizeof(why), "dlopen(librccl.so.1) failed: %s", e ? e : "?"); return; }
#define RCCL_SYM(field, name)                                                        \
        r.field = reinterpret_cast<decltype(r.field)>(dlsym(lib, name));             \
        if (!r.field) { snprintf(why, sizeof(why), "librccl has no symbol %s", name); dlclose(lib); return; }
        RCCL_SYM(GetUniqueId, "ncclGetUniqueId")
        RCCL_SYM(CommInitRank, "ncclCommInitRank")
        RCCL_SYM(CommDestroy, "ncclCommDestroy")
        RCCL_SYM(AllGather, "ncclAllGather")
        RCCL_SYM(Broadcast, "ncclBroadcast")
        RCCL_SYM(Send, "ncclSend")
        RCCL_SYM(Recv, "ncclRecv")
        RCCL_SYM(GroupStart, "ncclGroupStart")
        RCCL_SYM(GroupEnd, "ncclGroupEnd")
        RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef RCCL_SYM
        r.lib = lib;  // last: a non-null lib means every entry point is bound
    });
    if (!r.lib) { set_err("%s", why[0] ? why : "librccl could not be loaded"); return nullptr; }
    return &r;
}

#define RCCL_TRY(R, expr, onfail)                                                          \
    do {                                                                                   \
        ncclResult_t r_ = (expr);                                                          \
        if (r_ != ncclSuccess) {                                                           \
            set_err("%s failed: %s (%s:%d)", #expr, (R)->GetErrorString(r_), __FILE__, __LINE__); \
            onfail;                                                                        \
        }                                                                                  \
    } while (0)

}  // namespace

struct Gather {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    int root = -1;  // >= 0: gather to that rank only (ncclSend / ncclRecv); -1: all-gather, every rank receives the batch
    std::vector<size_t> counts, offsets;  // envs per rank, first global row of each rank
    size_t total = 0;
    bool equal = true;
    // device-side global buffers: the caller's (device-buffer handles) or staging owned here (host-buffer handles)
    float* g_obs = nullptr;
    float* g_rew = nullptr;
    unsigned char* g_term = nullptr;
    unsigned char* g_trunc = nullptr;
    bool own_staging = false;
    // host-buffer handles: where the gathered batch is copied to after the collective
    float* h_obs = nullptr;
    float* h_rew = nullptr;
    unsigned char* h_term = nullptr;
    unsigned char* h_trunc = nullptr;
    bool h_registered[4] = {false, false, false, false};  // the global host buffers pinned by gather_init
    // ---- peer-store exchange (round 4; drone_vec_gather_init_peer): no collective. The root exported its global
    // buffers as IPC handles; every other rank mapped them and bound its OUTPUT pointers to its rows in them, so its step
    // kernel's stores land in the root's HBM over xGMI. What is left of the "gather" is a handshake through a page of
    // flags in host memory shared by the ranks: post[r] = launches rank r has published, ack = rounds the root has consumed.
    bool peer = false;
    void* peer_base[4] = {nullptr, nullptr, nullptr, nullptr};  // IPC mappings opened here (non-root ranks)
    volatile uint32_t* flags = nullptr;  // the shared page: post[world] then ack
    char* d_flags = nullptr;             // its device address (hipStreamWriteValue32 / hipStreamWaitValue32)
    bool flags_registered = false;
    bool stream_writes = false;          // DRONE_PEER_STREAM_WRITES=1: publish flags with hipStreamWriteValue32 instead of the one-wave kernel (measured SLOWER: see peer_post)
    bool gpu_waits = true;               // the handshake runs on the stream (two one-wave kernels); false (DRONE_PEER_HOST_WAIT=1): the host drains the stream and polls / stores
    uint32_t* h_err = nullptr;           // pinned + mapped word a stream-side wait sets when it gave up (a dead peer)
    uint32_t* d_err = nullptr;
    unsigned long long budget_ticks = 0; // of the 100 MHz real-time counter
    uint32_t seq = 0;                    // rounds this rank has published (non-root) / collected (root)
    uint32_t acked = 0;                  // root: last round whose consumption it has announced
    // round 5: the two publications ride on the launch that writes the outputs (drone_kernels.h LaunchSig) instead of being
    // one-wave launches of their own. DRONE_PEER_INKERNEL=0 keeps the separate launches (A/B; also what the host-side and
    // hipStreamWriteValue32 forms use).
    bool in_kernel = true;
    uint32_t* d_arrive = nullptr;        // HBM, the peer block (drone_kernels.h LaunchSig): [0] the arrival counter of the in-kernel post, [kPeerStopWord] the stop word
                                         // a stream-side wait raises when it gives up; allocated whenever the waits run on the stream
    uint32_t launch_posts = 0;           // non-root: the round the LAST output-writing launch publishes by itself when it ends (0: none)
    bool launched = false;               // an output-writing launch has gone out since the last drone_vec_gather: the next one must be the gather
    uint32_t own_order = 0;              // the handle's sweep order / load hints before the exchange (the peer instantiations carry no load hints)
    float* own_obs = nullptr; float* own_rew = nullptr; unsigned char* own_term = nullptr; unsigned char* own_trunc = nullptr;  // the handle's output bindings before the exchange took them over
};

// (defined at global scope like Gather: DroneVec names it)
struct SweepTune {
    static constexpr int kStart = 160, kBurst = 16, kLead = 4, kRounds = 2, kPairs = 48;
    uint32_t cand[4];
    int nc = 0;
    uint32_t table = 0;
    long seen = 0;            // step launches of this handle so far
    double sum_ms[4] = {0, 0, 0, 0};
    int samples[4] = {0, 0, 0, 0};
    struct Pair { hipEvent_t e0 = nullptr, e1 = nullptr; int cand = -1; } pairs[kPairs];
    int in_flight = 0;
};

namespace {

void write_variant(DroneVec* v, const char* tuned);

void gather_destroy(DroneVec* v) {
    Gather* g = v->gather;
    if (!g) return;
    if (g->comm) {
        Rccl* R = rccl();
        if (R) (void)R->CommDestroy(g->comm);
    }
    if (g->peer) {  // give the handle its own output buffers back, then drop the mappings
        if (v->stream) (void)hipStreamSynchronize(v->stream);
        v->dv.obs = g->own_obs; v->dv.rew = g->own_rew; v->dv.term = g->own_term; v->dv.trunc = g->own_trunc;
        if (v->dv.order != g->own_order) { v->dv.order = g->own_order; write_variant(v, nullptr); }
        for (int k = 0; k < 4; k++)
            if (g->peer_base[k]) (void)hipIpcCloseMemHandle(g->peer_base[k]);
        if (g->flags_registered) host_unregister(const_cast<uint32_t*>(g->flags), v, "peer-store flag page");
        if (g->h_err) (void)hipHostFree(g->h_err);
        if (g->d_arrive) (void)hipFree(g->d_arrive);
        // the export is consumed: a later drone_vec_gather_init_peer needs a fresh drone_vec_gather_peer_export (ADVICE r4:
        // stale pointers here would be reused for buffers the caller may have freed since)
        v->px_obs = nullptr; v->px_rew = nullptr; v->px_term = nullptr; v->px_trunc = nullptr;
    }
    void* hosts[4] = {g->h_obs, g->h_rew, g->h_term, g->h_trunc};
    for (int k = 0; k < 4; k++)
        if (g->h_registered[k]) host_unregister(hosts[k], v, "global gather buffer");
    if (g->own_staging) {
        (void)hipFree(g->g_obs);
        (void)hipFree(g->g_rew);
        (void)hipFree(g->g_term);
        (void)hipFree(g->g_trunc);
    }
    delete g;
    v->gather = nullptr;
}

// ---- peer-store exchange: the handshake ----
// Flags are words of a host-memory page shared by the ranks' processes; the counters only grow and are compared as signed
// differences (a wrap after 2^32 rounds is harmless). Waiting: on the stream, ONE one-wave kernel whose lanes poll the
// flags waited for (hipStreamWaitValue32 cannot: it takes only the calling process's signal memory), each lane giving up
// after the time budget and raising the handle's error word; or, with DRONE_PEER_HOST_WAIT=1, on the host (drain the
// stream, poll), where the same budget (DRONE_PEER_TIMEOUT_MS, default 10 s) makes a dead peer an immediate error.
// DRONE_PEER_TIMEOUT_MS, clamped to [1 ms, 10 min]; anything malformed, zero or negative is the default (0 made every
// wait give up at once, a negative value became a budget of centuries)
long peer_timeout_ms() {
    const char* t = getenv("DRONE_PEER_TIMEOUT_MS");
    if (!t || !*t) return 10000;
    char* end = nullptr;
    const long ms = strtol(t, &end, 10);
    if (end == t || *end != 0 || ms < 1) return 10000;
    return ms > 600000 ? 600000 : ms;
}

// flags [first, first + count) except `skip` (-1: none) have all reached `want`
bool peer_wait_ge(DroneVec* v, Gather* g, int first, int count, int skip, uint32_t want) {
    if (g->gpu_waits) {  // one launch, one lane per flag, polling the shared words from the stream; gives up after the budget and says so in *d_err
        HIP_TRY(launch_flag_wait(reinterpret_cast<const uint32_t*>(g->d_flags) + first, (uint32_t)count, (uint32_t)skip, want, g->d_err, g->d_arrive + kPeerStopWord, g->budget_ticks, v->stream), return false);
        return true;
    }
    HIP_TRY(hipStreamSynchronize(v->stream), return false);
    const long limit_ms = peer_timeout_ms();
    timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    int slot = first;
    for (uint32_t spins = 0;; spins++) {
        while (slot < first + count && (slot - first == skip || (int32_t)(__atomic_load_n(g->flags + slot, __ATOMIC_ACQUIRE) - want) >= 0)) slot++;
        if (slot == first + count) return true;
        if ((spins & 1023u) == 1023u) {
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((t1.tv_sec - t0.tv_sec) * 1000L + (t1.tv_nsec - t0.tv_nsec) / 1000000L > limit_ms) {
                set_err("peer-store exchange: flag %d did not reach %u within %ld ms (a rank died or did not call drone_vec_gather)", slot, want, limit_ms);
                return false;
            }
        }
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
    }
}

// publish `value` in flag `slot` behind everything enqueued on the stream so far
bool peer_post(DroneVec* v, Gather* g, int slot, uint32_t value) {
    if (g->gpu_waits) {
        // A one-wave kernel (system-scope fence + store): one dependent launch boundary per flag — one rank stepping 2^20
        // envs with the root's per-step acknowledgement takes 43.3 us per step against 40.4-40.7 without an exchange. The
        // stream memory operation that looks cheaper (hipStreamWriteValue32: no kernel) is not: 46.1 us on the same box
        // (profiles/r04_ab/peer_post_forms.txt). Kept behind DRONE_PEER_STREAM_WRITES=1.
        if (g->stream_writes) {
            if (hipStreamWriteValue32(v->stream, g->d_flags + 4 * slot, value, 0) == hipSuccess) return true;
            (void)hipGetLastError();
            g->stream_writes = false;
        }
        HIP_TRY(launch_flag_post(reinterpret_cast<uint32_t*>(g->d_flags + 4 * slot), value, v->stream), return false);
        return true;
    }
    HIP_TRY(hipStreamSynchronize(v->stream), return false);
    __atomic_store_n(g->flags + slot, value, __ATOMIC_RELEASE);
    return true;
}

// a stream-side wait of an earlier round gave up: surface it on the next call instead of delivering a stale batch
bool peer_check_err(Gather* g) {
    if (g->h_err && __atomic_load_n(g->h_err, __ATOMIC_ACQUIRE) != 0u) {
        set_err("peer-store exchange: a wait on the stream gave up after %ld ms (a rank died or did not call drone_vec_gather)", peer_timeout_ms());
        return false;
    }
    return true;
}

// Before any launch that writes the output buffers (reset, step, rollout): the back-pressure half of the handshake.
// A non-root rank's kernel is about to overwrite its rows of the root's buffers with round seq + 1: it may, once the
// root has said that round seq has been consumed. The root says so at the start of ITS next launch — the
// consumer's reads were enqueued on the same stream between drone_vec_gather and this call, so they are ordered ahead.
// `sig` (zeroed by the caller): what the launch that follows publishes by itself (round 5) — the root's acknowledgement
// from its first workgroup as the kernel starts, a non-root rank's "round seq + 1 has landed" from the last workgroup to
// finish — so that neither is a one-wave launch of its own; with DRONE_PEER_INKERNEL=0, host-side waits or
// hipStreamWriteValue32 they stay separate and `sig` stays empty.
bool peer_before_launch(DroneVec* v, LaunchSig* sig) {
    Gather* g = v->gather;
    if (!g || !g->peer) return true;
    if (!peer_check_err(g)) return false;  // a stream-side wait of an earlier round gave up: that is what the caller must hear first
    // One batch per round: a second output-writing launch before the gather would find its wait already satisfied and overwrite
    // this rank's rows in the root's HBM while the root may be consuming the round the first launch announced.
    if (g->launched) {
        set_err("peer-store exchange: drone_vec_gather must follow every reset / step / rollout while the exchange is active (two launches without a gather in between)");
        return false;
    }
    const bool in_kernel = g->in_kernel && g->gpu_waits && !g->stream_writes;
    if (g->gpu_waits) {  // launches queued behind a wait that gives up must store nothing: the peer instantiations read the stop word
        sig->peer = 1u;
        sig->arrive = g->d_arrive;
    }
    g->launched = true;
    if (g->rank != g->root) {
        g->launch_posts = 0;
        if (in_kernel) {
            sig->post_flag = reinterpret_cast<uint32_t*>(g->d_flags + 4 * g->rank);
            sig->post_value = g->seq + 1u;
            g->launch_posts = sig->post_value;
        }
    }
    if (g->seq == 0) return true;
    if (g->rank == g->root) {
        if (g->acked != g->seq) {
            if (in_kernel) {
                sig->ack_flag = reinterpret_cast<uint32_t*>(g->d_flags + 4 * g->world);
                sig->ack_value = g->seq;
            } else if (!peer_post(v, g, g->world, g->seq)) {
                return false;
            }
            g->acked = g->seq;
        }
        return true;
    }
    return peer_wait_ge(v, g, g->world, 1, -1, g->seq);
}

// the launch peer_before_launch prepared did not go out: nothing will publish its round
void peer_launch_failed(DroneVec* v) {
    Gather* g = v->gather;
    if (!g || !g->peer) return;
    g->launch_posts = 0;
    g->launched = false;
}

// drone_vec_variant's text; `tuned`: " autotuned=1 table=8 tried=o8:170.1,o0:178.8,o6:170.3" once the handle has measured the candidates (SweepTune below)
void write_variant(DroneVec* v, const char* tuned) {
    snprintf(v->variant, sizeof(v->variant), "drone_step_kernel<task=%d,compact=%d,mem=%u,dt=%d> order=%u line_complete=%u packed_rk4=%u bytes=%d%s",
             v->cfg.task, v->dv.done_ids ? 1 : 0, (v->dv.order >> 2) & 3u, v->dv.derived_target ? 1 : 0, v->dv.order, v->dv.line_complete,
             v->dv.packed_rk4, drone_vec_bytes_per_env_step(v), tuned ? tuned : "");
}

// Pick the per-step kernel's sweep order / load hints by MEASUREMENT (round 5) — ONLINE, on the handle's real
// steps. Candidates: the footprint table's entry and its neighbours — a plain round-robin sweep (0), the sweep that turns
// around on odd steps with streamed action rows (6), a plain sweep with non-temporal state loads (8) — all instantiations the
// parity suite and the soak cover: the order only permutes which workgroup takes which chunk and which loads carry a hint,
// never a result, so real steps may run under any of them. From the handle's 161st step launch on, the candidates take turns
// in bursts of sixteen steps, twice each; every launch of a burst but its first four (the cache is still in the previous
// candidate's state) sits between two HIP events on the stream, read back lazily (hipEventQuery) when later calls find them
// complete — nothing waits, nothing extra is launched, no state or output is touched. When every candidate has its samples
// the fastest becomes the handle's order and drone_vec_variant() says what was measured
// (" autotuned=1 table=8 tried=o8:170.1,o0:178.8,o6:170.3").
// Why online: a first version timed trial steps right behind the first reset (profiles/r05_ab/autotune_offline_*.txt). No
// episode ends that soon after a reset — and the ranking depends on them: at 2^21 hover envs the plain sweep ran 80.8 us in
// that trial and 94.5 in steady state under the random policy (one episode end per 146 env-steps; the scattered log-plane and
// target updates cost it its cache residency), where the non-temporal sweep it "beat" runs 86.2 either way; at 2^23 the trial
// picked order 8, 4 % behind order 6 in steady state. The table, tuned in steady state, was right in all eight cases on two
// boxes; the offline trial in five. What a handle should measure is the workload it actually runs.
void tune_free(DroneVec* v) {
    if (!v->tune) return;
    for (auto& p : v->tune->pairs) {
        if (p.e0) (void)hipEventDestroy(p.e0);
        if (p.e1) (void)hipEventDestroy(p.e1);
    }
    delete v->tune;
    v->tune = nullptr;
}

void tune_harvest(SweepTune* t) {
    for (auto& p : t->pairs) {
        if (p.cand < 0) continue;
        const hipError_t q = hipEventQuery(p.e1);
        if (q == hipErrorNotReady) { (void)hipGetLastError(); continue; }
        float ms = 0.f;
        if (q == hipSuccess && hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
            t->sum_ms[p.cand] += ms;
            t->samples[p.cand] += 1;
        } else {
            (void)hipGetLastError();
        }
        p.cand = -1;
        t->in_flight -= 1;
    }
}

// called around every per-step launch of a handle that is still measuring: before it (returns the event pair to close behind the
// launch, or null) — may change v->dv.order for this launch
SweepTune::Pair* tune_before_step(DroneVec* v) {
    SweepTune* t = v->tune;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(v->stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone || v->dv.ctr || (v->gather && v->gather->peer)) {
        (void)hipGetLastError();  // a capture, graph-safe counters or a peer-store exchange: the table's choice stands for good
        v->dv.order = t->table;
        tune_free(v);
        return nullptr;
    }
    tune_harvest(t);
    const long k = t->seen++ - SweepTune::kStart;
    if (k < 0) return nullptr;
    const long burst = k / SweepTune::kBurst;
    if (burst >= (long)t->nc * SweepTune::kRounds) {  // exploration is over: decide once every pair has been read
        v->dv.order = t->table;
        if (t->in_flight > 0) return nullptr;
        int best = -1;
        for (int c = 0; c < t->nc; c++)
            if (t->samples[c] >= SweepTune::kBurst / 2 && (best < 0 || t->sum_ms[c] / t->samples[c] < t->sum_ms[best] / t->samples[best])) best = c;
        char tuned[200];
        int at = 0;
        if (best >= 0 && t->samples[0] >= SweepTune::kBurst / 2) {  // (without enough samples of the table's own entry there is nothing to compare with)
            v->dv.order = t->cand[best];
            at = snprintf(tuned, sizeof(tuned), " autotuned=1 table=%u tried=", t->table);
            for (int c = 0; c < t->nc && at < (int)sizeof(tuned) - 16; c++)
                at += snprintf(tuned + at, sizeof(tuned) - at, "%so%u:%.1f", c ? "," : "", t->cand[c], t->samples[c] ? t->sum_ms[c] * 1e3 / t->samples[c] : 0.0);
        } else {
            snprintf(tuned, sizeof(tuned), " autotuned=0 table=%u", t->table);
        }
        tune_free(v);
        write_variant(v, tuned);
        return nullptr;
    }
    const int c = (int)(burst % t->nc);
    v->dv.order = t->cand[c];
    if (k % SweepTune::kBurst < SweepTune::kLead) return nullptr;
    for (auto& p : t->pairs) {
        if (p.cand >= 0) continue;
        if (!p.e0 && (hipEventCreate(&p.e0) != hipSuccess || hipEventCreate(&p.e1) != hipSuccess)) { (void)hipGetLastError(); return nullptr; }
        if (hipEventRecord(p.e0, v->stream) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        p.cand = c;
        t->in_flight += 1;
        return &p;
    }
    return nullptr;  // every pair is still in flight (a caller far ahead of the device): this launch goes unmeasured
}

}  // namespace

extern "C" {

const char* drone_last_error(void) { return g_err; }

void drone_config_default(DroneConfig* c, int task) {
    memset(c, 0, sizeof(*c));
    c->struct_size = (uint32_t)sizeof(DroneConfig);
    c->task = task;
    c->buffer_kind = DRONE_BUFFERS_HOST;
    c->horizon = 1024;
    c->substeps = 1;
    c->dt = 0.01f;
    // Crazyflie-2.x-class airframe (public datasheet-level numbers)
    c->mass = 0.027f;
    c->arm = 0.0397f;
    c->ixx = 1.4e-5f;
    c->iyy = 1.4e-5f;
    c->izz = 2.17e-5f;
    c->k_thrust = 3.16e-10f;
    c->k_torque = 7.94e-12f;
    c->k_drag = 0.0027f;
    c->k_ang_damp = 1.0e-6f;
    c->gravity = 9.81f;
    c->max_rpm = 21702.0f;
    c->motor_tau = 0.05f;
    c->max_vel = 20.0f;
    c->max_omega = 50.0f;
    c->bound = 5.0f;
    c->spawn_extent = 3.0f;
    c->target_extent = 3.0f;
    c->tilt_init = 0.1f;
    c->hover_radius = 0.5f;
    c->waypoint_radius = 0.5f;
    c->wind_theta = 0.5f;
    c->wind_sigma = 1.0f;
    c->wind_max = 5.0f;
    c->c_omega = 1.0e-4f;
    c->c_action = 0.01f;
    c->crash_penalty = 1.0f;
    c->progress_scale = 1.0f;
    c->waypoint_bonus = 1.0f;
    c->agents_per_env = task == DRONE_TASK_SWARM ? 8 : 1;
    c->collision_radius = 0.15f;
    c->proximity_radius = 1.0f;
    c->c_proximity = 0.5f;
    c->gate_radius = 0.75f;
}

int drone_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int drone_vec_bytes_per_env_step(const DroneVec* v) {
    if (!v) return 0;
    const int task = v->cfg.task;
    const int aux = (task == DRONE_TASK_WAYPOINT || task == DRONE_TASK_RACE) ? 1 : 0, dt = v->dv.derived_target ? 1 : 0;
    // read: planes (6 / 7, or 5 derived-target) + action row; written: planes but the target one (+ the wind plane every step for
    // task 1; the gate normal of task 3 only per gate) + observation row + reward + 2 flag bytes
    const int planes_read = 6 + aux - dt, planes_written = 5 + (task == DRONE_TASK_WAYPOINT ? 1 : 0);
    return 16 * (planes_read + planes_written) + 16 + 4 * drone_obs_dim(task) + 4 + 2;
}

const char* drone_vec_variant(const DroneVec* v) { return v ? v->variant : ""; }

int drone_vec_host_transport(const DroneVec* v) {
    if (!v || !v->host_buffers) return -1;
    if (!v->zero_copy) return 0;
    for (int k = 0; k < 5; k++)
        if (v->bounce[k]) return v->threaded ? 3 : 2;
    return 1;
}

// Pin a host block the caller owns (on the handle's device), under the same page-ownership rule as the buffers given to init.
int drone_vec_host_pin(DroneVec* v, void* p, size_t bytes, int pages_exclusive) {
    Entry in(v);
    if (!in) return -1;
    if (!p || !bytes) { set_err("host_pin: NULL block or zero size"); return -1; }
    if (already_pinned(p, bytes)) return 0;  // the owner's registration: used as it is, never dropped by host_unpin
    const bool aligned = (reinterpret_cast<uintptr_t>(p) % kPage) == 0;
    if (!aligned || !pages_exclusive) {
        set_err("host_pin: the block must start on a 4 KiB boundary and be vouched for (pages_exclusive = 1: a mapping of its own - mmap, shm - padded to whole pages; not a block of the malloc heap), see DroneConfig.host_pages_exclusive");
        return -1;
    }
    const int cap = (int)(sizeof(v->pinned_blocks) / sizeof(v->pinned_blocks[0]));
    if (v->n_pinned_blocks >= cap) { set_err("host_pin: this handle already holds %d pinned blocks (unpin some first)", cap); return -1; }
    HIP_TRY(host_register(p, (bytes + kPage - 1) / kPage * kPage, v, "drone_vec_host_pin"), return -1);
    v->pinned_blocks[v->n_pinned_blocks++] = p;
    return 0;
}

int drone_vec_host_unpin(DroneVec* v, void* p) {
    Entry in(v);
    if (!in || !idle(v, "host_unpin")) return -1;
    if (!p) { set_err("host_unpin: NULL block"); return -1; }
    int k = 0;
    while (k < v->n_pinned_blocks && v->pinned_blocks[k] != p) k++;
    if (k == v->n_pinned_blocks) return 0;  // not registered by host_pin on this handle (the caller's own pin, or never pinned): not ours to drop
    HIP_TRY(hipStreamSynchronize(v->stream), return -1);  // nothing of this handle may still be writing the block
    v->pinned_blocks[k] = v->pinned_blocks[--v->n_pinned_blocks];
    HIP_TRY(hipHostUnregister(p), return -1);
    return 0;
}

int drone_vec_buffers(const DroneVec* v, float** observations, float** actions, float** rewards, unsigned char** terminals, unsigned char** truncations) {
    if (!v) { set_err("NULL handle"); return -1; }
    if (observations) *observations = v->u_obs;
    if (actions) *actions = v->u_act;
    if (rewards) *rewards = v->u_rew;
    if (terminals) *terminals = v->u_term;
    if (truncations) *truncations = v->u_trunc;
    return 0;
}

int drone_vec_device(const DroneVec* v) { return v ? v->device : -1; }

int drone_obs_dim(int task) { return (task == DRONE_TASK_SWARM || task == DRONE_TASK_RACE) ? DRONE_OBS_DIM_MAX : DRONE_OBS_DIM; }

DroneVec* drone_vec_init(float* observations, float* actions, float* rewards, unsigned char* terminals,
                         unsigned char* truncations, int num_envs, uint64_t seed, const DroneConfig* cfg) {
    g_err[0] = 0;
    if (!validate(cfg, num_envs)) return nullptr;
    // all five NULL on a device-buffer handle: the library allocates them in HBM (drone_vec_buffers hands them out)
    const bool lib_buffers = cfg->buffer_kind == DRONE_BUFFERS_DEVICE && !observations && !actions && !rewards && !terminals && !truncations;
    if (!lib_buffers && (!observations || !actions || !rewards || !terminals || !truncations)) {
        set_err("buffer pointer is NULL (only a DRONE_BUFFERS_DEVICE handle may pass all five as NULL: library-owned buffers)");
        return nullptr;
    }
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev), return nullptr);
    if (ndev <= 0 || cfg->device < 0 || cfg->device >= ndev) { set_err("HIP device %d not available (%d devices): this library has no CPU path", cfg->device, ndev); return nullptr; }
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, cfg->device), return nullptr);
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) { set_err("device %d is %s; this library is built for gfx950 only", cfg->device, prop.gcnArchName); return nullptr; }

    DeviceRestore restore_callers_device;
    DroneVec* v = new (std::nothrow) DroneVec();
    if (!v) { set_err("out of memory"); return nullptr; }
    memset(v, 0, sizeof(*v));
    v->cfg = *cfg;
    v->seed = seed;
    v->n = num_envs;
    v->n_pad = (uint32_t)(((uint64_t)num_envs + kBlock - 1) / kBlock * kBlock);
    v->stride = v->n_pad + plane_pad_elems();
    v->device = cfg->device;
    v->host_buffers = cfg->buffer_kind == DRONE_BUFFERS_HOST;
    v->u_obs = observations; v->u_act = actions; v->u_rew = rewards; v->u_term = terminals; v->u_trunc = truncations;

    const size_t n = (size_t)num_envs;
#define INIT_TRY(expr) HIP_TRY(expr, { drone_vec_close(v); return nullptr; })
    INIT_TRY(hipSetDevice(v->device));
    INIT_TRY(hipStreamCreateWithFlags(&v->stream, hipStreamNonBlocking));
    v->own_stream = true;
    INIT_TRY(hipEventCreate(&v->ev0));
    INIT_TRY(hipEventCreate(&v->ev1));
    {   // Derived-target layout (drone_params.hpp): hover / swarm handles whose step is HBM-bound drop the target plane
        // (16 of hover's 278 bytes per env-step) and re-derive the target from three hashes per step. It needs tick
        // and score_count in 16 bits each (horizon <= 65 535). Chosen by footprint like the other layout knobs
        // (profiles/r03_ab/ab_dt_*.txt: -4.4 % at 2^19 envs, -5.2 % at 2^20, -6.4 % at 2^21, -4.7 % at 2^22; neutral at
        // 2^18 and +1.4 % at 131 072, where the step is bound by one or two waves per SIMD issuing VALU, not by bytes);
        // DRONE_DERIVED_TARGET=0/1 forces it (1 is ignored where the layout cannot represent the handle).
        // Who decides: DRONE_DERIVED_TARGET (tuning tools) over DroneConfig.state_layout (the caller's declared choice) over
        // the footprint rule.
        const bool can = task_has_derived_target(cfg->task) && cfg->horizon <= 65535;
        const char* e = getenv("DRONE_DERIVED_TARGET");
        const size_t per_step = (size_t)num_envs * 278u;
        if (e && *e) v->dv.derived_target = can && atoi(e) != 0;
        else if (cfg->state_layout == DRONE_LAYOUT_DERIVED_TARGET) v->dv.derived_target = 1;  // validate() checked that it can
        else if (cfg->state_layout == DRONE_LAYOUT_TARGET_PLANE) v->dv.derived_target = 0;
        else v->dv.derived_target = can && per_step >= ((size_t)100 << 20);
    }
    const bool dt = v->dv.derived_target != 0;
    const size_t hot_elems = (size_t)v->n_pad * hot_planes(cfg->task, dt), cold_elems = (size_t)2 * v->stride;
    INIT_TRY(hipMalloc((void**)&v->dv.planes, sizeof(float4) * hot_elems));
    INIT_TRY(hipMemsetAsync(v->dv.planes, 0, sizeof(float4) * hot_elems, v->stream));
    INIT_TRY(hipMalloc((void**)&v->dv.cold, sizeof(float4) * cold_elems));
    INIT_TRY(hipMemsetAsync(v->dv.cold, 0, sizeof(float4) * cold_elems, v->stream));
    INIT_TRY(hipMalloc((void**)&v->d_kp, sizeof(KParams)));
    INIT_TRY(hipMalloc((void**)&v->dv.pad_sink, sizeof(float) * kBlock));
#if defined(DRONE_STAMPS) && DRONE_STAMPS  // diagnostic build (tools/stamps.py)
    INIT_TRY(hipMalloc((void**)&v->dv.stamps, sizeof(unsigned long long) * kStampSlots * (v->n_pad / 64)));
#endif
    INIT_TRY(hipMalloc((void**)&v->d_partials, sizeof(double) * 6 * kLogMaxGrid));
    INIT_TRY(hipHostMalloc((void**)&v->h_partials, sizeof(double) * 6 * kLogMaxGrid, hipHostMallocDefault));
    if (cfg->compact_done) {
        INIT_TRY(hipMalloc((void**)&v->dv.done_ids, sizeof(uint32_t) * n));
        INIT_TRY(hipMalloc((void**)&v->dv.done_count, sizeof(uint32_t) * 2));
        INIT_TRY(hipMemsetAsync(v->dv.done_count, 0, sizeof(uint32_t) * 2, v->stream));
    }
    if (v->host_buffers) {
        INIT_TRY(hipMalloc((void**)&v->d_obs, n * (size_t)drone_obs_dim(v->cfg.task) * sizeof(float)));
        INIT_TRY(hipMalloc((void**)&v->d_act, n * DRONE_ACT_DIM * sizeof(float)));
        INIT_TRY(hipMalloc((void**)&v->d_rew, n * sizeof(float)));
        INIT_TRY(hipMalloc((void**)&v->d_term, n));
        INIT_TRY(hipMalloc((void**)&v->d_trunc, n));
        void* const host[5] = {observations, actions, rewards, terminals, truncations};
        const size_t bytes[5] = {n * (size_t)drone_obs_dim(v->cfg.task) * sizeof(float), n * DRONE_ACT_DIM * sizeof(float), n * sizeof(float), n, n};
        bool pinned[5];
        size_t unpinned_bytes = 0;
        for (int k = 0; k < 5; k++) {
            pinned[k] = pin_caller_buffer(v, k, host[k], bytes[k]);
            if (!pinned[k]) unpinned_bytes += bytes[k];
        }
        v->dv.obs = v->d_obs; v->dv.act = v->d_act; v->dv.rew = v->d_rew; v->dv.term = v->d_term; v->dv.trunc = v->d_trunc;
        // Buffers that cannot be pinned (a worker's unaligned slices of a shared-memory block: the one-byte flag slices
        // practically always) get pinned stand-ins owned here, as long as copying them on the host is cheaper than the
        // mirror transport's DMA copies: up to DRONE_HOST_BOUNCE_MAX_BYTES in total (default 1 MiB; 0 = never).
        const char* bm = getenv("DRONE_HOST_BOUNCE_MAX_BYTES");
        const size_t bounce_max = (bm && *bm) ? (size_t)atoll(bm) : ((size_t)1 << 20);
        // Round 5: beyond that budget — the mid-size shards of a vec-env whose slices cannot be pinned, 16 384
        // to ~10^5 envs — the stand-ins are moved by the host copy pool instead (drone_host_copy.hpp; transport 3): the action
        // rows in as parallel slices, the outputs out chunk by chunk while the kernel is still writing over PCIe. Up to
        // DRONE_HOST_MT_MAX_BYTES (default 64 MiB of unpinnable buffers; beyond, the step is PCIe-bound for milliseconds and the
        // mirror transport's DMA copies are as good); DRONE_HOST_COPY_THREADS=1 (no pool) keeps the mirror transport.
        const char* mm = getenv("DRONE_HOST_MT_MAX_BYTES");
        const size_t mt_max = (mm && *mm) ? (size_t)atoll(mm) : ((size_t)64 << 20);
        // Where the pool takes over from the single memcpy: DRONE_HOST_POOL_MIN_BYTES, default 512 KiB (~5 000 hover envs; measured at
        // equal cost at 4 096 envs, 36 against 40 us at 6 144, 39 against 54 at 8 192: profiles/r05_ab/pool_hand_over.txt) — the single
        // memcpy stays the fallback up to DRONE_HOST_BOUNCE_MAX_BYTES in a process without the pool; a budget set by hand moves both.
        const char* pm = getenv("DRONE_HOST_POOL_MIN_BYTES");
        const size_t pool_min = (pm && *pm) ? (size_t)atoll(pm) : (bm && *bm) ? bounce_max : ((size_t)512 << 10);
        const bool threaded = bounce_max > 0 && unpinned_bytes > pool_min && unpinned_bytes <= mt_max && CopyPool::get().parts() > 1;  // (a budget of 0 turns stand-ins of either kind off)
        if (want_zero_copy(num_envs) && (unpinned_bytes <= bounce_max || threaded)) {
            void* mapped[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
            bool have_all = true;
            for (int k = 0; k < 5 && have_all; k++) {
                if (!pinned[k]) {
                    if (hipHostMalloc(&v->bounce[k], bytes[k], hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); v->bounce[k] = nullptr; have_all = false; break; }
                    v->bounce_bytes[k] = bytes[k];
                    memset(v->bounce[k], 0, bytes[k]);
                }
                mapped[k] = mapped_ptr(v->bounce[k] ? v->bounce[k] : host[k]);
                have_all = mapped[k] != nullptr;
            }
            v->m_obs = (float*)mapped[0];
            v->m_act = (float*)mapped[1];
            v->m_rew = (float*)mapped[2];
            v->m_term = (unsigned char*)mapped[3];
            v->m_trunc = (unsigned char*)mapped[4];
            const bool ok = v->m_obs && v->m_act && v->m_rew && v->m_term && v->m_trunc &&
                            !(reinterpret_cast<uintptr_t>(v->m_obs) & 15u) && !(reinterpret_cast<uintptr_t>(v->m_act) & 15u) &&
                            !(reinterpret_cast<uintptr_t>(v->m_rew) & 3u);
            bool words = true;
            if (ok && threaded) {  // the per-chunk completion words
                v->n_wg = v->n_pad / (uint32_t)kBlock;
                void* hw = nullptr;
                words = hipHostMalloc(&hw, sizeof(uint32_t) * v->n_wg, hipHostMallocMapped) == hipSuccess;
                if (words) {
                    v->h_wg_done = static_cast<uint32_t*>(hw);
                    memset(hw, 0, sizeof(uint32_t) * v->n_wg);
                    v->d_wg_done = static_cast<uint32_t*>(mapped_ptr(hw));
                    words = v->d_wg_done != nullptr;
                } else {
                    (void)hipGetLastError();
                }
            }
            if (ok && words) {
                v->zero_copy = true;
                v->threaded = threaded;
                ensure_flag(v);
                v->dv.obs = v->m_obs; v->dv.act = v->m_act; v->dv.rew = v->m_rew; v->dv.term = v->m_term; v->dv.trunc = v->m_trunc;
            } else {
                drop_bounce(v);  // mirror transport after all
            }
        }
    } else {
        if (lib_buffers) {  // freed at close like the host handles' mirrors, which these fields otherwise hold
            const size_t ob = n * (size_t)drone_obs_dim(v->cfg.task) * sizeof(float);
            INIT_TRY(hipMalloc((void**)&v->d_obs, ob));
            INIT_TRY(hipMalloc((void**)&v->d_act, n * DRONE_ACT_DIM * sizeof(float)));
            INIT_TRY(hipMalloc((void**)&v->d_rew, n * sizeof(float)));
            INIT_TRY(hipMalloc((void**)&v->d_term, n));
            INIT_TRY(hipMalloc((void**)&v->d_trunc, n));
            INIT_TRY(hipMemsetAsync(v->d_obs, 0, ob, v->stream));
            INIT_TRY(hipMemsetAsync(v->d_act, 0, n * DRONE_ACT_DIM * sizeof(float), v->stream));
            INIT_TRY(hipMemsetAsync(v->d_rew, 0, n * sizeof(float), v->stream));
            INIT_TRY(hipMemsetAsync(v->d_term, 0, n, v->stream));
            INIT_TRY(hipMemsetAsync(v->d_trunc, 0, n, v->stream));
            observations = v->d_obs; actions = v->d_act; rewards = v->d_rew; terminals = v->d_term; truncations = v->d_trunc;
            v->u_obs = observations; v->u_act = actions; v->u_rew = rewards; v->u_term = terminals; v->u_trunc = truncations;
        }
        if ((reinterpret_cast<uintptr_t>(observations) & 15u) || (reinterpret_cast<uintptr_t>(actions) & 15u) || (reinterpret_cast<uintptr_t>(rewards) & 3u)) {
            set_err("device buffers must be 16-byte aligned (observations, actions) and 4-byte aligned (rewards)");
            drone_vec_close(v);
            return nullptr;
        }
        v->dv.obs = observations; v->dv.act = actions; v->dv.rew = rewards; v->dv.term = terminals; v->dv.trunc = truncations;
    }
    v->dv.n = (uint32_t)num_envs;
    v->dv.n_pad = v->n_pad;
    v->dv.stride = v->stride;
    {   // Partial-line plane updates (ended episodes) cost an HBM read-modify-write each unless the Infinity Cache absorbs
        // them: widen them to whole lines once a step touches more than twice its 256 MiB (measured cross-over between
        // 2^20 and 2^22 envs, profiles/r02_ab/). DRONE_LINE_COMPLETE=0/1 forces it.
        const char* e = getenv("DRONE_LINE_COMPLETE");
        const size_t touched = n * (sizeof(float4) * (dt ? 2 * hot_planes(cfg->task, true) : 2 * hot_planes(cfg->task) - 1) + (size_t)drone_obs_dim(cfg->task) * 4 + 16 + 6);
        v->dv.line_complete = (e && *e) ? (atoi(e) != 0) : (touched > ((size_t)512 << 20));
        // Sweep order of the step kernel, by the same footprint (DRONE_SWEEP_ORDER=0..15 forces it): up to ~1.5x the
        // Infinity Cache, one contiguous eighth per XCD (-2.7 % at 2^20 envs); beyond, one global round-robin sweep
        // that, in some bands, turns around on odd steps (the tail of one step is the head of the next and is still
        // cached: -6 % at 2^22 envs in round 2) with the action rows streamed (bit 2: -2 % there).
        // Round 4, hover task in the derived-target layout (where the band was swept end to end, profiles/r04_ab/band_*.txt,
        // upper_*.txt): between ~400 MiB and ~1.1 GiB touched per step — a step that is one to four times the Infinity
        // Cache, the LRU worst case — the STATE loads carry the non-temporal hint (bit 3) on a plain round-robin sweep:
        // -3 % at 426 MiB, -10 ... -11.5 % from 458 to 655 MiB (2^21 envs: 97.6 -> 87.9 us), -10 % at 786 MiB (where the
        // old rule's reversed sweep was 12 % behind a plain one), -6 % at 917 MiB, -1.8 % at 2^22 envs; beyond ~1.1 GiB the
        // reversed sweep with streamed action rows wins again (+2 ... +4 % for the hint from 4.5 M envs on). The hint costs
        // +21 % at 2^20 envs, hence a band and not a switch; and it is the hover task's band only: the seven-plane tasks
        // lose 4 % to it at the same footprints (waypoint 1.75 M envs, race 2 M), the swarm task prefers order 6 there.
        const char* o = getenv("DRONE_SWEEP_ORDER");
        const bool hover_dt = cfg->task == DRONE_TASK_HOVER && dt;
        // The other tasks, and the bands around it (round 4 re-sweep of round 2's thresholds, profiles/r04_ab/sweep_*.txt:
        // waypoint / swarm / race at 1.3 ... 4.2 M envs, orders 0 1 2 6 8): the reversed sweep pays between ~450 and ~600 MiB
        // (-4 ... -10 % against a plain sweep) and again beyond ~900 MiB, but LOSES to a plain sweep in between, most at
        // about three times the Infinity Cache (775 ... 835 MiB: +3 ... +12 %, all four tasks) — the old rule switched to it
        // at 768 MiB, exactly there.
        const size_t mib = touched >> 20;
        const size_t rev_to = cfg->task == DRONE_TASK_SWARM ? 720 : 600;  // the swarm task keeps the reversed sweep's band longer (625 MiB: 103.2 against 116.1 us plain; 695: 124.3 / 130.4; 764: plain wins)
        uint32_t order = mib <= 400 ? 1u : mib <= 450 ? 0u : mib <= rev_to ? 6u : mib <= 900 ? 0u : 6u;
        if (hover_dt && mib > 400 && mib <= 1100) order = 8u;
        v->dv.order = (o && *o) ? (uint32_t)atoi(o) : order;
        // DRONE_AUTOTUNE=1 (opt-in since round 6): beyond 400 MiB the handle's own steps 161 ... 256 time the table's entry and
        // its neighbours on this box, on its own buffers, under its own workload (SweepTune), and the fastest becomes the
        // handle's order. Round 5 ran this by default and logged 12 of 12 cases on three boxes in which the measurement picked
        // exactly the table's entry, at a cost of 0.1 ... 0.8 % (profiles/r05_ab/autotune_online_box*.txt):
        // the table stands, the measurement is there for a box or a workload someone has reason to distrust it on. Not for a
        // forced order or host buffers (PCIe-bound at these sizes).
        const char* at = getenv("DRONE_AUTOTUNE");
        v->touched_mib = mib;
        if (mib > 400 && !(o && *o) && !v->host_buffers && at && *at && atoi(at) != 0) {
            v->tune = new (std::nothrow) SweepTune();
            if (v->tune) {
                v->tune->table = order;
                const uint32_t all[3] = {0u, 6u, 8u};
                v->tune->cand[v->tune->nc++] = order;
                for (uint32_t c : all)
                    if (c != order) v->tune->cand[v->tune->nc++] = c;
            }
        }
    }
    {   // Packed-f32 RK4 in the register-resident kernels (fused rollout, step_many): wins only while a SIMD holds ONE
        // wave (<= 65 536 envs on the 1024 SIMDs: rollout -8.8 %, step_many -3.9 %; waypoint / race -3...4 %), where the
        // issue rate of one wave is the limit and a packed instruction costs 1.1x a scalar one for two results. With two
        // waves per SIMD it already loses (131 072 envs: +3.3 % / +3.9 %), on a full chip clearly (2^20: rollout +11 %) —
        // profiles/r03_ab/ab_pk_*.txt. DRONE_PACKED_RK4=0/1 forces it.
        const char* e = getenv("DRONE_PACKED_RK4");
        v->dv.packed_rk4 = (e && *e) ? (atoi(e) != 0) : (v->n_pad <= 65536u);
    }
    v->dv.kp = v->d_kp;
    v->dv.kp_host = &v->kp;
    if (!upload_params(v)) { drone_vec_close(v); return nullptr; }
    write_variant(v, nullptr);
#undef INIT_TRY
    if (debug_reg()) fprintf(stderr, "[drone reg] init %p n=%d %s%s obs=%p act=%p planes=%p\n", (void*)v, v->n, v->host_buffers ? "host" : "device", v->zero_copy ? " zero-copy" : "", (void*)observations, (void*)actions, (void*)v->dv.planes);
    return v;
}

void drone_vec_reset(DroneVec* v, uint64_t seed) {
    Entry in(v);
    if (!in || !idle(v, "reset")) return;
    v->seed = seed;
    v->gstep = 0;
    v->step_launches = 0;  // the reset kernel zeroes both done-count slots
    v->list_valid = false;
    v->many_k = 0;
    LaunchSig sig = {nullptr, nullptr, nullptr, nullptr, 0u, 0u, 0u, 0u};
    if (!upload_params(v) || !push_counters(v) || !peer_before_launch(v, &sig)) return;
    HIP_TRY(launch_reset(v->dv, v->cfg.task, v->stream, &sig), { peer_launch_failed(v); return; });
    if (v->host_buffers) device_to_host_outputs(v);
}

namespace {
bool step_send_impl(DroneVec* v) {
#if DRONE_HOST_STAMPS
    g_stamps.enter();
#endif
    if (v->host_buffers && !host_to_device_actions(v)) return false;
    HOST_STAMP(kActionsIn);
    LaunchSig sig = {nullptr, nullptr, nullptr, nullptr, 0u, 0u, 0u, 0u};
    if (!peer_before_launch(v, &sig)) return false;
    const bool copy_out = v->host_buffers && v->zero_copy && v->threaded && v->h_flag;  // (without the completion flag the plain wait + whole copy is used)
    if (copy_out) {
        sig.wg_done = v->d_wg_done;
        sig.wg_done_value = ++v->wg_seq;
    }
    SweepTune::Pair* timing = v->tune ? tune_before_step(v) : nullptr;  // an HBM-bound handle still measuring its sweep order (may set dv.order for this launch)
    HIP_TRY(launch_step(v->dv, v->cfg.task, v->gstep, v->step_launches & 1u, v->stream, &sig), { peer_launch_failed(v); return false; });
    HOST_STAMP(kLaunched);
    if (timing && hipEventRecord(timing->e1, v->stream) != hipSuccess) {  // (v->tune is still there: a pair is only handed out while measuring)
        (void)hipGetLastError();
        timing->cand = -1;
        v->tune->in_flight -= 1;
    }
    v->gstep += 1;
    v->step_launches += 1;
    v->list_valid = true;
    v->many_k = 0;
    if (!v->host_buffers) return true;
    if (!enqueue_host_outputs(v)) return false;
    if (copy_out) {  // the helpers start following the chunks' words now; the caller joins in finish_host_outputs
        __atomic_store_n(&v->stream_idle, 0, __ATOMIC_RELEASE);
        __atomic_store_n(&v->copy_abort, 0, __ATOMIC_RELEASE);
        v->copy_started = CopyPool::get().try_start(copy_outputs_part, v);  // (busy with another handle's step: this one's outputs are copied after the wait, by this thread)
        HOST_STAMP(kPoolStarted);
    }
    return true;
}
}  // namespace

void drone_vec_step(DroneVec* v) {
    Entry in(v);
    if (!in || !idle(v, "step")) return;
    if (step_send_impl(v) && v->host_buffers) finish_host_outputs(v);
}

void drone_vec_step_send(DroneVec* v) {
    Entry in(v);
    if (!in || !idle(v, "step_send")) return;
    if (step_send_impl(v)) v->pending = true;
}

void drone_vec_step_recv(DroneVec* v) {
    Entry in(v);
    if (!in) return;
    if (!v->pending) { set_err("step_recv: no step was sent (drone_vec_step_send)"); return; }
    v->pending = false;
    if (v->host_buffers) finish_host_outputs(v);
}

void drone_vec_rollout(DroneVec* v, int horizon) {
    Entry in(v);
    if (!in || !idle(v, "rollout")) return;
    if (horizon <= 0) { set_err("rollout: horizon must be positive, got %d", horizon); return; }
    LaunchSig sig = {nullptr, nullptr, nullptr, nullptr, 0u, 0u, 0u, 0u};
    if (!peer_before_launch(v, &sig)) return;
    HIP_TRY(launch_rollout(v->dv, v->cfg.task, v->gstep, (uint32_t)horizon, v->stream, &sig), { peer_launch_failed(v); return; });
    v->gstep += (uint32_t)horizon;
    v->list_valid = false;  // the fused rollout builds no done-id list
    v->many_k = 0;
    if (v->host_buffers) device_to_host_outputs(v);
}

namespace {

// grow the K-dependent device storage of drone_vec_step_many (drained first: an earlier launch may still use the old blocks)
bool many_reserve(DroneVec* v, int k_steps, bool need_staging) {
    const size_t n = (size_t)v->n, od = (size_t)drone_obs_dim(v->cfg.task), K = (size_t)k_steps;
    if (v->cfg.compact_done && k_steps > v->many_cap) {
        HIP_TRY(hipStreamSynchronize(v->stream), return false);
        (void)hipFree(v->many_ids); v->many_ids = nullptr;
        (void)hipFree(v->many_count); v->many_count = nullptr;
        v->many_cap = 0;
        HIP_TRY(hipMalloc((void**)&v->many_ids, sizeof(uint32_t) * K * n), return false);
        HIP_TRY(hipMalloc((void**)&v->many_count, sizeof(uint32_t) * K), return false);
        v->many_cap = k_steps;
    }
    if (v->host_buffers && need_staging && k_steps > v->stage_cap) {
        HIP_TRY(hipStreamSynchronize(v->stream), return false);
        (void)hipFree(v->s_act); (void)hipFree(v->s_obs); (void)hipFree(v->s_rew); (void)hipFree(v->s_term); (void)hipFree(v->s_trunc);
        v->s_act = v->s_obs = v->s_rew = nullptr; v->s_term = v->s_trunc = nullptr;
        v->stage_cap = 0;
        HIP_TRY(hipMalloc((void**)&v->s_act, K * n * DRONE_ACT_DIM * sizeof(float)), return false);
        HIP_TRY(hipMalloc((void**)&v->s_obs, K * n * od * sizeof(float)), return false);
        HIP_TRY(hipMalloc((void**)&v->s_rew, K * n * sizeof(float)), return false);
        HIP_TRY(hipMalloc((void**)&v->s_term, K * n), return false);
        HIP_TRY(hipMalloc((void**)&v->s_trunc, K * n), return false);
        v->stage_cap = k_steps;
    }
    return true;
}

}  // namespace

namespace {
void step_many_impl(DroneVec* v, int k_steps, const float* actions, bool repeat, float* observations, float* rewards,
                    unsigned char* terminals, unsigned char* truncations);
}

void drone_vec_step_many(DroneVec* v, int k_steps, const float* actions, float* observations, float* rewards,
                         unsigned char* terminals, unsigned char* truncations) {
    step_many_impl(v, k_steps, actions, false, observations, rewards, terminals, truncations);
}

void drone_vec_step_repeat(DroneVec* v, int k_steps, const float* actions, float* observations, float* rewards,
                           unsigned char* terminals, unsigned char* truncations) {
    if (!actions) { Entry in(v); if (in) set_err("step_repeat: actions is NULL (the in-kernel policy is drone_vec_step_many with actions = NULL)"); return; }
    step_many_impl(v, k_steps, actions, true, observations, rewards, terminals, truncations);
}

namespace {

// `repeat`: `actions` is ONE [N][4] block applied to all k_steps steps (action repeat / frame skip)
void step_many_impl(DroneVec* v, int k_steps, const float* actions, bool repeat, float* observations, float* rewards,
                    unsigned char* terminals, unsigned char* truncations) {
    Entry in(v);
    if (!in || !idle(v, "step_many")) return;
    if (k_steps < 1) { set_err("step_many: k_steps must be positive, got %d", k_steps); return; }
    if (!observations || !rewards || !terminals || !truncations) { set_err("step_many: NULL output block"); return; }
    if (v->gather && v->gather->peer) {  // it writes the caller's blocks, not this rank's rows of the root's batch, and runs outside the handshake
        set_err("step_many / step_repeat: not while the peer-store exchange is active (the root's batch holds one step per round; drone_vec_gather_close first)");
        return;
    }
    const size_t n = (size_t)v->n, od = (size_t)drone_obs_dim(v->cfg.task), K = (size_t)k_steps;
    // Host handles: blocks the caller pinned beforehand (drone_host_pin, hipHostMalloc, hipHostRegister) are accessed by
    // the kernel in place over PCIe — no staging, no copy commands; anything else goes through device staging and DMA.
    const float* d_act = actions;
    float* d_obs = observations;
    float* d_rew = rewards;
    unsigned char* d_term = terminals;
    unsigned char* d_trunc = truncations;
    bool direct = false;
    if (v->host_buffers) {
        const char* zc = getenv("DRONE_HOST_ZEROCOPY");
        const size_t act_bytes = (repeat ? 1 : K) * n * DRONE_ACT_DIM * sizeof(float);
        if (!(zc && *zc && atoi(zc) == 0) && (!actions || already_pinned(actions, act_bytes)) && already_pinned(observations, K * n * od * sizeof(float)) &&
            already_pinned(rewards, K * n * sizeof(float)) && already_pinned(terminals, K * n) && already_pinned(truncations, K * n)) {
            void* m[5] = {actions ? mapped_ptr(const_cast<float*>(actions)) : nullptr, mapped_ptr(observations), mapped_ptr(rewards), mapped_ptr(terminals), mapped_ptr(truncations)};
            direct = (!actions || m[0]) && m[1] && m[2] && m[3] && m[4] && !(reinterpret_cast<uintptr_t>(m[0]) & 15u) && !(reinterpret_cast<uintptr_t>(m[1]) & 15u) &&
                     !(reinterpret_cast<uintptr_t>(m[2]) & 3u);
            if (direct) {
                d_act = static_cast<const float*>(m[0]); d_obs = static_cast<float*>(m[1]); d_rew = static_cast<float*>(m[2]);
                d_term = static_cast<unsigned char*>(m[3]); d_trunc = static_cast<unsigned char*>(m[4]);
            }
        }
    }
    if (!many_reserve(v, k_steps, !direct)) return;
    if (v->host_buffers && !direct) {
        d_act = actions ? v->s_act : nullptr;
        d_obs = v->s_obs; d_rew = v->s_rew; d_term = v->s_term; d_trunc = v->s_trunc;
        if (actions) HIP_TRY(hipMemcpyAsync(v->s_act, actions, (repeat ? 1 : K) * n * DRONE_ACT_DIM * sizeof(float), hipMemcpyHostToDevice, v->stream), return);
    } else if (!v->host_buffers && ((reinterpret_cast<uintptr_t>(observations) & 15u) || (reinterpret_cast<uintptr_t>(actions) & 15u) || (reinterpret_cast<uintptr_t>(rewards) & 3u))) {
        set_err("step_many: device blocks must be 16-byte aligned (observations, actions) and 4-byte aligned (rewards)");
        return;
    }
    if (v->cfg.compact_done) HIP_TRY(hipMemsetAsync(v->many_count, 0, sizeof(uint32_t) * K, v->stream), return);
    HIP_TRY(launch_step_many(v->dv, v->cfg.task, v->gstep, (uint32_t)k_steps, d_act, repeat ? 0u : (uint32_t)v->n, d_obs, d_rew, d_term, d_trunc,
                             v->cfg.compact_done ? v->many_ids : nullptr, v->cfg.compact_done ? v->many_count : nullptr, v->stream), return);
    v->gstep += (uint32_t)k_steps;
    v->list_valid = false;
    v->many_k = k_steps;
    if (v->host_buffers && direct) {
        ensure_flag(v);
        (void)wait_zero_copy(v);
    } else if (v->host_buffers) {
        HIP_TRY(hipMemcpyAsync(observations, v->s_obs, K * n * od * sizeof(float), hipMemcpyDeviceToHost, v->stream), return);
        HIP_TRY(hipMemcpyAsync(rewards, v->s_rew, K * n * sizeof(float), hipMemcpyDeviceToHost, v->stream), return);
        HIP_TRY(hipMemcpyAsync(terminals, v->s_term, K * n, hipMemcpyDeviceToHost, v->stream), return);
        HIP_TRY(hipMemcpyAsync(truncations, v->s_trunc, K * n, hipMemcpyDeviceToHost, v->stream), return);
        HIP_TRY(hipStreamSynchronize(v->stream), return);
    }
}

}  // namespace

void drone_vec_log(DroneVec* v, DroneLog* out) {
    if (!out) return;
    memset(out, 0, sizeof(*out));
    Entry in(v);
    if (!in || !idle(v, "log")) return;
    int grid = 0;
    HIP_TRY(launch_log_reduce(v->dv, v->d_partials, kLogMaxGrid, &grid, v->stream), return);
    HIP_TRY(hipMemcpyAsync(v->h_partials, v->d_partials, sizeof(double) * 6 * grid, hipMemcpyDeviceToHost, v->stream), return);
    HIP_TRY(hipStreamSynchronize(v->stream), return);
    double s[6] = {0, 0, 0, 0, 0, 0};
    for (int b = 0; b < grid; b++)
        for (int k = 0; k < 6; k++) s[k] += v->h_partials[b * 6 + k];
    const double n = s[4];
    if (n > 0) {
        // SPEC.md section 8 (v5): hover / swarm episodes log the count of steps within hover_radius; reported per step flown
        const bool per_step = v->cfg.task == DRONE_TASK_HOVER || v->cfg.task == DRONE_TASK_SWARM;
        out->perf = (float)(s[0] / (per_step ? s[3] : n));
        out->score = (float)(s[1] / (per_step ? s[3] : n));
        out->episode_return = (float)(s[2] / n);
        out->episode_length = (float)(s[3] / n);
        out->oob = (float)(s[5] / n);
    }
    out->n = (float)n;
}

void drone_vec_close(DroneVec* v) {
    if (!v) return;
    if (debug_reg()) fprintf(stderr, "[drone reg] close %p\n", (void*)v);
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; }
    (void)hipSetDevice(v->device);
    if (v->stream) (void)hipStreamSynchronize(v->stream);
    if (v->copy_started) (void)finish_threaded_copy(v);  // closed between step_send and step_recv: the pool must let go of this handle first
    tune_free(v);
    gather_destroy(v);
    if (v->h_wg_done) (void)hipHostFree(v->h_wg_done);
    for (int i = 0; i < 5; i++)
        if (v->registered[i]) host_unregister(v->registered_ptr[i], v, "caller buffer at close");
    for (int i = 0; i < v->n_pinned_blocks; i++) host_unregister(v->pinned_blocks[i], v, "drone_vec_host_pin block at close");
    v->n_pinned_blocks = 0;
    (void)hipFree(v->dv.planes);
    (void)hipFree(v->dv.cold);
    (void)hipFree(v->d_kp);
    (void)hipFree(v->dv.pad_sink);
    (void)hipFree(v->dv.ctr);
    (void)hipFree(v->dv.stamps);
    (void)hipFree(v->d_partials);
    if (v->h_partials) (void)hipHostFree(v->h_partials);
    if (v->h_flag) (void)hipHostFree(const_cast<uint32_t*>(v->h_flag));
    drop_bounce(v);
    (void)hipFree(v->dv.done_ids);
    (void)hipFree(v->dv.done_count);
    (void)hipFree(v->many_ids);
    (void)hipFree(v->many_count);
    (void)hipFree(v->s_act);
    (void)hipFree(v->s_obs);
    (void)hipFree(v->s_rew);
    (void)hipFree(v->s_term);
    (void)hipFree(v->s_trunc);
    (void)hipFree(v->d_obs);
    (void)hipFree(v->d_act);
    (void)hipFree(v->d_rew);
    (void)hipFree(v->d_term);
    (void)hipFree(v->d_trunc);
    if (v->ev0) (void)hipEventDestroy(v->ev0);
    if (v->ev1) (void)hipEventDestroy(v->ev1);
    if (v->own_stream && v->stream) (void)hipStreamDestroy(v->stream);
    if (prev >= 0 && prev != v->device) (void)hipSetDevice(prev);
    delete v;
}

int drone_vec_set_stream(DroneVec* v, void* hip_stream) {
    if (!v) return -1;
    if ((hipStream_t)hip_stream == v->stream && !v->own_stream) return 0;  // cheap to call every step
    Entry in(v);
    if (!in || !idle(v, "set_stream")) return -1;
    HIP_TRY(hipStreamSynchronize(v->stream), return -1);
    if (v->own_stream && v->stream) (void)hipStreamDestroy(v->stream);
    v->stream = (hipStream_t)hip_stream;
    v->own_stream = false;
    return 0;
}

int drone_vec_sync(DroneVec* v) {
    Entry in(v);
    if (!in) return -1;
    HIP_TRY(hipStreamSynchronize(v->stream), return -1);
    return 0;
}

// Like every entry point that takes a handle, the two rebinds go through Entry: they may call hipHostUnregister (on the
// handle's device, not whatever device the calling thread has current) and a failure sticks to the handle.
int drone_vec_bind_actions(DroneVec* v, float* actions) {
    Entry in(v);
    if (!in || !idle(v, "bind_actions")) return -1;
    if (!actions) { set_err("bind_actions: NULL argument"); return -1; }
    if (v->host_buffers) {
        // an unregistered buffer: back to the mirror transport — unless the actions already go through a stand-in, which
        // takes them from wherever the caller keeps them
        if (v->zero_copy && actions != v->u_act && !v->bounce[1]) leave_zero_copy(v);
        unpin_if_rebound(v, 1, actions);
        v->u_act = actions;  // copied (pageable unless the caller pinned it) at the next step
    } else {
        if (reinterpret_cast<uintptr_t>(actions) & 15u) { set_err("actions must be 16-byte aligned"); return -1; }
        v->u_act = actions;
        v->dv.act = actions;
    }
    return 0;
}

int drone_vec_bind_outputs(DroneVec* v, float* observations, float* rewards, unsigned char* terminals, unsigned char* truncations) {
    Entry in(v);
    if (!in || !idle(v, "bind_outputs")) return -1;
    if (!observations || !rewards || !terminals || !truncations) { set_err("bind_outputs: NULL argument"); return -1; }
    if (v->gather && v->gather->peer) { set_err("bind_outputs: the peer-store exchange owns the output bindings (drone_vec_gather_close first)"); return -1; }
    if (!v->host_buffers) {
        if ((reinterpret_cast<uintptr_t>(observations) & 15u) || (reinterpret_cast<uintptr_t>(rewards) & 3u)) {
            set_err("device buffers must be 16-byte aligned (observations) and 4-byte aligned (rewards)");
            return -1;
        }
        v->dv.obs = observations; v->dv.rew = rewards; v->dv.term = terminals; v->dv.trunc = truncations;
    }
    // host mode: the device mirrors stay; the next step copies out to the new addresses
    // (pageable unless the caller pinned them)
    if (v->host_buffers && v->zero_copy &&
        ((observations != v->u_obs && !v->bounce[0]) || (rewards != v->u_rew && !v->bounce[2]) || (terminals != v->u_term && !v->bounce[3]) ||
         (truncations != v->u_trunc && !v->bounce[4])))
        leave_zero_copy(v);  // a directly mapped buffer was replaced (stand-ins deliver to wherever the caller points)
    if (v->host_buffers) {
        unpin_if_rebound(v, 0, observations);
        unpin_if_rebound(v, 2, rewards);
        unpin_if_rebound(v, 3, terminals);
        unpin_if_rebound(v, 4, truncations);
    }
    v->u_obs = observations; v->u_rew = rewards; v->u_term = terminals; v->u_trunc = truncations;
    return 0;
}

int drone_vec_fill_random_actions(DroneVec* v, float* actions, uint32_t gstep) {
    Entry in(v);
    if (!in || !idle(v, "fill_random_actions")) return -1;
    if (!actions) { set_err("fill_random_actions: NULL buffer"); return -1; }
    if (v->host_buffers && v->zero_copy && actions == v->u_act) {
        // the bound action buffer is mapped (itself or through its stand-in): the kernel writes it over PCIe, no copy command
        HIP_TRY(launch_fill_actions(v->dv, v->m_act, gstep, v->stream), return -1);
        if (!wait_zero_copy(v)) return -1;
        if (v->bounce[1]) memcpy(v->u_act, v->bounce[1], v->bounce_bytes[1]);
    } else if (v->host_buffers) {
        // generate on the device into the action mirror, then hand the host its copy
        HIP_TRY(launch_fill_actions(v->dv, v->d_act, gstep, v->stream), return -1);
        HIP_TRY(hipMemcpyAsync(actions, v->d_act, (size_t)v->n * DRONE_ACT_DIM * sizeof(float), hipMemcpyDeviceToHost, v->stream), return -1);
        HIP_TRY(hipStreamSynchronize(v->stream), return -1);
    } else {
        if (reinterpret_cast<uintptr_t>(actions) & 15u) { set_err("actions must be 16-byte aligned"); return -1; }
        HIP_TRY(launch_fill_actions(v->dv, actions, gstep, v->stream), return -1);
    }
    return 0;
}

uint32_t drone_vec_gstep(const DroneVec* v) {
    if (!v) return 0u;
    if (v->dv.ctr) {  // graph-safe stepping: the device owns the counter (replays advance it without a host call)
        Entry in(v);
        if (in) (void)pull_counters(const_cast<DroneVec*>(v));
    }
    return v->gstep;
}

int drone_vec_set_gstep(DroneVec* v, uint32_t gstep) {
    Entry in(v);
    if (!in || !idle(v, "set_gstep")) return -1;
    if (!pull_counters(v)) return -1;  // keep the device's step-launch count
    v->gstep = gstep;
    v->list_valid = false;
    return push_counters(v) ? 0 : -1;
}

int drone_vec_enable_graph_capture(DroneVec* v, int on) {
    Entry in(v);
    if (!in || !idle(v, "enable_graph_capture")) return -1;
    if (v->host_buffers) { set_err("graph-safe stepping needs device buffers (host-buffer steps end in a stream sync, which cannot be captured)"); return -1; }
    if (on && v->gather && v->gather->peer) {  // the round numbers of the handshake are launch arguments kept by the host; a replay would wait for / publish a stale round
        set_err("graph-safe stepping cannot be combined with the peer-store exchange (its handshake's round numbers are host state baked into each launch); drone_vec_gather_close first");
        return -1;
    }
    if (on && !v->dv.ctr) {
        HIP_TRY(hipMalloc((void**)&v->dv.ctr, 3 * sizeof(uint32_t)), return -1);
        if (!push_counters(v)) return -1;
    } else if (!on && v->dv.ctr) {
        if (!pull_counters(v)) return -1;
        HIP_TRY(hipStreamSynchronize(v->stream), return -1);
        (void)hipFree(v->dv.ctr);
        v->dv.ctr = nullptr;
    }
    return 0;
}

int drone_vec_status(const DroneVec* v) { return v ? v->status : -1; }
const char* drone_vec_status_message(const DroneVec* v) { return v ? v->status_msg : "handle is NULL"; }
void drone_vec_clear_status(DroneVec* v) {
    if (!v) return;
    v->status = 0;
    v->status_msg[0] = 0;
}
int drone_vec_num_envs(const DroneVec* v) { return v ? v->n : 0; }

// ---- AoS import / export (tests, checkpoints): plain copies + host repack ----
namespace {

// host image of the tiles that cover envs [first, first + count) plus the matching pieces of the two cold planes
struct StateImage {
    uint32_t nph, tile0, ntiles, first;
    std::vector<float4> hot, cold;  // hot: the covered tiles, laid out like the device region but for ntiles * 64 drones; cold: [2][count]
    float4& at(uint32_t plane, uint32_t env) { return hot[hot_index(nph, plane, env - tile0 * kTile, ntiles * kTile)]; }
};

// copy the covered part of the hot region between the device and the image (one piece when tiled, one per plane otherwise)
bool image_copy(DroneVec* v, StateImage& im, bool to_device) {
#if DRONE_TILED_STATE
    float4* dev = v->dv.planes + (size_t)im.tile0 * im.nph * kTile;
    if (to_device) HIP_TRY(hipMemcpyAsync(dev, im.hot.data(), sizeof(float4) * im.hot.size(), hipMemcpyHostToDevice, v->stream), return false);
    else HIP_TRY(hipMemcpyAsync(im.hot.data(), dev, sizeof(float4) * im.hot.size(), hipMemcpyDeviceToHost, v->stream), return false);
#else
    const size_t w = (size_t)im.ntiles * kTile;
    for (uint32_t p = 0; p < im.nph; p++) {
        float4* dev = v->dv.planes + (size_t)p * v->n_pad + (size_t)im.tile0 * kTile;
        float4* host = im.hot.data() + (size_t)p * w;
        if (to_device) HIP_TRY(hipMemcpyAsync(dev, host, sizeof(float4) * w, hipMemcpyHostToDevice, v->stream), return false);
        else HIP_TRY(hipMemcpyAsync(host, dev, sizeof(float4) * w, hipMemcpyDeviceToHost, v->stream), return false);
    }
#endif
    return true;
}

bool image_fetch(DroneVec* v, int first, int count, StateImage& im) {
    im.nph = hot_planes(v->cfg.task, v->dv.derived_target != 0);
    im.first = (uint32_t)first;
    im.tile0 = (uint32_t)first / kTile;
    im.ntiles = ((uint32_t)(first + count) + kTile - 1) / kTile - im.tile0;
    im.hot.resize((size_t)im.ntiles * im.nph * kTile);
    im.cold.resize((size_t)2 * count);
    if (count == 0) return true;
    if (!image_copy(v, im, false)) return false;
    for (int k = 0; k < 2; k++)
        HIP_TRY(hipMemcpyAsync(im.cold.data() + (size_t)k * count, v->dv.cold + (size_t)k * v->stride + first, sizeof(float4) * count, hipMemcpyDeviceToHost, v->stream), return false);
    HIP_TRY(hipStreamSynchronize(v->stream), return false);
    return true;
}

}  // namespace

int drone_vec_get_state(DroneVec* v, DroneStateRow* rows, int first, int count) {
    Entry in(v);
    if (!in || !idle(v, "get_state")) return -1;
    if (!rows || first < 0 || count < 0 || first + count > v->n) { set_err("get_state: bad range"); return -1; }
    StateImage im;
    if (!image_fetch(v, first, count, im)) return -1;
    auto u = [](float f) { uint32_t x; memcpy(&x, &f, 4); return x; };
    const bool aux = im.nph == 7, dt = v->dv.derived_target != 0;
    for (int k = 0; k < count; k++) {
        const uint32_t e = (uint32_t)(first + k);
        const float4 a = im.at(kP0, e), b = im.at(kP1, e), c = im.at(kP2, e), d = im.at(kP3, e), ee = im.at(kP4, e);
        const float4 t = dt ? make_float4(0.f, 0.f, 0.f, 0.f) : im.at(kPT, e);
        const float4 w = aux ? im.at(kPW, e) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 l0 = im.cold[k], l1 = im.cold[(size_t)count + k];
        DroneStateRow& r = rows[k];
        r.pos[0] = a.x; r.pos[1] = a.y; r.pos[2] = a.z; r.vel[0] = a.w;
        r.vel[1] = b.x; r.vel[2] = b.y; r.quat[0] = b.z; r.quat[1] = b.w;
        r.quat[2] = c.x; r.quat[3] = c.y; r.omega[0] = c.z; r.omega[1] = c.w;
        r.omega[2] = d.x; r.rpm[0] = d.y; r.rpm[1] = d.z; r.rpm[2] = d.w;
        r.rpm[3] = ee.x; r.ep_return = ee.y;
        if (dt) {  // derived-target layout: counters packed in P4, the target re-derived as the kernels do
            r.tick = u(ee.z) & 0xFFFFu; r.score_count = u(ee.z) >> 16; r.episode = u(ee.w);
            derive_target(v->kp, v->kp.env_offset + e, r.episode, r.target);
        } else {
            r.tick = u(ee.z); r.score_count = u(ee.w);
            r.target[0] = t.x; r.target[1] = t.y; r.target[2] = t.z; r.episode = u(t.w);
        }
        r.wind[0] = w.x; r.wind[1] = w.y; r.wind[2] = w.z;
        r.perf_sum = l0.x; r.score_sum = l0.y; r.ret_sum = l0.z; r.len_sum = l0.w;
        r.n_sum = l1.x; r.oob_sum = l1.y;
    }
    return 0;
}

int drone_vec_set_state(DroneVec* v, const DroneStateRow* rows, int first, int count) {
    Entry in(v);
    if (!in || !idle(v, "set_state")) return -1;
    if (!rows || first < 0 || count < 0 || first + count > v->n) { set_err("set_state: bad range"); return -1; }
    if (count == 0) return 0;
    // the tiles at the edges of the range also hold neighbours: fetch, patch the rows, write the tiles back
    StateImage im;
    if (!image_fetch(v, first, count, im)) return -1;
    auto f = [](uint32_t x) { float y; memcpy(&y, &x, 4); return y; };
    const bool aux = im.nph == 7, dt = v->dv.derived_target != 0;
    for (int k = 0; k < count; k++) {
        const uint32_t e = (uint32_t)(first + k);
        const DroneStateRow& r = rows[k];
        if (dt) {  // this layout stores no target: the row's must be the one its (env, episode) implies, and the counters must fit
            float want[3];
            derive_target(v->kp, v->kp.env_offset + e, r.episode, want);
            if (memcmp(want, r.target, sizeof(want)) != 0 || r.tick > 0xFFFFu || r.score_count > 0xFFFFu) {
                set_err("set_state: env %u: the derived-target layout (hover / swarm, DRONE_DERIVED_TARGET) cannot hold a target other than the one "
                        "SPEC.md section 6 draws for (env, episode), nor counters beyond 65535; create the handle with DRONE_DERIVED_TARGET=0 for free-form states", e);
                return -1;
            }
        }
        im.at(kP0, e) = make_float4(r.pos[0], r.pos[1], r.pos[2], r.vel[0]);
        im.at(kP1, e) = make_float4(r.vel[1], r.vel[2], r.quat[0], r.quat[1]);
        im.at(kP2, e) = make_float4(r.quat[2], r.quat[3], r.omega[0], r.omega[1]);
        im.at(kP3, e) = make_float4(r.omega[2], r.rpm[0], r.rpm[1], r.rpm[2]);
        if (dt) {
            im.at(kP4, e) = make_float4(r.rpm[3], r.ep_return, f(r.tick | (r.score_count << 16)), f(r.episode));
        } else {
            im.at(kP4, e) = make_float4(r.rpm[3], r.ep_return, f(r.tick), f(r.score_count));
            im.at(kPT, e) = make_float4(r.target[0], r.target[1], r.target[2], f(r.episode));
        }
        if (aux) im.at(kPW, e) = make_float4(r.wind[0], r.wind[1], r.wind[2], 0.0f);
        im.cold[k] = make_float4(r.perf_sum, r.score_sum, r.ret_sum, r.len_sum);
        im.cold[(size_t)count + k] = make_float4(r.n_sum, r.oob_sum, 0.0f, 0.0f);
    }
    if (!image_copy(v, im, true)) return -1;
    for (int k = 0; k < 2; k++)
        HIP_TRY(hipMemcpyAsync(v->dv.cold + (size_t)k * v->stride + first, im.cold.data() + (size_t)k * count, sizeof(float4) * count, hipMemcpyHostToDevice, v->stream), return -1);
    HIP_TRY(hipStreamSynchronize(v->stream), return -1);
    return 0;
}

namespace {

// `cnt_dev`: the device counter of one list, `ids_dev` its ids. The 4-byte count is read and CHECKED after the stream
// has drained (a stack destination of an async copy holds nothing before that).
int fetch_done_list(DroneVec* v, const uint32_t* cnt_dev, const uint32_t* ids_dev, uint32_t* ids, int cap) {
    uint32_t cnt = 0;
    HIP_TRY(hipMemcpyAsync(&cnt, cnt_dev, sizeof(uint32_t), hipMemcpyDeviceToHost, v->stream), return -1);
    HIP_TRY(hipStreamSynchronize(v->stream), return -1);
    if (cnt > (uint32_t)v->n) { set_err("done list count %u exceeds num_envs %d (corrupt counter)", cnt, v->n); return -1; }
    const int take = (int)cnt < cap ? (int)cnt : cap;
    if (ids && take > 0) {
        HIP_TRY(hipMemcpyAsync(ids, ids_dev, sizeof(uint32_t) * take, hipMemcpyDeviceToHost, v->stream), return -1);
        HIP_TRY(hipStreamSynchronize(v->stream), return -1);
    }
    return (int)cnt;
}

}  // namespace

int drone_vec_done_list(DroneVec* v, uint32_t* ids, int cap) {
    Entry in(v);
    if (!in || !idle(v, "done_list")) return -1;
    if (!v->dv.done_ids) { set_err("done list not enabled (compact_done=0)"); return -1; }
    if (!pull_counters(v)) return -1;
    if (!v->list_valid || v->step_launches == 0) return 0;  // after reset / after a fused rollout there is no list
    return fetch_done_list(v, v->dv.done_count + ((v->step_launches - 1u) & 1u), v->dv.done_ids, ids, cap);
}

int drone_vec_done_list_at(DroneVec* v, int k, uint32_t* ids, int cap) {
    Entry in(v);
    if (!in) return -1;
    if (!v->cfg.compact_done) { set_err("done list not enabled (compact_done=0)"); return -1; }
    if (v->many_k <= 0) { set_err("done_list_at: the last path call was not drone_vec_step_many"); return -1; }
    if (k < 0 || k >= v->many_k) { set_err("done_list_at: step %d outside the last step_many's %d steps", k, v->many_k); return -1; }
    return fetch_done_list(v, v->many_count + k, v->many_ids + (size_t)k * (size_t)v->n, ids, cap);
}

void* drone_device_malloc(int device, size_t bytes) {
    g_err[0] = 0;
    DeviceRestore restore;
    void* p = nullptr;
    HIP_TRY(hipSetDevice(device), return nullptr);
    HIP_TRY(hipMalloc(&p, bytes ? bytes : 1), return nullptr);
    HIP_TRY(hipMemset(p, 0, bytes ? bytes : 1), { (void)hipFree(p); return nullptr; });
    return p;
}

void drone_device_free(int device, void* p) {
    if (!p) return;
    DeviceRestore restore;
    if (hipSetDevice(device) == hipSuccess) (void)hipFree(p);
    else (void)hipGetLastError();
}

int drone_vec_copy_to_host(DroneVec* v, void* host_dst, const void* device_src, size_t bytes) {
    Entry in(v);
    if (!in) return -1;
    if (!host_dst || !device_src) { set_err("copy_to_host: NULL argument"); return -1; }
    HIP_TRY(hipMemcpyAsync(host_dst, device_src, bytes, hipMemcpyDeviceToHost, v->stream), return -1);
    HIP_TRY(hipStreamSynchronize(v->stream), return -1);
    if (v->gather && v->gather->peer && !peer_check_err(v->gather)) return -1;  // a wait ahead of this copy gave up: the batch is not this round's
    return 0;
}

int drone_vec_timer_start(DroneVec* v) {
    Entry in(v);
    if (!in) return -1;
    HIP_TRY(hipEventRecord(v->ev0, v->stream), return -1);
    return 0;
}

int drone_vec_timer_stop(DroneVec* v, float* elapsed_ms) {
    Entry in(v);
    if (!in || !elapsed_ms) return -1;
    HIP_TRY(hipEventRecord(v->ev1, v->stream), return -1);
    HIP_TRY(hipEventSynchronize(v->ev1), return -1);
    HIP_TRY(hipEventElapsedTime(elapsed_ms, v->ev0, v->ev1), return -1);
    return 0;
}

#if defined(DRONE_STAMPS) && DRONE_STAMPS
// diagnostic build only; not part of include/drone_vec.h
int drone_debug_stamps(DroneVec* v, unsigned long long* out, int max_rows) {
    Entry in(v);
    if (!in || !v->dv.stamps) return -1;
    const int rows = (int)(v->n_pad / 64) < max_rows ? (int)(v->n_pad / 64) : max_rows;
    HIP_TRY(hipMemcpyAsync(out, v->dv.stamps, sizeof(unsigned long long) * kStampSlots * rows, hipMemcpyDeviceToHost, v->stream), return -1);
    HIP_TRY(hipStreamSynchronize(v->stream), return -1);
    return rows;
}
#endif

// ---- host-boundary all-gather (RCCL) ----
int drone_gather_unique_id(unsigned char* id) {
    g_err[0] = 0;
    if (!id) { set_err("gather_unique_id: NULL buffer"); return -1; }
    Rccl* R = rccl();
    if (!R) return -1;
    static_assert(sizeof(ncclUniqueId) == DRONE_GATHER_ID_BYTES, "DRONE_GATHER_ID_BYTES must match ncclUniqueId");
    ncclUniqueId u;
    RCCL_TRY(R, R->GetUniqueId(&u), return -1);
    memcpy(id, &u, sizeof(u));
    return 0;
}

int drone_vec_gather_init_root(DroneVec* v, const unsigned char* id, int rank, int world, const int* counts, int root,
                               float* all_observations, float* all_rewards, unsigned char* all_terminals, unsigned char* all_truncations) {
    Entry in(v);
    if (!in || !idle(v, "gather_init")) return -1;
    if (v->gather) { set_err("gather already initialised on this handle"); return -1; }
    if (!id || world < 1 || rank < 0 || rank >= world) { set_err("gather_init: bad id / rank %d / world %d", rank, world); return -1; }
    if (root < -1 || root >= world) { set_err("gather_init: root %d outside [-1, %d)", root, world); return -1; }
    const bool receives = root < 0 || root == rank;  // only a receiving rank needs the global buffers
    if (receives && (!all_observations || !all_rewards || !all_terminals || !all_truncations)) { set_err("gather_init: NULL global buffer"); return -1; }
    Rccl* R = rccl();
    if (!R) return -1;
    Gather* g = new (std::nothrow) Gather();
    if (!g) { set_err("out of memory"); return -1; }
    v->gather = g;
    g->rank = rank;
    g->world = world;
    g->root = root;
    g->counts.resize(world);
    g->offsets.resize(world);
    for (int r = 0; r < world; r++) {
        const int c = counts ? counts[r] : v->n;
        if (c <= 0) { set_err("gather_init: counts[%d] = %d", r, c); gather_destroy(v); return -1; }
        g->counts[r] = (size_t)c;
        g->offsets[r] = g->total;
        g->total += (size_t)c;
        if (c != v->n) g->equal = false;
    }
    if (getenv("DRONE_GATHER_FORCE_V")) g->equal = false;  // tests: take the all-gather-v branch even with equal shards
    if (g->counts[rank] != (size_t)v->n) { set_err("gather_init: counts[rank] = %zu but this handle has %d envs", g->counts[rank], v->n); gather_destroy(v); return -1; }
    const size_t od = (size_t)drone_obs_dim(v->cfg.task);
    if (v->host_buffers) {
        // the collective reads device memory: step into the device mirrors, gather into staging, copy the batch out
        if (v->zero_copy) leave_zero_copy(v);
        if (receives) {
            g->own_staging = true;
            g->h_obs = all_observations; g->h_rew = all_rewards; g->h_term = all_terminals; g->h_trunc = all_truncations;
            // The local output buffers were pinned at init. Where they are slices of the global ones (the usual layout), a
            // copy into the whole global buffer would then span pinned and pageable pages, which HIP rejects: drop the
            // local pins (the mirror transport does not need them) and pin the global buffers whole instead, best effort.
            for (int slot : {0, 2, 3, 4})
                if (v->registered[slot]) { host_unregister(v->registered_ptr[slot], v, "local output (gather takes over)"); v->registered[slot] = false; }
            void* hosts[4] = {all_observations, all_rewards, all_terminals, all_truncations};
            const size_t bytes[4] = {g->total * od * sizeof(float), g->total * sizeof(float), g->total, g->total};
            for (int k = 0; k < 4; k++) {  // pinned only when the pages are the buffer's own (pin_caller_buffer's rule)
                const bool own_pages = (reinterpret_cast<uintptr_t>(hosts[k]) % kPage) == 0 && v->cfg.host_pages_exclusive;
                g->h_registered[k] = own_pages && !already_pinned(hosts[k], bytes[k]) &&
                                     host_register(hosts[k], (bytes[k] + kPage - 1) / kPage * kPage, v, "global gather buffer") == hipSuccess;
                if (own_pages && !g->h_registered[k]) (void)hipGetLastError();
            }
#define G_TRY(expr) HIP_TRY(expr, { gather_destroy(v); return -1; })
            G_TRY(hipMalloc((void**)&g->g_obs, g->total * od * sizeof(float)));
            G_TRY(hipMalloc((void**)&g->g_rew, g->total * sizeof(float)));
            G_TRY(hipMalloc((void**)&g->g_term, g->total));
            G_TRY(hipMalloc((void**)&g->g_trunc, g->total));
#undef G_TRY
        }
    } else if (receives) {
        if (reinterpret_cast<uintptr_t>(all_observations) & 15u) { set_err("gather_init: global observations must be 16-byte aligned"); gather_destroy(v); return -1; }
        g->g_obs = all_observations; g->g_rew = all_rewards; g->g_term = all_terminals; g->g_trunc = all_truncations;
    }
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    RCCL_TRY(R, R->CommInitRank(&g->comm, world, u, rank), { g->comm = nullptr; gather_destroy(v); return -1; });
    return 0;
}

int drone_vec_gather_init(DroneVec* v, const unsigned char* id, int rank, int world, const int* counts,
                          float* all_observations, float* all_rewards, unsigned char* all_terminals, unsigned char* all_truncations) {
    return drone_vec_gather_init_root(v, id, rank, world, counts, -1, all_observations, all_rewards, all_terminals, all_truncations);
}

// ---- the same exchange as peer stores (round 4) ----
namespace {
struct PeerBuf {
    hipIpcMemHandle_t handle;  // of the ALLOCATION the buffer lives in (a torch tensor sits somewhere inside a caching-allocator segment)
    uint64_t offset;           // of the buffer inside it
};
static_assert(sizeof(PeerBuf) * 4 == DRONE_PEER_TOKEN_BYTES, "DRONE_PEER_TOKEN_BYTES must hold four IPC handles + offsets");
}  // namespace

int drone_vec_gather_peer_export(DroneVec* v, float* all_observations, float* all_rewards, unsigned char* all_terminals,
                                 unsigned char* all_truncations, unsigned char* token) {
    Entry in(v);
    if (!in || !idle(v, "gather_peer_export")) return -1;
    if (v->host_buffers) { set_err("gather_peer_export: the peer-store exchange needs device buffers (peers write HBM, not host memory)"); return -1; }
    if (!all_observations || !all_rewards || !all_terminals || !all_truncations || !token) { set_err("gather_peer_export: NULL argument"); return -1; }
    if (reinterpret_cast<uintptr_t>(all_observations) & 15u) { set_err("gather_peer_export: global observations must be 16-byte aligned"); return -1; }
    void* bufs[4] = {all_observations, all_rewards, all_terminals, all_truncations};
    PeerBuf out[4];
    memset(out, 0, sizeof(out));
    for (int k = 0; k < 4; k++) {
        hipDeviceptr_t base = nullptr;
        size_t size = 0;
        HIP_TRY(hipMemGetAddressRange(&base, &size, bufs[k]), return -1);
        HIP_TRY(hipIpcGetMemHandle(&out[k].handle, base), return -1);
        out[k].offset = (uint64_t)(static_cast<char*>(bufs[k]) - static_cast<char*>(base));
    }
    memcpy(token, out, sizeof(out));
    v->px_obs = all_observations; v->px_rew = all_rewards; v->px_term = all_terminals; v->px_trunc = all_truncations;
    return 0;
}

int drone_vec_gather_init_peer(DroneVec* v, const unsigned char* token, void* shared_flags, int rank, int world, const int* counts, int root) {
    Entry in(v);
    if (!in || !idle(v, "gather_init_peer")) return -1;
    if (v->gather) { set_err("gather already initialised on this handle"); return -1; }
    if (v->host_buffers) { set_err("gather_init_peer: the peer-store exchange needs device buffers"); return -1; }
    if (v->dv.ctr) { set_err("gather_init_peer: not on a handle in graph-safe mode (drone_vec_enable_graph_capture): a captured launch would replay the handshake with a stale round number"); return -1; }
    if (!token || !shared_flags || world < 1 || rank < 0 || rank >= world || root < 0 || root >= world) { set_err("gather_init_peer: bad token / flags / rank %d / world %d / root %d", rank, world, root); return -1; }
    if ((reinterpret_cast<uintptr_t>(shared_flags) % kPage) != 0 || (size_t)(world + 1) * 4u > kPage) { set_err("gather_init_peer: the flag block must be one 4 KiB page of memory shared by all ranks, page-aligned (world <= 1023)"); return -1; }
    if (rank == root && !v->px_obs) { set_err("gather_init_peer: the root must export its global buffers first (drone_vec_gather_peer_export)"); return -1; }
    Gather* g = new (std::nothrow) Gather();
    if (!g) { set_err("out of memory"); return -1; }
    g->peer = true;
    g->rank = rank; g->world = world; g->root = root;
    g->counts.resize(world);
    g->offsets.resize(world);
    for (int r = 0; r < world; r++) {
        const int c = counts ? counts[r] : v->n;
        if (c <= 0) { set_err("gather_init_peer: counts[%d] = %d", r, c); delete g; return -1; }
        g->counts[r] = (size_t)c;
        g->offsets[r] = g->total;
        g->total += (size_t)c;
    }
    if (g->counts[rank] != (size_t)v->n) { set_err("gather_init_peer: counts[rank] = %zu but this handle has %d envs", g->counts[rank], v->n); delete g; return -1; }
    g->own_obs = v->dv.obs; g->own_rew = v->dv.rew; g->own_term = v->dv.term; g->own_trunc = v->dv.trunc;
    if (v->tune) { v->dv.order = v->tune->table; tune_free(v); }  // (DRONE_AUTOTUNE=1) no measuring under the exchange: the table's choice stands
    g->own_order = v->dv.order;
    v->gather = g;  // from here on gather_destroy undoes whatever was done
    // the flag page: pinned + mapped so that stream memory operations can reach it (it owns its page: the rule of pin_caller_buffer)
    if (!already_pinned(shared_flags, kPage)) {
        HIP_TRY(host_register(shared_flags, kPage, v, "peer-store flag page"), { gather_destroy(v); return -1; });
        g->flags_registered = true;
    }
    g->flags = static_cast<volatile uint32_t*>(shared_flags);
    g->d_flags = static_cast<char*>(mapped_ptr(shared_flags));
    if (!g->d_flags) { set_err("gather_init_peer: the flag page could not be mapped into the device address space"); gather_destroy(v); return -1; }
    const char* hw = getenv("DRONE_PEER_HOST_WAIT");
    if (hw && *hw && atoi(hw) != 0) g->gpu_waits = false;
    const char* sw = getenv("DRONE_PEER_STREAM_WRITES");  // 1: publish flags with hipStreamWriteValue32 where the runtime takes the page
    if (sw && *sw) g->stream_writes = atoi(sw) != 0;
    const char* ik = getenv("DRONE_PEER_INKERNEL");       // 0: the flag publications as one-wave launches of their own (round 4's form; A/B)
    if (ik && *ik) g->in_kernel = atoi(ik) != 0;
    if (g->gpu_waits) {
        HIP_TRY(hipMalloc((void**)&g->d_arrive, kPeerBlockBytes), { gather_destroy(v); return -1; });
        HIP_TRY(hipMemsetAsync(g->d_arrive, 0, kPeerBlockBytes, v->stream), { gather_destroy(v); return -1; });
        if (v->dv.order & 12u) {  // the peer instantiations of the step kernel carry no load hints (a speed choice, never a result)
            v->dv.order &= 3u;
            write_variant(v, " peer=1");
        }
        void* he = nullptr;
        HIP_TRY(hipHostMalloc(&he, 64, hipHostMallocMapped), { gather_destroy(v); return -1; });
        g->h_err = static_cast<uint32_t*>(he);
        *g->h_err = 0u;
        g->d_err = static_cast<uint32_t*>(mapped_ptr(he));
        if (!g->d_err) { set_err("gather_init_peer: the error word could not be mapped"); gather_destroy(v); return -1; }
        g->budget_ticks = (unsigned long long)peer_timeout_ms() * 100000ull;  // s_memrealtime counts at 100 MHz
    }
    const size_t od = (size_t)drone_obs_dim(v->cfg.task), o = g->offsets[rank];
    char* glob[4];
    if (rank == root) {
        glob[0] = reinterpret_cast<char*>(v->px_obs); glob[1] = reinterpret_cast<char*>(v->px_rew);
        glob[2] = reinterpret_cast<char*>(v->px_term); glob[3] = reinterpret_cast<char*>(v->px_trunc);
    } else {
        PeerBuf in4[4];
        void* opened[4] = {nullptr, nullptr, nullptr, nullptr};
        memcpy(in4, token, sizeof(in4));
        for (int k = 0; k < 4; k++) {
            // several of the four buffers may live in ONE allocation (a caching allocator's segment): map each allocation once
            void* base = nullptr;
            for (int j = 0; j < k && !base; j++)
                if (memcmp(&in4[j].handle, &in4[k].handle, sizeof(hipIpcMemHandle_t)) == 0) base = opened[j];
            if (!base) {
                HIP_TRY(hipIpcOpenMemHandle(&base, in4[k].handle, hipIpcMemLazyEnablePeerAccess), { gather_destroy(v); return -1; });
                g->peer_base[k] = base;  // closed by gather_destroy
            }
            opened[k] = base;
            glob[k] = static_cast<char*>(base) + in4[k].offset;
        }
    }
    g->g_obs = reinterpret_cast<float*>(glob[0]); g->g_rew = reinterpret_cast<float*>(glob[1]);
    g->g_term = reinterpret_cast<unsigned char*>(glob[2]); g->g_trunc = reinterpret_cast<unsigned char*>(glob[3]);
    // from now on this rank's kernels write ITS ROWS OF THE ROOT'S BUFFERS: local HBM on the root, xGMI stores elsewhere
    v->dv.obs = g->g_obs + o * od;   // row offsets are multiples of 80 / 96 bytes: 16-byte alignment of the base carries over
    v->dv.rew = g->g_rew + o;
    v->dv.term = g->g_term + o;
    v->dv.trunc = g->g_trunc + o;
    if (reinterpret_cast<uintptr_t>(v->dv.obs) & 15u) { set_err("gather_init_peer: this rank's rows of the global observations are not 16-byte aligned"); gather_destroy(v); return -1; }
    return 0;
}

int drone_vec_gather(DroneVec* v) {
    Entry in(v);
    if (!in || !idle(v, "gather")) return -1;
    Gather* g = v->gather;
    if (!g) { set_err("gather not initialised (drone_vec_gather_init)"); return -1; }
    if (g->peer) {
        // Peer stores: the rows are already where they belong (the kernels wrote them there). A non-root rank publishes
        // "my launch #seq has landed" behind its kernel; the root's stream waits until every other rank has said so.
        g->launched = false;
        if (!peer_check_err(g)) return -1;
        g->seq += 1u;
        if (g->rank != g->root) {
            // the launch this call follows publishes the round itself when its last workgroup ends (LaunchSig): nothing to enqueue.
            // Anything else (the separate-launch forms; a gather that follows no launch) gets the one-wave post.
            if (g->launch_posts == g->seq) { g->launch_posts = 0; return 0; }
            return peer_post(v, g, g->rank, g->seq) ? 0 : -1;
        }
        if (g->world > 1 && !peer_wait_ge(v, g, 0, g->world, g->root, g->seq)) return -1;
        return 0;
    }
    Rccl* R = rccl();
    if (!R) return -1;
    const size_t od = (size_t)drone_obs_dim(v->cfg.task);
    const size_t n = (size_t)v->n;
    // sources: whatever the kernels currently write (the caller's device buffers or the mirrors)
    const float* s_obs = v->dv.obs;
    const float* s_rew = v->dv.rew;
    const unsigned char* s_term = v->dv.term;
    const unsigned char* s_trunc = v->dv.trunc;
    // one grouped launch for the four buffers; a send buffer that already is this rank's slice of the
    // global buffer makes the collective in-place
    if (g->root >= 0 && g->rank == g->root) {
        // the root's own rows need no link: a device copy, unless the kernels already write them in place
        const size_t o = g->offsets[g->rank];
        if (s_obs != g->g_obs + o * od) HIP_TRY(hipMemcpyAsync(g->g_obs + o * od, s_obs, n * od * sizeof(float), hipMemcpyDeviceToDevice, v->stream), return -1);
        if (s_rew != g->g_rew + o) HIP_TRY(hipMemcpyAsync(g->g_rew + o, s_rew, n * sizeof(float), hipMemcpyDeviceToDevice, v->stream), return -1);
        if (s_term != g->g_term + o) HIP_TRY(hipMemcpyAsync(g->g_term + o, s_term, n, hipMemcpyDeviceToDevice, v->stream), return -1);
        if (s_trunc != g->g_trunc + o) HIP_TRY(hipMemcpyAsync(g->g_trunc + o, s_trunc, n, hipMemcpyDeviceToDevice, v->stream), return -1);
    }
    RCCL_TRY(R, R->GroupStart(), return -1);
    bool ok = true;
    if (g->root >= 0) {
        // gather to ONE rank: every other rank sends its rows once; the root receives each rank's rows into their place.
        // Against the all-gather the 7 non-root GPUs of a node stop receiving (and writing to HBM) 7/8 of the batch each.
        if (g->rank != g->root) {
            ok = ok && R->Send(s_obs, n * od, ncclFloat, g->root, g->comm, v->stream) == ncclSuccess;
            ok = ok && R->Send(s_rew, n, ncclFloat, g->root, g->comm, v->stream) == ncclSuccess;
            ok = ok && R->Send(s_term, n, ncclUint8, g->root, g->comm, v->stream) == ncclSuccess;
            ok = ok && R->Send(s_trunc, n, ncclUint8, g->root, g->comm, v->stream) == ncclSuccess;
        } else {
            for (int r = 0; r < g->world && ok; r++) {
                if (r == g->rank) continue;
                const size_t c = g->counts[r], o = g->offsets[r];
                ok = ok && R->Recv(g->g_obs + o * od, c * od, ncclFloat, r, g->comm, v->stream) == ncclSuccess;
                ok = ok && R->Recv(g->g_rew + o, c, ncclFloat, r, g->comm, v->stream) == ncclSuccess;
                ok = ok && R->Recv(g->g_term + o, c, ncclUint8, r, g->comm, v->stream) == ncclSuccess;
                ok = ok && R->Recv(g->g_trunc + o, c, ncclUint8, r, g->comm, v->stream) == ncclSuccess;
            }
        }
    } else if (g->equal) {
        ok = ok && R->AllGather(s_obs, g->g_obs, n * od, ncclFloat, g->comm, v->stream) == ncclSuccess;
        ok = ok && R->AllGather(s_rew, g->g_rew, n, ncclFloat, g->comm, v->stream) == ncclSuccess;
        ok = ok && R->AllGather(s_term, g->g_term, n, ncclUint8, g->comm, v->stream) == ncclSuccess;
        ok = ok && R->AllGather(s_trunc, g->g_trunc, n, ncclUint8, g->comm, v->stream) == ncclSuccess;
    } else {  // ragged shards: one broadcast per rank into its rows (an all-gather-v)
        for (int r = 0; r < g->world && ok; r++) {
            const size_t c = g->counts[r], o = g->offsets[r];
            const bool me = r == g->rank;
            ok = ok && R->Broadcast(me ? (const void*)s_obs : (const void*)(g->g_obs + o * od), g->g_obs + o * od, c * od, ncclFloat, r, g->comm, v->stream) == ncclSuccess;
            ok = ok && R->Broadcast(me ? (const void*)s_rew : (const void*)(g->g_rew + o), g->g_rew + o, c, ncclFloat, r, g->comm, v->stream) == ncclSuccess;
            ok = ok && R->Broadcast(me ? (const void*)s_term : (const void*)(g->g_term + o), g->g_term + o, c, ncclUint8, r, g->comm, v->stream) == ncclSuccess;
            ok = ok && R->Broadcast(me ? (const void*)s_trunc : (const void*)(g->g_trunc + o), g->g_trunc + o, c, ncclUint8, r, g->comm, v->stream) == ncclSuccess;
        }
    }
    RCCL_TRY(R, R->GroupEnd(), return -1);
    if (!ok) { set_err("an RCCL collective of drone_vec_gather failed to enqueue"); return -1; }
    if (v->host_buffers && g->own_staging) {  // receiving ranks only: the batch goes out to the caller's host buffers
        HIP_TRY(hipMemcpyAsync(g->h_obs, g->g_obs, g->total * od * sizeof(float), hipMemcpyDeviceToHost, v->stream), return -1);
        HIP_TRY(hipMemcpyAsync(g->h_rew, g->g_rew, g->total * sizeof(float), hipMemcpyDeviceToHost, v->stream), return -1);
        HIP_TRY(hipMemcpyAsync(g->h_term, g->g_term, g->total, hipMemcpyDeviceToHost, v->stream), return -1);
        HIP_TRY(hipMemcpyAsync(g->h_trunc, g->g_trunc, g->total, hipMemcpyDeviceToHost, v->stream), return -1);
        HIP_TRY(hipStreamSynchronize(v->stream), return -1);
    }
    return 0;
}

void drone_vec_gather_close(DroneVec* v) {
    Entry in(v);
    if (!in) return;
    if (v->stream) (void)hipStreamSynchronize(v->stream);
    gather_destroy(v);
}

}  // extern "C"

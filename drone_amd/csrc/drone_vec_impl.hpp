// drone_vec_impl.hpp — what the translation units behind include/drone_vec.h share: the handle, the exchange state, the
// entry guard and the helpers that cross unit boundaries. Round 6 split the 2 300-line drone_vec.cpp:
//   drone_vec.cpp        the path itself: init / reset / step / rollout / step_many / log / close and the small accessors
//   drone_transport.cpp  host-buffer transports: mirror, zero-copy, pinned stand-ins, the host copy pool's jobs, pinning rules
//   drone_gather.cpp     the host-boundary exchange: RCCL (dlopen'ed) and the peer-store handshake
//   drone_state.cpp      state import / export, done lists, device memory helpers, timers; the opt-in sweep autotuner
// Nothing here is part of the ABI; everything lives in drone_impl (hidden visibility).
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <rccl/rccl.h>  // types only: the library is dlopen'ed on first use (drone_vec_gather_init), never linked

#include "drone_kernels.h"

using namespace drone;  // (an internal header: every unit that includes it is part of this library)

struct Gather;
struct SweepTune;

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

#define DRONE_IMPL_NS namespace drone_impl __attribute__((visibility("hidden")))
DRONE_IMPL_NS {

// the calling thread's error text (512 bytes; drone_last_error). A function, not an `extern thread_local`: a hidden-visibility
// thread_local referenced from another unit makes the compiler call its (non-existent, weak) TLS init function through a
// PC-relative address that is never null in a shared object — the first build of the split crashed in every entry point
// outside drone_vec.cpp that way.
constexpr size_t kErrBytes = 512;
char* err_text();
void set_err(const char* fmt, ...) __attribute__((format(printf, 1, 2)));

#define HIP_TRY(expr, onfail)                                                        \
    do {                                                                             \
        hipError_t e_ = (expr);                                                      \
        if (e_ != hipSuccess) {                                                      \
            set_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            onfail;                                                                  \
        }                                                                            \
    } while (0)

constexpr int kLogMaxGrid = 1024;
constexpr uintptr_t kPage = 4096;

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
        err_text()[0] = 0;
        if (!v) { set_err("handle is NULL"); return; }
        if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; }
        if (prev != v->device) {
            hipError_t e = hipSetDevice(v->device);
            if (e != hipSuccess) { set_err("hipSetDevice(%d) failed: %s", v->device, hipGetErrorString(e)); return; }
        }
        ok = true;
    }
    ~Entry() {
        const char* err = v ? err_text() : nullptr;
        if (v && err[0] && v->status == 0) {
            v->status = 1;
            snprintf(v->status_msg, sizeof(v->status_msg), "%s", err);
        }
        if (v && prev >= 0 && prev != v->device) (void)hipSetDevice(prev);
    }
    explicit operator bool() const { return ok; }
};

// drone_vec.cpp
bool idle(DroneVec* v, const char* what);
uint32_t plane_pad_elems();
bool push_counters(DroneVec* v);
bool pull_counters(DroneVec* v);
void write_variant(DroneVec* v, const char* tuned);

// drone_transport.cpp
bool debug_reg();
hipError_t host_register(void* p, size_t bytes, const void* who, const char* what);
void host_unregister(void* p, const void* who, const char* what);
void choose_host_transport(DroneVec* v);  // drone_vec_init, host-buffer handles: pin / stand-in / mirror (sets dv's buffer pointers)
bool host_to_device_actions(DroneVec* v);
void ensure_flag(DroneVec* v);
bool wait_zero_copy(DroneVec* v);
bool enqueue_host_outputs(DroneVec* v);
bool finish_threaded_copy(DroneVec* v);
bool finish_host_outputs(DroneVec* v);
bool device_to_host_outputs(DroneVec* v);
bool start_threaded_copy(DroneVec* v);   // the pool begins to follow the step's per-chunk words (transport 3)
void* mapped_ptr(void* host);
void drop_bounce(DroneVec* v);
void leave_zero_copy(DroneVec* v);
bool already_pinned(const void* p, size_t bytes);

// drone_gather.cpp
void gather_destroy(DroneVec* v);
bool peer_before_launch(DroneVec* v, LaunchSig* sig);
void peer_launch_failed(DroneVec* v);
bool peer_check_err(Gather* g);

// drone_state.cpp
void tune_free(DroneVec* v);
SweepTune::Pair* tune_before_step(DroneVec* v);

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
extern HostStamps g_stamps;
#define HOST_STAMP(what) g_stamps.at(HostStamps::what)
#else
#define HOST_STAMP(what) ((void)0)
#endif

}  // namespace drone_impl

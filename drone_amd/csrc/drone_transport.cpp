// drone_transport.cpp — how a HOST-buffer handle (the PufferLib drop-in case: numpy buffers) gets its actions to the kernel and
// its outputs back: mirror copies, zero-copy mapping, pinned stand-ins moved by one memcpy or by the host copy pool, and the
// rules for what may be pinned. Split out of drone_vec.cpp in round 6; see drone_vec_impl.hpp.
#include "drone_host_copy.hpp"
#include "drone_vec_impl.hpp"

DRONE_IMPL_NS {

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
#if DRONE_HOST_STAMPS
HostStamps g_stamps;
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

bool start_threaded_copy(DroneVec* v) { return CopyPool::get().try_start(copy_outputs_part, v); }

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

// drone_vec_init, host-buffer handles: which transport this handle gets (comments at want_zero_copy above)
void choose_host_transport(DroneVec* v) {
    void* const host[5] = {v->u_obs, v->u_act, v->u_rew, v->u_term, v->u_trunc};
    const size_t n = (size_t)v->n;
    const int num_envs = v->n;
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
}

}  // namespace drone_impl

using namespace drone_impl;

extern "C" {

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

}  // extern "C"

// drone_gather.cpp — the one exchange step of the path, at the host boundary: RCCL all-gather / gather to
// one rank (librccl dlopen'ed on first use) or peer stores with a flag handshake. Split out of drone_vec.cpp in round 6; see
// drone_vec_impl.hpp.
#include <dlfcn.h>

#include <mutex>
#include <new>

#include "drone_vec_impl.hpp"

DRONE_IMPL_NS {

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


}  // namespace drone_impl

using namespace drone_impl;

extern "C" {

// ---- host-boundary all-gather (RCCL) ----
int drone_gather_unique_id(unsigned char* id) {
    err_text()[0] = 0;
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
        const hipError_t ipc = hipIpcGetMemHandle(&out[k].handle, base);
        if (ipc != hipSuccess) {
            // what fails here in practice: memory that is not ONE plain allocation — a virtual-memory mapping (hipMemCreate / hipMemMap;
            // torch's expandable segments) has no IPC handle. The way out is a plain allocation.
            (void)hipGetLastError();
            set_err("gather_peer_export: hipIpcGetMemHandle failed for buffer %d (%s): the global batch must live in plain device allocations that can be shared "
                    "between processes — not in a virtual-memory mapping (hipMemCreate / hipMemMap, torch's expandable_segments allocator). Allocate it with "
                    "drone_device_malloc (a plain hipMalloc on the handle's device) and export that",
                    k, hipGetErrorString(ipc));
            return -1;
        }
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

// drone_kernels.hip — gfx950 (CDNA4, MI355X) kernels of the drone env.
//
// One lane = one drone. The state a step always touches lives in HBM as float4
// planes interleaved per 64-drone wave tile (drone_params.hpp), so a wave
// streams ONE contiguous 6-7 KiB piece with global_load/store_dwordx4 (16 B per
// lane, one base address + immediate offsets). Outputs leave in whole cache
// lines and non-temporally (nothing on the GPU re-reads them in this path):
//   * observation rows ([N][20] AoS, what a vec-env consumer expects) are
//     transposed through a wave-private LDS tile (wave-scope fences only, no
//     workgroup barrier), so a wave emits five 1-KiB stores instead of 64
//     strided 80-B rows (2.5x on the whole kernel);
//   * terminal / truncation bytes are built from the waves' __ballot masks,
//     gathered per workgroup in LDS and written as 16-B pieces by the first 32
//     lanes — partial-line dword stores of the same bytes cost 11 % of the kernel;
//   * the optional done-id list is compacted with ballot + mbcnt + one atomic
//     per wave;
//   * the rare per-lane plane updates of an ended episode are widened to whole
//     128-B lines when the footprint exceeds the Infinity Cache (whole_lines).
// vmcnt retires in issue order, stores included, and the wait-count pass merges
// both paths' outstanding counts at a join: hence no branch around the loads or
// the state stores, the log-plane loads ahead of the stores, their fold last.
// The sweep order over the envs, the line widening and the action-load hint are
// launch arguments chosen by the host from the step's footprint (DeviceView).
// The 57-word constants block (KParams) reaches the lanes through the kernarg
// segment (scalar loads -> SGPR operands; default) or staged through LDS by each
// workgroup (DRONE_PARAMS_IN_LDS=1).
//
// The path is elementwise: no MFMA. Roofline = HBM for the per-step kernel,
// f32 VALU for the fused rollout.
// Implements SPEC.md; reference file:line cannot be cited (no source in
// /root/reference — .gitmodules:1-3).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "drone_kernels.h"
#include "drone_lane.hpp"

// ---- compile-time variants. What used to be knobs here and lost its measurement now lives as a patch beside its log
// (profiles/r06_pruned/README.md: one patch that puts them all back, each with its log): plain instead of non-temporal output / state stores (+21 % / +4 %), non-temporal state
// loads everywhere, several chunks per workgroup, per-wave outputs in the step kernel, wave-count caps, constants by scalar
// loads from HBM, the hand-issued first loads, scalar-unit reset hashing, priority rotation, the RK4's constants in vector
// registers. ----
#ifndef DRONE_PARAMS_IN_LDS  // 1: stage KParams HBM -> LDS per workgroup (the north-star's wording; tests/test_parity_gpu.py builds and checks it); 0: kernarg scalar loads (-2.4 % step, -13 % rollout)
#define DRONE_PARAMS_IN_LDS 0
#endif
constexpr int kStepMinWaves = 5;  // __launch_bounds__ 2nd argument of the per-step kernel: keeps the race task at 93 VGPRs (97 unbounded), no scratch

// DIAGNOSTIC BUILD ONLY (tools/stamps.py): s_memtime stamps at the phase boundaries of the step kernel, one row per
// wave, to see where a small shard's few microseconds go. Never timed as a whole: the stamps' fences forbid overlaps.
#ifndef DRONE_STAMPS
#define DRONE_STAMPS 0
#endif
#if DRONE_STAMPS
#define DRONE_STAMP(k)                                                                                       \
    do {                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        unsigned long long t_;                                                                               \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                           \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        stamp_[k] = t_;                                                                                      \
    } while (0)
#else
#define DRONE_STAMP(k) do {} while (0)
#endif

#ifndef DRONE_EARLY_ARGS  // the words the per-step kernel's state-load addresses depend on — 0 (shipped since round 6): wherever the compiler sinks their scalar loads; 2: preloaded into SGPRs with the wave (leading scalar kernel arguments + -amdgpu-kernarg-preload-count: Makefile PRELOAD, where the reason for the default is). Round 4; the two forms in between (one scalar batch; the first loads issued by hand) are in profiles/r06_pruned/
#define DRONE_EARLY_ARGS 0
#endif
#if DRONE_EARLY_ARGS != 0 && DRONE_EARLY_ARGS != 2
#error "DRONE_EARLY_ARGS is 0 or 2"
#endif

namespace drone {

namespace {

#include "drone_planes.hpp"  // (inside drone::{anonymous}: the state planes <-> registers, the streaming stores)

struct StepArgs {
    DeviceView v;
    uint32_t gstep;
    uint32_t flags_aligned;  // bit0: terminals 16-B aligned, bit1: truncations 16-B aligned
    uint32_t done_slot;      // which of the two done-list counters this step launch adds to (the host alternates per STEP launch)
    uint32_t nwg;            // workgroups of this launch (= gridDim.x, which the kernel would otherwise fetch from the hidden arguments in a scalar round trip of its own)
    LaunchSig sig;           // what this launch publishes itself (drone_kernels.h): peer-store handshake flags, per-chunk completion words; all null otherwise
#if !DRONE_PARAMS_IN_LDS
    KParams kp;              // constants by value: scalar loads from the kernarg segment
#endif
};

// LDS of one workgroup
struct Shared {
    float4 obs_tile[kWavesPerBlock][kWave * kObsVecMax];  // wave-private observation tiles
    uint64_t masks[2][2][kWavesPerBlock];              // ballot masks: [chunk parity][terminal | truncation][wave]
#if DRONE_PARAMS_IN_LDS
    KParams kp;
#endif
};

#if DRONE_PARAMS_IN_LDS
#define DRONE_PARAMS(sh, a) stage_params((sh).kp, (a).v.kp)
__device__ __forceinline__ const KParams& stage_params(KParams& sp, const uint32_t* __restrict__ kp) {
    if (threadIdx.x < kParamWords) reinterpret_cast<uint32_t*>(&sp)[threadIdx.x] = kp[threadIdx.x];
    __syncthreads();
    return sp;
}
#else
#define DRONE_PARAMS(sh, a) ((a).kp)
#endif

// 4 mask bits -> 4 bytes of 0/1
__device__ __forceinline__ uint32_t spread4(uint32_t nib) { return ((nib & 0xFu) * 0x00204081u) & 0x01010101u; }

// The whole workgroup's outputs for its 256 drones: observation rows, rewards
// are already stored by the caller; here obs + terminal/truncation bytes.
// Every thread of the workgroup must call this (it contains the barrier).
template <int OBSV>
__device__ __forceinline__ void write_outputs(Shared& sh, const DeviceView& v, uint32_t flags_aligned, const float (&o)[DRONE_OBS_DIM_MAX],
                                              bool term, bool trunc, uint32_t i, uint32_t block_base, uint32_t parity = 0) {
    const uint32_t n = v.n;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = threadIdx.x / kWave;
    const uint32_t wave_base = i - lane;
    const bool valid = i < n;
    const uint64_t m_term = __ballot(valid && term);
    const uint64_t m_trunc = __ballot(valid && trunc);
    // a workgroup that walks several chunks alternates between two mask sets: the one barrier per chunk keeps its
    // waves at most one chunk apart
    if (lane == 0) {
        sh.masks[parity][0][wave] = m_term;
        sh.masks[parity][1][wave] = m_trunc;
    }
    // Observation rows: the tile is private to this wave and a wave's LDS operations execute in order, so the
    // transpose needs only compiler ordering (wave-scope fences) — no workgroup barrier on this path. Rows go in
    // row-major ([lane][4*OBSV]: ds_write_b128, conflict-free at the 80-B row stride), come back flat, and leave as
    // OBSV x 1 KiB contiguous stores per wave.
    float4* tile = sh.obs_tile[wave];
#pragma unroll
    for (int k = 0; k < OBSV; k++) tile[lane * OBSV + k] = make_float4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (wave_base + kWave <= n) {  // a full wave (wave-uniform): all reads first, then the stores, no per-piece guards
        float4* dst = reinterpret_cast<float4*>(v.obs + (size_t)wave_base * (4 * OBSV));
        float4 piece[OBSV];
#pragma unroll
        for (int k = 0; k < OBSV; k++) piece[k] = tile[k * kWave + lane];
#pragma unroll
        for (int k = 0; k < OBSV; k++) out_store(&dst[k * kWave + lane], piece[k]);
    } else if (wave_base < n) {  // the ragged last wave
        const uint32_t rows = n - wave_base;
        float4* dst = reinterpret_cast<float4*>(v.obs + (size_t)wave_base * (4 * OBSV));
#pragma unroll
        for (int k = 0; k < OBSV; k++) {
            const uint32_t j = k * kWave + lane;
            if (j < rows * OBSV) out_store(&dst[j], tile[j]);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the reads are done before the tile is written again
    __builtin_amdgcn_wave_barrier();

    __syncthreads();  // the flag bytes are assembled from all four waves' masks

    const bool full = block_base + kBlock <= n;  // workgroup-uniform
    if (full && (flags_aligned & 3u) == 3u) {
        if (threadIdx.x < 2 * kFlagLanes) {  // lanes 0..15: terminals, 16..31: truncations; 16 drones = 16 B each
            const uint32_t which = threadIdx.x / kFlagLanes, j = threadIdx.x % kFlagLanes;
            const uint32_t bits = (uint32_t)(sh.masks[parity][which][j >> 2] >> ((j & 3u) * 16u)) & 0xFFFFu;
            const u4_t packed = {spread4(bits), spread4(bits >> 4), spread4(bits >> 8), spread4(bits >> 12)};
            unsigned char* base = which ? v.trunc : v.term;
            out_store(reinterpret_cast<u4_t*>(base + block_base) + j, packed);
        }
    } else if (valid) {
        v.term[i] = term ? 1 : 0;
        v.trunc[i] = trunc ? 1 : 0;
    }
}

// A FULL wave's outputs of one step, on ONE static path with a fixed number of vector-memory operations and no
// workgroup barrier: the wait-count pass keeps, at every join, the most conservative outstanding-store count of the
// joined paths, and the exec-skip branch the compiler puts around any divergent store is such a path — a single
// guarded store in the loop body makes the wait for the prefetched action row drain the previous step's stores.
// So: every lane stores, always. Observation rows as in write_outputs (wave-private LDS transpose, OBSV x 1 KiB).
// Flag bytes per wave instead of per workgroup (no LDS masks, no barrier): the wave's 64 + 64 bytes leave as eight
// 16-byte pieces; all 64 lanes take part, lanes l and l + 8k writing the same bytes to the same address.
template <int OBSV>
__device__ __forceinline__ void write_outputs_wave(float4* tile, float* obs_wave, unsigned char* term_wave, unsigned char* trunc_wave, bool flags16,
                                                   const float (&o)[DRONE_OBS_DIM_MAX], bool term, bool trunc) {
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint64_t m_term = __ballot(term);
    const uint64_t m_trunc = __ballot(trunc);
#pragma unroll
    for (int k = 0; k < OBSV; k++) tile[lane * OBSV + k] = make_float4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float4* dst = reinterpret_cast<float4*>(obs_wave);
    float4 piece[OBSV];
#pragma unroll
    for (int k = 0; k < OBSV; k++) piece[k] = tile[k * kWave + lane];
#pragma unroll
    for (int k = 0; k < OBSV; k++) out_store(&dst[k * kWave + lane], piece[k]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the reads are done before the tile is written again
    __builtin_amdgcn_wave_barrier();
    if (flags16) {  // launch-uniform: both flag blocks and the env count are 16-byte multiples
        const uint32_t which = (lane >> 2) & 1u, j = lane & 3u;
        const uint32_t bits = (uint32_t)((which ? m_trunc : m_term) >> (j * 16u)) & 0xFFFFu;
        const u4_t packed = {spread4(bits), spread4(bits >> 4), spread4(bits >> 8), spread4(bits >> 12)};
        out_store(reinterpret_cast<u4_t*>(which ? trunc_wave : term_wave) + j, packed);
    } else {
        term_wave[lane] = term ? 1 : 0;
        trunc_wave[lane] = trunc ? 1 : 0;
    }
}

// SPEC.md §10: nearest neighbour among the lanes of this drone's swarm — the
// A = P.agents consecutive lanes starting at lane & ~(A-1). Each lane parks its
// position as one float4 in the wave's (otherwise idle) observation tile and
// reads its neighbours back with one ds_read_b128 each — a third of the LDS
// operations of three __shfl (ds_bpermute) per neighbour, which is what a swarm
// of 64 is bound by (tools/swarm_scan.py: 85 -> 65 us per step at A = 64). The
// tile is private to the wave and a wave's DS operations execute in order, so
// wave-scope fences (compiler ordering only) are all the synchronisation needed.
__device__ __forceinline__ void swarm_neighbour(const KParams& P, const Lane& L, float4* tile, float& nn_d2, float (&nn_e)[3]) {
    const uint32_t lane = threadIdx.x & (kWave - 1), mask = P.agents - 1u, base = lane & ~mask;
    tile[lane] = make_float4(L.s.p[0], L.s.p[1], L.s.p[2], 0.0f);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    nearest_neighbour(P, [&](uint32_t d, float (&e)[3]) {
        const float4 q = tile[base | ((lane + d) & mask)];
        e[0] = q.x - L.s.p[0];
        e[1] = q.y - L.s.p[1];
        e[2] = q.z - L.s.p[2];
    }, nn_d2, nn_e);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the reads are done before the tile is written again
    __builtin_amdgcn_wave_barrier();
}

// steps 1-9 of one env step for any task (the swarm task looks at its neighbours in between).
// CARRY: the state stays in registers from step to step and carries the rotor inputs (Lane::u) with it.
// PK: the RK4 substep in packed f32 instructions (small shards; drone_pk.hpp).
// INRANGE: `act` was drawn by random_action in this kernel (values in [-1, 1): the clamp of SPEC.md section 5 step 1 is the identity).
template <int TASK, bool CARRY = false, bool PK = false, bool INRANGE = false>
__device__ __forceinline__ void step_any(const KParams& P, Lane& L, float4* tile, const float (&act)[4], uint32_t env, uint32_t gstep, StepOut& out) {
    if (TASK == DRONE_TASK_SWARM) {
        StepCtx ctx;
        lane_integrate<TASK, CARRY, PK, INRANGE>(P, L, act, env, gstep, ctx);
        float nn_d2, nn_e[3];
        swarm_neighbour(P, L, tile, nn_d2, nn_e);
        lane_finish<TASK, CARRY>(P, L, env, ctx, nn_d2, out);
    } else {
        lane_step<TASK, CARRY, PK, INRANGE>(P, L, act, env, gstep, out);
    }
}

// the observation row of the (possibly fresh) state
template <int TASK>
__device__ __forceinline__ void obs_any(const KParams& P, const Lane& L, float4* tile, float (&o)[DRONE_OBS_DIM_MAX]) {
    lane_obs(P, L, o);
    if (TASK == DRONE_TASK_RACE) lane_obs_gate(P, L, o);
    if (TASK == DRONE_TASK_SWARM) {
        float nn_d2, nn_e[3];
        swarm_neighbour(P, L, tile, nn_d2, nn_e);  // on the positions after resets
        lane_obs_neighbour(P, L, nn_d2, nn_e, o);
    }
}

// Graph-safe stepping (drone_vec_enable_graph_capture): the vec-level step counter and the step-launch counter live
// in HBM (ctr[0], ctr[1]) instead of in the launch arguments, so that a captured launch replays with advancing
// counters. Every workgroup reads them at its start; the LAST workgroup to finish (arrival counter ctr[2]) advances
// them — by then every other workgroup of this launch has read them — and the kernel boundary publishes the new
// values to the next launch. One extra scalar-memory round trip per wave and one atomic per workgroup: off by default.
struct Counters {
    uint32_t gstep, launches;
};
__device__ __forceinline__ Counters read_counters(const StepArgs& a) {
    Counters c = {a.gstep, a.done_slot};
    if (a.v.ctr) {  // launch-uniform
        c.gstep = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&a.v.ctr[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        c.launches = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&a.v.ctr[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    return c;
}
__device__ __forceinline__ void advance_counters(const StepArgs& a, const Counters& c, uint32_t steps, uint32_t launches) {
    if (a.v.ctr && threadIdx.x == 0) {
        if (atomicAdd(&a.v.ctr[2], 1u) == gridDim.x - 1u) {  // the last workgroup of this launch
            a.v.ctr[2] = 0u;
            a.v.ctr[0] = c.gstep + steps;
            a.v.ctr[1] = c.launches + launches;
        }
    }
}

// Peer-store exchange: the handshake's two publications from inside the launch that writes the outputs (LaunchSig).
// ack: the root's stream has reached this launch, so whatever consumed the previous batch is done (stream order); one
// relaxed system-scope store by the first lane of the grid — it carries no data, so no fence.
// a stream-side wait of this handle has given up (LaunchSig: the stop word): the handshake is broken — the root may still be reading
// the batch this launch would overwrite, or nobody is left to read it. Peer instantiations only; launch-uniform.
__device__ __forceinline__ uint32_t peer_stop_word(const StepArgs& a) {
    return __hip_atomic_load(a.sig.arrive + kPeerStopWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool peer_stopped(const StepArgs& a) { return __builtin_amdgcn_readfirstlane(peer_stop_word(a)) != 0u; }
__device__ __forceinline__ void peer_ack(const StepArgs& a) {
    if (a.sig.ack_flag && blockIdx.x == 0 && threadIdx.x == 0)
        __hip_atomic_store(a.sig.ack_flag, a.sig.ack_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// post / wg_done: called by EVERY thread of EVERY workgroup as the kernel's last statement (it contains a workgroup barrier);
// `chunk`: the 256-drone chunk (env order) this workgroup wrote.
// A flag may only become visible behind all the rows this workgroup (wg_done) or this launch (post) stored — into the root's
// buffers (remote HBM on a multi-GPU node) or into pinned host memory over PCIe: each wave waits for its own stores to be
// acknowledged, the workgroup meets, one lane writes this XCD's L2 back at system scope (a release fence; the asm wait behind
// it keeps the compiler from dropping the fence's own wait — MI355X_MICROARCH.md "Compiler hazard"). Then, wg_done: the chunk's
// word; post: the workgroup counts itself in with a device-scope atomic, and the one whose count comes last knows that every
// other workgroup's release completed before its add, and publishes the flag.
__device__ __forceinline__ void peer_post(const StepArgs& a, uint32_t chunk) {
    if (!a.sig.post_flag && !a.sig.wg_done) return;  // launch-uniform
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (a.sig.wg_done) __hip_atomic_store(a.sig.wg_done + chunk, a.sig.wg_done_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (a.sig.post_flag && atomicAdd(a.sig.arrive, 1u) == gridDim.x - 1u) {
            __hip_atomic_store(a.sig.arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // the kernel boundary publishes it to the next launch
            __hip_atomic_store(a.sig.post_flag, a.sig.post_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// which 256-drone chunk this workgroup owns. `order` (a launch argument, chosen by the host from the step's
// footprint; DeviceView::order) — bit 0: workgroups that share an XCD (blockIdx % 8) take one contiguous eighth of the
// envs instead of being dealt round-robin over one global sweep; bits 2 and 3: non-temporal action / state loads (load_raw);
// bit 1 (odd steps only): sweep in reverse, so the
// lines touched last in one step are the first touched in the next and are still in the Infinity Cache.
// Bijective for any grid size; a speed choice only (profiles/r02_ab/ab_zz_*.txt, ab_order_*.txt).
__device__ __forceinline__ uint32_t my_chunk(uint32_t order, uint32_t gstep, uint32_t nwg) {
    uint32_t c = blockIdx.x;
    if (order & 1u) {
        const uint32_t xcd = blockIdx.x & 7u, q = nwg >> 3, r = nwg & 7u;
        c = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    if ((order & 2u) && (gstep & 1u)) c = nwg - 1u - c;
    return c;
}

// =====================================================================
// per-step kernel (SPEC.md §5): configs 1–4
// =====================================================================
// The leading scalar arguments repeat the words of `a` that the state-load addresses depend on. Built with
// -mllvm -amdgpu-kernarg-preload-count and -DDRONE_EARLY_ARGS=2 (Makefile PRELOAD; not the shipped build) they arrive in SGPRs
// with the wave — no scalar load, no wait — so the state loads go out a scalar-memory round trip earlier; in the shipped
// build (DRONE_EARLY_ARGS=0) they are ordinary, unused kernel arguments (kept: the two builds share one signature).
#define DRONE_STEP_PRE_PARAMS const float4* __restrict__ pre_planes, const float* __restrict__ pre_act, const uint32_t* __restrict__ pre_ctr, uint32_t pre_n, uint32_t pre_n_pad, \
                              uint32_t pre_order, uint32_t pre_nwg, uint32_t pre_gstep, uint32_t pre_slot

// PEER: the handle is in a peer-store exchange with stream-side waits (LaunchSig::peer) — the launch first looks at the stop word.
// A trailing template argument of the kernel itself, so that the PEER = false instantiations are, instruction for instruction,
// the kernels round 5 measured (moving the body into a function shared by two kernels reordered a handful of instructions).
template <int TASK, bool COMPACT, int MEM, bool DT, bool PEER = false>
__global__ __launch_bounds__(kBlock, kStepMinWaves) void drone_step_kernel(DRONE_STEP_PRE_PARAMS, StepArgs a) {
    __shared__ Shared sh;
    const KParams& P = DRONE_PARAMS(sh, a);
#if DRONE_EARLY_ARGS == 2
    a.v.planes = const_cast<float4*>(pre_planes); a.v.act = pre_act; a.v.ctr = const_cast<uint32_t*>(pre_ctr);
    a.v.n = pre_n; a.v.n_pad = pre_n_pad; a.v.order = pre_order; a.nwg = pre_nwg; a.gstep = pre_gstep; a.done_slot = pre_slot;
#endif
    const uint32_t n = a.v.n, np = a.v.stride;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    float4* const tile = sh.obs_tile[threadIdx.x / kWave];
    // One 256-drone chunk per workgroup (two or more, software-pipelined, measured +13 %: profiles/r06_pruned/).
    // Lanes [n, n_pad) exist in the planes and hold a valid reset state: they load and compute like the rest and store nothing
    // into the caller's buffers.
    const Counters ctr = read_counters(a);
    const uint32_t gstep = ctr.gstep, done_slot = ctr.launches & 1u;
    const uint32_t block_base = my_chunk(a.v.order, gstep, a.nwg) * (uint32_t)kBlock;
#if DRONE_STAMPS
    unsigned long long stamp_[kStampSlots];
    stamp_[8] = __builtin_amdgcn_s_memrealtime();
#endif
    DRONE_STAMP(0);  // entry
    RawLane<TASK> cur;
    load_raw<TASK, MEM, DT>(a.v.planes, a.v.act, a.v.n_pad, block_base + threadIdx.x, min(block_base + threadIdx.x, n - 1u), cur);
#if DRONE_EARLY_ARGS == 2
    // everything above came out of preloaded SGPRs: the state loads are in flight before the kernel's first scalar-memory
    // wait. Nothing may be scheduled across this point (a hoisted s_load + s_waitcnt, or a load sunk below one).
    __builtin_amdgcn_sched_barrier(0);
#endif
    if (COMPACT && blockIdx.x == 0 && threadIdx.x == 0) a.v.done_count[done_slot ^ 1u] = 0u;  // arm the next step launch's counter
    if (PEER && peer_stopped(a)) return;  // nothing is stored, nothing published
    peer_ack(a);
    const uint32_t i = block_base + threadIdx.x;
    const bool valid = i < n;

    Lane L;
    float act[4];
    DRONE_STAMP(1);  // loads issued
    unpack_lane<TASK, DT>(P, cur, P.env_offset + i, L, act);
#if DRONE_STAMPS
    asm volatile("" ::"v"(L.s.p[0]), "v"(L.s.r[3]), "v"(L.tgt[0]), "v"(act[0]));  // everything has landed
#endif
    DRONE_STAMP(2);  // data arrived
    StepOut out;
    step_any<TASK>(P, L, tile, act, P.env_offset + i, gstep, out);
#if DRONE_STAMPS
    asm volatile("" ::"v"(L.s.p[0]), "v"(L.s.q[0]), "v"(out.reward));
#endif
    DRONE_STAMP(3);  // integrated, reward, reset
    const bool done = valid && (out.oob || out.trunc);

    // Episode ends (~1 % of the lanes, but some lane in about half of the waves): the per-env log sums are a
    // read-modify-write of two cold planes. Their loads go out FIRST, ahead of this lane's stores, and are
    // consumed at the very end of the chunk: vmcnt retires in issue order, so a load issued behind the
    // state stores could only be waited for together with the write acknowledgements of those (non-temporal,
    // HBM-latency) stores — which stalled the wave, and through the barrier its whole workgroup (−10 % at 2^22 envs).
    // No branch around the stores either (a join would merge the outstanding-access counts of both paths and
    // force a full drain): the padding lanes [n, n_pad) own their plane slots and simply evolve like phantom
    // envs; only the caller's buffers are exactly n long, and there the padding lanes' reward goes to a sink.
    const bool ended = out.oob || out.trunc;  // padding lanes included: their log slots exist too
    const bool log_lane = lane_bit(whole_lines(__ballot(ended), a.v.line_complete));   // this lane's log slots share a line with an ended episode's
    const bool tgt_lane = !DT && lane_bit(whole_lines(__ballot(out.target_changed), a.v.line_complete));
    float4 l0, l1;
    if (log_lane) {
        l0 = a.v.cold[i];
        l1 = a.v.cold[np + i];
    }
    store_lane<TASK>(a.v.planes, a.v.n_pad, i, L, tgt_lane, DT);
    out_store(valid ? &a.v.rew[i] : &a.v.pad_sink[threadIdx.x], out.reward);
    DRONE_STAMP(4);  // state stores issued

    if (COMPACT) {  // done-id list: ballot -> one atomic per wave -> mbcnt rank
        const uint64_t m_done = __ballot(done);
        if (m_done != 0) {  // wave-uniform
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(a.v.done_count + done_slot, (uint32_t)__popcll(m_done));
            base = __shfl(base, 0);
            if (done) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m_done >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_done, 0u));
                a.v.done_ids[base + rank] = i;
            }
        }
    }

    float o[DRONE_OBS_DIM_MAX];
    obs_any<TASK>(P, L, tile, o);
    DRONE_STAMP(5);  // observation math done
    write_outputs<obs_vec<TASK>()>(sh, a.v, a.flags_aligned, o, out.oob, out.trunc, i, block_base);
    DRONE_STAMP(6);  // LDS transpose, barrier, observation / flag stores issued
    if (log_lane) {  // last of all: the two loads have been in flight since before the state stores
        asm volatile("" : "+v"(l0.x), "+v"(l1.x));  // pins the fold down here (the optimiser would hoist it up to the loads and wait there)
        if (ended) fold_log(l0, l1, out);
        a.v.cold[i] = l0;
        a.v.cold[np + i] = l1;
    }
#if DRONE_STAMPS
    DRONE_STAMP(7);  // log fold done
    stamp_[9] = __builtin_amdgcn_s_memrealtime();
    if (a.v.stamps && lane == 0) {
        unsigned long long* row = a.v.stamps + (size_t)(i / kWave) * kStampSlots;
        for (int k = 0; k < kStampSlots; k++) row[k] = stamp_[k];
    }
#endif
    advance_counters(a, ctr, 1u, 1u);
    peer_post(a, block_base / (uint32_t)kBlock);
}

// =====================================================================
// vec_reset (SPEC.md §6). Grid covers n_pad so the padding lanes are valid too.
// =====================================================================
template <int TASK, bool PEER = false>
__global__ __launch_bounds__(kBlock) void drone_reset_kernel(StepArgs a) {
    __shared__ Shared sh;
    const KParams& P = DRONE_PARAMS(sh, a);
    const uint32_t n = a.v.n, np = a.v.stride;
    const uint32_t block_base = blockIdx.x * kBlock;
    const uint32_t i = block_base + threadIdx.x;
    if (PEER && peer_stopped(a)) return;
    peer_ack(a);
    Lane L;
    L.episode = 0u;
    lane_reset<TASK>(P, L, P.env_offset + i);
    store_lane<TASK>(a.v.planes, a.v.n_pad, i, L, true, a.v.derived_target != 0);  // every plane of this task's tile (the wind plane starts at the zeros of the allocation's memset)
    a.v.cold[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    a.v.cold[np + i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n) a.v.rew[i] = 0.0f;
    if (a.v.done_count && i < 2) a.v.done_count[i] = 0u;
    float o[DRONE_OBS_DIM_MAX];
    obs_any<TASK>(P, L, sh.obs_tile[threadIdx.x / kWave], o);
    write_outputs<obs_vec<TASK>()>(sh, a.v, a.flags_aligned, o, false, false, i, block_base);
    peer_post(a, blockIdx.x);
}

// =====================================================================
// fused rollout (SPEC.md §9): config 5. State stays in registers for the
// whole horizon; actions come from the counter RNG; HBM is touched once on
// the way in and once on the way out.
// =====================================================================
template <int TASK, bool PK, bool PEER = false>
__global__ __launch_bounds__(kBlock) void drone_rollout_kernel(StepArgs a, uint32_t horizon) {
    __shared__ Shared sh;
    const KParams& P = DRONE_PARAMS(sh, a);
    const uint32_t n = a.v.n, np = a.v.stride;
    const Counters ctr = read_counters(a);
    const uint32_t gstep0 = ctr.gstep;
    const uint32_t block_base = my_chunk(a.v.order & 1u, 0u, a.nwg) * kBlock;
    const uint32_t i = block_base + threadIdx.x;
    const bool valid = i < n;
    Lane L;
    const bool dt = a.v.derived_target != 0;
    load_lane<TASK>(P, a.v.planes, a.v.n_pad, i, dt, L);
    L.u = rotor_inputs(P, L.s.r);  // carried from here on (step_any<TASK, true>: -18 operations per substep, bit-identical)
    float4 l0 = a.v.cold[i], l1 = a.v.cold[np + i];
    const uint32_t env = P.env_offset + i;
    // the stop word: loaded here, beside the state, and looked at behind the step loop — an early exit up here makes every wave of the
    // launch wait for this load ahead of the loop's first instruction (round 5: +2.5 % at 2^20 envs; behind the loop +0.5 %,
    // profiles/r05_ab/ab_stop_rollout_*.txt — and that only in the peer instantiation)
    uint32_t stop_word = 0u;
    if (PEER) stop_word = peer_stop_word(a);
    peer_ack(a);
    float rsum = 0.0f;
    bool any_term = false, any_trunc = false, any_target = false;
#if DRONE_STAMPS  // diagnostic build: the shader clock this kernel holds = delta s_memtime / delta s_memrealtime x 100 MHz (tools/rollout_clock.py)
    const unsigned long long ck0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    const KParams Pv = P;  // the step loop's own copy of the constants (left from the round-5 vector-register trial: dropping it moves 8-10 instructions of every instantiation; kept so that the measured ISA ships, tests/test_isa_frozen.py)
    for (uint32_t t = 0; t < horizon; t++) {
        float act[4];
        random_action(P.key_action, env, gstep0 + t, act);
        StepOut out;
        step_any<TASK, true, PK, true>(Pv, L, sh.obs_tile[threadIdx.x / kWave], act, env, gstep0 + t, out);
        rsum = rsum + out.reward;
        any_term |= out.oob;
        any_trunc |= out.trunc;
        any_target |= out.target_changed;
        if (out.oob || out.trunc) fold_log(l0, l1, out);
    }
#if DRONE_STAMPS
    asm volatile("" ::"v"(rsum), "v"(L.s.p[0]));
    if (a.v.stamps && (threadIdx.x & (kWave - 1)) == 0) {
        unsigned long long* row = a.v.stamps + (size_t)(i / kWave) * kStampSlots;
        row[0] = ck0; row[1] = __builtin_amdgcn_s_memtime(); row[8] = rt0; row[9] = __builtin_amdgcn_s_memrealtime();
        // where this wave ran (tools/wg_census.py): HW_REG_HW_ID (wave / SIMD / CU / shader array / shader engine) and HW_REG_XCC_ID
        row[2] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);
        row[3] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
#endif
    if (PEER && __builtin_amdgcn_readfirstlane(stop_word) != 0u) return;  // a wait of this handle gave up before this launch: nothing is stored or published
    // padding lanes [n, n_pad) own their plane slots (see the step kernel); rare updates go out as whole lines
    store_lane<TASK>(a.v.planes, a.v.n_pad, i, L, lane_bit(whole_lines(__ballot(any_target), a.v.line_complete)), dt);
    out_store(valid ? &a.v.rew[i] : &a.v.pad_sink[threadIdx.x], rsum);
    if (lane_bit(whole_lines(__ballot(any_term || any_trunc), a.v.line_complete))) {
        a.v.cold[i] = l0;
        a.v.cold[np + i] = l1;
    }
    float o[DRONE_OBS_DIM_MAX];
    obs_any<TASK>(P, L, sh.obs_tile[threadIdx.x / kWave], o);
    write_outputs<obs_vec<TASK>()>(sh, a.v, a.flags_aligned, o, any_term, any_trunc, i, block_base);
    advance_counters(a, ctr, horizon, 0u);  // a rollout builds no done-id list: the step-launch counter stays
    peer_post(a, block_base / (uint32_t)kBlock);
}

// =====================================================================
// K env steps per launch WITH per-step outputs (drone_vec_step_many).
// Between the per-step kernel (state through HBM every step, one dependent
// launch boundary per step) and the fused rollout (outputs only at the
// horizon): state is read once, stays in registers for K steps and is written
// once; every step reads its action row from the caller's [K][N][4] block (or
// draws it from the counter-RNG policy when there is none) and writes its
// observation rows / reward / flag bytes into the caller's [K][N]... blocks,
// exactly what K calls of drone_vec_step would have left there. What it buys:
// at small shards the 2.5 us dependent-launch boundary and the per-launch
// prologue are paid once per K steps instead of once per step (65 536 envs:
// 5.2 us per step -> profiles/r03_step_many_65536/), at large ones the state
// planes leave the per-step byte count (hover: 278 -> 102 + 176 / K bytes).
// The next step's action row is requested before this step's arithmetic, so its
// latency hides behind ~450 VALU instructions; vmcnt retires in order, so the
// row is consumed one iteration later with this iteration's stores still in
// flight (the first iteration is peeled by hand to give the loop entry the same
// outstanding-store count as the back edge — at a join the wait-count pass
// keeps the more conservative of the two).
// =====================================================================
struct ManyArgs {
    const float* act;      // [K][n][4] (act_stride = n), [n][4] applied to all K steps (act_stride = 0: action repeat), or null: the SPEC.md section 2 random policy, drawn in the kernel
    uint32_t act_stride;   // rows between the action blocks of consecutive steps
    float* obs;            // [K][n][obs_dim]
    float* rew;            // [K][n]
    unsigned char* term;   // [K][n]
    unsigned char* trunc;  // [K][n]
    uint32_t* done_ids;    // [K][n] (compact_done) or null
    uint32_t* done_count;  // [K], zeroed by the host before the launch
    uint32_t k_steps;
};

// one env step of the K: everything between "action row in registers" and "outputs of step k issued".
// FULL: every lane of this workgroup is a real env (all but the last workgroup of a ragged shard).
template <int TASK, bool COMPACT, bool POLICY, bool FULL, bool PK>
__device__ __forceinline__ void many_step(const KParams& P, Shared& sh, const StepArgs& a, const ManyArgs& m, Lane& L, float4& l0, float4& l1,
                                          const float4& arow, uint32_t k, uint32_t gstep, uint32_t i, uint32_t block_base, bool& any_target, bool& any_end) {
    const uint32_t n = a.v.n;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const bool valid = FULL || i < n;
    float4* const tile = sh.obs_tile[threadIdx.x / kWave];
    const uint32_t env = P.env_offset + i;
    float act[4];
    if (POLICY) random_action(P.key_action, env, gstep, act);
    else { act[0] = arow.x; act[1] = arow.y; act[2] = arow.z; act[3] = arow.w; }
    StepOut out;
    step_any<TASK, true, PK, POLICY>(P, L, tile, act, env, gstep, out);
    const bool ended = out.oob || out.trunc;
    any_target |= out.target_changed;
    any_end |= ended;
    if (ended) fold_log(l0, l1, out);  // log sums stay in registers; written once at the end of the launch
    const size_t row0 = (size_t)k * n;  // first row of step k in the caller's [K][n] blocks
    out_store(valid ? &m.rew[row0 + i] : &a.v.pad_sink[threadIdx.x], out.reward);
    if (COMPACT) {  // per-step done-id list: ballot -> one atomic per wave -> mbcnt rank
        const uint64_t m_done = __ballot(valid && ended);
        if (m_done != 0) {  // wave-uniform
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(m.done_count + k, (uint32_t)__popcll(m_done));
            base = __shfl(base, 0);
            if (valid && ended) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m_done >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_done, 0u));
                m.done_ids[row0 + base + rank] = i;
            }
        }
    }
    float o[DRONE_OBS_DIM_MAX];
    obs_any<TASK>(P, L, tile, o);
    constexpr int OBSV = obs_vec<TASK>();
    if (FULL) {
        const uint32_t wave_base = i - lane;
        // 16-byte flag pieces need step k's slices 16-byte aligned: block bases aligned and n a multiple of 16 (launch-uniform)
        const bool flags16 = ((reinterpret_cast<uintptr_t>(m.term) | reinterpret_cast<uintptr_t>(m.trunc) | (uintptr_t)n) & 15u) == 0;
        write_outputs_wave<OBSV>(tile, m.obs + (row0 + wave_base) * (size_t)(4 * OBSV), m.term + row0 + wave_base, m.trunc + row0 + wave_base, flags16, o, out.oob, out.trunc);
    } else {
        DeviceView w = a.v;  // step k's slices of the caller's blocks (only the output pointers differ)
        w.obs = m.obs + row0 * (size_t)(4 * OBSV);
        w.term = m.term + row0;
        w.trunc = m.trunc + row0;
        const uint32_t fa = ((reinterpret_cast<uintptr_t>(w.term) & 15u) == 0 ? 1u : 0u) | ((reinterpret_cast<uintptr_t>(w.trunc) & 15u) == 0 ? 2u : 0u);
        write_outputs<OBSV>(sh, w, fa, o, out.oob, out.trunc, i, block_base, k & 1u);
    }
}

template <int TASK, bool COMPACT, bool POLICY, bool FULL, bool PK>
__device__ __forceinline__ void many_loop(const KParams& P, Shared& sh, const StepArgs& a, const ManyArgs& m, Lane& L, float4& l0, float4& l1,
                                          uint32_t gstep0, uint32_t i, uint32_t block_base, bool& any_target, bool& any_end) {
    const uint32_t n = a.v.n, K = m.k_steps;
    // action rows: lane i's row of step k is actp[k * n]; lanes >= n read the last env's row and store nothing
    const float4* actp = POLICY ? nullptr : reinterpret_cast<const float4*>(m.act) + (FULL ? i : min(i, n - 1u));
    float4 a_cur = make_float4(0.f, 0.f, 0.f, 0.f), a_nxt = a_cur;
    if (!POLICY) {
        a_cur = actp[0];
        a_nxt = actp[(size_t)min(1u, K - 1u) * m.act_stride];
    }
    // step 0, peeled: the loop below is entered with step 0's stores behind the load of a_nxt, like every later entry
    many_step<TASK, COMPACT, POLICY, FULL, PK>(P, sh, a, m, L, l0, l1, a_cur, 0u, gstep0, i, block_base, any_target, any_end);
    for (uint32_t k = 1; k < K; k++) {
        a_cur = a_nxt;
        if (!POLICY) a_nxt = actp[(size_t)min(k + 1u, K - 1u) * m.act_stride];  // the NEXT step's row: in flight during this step's arithmetic
        many_step<TASK, COMPACT, POLICY, FULL, PK>(P, sh, a, m, L, l0, l1, a_cur, k, gstep0 + k, i, block_base, any_target, any_end);
    }
}

template <int TASK, bool COMPACT, bool POLICY, bool PK>
__global__ __launch_bounds__(kBlock) void drone_step_many_kernel(StepArgs a, ManyArgs m) {
    __shared__ Shared sh;
    const KParams& P = DRONE_PARAMS(sh, a);
    const uint32_t n = a.v.n, np = a.v.stride;
    const Counters ctr = read_counters(a);
    const uint32_t block_base = my_chunk(a.v.order & 1u, 0u, a.nwg) * kBlock;
    const uint32_t i = block_base + threadIdx.x;
    Lane L;
    const bool dt = a.v.derived_target != 0;
    load_lane<TASK>(P, a.v.planes, a.v.n_pad, i, dt, L);
    L.u = rotor_inputs(P, L.s.r);  // carried through the K steps (step_any<TASK, true>)
    float4 l0 = a.v.cold[i], l1 = a.v.cold[np + i];
    bool any_target = false, any_end = false;
    const KParams Pv = P;  // (as in the fused rollout: the full workgroups' loop sees a copy)
    if (block_base + kBlock <= n) many_loop<TASK, COMPACT, POLICY, true, PK>(Pv, sh, a, m, L, l0, l1, ctr.gstep, i, block_base, any_target, any_end);  // workgroup-uniform
    else many_loop<TASK, COMPACT, POLICY, false, false>(P, sh, a, m, L, l0, l1, ctr.gstep, i, block_base, any_target, any_end);  // the last workgroup of a ragged shard (one workgroup: scalar form, less code)
    // padding lanes [n, n_pad) own their plane slots (see the step kernel); rare updates go out as whole lines
    store_lane<TASK>(a.v.planes, a.v.n_pad, i, L, lane_bit(whole_lines(__ballot(any_target), a.v.line_complete)), dt);
    if (lane_bit(whole_lines(__ballot(any_end), a.v.line_complete))) {
        a.v.cold[i] = l0;
        a.v.cold[np + i] = l1;
    }
    advance_counters(a, ctr, m.k_steps, 0u);  // the per-step done-list ping-pong of drone_vec_step is not touched
}

// =====================================================================
// synthetic random policy into an action buffer (bench / tests)
// =====================================================================
__global__ __launch_bounds__(kBlock) void drone_fill_actions_kernel(StepArgs a, float4* __restrict__ actions) {
#if DRONE_PARAMS_IN_LDS
    __shared__ Shared sh;
#endif
    const KParams& P = DRONE_PARAMS(sh, a);
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.v.n) return;
    float act[4];
    random_action(P.key_action, P.env_offset + i, a.gstep, act);
    actions[i] = make_float4(act[0], act[1], act[2], act[3]);
}

// =====================================================================
// vec_log: sum the per-env log planes (double), clear them.
// Wave reduction by __shfl_down, one partial row per workgroup; the host adds
// the rows in workgroup order, so the result is reproducible run to run.
// =====================================================================
__global__ __launch_bounds__(kBlock) void drone_log_reduce_kernel(float4* __restrict__ cold, uint32_t n, uint32_t np,
                                                                  double* __restrict__ partials) {
    __shared__ double red[kWavesPerBlock][6];
    double s[6] = {0, 0, 0, 0, 0, 0};
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const float4 l0 = cold[i];
        const float4 l1 = cold[np + i];
        s[0] += l0.x; s[1] += l0.y; s[2] += l0.z; s[3] += l0.w; s[4] += l1.x; s[5] += l1.y;
        if (l1.x != 0.0f) {
            cold[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            cold[np + i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    const uint32_t lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
#pragma unroll
    for (int k = 0; k < 6; k++) {
#pragma unroll
        for (int off = kWave / 2; off > 0; off >>= 1) s[k] += __shfl_down(s[k], off);
        if (lane == 0) red[wave][k] = s[k];
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        double t = 0;
#pragma unroll
        for (int w = 0; w < kWavesPerBlock; w++) t += red[w][threadIdx.x];
        partials[blockIdx.x * 6 + threadIdx.x] = t;
    }
}

// =====================================================================
// peer-store exchange: the flag handshake as two one-wave kernels (hipStreamWaitValue32 only accepts signal memory of
// the calling process; these flags live in host memory that several processes share)
// =====================================================================
__global__ __launch_bounds__(64) void drone_flag_post_kernel(uint32_t* flag, uint32_t value) {
    if (threadIdx.x == 0) {
        __threadfence_system();  // the preceding kernels' stores (kernel boundary) and anything of ours: ahead of the flag
        __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// One launch waits for ALL the flags it is given: lane l polls flags l, l + 64, ... (r < count, r != skip); the wave — all
// lanes together, the loop is wave-uniform — retires when every flag has reached `want`, when the time budget has run
// out (lane 0 then raises *err), or AT ONCE when *err is already raised: an earlier wait of this handle gave up (a dead
// peer), and a caller that does not sync every launch has queued every later step's wait behind it — each would
// otherwise spin its full budget in turn. The error word is loaded together with the flags (both live in
// host memory: one PCIe round trip either way).
__global__ __launch_bounds__(64) void drone_flag_wait_kernel(const uint32_t* flags, uint32_t count, uint32_t skip, uint32_t want, uint32_t* err, uint32_t* stop, unsigned long long budget_ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        const uint32_t failed = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        bool ok = true;
        for (uint32_t r = threadIdx.x; r < count; r += 64u)
            if (r != skip) ok = ok && (int32_t)(__hip_atomic_load(flags + r, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - want) >= 0;
        if (__all(ok) || failed != 0u) break;  // wave-uniform (err is one address: every lane read the same word)
        if (__builtin_amdgcn_s_memrealtime() - t0 > budget_ticks) {  // a dead peer: report, do not hang the queue
            if (threadIdx.x == 0) {
                __hip_atomic_store(err, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(stop, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // what the launches queued behind this wait read from HBM (the kernel boundary publishes it)
            }
            break;
        }
        __builtin_amdgcn_s_sleep(64);
    }
}

StepArgs make_args(const DeviceView& v, uint32_t gstep) {
    StepArgs a;
    a.v = v;
    a.gstep = gstep;
    a.done_slot = 0;
    a.nwg = 0;  // set by the launchers that deal chunks (step, rollout, step_many)
    a.sig = LaunchSig{nullptr, nullptr, nullptr, nullptr, 0u, 0u, 0u, 0u};
    a.flags_aligned = ((reinterpret_cast<uintptr_t>(v.term) & 15u) == 0 ? 1u : 0u) | ((reinterpret_cast<uintptr_t>(v.trunc) & 15u) == 0 ? 2u : 0u);
#if !DRONE_PARAMS_IN_LDS
    a.kp = *v.kp_host;
#endif
    return a;
}

inline unsigned grid_for(uint32_t n) { return (n + kBlock - 1) / kBlock; }

// The packed-f32 RK4 form for the register-resident kernels: chosen per handle by the host (DeviceView::packed_rk4)
inline bool use_packed(const DeviceView& v) { return DRONE_PK_RK4 && v.packed_rk4 != 0; }

// hipGetLastError() reports the calling thread's last runtime error, whoever caused it (another library's call that
// failed benignly, a query that returned not-ready). Drop anything stale first so that what a launch_* returns is
// the launch's own status.
inline void drop_stale_error() { (void)hipGetLastError(); }

}  // namespace

hipError_t launch_reset(const DeviceView& v, int task, hipStream_t s, const LaunchSig* sig) {
    drop_stale_error();
    const dim3 g(v.n_pad / kBlock), b(kBlock);
    StepArgs a = make_args(v, 0);
    if (sig) a.sig = *sig;
    if (a.sig.peer) {
        if (task == DRONE_TASK_SWARM) drone_reset_kernel<DRONE_TASK_SWARM, true><<<g, b, 0, s>>>(a);
        else if (task == DRONE_TASK_RACE) drone_reset_kernel<DRONE_TASK_RACE, true><<<g, b, 0, s>>>(a);
        else if (task == DRONE_TASK_WAYPOINT) drone_reset_kernel<DRONE_TASK_WAYPOINT, true><<<g, b, 0, s>>>(a);
        else drone_reset_kernel<DRONE_TASK_HOVER, true><<<g, b, 0, s>>>(a);
        return hipGetLastError();
    }
    if (task == DRONE_TASK_SWARM) drone_reset_kernel<DRONE_TASK_SWARM><<<g, b, 0, s>>>(a);
    else if (task == DRONE_TASK_RACE) drone_reset_kernel<DRONE_TASK_RACE><<<g, b, 0, s>>>(a);
    else if (task == DRONE_TASK_WAYPOINT) drone_reset_kernel<DRONE_TASK_WAYPOINT><<<g, b, 0, s>>>(a);  // its tiles carry the wind plane
    else drone_reset_kernel<DRONE_TASK_HOVER><<<g, b, 0, s>>>(a);
    return hipGetLastError();
}

hipError_t launch_step(const DeviceView& v, int task, uint32_t gstep, uint32_t done_slot, hipStream_t s, const LaunchSig* sig) {
    drop_stale_error();
    StepArgs a = make_args(v, gstep);
    if (sig) a.sig = *sig;
    a.done_slot = done_slot & 1u;
    const dim3 g(grid_for(v.n)), b(kBlock);
    a.nwg = g.x;
    const bool compact = v.done_ids != nullptr;
    const int mem = (int)((v.order >> 2) & 3u);  // bit 0: non-temporal action loads, bit 1: non-temporal state loads (DeviceView::order bits 2, 3)
#define DRONE_PRE_ARGS a.v.planes, a.v.act, a.v.ctr, a.v.n, a.v.n_pad, a.v.order, a.nwg, a.gstep, a.done_slot
    const bool dt = v.derived_target != 0;
    if (a.sig.peer) {  // the peer instantiations (stop word): task x compact x layout, no load hints
#define DRONE_LAUNCH_PEER2(T, D) do { if (compact) drone_step_kernel<T, true, 0, D, true><<<g, b, 0, s>>>(DRONE_PRE_ARGS, a); else drone_step_kernel<T, false, 0, D, true><<<g, b, 0, s>>>(DRONE_PRE_ARGS, a); } while (0)
        if (task == DRONE_TASK_HOVER) { if (dt) DRONE_LAUNCH_PEER2(DRONE_TASK_HOVER, true); else DRONE_LAUNCH_PEER2(DRONE_TASK_HOVER, false); }
        else if (task == DRONE_TASK_SWARM) { if (dt) DRONE_LAUNCH_PEER2(DRONE_TASK_SWARM, true); else DRONE_LAUNCH_PEER2(DRONE_TASK_SWARM, false); }
        else if (task == DRONE_TASK_RACE) DRONE_LAUNCH_PEER2(DRONE_TASK_RACE, false);
        else DRONE_LAUNCH_PEER2(DRONE_TASK_WAYPOINT, false);
#undef DRONE_LAUNCH_PEER2
        return hipGetLastError();
    }
#define DRONE_LAUNCH_STEP3(T, C, D)                                                          \
    do {                                                                                     \
        if (mem == 0) drone_step_kernel<T, C, 0, D><<<g, b, 0, s>>>(DRONE_PRE_ARGS, a);      \
        else if (mem == 1) drone_step_kernel<T, C, 1, D><<<g, b, 0, s>>>(DRONE_PRE_ARGS, a); \
        else if (mem == 2) drone_step_kernel<T, C, 2, D><<<g, b, 0, s>>>(DRONE_PRE_ARGS, a); \
        else drone_step_kernel<T, C, 3, D><<<g, b, 0, s>>>(DRONE_PRE_ARGS, a);               \
    } while (0)
#define DRONE_LAUNCH_STEP2(T, D) do { if (compact) DRONE_LAUNCH_STEP3(T, true, D); else DRONE_LAUNCH_STEP3(T, false, D); } while (0)
#define DRONE_LAUNCH_STEP(T) DRONE_LAUNCH_STEP2(T, false)
    if (task == DRONE_TASK_HOVER) { if (dt) DRONE_LAUNCH_STEP2(DRONE_TASK_HOVER, true); else DRONE_LAUNCH_STEP(DRONE_TASK_HOVER); }
    else if (task == DRONE_TASK_SWARM) { if (dt) DRONE_LAUNCH_STEP2(DRONE_TASK_SWARM, true); else DRONE_LAUNCH_STEP(DRONE_TASK_SWARM); }
    else if (task == DRONE_TASK_RACE) DRONE_LAUNCH_STEP(DRONE_TASK_RACE);
    else DRONE_LAUNCH_STEP(DRONE_TASK_WAYPOINT);
#undef DRONE_LAUNCH_STEP3
#undef DRONE_LAUNCH_STEP2
#undef DRONE_LAUNCH_STEP
#undef DRONE_PRE_ARGS
    return hipGetLastError();
}

hipError_t launch_rollout(const DeviceView& v, int task, uint32_t gstep0, uint32_t horizon, hipStream_t s, const LaunchSig* sig) {
    drop_stale_error();
    StepArgs a = make_args(v, gstep0);
    if (sig) a.sig = *sig;
    const dim3 g(grid_for(v.n)), b(kBlock);
    a.nwg = g.x;
    const bool pk = use_packed(v);
    const bool peer = a.sig.peer != 0;
#define DRONE_LAUNCH_ROLLOUT(T)                                                                                                                              \
    do {                                                                                                                                                     \
        if (peer) { if (pk) drone_rollout_kernel<T, true, true><<<g, b, 0, s>>>(a, horizon); else drone_rollout_kernel<T, false, true><<<g, b, 0, s>>>(a, horizon); } \
        else { if (pk) drone_rollout_kernel<T, true><<<g, b, 0, s>>>(a, horizon); else drone_rollout_kernel<T, false><<<g, b, 0, s>>>(a, horizon); }               \
    } while (0)
    if (task == DRONE_TASK_HOVER) DRONE_LAUNCH_ROLLOUT(DRONE_TASK_HOVER);
    else if (task == DRONE_TASK_SWARM) DRONE_LAUNCH_ROLLOUT(DRONE_TASK_SWARM);
    else if (task == DRONE_TASK_RACE) DRONE_LAUNCH_ROLLOUT(DRONE_TASK_RACE);
    else DRONE_LAUNCH_ROLLOUT(DRONE_TASK_WAYPOINT);
#undef DRONE_LAUNCH_ROLLOUT
    return hipGetLastError();
}

hipError_t launch_step_many(const DeviceView& v, int task, uint32_t gstep0, uint32_t k_steps, const float* act, uint32_t act_stride, float* obs, float* rew,
                            unsigned char* term, unsigned char* trunc, uint32_t* done_ids, uint32_t* done_count, hipStream_t s) {
    drop_stale_error();
    StepArgs a = make_args(v, gstep0);
    a.nwg = grid_for(v.n);
    ManyArgs m;
    m.act = act; m.act_stride = act_stride; m.obs = obs; m.rew = rew; m.term = term; m.trunc = trunc;
    m.done_ids = done_ids; m.done_count = done_count; m.k_steps = k_steps;
    const dim3 g(grid_for(v.n)), b(kBlock);
    const bool compact = done_ids != nullptr, policy = act == nullptr, pk = use_packed(v);
#define DRONE_LAUNCH_MANY2(T, C, PO) do { if (pk) drone_step_many_kernel<T, C, PO, true><<<g, b, 0, s>>>(a, m); else drone_step_many_kernel<T, C, PO, false><<<g, b, 0, s>>>(a, m); } while (0)
#define DRONE_LAUNCH_MANY(T)                                                                 \
    do {                                                                                     \
        if (compact) { if (policy) DRONE_LAUNCH_MANY2(T, true, true); else DRONE_LAUNCH_MANY2(T, true, false); } \
        else { if (policy) DRONE_LAUNCH_MANY2(T, false, true); else DRONE_LAUNCH_MANY2(T, false, false); }    \
    } while (0)
    if (task == DRONE_TASK_HOVER) DRONE_LAUNCH_MANY(DRONE_TASK_HOVER);
    else if (task == DRONE_TASK_SWARM) DRONE_LAUNCH_MANY(DRONE_TASK_SWARM);
    else if (task == DRONE_TASK_RACE) DRONE_LAUNCH_MANY(DRONE_TASK_RACE);
    else DRONE_LAUNCH_MANY(DRONE_TASK_WAYPOINT);
#undef DRONE_LAUNCH_MANY
#undef DRONE_LAUNCH_MANY2
    return hipGetLastError();
}

hipError_t launch_fill_actions(const DeviceView& v, float* actions, uint32_t gstep, hipStream_t s) {
    drop_stale_error();
    drone_fill_actions_kernel<<<dim3(grid_for(v.n)), dim3(kBlock), 0, s>>>(make_args(v, gstep), reinterpret_cast<float4*>(actions));
    return hipGetLastError();
}

hipError_t launch_flag_post(uint32_t* flag, uint32_t value, hipStream_t s) {
    drop_stale_error();
    drone_flag_post_kernel<<<dim3(1), dim3(64), 0, s>>>(flag, value);
    return hipGetLastError();
}

hipError_t launch_flag_wait(const uint32_t* flags, uint32_t count, uint32_t skip, uint32_t want, uint32_t* err, uint32_t* stop, unsigned long long budget_ticks, hipStream_t s) {
    drop_stale_error();
    drone_flag_wait_kernel<<<dim3(1), dim3(64), 0, s>>>(flags, count, skip, want, err, stop, budget_ticks);
    return hipGetLastError();
}

hipError_t launch_log_reduce(const DeviceView& v, double* partials, int max_grid, int* grid_out, hipStream_t s) {
    drop_stale_error();
    int g = (int)grid_for(v.n);
    if (g > max_grid) g = max_grid;
    *grid_out = g;
    drone_log_reduce_kernel<<<dim3(g), dim3(kBlock), 0, s>>>(v.cold, v.n, v.stride, partials);
    return hipGetLastError();
}

}  // namespace drone

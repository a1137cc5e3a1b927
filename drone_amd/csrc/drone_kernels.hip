// drone_kernels.hip — gfx950 (CDNA4, MI355X) kernels of the drone env.
//
// One lane = one drone. State lives in HBM as float4 planes (drone_params.hpp),
// so every state access is a 16-B-per-lane, 1-KiB-per-wave coalesced
// global_load/store_dwordx4. The per-env constants block (KParams, 48 words)
// is staged into LDS once per workgroup and read by broadcast. Observation
// rows ([N][20] AoS, what a vec-env consumer expects) are transposed through a
// wave-private LDS tile so they leave as five fully coalesced 1-KiB stores per
// wave instead of 64 strided 80-B rows. Terminal / truncation bytes are built
// from the wave's __ballot mask (16 lanes store 4 packed bytes each), and the
// optional done-id list is compacted with ballot + mbcnt + one atomic per wave.
//
// The path is elementwise: no MFMA. Roofline = HBM.
// Implements SPEC.md; reference file:line cannot be cited (no source in
// /root/reference — .gitmodules:1-3).
#include <hip/hip_runtime.h>

#include "drone_kernels.h"
#include "drone_lane.hpp"

// ---- tuning knobs (compile-time; defaults are the measured best: DESIGN.md "Tuning log", gpurun_out/ab*.txt) ----
#ifndef DRONE_STEP_MIN_WAVES  // __launch_bounds__ 2nd argument (waves per SIMD) of the per-step kernel; 0 = unset
#define DRONE_STEP_MIN_WAVES 0
#endif
#ifndef DRONE_OBS_VIA_LDS  // 1: transpose observation rows through LDS; 0: strided per-lane row stores
#define DRONE_OBS_VIA_LDS 1
#endif
#ifndef DRONE_NT_STORES  // 1: non-temporal stores for the outputs this path never re-reads (obs, rewards): -5 % at equal placement
#define DRONE_NT_STORES 1
#endif
#ifndef DRONE_NT_STATE  // 1: non-temporal loads and stores for the state planes and actions too
#define DRONE_NT_STATE 0
#endif
#ifndef DRONE_NT_ACT  // 1: non-temporal loads for the action rows (read once, never again)
#define DRONE_NT_ACT 0
#endif
#ifndef DRONE_NT_FLAGS  // 1: non-temporal stores for the packed terminal / truncation dwords
#define DRONE_NT_FLAGS 0
#endif
#ifndef DRONE_STEP_MAX_WAVES  // >0: cap waves per SIMD of the per-step kernel (amdgpu_waves_per_eu)
#define DRONE_STEP_MAX_WAVES 0
#endif
#ifndef DRONE_XCD_REMAP  // 1: workgroups that share an XCD (blockIdx % 8) take one contiguous eighth of the envs (a further -3 % with NT stores)
#define DRONE_XCD_REMAP 1
#endif
#ifndef DRONE_PARAMS_IN_SGPR  // 1: constants from the kernarg segment (scalar loads) instead of the LDS block
#define DRONE_PARAMS_IN_SGPR 0
#endif
#ifndef DRONE_OBS_WAVE_SYNC  // 1: order the wave-private LDS tile with wave-scope fences instead of __syncthreads()
#define DRONE_OBS_WAVE_SYNC 0
#endif
#ifndef DRONE_PERSISTENT_BLOCKS_PER_CU  // >0: cap the grid at this many workgroups per CU and loop over chunks
#define DRONE_PERSISTENT_BLOCKS_PER_CU 0
#endif

namespace drone {

namespace {

constexpr int kWave = 64;
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kObsVec = DRONE_OBS_DIM / 4;  // float4 per obs row = 5

__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }

template <typename T>
__device__ __forceinline__ void out_store(T* p, const T& v) {
#if DRONE_NT_STORES
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
__device__ __forceinline__ void out_store(float4* p, const float4& v) {
#if DRONE_NT_STORES
    typedef float f4_t __attribute__((ext_vector_type(4)));
    f4_t x = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(x, reinterpret_cast<f4_t*>(p));
#else
    *p = v;
#endif
}

typedef float f4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 state_load(const float4* p) {
#if DRONE_NT_STATE
    const f4_t x = __builtin_nontemporal_load(reinterpret_cast<const f4_t*>(p));
    return make_float4(x.x, x.y, x.z, x.w);
#else
    return *p;
#endif
}
__device__ __forceinline__ float4 act_load(const float4* p) {
#if DRONE_NT_ACT
    const f4_t x = __builtin_nontemporal_load(reinterpret_cast<const f4_t*>(p));
    return make_float4(x.x, x.y, x.z, x.w);
#else
    return state_load(p);
#endif
}
__device__ __forceinline__ void state_store(float4* p, const float4& v) {
#if DRONE_NT_STATE
    f4_t x = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(x, reinterpret_cast<f4_t*>(p));
#else
    *p = v;
#endif
}

// ---- constants: HBM -> LDS, once per workgroup ----
__device__ __forceinline__ void stage_params(KParams& sp, const uint32_t* __restrict__ kp) {
    if (threadIdx.x < kParamWords) reinterpret_cast<uint32_t*>(&sp)[threadIdx.x] = kp[threadIdx.x];
    __syncthreads();
}

// ---- plane <-> register marshalling ----
template <int TASK>
__device__ __forceinline__ void load_lane(const float4* __restrict__ pl, uint32_t np, uint32_t i, Lane& L) {
    const float4 a = state_load(&pl[kP0 * np + i]);
    const float4 b = state_load(&pl[kP1 * np + i]);
    const float4 c = state_load(&pl[kP2 * np + i]);
    const float4 d = state_load(&pl[kP3 * np + i]);
    const float4 e = state_load(&pl[kP4 * np + i]);
    const float4 t = state_load(&pl[kPT * np + i]);
    L.s.p[0] = a.x; L.s.p[1] = a.y; L.s.p[2] = a.z; L.s.v[0] = a.w;
    L.s.v[1] = b.x; L.s.v[2] = b.y; L.s.q[0] = b.z; L.s.q[1] = b.w;
    L.s.q[2] = c.x; L.s.q[3] = c.y; L.s.o[0] = c.z; L.s.o[1] = c.w;
    L.s.o[2] = d.x; L.s.r[0] = d.y; L.s.r[1] = d.z; L.s.r[2] = d.w;
    L.s.r[3] = e.x; L.ep_return = e.y; L.tick = f2u(e.z); L.score_count = f2u(e.w);
    L.tgt[0] = t.x; L.tgt[1] = t.y; L.tgt[2] = t.z; L.episode = f2u(t.w);
    if (TASK == DRONE_TASK_WAYPOINT) {
        const float4 w = state_load(&pl[kPW * np + i]);
        L.wind[0] = w.x; L.wind[1] = w.y; L.wind[2] = w.z;
    } else {
        L.wind[0] = L.wind[1] = L.wind[2] = 0.0f;
    }
}

template <int TASK>
__device__ __forceinline__ void store_lane(float4* __restrict__ pl, uint32_t np, uint32_t i, const Lane& L, bool target_changed) {
    state_store(&pl[kP0 * np + i], make_float4(L.s.p[0], L.s.p[1], L.s.p[2], L.s.v[0]));
    state_store(&pl[kP1 * np + i], make_float4(L.s.v[1], L.s.v[2], L.s.q[0], L.s.q[1]));
    state_store(&pl[kP2 * np + i], make_float4(L.s.q[2], L.s.q[3], L.s.o[0], L.s.o[1]));
    state_store(&pl[kP3 * np + i], make_float4(L.s.o[2], L.s.r[0], L.s.r[1], L.s.r[2]));
    state_store(&pl[kP4 * np + i], make_float4(L.s.r[3], L.ep_return, u2f(L.tick), u2f(L.score_count)));
    if (target_changed) state_store(&pl[kPT * np + i], make_float4(L.tgt[0], L.tgt[1], L.tgt[2], u2f(L.episode)));
    if (TASK == DRONE_TASK_WAYPOINT) state_store(&pl[kPW * np + i], make_float4(L.wind[0], L.wind[1], L.wind[2], 0.0f));
}

// ---- wave-cooperative outputs ----

// Observation rows of one wave: registers -> wave-private LDS tile (row-major,
// ds_write_b128, conflict-free at the 80-B row stride) -> read back flat ->
// global_store_dwordx4 at consecutive 16-B slots: 5 × 1 KiB per wave.
__device__ __forceinline__ void store_obs_wave(float* __restrict__ obs, float4* tile, const float (&o)[DRONE_OBS_DIM],
                                               uint32_t wave_base, uint32_t n, uint32_t lane) {
#pragma unroll
    for (int k = 0; k < kObsVec; k++) tile[lane * kObsVec + k] = make_float4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]);
#if DRONE_OBS_WAVE_SYNC
    // the tile belongs to this wave alone and a wave's DS operations execute in
    // order: only the compiler has to be kept from reordering across this point
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#else
    __syncthreads();
#endif
    const uint32_t rows = n - wave_base < (uint32_t)kWave ? n - wave_base : (uint32_t)kWave;  // caller guarantees wave_base < n for active waves
    float4* dst = reinterpret_cast<float4*>(obs + (size_t)wave_base * DRONE_OBS_DIM);
#pragma unroll
    for (int k = 0; k < kObsVec; k++) {
        const uint32_t j = k * kWave + lane;
        if (j < rows * kObsVec) out_store(&dst[j], tile[j]);
    }
}

// 64 one-byte flags of a wave from its ballot mask: lane l < 16 spreads mask
// bits 4l..4l+3 into 4 bytes and stores one dword (needs a 4-B aligned base).
__device__ __forceinline__ void store_flags_wave(unsigned char* __restrict__ dst, uint64_t mask, bool flag, uint32_t wave_base,
                                                 uint32_t i, uint32_t n, uint32_t lane, bool aligned4) {
    if (aligned4 && wave_base + kWave <= n) {
        if (lane < 16) {
            const uint32_t nib = (uint32_t)(mask >> (4u * lane)) & 0xFu;
            const uint32_t packed = (nib * 0x00204081u) & 0x01010101u;
#if DRONE_NT_FLAGS
            __builtin_nontemporal_store(packed, &reinterpret_cast<uint32_t*>(dst + wave_base)[lane]);
#else
            reinterpret_cast<uint32_t*>(dst + wave_base)[lane] = packed;
#endif
        }
    } else if (i < n) {
        dst[i] = flag ? 1 : 0;
    }
}

struct StepArgs {
    DeviceView v;
    uint32_t gstep;
    uint32_t flags_aligned;  // bit0: term 4-B aligned, bit1: trunc 4-B aligned
#if DRONE_PARAMS_IN_SGPR
    KParams kp_val;          // constants by value: read with scalar loads from the kernarg segment
#endif
};

// Observation rows without the LDS transpose: each lane stores its own 80-B row.
[[maybe_unused]] __device__ __forceinline__ void store_obs_direct(float* __restrict__ obs, const float (&o)[DRONE_OBS_DIM], uint32_t i, bool valid) {
    if (!valid) return;
    float4* dst = reinterpret_cast<float4*>(obs + (size_t)i * DRONE_OBS_DIM);
#pragma unroll
    for (int k = 0; k < kObsVec; k++) out_store(&dst[k], make_float4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]));
}

// One 256-drone chunk of the per-step path.
template <int TASK, bool COMPACT>
__device__ __forceinline__ void step_chunk(const StepArgs& a, const KParams& sp, float4 (*obs_tile)[kWave * kObsVec], uint32_t chunk) {
    const uint32_t n = a.v.n, np = a.v.stride;
    const uint32_t i = chunk * kBlock + threadIdx.x;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = threadIdx.x / kWave;
    const uint32_t wave_base = i - lane;
    const bool valid = i < n;
    // planes are padded to n_pad, so loads of the padding lanes are in bounds;
    // they compute on the reset state and store nothing.
    Lane L;
    load_lane<TASK>(a.v.planes, np, i, L);
    float act[4];
    {
        const float4 av = valid ? act_load(&reinterpret_cast<const float4*>(a.v.act)[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
        act[0] = av.x; act[1] = av.y; act[2] = av.z; act[3] = av.w;
    }
    const uint32_t env = sp.env_offset + i;
    StepOut out;
    lane_step<TASK>(sp, L, act, env, a.gstep, out);
    const bool done = valid && (out.oob || out.trunc);

    if (valid) {
        store_lane<TASK>(a.v.planes, np, i, L, out.target_changed);
        out_store(&a.v.rew[i], out.reward);
        if (done) {  // rare: fold the finished episode into this env's log sums
            float4 l0 = a.v.planes[kL0 * np + i];
            float4 l1 = a.v.planes[kL1 * np + i];
            l0.x += out.perf; l0.y += out.score; l0.z += out.ep_return; l0.w += out.ep_len;
            l1.x += 1.0f; l1.y += out.oob ? 1.0f : 0.0f;
            a.v.planes[kL0 * np + i] = l0;
            a.v.planes[kL1 * np + i] = l1;
        }
    }

    // done-mask work on the wave's ballots
    const uint64_t m_term = __ballot(valid && out.oob);
    const uint64_t m_trunc = __ballot(valid && out.trunc);
    store_flags_wave(a.v.term, m_term, out.oob, wave_base, i, n, lane, a.flags_aligned & 1u);
    store_flags_wave(a.v.trunc, m_trunc, out.trunc, wave_base, i, n, lane, a.flags_aligned & 2u);
    if (COMPACT) {
        const uint64_t m_done = m_term | m_trunc;
        uint32_t* cnt = a.v.done_count + (a.gstep & 1u);
        if (chunk == 0 && threadIdx.x == 0) a.v.done_count[(a.gstep + 1u) & 1u] = 0u;  // arm the next step's counter
        if (m_done != 0) {  // wave-uniform
            const uint32_t total = (uint32_t)__popcll(m_done);
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(cnt, total);
            base = __shfl(base, 0);
            if (done) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m_done >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_done, 0u));
                a.v.done_ids[base + rank] = i;
            }
        }
    }

    float o[DRONE_OBS_DIM];
    lane_obs(sp, L, o);
#if DRONE_OBS_VIA_LDS
    store_obs_wave(a.v.obs, obs_tile[wave], o, wave_base < n ? wave_base : 0u, wave_base < n ? n : 0u, lane);
#else
    store_obs_direct(a.v.obs, o, i, valid);
#endif
}

// =====================================================================
// per-step kernel (SPEC.md §5): configs 1–4
// =====================================================================
#if DRONE_STEP_MIN_WAVES > 0
#define DRONE_STEP_BOUNDS __launch_bounds__(kBlock, DRONE_STEP_MIN_WAVES)
#else
#define DRONE_STEP_BOUNDS __launch_bounds__(kBlock)
#endif
#if DRONE_STEP_MAX_WAVES > 0
#define DRONE_STEP_WAVES __attribute__((amdgpu_waves_per_eu(1, DRONE_STEP_MAX_WAVES)))
#else
#define DRONE_STEP_WAVES
#endif

template <int TASK, bool COMPACT>
__global__ DRONE_STEP_BOUNDS DRONE_STEP_WAVES void drone_step_kernel(StepArgs a) {
#if DRONE_OBS_VIA_LDS
    __shared__ float4 obs_tile[kWavesPerBlock][kWave * kObsVec];
#else
    float4 (*obs_tile)[kWave * kObsVec] = nullptr;
#endif
#if DRONE_PARAMS_IN_SGPR
    const KParams& sp = a.kp_val;
#else
    __shared__ KParams sp;
    stage_params(sp, a.v.kp);
#endif
#if DRONE_PERSISTENT_BLOCKS_PER_CU > 0
    const uint32_t chunks = (a.v.n + kBlock - 1) / kBlock;
    for (uint32_t chunk = blockIdx.x; chunk < chunks; chunk += gridDim.x) {
        step_chunk<TASK, COMPACT>(a, sp, obs_tile, chunk);
#if DRONE_OBS_VIA_LDS
        __syncthreads();  // the tile is reused by the next chunk
#endif
    }
#elif DRONE_XCD_REMAP
    {   // bijective for any grid size (cdna_hip_programming.md §5 "XCD swizzle must be bijective")
        const uint32_t nwg = gridDim.x, xcd = blockIdx.x & 7u, q = nwg >> 3, r = nwg & 7u;
        const uint32_t chunk = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + (blockIdx.x >> 3);
        step_chunk<TASK, COMPACT>(a, sp, obs_tile, chunk);
    }
#else
    step_chunk<TASK, COMPACT>(a, sp, obs_tile, blockIdx.x);
#endif
}

// =====================================================================
// vec_reset (SPEC.md §6)
// =====================================================================
__global__ __launch_bounds__(kBlock) void drone_reset_kernel(StepArgs a) {
    __shared__ KParams sp;
    __shared__ float4 obs_tile[kWavesPerBlock][kWave * kObsVec];
    stage_params(sp, a.v.kp);
    const uint32_t n = a.v.n, np = a.v.stride;
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = threadIdx.x / kWave;
    const uint32_t wave_base = i - lane;
    Lane L;
    L.episode = 0u;
    lane_reset(sp, L, sp.env_offset + i);
    // every plane, padding lanes included (keeps padding finite)
    store_lane<DRONE_TASK_WAYPOINT>(a.v.planes, np, i, L, true);
    a.v.planes[kL0 * np + i] = make_float4(0.f, 0.f, 0.f, 0.f);
    a.v.planes[kL1 * np + i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n) {
        a.v.rew[i] = 0.0f;
        a.v.term[i] = 0;
        a.v.trunc[i] = 0;
    }
    if (a.v.done_count && i < 2) a.v.done_count[i] = 0u;
    float o[DRONE_OBS_DIM];
    lane_obs(sp, L, o);
    store_obs_wave(a.v.obs, obs_tile[wave], o, wave_base < n ? wave_base : 0u, wave_base < n ? n : 0u, lane);
}

// =====================================================================
// fused rollout (SPEC.md §9): config 5. State stays in registers for the
// whole horizon; actions come from the counter RNG; HBM is touched once on
// the way in and once on the way out.
// =====================================================================
template <int TASK>
__global__ __launch_bounds__(kBlock) void drone_rollout_kernel(StepArgs a, uint32_t horizon) {
    __shared__ KParams sp;
    __shared__ float4 obs_tile[kWavesPerBlock][kWave * kObsVec];
    stage_params(sp, a.v.kp);
    const uint32_t n = a.v.n, np = a.v.stride;
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = threadIdx.x / kWave;
    const uint32_t wave_base = i - lane;
    const bool valid = i < n;
    Lane L;
    load_lane<TASK>(a.v.planes, np, i, L);
    float4 l0 = a.v.planes[kL0 * np + i];
    float4 l1 = a.v.planes[kL1 * np + i];
    const uint32_t env = sp.env_offset + i;
    float rsum = 0.0f;
    bool any_term = false, any_trunc = false, any_target = false;
    for (uint32_t t = 0; t < horizon; t++) {
        float act[4];
        random_action(sp.key_action, env, a.gstep + t, act);
        StepOut out;
        lane_step<TASK>(sp, L, act, env, a.gstep + t, out);
        rsum = rsum + out.reward;
        any_term |= out.oob;
        any_trunc |= out.trunc;
        any_target |= out.target_changed;
        if (out.oob || out.trunc) {
            l0.x += out.perf; l0.y += out.score; l0.z += out.ep_return; l0.w += out.ep_len;
            l1.x += 1.0f; l1.y += out.oob ? 1.0f : 0.0f;
        }
    }
    if (valid) {
        store_lane<TASK>(a.v.planes, np, i, L, any_target);
        a.v.rew[i] = rsum;
        if (any_term || any_trunc) {
            a.v.planes[kL0 * np + i] = l0;
            a.v.planes[kL1 * np + i] = l1;
        }
    }
    const uint64_t m_term = __ballot(valid && any_term);
    const uint64_t m_trunc = __ballot(valid && any_trunc);
    store_flags_wave(a.v.term, m_term, any_term, wave_base, i, n, lane, a.flags_aligned & 1u);
    store_flags_wave(a.v.trunc, m_trunc, any_trunc, wave_base, i, n, lane, a.flags_aligned & 2u);
    float o[DRONE_OBS_DIM];
    lane_obs(sp, L, o);
    store_obs_wave(a.v.obs, obs_tile[wave], o, wave_base < n ? wave_base : 0u, wave_base < n ? n : 0u, lane);
}

// =====================================================================
// synthetic random policy into an action buffer (bench / tests)
// =====================================================================
__global__ __launch_bounds__(kBlock) void drone_fill_actions_kernel(const uint32_t* __restrict__ kp, float4* __restrict__ actions,
                                                                    uint32_t n, uint32_t gstep) {
    __shared__ KParams sp;
    stage_params(sp, kp);
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float a[4];
    random_action(sp.key_action, sp.env_offset + i, gstep, a);
    actions[i] = make_float4(a[0], a[1], a[2], a[3]);
}

// =====================================================================
// vec_log: sum the per-env log planes (double), clear them.
// Wave reduction by __shfl_down, one partial row per workgroup; the host adds
// the rows in workgroup order, so the result is reproducible run to run.
// =====================================================================
__global__ __launch_bounds__(kBlock) void drone_log_reduce_kernel(float4* __restrict__ planes, uint32_t n, uint32_t np,
                                                                  double* __restrict__ partials) {
    __shared__ double red[kWavesPerBlock][6];
    double s[6] = {0, 0, 0, 0, 0, 0};
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const float4 l0 = planes[kL0 * np + i];
        const float4 l1 = planes[kL1 * np + i];
        s[0] += l0.x; s[1] += l0.y; s[2] += l0.z; s[3] += l0.w; s[4] += l1.x; s[5] += l1.y;
        if (l1.x != 0.0f) {
            planes[kL0 * np + i] = make_float4(0.f, 0.f, 0.f, 0.f);
            planes[kL1 * np + i] = make_float4(0.f, 0.f, 0.f, 0.f);
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

StepArgs make_args(const DeviceView& v, uint32_t gstep) {
    StepArgs a;
    a.v = v;
    a.gstep = gstep;
#if DRONE_PARAMS_IN_SGPR
    a.kp_val = *v.kp_host;
#endif
    a.flags_aligned = ((reinterpret_cast<uintptr_t>(v.term) & 3u) == 0 ? 1u : 0u) | ((reinterpret_cast<uintptr_t>(v.trunc) & 3u) == 0 ? 2u : 0u);
    return a;
}

inline unsigned grid_for(uint32_t n) { return (n + kBlock - 1) / kBlock; }

}  // namespace

hipError_t launch_reset(const DeviceView& v, hipStream_t s) {
    // covers the padding lanes too: grid over n_pad
    drone_reset_kernel<<<dim3(v.n_pad / kBlock), dim3(kBlock), 0, s>>>(make_args(v, 0));
    return hipGetLastError();
}

hipError_t launch_step(const DeviceView& v, int task, uint32_t gstep, hipStream_t s) {
    const StepArgs a = make_args(v, gstep);
    unsigned blocks = grid_for(v.n);
#if DRONE_PERSISTENT_BLOCKS_PER_CU > 0
    if (blocks > 256u * DRONE_PERSISTENT_BLOCKS_PER_CU) blocks = 256u * DRONE_PERSISTENT_BLOCKS_PER_CU;
#endif
    const dim3 g(blocks), b(kBlock);
    const bool compact = v.done_ids != nullptr;
    if (task == DRONE_TASK_HOVER) {
        if (compact) drone_step_kernel<DRONE_TASK_HOVER, true><<<g, b, 0, s>>>(a);
        else drone_step_kernel<DRONE_TASK_HOVER, false><<<g, b, 0, s>>>(a);
    } else {
        if (compact) drone_step_kernel<DRONE_TASK_WAYPOINT, true><<<g, b, 0, s>>>(a);
        else drone_step_kernel<DRONE_TASK_WAYPOINT, false><<<g, b, 0, s>>>(a);
    }
    return hipGetLastError();
}

hipError_t launch_rollout(const DeviceView& v, int task, uint32_t gstep0, uint32_t horizon, hipStream_t s) {
    const StepArgs a = make_args(v, gstep0);
    const dim3 g(grid_for(v.n)), b(kBlock);
    if (task == DRONE_TASK_HOVER) drone_rollout_kernel<DRONE_TASK_HOVER><<<g, b, 0, s>>>(a, horizon);
    else drone_rollout_kernel<DRONE_TASK_WAYPOINT><<<g, b, 0, s>>>(a, horizon);
    return hipGetLastError();
}

hipError_t launch_fill_actions(const DeviceView& v, float* actions, uint32_t gstep, hipStream_t s) {
    drone_fill_actions_kernel<<<dim3(grid_for(v.n)), dim3(kBlock), 0, s>>>(v.kp, reinterpret_cast<float4*>(actions), v.n, gstep);
    return hipGetLastError();
}

hipError_t launch_log_reduce(const DeviceView& v, double* partials, int max_grid, int* grid_out, hipStream_t s) {
    int g = (int)grid_for(v.n);
    if (g > max_grid) g = max_grid;
    *grid_out = g;
    drone_log_reduce_kernel<<<dim3(g), dim3(kBlock), 0, s>>>(v.planes, v.n, v.stride, partials);
    return hipGetLastError();
}

}  // namespace drone

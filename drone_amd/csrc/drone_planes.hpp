// drone_planes.hpp — how one lane's state travels between the float4 planes in HBM (drone_params.hpp: interleaved per
// 64-drone tile) and registers, and the streaming stores of the output path. Part of drone_kernels.hip (included inside its
// drone::{anonymous} namespace, device code only); split out in round 6 to keep the kernels file readable.
// Included exactly once, by drone_kernels.hip, behind drone_lane.hpp and drone_kernels.h.
#pragma once

constexpr int kWave = 64;
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kObsVecMax = DRONE_OBS_DIM_MAX / 4;  // float4 per observation row: 5 (tasks 0, 1) or 6 (swarm)
template <int TASK> constexpr int obs_vec() { return (TASK == DRONE_TASK_SWARM || TASK == DRONE_TASK_RACE) ? 6 : 5; }
template <int TASK> constexpr bool has_aux_plane() { return TASK == DRONE_TASK_WAYPOINT || TASK == DRONE_TASK_RACE; }  // wind / gate normal
constexpr int kFlagLanes = kBlock / 16;        // lanes that write one flag array of a workgroup, 16 B each
static_assert(kBlock % kWave == 0 && 2 * kFlagLanes <= kWave, "workgroup must be 64..512 threads");

typedef float f4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }

// stores of data this path never reads back: non-temporal (plain stores: +21 % at 2^20 envs, +11 % at 131 072, 0 at 2^22; profiles/r01_ab/ab_nt_*)
__device__ __forceinline__ void out_store(float* p, float v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void out_store(float4* p, const float4& v) {
    const f4_t x = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(x, reinterpret_cast<f4_t*>(p));
}
__device__ __forceinline__ void out_store(u4_t* p, const u4_t& v) { __builtin_nontemporal_store(v, p); }

// the state planes too: the next step's loads come from HBM or the Infinity Cache either way (plain: +0.4 % at 2^20 envs, +4 % at 131 072)
__device__ __forceinline__ void state_store(float4* p, const float4& v) {
    const f4_t x = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(x, reinterpret_cast<f4_t*>(p));
}

// A scattered 16-byte store into HBM is a partial-line write: the memory controller turns it into a read-modify-write
// that occupies its channel for tens of nanoseconds (measured: one such store per ended episode, 0.8 % of the lanes, cost 5 % of the
// step kernel at 2^22 envs; profiles/r02_ab/ab_ends_*.txt). So rare per-lane plane updates are widened to whole
// 128-byte lines: if any of the 8 lanes that share a line needs the update, all 8 store (the others rewrite what they
// hold). `m` is a ballot mask; the result has every aligned group of 8 bits set in which `m` had a bit. While the
// step's working set fits the 256 MiB Infinity Cache the partial writes are absorbed there and the widening only adds
// bytes (+0.6 % at 2^20 envs), so the host enables it per handle by footprint (DeviceView::line_complete).
__device__ __forceinline__ uint64_t whole_lines(uint64_t m, uint32_t enabled) {
    if (!enabled) return m;  // wave-uniform (a launch argument)
    m |= m >> 1;
    m |= m >> 2;
    m |= m >> 4;
    return (m & 0x0101010101010101ull) * 0xFFull;
}
__device__ __forceinline__ bool lane_bit(uint64_t m) { return (m >> (threadIdx.x & (kWave - 1))) & 1ull; }

// ---- plane <-> register marshalling ----
// one lane's state as it sits in HBM: issued as a block of loads, unpacked when first needed
template <int TASK>
struct RawLane {
    float4 a, b, c, d, e, t, w, act;
};

// No branches in here (the wait-count pass merges the outstanding-load state of all paths into a join). `ia` is the action
// row to read, already clamped into the buffer by the caller (lanes >= n read the last row and never store anything).
// DT: the derived-target layout (drone_params.hpp): five planes per tile, no target plane to read.
// MEM: which of the loads carry the non-temporal hint — bit 0 the action rows, bit 1 the state planes. A compile-time choice
// (a run-time branch here would make the wait-count pass drain all loads at the join), made by the host per handle from the
// bytes one step touches (DeviceView::order bits 2 and 3): a hint is worth something only where the line would not have
// been served from a cache anyway, and costs dearly where it would (state planes: +21 % at 2^20 envs, -10 % at 2^21;
// action rows: +19 % at 2^20, -2 % at 2^23 — profiles/r04_ab/band_*.txt, r02_ab/ab_o6_*.txt).
template <int TASK, int MEM, bool DT>
__device__ __forceinline__ void load_raw(const float4* __restrict__ pl, const float* __restrict__ actions, uint32_t np, uint32_t i, uint32_t ia, RawLane<TASK>& R) {
    constexpr bool NT_STATE = (MEM & 2) != 0, NT_ACT = (MEM & 1) != 0;
    auto plane = [&](int p) {
        const uint32_t e = hot_index(hot_planes(TASK, DT), p, i, np);
        if (NT_STATE) {
            const f4_t x = __builtin_nontemporal_load(reinterpret_cast<const f4_t*>(pl) + e);
            return make_float4(x.x, x.y, x.z, x.w);
        }
        return pl[e];
    };
    R.a = plane(kP0);
    R.b = plane(kP1);
    R.c = plane(kP2);
    R.d = plane(kP3);
    R.e = plane(kP4);
    if (!DT) R.t = plane(kPT);
    if (has_aux_plane<TASK>()) R.w = plane(kPW);
    if (NT_ACT) {
        const f4_t av = __builtin_nontemporal_load(reinterpret_cast<const f4_t*>(actions) + ia);
        R.act = make_float4(av.x, av.y, av.z, av.w);
    } else {
        R.act = reinterpret_cast<const float4*>(actions)[ia];
    }
}

// the words of P4 and PT that are not floats: (tick, score_count, episode) and the target, for either layout
__device__ __forceinline__ void unpack_counters(const KParams& P, bool dt, const float4& e, const float4& t, uint32_t env, Lane& L) {
    if (dt) {
        const uint32_t ts = f2u(e.z);
        L.tick = ts & 0xFFFFu;
        L.score_count = ts >> 16;
        L.episode = f2u(e.w);
        derive_target(P, env, L.episode, L.tgt);
    } else {
        L.tick = f2u(e.z);
        L.score_count = f2u(e.w);
        L.tgt[0] = t.x; L.tgt[1] = t.y; L.tgt[2] = t.z;
        L.episode = f2u(t.w);
    }
}

template <int TASK, bool DT>
__device__ __forceinline__ void unpack_lane(const KParams& P, const RawLane<TASK>& R, uint32_t env, Lane& L, float (&act)[4]) {
    const float4 &a = R.a, &b = R.b, &c = R.c, &d = R.d, &e = R.e;
    L.s.p[0] = a.x; L.s.p[1] = a.y; L.s.p[2] = a.z; L.s.v[0] = a.w;
    L.s.v[1] = b.x; L.s.v[2] = b.y; L.s.q[0] = b.z; L.s.q[1] = b.w;
    L.s.q[2] = c.x; L.s.q[3] = c.y; L.s.o[0] = c.z; L.s.o[1] = c.w;
    L.s.o[2] = d.x; L.s.r[0] = d.y; L.s.r[1] = d.z; L.s.r[2] = d.w;
    L.s.r[3] = e.x; L.ep_return = e.y;
    unpack_counters(P, DT, e, R.t, env, L);
    if (has_aux_plane<TASK>()) {
        L.wind[0] = R.w.x; L.wind[1] = R.w.y; L.wind[2] = R.w.z;
    } else {
        L.wind[0] = L.wind[1] = L.wind[2] = 0.0f;
    }
    act[0] = R.act.x; act[1] = R.act.y; act[2] = R.act.z; act[3] = R.act.w;
}

// `dt` is launch-uniform here (the register-resident kernels load and store the state once per launch: a branch costs nothing)
template <int TASK>
__device__ __forceinline__ void load_lane(const KParams& P, const float4* __restrict__ pl, uint32_t np, uint32_t i, bool dt, Lane& L) {
    const uint32_t nph = hot_planes(TASK, dt);
    const float4 a = pl[hot_index(nph, kP0, i, np)];
    const float4 b = pl[hot_index(nph, kP1, i, np)];
    const float4 c = pl[hot_index(nph, kP2, i, np)];
    const float4 d = pl[hot_index(nph, kP3, i, np)];
    const float4 e = pl[hot_index(nph, kP4, i, np)];
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!dt) t = pl[hot_index(nph, kPT, i, np)];
    L.s.p[0] = a.x; L.s.p[1] = a.y; L.s.p[2] = a.z; L.s.v[0] = a.w;
    L.s.v[1] = b.x; L.s.v[2] = b.y; L.s.q[0] = b.z; L.s.q[1] = b.w;
    L.s.q[2] = c.x; L.s.q[3] = c.y; L.s.o[0] = c.z; L.s.o[1] = c.w;
    L.s.o[2] = d.x; L.s.r[0] = d.y; L.s.r[1] = d.z; L.s.r[2] = d.w;
    L.s.r[3] = e.x; L.ep_return = e.y;
    unpack_counters(P, dt, e, t, P.env_offset + i, L);
    if (has_aux_plane<TASK>()) {
        const float4 w = pl[hot_index(nph, kPW, i, np)];
        L.wind[0] = w.x; L.wind[1] = w.y; L.wind[2] = w.z;
    } else {
        L.wind[0] = L.wind[1] = L.wind[2] = 0.0f;
    }
}

// `dt`: a template constant in the per-step kernel (no branch may surround its stores), launch-uniform elsewhere
template <int TASK>
__device__ __forceinline__ void store_lane(float4* __restrict__ pl, uint32_t np, uint32_t i, const Lane& L, bool target_changed, bool dt) {
    const uint32_t nph = hot_planes(TASK, dt);
    state_store(&pl[hot_index(nph, kP0, i, np)], make_float4(L.s.p[0], L.s.p[1], L.s.p[2], L.s.v[0]));
    state_store(&pl[hot_index(nph, kP1, i, np)], make_float4(L.s.v[1], L.s.v[2], L.s.q[0], L.s.q[1]));
    state_store(&pl[hot_index(nph, kP2, i, np)], make_float4(L.s.q[2], L.s.q[3], L.s.o[0], L.s.o[1]));
    state_store(&pl[hot_index(nph, kP3, i, np)], make_float4(L.s.o[2], L.s.r[0], L.s.r[1], L.s.r[2]));
    if (dt) {  // the episode counter lives here; tick and score_count share a word (both <= horizon <= 65 535); no target plane
        state_store(&pl[hot_index(nph, kP4, i, np)], make_float4(L.s.r[3], L.ep_return, u2f(L.tick | (L.score_count << 16)), u2f(L.episode)));
    } else {
        state_store(&pl[hot_index(nph, kP4, i, np)], make_float4(L.s.r[3], L.ep_return, u2f(L.tick), u2f(L.score_count)));
        if (target_changed) pl[hot_index(nph, kPT, i, np)] = make_float4(L.tgt[0], L.tgt[1], L.tgt[2], u2f(L.episode));
    }
    // wind changes every step; a gate normal only together with its centre
    if (TASK == DRONE_TASK_WAYPOINT || (TASK == DRONE_TASK_RACE && target_changed)) pl[hot_index(nph, kPW, i, np)] = make_float4(L.wind[0], L.wind[1], L.wind[2], 0.0f);
}

// per-env log sums: touched only when an episode ended
__device__ __forceinline__ void fold_log(float4& l0, float4& l1, const StepOut& out) {
    l0.x += out.perf; l0.y += out.score; l0.z += out.ep_return; l0.w += out.ep_len;
    l1.x += 1.0f; l1.y += out.oob ? 1.0f : 0.0f;
}

// drone_params.hpp — the constants block every kernel receives (KParams), the
// counter RNG, and the plane layout of the device state.
//
// Follows SPEC.md §1–§3 (this repo's spec; the reference has no source to
// cite — /root/reference/.gitmodules:1-3). Written
// independently of oracle/: nothing here includes or links the oracle.
#pragma once

#include <math.h>
#include <stdint.h>

#include "../../include/drone_vec.h"

#if defined(__HIPCC__)
#define DRONE_FN __device__ __host__ __forceinline__
#else
#define DRONE_FN static inline __attribute__((always_inline))
#endif

namespace drone {

enum Stream : uint32_t { kReset = 0, kAction = 1, kWind = 2, kWaypoint = 3 };

// ---- device state (DESIGN.md "Data layout") ----
// One lane reads/writes one float4 per plane: 16 B/lane, 1 KiB per wave-instruction, fully coalesced.
// The planes a step touches every time (P0..PT, + PW for tasks 1 and 3) are interleaved per 64-drone wave tile:
//   hot[tile = i / 64][plane][lane = i % 64]      (float4 elements)
// so a wave streams ONE contiguous 6-7 KiB piece instead of 6-7 separate 1-KiB pieces 2^k bytes apart (+3 % at 2^22
// envs where nothing is cached, profiles/r02_baseline/stream_mix.txt; and one base address + immediate offsets
// instead of seven 64-bit address computations). The two log planes are cold (touched when an episode ends) and stay
// plain planes: cold[k][stride].
enum Plane : int {
    kP0 = 0,    // pos.x pos.y pos.z vel.x
    kP1 = 1,    // vel.y vel.z q.w q.x
    kP2 = 2,    // q.y q.z om.x om.y
    kP3 = 3,    // om.z rpm0 rpm1 rpm2
    kP4 = 4,    // rpm3 ep_return tick(u32) score_count(u32)
    kPT = 5,    // target.x target.y target.z episode(u32)   — written only when it changes; ABSENT in the derived-target layout
    kPW = 6,    // aux: wind.xyz (task 1) / gate normal (task 3)  — tasks 1 and 3 only
    kL0 = 7,    // perf_sum score_sum ret_sum len_sum        — touched only when an episode ends
    kL1 = 8,    // n_sum oob_sum (pad) (pad)
    kNumPlanes = 9
};

// Everything a lane needs that is the same for all lanes: 57 words. Reaches
// the lanes through the kernarg segment (scalar loads) or, with
// -DDRONE_PARAMS_IN_LDS=1, staged HBM -> LDS once per workgroup and read by
// broadcast ds_read (drone_kernels.hip).
struct KParams {
    // integrator / dynamics (premultiplied as SPEC.md §1 defines them)
    float h, h_half, h_sixth;
    float hq, hq_half, hq_sixth;  // quaternion rows integrate 2*qdot with halved steps
    float kT2_m, cx, cy, cz;
    float gxi, gyi, gzi;
    float kdx, kdy, kdz;
    float drag_m, e_half, e_full, gravity;  // e_*: rotor lag over half a substep / a substep, exp(-h / (2 tau)), exp(-h / tau)
    float half_max_rpm, hover_rpm, max_rpm, max_vel, max_omega;
    // observation scales
    float inv_max_vel, inv_max_omega, inv_max_rpm, inv_bound, half_inv_bound;
    // task
    float bound, spawn_extent, target_extent, tilt_init;
    float hover_radius, waypoint_radius;
    float wind_decay, wind_gain, wind_max;
    float c_omega, c_action, crash_penalty, progress_scale, waypoint_bonus;
    float coll_r2, inv_prox_r2, nn_far2, c_proximity;  // task 2 (SPEC.md §10)
    float gate_r2;                                      // task 3 (SPEC.md §11)
    // integers
    uint32_t horizon, substeps;
    uint32_t key_reset, key_action, key_wind, key_waypoint;
    uint32_t env_offset;
    uint32_t agents;  // drones per swarm (task 2), else 1
};
constexpr uint32_t kTile = 64;  // drones per state tile = one wavefront
// planes per hot tile: the aux plane (wind / gate normal) exists only for the tasks that use it
// Derived-target layout (`dt`; hover and swarm tasks only): their target is a pure function of (reset key, env, episode) —
// SPEC.md section 6 draws it at reset and nothing else writes it — so the target plane need not exist: the episode
// counter moves into P4 (tick and score_count share one word, 16 bits each: both are <= horizon <= 65 535) and the
// kernels re-derive the target from three hashes instead of reading 16 bytes per env and step (278 -> 262 B for hover).
// Chosen per handle by the host where the step is HBM-bound (DeviceView::derived_target).
DRONE_FN constexpr uint32_t hot_planes(int task, bool dt = false) { return (task == DRONE_TASK_WAYPOINT || task == DRONE_TASK_RACE) ? 7u : dt ? 5u : 6u; }
DRONE_FN constexpr bool task_has_derived_target(int task) { return task == DRONE_TASK_HOVER || task == DRONE_TASK_SWARM; }
#ifndef DRONE_TILED_STATE  // 0: plain planes [plane][n_pad] (the round-1 layout; kept for A/B at equal placement)
#define DRONE_TILED_STATE 1
#endif
// float4 element of plane `p` of drone `i` in the hot region (`n_pad`: drones the region is laid out for)
DRONE_FN constexpr uint32_t hot_index(uint32_t planes_per_tile, uint32_t p, uint32_t i, uint32_t n_pad) {
#if DRONE_TILED_STATE
    return ((i / kTile) * planes_per_tile + p) * kTile + (i % kTile);
#else
    return p * n_pad + i;
#endif
}

static_assert(sizeof(KParams) == 57 * 4, "KParams is passed / staged as 57 words");
constexpr int kParamWords = 57;

// ---- SPEC.md §2: counter RNG ----
DRONE_FN uint32_t hash32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
DRONE_FN uint32_t rng_base(uint32_t key, uint32_t env, uint32_t ctr) {
    return hash32(hash32(key ^ env) + ctr * 0x9E3779B9u);
}
DRONE_FN uint32_t rng_draw(uint32_t base, uint32_t d) { return hash32(base + d * 0x85EBCA6Bu); }
// SPEC.md §2 states sym(u) = 2 (u >> 8) 2^-24 - 1 and s16(h) = (h - 32768) 2^-15. Both are computed here as ONE fused
// multiply-add on the converted integer: every intermediate of either form is exact (24- / 16-bit integers, powers of
// two), so the two formulations round the same real number once and agree bit for bit (the oracle keeps the SPEC's
// literal form: tests/test_lane_host.py compares them over all 2^16 halves); one operation fewer per value.
DRONE_FN float sym(uint32_t u) { return __builtin_fmaf((float)(u >> 8), 1.1920928955078125e-7f, -1.0f); }
DRONE_FN float s16(uint32_t h) { return __builtin_fmaf((float)h, 3.0517578125e-5f, -1.0f); }
// j-th 16-bit half of a run of 32-bit draws, low half first
DRONE_FN uint32_t half16(const uint32_t* u, uint32_t j) { return (j & 1u) ? (u[j >> 1] >> 16) : (u[j >> 1] & 0xFFFFu); }

// SPEC.md §6, the target alone: what lane_reset draws for episode `episode` of env `env` (halves 3, 4, 5 of the five
// reset draws). Used by the derived-target layout, which stores no target plane for the hover and swarm tasks.
DRONE_FN void derive_target(const KParams& P, uint32_t env, uint32_t episode, float (&tgt)[3]) {
    const uint32_t b = rng_base(P.key_reset, env, episode);
    const uint32_t u1 = rng_draw(b, 1u), u2 = rng_draw(b, 2u);
    tgt[0] = P.target_extent * s16(u1 >> 16);
    tgt[1] = P.target_extent * s16(u2 & 0xFFFFu);
    tgt[2] = P.target_extent * s16(u2 >> 16);
}

inline uint32_t stream_key(uint64_t seed, uint32_t stream) {
    return hash32((uint32_t)seed ^ hash32((uint32_t)(seed >> 32) ^ (0x9E3779B9u * (stream + 1u))));
}

// ---- SPEC.md §1: derived parameters (host, float, fixed order) ----
inline void derive_kparams(const DroneConfig& c, uint64_t seed, KParams& p) {
    p.h = c.dt / (float)c.substeps;
    p.h_half = 0.5f * p.h;
    p.h_sixth = p.h / 6.0f;
    p.hq = 0.5f * p.h;
    p.hq_half = 0.5f * p.h_half;
    p.hq_sixth = 0.5f * p.h_sixth;
    const float inv_mass = 1.0f / c.mass;
    const float inv_ixx = 1.0f / c.ixx, inv_iyy = 1.0f / c.iyy, inv_izz = 1.0f / c.izz;
    const float arm_xy = c.arm * 0.70710678f;
    const float arm_k = arm_xy * c.k_thrust;
    p.kT2_m = (2.0f * c.k_thrust) * inv_mass;
    p.cx = arm_k * inv_ixx;
    p.cy = arm_k * inv_iyy;
    p.cz = c.k_torque * inv_izz;
    p.gxi = (c.izz - c.iyy) * inv_ixx;
    p.gyi = (c.ixx - c.izz) * inv_iyy;
    p.gzi = (c.iyy - c.ixx) * inv_izz;
    p.kdx = c.k_ang_damp * inv_ixx;
    p.kdy = c.k_ang_damp * inv_iyy;
    p.kdz = c.k_ang_damp * inv_izz;
    p.drag_m = c.k_drag * inv_mass;
    p.e_half = (float)exp(-0.5 * (double)p.h / (double)c.motor_tau);  // double exp, rounded once: the same libm on both sides
    p.e_full = (float)exp(-(double)p.h / (double)c.motor_tau);
    p.gravity = c.gravity;
    p.half_max_rpm = 0.5f * c.max_rpm;
    p.hover_rpm = sqrtf((c.mass * c.gravity) / (4.0f * c.k_thrust));
    p.max_rpm = c.max_rpm;
    p.max_vel = c.max_vel;
    p.max_omega = c.max_omega;
    p.inv_max_vel = 1.0f / c.max_vel;
    p.inv_max_omega = 1.0f / c.max_omega;
    p.inv_max_rpm = 1.0f / c.max_rpm;
    p.inv_bound = 1.0f / c.bound;
    p.half_inv_bound = 0.5f * p.inv_bound;
    p.bound = c.bound;
    p.spawn_extent = c.spawn_extent;
    p.target_extent = c.target_extent;
    p.tilt_init = c.tilt_init;
    p.hover_radius = c.hover_radius;
    p.waypoint_radius = c.waypoint_radius;
    p.wind_decay = 1.0f - c.wind_theta * c.dt;
    p.wind_gain = (c.wind_sigma * sqrtf(c.dt)) * 0.0067658754f;
    p.wind_max = c.wind_max;
    p.c_omega = c.c_omega;
    p.c_action = c.c_action;
    p.crash_penalty = c.crash_penalty;
    p.progress_scale = c.progress_scale;
    p.waypoint_bonus = c.waypoint_bonus;
    p.coll_r2 = c.collision_radius * c.collision_radius;
    p.inv_prox_r2 = 1.0f / (c.proximity_radius * c.proximity_radius);
    p.nn_far2 = (4.0f * c.bound) * (4.0f * c.bound);
    p.c_proximity = c.c_proximity;
    p.gate_r2 = c.gate_radius * c.gate_radius;
    p.horizon = (uint32_t)c.horizon;
    p.substeps = (uint32_t)c.substeps;
    p.key_reset = stream_key(seed, kReset);
    p.key_action = stream_key(seed, kAction);
    p.key_wind = stream_key(seed, kWind);
    p.key_waypoint = stream_key(seed, kWaypoint);
    p.env_offset = c.env_offset;
    p.agents = c.task == DRONE_TASK_SWARM ? (uint32_t)c.agents_per_env : 1u;
}

}  // namespace drone

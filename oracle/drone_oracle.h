/*
 * drone_oracle.h — scalar CPU restatement of the drone env. TEST INFRASTRUCTURE.
 *
 * PARITY UNPINNED: the reference snapshot contains no simulator source (the
 * `pufferlib` submodule is an empty directory, /root/reference/.gitmodules:1-3;
 * SURVEY.md §8c), no tests and no golden vectors. This file therefore follows
 * the stage list of BASELINE.json `north_star` (SURVEY.md §8a rows a1–a7) as
 * fixed by this repo's SPEC.md, not a reference file:line. It is the oracle of
 * record for this repo only.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * use anything in oracle/. The product (drone_amd/, include/) never does.
 *
 * Shape: one `Drone` struct per env with PufferLib-ocean-style entry points
 * init / c_reset / c_step over caller-owned obs / action / reward / terminal
 * buffers, one env at a time, array-of-structs, plain float — the opposite of
 * the device layout. What agreement with the HIP path does and does not show:
 * the two share no code in the linker's sense, but they are ONE reading of
 * SPEC.md typed twice by one author (compare c_step below with
 * drone_amd/csrc/drone_lane.hpp): bit-exact agreement proves that the layouts,
 * the kernels' plumbing (tiles, LDS transposes, ballots, resets in flight) and
 * the gcc / hipcc numerics contracts agree — not that SPEC.md was read
 * correctly. The second opinion on the READING is tests/spec_numpy.py, a
 * float64 statement in a different evaluation order, checked on the CPU at
 * <= 1e-5 (tests/test_oracle_independent.py); the pin that would settle it —
 * upstream's C step() — is not in /root/reference.
 *
 * Build with -ffp-contract=off: the only fused operations are the fmaf()
 * calls written out below (SPEC.md preamble).
 */
#ifndef DRONE_ORACLE_H
#define DRONE_ORACLE_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../include/drone_vec.h"

#define STREAM_RESET 0u
#define STREAM_ACTION 1u
#define STREAM_WIND 2u
#define STREAM_WAYPOINT 3u

/* SPEC.md §1: derived parameters, computed once, in float, in this order. */
typedef struct Params {
    float h, h_half, h_sixth;
    float hq, hq_half, hq_sixth;
    float kT2_m, cx, cy, cz;
    float gxi, gyi, gzi;
    float kdx, kdy, kdz;
    float drag_m, e_half, e_full;
    float half_max_rpm, hover_rpm;
    float inv_max_vel, inv_max_omega, inv_max_rpm;
    float inv_bound, half_inv_bound;
    float wind_decay, wind_gain;
    float coll_r2, inv_prox_r2, nn_far2;
    float gate_r2;
} Params;

static inline void params_derive(const DroneConfig* c, Params* p) {
    p->h = c->dt / (float)c->substeps;
    p->h_half = 0.5f * p->h;
    p->h_sixth = p->h / 6.0f;
    p->hq = 0.5f * p->h;
    p->hq_half = 0.5f * p->h_half;
    p->hq_sixth = 0.5f * p->h_sixth;
    const float inv_mass = 1.0f / c->mass;
    const float inv_ixx = 1.0f / c->ixx;
    const float inv_iyy = 1.0f / c->iyy;
    const float inv_izz = 1.0f / c->izz;
    const float arm_xy = c->arm * 0.70710678f;
    const float arm_k = arm_xy * c->k_thrust;
    p->kT2_m = (2.0f * c->k_thrust) * inv_mass;
    p->cx = arm_k * inv_ixx;
    p->cy = arm_k * inv_iyy;
    p->cz = c->k_torque * inv_izz;
    p->gxi = (c->izz - c->iyy) * inv_ixx;
    p->gyi = (c->ixx - c->izz) * inv_iyy;
    p->gzi = (c->iyy - c->ixx) * inv_izz;
    p->kdx = c->k_ang_damp * inv_ixx;
    p->kdy = c->k_ang_damp * inv_iyy;
    p->kdz = c->k_ang_damp * inv_izz;
    p->drag_m = c->k_drag * inv_mass;
    p->e_half = (float)exp(-0.5 * (double)p->h / (double)c->motor_tau);
    p->e_full = (float)exp(-(double)p->h / (double)c->motor_tau);
    p->half_max_rpm = 0.5f * c->max_rpm;
    p->hover_rpm = sqrtf((c->mass * c->gravity) / (4.0f * c->k_thrust));
    p->inv_max_vel = 1.0f / c->max_vel;
    p->inv_max_omega = 1.0f / c->max_omega;
    p->inv_max_rpm = 1.0f / c->max_rpm;
    p->inv_bound = 1.0f / c->bound;
    p->half_inv_bound = 0.5f * p->inv_bound;
    p->wind_decay = 1.0f - c->wind_theta * c->dt;
    p->wind_gain = (c->wind_sigma * sqrtf(c->dt)) * 0.0067658754f;
    p->coll_r2 = c->collision_radius * c->collision_radius;
    p->inv_prox_r2 = 1.0f / (c->proximity_radius * c->proximity_radius);
    p->nn_far2 = (4.0f * c->bound) * (4.0f * c->bound);
    p->gate_r2 = c->gate_radius * c->gate_radius;
}

/* SPEC.md §2 */
static inline uint32_t hash32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
static inline uint32_t stream_key(uint64_t seed, uint32_t stream) {
    return hash32((uint32_t)seed ^ hash32((uint32_t)(seed >> 32) ^ (0x9E3779B9u * (stream + 1u))));
}
static inline uint32_t rng_base(uint32_t key, uint32_t env, uint32_t ctr) {
    return hash32(hash32(key ^ env) + ctr * 0x9E3779B9u);
}
static inline uint32_t rng_draw(uint32_t base, uint32_t d) { return hash32(base + d * 0x85EBCA6Bu); }
static inline float u01(uint32_t u) { return (float)(u >> 8) * 5.9604645e-8f; }
static inline float sym(uint32_t u) { return fmaf(2.0f, u01(u), -1.0f); }
static inline float s16(uint32_t h) { return (float)((int)h - 32768) * 3.0517578125e-5f; }
/* C99 fmaxf / fminf semantics for the operands the spec feeds them (a NaN operand
 * yields the other one; no signalling NaNs, no zero-vs-zero ties), written out so
 * that the compiler inlines compare + select instead of calling libm. */
static inline float fmax_(float a, float b) { return (a >= b || b != b) ? a : b; }
static inline float fmin_(float a, float b) { return (a <= b || b != b) ? a : b; }
static inline float clampc(float x, float lo, float hi) { return fmin_(fmax_(x, lo), hi); }

static inline void random_action(uint32_t key_action, uint32_t env, uint32_t gstep, float a[4]) {
    const uint32_t k = hash32(key_action ^ env);
    const uint32_t h0 = hash32(k + (2u * gstep) * 0x9E3779B9u);
    const uint32_t h1 = hash32(k + (2u * gstep + 1u) * 0x9E3779B9u);
    a[0] = s16(h0 & 0xFFFFu);
    a[1] = s16(h0 >> 16);
    a[2] = s16(h1 & 0xFFFFu);
    a[3] = s16(h1 >> 16);
}

typedef struct Log {
    float perf, score, episode_return, episode_length, n, oob;
} Log;

/* SPEC.md §3: pos, vel, quat, omega advance by RK4; rpm in closed form. */
typedef struct State {
    float pos[3], vel[3], quat[4], omega[3], rpm[4];
} State;

typedef struct Drone {
    Log log; /* per-env accumulators (sums), drained by vec_log */
    float* observations;
    float* actions;
    float* rewards;
    unsigned char* terminals;
    unsigned char* truncations;
    State s;
    float target[3], wind[3];
    float gate_n[3]; /* task 3: unit normal of the current gate (its centre is `target`) */
    float ep_return;
    uint32_t tick, episode, score_count;
    uint32_t env_id; /* global id */
    float scratch_a2, scratch_prev_dist; /* carried from step_integrate to step_finish */
    float scratch_p0[3];                 /* task 3: position before integrating */
    /* shared, read-only */
    const DroneConfig* cfg;
    const Params* par;
    const uint32_t* keys;  /* [4] stream keys */
    const uint32_t* gstep; /* vec-level step counter */
} Drone;

/* SPEC.md §4: rotor inputs at one instant: twice the thrust acceleration, torques over inertia. */
typedef struct Rotor {
    float aT2, tx, ty, tz;
} Rotor;

static inline Rotor rotor_inputs(const Params* p, const float rpm[4]) {
    float q[4];
    for (int i = 0; i < 4; i++) q[i] = rpm[i] * rpm[i];
    const float s01 = q[0] + q[1];
    const float s23 = q[2] + q[3];
    Rotor u;
    u.aT2 = p->kT2_m * (s01 + s23);
    u.tx = p->cx * (s01 - s23);
    u.ty = p->cy * ((q[1] + q[2]) - (q[0] + q[3]));
    u.tz = p->cz * ((q[0] + q[2]) - (q[1] + q[3]));
    return u;
}

/* SPEC.md §4: deriv of (vel, quat, omega) given the rotor inputs of that instant. Only those ten
 * components of `D` are written; dpos = vel is handled by the caller, rpm in closed form. */
static inline void deriv(const Drone* env, const State* S, const Rotor* u, State* D) {
    const DroneConfig* c = env->cfg;
    const Params* p = env->par;
    const float w = S->quat[0], x = S->quat[1], y = S->quat[2], z = S->quat[3];
    const float ox = S->omega[0], oy = S->omega[1], oz = S->omega[2];
    const float zx = fmaf(x, z, w * y);
    const float zy = fmaf(y, z, -(w * x));
    const float zzh = 0.5f - fmaf(x, x, y * y);
    D->vel[0] = fmaf(u->aT2, zx, -(p->drag_m * (S->vel[0] - env->wind[0])));
    D->vel[1] = fmaf(u->aT2, zy, -(p->drag_m * (S->vel[1] - env->wind[1])));
    D->vel[2] = fmaf(-p->drag_m, S->vel[2] - env->wind[2], fmaf(u->aT2, zzh, -c->gravity));
    D->omega[0] = fmaf(-p->gxi, oy * oz, fmaf(-p->kdx, ox, u->tx));
    D->omega[1] = fmaf(-p->gyi, oz * ox, fmaf(-p->kdy, oy, u->ty));
    D->omega[2] = fmaf(-p->gzi, ox * oy, fmaf(-p->kdz, oz, u->tz));
    D->quat[0] = -fmaf(x, ox, fmaf(y, oy, z * oz)); /* q (x) (0, omega) = 2 qdot; the 1/2 is in hq* */
    D->quat[1] = fmaf(w, ox, fmaf(y, oz, -(z * oy)));
    D->quat[2] = fmaf(w, oy, fmaf(z, ox, -(x * oz)));
    D->quat[3] = fmaf(w, oz, fmaf(x, oy, -(y * ox)));
}

#define BODY_FIRST 3  /* State components 3..12 are vel, quat, omega: the ones RK4 feeds back */
#define BODY_LAST 12
#define QUAT_FIRST 6 /* components 6..9 of State are the quaternion */
#define QUAT_LAST 9

/* SPEC.md §4: one substep. cmd[i] = commanded rotor speed, held over the step. */
static inline void rk4_substep(Drone* env, const float cmd[4]) {
    const Params* p = env->par;
    State k, A, acc;
    float* S = (float*)&env->s;
    float* kk = (float*)&k;
    float* AA = (float*)&A;
    float* ac = (float*)&acc;
    float H[13], Hh[13], H6[13];
    for (int c = BODY_FIRST; c <= BODY_LAST; c++) {
        const int is_q = c >= QUAT_FIRST && c <= QUAT_LAST;
        H[c] = is_q ? p->hq : p->h;
        Hh[c] = is_q ? p->hq_half : p->h_half;
        H6[c] = is_q ? p->hq_sixth : p->h_sixth;
    }
    /* rotor speeds: exact first-order lag toward cmd, at t + h/2 and t + h */
    float r_half[4], r_full[4];
    for (int i = 0; i < 4; i++) {
        const float d = env->s.rpm[i] - cmd[i];
        r_half[i] = fmaf(p->e_half, d, cmd[i]);
        r_full[i] = fmaf(p->e_full, d, cmd[i]);
    }
    const Rotor u0 = rotor_inputs(p, env->s.rpm), uh = rotor_inputs(p, r_half), uf = rotor_inputs(p, r_full);
    float vsum[3]; /* v1 + 2 v2 + 2 v3 (+ v4 at the end): dpos = vel */
    A = env->s;
    deriv(env, &env->s, &u0, &k);
    for (int i = 0; i < 3; i++) vsum[i] = env->s.vel[i];
    for (int c = BODY_FIRST; c <= BODY_LAST; c++) {
        ac[c] = kk[c];
        AA[c] = fmaf(Hh[c], kk[c], S[c]);
    }
    deriv(env, &A, &uh, &k);
    for (int i = 0; i < 3; i++) vsum[i] = fmaf(2.0f, A.vel[i], vsum[i]);
    for (int c = BODY_FIRST; c <= BODY_LAST; c++) {
        ac[c] = fmaf(2.0f, kk[c], ac[c]);
        AA[c] = fmaf(Hh[c], kk[c], S[c]);
    }
    deriv(env, &A, &uh, &k);
    for (int i = 0; i < 3; i++) vsum[i] = fmaf(2.0f, A.vel[i], vsum[i]);
    for (int c = BODY_FIRST; c <= BODY_LAST; c++) {
        ac[c] = fmaf(2.0f, kk[c], ac[c]);
        AA[c] = fmaf(H[c], kk[c], S[c]);
    }
    deriv(env, &A, &uf, &k);
    for (int i = 0; i < 3; i++) env->s.pos[i] = fmaf(p->h_sixth, vsum[i] + A.vel[i], env->s.pos[i]);
    for (int c = BODY_FIRST; c <= BODY_LAST; c++) {
        ac[c] = ac[c] + kk[c];
        S[c] = fmaf(H6[c], ac[c], S[c]);
    }
    for (int i = 0; i < 4; i++) env->s.rpm[i] = r_full[i];
}

static inline float target_dist(const Drone* env) {
    const float dx = env->target[0] - env->s.pos[0];
    const float dy = env->target[1] - env->s.pos[1];
    const float dz = env->target[2] - env->s.pos[2];
    return sqrtf(fmaf(dx, dx, fmaf(dy, dy, dz * dz)));
}

/* SPEC.md §7 */
static inline void compute_observations(Drone* env) {
    const DroneConfig* c = env->cfg;
    const Params* p = env->par;
    const State* s = &env->s;
    const float w = s->quat[0], x = s->quat[1], y = s->quat[2], z = s->quat[3];
    const float r00 = fmaf(-2.0f, fmaf(y, y, z * z), 1.0f);
    const float r01 = 2.0f * fmaf(x, y, -(w * z));
    const float r02 = 2.0f * fmaf(x, z, w * y);
    const float r10 = 2.0f * fmaf(x, y, w * z);
    const float r11 = fmaf(-2.0f, fmaf(x, x, z * z), 1.0f);
    const float r12 = 2.0f * fmaf(y, z, -(w * x));
    const float r20 = 2.0f * fmaf(x, z, -(w * y));
    const float r21 = 2.0f * fmaf(y, z, w * x);
    const float r22 = fmaf(-2.0f, fmaf(x, x, y * y), 1.0f);
    float* o = env->observations;
    const float* v = s->vel;
    o[0] = fmaf(r00, v[0], fmaf(r10, v[1], r20 * v[2])) * p->inv_max_vel;
    o[1] = fmaf(r01, v[0], fmaf(r11, v[1], r21 * v[2])) * p->inv_max_vel;
    o[2] = fmaf(r02, v[0], fmaf(r12, v[1], r22 * v[2])) * p->inv_max_vel;
    for (int i = 0; i < 3; i++) o[3 + i] = s->omega[i] * p->inv_max_omega;
    for (int i = 0; i < 4; i++) o[6 + i] = s->quat[i];
    for (int i = 0; i < 4; i++) o[10 + i] = s->rpm[i] * p->inv_max_rpm;
    const float e0 = env->target[0] - s->pos[0];
    const float e1 = env->target[1] - s->pos[1];
    const float e2 = env->target[2] - s->pos[2];
    o[14] = fmaf(r00, e0, fmaf(r10, e1, r20 * e2)) * p->half_inv_bound;
    o[15] = fmaf(r01, e0, fmaf(r11, e1, r21 * e2)) * p->half_inv_bound;
    o[16] = fmaf(r02, e0, fmaf(r12, e1, r22 * e2)) * p->half_inv_bound;
    for (int i = 0; i < 3; i++) o[17 + i] = s->pos[i] * p->inv_bound;
    (void)c;
}

static inline float dot3(const float a[3], const float b[3]) { return fmaf(a[0], b[0], fmaf(a[1], b[1], a[2] * b[2])); }

/* SPEC.md §11: unit(e) */
static inline void unit3(const float e[3], float out[3]) {
    const float inv = 1.0f / sqrtf(dot3(e, e) + 1e-12f);
    for (int i = 0; i < 3; i++) out[i] = e[i] * inv;
}

/* SPEC.md §6 (state only; the episode counter is the caller's business) */
static inline void reset_state(Drone* env) {
    const DroneConfig* c = env->cfg;
    const uint32_t b = rng_base(env->keys[STREAM_RESET], env->env_id, env->episode);
    /* nine values from five 32-bit draws, 16 bits each, low half first */
    float val[9];
    for (uint32_t j = 0; j < 9; j++) {
        const uint32_t u = rng_draw(b, j / 2u);
        val[j] = s16((j & 1u) ? (u >> 16) : (u & 0xFFFFu));
    }
    float t[3];
    for (uint32_t i = 0; i < 3; i++) {
        env->s.pos[i] = c->spawn_extent * val[i];
        env->target[i] = c->target_extent * val[3 + i];
        t[i] = c->tilt_init * val[6 + i];
    }
    const float n2 = fmaf(t[0], t[0], fmaf(t[1], t[1], fmaf(t[2], t[2], 1.0f)));
    /* SPEC v5: two Newton steps of 1/sqrt(n2) about 1 — no square root, no division */
    const float s1 = fmaf(-0.5f, n2, 1.5f);
    const float m = (n2 * s1) * s1;
    const float s2 = fmaf(-0.5f, m, 1.5f);
    const float sc = s1 * s2;
    env->s.quat[0] = sc;
    env->s.quat[1] = t[0] * sc;
    env->s.quat[2] = t[1] * sc;
    env->s.quat[3] = t[2] * sc;
    for (int i = 0; i < 3; i++) {
        env->s.vel[i] = 0.0f;
        env->s.omega[i] = 0.0f;
        env->wind[i] = 0.0f;
    }
    for (int i = 0; i < 4; i++) env->s.rpm[i] = env->par->hover_rpm;
    env->tick = 0;
    env->score_count = 0;
    env->ep_return = 0.0f;
    if (c->task == DRONE_TASK_RACE) { /* SPEC.md §11: gate 0 faces the spawn point */
        float e[3];
        for (int i = 0; i < 3; i++) e[i] = env->target[i] - env->s.pos[i];
        unit3(e, env->gate_n);
    }
}

static inline void init(Drone* env) {
    memset(&env->log, 0, sizeof(Log));
    memset(&env->s, 0, sizeof(State));
    env->episode = 0;
}

static inline void race_observations(Drone* env);

/* First episode of this env (vec_reset): SPEC.md §6 last paragraph. */
static inline void c_reset(Drone* env) {
    memset(&env->log, 0, sizeof(Log));
    env->episode = 0;
    reset_state(env);
    compute_observations(env);
    if (env->cfg->task == DRONE_TASK_RACE) race_observations(env);
    env->rewards[0] = 0.0f;
    env->terminals[0] = 0;
    env->truncations[0] = 0;
}

/* SPEC.md §5 steps 1-5: actions, wind, integration, tick, distance. What the
 * later phases need is parked in the env (scratch fields). */
static inline void step_integrate(Drone* env) {
    const DroneConfig* c = env->cfg;
    const Params* p = env->par;
    State* s = &env->s;
    float a[4], cmd[4];
    for (int i = 0; i < 4; i++) {
        a[i] = clampc(env->actions[i], -1.0f, 1.0f);
        cmd[i] = p->half_max_rpm * (a[i] + 1.0f);
    }
    env->scratch_a2 = fmaf(a[0], a[0], fmaf(a[1], a[1], fmaf(a[2], a[2], a[3] * a[3])));
    env->scratch_prev_dist = 0.0f;
    if (c->task == DRONE_TASK_WAYPOINT) {
        const uint32_t b = rng_base(env->keys[STREAM_WIND], env->env_id, *env->gstep);
        for (uint32_t i = 0; i < 3; i++) {
            const uint32_t u = rng_draw(b, i);
            const uint32_t sum = (u & 255u) + ((u >> 8) & 255u) + ((u >> 16) & 255u) + (u >> 24);
            const float xi = (float)((int)sum - 510);
            env->wind[i] = clampc(fmaf(p->wind_decay, env->wind[i], p->wind_gain * xi), -c->wind_max, c->wind_max);
        }
        env->scratch_prev_dist = target_dist(env);
    }
    if (c->task == DRONE_TASK_RACE) {
        for (int i = 0; i < 3; i++) env->scratch_p0[i] = s->pos[i];
        env->scratch_prev_dist = target_dist(env);
    }
    for (int k = 0; k < c->substeps; k++) rk4_substep(env, cmd);
    {
        float* q = s->quat;
        const float n2 = fmaf(q[0], q[0], fmaf(q[1], q[1], fmaf(q[2], q[2], q[3] * q[3])));
        const float sc = fmaf(-0.5f, n2, 1.5f); /* one Newton step of 1/sqrt(n2) about 1 */
        for (int i = 0; i < 4; i++) q[i] = q[i] * sc;
        for (int i = 0; i < 3; i++) s->vel[i] = clampc(s->vel[i], -c->max_vel, c->max_vel);
        for (int i = 0; i < 3; i++) s->omega[i] = clampc(s->omega[i], -c->max_omega, c->max_omega);
        /* rotor speeds stay between their old value and cmd, both in [0, max_rpm]: no clamp */
    }
    env->tick += 1;
}

/* SPEC.md §10: nearest neighbour of agent `i` among the `A` agents of its swarm. */
static inline void neighbour(const Drone* swarm, int A, int i, float* nn_d2, float nn_e[3]) {
    const Params* p = swarm[i].par;
    float best = p->nn_far2;
    nn_e[0] = nn_e[1] = nn_e[2] = 0.0f;
    for (int d = 1; d < A; d++) {
        const Drone* o = &swarm[(i + d) % A];
        const float ex = o->s.pos[0] - swarm[i].s.pos[0];
        const float ey = o->s.pos[1] - swarm[i].s.pos[1];
        const float ez = o->s.pos[2] - swarm[i].s.pos[2];
        const float d2 = fmaf(ex, ex, fmaf(ey, ey, ez * ez));
        if (d2 < best) {
            best = d2;
            nn_e[0] = ex;
            nn_e[1] = ey;
            nn_e[2] = ez;
        }
    }
    *nn_d2 = best;
}

/* SPEC.md §5 steps 5-9 (+ §10 steps 6-7 for the swarm task). */
static inline void step_finish(Drone* env, float nn_d2) {
    const DroneConfig* c = env->cfg;
    const Params* p = env->par;
    State* s = &env->s;
    const float dist = target_dist(env);
    int oob = !(fabsf(s->pos[0]) <= c->bound) || !(fabsf(s->pos[1]) <= c->bound) || !(fabsf(s->pos[2]) <= c->bound);
    if (c->task == DRONE_TASK_SWARM) oob = oob || (nn_d2 < p->coll_r2); /* crash = out of the box or collided */
    const int trunc = !oob && env->tick >= (uint32_t)c->horizon;

    const float w2 = fmaf(s->omega[0], s->omega[0], fmaf(s->omega[1], s->omega[1], s->omega[2] * s->omega[2]));
    const float pen = fmaf(c->c_omega, w2, c->c_action * env->scratch_a2);
    float r;
    if (c->task == DRONE_TASK_RACE) {
        r = c->progress_scale * (env->scratch_prev_dist - dist) - pen;
        float d0[3], d1[3];
        for (int i = 0; i < 3; i++) {
            d0[i] = env->scratch_p0[i] - env->target[i];
            d1[i] = s->pos[i] - env->target[i];
        }
        const float s0 = dot3(env->gate_n, d0), s1 = dot3(env->gate_n, d1);
        if (!oob && s0 < 0.0f && s1 >= 0.0f) { /* crossed the gate plane forwards */
            const float t = s0 / (s0 - s1);
            float m[3];
            for (int i = 0; i < 3; i++) m[i] = fmaf(t, s->pos[i] - env->scratch_p0[i], env->scratch_p0[i]) - env->target[i];
            if (dot3(m, m) < p->gate_r2) { /* through the ring */
                r += c->waypoint_bonus;
                env->score_count += 1;
                const uint32_t b = rng_base(env->keys[STREAM_WAYPOINT], env->env_id, env->episode);
                float cn[3], e[3];
                for (uint32_t i = 0; i < 3; i++) {
                    cn[i] = c->target_extent * sym(rng_draw(b, 3u * env->score_count + i));
                    e[i] = cn[i] - env->target[i];
                }
                unit3(e, env->gate_n);
                for (int i = 0; i < 3; i++) env->target[i] = cn[i];
            }
        }
    } else if (c->task != DRONE_TASK_WAYPOINT) {
        r = fmaf(-p->half_inv_bound, dist, 1.0f) - pen;
        if (dist < c->hover_radius) env->score_count += 1;
        if (c->task == DRONE_TASK_SWARM) r = r - c->c_proximity * fmax_(0.0f, fmaf(-nn_d2, p->inv_prox_r2, 1.0f));
    } else {
        r = c->progress_scale * (env->scratch_prev_dist - dist) - pen;
        if (!oob && dist < c->waypoint_radius) {
            r += c->waypoint_bonus;
            env->score_count += 1;
            const uint32_t b = rng_base(env->keys[STREAM_WAYPOINT], env->env_id, env->episode);
            for (uint32_t i = 0; i < 3; i++)
                env->target[i] = c->target_extent * sym(rng_draw(b, 3u * env->score_count + i));
        }
    }
    if (oob) r -= c->crash_penalty;

    env->ep_return += r;
    env->rewards[0] = r;
    env->terminals[0] = (unsigned char)oob;
    env->truncations[0] = (unsigned char)trunc;

    if (oob || trunc) {
        /* SPEC v5: tasks 0 and 2 log the COUNT of steps within hover_radius; vec_log divides by the steps flown */
        const float score = (float)env->score_count;
        float perf = score;
        if (c->task == DRONE_TASK_WAYPOINT || c->task == DRONE_TASK_RACE) perf = env->score_count >= 8u ? 1.0f : score * 0.125f;
        env->log.perf += perf;
        env->log.score += score;
        env->log.episode_return += env->ep_return;
        env->log.episode_length += (float)env->tick;
        env->log.n += 1.0f;
        env->log.oob += oob ? 1.0f : 0.0f;
        env->episode += 1;
        reset_state(env);
    }
}

/* SPEC.md §10 step 10: the four neighbour observations of agent `i`. */
static inline void swarm_observations(Drone* swarm, int A, int i) {
    Drone* env = &swarm[i];
    const Params* p = env->par;
    float nn_d2, e[3];
    neighbour(swarm, A, i, &nn_d2, e);
    const float w = env->s.quat[0], x = env->s.quat[1], y = env->s.quat[2], z = env->s.quat[3];
    const float r00 = fmaf(-2.0f, fmaf(y, y, z * z), 1.0f), r01 = 2.0f * fmaf(x, y, -(w * z)), r02 = 2.0f * fmaf(x, z, w * y);
    const float r10 = 2.0f * fmaf(x, y, w * z), r11 = fmaf(-2.0f, fmaf(x, x, z * z), 1.0f), r12 = 2.0f * fmaf(y, z, -(w * x));
    const float r20 = 2.0f * fmaf(x, z, -(w * y)), r21 = 2.0f * fmaf(y, z, w * x), r22 = fmaf(-2.0f, fmaf(x, x, y * y), 1.0f);
    float* o = env->observations;
    o[20] = fmaf(r00, e[0], fmaf(r10, e[1], r20 * e[2])) * p->half_inv_bound;
    o[21] = fmaf(r01, e[0], fmaf(r11, e[1], r21 * e[2])) * p->half_inv_bound;
    o[22] = fmaf(r02, e[0], fmaf(r12, e[1], r22 * e[2])) * p->half_inv_bound;
    o[23] = (nn_d2 * p->inv_bound) * p->inv_bound;
}

/* SPEC.md §11 step 10: the gate normal in the body frame and the signed distance to its plane. */
static inline void race_observations(Drone* env) {
    const Params* p = env->par;
    const float w = env->s.quat[0], x = env->s.quat[1], y = env->s.quat[2], z = env->s.quat[3];
    const float r00 = fmaf(-2.0f, fmaf(y, y, z * z), 1.0f), r01 = 2.0f * fmaf(x, y, -(w * z)), r02 = 2.0f * fmaf(x, z, w * y);
    const float r10 = 2.0f * fmaf(x, y, w * z), r11 = fmaf(-2.0f, fmaf(x, x, z * z), 1.0f), r12 = 2.0f * fmaf(y, z, -(w * x));
    const float r20 = 2.0f * fmaf(x, z, -(w * y)), r21 = 2.0f * fmaf(y, z, w * x), r22 = fmaf(-2.0f, fmaf(x, x, y * y), 1.0f);
    const float* n = env->gate_n;
    float* o = env->observations;
    o[20] = fmaf(r00, n[0], fmaf(r10, n[1], r20 * n[2]));
    o[21] = fmaf(r01, n[0], fmaf(r11, n[1], r21 * n[2]));
    o[22] = fmaf(r02, n[0], fmaf(r12, n[1], r22 * n[2]));
    float d[3];
    for (int i = 0; i < 3; i++) d[i] = env->s.pos[i] - env->target[i];
    o[23] = dot3(n, d) * p->inv_bound;
}

/* One step of a single-agent env (tasks 0, 1 and 3): SPEC.md §5, §11. */
static inline void c_step(Drone* env) {
    step_integrate(env);
    step_finish(env, 0.0f);
    compute_observations(env);
    if (env->cfg->task == DRONE_TASK_RACE) race_observations(env);
}

/* One step of a swarm of A agents (task 2): SPEC.md §10. */
static inline void c_step_swarm(Drone* swarm, int A) {
    float nn_d2[64], e[3];
    for (int i = 0; i < A; i++) step_integrate(&swarm[i]);
    for (int i = 0; i < A; i++) neighbour(swarm, A, i, &nn_d2[i], e); /* all on post-integration positions */
    for (int i = 0; i < A; i++) step_finish(&swarm[i], nn_d2[i]);
    for (int i = 0; i < A; i++) {
        compute_observations(&swarm[i]);
        swarm_observations(swarm, A, i);
    }
}

static inline void c_reset_swarm(Drone* swarm, int A) {
    for (int i = 0; i < A; i++) c_reset(&swarm[i]);
    for (int i = 0; i < A; i++) swarm_observations(swarm, A, i);
}

#endif /* DRONE_ORACLE_H */

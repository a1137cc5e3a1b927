// drone_lane.hpp — what ONE lane does for ONE drone, entirely in registers:
// rotor model, RK4 Newton–Euler integration, bounds, reward, episodic reset,
// observation. Implements SPEC.md §4–§7 (BASELINE.json north_star stages
// a2–a7; the reference has no source to cite — /root/reference/.gitmodules:1-3).
//
// Numerics contract: compiled with -ffp-contract=off; the only fused
// operations are the fma() calls below, `/` and sqrtf are correctly rounded
// (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt), no transcendental
// calls. That is what makes the result bit-identical to the scalar CPU oracle.
//
// `P` is a reference to the KParams block (kernarg segment or LDS in the kernels).
#pragma once

#include "drone_params.hpp"

// The RK4 substep exists twice: scalar, and in two-wide f32 instructions with hand-placed modifiers (drone_pk.hpp).
// Same results bit for bit (same expression trees). Which one runs is a template argument (PK) that the kernels choose
// at launch: packed wins where a SIMD holds one wave (65 536 envs: fused rollout -8.8 %, step_many -3.9 %) and loses
// from two waves per SIMD on (131 072 envs: +3...4 %; 2^20 envs: rollout +11 %) — profiles/r03_ab/ab_pk_*.txt.
// DRONE_PK_RK4=0 compiles the packed form out (it needs the constants in SGPRs: not with the LDS-staged constants variant).
// DRONE_PK_DEFAULT: what PK defaults to where the caller does not say (the host test harness builds once with each).
#ifndef DRONE_PK_RK4
#define DRONE_PK_RK4 1
#endif
#ifndef DRONE_PK_DEFAULT
#define DRONE_PK_DEFAULT 0
#endif
#if defined(DRONE_PARAMS_IN_LDS) && DRONE_PARAMS_IN_LDS  // the packed form takes its constants as SGPR pairs: kernarg constants only
#undef DRONE_PK_RK4
#define DRONE_PK_RK4 0
#endif
#if DRONE_PK_RK4
#include "drone_pk.hpp"
#endif

namespace drone {

#define fma_(a, b, c) __builtin_fmaf((a), (b), (c))

// the state the integrator advances: 13 rigid-body components by RK4, 4 rotor speeds in closed form
struct Dyn {
    float p[3], v[3], q[4], o[3], r[4];
};

// SPEC.md §4: what the rotors feed into the rigid body at one instant — twice the thrust acceleration and
// the three body torques over the inertia.
struct RotorIn {
    float aT2, tx, ty, tz;
};

struct Lane {
    Dyn s;
    RotorIn u;  // rotor inputs at the CURRENT rotor speeds s.r. Kernels that keep the state in registers from step to step
                // (CARRY) maintain it: the inputs at the end of one substep are those at the start of the next — the same
                // function of the same floats, so carrying them is bit-identical to recomputing them (18 operations saved
                // per substep); kernels that load the state from HBM recompute it at the start of the step.
    float tgt[3];
    float wind[3];  // task 1: wind; task 3: unit normal of the current gate (the dynamics see no wind there)
    float ep_return;
    uint32_t tick, episode, score_count;
};

struct StepOut {
    float reward;
    bool oob, trunc;
    bool target_changed;  // target plane must be written back
    // valid when oob || trunc: this episode's contribution to the log sums
    float perf, score, ep_return, ep_len;
};

// SPEC.md §4 clampc: fminf(fmaxf(x, lo), hi) with C99 NaN semantics (a NaN x yields lo). v_med3_f32 returns
// min3 when an operand is NaN, i.e. lo as well, and the median otherwise: one instruction instead of two.
DRONE_FN float clampc(float x, float lo, float hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_fmed3f(x, lo, hi);
#else
    return __builtin_fminf(__builtin_fmaxf(x, lo), hi);
#endif
}

DRONE_FN float dot3(const float (&a)[3], const float (&b)[3]) { return fma_(a[0], b[0], fma_(a[1], b[1], a[2] * b[2])); }

// SPEC.md §11: unit(e)
DRONE_FN void unit3(const float (&e)[3], float (&out)[3]) {
    const float inv = 1.0f / sqrtf(dot3(e, e) + 1e-12f);
#pragma unroll
    for (int i = 0; i < 3; i++) out[i] = e[i] * inv;
}

// SPEC.md §4: RotorIn from the rotor speeds at one instant. 18 operations.
DRONE_FN RotorIn rotor_inputs(const KParams& P, const float (&r)[4]) {
    const float q0 = r[0] * r[0], q1 = r[1] * r[1], q2 = r[2] * r[2], q3 = r[3] * r[3];
    const float s01 = q0 + q1, s23 = q2 + q3;
    RotorIn u;
    u.aT2 = P.kT2_m * (s01 + s23);
    u.tx = P.cx * (s01 - s23);
    u.ty = P.cy * ((q1 + q2) - (q0 + q3));
    u.tz = P.cz * ((q0 + q2) - (q1 + q3));
    return u;
}

// SPEC.md §4: derivative of the 10 components (v, q, o) that feed back; dp = v needs no work. 34 operations (37 with wind).
struct Body {
    float v[3], q[4], o[3];
};
template <int TASK>
DRONE_FN void deriv(const KParams& P, const Body& S, const RotorIn& u, const float (&wind)[3], Body& D) {
    const float w = S.q[0], x = S.q[1], y = S.q[2], z = S.q[3];
    const float ox = S.o[0], oy = S.o[1], oz = S.o[2];
    const float zx = fma_(x, z, w * y);
    const float zy = fma_(y, z, -(w * x));
    const float zzh = 0.5f - fma_(x, x, y * y);
    if (TASK == DRONE_TASK_WAYPOINT) {
        D.v[0] = fma_(u.aT2, zx, -(P.drag_m * (S.v[0] - wind[0])));
        D.v[1] = fma_(u.aT2, zy, -(P.drag_m * (S.v[1] - wind[1])));
        D.v[2] = fma_(-P.drag_m, S.v[2] - wind[2], fma_(u.aT2, zzh, -P.gravity));
    } else {  // wind == 0 and v - 0 is exact (SPEC.md §4)
        D.v[0] = fma_(u.aT2, zx, -(P.drag_m * S.v[0]));
        D.v[1] = fma_(u.aT2, zy, -(P.drag_m * S.v[1]));
        D.v[2] = fma_(-P.drag_m, S.v[2], fma_(u.aT2, zzh, -P.gravity));
    }
    D.o[0] = fma_(-P.gxi, oy * oz, fma_(-P.kdx, ox, u.tx));
    D.o[1] = fma_(-P.gyi, oz * ox, fma_(-P.kdy, oy, u.ty));
    D.o[2] = fma_(-P.gzi, ox * oy, fma_(-P.kdz, oz, u.tz));
    D.q[0] = -fma_(x, ox, fma_(y, oy, z * oz));  // q (x) (0, omega) = 2 qdot; the 1/2 is in hq*
    D.q[1] = fma_(w, ox, fma_(y, oz, -(z * oy)));
    D.q[2] = fma_(w, oy, fma_(z, ox, -(x * oz)));
    D.q[3] = fma_(w, oz, fma_(x, oy, -(y * ox)));
}

#if DRONE_PK_RK4
// ---------------------------------------------------------------------------------------------------------------
// The substep in pair layout. Pairs (low, high):  V = (v0, v1)   X = (v2, o0)   O = (o1, o2)   Q = (q1, q2) = (x, y)
// W = (q0, q3) = (w, z)   R01, R23 rotor speeds   U0 = (aT2, tx)   Utz = (ty, tz).
// Rows that integrate with h (V, X, O) and with hq (Q, W) never share a pair; the step sizes travel as SGPR pairs
// (h*, hq*) and each instruction picks its half. The W pair carries MINUS the derivative of q0 in its low lane
// (SPEC: dq0 = -fma(x, ox, fma(y, oy, z oz))): the sign is applied by a neg modifier wherever that lane is consumed
// (-(a + b) = (-a) + (-b) and fma(c, -k, -a) = -fma(c, k, a) hold exactly in IEEE arithmetic), which saves the four
// negations per substep. 181 instructions (85 of them packed) instead of 266.
// ---------------------------------------------------------------------------------------------------------------
struct BodyPk {
    pk::f2 V, X, O, Q, W;
};
struct RotorPk {
    pk::f2 U0, Utz;
};
struct PkConsts {
    pk::f2 hH, hF, hS, two, eHF, KD, G, C, dr;
};
// A constant on its way into an SGPR pair. The empty asm makes the value opaque: without it the optimiser merges the
// loads of neighbouring KParams fields (kdy kdz, gyi gzi, ...) into <2 x float> loads of the kernarg block, and the
// mixed vector / scalar view of that block then keeps 100 bytes of it in scratch memory, re-read inside the substep loop.
DRONE_FN float sgpr_(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+s"(x));
#endif
    return x;
}
DRONE_FN PkConsts pk_consts(const KParams& P) {
    PkConsts c;
    c.hH = pk::make(sgpr_(P.h_half), sgpr_(P.hq_half));
    c.hF = pk::make(sgpr_(P.h), sgpr_(P.hq));
    c.hS = pk::make(sgpr_(P.h_sixth), sgpr_(P.hq_sixth));
    c.two = pk::make(2.0f, 2.0f);
    c.eHF = pk::make(sgpr_(P.e_half), sgpr_(P.e_full));
    c.KD = pk::make(sgpr_(P.kdy), sgpr_(P.kdz));
    c.G = pk::make(sgpr_(P.gyi), sgpr_(P.gzi));
    c.C = pk::make(sgpr_(P.cy), sgpr_(P.cz));
    c.dr = pk::make(sgpr_(P.drag_m), 0.0f);
    return c;
}

// SPEC.md section 4 rotor inputs from the rotor-speed pairs: 12 instructions (18 scalar)
DRONE_FN RotorPk rotor_inputs_pk(const KParams& P, const PkConsts& c, pk::f2 R01, pk::f2 R23) {
    const pk::f2 Q01 = pk::mul(R01, R01), Q23 = pk::mul(R23, R23);  // (q0, q1), (q2, q3)
    const float s01 = Q01.x + Q01.y, s23 = Q23.x + Q23.y;
    RotorPk u;
    u.U0 = pk::make(P.kT2_m * (s01 + s23), P.cx * (s01 - s23));
    const pk::f2 A = pk::add_swap_lo(Q01, Q23);  // (q1 + q2, q0 + q2)
    const pk::f2 B = pk::add_same_hi(Q01, Q23);  // (q0 + q3, q1 + q3)
    u.Utz = pk::mul_s(c.C, pk::sub(A, B));       // (cy ((q1+q2) - (q0+q3)), cz ((q0+q2) - (q1+q3)))
    return u;
}

// derivative of the 10 feedback components at state B under rotor inputs u: 27 instructions (34 scalar)
template <int TASK>
DRONE_FN void deriv_pk(const KParams& P, const PkConsts& c, const BodyPk& B, const RotorPk& u, const float (&wind)[3], BodyPk& K) {
    const float x = B.Q.x, y = B.Q.y, w = B.W.x, z = B.W.y;
    const float v2 = B.X.x, ox = B.X.y, oy = B.O.x, oz = B.O.y;
    const pk::f2 t = pk::mul_lo_swap_nhi(B.W, B.Q);  // (w y, -(w x))
    const pk::f2 Z = pk::fma_b_hi(B.Q, B.W, t);      // zx = fma(x, z, w y), zy = fma(y, z, -(w x))
    const float zzh = 0.5f - fma_(x, x, y * y);
    float dv2;
    if (TASK == DRONE_TASK_WAYPOINT) {
        const pk::f2 rel = pk::sub(B.V, pk::make_v(wind[0], wind[1]));
        K.V = pk::fma_vlo_nc(u.U0, Z, pk::mul_slo(c.dr, rel));  // fma(aT2, z*, -(drag (v - wind)))
        dv2 = fma_(-P.drag_m, v2 - wind[2], fma_(u.U0.x, zzh, -P.gravity));
    } else {
        K.V = pk::fma_vlo_nc(u.U0, Z, pk::mul_slo(c.dr, B.V));
        dv2 = fma_(-P.drag_m, v2, fma_(u.U0.x, zzh, -P.gravity));
    }
    const float do0 = fma_(-P.gxi, oy * oz, fma_(-P.kdx, ox, u.U0.y));
    K.X = pk::make(dv2, do0);
    const pk::f2 m = pk::mul_swap_hi(B.O, B.X);                // (oz ox, oy ox)
    K.O = pk::fma_ns(c.G, m, pk::fma_ns(c.KD, B.O, u.Utz));    // fma(-g, m, fma(-kd, o, t)) for rows y, z
    const float nq0 = fma_(x, ox, fma_(y, oy, z * oz));       // = -dq0: the sign rides on the consumers' neg modifier
    const float dq1 = fma_(w, ox, fma_(y, oz, -(z * oy)));
    const float dq2 = fma_(w, oy, fma_(z, ox, -(x * oz)));
    const float dq3 = fma_(w, oz, fma_(x, oy, -(y * ox)));
    K.Q = pk::make(dq1, dq2);
    K.W = pk::make(nq0, dq3);
}

template <int TASK>
DRONE_FN void rk4_substep_pk(const KParams& P, Dyn& S, const float (&cmd)[4], const float (&wind)[3], RotorIn& u0, const PkConsts& c) {
    const pk::f2 C01 = pk::make_v(cmd[0], cmd[1]), C23 = pk::make_v(cmd[2], cmd[3]);
    const pk::f2 d01 = pk::sub(pk::make_v(S.r[0], S.r[1]), C01), d23 = pk::sub(pk::make_v(S.r[2], S.r[3]), C23);
    const pk::f2 rh01 = pk::fma_slo(c.eHF, d01, C01), rh23 = pk::fma_slo(c.eHF, d23, C23);  // rotor speeds at t + h/2
    const pk::f2 rf01 = pk::fma_shi(c.eHF, d01, C01), rf23 = pk::fma_shi(c.eHF, d23, C23);  // and at t + h
    RotorPk U;
    U.U0 = pk::make_v(u0.aT2, u0.tx);
    U.Utz = pk::make_v(u0.ty, u0.tz);
    const RotorPk uh = rotor_inputs_pk(P, c, rh01, rh23), uf = rotor_inputs_pk(P, c, rf01, rf23);
    BodyPk B, k, A, acc;
    B.V = pk::make_v(S.v[0], S.v[1]);
    B.X = pk::make_v(S.v[2], S.o[0]);
    B.O = pk::make_v(S.o[1], S.o[2]);
    B.Q = pk::make_v(S.q[1], S.q[2]);
    B.W = pk::make_v(S.q[0], S.q[3]);
    pk::f2 P01 = pk::make_v(S.p[0], S.p[1]);
    // stage 1
    deriv_pk<TASK>(P, c, B, U, wind, k);
    pk::f2 pacc01 = B.V;
    float pacc2 = B.X.x;
    acc = k;
#define DRONE_PK_STAGE_A(HP)                    \
    A.V = pk::fma_slo(HP, k.V, B.V);            \
    A.X = pk::fma_slo(HP, k.X, B.X);            \
    A.O = pk::fma_slo(HP, k.O, B.O);            \
    A.Q = pk::fma_shi(HP, k.Q, B.Q);            \
    A.W = pk::fma_shi_nblo(HP, k.W, B.W);
#define DRONE_PK_ACC2()                         \
    acc.V = pk::fma_slo(c.two, k.V, acc.V);     \
    acc.X = pk::fma_slo(c.two, k.X, acc.X);     \
    acc.O = pk::fma_slo(c.two, k.O, acc.O);     \
    acc.Q = pk::fma_slo(c.two, k.Q, acc.Q);     \
    acc.W = pk::fma_slo(c.two, k.W, acc.W);
    DRONE_PK_STAGE_A(c.hH)
    // stage 2
    deriv_pk<TASK>(P, c, A, uh, wind, k);
    pacc01 = pk::fma_slo(c.two, A.V, pacc01);
    pacc2 = fma_(2.0f, A.X.x, pacc2);
    DRONE_PK_ACC2()
    DRONE_PK_STAGE_A(c.hH)
    // stage 3
    deriv_pk<TASK>(P, c, A, uh, wind, k);
    pacc01 = pk::fma_slo(c.two, A.V, pacc01);
    pacc2 = fma_(2.0f, A.X.x, pacc2);
    DRONE_PK_ACC2()
    DRONE_PK_STAGE_A(c.hF)
    // stage 4
    deriv_pk<TASK>(P, c, A, uf, wind, k);
    P01 = pk::fma_slo(c.hS, pk::add(pacc01, A.V), P01);
    S.p[0] = P01.x;
    S.p[1] = P01.y;
    S.p[2] = fma_(P.h_sixth, pacc2 + A.X.x, S.p[2]);
    const pk::f2 nV = pk::fma_slo(c.hS, pk::add(acc.V, k.V), B.V);
    const pk::f2 nX = pk::fma_slo(c.hS, pk::add(acc.X, k.X), B.X);
    const pk::f2 nO = pk::fma_slo(c.hS, pk::add(acc.O, k.O), B.O);
    const pk::f2 nQ = pk::fma_shi(c.hS, pk::add(acc.Q, k.Q), B.Q);
    const pk::f2 nW = pk::fma_shi_nblo(c.hS, pk::add(acc.W, k.W), B.W);
#undef DRONE_PK_STAGE_A
#undef DRONE_PK_ACC2
    S.v[0] = nV.x; S.v[1] = nV.y; S.v[2] = nX.x;
    S.o[0] = nX.y; S.o[1] = nO.x; S.o[2] = nO.y;
    S.q[1] = nQ.x; S.q[2] = nQ.y; S.q[0] = nW.x; S.q[3] = nW.y;
    S.r[0] = rf01.x; S.r[1] = rf01.y; S.r[2] = rf23.x; S.r[3] = rf23.y;
    u0.aT2 = uf.U0.x; u0.tx = uf.U0.y; u0.ty = uf.Utz.x; u0.tz = uf.Utz.y;
}
#endif  // DRONE_PK_RK4

// one RK4 stage update over the 10 feedback components; the quaternion rows use the hq* steps
#define DRONE_FOR_COMPONENTS(BODY)                                      \
    _Pragma("unroll") for (int i = 0; i < 3; i++) { BODY(v, i, h_) }    \
    _Pragma("unroll") for (int i = 0; i < 4; i++) { BODY(q, i, hq_) }   \
    _Pragma("unroll") for (int i = 0; i < 3; i++) { BODY(o, i, h_) }

// SPEC.md §4: one substep of size h. Rotor speeds relax to cmd exactly (first-order lag, constant command), so the
// rotor inputs are known functions of time: evaluated at t, t + h/2 (stages 2 and 3 share it) and t + h.
// `u0`: rotor inputs at the substep's start (= at S.r) on entry, at its end on return.
template <int TASK>
DRONE_FN void rk4_substep(const KParams& P, Dyn& S, const float (&cmd)[4], const float (&wind)[3], RotorIn& u0) {
    float rh[4], rf[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float d = S.r[i] - cmd[i];
        rh[i] = fma_(P.e_half, d, cmd[i]);
        rf[i] = fma_(P.e_full, d, cmd[i]);
    }
    const RotorIn uh = rotor_inputs(P, rh), uf = rotor_inputs(P, rf);
    Body B, k, A, acc;
#pragma unroll
    for (int i = 0; i < 3; i++) { B.v[i] = S.v[i]; B.o[i] = S.o[i]; }
#pragma unroll
    for (int i = 0; i < 4; i++) B.q[i] = S.q[i];
    float pacc[3];  // dp = v: sum of the stage velocities v1 + 2 v2 + 2 v3 + v4
    const float h_full = P.h, h_half = P.h_half, h_sixth = P.h_sixth;
    const float hq_full = P.hq, hq_half = P.hq_half, hq_sixth = P.hq_sixth;
    deriv<TASK>(P, B, u0, wind, k);
#pragma unroll
    for (int i = 0; i < 3; i++) pacc[i] = B.v[i];
#define STAGE1(f, i, H) acc.f[i] = k.f[i]; A.f[i] = fma_(H##half, k.f[i], B.f[i]);
    DRONE_FOR_COMPONENTS(STAGE1)
    deriv<TASK>(P, A, uh, wind, k);
#pragma unroll
    for (int i = 0; i < 3; i++) pacc[i] = fma_(2.0f, A.v[i], pacc[i]);
#define STAGE2(f, i, H) acc.f[i] = fma_(2.0f, k.f[i], acc.f[i]); A.f[i] = fma_(H##half, k.f[i], B.f[i]);
    DRONE_FOR_COMPONENTS(STAGE2)
    deriv<TASK>(P, A, uh, wind, k);
#pragma unroll
    for (int i = 0; i < 3; i++) pacc[i] = fma_(2.0f, A.v[i], pacc[i]);
#define STAGE3(f, i, H) acc.f[i] = fma_(2.0f, k.f[i], acc.f[i]); A.f[i] = fma_(H##full, k.f[i], B.f[i]);
    DRONE_FOR_COMPONENTS(STAGE3)
    deriv<TASK>(P, A, uf, wind, k);
#pragma unroll
    for (int i = 0; i < 3; i++) S.p[i] = fma_(h_sixth, pacc[i] + A.v[i], S.p[i]);
#define STAGE4(f, i, H) acc.f[i] = acc.f[i] + k.f[i]; S.f[i] = fma_(H##sixth, acc.f[i], B.f[i]);
    DRONE_FOR_COMPONENTS(STAGE4)
#undef STAGE1
#undef STAGE2
#undef STAGE3
#undef STAGE4
#pragma unroll
    for (int i = 0; i < 4; i++) S.r[i] = rf[i];
    u0 = uf;
}

// all substeps of one env step, in the form the caller picked
template <int TASK, bool PK>
DRONE_FN void rk4_run(const KParams& P, Dyn& S, const float (&cmd)[4], const float (&wind)[3], RotorIn& u) {
#if DRONE_PK_RK4
    if constexpr (PK) {
        const PkConsts c = pk_consts(P);  // once per env step, outside the substep loop
        for (uint32_t k = 0; k < P.substeps; k++) rk4_substep_pk<TASK>(P, S, cmd, wind, u, c);
        return;
    }
#endif
    for (uint32_t k = 0; k < P.substeps; k++) rk4_substep<TASK>(P, S, cmd, wind, u);
}

DRONE_FN float target_dist(const Lane& L) {
    const float dx = L.tgt[0] - L.s.p[0], dy = L.tgt[1] - L.s.p[1], dz = L.tgt[2] - L.s.p[2];
    return sqrtf(fma_(dx, dx, fma_(dy, dy, dz * dz)));
}

// SPEC.md §6 from the five reset draws u[0..4] of (env, episode): nine values from their 16-bit halves, low half first
template <int TASK = DRONE_TASK_HOVER, bool CARRY = false>
DRONE_FN void lane_reset_from_draws(const KParams& P, Lane& L, const uint32_t (&u)[5]) {
    float t[3];
#pragma unroll
    for (uint32_t i = 0; i < 3; i++) {
        L.s.p[i] = P.spawn_extent * s16(half16(u, i));
        L.tgt[i] = P.target_extent * s16(half16(u, 3u + i));
        t[i] = P.tilt_init * s16(half16(u, 6u + i));
    }
    // SPEC v5: (1, t) scaled to unit length by two Newton steps of 1/sqrt(n2) about 1 — 5 operations where the correctly
    // rounded sqrt + divide took 28 (two quarter-rate transcendentals among them) on the episode-end path
    const float n2 = fma_(t[0], t[0], fma_(t[1], t[1], fma_(t[2], t[2], 1.0f)));
    const float s1 = fma_(-0.5f, n2, 1.5f);
    const float m = (n2 * s1) * s1;
    const float s2 = fma_(-0.5f, m, 1.5f);
    const float sc = s1 * s2;
    L.s.q[0] = sc;
    L.s.q[1] = t[0] * sc;
    L.s.q[2] = t[1] * sc;
    L.s.q[3] = t[2] * sc;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        L.s.v[i] = 0.0f;
        L.s.o[i] = 0.0f;
        L.wind[i] = 0.0f;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) L.s.r[i] = P.hover_rpm;
    L.tick = 0;
    L.score_count = 0;
    L.ep_return = 0.0f;
    if (TASK == DRONE_TASK_RACE) {  // SPEC.md §11: gate 0 faces the spawn point
        const float e[3] = {L.tgt[0] - L.s.p[0], L.tgt[1] - L.s.p[1], L.tgt[2] - L.s.p[2]};
        unit3(e, L.wind);
    }
    if (CARRY) L.u = rotor_inputs(P, L.s.r);  // the carried rotor inputs follow the fresh rotor speeds
}

// SPEC.md §6. `env` is the global env id.
template <int TASK = DRONE_TASK_HOVER, bool CARRY = false>
DRONE_FN void lane_reset(const KParams& P, Lane& L, uint32_t env) {
    const uint32_t b = rng_base(P.key_reset, env, L.episode);
    uint32_t u[5];
#pragma unroll
    for (uint32_t k = 0; k < 5; k++) u[k] = rng_draw(b, k);
    lane_reset_from_draws<TASK, CARRY>(P, L, u);
}

// SPEC.md §2: the synthetic random policy.
DRONE_FN void random_action(uint32_t key_action, uint32_t env, uint32_t gstep, float (&a)[4]) {
    const uint32_t k = hash32(key_action ^ env);  // invariant over the steps of a fused rollout
    const uint32_t h0 = hash32(k + (2u * gstep) * 0x9E3779B9u), h1 = hash32(k + (2u * gstep + 1u) * 0x9E3779B9u);
    a[0] = s16(h0 & 0xFFFFu);
    a[1] = s16(h0 >> 16);
    a[2] = s16(h1 & 0xFFFFu);
    a[3] = s16(h1 >> 16);
}

// What lane_integrate hands to lane_finish.
struct StepCtx {
    float a2;         // |clamped action|^2
    float prev_dist;  // waypoint / race tasks: distance before integrating
    float p0[3];      // race task: position before integrating
};

// SPEC.md §5 steps 1–4: actions, wind, RK4, renormalise, clamp, tick.
// INRANGE: the caller drew `act` from the SPEC.md section 2 policy itself (random_action: s16 values, all in [-1, 1)), so the
// clamp of step 1 is the identity and is left out — four instructions per step in the kernels that draw their own actions
// (fused rollout, step_many with the in-kernel policy); bit-identical by construction, and pinned on the CPU by
// tests/test_lane_host.py, whose rollout harness compiles this form against the oracle's (clamping) rollout.
template <int TASK, bool CARRY = false, bool PK = DRONE_PK_DEFAULT, bool INRANGE = false>
DRONE_FN void lane_integrate(const KParams& P, Lane& L, const float (&act)[4], uint32_t env, uint32_t gstep, StepCtx& ctx) {
    float a[4], cmd[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        a[i] = INRANGE ? act[i] : clampc(act[i], -1.0f, 1.0f);
        cmd[i] = P.half_max_rpm * (a[i] + 1.0f);
    }
    ctx.a2 = fma_(a[0], a[0], fma_(a[1], a[1], fma_(a[2], a[2], a[3] * a[3])));
    ctx.prev_dist = 0.0f;
    if (TASK == DRONE_TASK_WAYPOINT) {
        const uint32_t b = rng_base(P.key_wind, env, gstep);
#pragma unroll
        for (uint32_t i = 0; i < 3; i++) {
            const uint32_t u = rng_draw(b, i);
            const uint32_t sum = (u & 255u) + ((u >> 8) & 255u) + ((u >> 16) & 255u) + (u >> 24);
            const float xi = (float)((int)sum - 510);
            L.wind[i] = clampc(fma_(P.wind_decay, L.wind[i], P.wind_gain * xi), -P.wind_max, P.wind_max);
        }
        ctx.prev_dist = target_dist(L);
    }
    if (TASK == DRONE_TASK_RACE) {
#pragma unroll
        for (int i = 0; i < 3; i++) ctx.p0[i] = L.s.p[i];
        ctx.prev_dist = target_dist(L);
    }

    if (!CARRY) L.u = rotor_inputs(P, L.s.r);  // state fresh from HBM: nothing carried over from the previous step
    rk4_run<TASK, PK>(P, L.s, cmd, L.wind, L.u);

    {
        float* q = L.s.q;
        const float n2 = fma_(q[0], q[0], fma_(q[1], q[1], fma_(q[2], q[2], q[3] * q[3])));
        const float sc = fma_(-0.5f, n2, 1.5f);  // one Newton step of 1/sqrt(n2) about 1
#pragma unroll
        for (int i = 0; i < 4; i++) q[i] = q[i] * sc;
#pragma unroll
        for (int i = 0; i < 3; i++) L.s.v[i] = clampc(L.s.v[i], -P.max_vel, P.max_vel);
#pragma unroll
        for (int i = 0; i < 3; i++) L.s.o[i] = clampc(L.s.o[i], -P.max_omega, P.max_omega);
        // rotor speeds need no clamp: each stays between its old value and cmd in [0, max_rpm]
    }
    L.tick += 1u;
}

// SPEC.md §10: nearest neighbour among the A agents of a swarm. `other(d, e)`
// yields e = p_j - p_i for the agent d places further round the swarm — lane
// shuffles in the kernels, an array walk in the host test harness.
template <class Other>
DRONE_FN void nearest_neighbour(const KParams& P, Other other, float& nn_d2, float (&nn_e)[3]) {
    nn_d2 = P.nn_far2;
    nn_e[0] = nn_e[1] = nn_e[2] = 0.0f;
    for (uint32_t d = 1; d < P.agents; d++) {
        float e[3];
        other(d, e);
        const float d2 = fma_(e[0], e[0], fma_(e[1], e[1], e[2] * e[2]));
        if (d2 < nn_d2) {
            nn_d2 = d2;
            nn_e[0] = e[0];
            nn_e[1] = e[1];
            nn_e[2] = e[2];
        }
    }
}

// SPEC.md §5 steps 5–9 (§10 steps 6–7 for the swarm task): distance, bounds,
// reward, episode end and reset. `nn_d2` is read only by the swarm task.
// (The reset draws of ended episodes hashed on the scalar unit, one ended lane at a time, was measured and not kept:
// profiles/r06_pruned/.)
template <int TASK, bool CARRY = false>
DRONE_FN void lane_finish(const KParams& P, Lane& L, uint32_t env, const StepCtx& ctx, float nn_d2, StepOut& out) {
    const float dist = target_dist(L);
    bool oob = !(fabsf(L.s.p[0]) <= P.bound) || !(fabsf(L.s.p[1]) <= P.bound) || !(fabsf(L.s.p[2]) <= P.bound);
    if (TASK == DRONE_TASK_SWARM) oob = oob || (nn_d2 < P.coll_r2);  // crash = left the box or collided
    const bool trunc = !oob && L.tick >= P.horizon;

    const float w2 = fma_(L.s.o[0], L.s.o[0], fma_(L.s.o[1], L.s.o[1], L.s.o[2] * L.s.o[2]));
    const float pen = fma_(P.c_omega, w2, P.c_action * ctx.a2);
    float r;
    bool target_changed = false;
    if (TASK == DRONE_TASK_RACE) {
        r = P.progress_scale * (ctx.prev_dist - dist) - pen;
        const float d0[3] = {ctx.p0[0] - L.tgt[0], ctx.p0[1] - L.tgt[1], ctx.p0[2] - L.tgt[2]};
        const float d1[3] = {L.s.p[0] - L.tgt[0], L.s.p[1] - L.tgt[1], L.s.p[2] - L.tgt[2]};
        const float s0 = dot3(L.wind, d0), s1 = dot3(L.wind, d1);
        if (!oob && s0 < 0.0f && s1 >= 0.0f) {  // crossed the gate plane forwards
            const float t = s0 / (s0 - s1);
            float m[3];
#pragma unroll
            for (int i = 0; i < 3; i++) m[i] = fma_(t, L.s.p[i] - ctx.p0[i], ctx.p0[i]) - L.tgt[i];
            if (dot3(m, m) < P.gate_r2) {  // through the ring
                r += P.waypoint_bonus;
                L.score_count += 1u;
                const uint32_t b = rng_base(P.key_waypoint, env, L.episode);
                float cn[3], e[3];
#pragma unroll
                for (uint32_t i = 0; i < 3; i++) {
                    cn[i] = P.target_extent * sym(rng_draw(b, 3u * L.score_count + i));
                    e[i] = cn[i] - L.tgt[i];
                }
                unit3(e, L.wind);
#pragma unroll
                for (int i = 0; i < 3; i++) L.tgt[i] = cn[i];
                target_changed = true;
            }
        }
    } else if (TASK != DRONE_TASK_WAYPOINT) {
        r = fma_(-P.half_inv_bound, dist, 1.0f) - pen;
        if (dist < P.hover_radius) L.score_count += 1u;
        if (TASK == DRONE_TASK_SWARM) r = r - P.c_proximity * __builtin_fmaxf(0.0f, fma_(-nn_d2, P.inv_prox_r2, 1.0f));
    } else {
        r = P.progress_scale * (ctx.prev_dist - dist) - pen;
        if (!oob && dist < P.waypoint_radius) {
            r += P.waypoint_bonus;
            L.score_count += 1u;
            const uint32_t b = rng_base(P.key_waypoint, env, L.episode);
#pragma unroll
            for (uint32_t i = 0; i < 3; i++) L.tgt[i] = P.target_extent * sym(rng_draw(b, 3u * L.score_count + i));
            target_changed = true;
        }
    }
    if (oob) r -= P.crash_penalty;
    L.ep_return += r;

    out.reward = r;
    out.oob = oob;
    out.trunc = trunc;
    out.perf = out.score = out.ep_return = out.ep_len = 0.0f;
    if (oob || trunc) {
        // SPEC v5: hover / swarm log the COUNT of steps within hover_radius (vec_log divides by the steps flown): no division here
        const float score = (float)L.score_count;
        float perf = score;
        if (TASK == DRONE_TASK_WAYPOINT || TASK == DRONE_TASK_RACE) perf = L.score_count >= 8u ? 1.0f : score * 0.125f;
        out.perf = perf;
        out.score = score;
        out.ep_return = L.ep_return;
        out.ep_len = (float)L.tick;
        L.episode += 1u;
        lane_reset<TASK, CARRY>(P, L, env);
        target_changed = true;
    }
    out.target_changed = target_changed;
}

// Single-agent tasks: the whole of SPEC.md §5 steps 1–9.
template <int TASK, bool CARRY = false, bool PK = DRONE_PK_DEFAULT, bool INRANGE = false>
DRONE_FN void lane_step(const KParams& P, Lane& L, const float (&act)[4], uint32_t env, uint32_t gstep, StepOut& out) {
    StepCtx ctx;
    lane_integrate<TASK, CARRY, PK, INRANGE>(P, L, act, env, gstep, ctx);
    lane_finish<TASK, CARRY>(P, L, env, ctx, 0.0f, out);
}

// SPEC.md §7
DRONE_FN void lane_obs(const KParams& P, const Lane& L, float (&o)[DRONE_OBS_DIM_MAX]) {
    const float w = L.s.q[0], x = L.s.q[1], y = L.s.q[2], z = L.s.q[3];
    const float r00 = fma_(-2.0f, fma_(y, y, z * z), 1.0f);
    const float r01 = 2.0f * fma_(x, y, -(w * z));
    const float r02 = 2.0f * fma_(x, z, w * y);
    const float r10 = 2.0f * fma_(x, y, w * z);
    const float r11 = fma_(-2.0f, fma_(x, x, z * z), 1.0f);
    const float r12 = 2.0f * fma_(y, z, -(w * x));
    const float r20 = 2.0f * fma_(x, z, -(w * y));
    const float r21 = 2.0f * fma_(y, z, w * x);
    const float r22 = fma_(-2.0f, fma_(x, x, y * y), 1.0f);
    const float vx = L.s.v[0], vy = L.s.v[1], vz = L.s.v[2];
    o[0] = fma_(r00, vx, fma_(r10, vy, r20 * vz)) * P.inv_max_vel;
    o[1] = fma_(r01, vx, fma_(r11, vy, r21 * vz)) * P.inv_max_vel;
    o[2] = fma_(r02, vx, fma_(r12, vy, r22 * vz)) * P.inv_max_vel;
#pragma unroll
    for (int i = 0; i < 3; i++) o[3 + i] = L.s.o[i] * P.inv_max_omega;
#pragma unroll
    for (int i = 0; i < 4; i++) o[6 + i] = L.s.q[i];
#pragma unroll
    for (int i = 0; i < 4; i++) o[10 + i] = L.s.r[i] * P.inv_max_rpm;
    const float ex = L.tgt[0] - L.s.p[0], ey = L.tgt[1] - L.s.p[1], ez = L.tgt[2] - L.s.p[2];
    o[14] = fma_(r00, ex, fma_(r10, ey, r20 * ez)) * P.half_inv_bound;
    o[15] = fma_(r01, ex, fma_(r11, ey, r21 * ez)) * P.half_inv_bound;
    o[16] = fma_(r02, ex, fma_(r12, ey, r22 * ez)) * P.half_inv_bound;
#pragma unroll
    for (int i = 0; i < 3; i++) o[17 + i] = L.s.p[i] * P.inv_bound;
}

// SPEC.md §11 step 10: gate normal in the body frame, signed distance to the gate plane.
DRONE_FN void lane_obs_gate(const KParams& P, const Lane& L, float (&o)[DRONE_OBS_DIM_MAX]) {
    const float w = L.s.q[0], x = L.s.q[1], y = L.s.q[2], z = L.s.q[3];
    const float r00 = fma_(-2.0f, fma_(y, y, z * z), 1.0f), r01 = 2.0f * fma_(x, y, -(w * z)), r02 = 2.0f * fma_(x, z, w * y);
    const float r10 = 2.0f * fma_(x, y, w * z), r11 = fma_(-2.0f, fma_(x, x, z * z), 1.0f), r12 = 2.0f * fma_(y, z, -(w * x));
    const float r20 = 2.0f * fma_(x, z, -(w * y)), r21 = 2.0f * fma_(y, z, w * x), r22 = fma_(-2.0f, fma_(x, x, y * y), 1.0f);
    const float(&n)[3] = L.wind;
    o[20] = fma_(r00, n[0], fma_(r10, n[1], r20 * n[2]));
    o[21] = fma_(r01, n[0], fma_(r11, n[1], r21 * n[2]));
    o[22] = fma_(r02, n[0], fma_(r12, n[1], r22 * n[2]));
    const float d[3] = {L.s.p[0] - L.tgt[0], L.s.p[1] - L.tgt[1], L.s.p[2] - L.tgt[2]};
    o[23] = dot3(n, d) * P.inv_bound;
}

// SPEC.md §10 step 10: the four neighbour observations (rows of 24 floats).
DRONE_FN void lane_obs_neighbour(const KParams& P, const Lane& L, float nn_d2, const float (&e)[3], float (&o)[DRONE_OBS_DIM_MAX]) {
    const float w = L.s.q[0], x = L.s.q[1], y = L.s.q[2], z = L.s.q[3];
    const float r00 = fma_(-2.0f, fma_(y, y, z * z), 1.0f), r01 = 2.0f * fma_(x, y, -(w * z)), r02 = 2.0f * fma_(x, z, w * y);
    const float r10 = 2.0f * fma_(x, y, w * z), r11 = fma_(-2.0f, fma_(x, x, z * z), 1.0f), r12 = 2.0f * fma_(y, z, -(w * x));
    const float r20 = 2.0f * fma_(x, z, -(w * y)), r21 = 2.0f * fma_(y, z, w * x), r22 = fma_(-2.0f, fma_(x, x, y * y), 1.0f);
    o[20] = fma_(r00, e[0], fma_(r10, e[1], r20 * e[2])) * P.half_inv_bound;
    o[21] = fma_(r01, e[0], fma_(r11, e[1], r21 * e[2])) * P.half_inv_bound;
    o[22] = fma_(r02, e[0], fma_(r12, e[1], r22 * e[2])) * P.half_inv_bound;
    o[23] = (nn_d2 * P.inv_bound) * P.inv_bound;
}

}  // namespace drone

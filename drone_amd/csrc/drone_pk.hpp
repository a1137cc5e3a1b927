// drone_pk.hpp — two-wide f32 arithmetic for the RK4 substep (DRONE_PK_RK4=1): hand-placed v_pk_fma_f32 /
// v_pk_mul_f32 / v_pk_add_f32 with explicit half selection (op_sel) and sign (neg) modifiers.
//
// Why by hand: one wave per SIMD (65 536 envs on 1024 SIMDs) issues one VALU instruction per ~4.7 cycles whatever it
// is, and a packed f32 instruction costs 5.2 — two FMAs for the price of 1.1 (profiles/micro_valu_issue.txt). The
// compiler's SLP pass finds such pairs too but pays for them in v_mov shuffles (+23 % at 65 536 envs,
// profiles/r02_ab/ab_slp_*): it does not keep the state in pair layout, and it folds neither "both lanes from the high
// half" nor a per-lane sign into the instruction's modifiers (it emits v_mov / v_pk_add ... 0 for those). Here the
// state lives in a fixed pair layout for the whole substep and every swap / splat / sign is a modifier bit, so no
// lane ever moves.
//
// Numerics: each lane of a packed instruction is the IEEE operation of its scalar twin (same rounding, same
// subnormal handling); a neg modifier is an exact sign flip of an operand. A result computed here is therefore bit
// for bit the result of the scalar expression it replaces, PROVIDED the expression tree is the same — which is how
// rk4_substep_pk (drone_lane.hpp) is written: SPEC.md's order, term by term. The host build (tests/lane_host, g++)
// gets the same functions as plain scalar code, so the CPU tests check the expression trees against the oracle.
//
// Modifier reminder (VOP3P): op_sel[i] = which half of source i feeds the LOW result lane (0 lo, 1 hi);
// op_sel_hi[i] = which half feeds the HIGH result lane; neg_lo[i] / neg_hi[i] negate source i for that lane.
#pragma once

#include "drone_params.hpp"

namespace drone {
namespace pk {

#if defined(__HIP_DEVICE_COMPILE__)
typedef float f2 __attribute__((ext_vector_type(2)));
#define DRONE_PK_ASM 1
#else
struct f2 {
    float x, y;
};
#define DRONE_PK_ASM 0
#endif

DRONE_FN f2 make(float lo, float hi) {
    f2 r;
    r.x = lo;
    r.y = hi;
    return r;
}

// A pair assembled from two per-lane scalars. The empty asm statements make the scalars opaque first: otherwise the
// optimiser fuses the loads of neighbouring fields of the lane's state struct into <2 x float> loads, and that mixed
// vector / scalar view keeps part of the state in scratch memory instead of registers (seen: 72 bytes, re-read and
// re-written inside the substep loop). No instruction is emitted for them.
DRONE_FN f2 make_v(float lo, float hi) {
#if DRONE_PK_ASM
    asm("" : "+v"(lo));
    asm("" : "+v"(hi));
#endif
    return make(lo, hi);
}

#if DRONE_PK_ASM
// V = both sources in VGPR pairs; S = first source in an SGPR pair (one scalar operand per instruction: the gfx9 constant bus)
#define DRONE_PK3(name, op, mods, c0)                                                      \
    __device__ __forceinline__ f2 name(f2 a, f2 b, f2 c) {                                 \
        f2 d;                                                                              \
        asm(op " %0, %1, %2, %3 " mods : "=v"(d) : c0(a), "v"(b), "v"(c));                 \
        return d;                                                                          \
    }
#define DRONE_PK2(name, op, mods, c0)                                                      \
    __device__ __forceinline__ f2 name(f2 a, f2 b) {                                       \
        f2 d;                                                                              \
        asm(op " %0, %1, %2 " mods : "=v"(d) : c0(a), "v"(b));                             \
        return d;                                                                          \
    }
#define DRONE_PK_HOST3(name, lo, hi)
#define DRONE_PK_HOST2(name, lo, hi)
#else
#define DRONE_PK3(name, op, mods, c0)
#define DRONE_PK2(name, op, mods, c0)
#define DRONE_PK_HOST3(name, lo, hi) \
    DRONE_FN f2 name(f2 a, f2 b, f2 c) { return make(lo, hi); }
#define DRONE_PK_HOST2(name, lo, hi) \
    DRONE_FN f2 name(f2 a, f2 b) { return make(lo, hi); }
#endif
#define FMA_ __builtin_fmaf

// ---- plain lane-wise operations ----
DRONE_PK3(fma, "v_pk_fma_f32", "", "v")
DRONE_PK_HOST3(fma, FMA_(a.x, b.x, c.x), FMA_(a.y, b.y, c.y))
DRONE_PK2(mul, "v_pk_mul_f32", "", "v")
DRONE_PK_HOST2(mul, a.x * b.x, a.y * b.y)
DRONE_PK2(add, "v_pk_add_f32", "", "v")
DRONE_PK_HOST2(add, a.x + b.x, a.y + b.y)
DRONE_PK2(sub, "v_pk_add_f32", "neg_lo:[0,1] neg_hi:[0,1]", "v")  // a - b
DRONE_PK_HOST2(sub, a.x - b.x, a.y - b.y)

// ---- a scalar held in one half of an SGPR pair, applied to both lanes ----
// fma(s.lo, b, c) / fma(s.hi, b, c): the RK4 stage updates (h for velocity / rate rows, hq for quaternion rows share a pair)
DRONE_PK3(fma_slo, "v_pk_fma_f32", "op_sel:[0,0,0] op_sel_hi:[0,1,1]", "s")
DRONE_PK_HOST3(fma_slo, FMA_(a.x, b.x, c.x), FMA_(a.x, b.y, c.y))
DRONE_PK3(fma_shi, "v_pk_fma_f32", "op_sel:[1,0,0] op_sel_hi:[1,1,1]", "s")
DRONE_PK_HOST3(fma_shi, FMA_(a.y, b.x, c.x), FMA_(a.y, b.y, c.y))
// the same with the LOW lane's b negated: the quaternion pair (q0, q3) carries -k for q0 (see rk4_substep_pk)
DRONE_PK3(fma_shi_nblo, "v_pk_fma_f32", "op_sel:[1,0,0] op_sel_hi:[1,1,1] neg_lo:[0,1,0]", "s")
DRONE_PK_HOST3(fma_shi_nblo, FMA_(a.y, -b.x, c.x), FMA_(a.y, b.y, c.y))
DRONE_PK2(mul_slo, "v_pk_mul_f32", "op_sel:[0,0] op_sel_hi:[0,1]", "s")
DRONE_PK_HOST2(mul_slo, a.x * b.x, a.x * b.y)
// a pair of DIFFERENT scalars (SGPR pair), lane-wise: cy, cz; and negated inside an fma: -(kdy, kdz), -(gyi, gzi)
DRONE_PK2(mul_s, "v_pk_mul_f32", "", "s")
DRONE_PK_HOST2(mul_s, a.x * b.x, a.y * b.y)
DRONE_PK3(fma_ns, "v_pk_fma_f32", "neg_lo:[1,0,0] neg_hi:[1,0,0]", "s")  // fma(-s, b, c) lane-wise
DRONE_PK_HOST3(fma_ns, FMA_(-a.x, b.x, c.x), FMA_(-a.y, b.y, c.y))

// ---- a per-lane scalar held in the LOW half of a VGPR pair, applied to both lanes ----
DRONE_PK3(fma_vlo_nc, "v_pk_fma_f32", "op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]", "v")  // fma(a.lo, b, -c)
DRONE_PK_HOST3(fma_vlo_nc, FMA_(a.x, b.x, -c.x), FMA_(a.x, b.y, -c.y))

// ---- the cross terms of SPEC.md section 4 (q = (w, x, y, z); pairs Q = (x, y), W = (w, z)) ----
// (w y, -(w x)) from W, Q
DRONE_PK2(mul_lo_swap_nhi, "v_pk_mul_f32", "op_sel:[0,1] op_sel_hi:[0,0] neg_hi:[0,1]", "v")
DRONE_PK_HOST2(mul_lo_swap_nhi, a.x * b.y, a.x * -b.x)
// (fma(x, z, c.lo), fma(y, z, c.hi)) from Q, W, c
DRONE_PK3(fma_b_hi, "v_pk_fma_f32", "op_sel:[0,1,0] op_sel_hi:[1,1,1]", "v")
DRONE_PK_HOST3(fma_b_hi, FMA_(a.x, b.y, c.x), FMA_(a.y, b.y, c.y))
// (a.hi b.hi, a.lo b.hi): (oz ox, oy ox) from O = (oy, oz), X = (v2, ox)
DRONE_PK2(mul_swap_hi, "v_pk_mul_f32", "op_sel:[1,1] op_sel_hi:[0,1]", "v")
DRONE_PK_HOST2(mul_swap_hi, a.y * b.y, a.x * b.y)
// rotor torques: (a.hi + b.lo, a.lo + b.lo) and (a.lo + b.hi, a.hi + b.hi)
DRONE_PK2(add_swap_lo, "v_pk_add_f32", "op_sel:[1,0] op_sel_hi:[0,0]", "v")
DRONE_PK_HOST2(add_swap_lo, a.y + b.x, a.x + b.x)
DRONE_PK2(add_same_hi, "v_pk_add_f32", "op_sel:[0,1] op_sel_hi:[1,1]", "v")
DRONE_PK_HOST2(add_same_hi, a.x + b.y, a.y + b.y)

#undef FMA_
#undef DRONE_PK3
#undef DRONE_PK2
#undef DRONE_PK_HOST3
#undef DRONE_PK_HOST2

}  // namespace pk
}  // namespace drone

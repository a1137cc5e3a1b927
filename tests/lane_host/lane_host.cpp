// lane_host.cpp — TEST-ONLY host compile of the kernel's per-lane math
// (drone_amd/csrc/drone_lane.hpp) so that its bit-exactness against the CPU
// oracle can be checked without a GPU. This object is never part of the
// product library; libdrone_hip.so has no CPU path.
#include <cstring>
#include <vector>

#include "../../drone_amd/csrc/drone_lane.hpp"

using namespace drone;

static void row_to_lane(const DroneStateRow& r, Lane& L) {
    memcpy(L.s.p, r.pos, 17 * sizeof(float));
    memcpy(L.tgt, r.target, 12);
    memcpy(L.wind, r.wind, 12);
    L.ep_return = r.ep_return;
    L.tick = r.tick;
    L.episode = r.episode;
    L.score_count = r.score_count;
}
static void lane_to_row(const Lane& L, DroneStateRow& r) {
    memcpy(r.pos, L.s.p, 17 * sizeof(float));
    memcpy(r.target, L.tgt, 12);
    memcpy(r.wind, L.wind, 12);
    r.ep_return = L.ep_return;
    r.tick = L.tick;
    r.episode = L.episode;
    r.score_count = L.score_count;
}

extern "C" {

// the swarm task's cross-lane exchange, done here by walking the lanes of the group
static void group_neighbour(const KParams& P, const std::vector<Lane>& lanes, int i, float& nn_d2, float (&nn_e)[3]) {
    const int A = (int)P.agents, base = i - (i % A);
    nearest_neighbour(P, [&](uint32_t d, float (&e)[3]) {
        const Lane& o = lanes[base + ((i - base + (int)d) % A)];
        for (int c = 0; c < 3; c++) e[c] = o.s.p[c] - lanes[i].s.p[c];
    }, nn_d2, nn_e);
}

static void write_obs(const DroneConfig* cfg, const KParams& P, const std::vector<Lane>& lanes, float* obs) {
    const int od = (cfg->task == DRONE_TASK_SWARM || cfg->task == DRONE_TASK_RACE) ? DRONE_OBS_DIM_MAX : DRONE_OBS_DIM;
    for (size_t i = 0; i < lanes.size(); i++) {
        float o[DRONE_OBS_DIM_MAX];
        lane_obs(P, lanes[i], o);
        if (cfg->task == DRONE_TASK_RACE) lane_obs_gate(P, lanes[i], o);
        if (cfg->task == DRONE_TASK_SWARM) {
            float nn_d2, nn_e[3];
            group_neighbour(P, lanes, (int)i, nn_d2, nn_e);
            lane_obs_neighbour(P, lanes[i], nn_d2, nn_e, o);
        }
        memcpy(obs + i * od, o, sizeof(float) * od);
    }
}

void lane_host_reset(const DroneConfig* cfg, uint64_t seed, DroneStateRow* rows, float* obs, int n) {
    KParams P;
    derive_kparams(*cfg, seed, P);
    std::vector<Lane> lanes(n);
    for (int i = 0; i < n; i++) {
        memset(&rows[i], 0, sizeof(DroneStateRow));
        lanes[i].episode = 0;
        if (cfg->task == DRONE_TASK_RACE) lane_reset<DRONE_TASK_RACE>(P, lanes[i], P.env_offset + (uint32_t)i);
        else lane_reset(P, lanes[i], P.env_offset + (uint32_t)i);
        lane_to_row(lanes[i], rows[i]);
    }
    write_obs(cfg, P, lanes, obs);
}

// One vec step over AoS rows with the kernel's lane code. random_policy != 0:
// actions come from the device policy (and are written back to `actions`).
void lane_host_step(const DroneConfig* cfg, uint64_t seed, uint32_t gstep, DroneStateRow* rows, float* actions, float* obs,
                    float* rew, unsigned char* term, unsigned char* trunc, int n, int random_policy) {
    KParams P;
    derive_kparams(*cfg, seed, P);
    std::vector<Lane> lanes(n);
    std::vector<StepCtx> ctx(n);
    std::vector<StepOut> outs(n);
    for (int i = 0; i < n; i++) {
        row_to_lane(rows[i], lanes[i]);
        const uint32_t env = P.env_offset + (uint32_t)i;
        float act[4];
        if (random_policy) {
            random_action(P.key_action, env, gstep, act);
            memcpy(actions + (size_t)i * 4, act, 16);
        } else {
            memcpy(act, actions + (size_t)i * 4, 16);
        }
        if (cfg->task == DRONE_TASK_HOVER) lane_step<DRONE_TASK_HOVER>(P, lanes[i], act, env, gstep, outs[i]);
        else if (cfg->task == DRONE_TASK_WAYPOINT) lane_step<DRONE_TASK_WAYPOINT>(P, lanes[i], act, env, gstep, outs[i]);
        else if (cfg->task == DRONE_TASK_RACE) lane_step<DRONE_TASK_RACE>(P, lanes[i], act, env, gstep, outs[i]);
        else lane_integrate<DRONE_TASK_SWARM>(P, lanes[i], act, env, gstep, ctx[i]);
    }
    if (cfg->task == DRONE_TASK_SWARM) {
        std::vector<float> nn(n);
        for (int i = 0; i < n; i++) {  // every agent has integrated: neighbours on post-integration positions
            float e[3];
            group_neighbour(P, lanes, i, nn[i], e);
        }
        for (int i = 0; i < n; i++) lane_finish<DRONE_TASK_SWARM>(P, lanes[i], P.env_offset + (uint32_t)i, ctx[i], nn[i], outs[i]);
    }
    for (int i = 0; i < n; i++) {
        const StepOut& out = outs[i];
        lane_to_row(lanes[i], rows[i]);
        if (out.oob || out.trunc) {
            rows[i].perf_sum += out.perf;
            rows[i].score_sum += out.score;
            rows[i].ret_sum += out.ep_return;
            rows[i].len_sum += out.ep_len;
            rows[i].n_sum += 1.0f;
            rows[i].oob_sum += out.oob ? 1.0f : 0.0f;
        }
        rew[i] = out.reward;
        term[i] = out.oob;
        trunc[i] = out.trunc;
    }
    write_obs(cfg, P, lanes, obs);
}

}  // extern "C"

// T steps under the device policy with the lanes kept "in registers" from step to step, the way the fused rollout and
// drone_vec_step_many run them: the rotor inputs are CARRIED across steps (Lane::u, CARRY = true) instead of being
// recomputed from the rotor speeds at the start of every step. Returns the reward sums; rows are updated in place.
template <int TASK>
static void rollout_task(const KParams& P, std::vector<Lane>& lanes, DroneStateRow* rows, float* rsum, uint32_t gstep0, int T) {
    const int n = (int)lanes.size();
    std::vector<StepCtx> ctx(n);
    std::vector<StepOut> outs(n);
    std::vector<float> nn(n);
    for (int i = 0; i < n; i++) lanes[i].u = rotor_inputs(P, lanes[i].s.r);
    for (int t = 0; t < T; t++) {
        for (int i = 0; i < n; i++) {
            const uint32_t env = P.env_offset + (uint32_t)i;
            float act[4];
            random_action(P.key_action, env, gstep0 + (uint32_t)t, act);
            // CARRY and INRANGE as the fused rollout kernel instantiates them (actions from random_action: no clamp)
            if (TASK == DRONE_TASK_SWARM) lane_integrate<TASK, true, DRONE_PK_DEFAULT, true>(P, lanes[i], act, env, gstep0 + (uint32_t)t, ctx[i]);
            else lane_step<TASK, true, DRONE_PK_DEFAULT, true>(P, lanes[i], act, env, gstep0 + (uint32_t)t, outs[i]);
        }
        if (TASK == DRONE_TASK_SWARM) {
            for (int i = 0; i < n; i++) {
                float e[3];
                group_neighbour(P, lanes, i, nn[i], e);
            }
            for (int i = 0; i < n; i++) lane_finish<TASK, true>(P, lanes[i], P.env_offset + (uint32_t)i, ctx[i], nn[i], outs[i]);
        }
        for (int i = 0; i < n; i++) {
            const StepOut& out = outs[i];
            rsum[i] = rsum[i] + out.reward;
            if (out.oob || out.trunc) {
                rows[i].perf_sum += out.perf;
                rows[i].score_sum += out.score;
                rows[i].ret_sum += out.ep_return;
                rows[i].len_sum += out.ep_len;
                rows[i].n_sum += 1.0f;
                rows[i].oob_sum += out.oob ? 1.0f : 0.0f;
            }
        }
    }
}

extern "C" {

void lane_host_rollout(const DroneConfig* cfg, uint64_t seed, uint32_t gstep0, int T, DroneStateRow* rows, float* rsum, int n) {
    KParams P;
    derive_kparams(*cfg, seed, P);
    std::vector<Lane> lanes(n);
    for (int i = 0; i < n; i++) {
        row_to_lane(rows[i], lanes[i]);
        rsum[i] = 0.0f;
    }
    if (cfg->task == DRONE_TASK_HOVER) rollout_task<DRONE_TASK_HOVER>(P, lanes, rows, rsum, gstep0, T);
    else if (cfg->task == DRONE_TASK_WAYPOINT) rollout_task<DRONE_TASK_WAYPOINT>(P, lanes, rows, rsum, gstep0, T);
    else if (cfg->task == DRONE_TASK_RACE) rollout_task<DRONE_TASK_RACE>(P, lanes, rows, rsum, gstep0, T);
    else rollout_task<DRONE_TASK_SWARM>(P, lanes, rows, rsum, gstep0, T);
    for (int i = 0; i < n; i++) lane_to_row(lanes[i], rows[i]);
}

// s16 / sym as the product computes them (one fused multiply-add), for an exhaustive comparison with SPEC.md's literal forms
void lane_host_s16_all(float* out65536) {
    for (uint32_t h = 0; h < 65536u; h++) out65536[h] = s16(h);
}
void lane_host_sym(const uint32_t* u, float* out, int n) {
    for (int i = 0; i < n; i++) out[i] = sym(u[i]);
}

void lane_host_kparams(const DroneConfig* cfg, uint64_t seed, uint32_t* out56) {
    KParams P;
    derive_kparams(*cfg, seed, P);
    memcpy(out56, &P, sizeof(P));
}
}

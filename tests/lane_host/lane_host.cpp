// lane_host.cpp — TEST-ONLY host compile of the kernel's per-lane math
// (drone_amd/csrc/drone_lane.hpp) so that its bit-exactness against the CPU
// oracle can be checked without a GPU. This object is never part of the
// product library; libdrone_hip.so has no CPU path.
#include <cstring>

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

void lane_host_reset(const DroneConfig* cfg, uint64_t seed, DroneStateRow* rows, float* obs, int n) {
    KParams P;
    derive_kparams(*cfg, seed, P);
    for (int i = 0; i < n; i++) {
        Lane L;
        memset(&rows[i], 0, sizeof(DroneStateRow));
        L.episode = 0;
        lane_reset(P, L, P.env_offset + (uint32_t)i);
        lane_to_row(L, rows[i]);
        float o[DRONE_OBS_DIM];
        lane_obs(P, L, o);
        memcpy(obs + (size_t)i * DRONE_OBS_DIM, o, sizeof(o));
    }
}

// One vec step over AoS rows with the kernel's lane code. random_policy != 0:
// actions come from the device policy (and are written back to `actions`).
void lane_host_step(const DroneConfig* cfg, uint64_t seed, uint32_t gstep, DroneStateRow* rows, float* actions, float* obs,
                    float* rew, unsigned char* term, unsigned char* trunc, int n, int random_policy) {
    KParams P;
    derive_kparams(*cfg, seed, P);
    for (int i = 0; i < n; i++) {
        Lane L;
        row_to_lane(rows[i], L);
        const uint32_t env = P.env_offset + (uint32_t)i;
        float act[4];
        if (random_policy) {
            random_action(P.key_action, env, gstep, act);
            memcpy(actions + (size_t)i * 4, act, 16);
        } else {
            memcpy(act, actions + (size_t)i * 4, 16);
        }
        StepOut out;
        if (cfg->task == DRONE_TASK_HOVER) lane_step<DRONE_TASK_HOVER>(P, L, act, env, gstep, out);
        else lane_step<DRONE_TASK_WAYPOINT>(P, L, act, env, gstep, out);
        lane_to_row(L, rows[i]);
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
        float o[DRONE_OBS_DIM];
        lane_obs(P, L, o);
        memcpy(obs + (size_t)i * DRONE_OBS_DIM, o, sizeof(o));
    }
}

void lane_host_kparams(const DroneConfig* cfg, uint64_t seed, uint32_t* out52) {
    KParams P;
    derive_kparams(*cfg, seed, P);
    memcpy(out52, &P, sizeof(P));
}
}

/*
 * drone_oracle_vec.c — vec-level wrapper over the scalar oracle env. TEST
 * INFRASTRUCTURE (see drone_oracle.h header: PARITY UNPINNED, reference has no
 * source — /root/reference/.gitmodules:1-3).
 *
 * Mirrors the C-ABI of include/drone_vec.h one-to-one under the `oracle_`
 * prefix so parity tests can drive both sides with the same calls. The loop
 * `for (i < num_envs) c_step(&envs[i])` is the shape of a PufferLib binding's
 * vec_step (HOT LOOP #1); OpenMP only splits that loop over
 * host cores for the cpu_baseline timing — envs are independent, so results do
 * not depend on the thread count.
 */
#include <stdio.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "drone_oracle.h"

typedef struct OracleVec {
    Drone* envs;
    int num_envs;
    DroneConfig cfg;
    Params par;
    uint32_t keys[4];
    uint32_t gstep;
    int threads;
    float* observations;
    float* actions;
    float* rewards;
    unsigned char* terminals;
    unsigned char* truncations;
} OracleVec;

void oracle_config_default(DroneConfig* c, int task) {
    memset(c, 0, sizeof(*c));
    c->struct_size = (uint32_t)sizeof(DroneConfig);
    c->task = task;
    c->buffer_kind = DRONE_BUFFERS_HOST;
    c->device = 0;
    c->env_offset = 0;
    c->horizon = 1024;
    c->substeps = 1;
    c->compact_done = 0;
    c->agents_per_env = task == DRONE_TASK_SWARM ? 8 : 1;
    c->dt = 0.01f;
    c->mass = 0.027f;
    c->arm = 0.0397f;
    c->ixx = 1.4e-5f;
    c->iyy = 1.4e-5f;
    c->izz = 2.17e-5f;
    c->k_thrust = 3.16e-10f;
    c->k_torque = 7.94e-12f;
    c->k_drag = 0.0027f;
    c->k_ang_damp = 1.0e-6f;
    c->gravity = 9.81f;
    c->max_rpm = 21702.0f;
    c->motor_tau = 0.05f;
    c->max_vel = 20.0f;
    c->max_omega = 50.0f;
    c->bound = 5.0f;
    c->spawn_extent = 3.0f;
    c->target_extent = 3.0f;
    c->tilt_init = 0.1f;
    c->hover_radius = 0.5f;
    c->waypoint_radius = 0.5f;
    c->wind_theta = 0.5f;
    c->wind_sigma = 1.0f;
    c->wind_max = 5.0f;
    c->c_omega = 1.0e-4f;
    c->c_action = 0.01f;
    c->crash_penalty = 1.0f;
    c->progress_scale = 1.0f;
    c->waypoint_bonus = 1.0f;
    c->collision_radius = 0.15f;
    c->proximity_radius = 1.0f;
    c->c_proximity = 0.5f;
    c->gate_radius = 0.75f;
}

int oracle_obs_dim(int task) { return (task == DRONE_TASK_SWARM || task == DRONE_TASK_RACE) ? DRONE_OBS_DIM_MAX : DRONE_OBS_DIM; }

static void set_keys(OracleVec* v, uint64_t seed) {
    for (uint32_t s = 0; s < 4; s++) v->keys[s] = stream_key(seed, s);
}

OracleVec* oracle_vec_init(float* observations, float* actions, float* rewards, unsigned char* terminals,
                           unsigned char* truncations, int num_envs, uint64_t seed, const DroneConfig* cfg) {
    if (!cfg || cfg->struct_size != sizeof(DroneConfig) || num_envs <= 0) return NULL;
    if (cfg->substeps < 1 || cfg->horizon < 1) return NULL;
    if (cfg->task == DRONE_TASK_WAYPOINT && !(cfg->wind_max > 0.0f)) return NULL; /* clamp bounds are never zero (SPEC.md §4) */
    if (cfg->task == DRONE_TASK_SWARM) {
        const int A = cfg->agents_per_env;
        if (A < 1 || A > 64 || (A & (A - 1)) || num_envs % A || cfg->env_offset % (uint32_t)A) return NULL;
    }
    OracleVec* v = (OracleVec*)calloc(1, sizeof(OracleVec));
    v->envs = (Drone*)calloc((size_t)num_envs, sizeof(Drone));
    v->num_envs = num_envs;
    v->cfg = *cfg;
    params_derive(&v->cfg, &v->par);
    set_keys(v, seed);
    v->gstep = 0;
    v->threads = 1;
    v->observations = observations;
    v->actions = actions;
    v->rewards = rewards;
    v->terminals = terminals;
    v->truncations = truncations;
    for (int i = 0; i < num_envs; i++) {
        Drone* e = &v->envs[i];
        e->observations = observations + (size_t)i * (size_t)oracle_obs_dim(cfg->task);
        e->actions = actions + (size_t)i * DRONE_ACT_DIM;
        e->rewards = rewards + i;
        e->terminals = terminals + i;
        e->truncations = truncations + i;
        e->env_id = cfg->env_offset + (uint32_t)i;
        e->cfg = &v->cfg;
        e->par = &v->par;
        e->keys = v->keys;
        e->gstep = &v->gstep;
        init(e);
    }
    return v;
}

void oracle_set_threads(OracleVec* v, int threads) { v->threads = threads < 1 ? 1 : threads; }

void oracle_vec_reset(OracleVec* v, uint64_t seed) {
    set_keys(v, seed);
    v->gstep = 0;
    if (v->cfg.task == DRONE_TASK_SWARM) {
        const int A = v->cfg.agents_per_env;
        for (int g = 0; g < v->num_envs / A; g++) c_reset_swarm(&v->envs[(size_t)g * A], A);
    } else {
        for (int i = 0; i < v->num_envs; i++) c_reset(&v->envs[i]);
    }
}

void oracle_vec_step(OracleVec* v) {
    const int n = v->num_envs;
    if (v->cfg.task == DRONE_TASK_SWARM) {
        const int A = v->cfg.agents_per_env, groups = n / A;
#pragma omp parallel for schedule(static) num_threads(v->threads) if (v->threads > 1)
        for (int g = 0; g < groups; g++) c_step_swarm(&v->envs[(size_t)g * A], A);
    } else {
#pragma omp parallel for schedule(static) num_threads(v->threads) if (v->threads > 1)
        for (int i = 0; i < n; i++) c_step(&v->envs[i]);
    }
    v->gstep += 1;
}

int oracle_vec_fill_random_actions(OracleVec* v, float* actions, uint32_t gstep) {
    const int n = v->num_envs;
#pragma omp parallel for schedule(static) num_threads(v->threads) if (v->threads > 1)
    for (int i = 0; i < n; i++)
        random_action(v->keys[STREAM_ACTION], v->cfg.env_offset + (uint32_t)i, gstep, actions + (size_t)i * 4);
    return 0;
}

/* SPEC.md §9, stated the slow way: T × { random actions; step }, reducing the
 * per-step outputs as the fused kernel defines them. Uses a private action
 * scratch so the caller's `actions` buffer is untouched. */
void oracle_vec_rollout(OracleVec* v, int horizon) {
    const int n = v->num_envs;
    float* rsum = (float*)calloc((size_t)n, sizeof(float));
    unsigned char* tany = (unsigned char*)calloc((size_t)n, 1);
    unsigned char* uany = (unsigned char*)calloc((size_t)n, 1);
    float* scratch = (float*)malloc((size_t)n * 4 * sizeof(float));
    for (int i = 0; i < n; i++) v->envs[i].actions = scratch + (size_t)i * 4;
    for (int t = 0; t < horizon; t++) {
        oracle_vec_fill_random_actions(v, scratch, v->gstep);
        oracle_vec_step(v);
        for (int i = 0; i < n; i++) {
            rsum[i] = rsum[i] + v->rewards[i];
            tany[i] |= v->terminals[i];
            uany[i] |= v->truncations[i];
        }
    }
    for (int i = 0; i < n; i++) {
        v->envs[i].actions = v->actions + (size_t)i * 4;
        v->rewards[i] = rsum[i];
        v->terminals[i] = tany[i];
        v->truncations[i] = uany[i];
    }
    free(rsum);
    free(tany);
    free(uany);
    free(scratch);
}

void oracle_vec_log(OracleVec* v, DroneLog* out) {
    double perf = 0, score = 0, ret = 0, len = 0, n = 0, oob = 0;
    for (int i = 0; i < v->num_envs; i++) {
        Log* l = &v->envs[i].log;
        perf += l->perf;
        score += l->score;
        ret += l->episode_return;
        len += l->episode_length;
        n += l->n;
        oob += l->oob;
        memset(l, 0, sizeof(Log));
    }
    memset(out, 0, sizeof(*out));
    if (n > 0) {
        /* SPEC.md §8 (v5): hover / swarm report the share of logged steps within hover_radius */
        const int per_step = v->cfg.task == DRONE_TASK_HOVER || v->cfg.task == DRONE_TASK_SWARM;
        out->perf = (float)(perf / (per_step ? len : n));
        out->score = (float)(score / (per_step ? len : n));
        out->episode_return = (float)(ret / n);
        out->episode_length = (float)(len / n);
        out->oob = (float)(oob / n);
    }
    out->n = (float)n;
}

void oracle_vec_close(OracleVec* v) {
    if (!v) return;
    free(v->envs);
    free(v);
}

uint32_t oracle_vec_gstep(const OracleVec* v) { return v->gstep; }
int oracle_vec_num_envs(const OracleVec* v) { return v->num_envs; }

int oracle_vec_get_state(OracleVec* v, DroneStateRow* rows, int first, int count) {
    if (first < 0 || count < 0 || first + count > v->num_envs) return -1;
    for (int i = 0; i < count; i++) {
        const Drone* e = &v->envs[first + i];
        DroneStateRow* r = &rows[i];
        memcpy(r->pos, e->s.pos, sizeof(State)); /* pos vel quat omega rpm are contiguous in both */
        memcpy(r->target, e->target, 12);
        memcpy(r->wind, v->cfg.task == DRONE_TASK_RACE ? e->gate_n : e->wind, 12); /* the aux slots */
        r->ep_return = e->ep_return;
        r->tick = e->tick;
        r->episode = e->episode;
        r->score_count = e->score_count;
        r->perf_sum = e->log.perf;
        r->score_sum = e->log.score;
        r->ret_sum = e->log.episode_return;
        r->len_sum = e->log.episode_length;
        r->n_sum = e->log.n;
        r->oob_sum = e->log.oob;
    }
    return 0;
}

int oracle_vec_set_state(OracleVec* v, const DroneStateRow* rows, int first, int count) {
    if (first < 0 || count < 0 || first + count > v->num_envs) return -1;
    for (int i = 0; i < count; i++) {
        Drone* e = &v->envs[first + i];
        const DroneStateRow* r = &rows[i];
        memcpy(e->s.pos, r->pos, sizeof(State));
        memcpy(e->target, r->target, 12);
        if (v->cfg.task == DRONE_TASK_RACE) memcpy(e->gate_n, r->wind, 12);
        else memcpy(e->wind, r->wind, 12);
        e->ep_return = r->ep_return;
        e->tick = r->tick;
        e->episode = r->episode;
        e->score_count = r->score_count;
        e->log.perf = r->perf_sum;
        e->log.score = r->score_sum;
        e->log.episode_return = r->ret_sum;
        e->log.episode_length = r->len_sum;
        e->log.n = r->n_sum;
        e->log.oob = r->oob_sum;
    }
    return 0;
}

/* Exposed for known-answer tests of the integer RNG and parameter derivation. */
uint32_t oracle_hash32(uint32_t x) { return hash32(x); }
uint32_t oracle_stream_key(uint64_t seed, uint32_t stream) { return stream_key(seed, stream); }
uint32_t oracle_rng_draw(uint32_t key, uint32_t env, uint32_t ctr, uint32_t d) {
    return rng_draw(rng_base(key, env, ctr), d);
}
void oracle_params_derive(const DroneConfig* c, float* out32) {
    Params p;
    params_derive(c, &p);
    memcpy(out32, &p, sizeof(Params));
}
int oracle_omp_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

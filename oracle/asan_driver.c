/* Sanitizer smoke driver for the oracle (CPU only; test infrastructure). */
#include <stdio.h>
#include <stdlib.h>
#include "../include/drone_vec.h"
typedef struct OracleVec OracleVec;
void oracle_config_default(DroneConfig*, int);
OracleVec* oracle_vec_init(float*, float*, float*, unsigned char*, unsigned char*, int, uint64_t, const DroneConfig*);
void oracle_vec_reset(OracleVec*, uint64_t);
void oracle_vec_step(OracleVec*);
void oracle_vec_rollout(OracleVec*, int);
int oracle_vec_fill_random_actions(OracleVec*, float*, uint32_t);
void oracle_vec_log(OracleVec*, DroneLog*);
void oracle_vec_close(OracleVec*);
uint32_t oracle_vec_gstep(const OracleVec*);
int main(void) {
    for (int task = 0; task < 4; task++) {
        const int n = task == 2 ? 256 : 257;
        DroneConfig c;
        oracle_config_default(&c, task);
        c.horizon = 64;
        c.collision_radius = 0.6f;
        c.gate_radius = 3.0f;
        float* obs = malloc(sizeof(float) * n * (task >= 2 ? DRONE_OBS_DIM_MAX : DRONE_OBS_DIM));
        float* act = malloc(sizeof(float) * n * DRONE_ACT_DIM);
        float* rew = malloc(sizeof(float) * n);
        unsigned char* term = malloc(n);
        unsigned char* trunc = malloc(n);
        OracleVec* v = oracle_vec_init(obs, act, rew, term, trunc, n, 7, &c);
        oracle_vec_reset(v, 7);
        for (int t = 0; t < 300; t++) {
            oracle_vec_fill_random_actions(v, act, oracle_vec_gstep(v));
            oracle_vec_step(v);
        }
        oracle_vec_rollout(v, 33);
        DroneLog l;
        oracle_vec_log(v, &l);
        printf("task %d: n=%g ret=%g len=%g oob=%g score=%g\n", task, l.n, l.episode_return, l.episode_length, l.oob, l.score);
        oracle_vec_close(v);
        free(obs); free(act); free(rew); free(term); free(trunc);
    }
    return 0;
}

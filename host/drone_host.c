/*
 * drone_host.c — plain-C host of the drone env, the way a C vec-env worker
 * would drive it: caller-owned host buffers, init / reset / step / log / close
 * through include/drone_vec.h, nothing else. (BASELINE.json north_star: "host
 * side stays C calling HIP through a thin C-ABI".) No HIP headers here: the
 * device is entirely behind the C-ABI.
 *
 *   drone_host [--envs N] [--steps K] [--task 0..3] [--rollout T] [--seed S] [--crc 1] [--many K] [--fill 0|1] [--heap 0|1]
 *
 * Prints env-steps/s for (a) per-step calls with host buffers — every step
 * pays H2D actions + D2H observations/rewards/flags over PCIe — and (b) the
 * fused rollout, which crosses PCIe once per T steps.
 * With --crc 1 it instead runs K random-policy steps from reset and prints a
 * CRC-32 (zlib polynomial) chained over every step's observations, rewards,
 * terminals and truncations: tests/test_c_host.py compares it with the CPU
 * oracle's, so this pure-C caller is parity-checked too.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/mman.h>
#include <string.h>
#include <time.h>

#include "drone_vec.h"

/* A zeroed buffer that owns its pages (page-aligned, padded to whole pages): what DroneConfig.host_pages_exclusive
 * vouches for, so that the library may pin it for the zero-copy transport. */
static void* page_alloc(size_t bytes) {
    /* a mapping of its own (round 5: a posix_memalign block owns its pages but lies inside the malloc heap, and registered heap
     * pages fault when the heap around them is trimmed while the GPU writes them: tools/debug/heap_interior_registration_stress.py) */
    const size_t span = bytes ? (bytes + 4095) / 4096 * 4096 : 4096;
    void* p = mmap(NULL, span, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) return NULL;
    memset(p, 0, span);
    return p;
}

/* CRC-32 (IEEE 802.3, reflected, as zlib.crc32): crc of `buf` continuing from `crc` */
static uint32_t crc32_update(uint32_t crc, const void* buf, size_t len) {
    static uint32_t table[256];
    static int ready = 0;
    if (!ready) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        ready = 1;
    }
    const unsigned char* p = (const unsigned char*)buf;
    crc = ~crc;
    for (size_t i = 0; i < len; i++) crc = table[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
    return ~crc;
}

/* The opposite: a zeroed buffer that does NOT own its pages (64 bytes into a malloc block), like a vec-env worker's
 * unaligned slice of a shared block: the library may not pin it (--heap 1). Never freed: the process is short-lived. */
static void* heap_alloc(size_t bytes) {
    char* p = (char*)calloc(bytes + 128, 1);
    return p ? p + 64 : NULL;
}

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int main(int argc, char** argv) {
    int envs = 65536, steps = 1000, task = DRONE_TASK_HOVER, rollout = 128, crc_mode = 0, many = 0, fill = 1, heap = 0;
    unsigned long long seed = 0;
    for (int i = 1; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "--envs")) envs = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--steps")) steps = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--task")) task = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--rollout")) rollout = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--seed")) seed = strtoull(argv[i + 1], NULL, 10);
        else if (!strcmp(argv[i], "--crc")) crc_mode = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--fill")) fill = atoi(argv[i + 1]); /* 0: time the env step alone (actions stay as first drawn); 1: a fresh random action batch per step, itself a device round trip */
        else if (!strcmp(argv[i], "--heap")) heap = atoi(argv[i + 1]); /* 1: buffers that share their pages with the heap (never pinned by the library) */
        else if (!strcmp(argv[i], "--many")) many = atoi(argv[i + 1]); /* K > 0: step through drone_vec_step_many, K env steps per call */
        else { fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
    }
    if (task != DRONE_TASK_HOVER && task != DRONE_TASK_WAYPOINT && task != DRONE_TASK_SWARM && task != DRONE_TASK_RACE) {
        fprintf(stderr, "unknown task %d (0 hover, 1 waypoint, 2 swarm, 3 race)\n", task);
        return 2;
    }
    if (many < 0 || many > 4096) { fprintf(stderr, "--many must be in [0, 4096]\n"); return 2; }
    if (envs <= 0 || steps <= 0 || rollout <= 0) { fprintf(stderr, "--envs, --steps and --rollout must be positive\n"); return 2; }
    const size_t obs_dim = (size_t)drone_obs_dim(task); /* 20, or 24 for the swarm and race tasks */
    void* (*const alloc)(size_t) = heap ? heap_alloc : page_alloc;
    float* obs = (float*)alloc(sizeof(float) * (size_t)envs * obs_dim);
    float* act = (float*)alloc(sizeof(float) * (size_t)envs * DRONE_ACT_DIM);
    float* rew = (float*)alloc(sizeof(float) * (size_t)envs);
    unsigned char* term = (unsigned char*)alloc((size_t)envs);
    unsigned char* trunc = (unsigned char*)alloc((size_t)envs);
    if (!obs || !act || !rew || !term || !trunc) { fprintf(stderr, "out of memory\n"); return 1; }

    DroneConfig cfg;
    drone_config_default(&cfg, task);
    cfg.buffer_kind = DRONE_BUFFERS_HOST;
    cfg.host_pages_exclusive = heap ? 0 : 1; /* page_alloc: every buffer owns its pages */
    DroneVec* v = drone_vec_init(obs, act, rew, term, trunc, envs, seed, &cfg);
    if (!v) { fprintf(stderr, "drone_vec_init failed: %s\n", drone_last_error()); return 1; }
    drone_vec_reset(v, seed);
    if (drone_vec_status(v)) { fprintf(stderr, "drone_vec_reset failed: %s\n", drone_vec_status_message(v)); return 1; }

    /* K-major blocks for drone_vec_step_many: actions [K][N][4] in, observations [K][N][O] / rewards / flags [K][N] out */
    float *m_act = NULL, *m_obs = NULL, *m_rew = NULL;
    unsigned char *m_term = NULL, *m_trunc = NULL;
    if (many > 0) {
        const size_t kn = (size_t)many * (size_t)envs;
        m_act = (float*)page_alloc(sizeof(float) * kn * DRONE_ACT_DIM);
        m_obs = (float*)page_alloc(sizeof(float) * kn * obs_dim);
        m_rew = (float*)page_alloc(sizeof(float) * kn);
        m_term = (unsigned char*)page_alloc(kn);
        m_trunc = (unsigned char*)page_alloc(kn);
        if (!m_act || !m_obs || !m_rew || !m_term || !m_trunc) { fprintf(stderr, "out of memory\n"); return 1; }
        /* page_alloc blocks own their pages: pinned, the kernel reads / writes them in place (DRONE_HOST_PIN_BLOCKS=0: staging + copies) */
        const char* pb = getenv("DRONE_HOST_PIN_BLOCKS");
        if (!(pb && pb[0] == '0') &&
            (drone_vec_host_pin(v, m_act, sizeof(float) * kn * DRONE_ACT_DIM, 1) || drone_vec_host_pin(v, m_obs, sizeof(float) * kn * obs_dim, 1) ||
             drone_vec_host_pin(v, m_rew, sizeof(float) * kn, 1) || drone_vec_host_pin(v, m_term, kn, 1) || drone_vec_host_pin(v, m_trunc, kn, 1))) {
            fprintf(stderr, "drone_vec_host_pin failed: %s\n", drone_last_error());
            return 1;
        }
    }

    if (crc_mode) {
        uint32_t crc = 0;
        crc = crc32_update(crc, obs, sizeof(float) * (size_t)envs * obs_dim); /* the reset observations */
        for (int t = 0; many > 0 && t < steps;) { /* the same CRC chain through K steps per call: step by step, obs / rew / term / trunc */
            const int k_now = steps - t < many ? steps - t : many;
            for (int k = 0; k < k_now; k++) drone_vec_fill_random_actions(v, m_act + (size_t)k * (size_t)envs * DRONE_ACT_DIM, drone_vec_gstep(v) + (uint32_t)k);
            drone_vec_step_many(v, k_now, m_act, m_obs, m_rew, m_term, m_trunc);
            for (int k = 0; k < k_now; k++) {
                const size_t r0 = (size_t)k * (size_t)envs;
                crc = crc32_update(crc, m_obs + r0 * obs_dim, sizeof(float) * (size_t)envs * obs_dim);
                crc = crc32_update(crc, m_rew + r0, sizeof(float) * (size_t)envs);
                crc = crc32_update(crc, m_term + r0, (size_t)envs);
                crc = crc32_update(crc, m_trunc + r0, (size_t)envs);
            }
            t += k_now;
        }
        for (int t = 0; many == 0 && t < steps; t++) {
            drone_vec_fill_random_actions(v, act, drone_vec_gstep(v));
            drone_vec_step(v);
            crc = crc32_update(crc, obs, sizeof(float) * (size_t)envs * obs_dim);
            crc = crc32_update(crc, rew, sizeof(float) * (size_t)envs);
            crc = crc32_update(crc, term, (size_t)envs);
            crc = crc32_update(crc, trunc, (size_t)envs);
        }
        if (drone_vec_status(v)) { fprintf(stderr, "step failed: %s\n", drone_vec_status_message(v)); return 1; }
        DroneLog lg;
        drone_vec_log(v, &lg);
        printf("{\"mode\": \"crc\", \"task\": %d, \"envs\": %d, \"steps\": %d, \"steps_per_call\": %d, \"crc32\": %u, \"episodes\": %.0f}\n", task, envs, steps, many > 0 ? many : 1, crc, lg.n);
        if (many > 0) { drone_vec_host_unpin(v, m_act); drone_vec_host_unpin(v, m_obs); drone_vec_host_unpin(v, m_rew); drone_vec_host_unpin(v, m_term); drone_vec_host_unpin(v, m_trunc); }
        drone_vec_close(v);
        /* (the buffers are mappings of their own or never-freed heap blocks: the process ends here) */
        return 0;
    }

    /* (a) per-step, host buffers: the random policy stands in for the caller's policy */
    for (int t = 0; t < 10; t++) { drone_vec_fill_random_actions(v, act, drone_vec_gstep(v)); drone_vec_step(v); }
    double t0 = now_s();
    long dones = 0;
    for (int t = 0; t < steps; t++) {
        if (fill) drone_vec_fill_random_actions(v, act, drone_vec_gstep(v));
        drone_vec_step(v);
        for (int i = 0; i < envs; i += 4096) dones += term[i] | trunc[i]; /* touch the outputs */
    }
    double el = now_s() - t0;
    printf("{\"mode\": \"per-step host buffers (PCIe inclusive)\", \"envs\": %d, \"steps\": %d, \"fresh_actions_per_step\": %d, \"transport\": %d, \"env_steps_per_s\": %.4g, \"ms_per_step\": %.4f}\n",
           envs, steps, fill, drone_vec_host_transport(v), (double)envs * steps / el, el * 1e3 / steps);

    /* (a') K env steps per call with every step's outputs (drone_vec_step_many): one launch and one round of copies per K steps */
    if (many > 0) {
        for (int k = 0; k < many; k++) drone_vec_fill_random_actions(v, m_act + (size_t)k * (size_t)envs * DRONE_ACT_DIM, drone_vec_gstep(v) + (uint32_t)k);
        drone_vec_step_many(v, many, m_act, m_obs, m_rew, m_term, m_trunc);
        const int calls = steps / many > 0 ? steps / many : 1;
        t0 = now_s();
        for (int c = 0; c < calls; c++) drone_vec_step_many(v, many, m_act, m_obs, m_rew, m_term, m_trunc);
        el = now_s() - t0;
        printf("{\"mode\": \"step_many host blocks (PCIe inclusive)\", \"envs\": %d, \"steps_per_call\": %d, \"calls\": %d, \"env_steps_per_s\": %.4g, \"ms_per_env_step\": %.4f}\n",
               envs, many, calls, (double)envs * many * calls / el, el * 1e3 / ((double)calls * many));
    }

    /* (b) fused rollout: PCIe once per horizon */
    drone_vec_rollout(v, rollout);
    int reps = steps / rollout > 0 ? steps / rollout : 1;
    t0 = now_s();
    for (int r = 0; r < reps; r++) drone_vec_rollout(v, rollout);
    el = now_s() - t0;
    printf("{\"mode\": \"fused rollout, host buffers at the horizon\", \"envs\": %d, \"horizon\": %d, \"launches\": %d, \"env_steps_per_s\": %.4g, \"ms_per_launch\": %.4f}\n",
           envs, rollout, reps, (double)envs * rollout * reps / el, el * 1e3 / reps);

    DroneLog log;
    drone_vec_log(v, &log);
    printf("{\"log\": {\"n\": %.0f, \"episode_return\": %.5g, \"episode_length\": %.5g, \"score\": %.5g, \"oob\": %.5g}, \"sampled_dones\": %ld}\n",
           log.n, log.episode_return, log.episode_length, log.score, log.oob, dones);
    if (drone_vec_status(v)) { fprintf(stderr, "a call on the handle failed: %s\n", drone_vec_status_message(v)); return 1; }
    if (many > 0) { drone_vec_host_unpin(v, m_act); drone_vec_host_unpin(v, m_obs); drone_vec_host_unpin(v, m_rew); drone_vec_host_unpin(v, m_term); drone_vec_host_unpin(v, m_trunc); }
    drone_vec_close(v);
    /* (the buffers are mappings of their own or never-freed heap blocks: the process ends here) */
    return 0;
}

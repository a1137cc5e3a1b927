// Sanitizer driver (TEST ONLY, CPU): the kernel's per-lane math (drone_amd/csrc/drone_lane.hpp, host-compiled through
// lane_host.cpp) under AddressSanitizer + UBSan for all four tasks, including NaN / huge / denormal states and actions.
// GPU sanitizers are unavailable on this pool; this is where out-of-range indexing, signed overflow or invalid shifts
// in the lane code would show.   g++ -fsanitize=address,undefined lane_host.cpp san_driver.cpp
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/drone_vec.h"

extern "C" {
void lane_host_reset(const DroneConfig* cfg, uint64_t seed, DroneStateRow* rows, float* obs, int n);
void lane_host_step(const DroneConfig* cfg, uint64_t seed, uint32_t gstep, DroneStateRow* rows, float* actions, float* obs, float* rew,
                    unsigned char* term, unsigned char* trunc, int n, int random_policy);
}

// default config without the library (this driver links no HIP): the fields the lane code reads
static void config(DroneConfig& c, int task) {
    memset(&c, 0, sizeof(c));
    c.struct_size = sizeof(c); c.task = task; c.horizon = 40; c.substeps = task == 1 ? 3 : 1; c.agents_per_env = task == 2 ? 8 : 1;
    c.dt = 0.01f; c.mass = 0.027f; c.arm = 0.0397f; c.ixx = 1.4e-5f; c.iyy = 1.4e-5f; c.izz = 2.17e-5f; c.k_thrust = 3.16e-10f;
    c.k_torque = 7.94e-12f; c.k_drag = 0.0027f; c.k_ang_damp = 1e-6f; c.gravity = 9.81f; c.max_rpm = 21702.0f; c.motor_tau = 0.05f;
    c.max_vel = 20.0f; c.max_omega = 50.0f; c.bound = 5.0f; c.spawn_extent = 3.0f; c.target_extent = 3.0f; c.tilt_init = 0.1f;
    c.hover_radius = 0.5f; c.waypoint_radius = 1.5f; c.wind_theta = 0.5f; c.wind_sigma = 1.0f; c.wind_max = 5.0f; c.c_omega = 1e-4f;
    c.c_action = 0.01f; c.crash_penalty = 1.0f; c.progress_scale = 1.0f; c.waypoint_bonus = 1.0f; c.collision_radius = 0.6f;
    c.proximity_radius = 1.0f; c.c_proximity = 0.5f; c.gate_radius = 3.0f; c.env_offset = 0xFFFFFF00u;  // global ids wrap around 2^32
}

int main() {
    const int n = 256;
    for (int task = 0; task < 4; task++) {
        DroneConfig c;
        config(c, task);
        const int od = task >= 2 ? DRONE_OBS_DIM_MAX : DRONE_OBS_DIM;
        std::vector<DroneStateRow> rows(n);
        std::vector<float> obs((size_t)n * od), act((size_t)n * 4), rew(n);
        std::vector<unsigned char> term(n), trunc(n);
        lane_host_reset(&c, 0xFEEDFACECAFEBEEFull, rows.data(), obs.data(), n);
        long ends = 0;
        for (uint32_t t = 0; t < 300; t++) {
            const uint32_t gstep = 0xFFFFFF80u + t;  // the step counter wraps too
            if (t == 100) {                          // hostile states: NaN, infinities, denormals, huge values
                const float bad[8] = {NAN, INFINITY, -INFINITY, 1e-42f, -1e-42f, 3e38f, -3e38f, 0.0f};
                for (int i = 0; i < n; i++) {
                    rows[i].pos[i % 3] = bad[i % 8];
                    rows[i].vel[(i + 1) % 3] = bad[(i + 3) % 8];
                    rows[i].quat[i % 4] = bad[(i + 5) % 8];
                    rows[i].rpm[i % 4] = bad[(i + 2) % 8];
                    rows[i].tick = (i % 5 == 0) ? 0xFFFFFFFFu : rows[i].tick;
                }
            }
            lane_host_step(&c, 0xFEEDFACECAFEBEEFull, gstep, rows.data(), act.data(), obs.data(), rew.data(), term.data(), trunc.data(), n, t % 2);
            if (t % 2 == 1) for (int i = 0; i < n * 4; i++) act[i] = (i % 7 == 0) ? NAN : (i % 11 == 0 ? 1e30f : act[i]);  // hostile actions for the next step
            for (int i = 0; i < n; i++) ends += term[i] | trunc[i];
        }
        printf("task %d: %ld episode ends, obs[0]=%g\n", task, ends, obs[0]);
        if (ends == 0) return 1;
    }
    return 0;
}

// Small shards from C++ (no Python in the launch path): one handle on one stream vs the same envs split over H
// handles with their own HIP streams — launched round-robin from one thread, or each from its own thread — to see
// whether independent streams hide the dependent-launch boundary that dominates a 65 536-env step.
//   hipcc -O2 -I include tools/micro/two_stream_host.cpp -L drone_amd -l:libdrone_hip.so -Wl,-rpath,$PWD/drone_amd -lpthread -o /tmp/two_stream_host
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "drone_vec.h"

struct Half {
    DroneVec* v;
    hipStream_t s;
    float *obs, *act, *rew;
    unsigned char *term, *trunc;
};

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const int steps = 4000;
    for (int total : {65536, 131072, 262144}) {
        for (int H : {1, 2, 4}) {
            for (int threaded = 0; threaded <= (H > 1 ? 1 : 0); threaded++) {
                const int n = total / H;
                std::vector<Half> hs(H);
                for (int k = 0; k < H; k++) {
                    Half& h = hs[k];
                    (void)hipStreamCreateWithFlags(&h.s, hipStreamNonBlocking);
                    (void)hipMalloc(&h.obs, sizeof(float) * n * 20);
                    (void)hipMalloc(&h.act, sizeof(float) * n * 4);
                    (void)hipMalloc(&h.rew, sizeof(float) * n);
                    (void)hipMalloc(&h.term, n);
                    (void)hipMalloc(&h.trunc, n);
                    DroneConfig c;
                    drone_config_default(&c, DRONE_TASK_HOVER);
                    c.buffer_kind = DRONE_BUFFERS_DEVICE;
                    c.env_offset = (uint32_t)(k * n);
                    h.v = drone_vec_init(h.obs, h.act, h.rew, h.term, h.trunc, n, 0, &c);
                    if (!h.v) { fprintf(stderr, "init failed: %s\n", drone_last_error()); return 1; }
                    drone_vec_set_stream(h.v, h.s);
                    drone_vec_reset(h.v, 0);
                    drone_vec_fill_random_actions(h.v, h.act, 0);
                }
                auto run = [&](int count) {
                    if (!threaded) {
                        for (int t = 0; t < count; t++)
                            for (auto& h : hs) drone_vec_step(h.v);
                    } else {
                        std::vector<std::thread> th;
                        for (auto& h : hs) th.emplace_back([&h, count]() { for (int t = 0; t < count; t++) drone_vec_step(h.v); });
                        for (auto& t : th) t.join();
                    }
                    for (auto& h : hs) drone_vec_sync(h.v);
                };
                run(200);
                const double t0 = now_us();
                run(steps);
                const double us = (now_us() - t0) / steps;
                printf("{\"envs\": %d, \"handles_x_streams\": %d, \"launch\": \"%s\", \"us_per_full_step\": %.3f, \"env_steps_per_s\": %.4g, \"frac_8TB\": %.3f}\n", total, H,
                       threaded ? "one host thread per handle" : "one host thread, round robin", us, total / us * 1e6, 278.0 * total / us / 1e6 / 8.0);
                for (auto& h : hs) {
                    drone_vec_close(h.v);
                    (void)hipFree(h.obs); (void)hipFree(h.act); (void)hipFree(h.rew); (void)hipFree(h.term); (void)hipFree(h.trunc);
                    (void)hipStreamDestroy(h.s);
                }
            }
        }
    }
    return 0;
}

// Host-only driver for drone_amd/csrc/drone_host_copy.hpp (the pool touches no HIP): built with g++ -fsanitize=thread by
// tests/test_copy_pool_host.py and run on the CPU. Covers what the library does with the pool: whole jobs (run), a job started
// on one thread and finished on another (drone_vec_step_send / drone_vec_step_recv), a second caller finding the pool busy,
// workers that went to sleep between jobs, the watch callback of finish().
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "drone_host_copy.hpp"

using drone::CopyPool;

struct Job {
    std::atomic<int> calls[32];
    std::atomic<int> parts_seen{0};
    std::vector<char> src, dst;
    Job(size_t bytes) : src(bytes), dst(bytes) {
        for (auto& c : calls) c.store(0);
        for (size_t i = 0; i < bytes; i++) src[i] = (char)(i * 131u + 7u);
    }
};

static void copy_part(void* ctx, int part, int parts) {
    Job* j = static_cast<Job*>(ctx);
    j->calls[part].fetch_add(1);
    j->parts_seen.store(parts);
    const size_t n = j->src.size(), b = n * (size_t)part / (size_t)parts, e = n * (size_t)(part + 1) / (size_t)parts;
    memcpy(j->dst.data() + b, j->src.data() + b, e - b);
}

static std::atomic<int> g_watch{0};
static void watch(void*) { g_watch.fetch_add(1); }
static void slow_part(void* ctx, int part, int parts) {
    if (part != 0) std::this_thread::sleep_for(std::chrono::milliseconds(3));  // the caller's part is done first: finish() has to wait, and watches
    copy_part(ctx, part, parts);
}

#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } } while (0)

int main() {
    CopyPool& pool = CopyPool::get();
    const int parts = pool.parts();
    CHECK(parts >= 2 && parts <= 16);
    // 1. whole jobs back to back: every part exactly once, every byte moved
    for (int rep = 0; rep < 2000; rep++) {
        Job j(64 * 1024 + (size_t)rep);
        pool.run(copy_part, &j);
        for (int p = 0; p < parts; p++) CHECK(j.calls[p].load() == 1);
        CHECK(j.parts_seen.load() == parts && j.src == j.dst);
    }
    // 2. workers asleep between jobs (they spin ~200 us, then wait on the condition variable)
    for (int rep = 0; rep < 20; rep++) {
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
        Job j(300000);
        pool.run(copy_part, &j);
        for (int p = 0; p < parts; p++) CHECK(j.calls[p].load() == 1);
        CHECK(j.src == j.dst);
    }
    // 3. started on one thread, finished on another; a third caller meanwhile finds the pool busy and works alone
    for (int rep = 0; rep < 200; rep++) {
        Job held(100000), other(50000);
        bool started = false;
        std::thread a([&] { started = pool.try_start(copy_part, &held); });
        a.join();
        CHECK(started);
        CHECK(!pool.try_start(copy_part, &other));  // busy: nothing started
        pool.run(copy_part, &other);                // ... so run() does the whole job on this thread
        CHECK(other.calls[0].load() == 1 && other.parts_seen.load() == 1 && other.src == other.dst);
        for (int p = 1; p < parts; p++) CHECK(other.calls[p].load() == 0);
        std::thread b([&] { pool.finish(); });
        b.join();
        for (int p = 0; p < parts; p++) CHECK(held.calls[p].load() == 1);
        CHECK(held.src == held.dst);
    }
    // 4. finish() keeps calling the watch while it waits for the helpers
    {
        Job j(4096);
        CHECK(pool.try_start(slow_part, &j));
        pool.finish(watch, nullptr);
        CHECK(g_watch.load() >= 1 && j.src == j.dst);
    }
    // 5. two callers racing for the pool: each job is complete whoever won it
    {
        std::atomic<int> bad{0};
        auto hammer = [&] {
            for (int rep = 0; rep < 500; rep++) {
                Job j(20000 + (size_t)rep);
                pool.run(copy_part, &j);
                if (!(j.src == j.dst)) bad.fetch_add(1);
            }
        };
        std::thread t1(hammer), t2(hammer);
        t1.join();
        t2.join();
        CHECK(bad.load() == 0);
    }
    printf("OK parts=%d watch=%d\n", parts, g_watch.load());
    return 0;
}

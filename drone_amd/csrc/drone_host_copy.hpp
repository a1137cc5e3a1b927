// drone_host_copy.hpp — a small pool of host threads for the host-buffer transports (round 5).
//
// PufferLib hands an env unaligned slices of one shared-memory block; such buffers may never be pinned
// (include/drone_vec.h, DroneConfig.host_pages_exclusive), so the kernel reads / writes pinned stand-ins the library owns
// and the HOST moves the bytes between them and the caller's memory. Up to ~1 MiB one memcpy does that; beyond, a single
// thread copies 6.6 MB per step at 65 536 envs in ~280 us against a 137 us step — and it did so strictly before and
// after the kernel. This pool lets (a) the action rows go in as parallel slices and (b) the outputs come out WHILE the
// kernel is still running: every workgroup of the step kernel raises a word in pinned host memory once its rows have
// landed (LaunchSig::wg_done), each thread follows the words of its slice of the envs and copies rows as they arrive.
//
// One pool per process, started on first use. Workers spin for a short while after a job (a vec-env steps every few
// hundred microseconds: a futex wake-up would cost a tenth of the step) and sleep on a condition variable when the
// handle goes quiet. One job at a time (a flag held from try_start() to finish()): whoever finds the pool busy does its
// copying alone, on its own thread. Nothing here touches HIP.
#pragma once

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <vector>
#if defined(__linux__)
#include <sched.h>
#endif

namespace drone {

class CopyPool {
public:
    typedef void (*Fn)(void* ctx, int part, int parts);

    // helpers: pool threads besides the caller. DRONE_HOST_COPY_THREADS=<total threads per job, caller included>; default: half
    // the CPUs this process may run on (its affinity mask, not the machine: a vec-env worker pinned to four cores gets two
    // threads), at most 8; 1 (or 0) = no pool: everything runs on the caller. Several worker processes on one box each have
    // a pool: give them DRONE_HOST_COPY_THREADS ~ cores / workers
    static CopyPool& get() {
        static CopyPool pool;
        return pool;
    }
    int parts() const { return (int)workers_.size() + 1; }

    // start fn(ctx, part, parts) on the helpers (parts 1 ... parts-1); the caller later runs part 0 itself inside finish().
    // false: the pool is busy with another job (another handle between its step_send and step_recv, possibly on this very
    // thread — waiting could never end): nothing was started, the caller does the work by itself, fn(ctx, 0, 1).
    bool try_start(Fn fn, void* ctx) {
        if (busy_.exchange(true, std::memory_order_acquire)) return false;
        fn_ = fn;
        ctx_ = ctx;
        done_.store(0, std::memory_order_relaxed);
        gen_.fetch_add(1);                 // seq_cst on purpose, like the sleepers' counter: "bump, then look for sleepers" against
        if (sleepers_.load() > 0) {        // "announce sleep, then look at the generation" must not both read the old value
            std::lock_guard<std::mutex> g(m_);
            cv_.notify_all();
        }
        return true;
    }
    // run part 0 here, then wait for the helpers; `watch(watch_ctx)` is called every few microseconds of that wait — the
    // caller's chance to tell the helpers that what they are waiting for will never come, or already has. (Part 0 runs BEFORE
    // the first watch: a job whose part 0 can stall must look for itself — drone_vec.cpp copy_outputs_part does.) A helper
    // that has been descheduled (an oversubscribed or quota-limited box) is waited for politely: after ~50 us of spinning the
    // caller yields its time slice between looks.
    void finish(void (*watch)(void*) = nullptr, void* watch_ctx = nullptr) {
        fn_(ctx_, 0, parts());
        const int want = (int)workers_.size();
        unsigned spins = 0;
        while (done_.load(std::memory_order_acquire) < want) {
            if (watch && (spins & 255u) == 0) watch(watch_ctx);
            if (++spins > 4096u) std::this_thread::yield();
            else cpu_relax();
        }
        busy_.store(false, std::memory_order_release);
    }
    void run(Fn fn, void* ctx) {
        if (try_start(fn, ctx)) finish();
        else fn(ctx, 0, 1);
    }

    static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#else
        std::this_thread::yield();
#endif
    }

private:
    CopyPool() {
        int hw = (int)std::thread::hardware_concurrency();
#if defined(__linux__)
        cpu_set_t mask;  // what this PROCESS may use: affinity masks and cpusets shrink it, hardware_concurrency does not notice
        if (sched_getaffinity(0, sizeof(mask), &mask) == 0 && CPU_COUNT(&mask) > 0 && CPU_COUNT(&mask) < hw) hw = CPU_COUNT(&mask);
#endif
        int total = hw >= 16 ? 8 : hw >= 4 ? hw / 2 : 1;  // half the machine's threads, at most eight (measured: 65 536 envs 211 us per step with four, 166 with eight; profiles/r05_ab/host_transports.txt)
        const char* e = getenv("DRONE_HOST_COPY_THREADS");
        if (e && *e) total = atoi(e);
        if (hw > 0 && total > hw) total = hw;
        if (total > 16) total = 16;
        for (int k = 1; k < total; k++) workers_.emplace_back([this, k] { loop(k); });
    }
    ~CopyPool() {
        stop_.store(true, std::memory_order_release);
        gen_.fetch_add(1, std::memory_order_release);
        {
            std::lock_guard<std::mutex> g(m_);
            cv_.notify_all();
        }
        for (auto& t : workers_) t.join();
    }
    void loop(int part) {
        unsigned seen = 0;
        for (;;) {
            // wait for the next generation: spin ~200 us, then sleep
            unsigned g = gen_.load(std::memory_order_acquire);
            if (g == seen) {
                const auto t0 = std::chrono::steady_clock::now();
                unsigned spins = 0;
                while ((g = gen_.load(std::memory_order_acquire)) == seen) {
                    cpu_relax();
                    if ((++spins & 1023u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(200)) {
                        std::unique_lock<std::mutex> lk(m_);
                        sleepers_.fetch_add(1);
                        cv_.wait(lk, [&] { return gen_.load() != seen; });
                        sleepers_.fetch_sub(1);
                        g = gen_.load(std::memory_order_acquire);
                        break;
                    }
                }
            }
            seen = g;
            if (stop_.load(std::memory_order_acquire)) return;
            fn_(ctx_, part, parts());
            done_.fetch_add(1, std::memory_order_release);
        }
    }

    std::vector<std::thread> workers_;
    std::atomic<bool> busy_{false};  // a flag, not a mutex: drone_vec_step_send and drone_vec_step_recv may come from different threads, and a mutex belongs to the thread that locked it
    std::mutex m_;
    std::condition_variable cv_;
    std::atomic<unsigned> gen_{0};
    std::atomic<int> done_{0}, sleepers_{0};
    std::atomic<bool> stop_{false};
    Fn fn_ = nullptr;
    void* ctx_ = nullptr;
};

}  // namespace drone

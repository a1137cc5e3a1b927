// drone_state.cpp — state import / export (tests, checkpoints), done-id lists, device memory helpers and timers of the C-ABI, and the
// opt-in sweep autotuner (DRONE_AUTOTUNE=1). Split out of drone_vec.cpp in round 6; see drone_vec_impl.hpp.
#include <new>

#include "drone_vec_impl.hpp"

DRONE_IMPL_NS {

// Pick the per-step kernel's sweep order / load hints by MEASUREMENT (round 5) — ONLINE, on the handle's real
// steps. Candidates: the footprint table's entry and its neighbours — a plain round-robin sweep (0), the sweep that turns
// around on odd steps with streamed action rows (6), a plain sweep with non-temporal state loads (8) — all instantiations the
// parity suite and the soak cover: the order only permutes which workgroup takes which chunk and which loads carry a hint,
// never a result, so real steps may run under any of them. From the handle's 161st step launch on, the candidates take turns
// in bursts of sixteen steps, twice each; every launch of a burst but its first four (the cache is still in the previous
// candidate's state) sits between two HIP events on the stream, read back lazily (hipEventQuery) when later calls find them
// complete — nothing waits, nothing extra is launched, no state or output is touched. When every candidate has its samples
// the fastest becomes the handle's order and drone_vec_variant() says what was measured
// (" autotuned=1 table=8 tried=o8:170.1,o0:178.8,o6:170.3").
// Why online: a first version timed trial steps right behind the first reset (profiles/r05_ab/autotune_offline_*.txt). No
// episode ends that soon after a reset — and the ranking depends on them: at 2^21 hover envs the plain sweep ran 80.8 us in
// that trial and 94.5 in steady state under the random policy (one episode end per 146 env-steps; the scattered log-plane and
// target updates cost it its cache residency), where the non-temporal sweep it "beat" runs 86.2 either way; at 2^23 the trial
// picked order 8, 4 % behind order 6 in steady state. The table, tuned in steady state, was right in all eight cases on two
// boxes; the offline trial in five. What a handle should measure is the workload it actually runs.
void tune_free(DroneVec* v) {
    if (!v->tune) return;
    for (auto& p : v->tune->pairs) {
        if (p.e0) (void)hipEventDestroy(p.e0);
        if (p.e1) (void)hipEventDestroy(p.e1);
    }
    delete v->tune;
    v->tune = nullptr;
}

void tune_harvest(SweepTune* t) {
    for (auto& p : t->pairs) {
        if (p.cand < 0) continue;
        const hipError_t q = hipEventQuery(p.e1);
        if (q == hipErrorNotReady) { (void)hipGetLastError(); continue; }
        float ms = 0.f;
        if (q == hipSuccess && hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
            t->sum_ms[p.cand] += ms;
            t->samples[p.cand] += 1;
        } else {
            (void)hipGetLastError();
        }
        p.cand = -1;
        t->in_flight -= 1;
    }
}

// called around every per-step launch of a handle that is still measuring: before it (returns the event pair to close behind the
// launch, or null) — may change v->dv.order for this launch
SweepTune::Pair* tune_before_step(DroneVec* v) {
    SweepTune* t = v->tune;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(v->stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone || v->dv.ctr || (v->gather && v->gather->peer)) {
        (void)hipGetLastError();  // a capture, graph-safe counters or a peer-store exchange: the table's choice stands for good
        v->dv.order = t->table;
        tune_free(v);
        return nullptr;
    }
    tune_harvest(t);
    const long k = t->seen++ - SweepTune::kStart;
    if (k < 0) return nullptr;
    const long burst = k / SweepTune::kBurst;
    if (burst >= (long)t->nc * SweepTune::kRounds) {  // exploration is over: decide once every pair has been read
        v->dv.order = t->table;
        if (t->in_flight > 0) return nullptr;
        int best = -1;
        for (int c = 0; c < t->nc; c++)
            if (t->samples[c] >= SweepTune::kBurst / 2 && (best < 0 || t->sum_ms[c] / t->samples[c] < t->sum_ms[best] / t->samples[best])) best = c;
        char tuned[200];
        int at = 0;
        if (best >= 0 && t->samples[0] >= SweepTune::kBurst / 2) {  // (without enough samples of the table's own entry there is nothing to compare with)
            v->dv.order = t->cand[best];
            at = snprintf(tuned, sizeof(tuned), " autotuned=1 table=%u tried=", t->table);
            for (int c = 0; c < t->nc && at < (int)sizeof(tuned) - 16; c++)
                at += snprintf(tuned + at, sizeof(tuned) - at, "%so%u:%.1f", c ? "," : "", t->cand[c], t->samples[c] ? t->sum_ms[c] * 1e3 / t->samples[c] : 0.0);
        } else {
            snprintf(tuned, sizeof(tuned), " autotuned=0 table=%u", t->table);
        }
        tune_free(v);
        write_variant(v, tuned);
        return nullptr;
    }
    const int c = (int)(burst % t->nc);
    v->dv.order = t->cand[c];
    if (k % SweepTune::kBurst < SweepTune::kLead) return nullptr;
    for (auto& p : t->pairs) {
        if (p.cand >= 0) continue;
        if (!p.e0 && (hipEventCreate(&p.e0) != hipSuccess || hipEventCreate(&p.e1) != hipSuccess)) { (void)hipGetLastError(); return nullptr; }
        if (hipEventRecord(p.e0, v->stream) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        p.cand = c;
        t->in_flight += 1;
        return &p;
    }
    return nullptr;  // every pair is still in flight (a caller far ahead of the device): this launch goes unmeasured
}


}  // namespace drone_impl

using namespace drone_impl;

extern "C" {

// ---- AoS import / export (tests, checkpoints): plain copies + host repack ----
namespace {

// host image of the tiles that cover envs [first, first + count) plus the matching pieces of the two cold planes
struct StateImage {
    uint32_t nph, tile0, ntiles, first;
    std::vector<float4> hot, cold;  // hot: the covered tiles, laid out like the device region but for ntiles * 64 drones; cold: [2][count]
    float4& at(uint32_t plane, uint32_t env) { return hot[hot_index(nph, plane, env - tile0 * kTile, ntiles * kTile)]; }
};

// copy the covered part of the hot region between the device and the image (one piece when tiled, one per plane otherwise)
bool image_copy(DroneVec* v, StateImage& im, bool to_device) {
#if DRONE_TILED_STATE
    float4* dev = v->dv.planes + (size_t)im.tile0 * im.nph * kTile;
    if (to_device) HIP_TRY(hipMemcpyAsync(dev, im.hot.data(), sizeof(float4) * im.hot.size(), hipMemcpyHostToDevice, v->stream), return false);
    else HIP_TRY(hipMemcpyAsync(im.hot.data(), dev, sizeof(float4) * im.hot.size(), hipMemcpyDeviceToHost, v->stream), return false);
#else
    const size_t w = (size_t)im.ntiles * kTile;
    for (uint32_t p = 0; p < im.nph; p++) {
        float4* dev = v->dv.planes + (size_t)p * v->n_pad + (size_t)im.tile0 * kTile;
        float4* host = im.hot.data() + (size_t)p * w;
        if (to_device) HIP_TRY(hipMemcpyAsync(dev, host, sizeof(float4) * w, hipMemcpyHostToDevice, v->stream), return false);
        else HIP_TRY(hipMemcpyAsync(host, dev, sizeof(float4) * w, hipMemcpyDeviceToHost, v->stream), return false);
    }
#endif
    return true;
}

bool image_fetch(DroneVec* v, int first, int count, StateImage& im) {
    im.nph = hot_planes(v->cfg.task, v->dv.derived_target != 0);
    im.first = (uint32_t)first;
    im.tile0 = (uint32_t)first / kTile;
    im.ntiles = ((uint32_t)(first + count) + kTile - 1) / kTile - im.tile0;
    im.hot.resize((size_t)im.ntiles * im.nph * kTile);
    im.cold.resize((size_t)2 * count);
    if (count == 0) return true;
    if (!image_copy(v, im, false)) return false;
    for (int k = 0; k < 2; k++)
        HIP_TRY(hipMemcpyAsync(im.cold.data() + (size_t)k * count, v->dv.cold + (size_t)k * v->stride + first, sizeof(float4) * count, hipMemcpyDeviceToHost, v->stream), return false);
    HIP_TRY(hipStreamSynchronize(v->stream), return false);
    return true;
}

}  // namespace

int drone_vec_get_state(DroneVec* v, DroneStateRow* rows, int first, int count) {
    Entry in(v);
    if (!in || !idle(v, "get_state")) return -1;
    if (!rows || first < 0 || count < 0 || first + count > v->n) { set_err("get_state: bad range"); return -1; }
    StateImage im;
    if (!image_fetch(v, first, count, im)) return -1;
    auto u = [](float f) { uint32_t x; memcpy(&x, &f, 4); return x; };
    const bool aux = im.nph == 7, dt = v->dv.derived_target != 0;
    for (int k = 0; k < count; k++) {
        const uint32_t e = (uint32_t)(first + k);
        const float4 a = im.at(kP0, e), b = im.at(kP1, e), c = im.at(kP2, e), d = im.at(kP3, e), ee = im.at(kP4, e);
        const float4 t = dt ? make_float4(0.f, 0.f, 0.f, 0.f) : im.at(kPT, e);
        const float4 w = aux ? im.at(kPW, e) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 l0 = im.cold[k], l1 = im.cold[(size_t)count + k];
        DroneStateRow& r = rows[k];
        r.pos[0] = a.x; r.pos[1] = a.y; r.pos[2] = a.z; r.vel[0] = a.w;
        r.vel[1] = b.x; r.vel[2] = b.y; r.quat[0] = b.z; r.quat[1] = b.w;
        r.quat[2] = c.x; r.quat[3] = c.y; r.omega[0] = c.z; r.omega[1] = c.w;
        r.omega[2] = d.x; r.rpm[0] = d.y; r.rpm[1] = d.z; r.rpm[2] = d.w;
        r.rpm[3] = ee.x; r.ep_return = ee.y;
        if (dt) {  // derived-target layout: counters packed in P4, the target re-derived as the kernels do
            r.tick = u(ee.z) & 0xFFFFu; r.score_count = u(ee.z) >> 16; r.episode = u(ee.w);
            derive_target(v->kp, v->kp.env_offset + e, r.episode, r.target);
        } else {
            r.tick = u(ee.z); r.score_count = u(ee.w);
            r.target[0] = t.x; r.target[1] = t.y; r.target[2] = t.z; r.episode = u(t.w);
        }
        r.wind[0] = w.x; r.wind[1] = w.y; r.wind[2] = w.z;
        r.perf_sum = l0.x; r.score_sum = l0.y; r.ret_sum = l0.z; r.len_sum = l0.w;
        r.n_sum = l1.x; r.oob_sum = l1.y;
    }
    return 0;
}

int drone_vec_set_state(DroneVec* v, const DroneStateRow* rows, int first, int count) {
    Entry in(v);
    if (!in || !idle(v, "set_state")) return -1;
    if (!rows || first < 0 || count < 0 || first + count > v->n) { set_err("set_state: bad range"); return -1; }
    if (count == 0) return 0;
    // the tiles at the edges of the range also hold neighbours: fetch, patch the rows, write the tiles back
    StateImage im;
    if (!image_fetch(v, first, count, im)) return -1;
    auto f = [](uint32_t x) { float y; memcpy(&y, &x, 4); return y; };
    const bool aux = im.nph == 7, dt = v->dv.derived_target != 0;
    for (int k = 0; k < count; k++) {
        const uint32_t e = (uint32_t)(first + k);
        const DroneStateRow& r = rows[k];
        if (dt) {  // this layout stores no target: the row's must be the one its (env, episode) implies, and the counters must fit
            float want[3];
            derive_target(v->kp, v->kp.env_offset + e, r.episode, want);
            if (memcmp(want, r.target, sizeof(want)) != 0 || r.tick > 0xFFFFu || r.score_count > 0xFFFFu) {
                set_err("set_state: env %u: the derived-target layout (hover / swarm, DRONE_DERIVED_TARGET) cannot hold a target other than the one "
                        "SPEC.md section 6 draws for (env, episode), nor counters beyond 65535; create the handle with DRONE_DERIVED_TARGET=0 for free-form states", e);
                return -1;
            }
        }
        im.at(kP0, e) = make_float4(r.pos[0], r.pos[1], r.pos[2], r.vel[0]);
        im.at(kP1, e) = make_float4(r.vel[1], r.vel[2], r.quat[0], r.quat[1]);
        im.at(kP2, e) = make_float4(r.quat[2], r.quat[3], r.omega[0], r.omega[1]);
        im.at(kP3, e) = make_float4(r.omega[2], r.rpm[0], r.rpm[1], r.rpm[2]);
        if (dt) {
            im.at(kP4, e) = make_float4(r.rpm[3], r.ep_return, f(r.tick | (r.score_count << 16)), f(r.episode));
        } else {
            im.at(kP4, e) = make_float4(r.rpm[3], r.ep_return, f(r.tick), f(r.score_count));
            im.at(kPT, e) = make_float4(r.target[0], r.target[1], r.target[2], f(r.episode));
        }
        if (aux) im.at(kPW, e) = make_float4(r.wind[0], r.wind[1], r.wind[2], 0.0f);
        im.cold[k] = make_float4(r.perf_sum, r.score_sum, r.ret_sum, r.len_sum);
        im.cold[(size_t)count + k] = make_float4(r.n_sum, r.oob_sum, 0.0f, 0.0f);
    }
    if (!image_copy(v, im, true)) return -1;
    for (int k = 0; k < 2; k++)
        HIP_TRY(hipMemcpyAsync(v->dv.cold + (size_t)k * v->stride + first, im.cold.data() + (size_t)k * count, sizeof(float4) * count, hipMemcpyHostToDevice, v->stream), return -1);
    HIP_TRY(hipStreamSynchronize(v->stream), return -1);
    return 0;
}

namespace {

// `cnt_dev`: the device counter of one list, `ids_dev` its ids. The 4-byte count is read and CHECKED after the stream
// has drained (a stack destination of an async copy holds nothing before that).
int fetch_done_list(DroneVec* v, const uint32_t* cnt_dev, const uint32_t* ids_dev, uint32_t* ids, int cap) {
    uint32_t cnt = 0;
    HIP_TRY(hipMemcpyAsync(&cnt, cnt_dev, sizeof(uint32_t), hipMemcpyDeviceToHost, v->stream), return -1);
    HIP_TRY(hipStreamSynchronize(v->stream), return -1);
    if (cnt > (uint32_t)v->n) { set_err("done list count %u exceeds num_envs %d (corrupt counter)", cnt, v->n); return -1; }
    const int take = (int)cnt < cap ? (int)cnt : cap;
    if (ids && take > 0) {
        HIP_TRY(hipMemcpyAsync(ids, ids_dev, sizeof(uint32_t) * take, hipMemcpyDeviceToHost, v->stream), return -1);
        HIP_TRY(hipStreamSynchronize(v->stream), return -1);
    }
    return (int)cnt;
}

}  // namespace

int drone_vec_done_list(DroneVec* v, uint32_t* ids, int cap) {
    Entry in(v);
    if (!in || !idle(v, "done_list")) return -1;
    if (!v->dv.done_ids) { set_err("done list not enabled (compact_done=0)"); return -1; }
    if (!pull_counters(v)) return -1;
    if (!v->list_valid || v->step_launches == 0) return 0;  // after reset / after a fused rollout there is no list
    return fetch_done_list(v, v->dv.done_count + ((v->step_launches - 1u) & 1u), v->dv.done_ids, ids, cap);
}

int drone_vec_done_list_at(DroneVec* v, int k, uint32_t* ids, int cap) {
    Entry in(v);
    if (!in) return -1;
    if (!v->cfg.compact_done) { set_err("done list not enabled (compact_done=0)"); return -1; }
    if (v->many_k <= 0) { set_err("done_list_at: the last path call was not drone_vec_step_many"); return -1; }
    if (k < 0 || k >= v->many_k) { set_err("done_list_at: step %d outside the last step_many's %d steps", k, v->many_k); return -1; }
    return fetch_done_list(v, v->many_count + k, v->many_ids + (size_t)k * (size_t)v->n, ids, cap);
}

void* drone_device_malloc(int device, size_t bytes) {
    err_text()[0] = 0;
    DeviceRestore restore;
    void* p = nullptr;
    HIP_TRY(hipSetDevice(device), return nullptr);
    HIP_TRY(hipMalloc(&p, bytes ? bytes : 1), return nullptr);
    HIP_TRY(hipMemset(p, 0, bytes ? bytes : 1), { (void)hipFree(p); return nullptr; });
    return p;
}

void drone_device_free(int device, void* p) {
    if (!p) return;
    DeviceRestore restore;
    if (hipSetDevice(device) == hipSuccess) (void)hipFree(p);
    else (void)hipGetLastError();
}

int drone_vec_copy_to_host(DroneVec* v, void* host_dst, const void* device_src, size_t bytes) {
    Entry in(v);
    if (!in) return -1;
    if (!host_dst || !device_src) { set_err("copy_to_host: NULL argument"); return -1; }
    HIP_TRY(hipMemcpyAsync(host_dst, device_src, bytes, hipMemcpyDeviceToHost, v->stream), return -1);
    HIP_TRY(hipStreamSynchronize(v->stream), return -1);
    if (v->gather && v->gather->peer && !peer_check_err(v->gather)) return -1;  // a wait ahead of this copy gave up: the batch is not this round's
    return 0;
}

int drone_vec_timer_start(DroneVec* v) {
    Entry in(v);
    if (!in) return -1;
    HIP_TRY(hipEventRecord(v->ev0, v->stream), return -1);
    return 0;
}

int drone_vec_timer_stop(DroneVec* v, float* elapsed_ms) {
    Entry in(v);
    if (!in || !elapsed_ms) return -1;
    HIP_TRY(hipEventRecord(v->ev1, v->stream), return -1);
    HIP_TRY(hipEventSynchronize(v->ev1), return -1);
    HIP_TRY(hipEventElapsedTime(elapsed_ms, v->ev0, v->ev1), return -1);
    return 0;
}

#if defined(DRONE_STAMPS) && DRONE_STAMPS
// diagnostic build only; not part of include/drone_vec.h
int drone_debug_stamps(DroneVec* v, unsigned long long* out, int max_rows) {
    Entry in(v);
    if (!in || !v->dv.stamps) return -1;
    const int rows = (int)(v->n_pad / 64) < max_rows ? (int)(v->n_pad / 64) : max_rows;
    HIP_TRY(hipMemcpyAsync(out, v->dv.stamps, sizeof(unsigned long long) * kStampSlots * rows, hipMemcpyDeviceToHost, v->stream), return -1);
    HIP_TRY(hipStreamSynchronize(v->stream), return -1);
    return rows;
}
#endif

}  // extern "C"

// drone_vec.cpp — host side of the C-ABI in include/drone_vec.h.
//
// Owns the device state planes, the KParams block, the stream and (in
// host-buffer mode) the device mirrors of the caller's buffers. Every entry
// point ends in a HIP launch from drone_kernels.hip: there is no CPU
// implementation of the env in this library, and it fails loudly (NULL /
// non-zero + drone_last_error()) when HIP or the device is unavailable.
//
// Replaces, on the PufferLib side, the binding's vec_init / vec_reset /
// vec_step / vec_log / vec_close loop over per-env c_step; the
// reference file:line cannot be cited — no source in /root/reference
// (.gitmodules:1-3).
#include <new>

#include "drone_vec_impl.hpp"

DRONE_IMPL_NS {

static thread_local char g_err_storage[kErrBytes] = "";
char* err_text() { return g_err_storage; }

void set_err(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err_storage, sizeof(g_err_storage), fmt, ap);
    va_end(ap);
}


// Plane stride padding, in float4 elements (DRONE_PLANE_PAD, tuning experiments
// only). With a power-of-two env count the planes sit exactly 2^k bytes apart;
// the HBM address hash copes: pads of 16...65552 elements measure within +-0.7 %
// of no pad at equal placement (profiles/r01_ab/ab14_pad.txt), so the default is 0.
uint32_t plane_pad_elems() {
    const char* e = getenv("DRONE_PLANE_PAD");
    if (e && *e) return (uint32_t)strtoul(e, nullptr, 10);
    return 0;
}

// graph-safe stepping: the device copies of the counters ({gstep, step launches, arrivals}); the host fields mirror them
bool push_counters(DroneVec* v) {
    if (!v->dv.ctr) return true;
    const uint32_t c[3] = {v->gstep, v->step_launches, 0u};
    HIP_TRY(hipMemcpyAsync(v->dv.ctr, c, sizeof(c), hipMemcpyHostToDevice, v->stream), return false);
    HIP_TRY(hipStreamSynchronize(v->stream), return false);  // `c` is on the stack
    return true;
}
bool pull_counters(DroneVec* v) {
    if (!v->dv.ctr) return true;
    uint32_t c[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(c, v->dv.ctr, sizeof(c), hipMemcpyDeviceToHost, v->stream), return false);
    HIP_TRY(hipStreamSynchronize(v->stream), return false);
    if (c[1] != v->step_launches) v->list_valid = true;  // steps ran (e.g. graph replays) since the host last looked
    v->gstep = c[0];
    v->step_launches = c[1];
    return true;
}

bool upload_params(DroneVec* v) {
    derive_kparams(v->cfg, v->seed, v->kp);
    HIP_TRY(hipMemcpyAsync(v->d_kp, &v->kp, sizeof(KParams), hipMemcpyHostToDevice, v->stream), return false);
    // the source is the handle's own field; make the copy complete before it can change again
    HIP_TRY(hipStreamSynchronize(v->stream), return false);
    return true;
}

bool validate(const DroneConfig* c, int num_envs) {
    if (!c) { set_err("config is NULL"); return false; }
    if (c->struct_size != sizeof(DroneConfig)) { set_err("DroneConfig.struct_size %u != %zu", c->struct_size, sizeof(DroneConfig)); return false; }
    if (num_envs <= 0) { set_err("num_envs must be positive"); return false; }
    if (c->task != DRONE_TASK_HOVER && c->task != DRONE_TASK_WAYPOINT && c->task != DRONE_TASK_SWARM && c->task != DRONE_TASK_RACE) { set_err("unknown task %d", c->task); return false; }
    if (c->task == DRONE_TASK_SWARM) {
        const int A = c->agents_per_env;
        if (A < 1 || A > 64 || (A & (A - 1))) { set_err("agents_per_env must be a power of two in [1, 64], got %d", A); return false; }
        if (num_envs % A || c->env_offset % (uint32_t)A) { set_err("num_envs and env_offset must be multiples of agents_per_env (%d)", A); return false; }
        if (!(c->proximity_radius > 0.0f)) { set_err("proximity_radius must be positive"); return false; }
    }
    {   // state addressing in the kernels is 32-bit: the hot region's n_pad x planes-per-tile float4 elements must fit
        const uint64_t n_pad = ((uint64_t)num_envs + kBlock - 1) / kBlock * kBlock;
        if (n_pad * 7u > 0xFFFFFFFFull || 2u * (n_pad + plane_pad_elems()) > 0xFFFFFFFFull) {
            set_err("num_envs %d too large: the state region must fit 32-bit element indices (max about %llu envs per handle; shard further)",
                    num_envs, (unsigned long long)(0xFFFFFFFFull / 7u - kBlock));
            return false;
        }
    }
    if (c->state_layout != DRONE_LAYOUT_AUTO && c->state_layout != DRONE_LAYOUT_TARGET_PLANE && c->state_layout != DRONE_LAYOUT_DERIVED_TARGET) { set_err("unknown state_layout %d", c->state_layout); return false; }
    if (c->state_layout == DRONE_LAYOUT_DERIVED_TARGET && !(task_has_derived_target(c->task) && c->horizon <= 65535)) {
        set_err("state_layout = DRONE_LAYOUT_DERIVED_TARGET needs the hover or swarm task and horizon <= 65535 (task %d, horizon %d)", c->task, c->horizon);
        return false;
    }
    if (c->buffer_kind != DRONE_BUFFERS_HOST && c->buffer_kind != DRONE_BUFFERS_DEVICE) { set_err("unknown buffer_kind %d", c->buffer_kind); return false; }
    if (c->substeps < 1 || c->horizon < 1) { set_err("substeps and horizon must be >= 1"); return false; }
    if (!(c->dt > 0.0f) || !(c->mass > 0.0f) || !(c->ixx > 0.0f) || !(c->iyy > 0.0f) || !(c->izz > 0.0f) || !(c->motor_tau > 0.0f) ||
        !(c->max_rpm > 0.0f) || !(c->max_vel > 0.0f) || !(c->max_omega > 0.0f) || !(c->bound > 0.0f) || !(c->k_thrust > 0.0f)) {
        set_err("physical constants must be positive");
        return false;
    }
    if (c->task == DRONE_TASK_WAYPOINT && !(c->wind_max > 0.0f)) { set_err("wind_max must be positive (SPEC.md §4: clamp bounds are never zero)"); return false; }
    return true;
}

// every path / plumbing call except drone_vec_step_recv and close: not while a sent step is in flight
bool idle(DroneVec* v, const char* what) {
    if (!v->pending) return true;
    set_err("%s: a step sent with drone_vec_step_send has not been received (drone_vec_step_recv)", what);
    return false;
}

// drone_vec_variant's text; `tuned`: " autotuned=1 table=8 tried=o8:170.1,o0:178.8,o6:170.3" once the handle has measured the candidates (SweepTune below)
void write_variant(DroneVec* v, const char* tuned) {
    snprintf(v->variant, sizeof(v->variant), "drone_step_kernel<task=%d,compact=%d,mem=%u,dt=%d> order=%u line_complete=%u packed_rk4=%u bytes=%d%s",
             v->cfg.task, v->dv.done_ids ? 1 : 0, (v->dv.order >> 2) & 3u, v->dv.derived_target ? 1 : 0, v->dv.order, v->dv.line_complete,
             v->dv.packed_rk4, drone_vec_bytes_per_env_step(v), tuned ? tuned : "");
}


}  // namespace drone_impl

using namespace drone_impl;

extern "C" {

const char* drone_last_error(void) { return err_text(); }

void drone_config_default(DroneConfig* c, int task) {
    memset(c, 0, sizeof(*c));
    c->struct_size = (uint32_t)sizeof(DroneConfig);
    c->task = task;
    c->buffer_kind = DRONE_BUFFERS_HOST;
    c->horizon = 1024;
    c->substeps = 1;
    c->dt = 0.01f;
    // Crazyflie-2.x-class airframe (public datasheet-level numbers)
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
    c->agents_per_env = task == DRONE_TASK_SWARM ? 8 : 1;
    c->collision_radius = 0.15f;
    c->proximity_radius = 1.0f;
    c->c_proximity = 0.5f;
    c->gate_radius = 0.75f;
}

int drone_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int drone_vec_bytes_per_env_step(const DroneVec* v) {
    if (!v) return 0;
    const int task = v->cfg.task;
    const int aux = (task == DRONE_TASK_WAYPOINT || task == DRONE_TASK_RACE) ? 1 : 0, dt = v->dv.derived_target ? 1 : 0;
    // read: planes (6 / 7, or 5 derived-target) + action row; written: planes but the target one (+ the wind plane every step for
    // task 1; the gate normal of task 3 only per gate) + observation row + reward + 2 flag bytes
    const int planes_read = 6 + aux - dt, planes_written = 5 + (task == DRONE_TASK_WAYPOINT ? 1 : 0);
    return 16 * (planes_read + planes_written) + 16 + 4 * drone_obs_dim(task) + 4 + 2;
}

const char* drone_vec_variant(const DroneVec* v) { return v ? v->variant : ""; }

int drone_vec_host_transport(const DroneVec* v) {
    if (!v || !v->host_buffers) return -1;
    if (!v->zero_copy) return 0;
    for (int k = 0; k < 5; k++)
        if (v->bounce[k]) return v->threaded ? 3 : 2;
    return 1;
}

int drone_vec_buffers(const DroneVec* v, float** observations, float** actions, float** rewards, unsigned char** terminals, unsigned char** truncations) {
    if (!v) { set_err("NULL handle"); return -1; }
    if (observations) *observations = v->u_obs;
    if (actions) *actions = v->u_act;
    if (rewards) *rewards = v->u_rew;
    if (terminals) *terminals = v->u_term;
    if (truncations) *truncations = v->u_trunc;
    return 0;
}

int drone_vec_device(const DroneVec* v) { return v ? v->device : -1; }

int drone_obs_dim(int task) { return (task == DRONE_TASK_SWARM || task == DRONE_TASK_RACE) ? DRONE_OBS_DIM_MAX : DRONE_OBS_DIM; }

DroneVec* drone_vec_init(float* observations, float* actions, float* rewards, unsigned char* terminals,
                         unsigned char* truncations, int num_envs, uint64_t seed, const DroneConfig* cfg) {
    err_text()[0] = 0;
    if (!validate(cfg, num_envs)) return nullptr;
    // all five NULL on a device-buffer handle: the library allocates them in HBM (drone_vec_buffers hands them out)
    const bool lib_buffers = cfg->buffer_kind == DRONE_BUFFERS_DEVICE && !observations && !actions && !rewards && !terminals && !truncations;
    if (!lib_buffers && (!observations || !actions || !rewards || !terminals || !truncations)) {
        set_err("buffer pointer is NULL (only a DRONE_BUFFERS_DEVICE handle may pass all five as NULL: library-owned buffers)");
        return nullptr;
    }
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev), return nullptr);
    if (ndev <= 0 || cfg->device < 0 || cfg->device >= ndev) { set_err("HIP device %d not available (%d devices): this library has no CPU path", cfg->device, ndev); return nullptr; }
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, cfg->device), return nullptr);
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) { set_err("device %d is %s; this library is built for gfx950 only", cfg->device, prop.gcnArchName); return nullptr; }

    DeviceRestore restore_callers_device;
    DroneVec* v = new (std::nothrow) DroneVec();
    if (!v) { set_err("out of memory"); return nullptr; }
    memset(v, 0, sizeof(*v));
    v->cfg = *cfg;
    v->seed = seed;
    v->n = num_envs;
    v->n_pad = (uint32_t)(((uint64_t)num_envs + kBlock - 1) / kBlock * kBlock);
    v->stride = v->n_pad + plane_pad_elems();
    v->device = cfg->device;
    v->host_buffers = cfg->buffer_kind == DRONE_BUFFERS_HOST;
    v->u_obs = observations; v->u_act = actions; v->u_rew = rewards; v->u_term = terminals; v->u_trunc = truncations;

    const size_t n = (size_t)num_envs;
#define INIT_TRY(expr) HIP_TRY(expr, { drone_vec_close(v); return nullptr; })
    INIT_TRY(hipSetDevice(v->device));
    INIT_TRY(hipStreamCreateWithFlags(&v->stream, hipStreamNonBlocking));
    v->own_stream = true;
    INIT_TRY(hipEventCreate(&v->ev0));
    INIT_TRY(hipEventCreate(&v->ev1));
    {   // Derived-target layout (drone_params.hpp): hover / swarm handles whose step is HBM-bound drop the target plane
        // (16 of hover's 278 bytes per env-step) and re-derive the target from three hashes per step. It needs tick
        // and score_count in 16 bits each (horizon <= 65 535). Chosen by footprint like the other layout knobs
        // (profiles/r03_ab/ab_dt_*.txt: -4.4 % at 2^19 envs, -5.2 % at 2^20, -6.4 % at 2^21, -4.7 % at 2^22; neutral at
        // 2^18 and +1.4 % at 131 072, where the step is bound by one or two waves per SIMD issuing VALU, not by bytes);
        // DRONE_DERIVED_TARGET=0/1 forces it (1 is ignored where the layout cannot represent the handle).
        // Who decides: DRONE_DERIVED_TARGET (tuning tools) over DroneConfig.state_layout (the caller's declared choice) over
        // the footprint rule.
        const bool can = task_has_derived_target(cfg->task) && cfg->horizon <= 65535;
        const char* e = getenv("DRONE_DERIVED_TARGET");
        const size_t per_step = (size_t)num_envs * 278u;
        if (e && *e) v->dv.derived_target = can && atoi(e) != 0;
        else if (cfg->state_layout == DRONE_LAYOUT_DERIVED_TARGET) v->dv.derived_target = 1;  // validate() checked that it can
        else if (cfg->state_layout == DRONE_LAYOUT_TARGET_PLANE) v->dv.derived_target = 0;
        else v->dv.derived_target = can && per_step >= ((size_t)100 << 20);
    }
    const bool dt = v->dv.derived_target != 0;
    const size_t hot_elems = (size_t)v->n_pad * hot_planes(cfg->task, dt), cold_elems = (size_t)2 * v->stride;
    INIT_TRY(hipMalloc((void**)&v->dv.planes, sizeof(float4) * hot_elems));
    INIT_TRY(hipMemsetAsync(v->dv.planes, 0, sizeof(float4) * hot_elems, v->stream));
    INIT_TRY(hipMalloc((void**)&v->dv.cold, sizeof(float4) * cold_elems));
    INIT_TRY(hipMemsetAsync(v->dv.cold, 0, sizeof(float4) * cold_elems, v->stream));
    INIT_TRY(hipMalloc((void**)&v->d_kp, sizeof(KParams)));
    INIT_TRY(hipMalloc((void**)&v->dv.pad_sink, sizeof(float) * kBlock));
#if defined(DRONE_STAMPS) && DRONE_STAMPS  // diagnostic build (tools/stamps.py)
    INIT_TRY(hipMalloc((void**)&v->dv.stamps, sizeof(unsigned long long) * kStampSlots * (v->n_pad / 64)));
#endif
    INIT_TRY(hipMalloc((void**)&v->d_partials, sizeof(double) * 6 * kLogMaxGrid));
    INIT_TRY(hipHostMalloc((void**)&v->h_partials, sizeof(double) * 6 * kLogMaxGrid, hipHostMallocDefault));
    if (cfg->compact_done) {
        INIT_TRY(hipMalloc((void**)&v->dv.done_ids, sizeof(uint32_t) * n));
        INIT_TRY(hipMalloc((void**)&v->dv.done_count, sizeof(uint32_t) * 2));
        INIT_TRY(hipMemsetAsync(v->dv.done_count, 0, sizeof(uint32_t) * 2, v->stream));
    }
    if (v->host_buffers) {
        INIT_TRY(hipMalloc((void**)&v->d_obs, n * (size_t)drone_obs_dim(v->cfg.task) * sizeof(float)));
        INIT_TRY(hipMalloc((void**)&v->d_act, n * DRONE_ACT_DIM * sizeof(float)));
        INIT_TRY(hipMalloc((void**)&v->d_rew, n * sizeof(float)));
        INIT_TRY(hipMalloc((void**)&v->d_term, n));
        INIT_TRY(hipMalloc((void**)&v->d_trunc, n));
        choose_host_transport(v);
    } else {
        if (lib_buffers) {  // freed at close like the host handles' mirrors, which these fields otherwise hold
            const size_t ob = n * (size_t)drone_obs_dim(v->cfg.task) * sizeof(float);
            INIT_TRY(hipMalloc((void**)&v->d_obs, ob));
            INIT_TRY(hipMalloc((void**)&v->d_act, n * DRONE_ACT_DIM * sizeof(float)));
            INIT_TRY(hipMalloc((void**)&v->d_rew, n * sizeof(float)));
            INIT_TRY(hipMalloc((void**)&v->d_term, n));
            INIT_TRY(hipMalloc((void**)&v->d_trunc, n));
            INIT_TRY(hipMemsetAsync(v->d_obs, 0, ob, v->stream));
            INIT_TRY(hipMemsetAsync(v->d_act, 0, n * DRONE_ACT_DIM * sizeof(float), v->stream));
            INIT_TRY(hipMemsetAsync(v->d_rew, 0, n * sizeof(float), v->stream));
            INIT_TRY(hipMemsetAsync(v->d_term, 0, n, v->stream));
            INIT_TRY(hipMemsetAsync(v->d_trunc, 0, n, v->stream));
            observations = v->d_obs; actions = v->d_act; rewards = v->d_rew; terminals = v->d_term; truncations = v->d_trunc;
            v->u_obs = observations; v->u_act = actions; v->u_rew = rewards; v->u_term = terminals; v->u_trunc = truncations;
        }
        if ((reinterpret_cast<uintptr_t>(observations) & 15u) || (reinterpret_cast<uintptr_t>(actions) & 15u) || (reinterpret_cast<uintptr_t>(rewards) & 3u)) {
            set_err("device buffers must be 16-byte aligned (observations, actions) and 4-byte aligned (rewards)");
            drone_vec_close(v);
            return nullptr;
        }
        v->dv.obs = observations; v->dv.act = actions; v->dv.rew = rewards; v->dv.term = terminals; v->dv.trunc = truncations;
    }
    v->dv.n = (uint32_t)num_envs;
    v->dv.n_pad = v->n_pad;
    v->dv.stride = v->stride;
    {   // Partial-line plane updates (ended episodes) cost an HBM read-modify-write each unless the Infinity Cache absorbs
        // them: widen them to whole lines once a step touches more than twice its 256 MiB (measured cross-over between
        // 2^20 and 2^22 envs, profiles/r02_ab/). DRONE_LINE_COMPLETE=0/1 forces it.
        const char* e = getenv("DRONE_LINE_COMPLETE");
        const size_t touched = n * (sizeof(float4) * (dt ? 2 * hot_planes(cfg->task, true) : 2 * hot_planes(cfg->task) - 1) + (size_t)drone_obs_dim(cfg->task) * 4 + 16 + 6);
        v->dv.line_complete = (e && *e) ? (atoi(e) != 0) : (touched > ((size_t)512 << 20));
        // Sweep order of the step kernel, by the same footprint (DRONE_SWEEP_ORDER=0..15 forces it): up to ~1.5x the
        // Infinity Cache, one contiguous eighth per XCD (-2.7 % at 2^20 envs); beyond, one global round-robin sweep
        // that, in some bands, turns around on odd steps (the tail of one step is the head of the next and is still
        // cached: -6 % at 2^22 envs in round 2) with the action rows streamed (bit 2: -2 % there).
        // Round 4, hover task in the derived-target layout (where the band was swept end to end, profiles/r04_ab/band_*.txt,
        // upper_*.txt): between ~400 MiB and ~1.1 GiB touched per step — a step that is one to four times the Infinity
        // Cache, the LRU worst case — the STATE loads carry the non-temporal hint (bit 3) on a plain round-robin sweep:
        // -3 % at 426 MiB, -10 ... -11.5 % from 458 to 655 MiB (2^21 envs: 97.6 -> 87.9 us), -10 % at 786 MiB (where the
        // old rule's reversed sweep was 12 % behind a plain one), -6 % at 917 MiB, -1.8 % at 2^22 envs; beyond ~1.1 GiB the
        // reversed sweep with streamed action rows wins again (+2 ... +4 % for the hint from 4.5 M envs on). The hint costs
        // +21 % at 2^20 envs, hence a band and not a switch; and it is the hover task's band only: the seven-plane tasks
        // lose 4 % to it at the same footprints (waypoint 1.75 M envs, race 2 M), the swarm task prefers order 6 there.
        const char* o = getenv("DRONE_SWEEP_ORDER");
        const bool hover_dt = cfg->task == DRONE_TASK_HOVER && dt;
        // The other tasks, and the bands around it (round 4 re-sweep of round 2's thresholds, profiles/r04_ab/sweep_*.txt:
        // waypoint / swarm / race at 1.3 ... 4.2 M envs, orders 0 1 2 6 8): the reversed sweep pays between ~450 and ~600 MiB
        // (-4 ... -10 % against a plain sweep) and again beyond ~900 MiB, but LOSES to a plain sweep in between, most at
        // about three times the Infinity Cache (775 ... 835 MiB: +3 ... +12 %, all four tasks) — the old rule switched to it
        // at 768 MiB, exactly there.
        const size_t mib = touched >> 20;
        const size_t rev_to = cfg->task == DRONE_TASK_SWARM ? 720 : 600;  // the swarm task keeps the reversed sweep's band longer (625 MiB: 103.2 against 116.1 us plain; 695: 124.3 / 130.4; 764: plain wins)
        uint32_t order = mib <= 400 ? 1u : mib <= 450 ? 0u : mib <= rev_to ? 6u : mib <= 900 ? 0u : 6u;
        if (hover_dt && mib > 400 && mib <= 1100) order = 8u;
        v->dv.order = (o && *o) ? (uint32_t)atoi(o) : order;
        // DRONE_AUTOTUNE=1 (opt-in since round 6): beyond 400 MiB the handle's own steps 161 ... 256 time the table's entry and
        // its neighbours on this box, on its own buffers, under its own workload (SweepTune), and the fastest becomes the
        // handle's order. Round 5 ran this by default and logged 12 of 12 cases on three boxes in which the measurement picked
        // exactly the table's entry, at a cost of 0.1 ... 0.8 % (profiles/r05_ab/autotune_online_box*.txt):
        // the table stands, the measurement is there for a box or a workload someone has reason to distrust it on. Not for a
        // forced order or host buffers (PCIe-bound at these sizes).
        const char* at = getenv("DRONE_AUTOTUNE");
        v->touched_mib = mib;
        if (mib > 400 && !(o && *o) && !v->host_buffers && at && *at && atoi(at) != 0) {
            v->tune = new (std::nothrow) SweepTune();
            if (v->tune) {
                v->tune->table = order;
                const uint32_t all[3] = {0u, 6u, 8u};
                v->tune->cand[v->tune->nc++] = order;
                for (uint32_t c : all)
                    if (c != order) v->tune->cand[v->tune->nc++] = c;
            }
        }
    }
    {   // Packed-f32 RK4 in the register-resident kernels (fused rollout, step_many): wins only while a SIMD holds ONE
        // wave (<= 65 536 envs on the 1024 SIMDs: rollout -8.8 %, step_many -3.9 %; waypoint / race -3...4 %), where the
        // issue rate of one wave is the limit and a packed instruction costs 1.1x a scalar one for two results. With two
        // waves per SIMD it already loses (131 072 envs: +3.3 % / +3.9 %), on a full chip clearly (2^20: rollout +11 %) —
        // profiles/r03_ab/ab_pk_*.txt. DRONE_PACKED_RK4=0/1 forces it.
        const char* e = getenv("DRONE_PACKED_RK4");
        v->dv.packed_rk4 = (e && *e) ? (atoi(e) != 0) : (v->n_pad <= 65536u);
    }
    v->dv.kp = v->d_kp;
    v->dv.kp_host = &v->kp;
    if (!upload_params(v)) { drone_vec_close(v); return nullptr; }
    write_variant(v, nullptr);
#undef INIT_TRY
    if (debug_reg()) fprintf(stderr, "[drone reg] init %p n=%d %s%s obs=%p act=%p planes=%p\n", (void*)v, v->n, v->host_buffers ? "host" : "device", v->zero_copy ? " zero-copy" : "", (void*)observations, (void*)actions, (void*)v->dv.planes);
    return v;
}

void drone_vec_reset(DroneVec* v, uint64_t seed) {
    Entry in(v);
    if (!in || !idle(v, "reset")) return;
    v->seed = seed;
    v->gstep = 0;
    v->step_launches = 0;  // the reset kernel zeroes both done-count slots
    v->list_valid = false;
    v->many_k = 0;
    LaunchSig sig = {nullptr, nullptr, nullptr, nullptr, 0u, 0u, 0u, 0u};
    if (!upload_params(v) || !push_counters(v) || !peer_before_launch(v, &sig)) return;
    HIP_TRY(launch_reset(v->dv, v->cfg.task, v->stream, &sig), { peer_launch_failed(v); return; });
    if (v->host_buffers) device_to_host_outputs(v);
}

namespace {
bool step_send_impl(DroneVec* v) {
#if DRONE_HOST_STAMPS
    g_stamps.enter();
#endif
    if (v->host_buffers && !host_to_device_actions(v)) return false;
    HOST_STAMP(kActionsIn);
    LaunchSig sig = {nullptr, nullptr, nullptr, nullptr, 0u, 0u, 0u, 0u};
    if (!peer_before_launch(v, &sig)) return false;
    const bool copy_out = v->host_buffers && v->zero_copy && v->threaded && v->h_flag;  // (without the completion flag the plain wait + whole copy is used)
    if (copy_out) {
        sig.wg_done = v->d_wg_done;
        sig.wg_done_value = ++v->wg_seq;
    }
    SweepTune::Pair* timing = v->tune ? tune_before_step(v) : nullptr;  // an HBM-bound handle still measuring its sweep order (may set dv.order for this launch)
    HIP_TRY(launch_step(v->dv, v->cfg.task, v->gstep, v->step_launches & 1u, v->stream, &sig), { peer_launch_failed(v); return false; });
    HOST_STAMP(kLaunched);
    if (timing && hipEventRecord(timing->e1, v->stream) != hipSuccess) {  // (v->tune is still there: a pair is only handed out while measuring)
        (void)hipGetLastError();
        timing->cand = -1;
        v->tune->in_flight -= 1;
    }
    v->gstep += 1;
    v->step_launches += 1;
    v->list_valid = true;
    v->many_k = 0;
    if (!v->host_buffers) return true;
    if (!enqueue_host_outputs(v)) return false;
    if (copy_out) {  // the helpers start following the chunks' words now; the caller joins in finish_host_outputs
        __atomic_store_n(&v->stream_idle, 0, __ATOMIC_RELEASE);
        __atomic_store_n(&v->copy_abort, 0, __ATOMIC_RELEASE);
        v->copy_started = start_threaded_copy(v);  // (busy with another handle's step: this one's outputs are copied after the wait, by this thread)
        HOST_STAMP(kPoolStarted);
    }
    return true;
}
}  // namespace

void drone_vec_step(DroneVec* v) {
    Entry in(v);
    if (!in || !idle(v, "step")) return;
    if (step_send_impl(v) && v->host_buffers) finish_host_outputs(v);
}

void drone_vec_step_send(DroneVec* v) {
    Entry in(v);
    if (!in || !idle(v, "step_send")) return;
    if (step_send_impl(v)) v->pending = true;
}

void drone_vec_step_recv(DroneVec* v) {
    Entry in(v);
    if (!in) return;
    if (!v->pending) { set_err("step_recv: no step was sent (drone_vec_step_send)"); return; }
    v->pending = false;
    if (v->host_buffers) finish_host_outputs(v);
}

void drone_vec_rollout(DroneVec* v, int horizon) {
    Entry in(v);
    if (!in || !idle(v, "rollout")) return;
    if (horizon <= 0) { set_err("rollout: horizon must be positive, got %d", horizon); return; }
    LaunchSig sig = {nullptr, nullptr, nullptr, nullptr, 0u, 0u, 0u, 0u};
    if (!peer_before_launch(v, &sig)) return;
    HIP_TRY(launch_rollout(v->dv, v->cfg.task, v->gstep, (uint32_t)horizon, v->stream, &sig), { peer_launch_failed(v); return; });
    v->gstep += (uint32_t)horizon;
    v->list_valid = false;  // the fused rollout builds no done-id list
    v->many_k = 0;
    if (v->host_buffers) device_to_host_outputs(v);
}

namespace {

// grow the K-dependent device storage of drone_vec_step_many (drained first: an earlier launch may still use the old blocks)
bool many_reserve(DroneVec* v, int k_steps, bool need_staging) {
    const size_t n = (size_t)v->n, od = (size_t)drone_obs_dim(v->cfg.task), K = (size_t)k_steps;
    if (v->cfg.compact_done && k_steps > v->many_cap) {
        HIP_TRY(hipStreamSynchronize(v->stream), return false);
        (void)hipFree(v->many_ids); v->many_ids = nullptr;
        (void)hipFree(v->many_count); v->many_count = nullptr;
        v->many_cap = 0;
        HIP_TRY(hipMalloc((void**)&v->many_ids, sizeof(uint32_t) * K * n), return false);
        HIP_TRY(hipMalloc((void**)&v->many_count, sizeof(uint32_t) * K), return false);
        v->many_cap = k_steps;
    }
    if (v->host_buffers && need_staging && k_steps > v->stage_cap) {
        HIP_TRY(hipStreamSynchronize(v->stream), return false);
        (void)hipFree(v->s_act); (void)hipFree(v->s_obs); (void)hipFree(v->s_rew); (void)hipFree(v->s_term); (void)hipFree(v->s_trunc);
        v->s_act = v->s_obs = v->s_rew = nullptr; v->s_term = v->s_trunc = nullptr;
        v->stage_cap = 0;
        HIP_TRY(hipMalloc((void**)&v->s_act, K * n * DRONE_ACT_DIM * sizeof(float)), return false);
        HIP_TRY(hipMalloc((void**)&v->s_obs, K * n * od * sizeof(float)), return false);
        HIP_TRY(hipMalloc((void**)&v->s_rew, K * n * sizeof(float)), return false);
        HIP_TRY(hipMalloc((void**)&v->s_term, K * n), return false);
        HIP_TRY(hipMalloc((void**)&v->s_trunc, K * n), return false);
        v->stage_cap = k_steps;
    }
    return true;
}

}  // namespace

namespace {
void step_many_impl(DroneVec* v, int k_steps, const float* actions, bool repeat, float* observations, float* rewards,
                    unsigned char* terminals, unsigned char* truncations);
}

void drone_vec_step_many(DroneVec* v, int k_steps, const float* actions, float* observations, float* rewards,
                         unsigned char* terminals, unsigned char* truncations) {
    step_many_impl(v, k_steps, actions, false, observations, rewards, terminals, truncations);
}

void drone_vec_step_repeat(DroneVec* v, int k_steps, const float* actions, float* observations, float* rewards,
                           unsigned char* terminals, unsigned char* truncations) {
    if (!actions) { Entry in(v); if (in) set_err("step_repeat: actions is NULL (the in-kernel policy is drone_vec_step_many with actions = NULL)"); return; }
    step_many_impl(v, k_steps, actions, true, observations, rewards, terminals, truncations);
}

namespace {

// `repeat`: `actions` is ONE [N][4] block applied to all k_steps steps (action repeat / frame skip)
void step_many_impl(DroneVec* v, int k_steps, const float* actions, bool repeat, float* observations, float* rewards,
                    unsigned char* terminals, unsigned char* truncations) {
    Entry in(v);
    if (!in || !idle(v, "step_many")) return;
    if (k_steps < 1) { set_err("step_many: k_steps must be positive, got %d", k_steps); return; }
    if (!observations || !rewards || !terminals || !truncations) { set_err("step_many: NULL output block"); return; }
    if (v->gather && v->gather->peer) {  // it writes the caller's blocks, not this rank's rows of the root's batch, and runs outside the handshake
        set_err("step_many / step_repeat: not while the peer-store exchange is active (the root's batch holds one step per round; drone_vec_gather_close first)");
        return;
    }
    const size_t n = (size_t)v->n, od = (size_t)drone_obs_dim(v->cfg.task), K = (size_t)k_steps;
    // Host handles: blocks the caller pinned beforehand (drone_host_pin, hipHostMalloc, hipHostRegister) are accessed by
    // the kernel in place over PCIe — no staging, no copy commands; anything else goes through device staging and DMA.
    const float* d_act = actions;
    float* d_obs = observations;
    float* d_rew = rewards;
    unsigned char* d_term = terminals;
    unsigned char* d_trunc = truncations;
    bool direct = false;
    if (v->host_buffers) {
        const char* zc = getenv("DRONE_HOST_ZEROCOPY");
        const size_t act_bytes = (repeat ? 1 : K) * n * DRONE_ACT_DIM * sizeof(float);
        if (!(zc && *zc && atoi(zc) == 0) && (!actions || already_pinned(actions, act_bytes)) && already_pinned(observations, K * n * od * sizeof(float)) &&
            already_pinned(rewards, K * n * sizeof(float)) && already_pinned(terminals, K * n) && already_pinned(truncations, K * n)) {
            void* m[5] = {actions ? mapped_ptr(const_cast<float*>(actions)) : nullptr, mapped_ptr(observations), mapped_ptr(rewards), mapped_ptr(terminals), mapped_ptr(truncations)};
            direct = (!actions || m[0]) && m[1] && m[2] && m[3] && m[4] && !(reinterpret_cast<uintptr_t>(m[0]) & 15u) && !(reinterpret_cast<uintptr_t>(m[1]) & 15u) &&
                     !(reinterpret_cast<uintptr_t>(m[2]) & 3u);
            if (direct) {
                d_act = static_cast<const float*>(m[0]); d_obs = static_cast<float*>(m[1]); d_rew = static_cast<float*>(m[2]);
                d_term = static_cast<unsigned char*>(m[3]); d_trunc = static_cast<unsigned char*>(m[4]);
            }
        }
    }
    if (!many_reserve(v, k_steps, !direct)) return;
    if (v->host_buffers && !direct) {
        d_act = actions ? v->s_act : nullptr;
        d_obs = v->s_obs; d_rew = v->s_rew; d_term = v->s_term; d_trunc = v->s_trunc;
        if (actions) HIP_TRY(hipMemcpyAsync(v->s_act, actions, (repeat ? 1 : K) * n * DRONE_ACT_DIM * sizeof(float), hipMemcpyHostToDevice, v->stream), return);
    } else if (!v->host_buffers && ((reinterpret_cast<uintptr_t>(observations) & 15u) || (reinterpret_cast<uintptr_t>(actions) & 15u) || (reinterpret_cast<uintptr_t>(rewards) & 3u))) {
        set_err("step_many: device blocks must be 16-byte aligned (observations, actions) and 4-byte aligned (rewards)");
        return;
    }
    if (v->cfg.compact_done) HIP_TRY(hipMemsetAsync(v->many_count, 0, sizeof(uint32_t) * K, v->stream), return);
    HIP_TRY(launch_step_many(v->dv, v->cfg.task, v->gstep, (uint32_t)k_steps, d_act, repeat ? 0u : (uint32_t)v->n, d_obs, d_rew, d_term, d_trunc,
                             v->cfg.compact_done ? v->many_ids : nullptr, v->cfg.compact_done ? v->many_count : nullptr, v->stream), return);
    v->gstep += (uint32_t)k_steps;
    v->list_valid = false;
    v->many_k = k_steps;
    if (v->host_buffers && direct) {
        ensure_flag(v);
        (void)wait_zero_copy(v);
    } else if (v->host_buffers) {
        HIP_TRY(hipMemcpyAsync(observations, v->s_obs, K * n * od * sizeof(float), hipMemcpyDeviceToHost, v->stream), return);
        HIP_TRY(hipMemcpyAsync(rewards, v->s_rew, K * n * sizeof(float), hipMemcpyDeviceToHost, v->stream), return);
        HIP_TRY(hipMemcpyAsync(terminals, v->s_term, K * n, hipMemcpyDeviceToHost, v->stream), return);
        HIP_TRY(hipMemcpyAsync(truncations, v->s_trunc, K * n, hipMemcpyDeviceToHost, v->stream), return);
        HIP_TRY(hipStreamSynchronize(v->stream), return);
    }
}

}  // namespace

void drone_vec_log(DroneVec* v, DroneLog* out) {
    if (!out) return;
    memset(out, 0, sizeof(*out));
    Entry in(v);
    if (!in || !idle(v, "log")) return;
    int grid = 0;
    HIP_TRY(launch_log_reduce(v->dv, v->d_partials, kLogMaxGrid, &grid, v->stream), return);
    HIP_TRY(hipMemcpyAsync(v->h_partials, v->d_partials, sizeof(double) * 6 * grid, hipMemcpyDeviceToHost, v->stream), return);
    HIP_TRY(hipStreamSynchronize(v->stream), return);
    double s[6] = {0, 0, 0, 0, 0, 0};
    for (int b = 0; b < grid; b++)
        for (int k = 0; k < 6; k++) s[k] += v->h_partials[b * 6 + k];
    const double n = s[4];
    if (n > 0) {
        // SPEC.md section 8 (v5): hover / swarm episodes log the count of steps within hover_radius; reported per step flown
        const bool per_step = v->cfg.task == DRONE_TASK_HOVER || v->cfg.task == DRONE_TASK_SWARM;
        out->perf = (float)(s[0] / (per_step ? s[3] : n));
        out->score = (float)(s[1] / (per_step ? s[3] : n));
        out->episode_return = (float)(s[2] / n);
        out->episode_length = (float)(s[3] / n);
        out->oob = (float)(s[5] / n);
    }
    out->n = (float)n;
}

void drone_vec_close(DroneVec* v) {
    if (!v) return;
    if (debug_reg()) fprintf(stderr, "[drone reg] close %p\n", (void*)v);
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; }
    (void)hipSetDevice(v->device);
    if (v->stream) (void)hipStreamSynchronize(v->stream);
    if (v->copy_started) (void)finish_threaded_copy(v);  // closed between step_send and step_recv: the pool must let go of this handle first
    tune_free(v);
    gather_destroy(v);
    if (v->h_wg_done) (void)hipHostFree(v->h_wg_done);
    for (int i = 0; i < 5; i++)
        if (v->registered[i]) host_unregister(v->registered_ptr[i], v, "caller buffer at close");
    for (int i = 0; i < v->n_pinned_blocks; i++) host_unregister(v->pinned_blocks[i], v, "drone_vec_host_pin block at close");
    v->n_pinned_blocks = 0;
    (void)hipFree(v->dv.planes);
    (void)hipFree(v->dv.cold);
    (void)hipFree(v->d_kp);
    (void)hipFree(v->dv.pad_sink);
    (void)hipFree(v->dv.ctr);
    (void)hipFree(v->dv.stamps);
    (void)hipFree(v->d_partials);
    if (v->h_partials) (void)hipHostFree(v->h_partials);
    if (v->h_flag) (void)hipHostFree(const_cast<uint32_t*>(v->h_flag));
    drop_bounce(v);
    (void)hipFree(v->dv.done_ids);
    (void)hipFree(v->dv.done_count);
    (void)hipFree(v->many_ids);
    (void)hipFree(v->many_count);
    (void)hipFree(v->s_act);
    (void)hipFree(v->s_obs);
    (void)hipFree(v->s_rew);
    (void)hipFree(v->s_term);
    (void)hipFree(v->s_trunc);
    (void)hipFree(v->d_obs);
    (void)hipFree(v->d_act);
    (void)hipFree(v->d_rew);
    (void)hipFree(v->d_term);
    (void)hipFree(v->d_trunc);
    if (v->ev0) (void)hipEventDestroy(v->ev0);
    if (v->ev1) (void)hipEventDestroy(v->ev1);
    if (v->own_stream && v->stream) (void)hipStreamDestroy(v->stream);
    if (prev >= 0 && prev != v->device) (void)hipSetDevice(prev);
    delete v;
}

int drone_vec_set_stream(DroneVec* v, void* hip_stream) {
    if (!v) return -1;
    if ((hipStream_t)hip_stream == v->stream && !v->own_stream) return 0;  // cheap to call every step
    Entry in(v);
    if (!in || !idle(v, "set_stream")) return -1;
    HIP_TRY(hipStreamSynchronize(v->stream), return -1);
    if (v->own_stream && v->stream) (void)hipStreamDestroy(v->stream);
    v->stream = (hipStream_t)hip_stream;
    v->own_stream = false;
    return 0;
}

int drone_vec_sync(DroneVec* v) {
    Entry in(v);
    if (!in) return -1;
    HIP_TRY(hipStreamSynchronize(v->stream), return -1);
    return 0;
}

int drone_vec_fill_random_actions(DroneVec* v, float* actions, uint32_t gstep) {
    Entry in(v);
    if (!in || !idle(v, "fill_random_actions")) return -1;
    if (!actions) { set_err("fill_random_actions: NULL buffer"); return -1; }
    if (v->host_buffers && v->zero_copy && actions == v->u_act) {
        // the bound action buffer is mapped (itself or through its stand-in): the kernel writes it over PCIe, no copy command
        HIP_TRY(launch_fill_actions(v->dv, v->m_act, gstep, v->stream), return -1);
        if (!wait_zero_copy(v)) return -1;
        if (v->bounce[1]) memcpy(v->u_act, v->bounce[1], v->bounce_bytes[1]);
    } else if (v->host_buffers) {
        // generate on the device into the action mirror, then hand the host its copy
        HIP_TRY(launch_fill_actions(v->dv, v->d_act, gstep, v->stream), return -1);
        HIP_TRY(hipMemcpyAsync(actions, v->d_act, (size_t)v->n * DRONE_ACT_DIM * sizeof(float), hipMemcpyDeviceToHost, v->stream), return -1);
        HIP_TRY(hipStreamSynchronize(v->stream), return -1);
    } else {
        if (reinterpret_cast<uintptr_t>(actions) & 15u) { set_err("actions must be 16-byte aligned"); return -1; }
        HIP_TRY(launch_fill_actions(v->dv, actions, gstep, v->stream), return -1);
    }
    return 0;
}

uint32_t drone_vec_gstep(const DroneVec* v) {
    if (!v) return 0u;
    if (v->dv.ctr) {  // graph-safe stepping: the device owns the counter (replays advance it without a host call)
        Entry in(v);
        if (in) (void)pull_counters(const_cast<DroneVec*>(v));
    }
    return v->gstep;
}

int drone_vec_set_gstep(DroneVec* v, uint32_t gstep) {
    Entry in(v);
    if (!in || !idle(v, "set_gstep")) return -1;
    if (!pull_counters(v)) return -1;  // keep the device's step-launch count
    v->gstep = gstep;
    v->list_valid = false;
    return push_counters(v) ? 0 : -1;
}

int drone_vec_enable_graph_capture(DroneVec* v, int on) {
    Entry in(v);
    if (!in || !idle(v, "enable_graph_capture")) return -1;
    if (v->host_buffers) { set_err("graph-safe stepping needs device buffers (host-buffer steps end in a stream sync, which cannot be captured)"); return -1; }
    if (on && v->gather && v->gather->peer) {  // the round numbers of the handshake are launch arguments kept by the host; a replay would wait for / publish a stale round
        set_err("graph-safe stepping cannot be combined with the peer-store exchange (its handshake's round numbers are host state baked into each launch); drone_vec_gather_close first");
        return -1;
    }
    if (on && !v->dv.ctr) {
        HIP_TRY(hipMalloc((void**)&v->dv.ctr, 3 * sizeof(uint32_t)), return -1);
        if (!push_counters(v)) return -1;
    } else if (!on && v->dv.ctr) {
        if (!pull_counters(v)) return -1;
        HIP_TRY(hipStreamSynchronize(v->stream), return -1);
        (void)hipFree(v->dv.ctr);
        v->dv.ctr = nullptr;
    }
    return 0;
}

int drone_vec_status(const DroneVec* v) { return v ? v->status : -1; }
const char* drone_vec_status_message(const DroneVec* v) { return v ? v->status_msg : "handle is NULL"; }
void drone_vec_clear_status(DroneVec* v) {
    if (!v) return;
    v->status = 0;
    v->status_msg[0] = 0;
}
int drone_vec_num_envs(const DroneVec* v) { return v ? v->n : 0; }

}  // extern "C"

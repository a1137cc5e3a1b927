/*
 * drone_vec.h — C-ABI of the MI355X-native vectorised drone environment.
 *
 * This is the drop-in boundary: the entry points a PufferLib
 * ocean-env binding would bind for the drone env's vec path — init / reset /
 * step / log / close over caller-owned observation / action / reward /
 * terminal / truncation buffers (BASELINE.json north_star: "PufferLib C env
 * API (init/reset/step, obs/action/reward/done buffers)").
 *
 * Reference interface each entry point replaces: NONE CAN BE CITED. The
 * reference snapshot has no binding source — the `pufferlib` submodule is an
 * empty directory (/root/reference/.gitmodules:1-3) and only an older Cython
 * shim is hinted at (/root/reference/.gitignore:14, `simulator/cy_env.c`).
 * The shape below follows the north-star sentence; see INTEGRATION.md for the
 * binding stub a maintainer would add on the PufferLib side.
 *
 * Plain C: pointers and sizes only, no torch / HIP types in any signature
 * (a HIP stream crosses as `void*`). Behaviour is defined by SPEC.md.
 * The library is HIP-only: every entry point fails loudly (NULL / non-zero,
 * message via drone_last_error()) when no gfx950 device is usable. There is
 * no CPU fallback.
 */
#ifndef DRONE_VEC_H
#define DRONE_VEC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DRONE_OBS_DIM 20     /* observation floats per env row, tasks 0 and 1 */
#define DRONE_OBS_DIM_MAX 24 /* tasks 2 and 3 append four floats: nearest neighbour / gate normal and plane distance */
#define DRONE_ACT_DIM 4

#define DRONE_TASK_HOVER 0
#define DRONE_TASK_WAYPOINT 1 /* waypoint tracking with OU wind gusts */
#define DRONE_TASK_SWARM 2    /* agents_per_env drones per env, coupled by a nearest-neighbour term (SPEC.md §10) */
#define DRONE_TASK_RACE 3     /* fly through a sequence of ring gates (SPEC.md §11) */

#define DRONE_BUFFERS_HOST 0   /* caller buffers are host memory. Buffers the caller vouches for (host_pages_exclusive: mappings of their own) are pinned + mapped and the kernel accesses them over PCIe; others get pinned stand-ins (small and mid-size shards) or go through H2D, kernel, D2H copies; step ends with a sync */
#define DRONE_BUFFERS_DEVICE 1 /* caller buffers are HBM on `device`: zero-copy, async on the stream */

/* Env kwargs. Fill with drone_config_default() first, then override. */
typedef struct DroneConfig {
    uint32_t struct_size; /* = sizeof(DroneConfig); checked at init */
    int32_t task;         /* DRONE_TASK_* */
    int32_t buffer_kind;  /* DRONE_BUFFERS_* */
    int32_t device;       /* HIP device ordinal */
    uint32_t env_offset;  /* global id of local env 0 (sharding; SPEC.md §2) */
    int32_t horizon;
    int32_t substeps;
    int32_t compact_done; /* 1: also build the compacted done-id list each step */
    int32_t agents_per_env; /* task 2: power of two in [1, 64]; num_envs and env_offset must be multiples of it */
    float dt;
    float mass, arm, ixx, iyy, izz;
    float k_thrust, k_torque, k_drag, k_ang_damp, gravity;
    float max_rpm, motor_tau, max_vel, max_omega;
    float bound, spawn_extent, target_extent, tilt_init;
    float hover_radius, waypoint_radius;
    float wind_theta, wind_sigma, wind_max;
    float c_omega, c_action, crash_penalty, progress_scale, waypoint_bonus;
    float collision_radius, proximity_radius, c_proximity; /* task 2 */
    float gate_radius;                                      /* task 3 */
    /* Host buffers only. 1: the caller guarantees that each of the five buffers is a MAPPING OF ITS OWN — mmap, POSIX shm, a
     * System V segment — that starts on a 4 KiB page boundary, with nothing else in its pages up to the end of its last page,
     * and that stays mapped while the handle lives. Such buffers are pinned and mapped for the zero-copy transport. 0
     * (default): a buffer is used in place only if the caller pinned it itself (hipHostMalloc / hipHostRegister); everything
     * else gets pinned stand-ins or is copied through pageable transfers and is never registered. Why the caller has to say
     * so: on ROCm 7, hipHostRegister / hipHostUnregister of a range that SHARES a page with other heap memory breaks the
     * runtime's own on-the-fly pinning of pageable copy destinations on that page ("Memory access fault by GPU ... on address
     * <heap address>", tools/debug/pageable_copy_stress.py reproduces it without this library) — and (round 5) a block that
     * owns its pages but lies INSIDE the malloc heap (posix_memalign, an array that happens to be page-aligned) faults the
     * same way once the heap around it is trimmed or reused while the GPU writes it
     * (tools/debug/heap_interior_registration_stress.py); a mapping of its own never does. Rounds 3-4 registered any
     * page-aligned whole-page buffer unasked; the library cannot tell the two kinds apart, so it no longer does. */
    int32_t host_pages_exclusive;
    /* How the device keeps the state (round 4; ADVICE r3: the choice is part of the handle's declared contract, not only of
     * its size). DRONE_LAYOUT_AUTO (0, default): hover / swarm handles whose step is HBM-bound (about 2^19 envs on) use the
     * derived-target layout — no target plane, the target re-derived from (reset key, env, episode), 262 instead of 278
     * bytes per env-step for hover — smaller ones keep the target plane. DRONE_LAYOUT_TARGET_PLANE / _DERIVED_TARGET force
     * one (derived-target needs task 0 or 2 and horizon <= 65535; init fails otherwise). What differs for the caller:
     * drone_vec_set_state on a derived-target handle refuses rows whose target is not the one SPEC.md section 6 draws for
     * (env, episode) or whose counters exceed 65535; get_state, checkpoints and every trajectory are identical. */
    int32_t state_layout;
} DroneConfig;

#define DRONE_LAYOUT_AUTO 0
#define DRONE_LAYOUT_TARGET_PLANE 1
#define DRONE_LAYOUT_DERIVED_TARGET 2

/* Aggregated episode statistics since the previous drone_vec_log (SPEC.md §8). */
typedef struct DroneLog {
    float perf;
    float score;
    float episode_return;
    float episode_length;
    float oob; /* fraction of episodes that ended by leaving the box */
    float n;   /* episodes aggregated */
} DroneLog;

/* One env's full state as an AoS row — test / checkpoint interface only;
 * the device keeps state as float4 planes. */
typedef struct DroneStateRow {
    float pos[3], vel[3], quat[4], omega[3], rpm[4];
    float target[3], wind[3]; /* task 3: target = gate centre, wind = gate normal */
    float ep_return;
    uint32_t tick, episode, score_count;
    float perf_sum, score_sum, ret_sum, len_sum, n_sum, oob_sum;
} DroneStateRow;

typedef struct DroneVec DroneVec;

void drone_config_default(DroneConfig* cfg, int task);

/* HIP devices visible to this process (0 if none / HIP unusable); a multi-process host maps ranks onto them. */
int drone_device_count(void);

/* Algorithmic HBM bytes the per-step kernel moves per env and step for THIS handle (task and state layout: hover 278, or
 * 262 when the handle uses the derived-target layout; waypoint / race 310; swarm 294 / 278) — what bench.py's roofline uses. */
int drone_vec_bytes_per_env_step(const DroneVec* v);

/* Which instantiation of the per-step kernel this handle launches and the per-handle launch choices, as text:
 * "drone_step_kernel<task=0,compact=0,mem=0,dt=1> order=1 line_complete=0 packed_rk4=0 bytes=262" — task, done-id
 * compaction, which loads carry the non-temporal hint (bit 0 the action rows, bit 1 the state planes), derived-target
 * layout (the four template arguments of drone_step_kernel), the sweep order word (DeviceView::order), whole-line
 * widening of rare plane updates, packed-f32 RK4 in the register-resident kernels, and the algorithmic bytes per
 * env-step. Tests use it to assert that the sizes bench.py times run the instantiations the parity suite covers
 * (tests/test_configs_gpu.py). The string lives in the handle. */
const char* drone_vec_variant(const DroneVec* v);

/* How host-buffer steps of this handle move their data: 1 = zero-copy (the kernel reads / writes the caller's pinned
 * buffers over PCIe); 2 = zero-copy through pinned stand-ins the library owns for those of the five buffers that could
 * not be pinned themselves, copied to / from the caller's memory on the host around each step (small shards only: up to
 * DRONE_HOST_POOL_MIN_BYTES, default 512 KiB of such buffers — DRONE_HOST_BOUNCE_MAX_BYTES, default 1 MiB, in a process
 * without the pool); 3 (round 5) = the same stand-ins for mid-size shards (up to
 * DRONE_HOST_MT_MAX_BYTES, default 64 MiB of unpinnable buffers — a vec-env worker's unaligned shared-memory slices at
 * 5 000 ... ~10^5 envs), moved by a small pool of host threads (DRONE_HOST_COPY_THREADS per job, caller included; default
 * half the machine's hardware threads, at most 8; 1 = off): the action rows go in as parallel slices, and the outputs come out WHILE the step kernel is still writing
 * over PCIe — each 256-drone chunk as soon as its workgroup says its rows have landed. The pool is one per process, started
 * on first use; its threads spin for ~200 us after a job and sleep otherwise; it serves one handle at a time (from
 * drone_vec_step_send to drone_vec_step_recv it belongs to that handle): a second handle stepped meanwhile does its own
 * copying on the calling thread. Do not fork() a process that has started it. 0 = device mirrors and DMA copies; -1 for device-buffer handles. */
int drone_vec_host_transport(const DroneVec* v);

/* Floats per observation row for a task: 20, or 24 for DRONE_TASK_SWARM and DRONE_TASK_RACE. */
int drone_obs_dim(int task);

/* observations [N][drone_obs_dim(task)] f32, actions [N][4] f32, rewards [N] f32,
 * terminals [N] u8, truncations [N] u8 — owned by the caller, never freed
 * here. Returns NULL on failure (see drone_last_error). Does not reset.
 * A DRONE_BUFFERS_DEVICE handle may pass all five as NULL: the library then allocates them (zeroed) in HBM on
 * cfg.device, frees them at close, and drone_vec_buffers hands out the addresses — for consumers that wrap device
 * memory they did not allocate (DLPack: bindings/drone_binding.c vec_dlpack). */
DroneVec* drone_vec_init(float* observations, float* actions, float* rewards,
                         unsigned char* terminals, unsigned char* truncations,
                         int num_envs, uint64_t seed, const DroneConfig* cfg);

/* Start every env's first episode from `seed`; writes observations. */
void drone_vec_reset(DroneVec* v, uint64_t seed);

/* Read `actions`, advance every env one step, overwrite observations /
 * rewards / terminals / truncations; finished envs auto-reset (SPEC.md §5). */
void drone_vec_step(DroneVec* v);

/* drone_vec_step in two halves, for a caller that has other work to do while the env steps (a vec-env's async
 * send / recv; two handles stepping out of phase so that one's PCIe transfers overlap the other's kernel):
 * step_send reads `actions` and enqueues the step and everything that can follow it on the stream; step_recv waits and
 * delivers the outputs into the caller's buffers. send + recv = step, bit for bit. Between the two, the caller must
 * not touch the five buffers, and every other call on the handle except drone_vec_sync / status / close fails with
 * "not received". Device-buffer handles: send is the (asynchronous) step, recv returns at once. */
void drone_vec_step_send(DroneVec* v);
void drone_vec_step_recv(DroneVec* v);

/* Fused rollout: `horizon` steps under the device-side random policy with
 * state held in registers; outputs written once at the horizon (SPEC.md §9).
 * `actions` is neither read nor written. */
void drone_vec_rollout(DroneVec* v, int horizon);

/* K env steps in ONE launch, with the per-step outputs of every step (round 3). Exactly what
 * `k_steps` calls of drone_vec_step would do — same trajectories, same auto-resets inside each step — except that the
 * caller stages the K action rows up front and receives K steps' outputs in K-major blocks:
 *   actions      [K][N][4] f32, or NULL: the SPEC.md §2 random policy is drawn in the kernel (as drone_vec_rollout does)
 *   observations [K][N][drone_obs_dim(task)] f32, rewards [K][N] f32, terminals / truncations [K][N] u8
 * all of the handle's buffer kind (device kind: 16-byte aligned observations / actions, written asynchronously on the
 * stream; host kind: blocks pinned beforehand — drone_vec_host_pin — are read / written by the kernel in place, others are
 * copied through device staging; returns when the outputs are in the caller's memory). The handle's bound per-step buffers are
 * neither read nor written. State stays in registers between the K steps: the dependent-launch boundary and the state
 * planes' HBM traffic are paid once per K steps (hover: 102 + 176 / K bytes per env-step instead of 278), which is what
 * small shards (one wave per SIMD, launch-boundary bound) need. Consumers: open-loop action segments, action repeat /
 * frame skip, a device-side policy. With compact_done=1 every step's done-id list is kept: drone_vec_done_list_at.
 * Capturable into a hipGraph after drone_vec_enable_graph_capture (device buffers): make one call outside the capture
 * first — it sizes the K-dependent storage, which needs a stream sync. */
void drone_vec_step_many(DroneVec* v, int k_steps, const float* actions, float* observations, float* rewards,
                         unsigned char* terminals, unsigned char* truncations);

/* Action repeat (frame skip): the same with ONE actions block [N][4] applied to all k_steps steps — what k_steps calls
 * of drone_vec_step with an unchanged action buffer would do. Outputs as for drone_vec_step_many: every step's, K-major. */
void drone_vec_step_repeat(DroneVec* v, int k_steps, const float* actions, float* observations, float* rewards,
                           unsigned char* terminals, unsigned char* truncations);

/* Host handles: pin (and map) a host block the caller owns, on the handle's device, so that the kernel can access it in
 * place — K-major blocks of drone_vec_step_many / drone_vec_step_repeat that are pinned (this call, hipHostMalloc,
 * hipHostRegister) are read / written over PCIe directly, without device staging and copy commands (1 024 envs, K = 32:
 * 5.4 -> ~2 us per env step). Same rule as DroneConfig.host_pages_exclusive: the block must start on a 4 KiB boundary AND
 * be vouched for (pages_exclusive = 1: a mapping of its own — mmap, shm — padded to whole pages; not a block of the malloc
 * heap, however aligned: round 5). Unpin before unmapping the block. 0 / -1 (drone_last_error). */
int drone_vec_host_pin(DroneVec* v, void* block, size_t bytes, int pages_exclusive);
/* Drops a registration drone_vec_host_pin made on this handle. A block host_pin found already pinned by its owner
 * (hipHostMalloc / the caller's own hipHostRegister) was never registered here and is left alone: returns 0. */
int drone_vec_host_unpin(DroneVec* v, void* block);

void drone_vec_log(DroneVec* v, DroneLog* out);
void drone_vec_close(DroneVec* v);

/* ---- plumbing around the path ---- */

/* Launch on this hipStream_t (passed as void*) from now on. Default: a
 * stream the handle creates. Device-buffer mode is asynchronous on it. */
int drone_vec_set_stream(DroneVec* v, void* hip_stream);
int drone_vec_sync(DroneVec* v);

/* Rebind the action buffer (same kind as at init) — lets a caller rotate
 * through pre-filled action buffers without copies. */
int drone_vec_bind_actions(DroneVec* v, float* actions);

/* Rebind the four output buffers (same kind as at init). A consumer that
 * overlaps a collective or a copy of step k's outputs with step k+1 alternates
 * between two sets (drone_amd/dist.py PipelinedGather). */
int drone_vec_bind_outputs(DroneVec* v, float* observations, float* rewards,
                           unsigned char* terminals, unsigned char* truncations);

/* Write the SPEC.md §2 random-policy actions for step `gstep` into `actions`
 * (same kind as the handle's buffers). */
int drone_vec_fill_random_actions(DroneVec* v, float* actions, uint32_t gstep);

uint32_t drone_vec_gstep(const DroneVec* v);
int drone_vec_num_envs(const DroneVec* v);

/* The five buffers the handle is bound to now (the caller's, the last rebind's, or the library-owned ones), any of
 * the out-pointers may be NULL; and the HIP device ordinal the handle lives on (-1 for a NULL handle). */
int drone_vec_buffers(const DroneVec* v, float** observations, float** actions, float** rewards,
                      unsigned char** terminals, unsigned char** truncations);
int drone_vec_device(const DroneVec* v);

/* Restore the vec-level step counter (checkpoints): wind gusts and the random
 * policy are keyed on it (SPEC.md §2), so a run restored with set_state +
 * set_gstep continues bit for bit like the uninterrupted one. */
int drone_vec_set_gstep(DroneVec* v, uint32_t gstep);

/* Graph-safe stepping. By default the step counter travels in the launch arguments, so a drone_vec_step captured into
 * a hipGraph (hipStreamBeginCapture on the handle's stream — e.g. torch.cuda.graph around "policy forward + env step")
 * would replay with a frozen counter. After drone_vec_enable_graph_capture(v, 1) the counters live in HBM and the
 * kernels advance them themselves, so a captured step / rollout replays correctly (device buffers only; the bound
 * action / output pointers are baked into the capture like any graph argument). drone_vec_gstep and drone_vec_done_list
 * then read the device counters (a stream sync). Costs one more memory round trip per wave: off by default. */
int drone_vec_enable_graph_capture(DroneVec* v, int on);

/* Sticky status of the handle. reset / step / rollout / log return void (the
 * PufferLib convention), so a failed launch or copy would otherwise go
 * unnoticed: the FIRST failure of any call on the handle is kept here until
 * cleared. 0 = every call so far succeeded. */
int drone_vec_status(const DroneVec* v);
const char* drone_vec_status_message(const DroneVec* v);
void drone_vec_clear_status(DroneVec* v);

/* Copy envs [first, first+count) to / from AoS rows (host memory). */
int drone_vec_get_state(DroneVec* v, DroneStateRow* rows, int first, int count);
int drone_vec_set_state(DroneVec* v, const DroneStateRow* rows, int first, int count);

/* compact_done=1 only: ids (local) of the envs that finished in the last
 * drone_vec_step, unordered; returns their count (or -1). Copies at most `cap`
 * ids. Returns 0 when the last path call was drone_vec_reset or
 * drone_vec_rollout (the fused rollout builds no list). */
int drone_vec_done_list(DroneVec* v, uint32_t* ids, int cap);

/* The same for step `k` (0 <= k < k_steps) of the last drone_vec_step_many; -1 if the last path call was not a
 * step_many or k is out of range. */
int drone_vec_done_list_at(DroneVec* v, int k, uint32_t* ids, int cap);

/* ---- multi-GPU: the host-boundary exchange ----
 * Envs shard over GPUs with no collective on the env path (one process and one
 * handle per GPU, cfg.env_offset = first global id). A consumer that wants the
 * whole batch in one place calls drone_vec_gather after a step / rollout: an
 * RCCL all-gather (xGMI) of observations, rewards, terminals, truncations of
 * every rank into global buffers in global-env order, enqueued on the handle's
 * stream. librccl is dlopen'ed on first use; nothing here needs it otherwise.
 *
 * Bootstrap, plain C: ONE rank calls drone_gather_unique_id and ships the
 * DRONE_GATHER_ID_BYTES bytes to the others by any means (pipe, shared
 * memory, MPI, a file); every rank then calls drone_vec_gather_init (collective).
 *   counts: envs per rank [world], or NULL when every rank has num_envs envs.
 *   all_*:  global buffers [sum(counts)] rows, same kind as the handle's
 *           buffers. Device kind: written asynchronously on the stream; a local
 *           buffer that IS this rank's slice of the global one makes the
 *           collective in-place. Host kind: the collective runs on device
 *           staging, the batch is then copied to these host buffers and the
 *           call returns after the copy (steps use the mirror transport). */
#define DRONE_GATHER_ID_BYTES 128
int drone_gather_unique_id(unsigned char* id);
int drone_vec_gather_init(DroneVec* v, const unsigned char* id, int rank, int world, const int* counts,
                          float* all_observations, float* all_rewards,
                          unsigned char* all_terminals, unsigned char* all_truncations);
/* The same bootstrap with a choice of exchange (round 3). root = -1: the all-gather above. root in [0, world): a
 * gather TO THAT RANK only — every other rank ncclSends its rows once, the root ncclRecvs each rank's rows into their
 * place in its global buffers (one grouped launch); non-root ranks receive nothing and may pass NULL for the four
 * all_* pointers. For one consumer process (a trainer on rank 0) this is the north-star's "RCCL gather": against the
 * all-gather, the other 7 GPUs of a node stop receiving and writing 7/8 of the batch each. */
int drone_vec_gather_init_root(DroneVec* v, const unsigned char* id, int rank, int world, const int* counts, int root,
                               float* all_observations, float* all_rewards,
                               unsigned char* all_terminals, unsigned char* all_truncations);
int drone_vec_gather(DroneVec* v);
void drone_vec_gather_close(DroneVec* v);

/* The same exchange WITHOUT a collective: peer stores (round 4). For one consumer process on rank `root` whose batch
 * lives in HBM. The root exports its four global device buffers as IPC handles (drone_vec_gather_peer_export fills
 * DRONE_PEER_TOKEN_BYTES bytes; ship them to the other ranks like the RCCL id); every rank then calls
 * drone_vec_gather_init_peer. From then on each rank's kernels write observations / rewards / flags straight into that
 * rank's rows of the ROOT's buffers — local HBM on the root, stores over xGMI everywhere else: no collective launch, no
 * second pass over the outputs, nothing received or written by the other GPUs. drone_vec_gather is then only a
 * handshake, one call per launch on every rank as with RCCL, through `shared_flags`: ONE page-aligned 4 KiB page of host
 * memory shared by all ranks (POSIX shm, or a MAP_SHARED mapping made before fork), zeroed by whoever creates it.
 * (hipStreamWaitValue32 cannot do this: it accepts only the calling process's signal memory.) Round 5: the two
 * publications ride on the launches that write the outputs — a non-root rank's reset / step / rollout kernel says "my
 * launch has landed" itself (every workgroup releases its stores system-wide, the last one to finish stores the flag) and
 * the root's next launch acknowledges the previous round from its first workgroup — so what is left as launches of their
 * own are the two WAITS, one bounded one-wave kernel each: the root's stream waits for all ranks' flags, a non-root
 * rank's next launch waits until the root has begun ITS next launch (back-pressure: the root enqueues that launch behind
 * whatever consumed the batch on its stream). DRONE_PEER_INKERNEL=0 keeps the publications as one-wave launches (round 4).
 * Device buffers only. While the exchange is active these fail with a message: drone_vec_bind_outputs (the exchange owns
 * the output bindings), drone_vec_step_many / step_repeat (they write the caller's blocks, outside the handshake) and
 * drone_vec_enable_graph_capture — and drone_vec_gather_init_peer itself on a handle already in graph-safe mode: the
 * handshake's round numbers are host state baked into each launch, so a replayed capture would wait for and publish a
 * stale round. drone_vec_gather_close gives the handle its own output buffers back and forgets the export (export again
 * before another drone_vec_gather_init_peer). A dead peer: every wait gives up after DRONE_PEER_TIMEOUT_MS (default
 * 10 000; clamped to 1 ... 600 000, anything else is the default) — a stream-side wait raises a flag that fails the next
 * call on the handle, makes every wait already queued behind it return at once, and (round 6) raises a stop word in HBM:
 * the reset / step / rollout launches already queued behind it store nothing and publish nothing, so the root's batch stays
 * what it was while they drain (handles in an exchange launch instantiations of their own for this; every other handle's
 * kernels do not know the word); DRONE_PEER_HOST_WAIT=1 moves the whole handshake to the host (the stream is drained, the
 * host polls / stores), where the timeout is an immediate error.
 * ONE drone_vec_gather per output-writing launch (round 6): a second reset / step / rollout before the gather fails with a
 * message — its wait would already be satisfied and it would overwrite rows the root may be consuming.
 * The exported buffers must be plain device allocations (drone_device_malloc, hipMalloc, torch's default allocator): a
 * virtual-memory mapping (hipMemCreate / hipMemMap, torch's expandable_segments) has no IPC handle, and
 * drone_vec_gather_peer_export says so. */
#define DRONE_PEER_TOKEN_BYTES 288
int drone_vec_gather_peer_export(DroneVec* v, float* all_observations, float* all_rewards,
                                 unsigned char* all_terminals, unsigned char* all_truncations, unsigned char* token);
int drone_vec_gather_init_peer(DroneVec* v, const unsigned char* token, void* shared_flags, int rank, int world,
                               const int* counts, int root);

/* For hosts that stay plain C (no HIP headers): raw, zeroed device memory on a HIP device — e.g. the root's global buffers
 * of the peer-store exchange — and a copy from device memory to the host, enqueued on the handle's stream behind its
 * work (so behind a drone_vec_gather's waits) and complete on return. */
void* drone_device_malloc(int device, size_t bytes);
void drone_device_free(int device, void* p);
int drone_vec_copy_to_host(DroneVec* v, void* host_dst, const void* device_src, size_t bytes);

/* HIP-event timer on the handle's stream: start, ..launches.., stop → ms. */
int drone_vec_timer_start(DroneVec* v);
int drone_vec_timer_stop(DroneVec* v, float* elapsed_ms);

/* Last error message of the calling thread ("" if none). */
const char* drone_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* DRONE_VEC_H */

/*
 * drone_host_mp.c — plain-C multi-GPU host: one PROCESS per GPU, forked before
 * anything touches HIP, each driving its contiguous shard of the envs through
 * include/drone_vec.h, with the north-star's one exchange step — the RCCL
 * all-gather of observations / rewards / flags at the host boundary — reached
 * from C through drone_vec_gather (BASELINE.json north_star:
 * "host side stays C calling HIP through a thin C-ABI"). No HIP or RCCL headers
 * here: both stay behind the C-ABI.
 *
 *   drone_host_mp [--gpus G] [--envs TOTAL] [--steps K] [--task 0..3] [--seed S]
 *                 [--gather 0|1] [--root R] [--exchange rccl|peer] [--rollout T] [--crc 1] [--share-devices 1] [--timeout SECONDS]
 *
 * Rank r takes envs [offset_r, offset_r + count_r) (the first TOTAL % G ranks get
 * one more) on device r. The RCCL unique id is made by rank 0 AFTER the fork and
 * handed to the other ranks through an anonymous shared mapping created before
 * the fork — the bootstrap needs nothing but plain C. Every rank keeps the whole
 * batch in host memory (its local buffers are its slice of the global ones).
 * Rank 0 prints one JSON line; with --crc 1 the CRC-32 chained over every step's
 * gathered batch, which tests/test_c_host.py compares with the CPU oracle's.
 * --root R (default -1): -1 = all-gather, every rank ends up with the whole batch; R >= 0 = gather to rank R only
 *   (ncclSend / ncclRecv): the batch lands in rank R's host buffers, the printed CRC is rank R's.
 * --exchange peer (round 4): the exchange WITHOUT a collective. Device buffers; rank R (--root, default 0) owns the batch
 *   in its HBM and exports it (drone_vec_gather_peer_export: IPC handles through the shared page); every other rank's step
 *   kernel stores its rows straight into it (xGMI on a multi-GPU node), drone_vec_gather is a flag handshake on a second
 *   shared page. The root copies each batch to its host buffers for the CRC (its stand-in consumer). Works with several
 *   ranks on ONE GPU too (--share-devices 1): the IPC mapping and the handshake are the same, only the stores stay local.
 * --rollout T: fused T-step rollouts with the gather once per horizon (configs[4]).
 * --share-devices 1: rank r uses device r %% (visible devices) — lets the fork / shard / barrier logic run with several
 *   ranks on a one-GPU box (without --gather: RCCL refuses two ranks on one device); every rank's CRC over ITS OWN
 *   slice is printed so a test can check each shard against the oracle.
 * --timeout S (default 300): wall-clock limit of the whole run. The parent reaps its children in ANY order; the first
 *   abnormal exit (non-zero status, a signal), or the limit, marks the run failed, SIGKILLs the remaining ranks and
 *   exits non-zero — a rank blocked in a collective whose peer died can never hang the host. The ranks' own spins
 *   (barrier, unique-id hand-over) watch the same flag and the same deadline.
 */
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

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

typedef struct Shared {
    volatile int id_ready;
    volatile int token_ready;
    unsigned char token[DRONE_PEER_TOKEN_BYTES]; /* --exchange peer: the root's exported batch */
    volatile int failed;
    unsigned char id[DRONE_GATHER_ID_BYTES];
    volatile int arrived[2]; /* sense-reversing barrier over the ranks */
    volatile int sense;
    double deadline; /* CLOCK_MONOTONIC seconds after which every spin gives up (set by the parent before the fork) */
    double rank_seconds[64];
    uint32_t rank_crc[64]; /* CRC-32 of each rank's own slice of the outputs, chained over the launches */
    uint32_t batch_crc[64]; /* CRC-32 of the whole gathered batch as each rank sees it (meaningful on receiving ranks) */
} Shared;

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

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* 0 = all ranks arrived; 1 = another rank failed or the deadline passed (the caller returns non-zero) */
static int barrier(Shared* sh, int world, int* local_sense) {
    *local_sense = !*local_sense;
    const int slot = *local_sense;
    if (__sync_add_and_fetch(&sh->arrived[slot], 1) == world) {
        sh->arrived[slot] = 0;
        __sync_synchronize();
        sh->sense = *local_sense;
    } else {
        while (sh->sense != *local_sense) {
            if (sh->failed || now_s() > sh->deadline) return 1;
            usleep(50);
        }
    }
    return 0;
}

typedef struct Opts {
    int gpus, total, steps, task, gather, rollout, crc, share, timeout, die_rank, root, peer;
    unsigned long long seed;
} Opts;

/* --exchange peer, the root's stand-in consumer: copy the batch from its HBM to the host — stream-ordered behind the
 * handshake's waits (drone_vec_copy_to_host) — and chain the CRC over it. After a reset only the observations are defined. */
typedef struct PeerBatch {
    float *g_obs, *g_rew, *h_obs, *h_rew;
    unsigned char *g_term, *g_trunc, *h_term, *h_trunc;
    size_t total, od;
} PeerBatch;

static int peer_consume(DroneVec* v, const PeerBatch* b, int after_reset, uint32_t* crc) {
    if (drone_vec_copy_to_host(v, b->h_obs, b->g_obs, sizeof(float) * b->total * b->od) != 0) return -1;
    *crc = crc32_update(*crc, b->h_obs, sizeof(float) * b->total * b->od);
    if (after_reset) return 0;
    if (drone_vec_copy_to_host(v, b->h_rew, b->g_rew, sizeof(float) * b->total) != 0 || drone_vec_copy_to_host(v, b->h_term, b->g_term, b->total) != 0 ||
        drone_vec_copy_to_host(v, b->h_trunc, b->g_trunc, b->total) != 0)
        return -1;
    *crc = crc32_update(*crc, b->h_rew, sizeof(float) * b->total);
    *crc = crc32_update(*crc, b->h_term, b->total);
    *crc = crc32_update(*crc, b->h_trunc, b->total);
    return 0;
}

/* --exchange peer: one rank's whole run. Device-buffer handle with library-owned local buffers; the root's global batch
 * lives in its HBM (drone_device_malloc) and is copied to the host after every gather for the CRC. */
static int run_rank_peer(const Opts* o, int rank, Shared* sh, void* flag_page) {
    const int world = o->gpus, root = o->root;
    int counts[64], offsets[64];
    for (int r = 0, off = 0; r < world; r++) {
        counts[r] = o->total / world + (r < o->total % world ? 1 : 0);
        offsets[r] = off;
        off += counts[r];
    }
    const int n = counts[rank];
    const size_t od = (size_t)drone_obs_dim(o->task), total = (size_t)o->total;
    DroneConfig cfg;
    drone_config_default(&cfg, o->task);
    cfg.buffer_kind = DRONE_BUFFERS_DEVICE;
    cfg.device = rank;
    if (o->share) {
        const int ndev = drone_device_count();
        if (ndev < 1) { fprintf(stderr, "rank %d: no HIP device\n", rank); return 1; }
        cfg.device = rank % ndev;
    }
    cfg.env_offset = (uint32_t)offsets[rank];
    DroneVec* v = drone_vec_init(NULL, NULL, NULL, NULL, NULL, n, o->seed, &cfg); /* library-owned HBM buffers */
    if (!v) { fprintf(stderr, "rank %d: drone_vec_init failed: %s\n", rank, drone_last_error()); return 1; }
    float* act = NULL;
    drone_vec_buffers(v, NULL, &act, NULL, NULL, NULL);
    float *g_obs = NULL, *g_rew = NULL, *h_obs = NULL, *h_rew = NULL;
    unsigned char *g_term = NULL, *g_trunc = NULL, *h_term = NULL, *h_trunc = NULL;
    if (rank == root) {
        g_obs = (float*)drone_device_malloc(cfg.device, sizeof(float) * total * od);
        g_rew = (float*)drone_device_malloc(cfg.device, sizeof(float) * total);
        g_term = (unsigned char*)drone_device_malloc(cfg.device, total);
        g_trunc = (unsigned char*)drone_device_malloc(cfg.device, total);
        h_obs = (float*)page_alloc(sizeof(float) * total * od);
        h_rew = (float*)page_alloc(sizeof(float) * total);
        h_term = (unsigned char*)page_alloc(total);
        h_trunc = (unsigned char*)page_alloc(total);
        if (!g_obs || !g_rew || !g_term || !g_trunc || !h_obs || !h_rew || !h_term || !h_trunc) { fprintf(stderr, "rank %d: out of memory (%s)\n", rank, drone_last_error()); return 1; }
        unsigned char token[DRONE_PEER_TOKEN_BYTES];
        if (drone_vec_gather_peer_export(v, g_obs, g_rew, g_term, g_trunc, token) != 0) { fprintf(stderr, "rank %d: peer export failed: %s\n", rank, drone_last_error()); return 1; }
        memcpy((void*)sh->token, token, sizeof(token));
        __sync_synchronize();
        sh->token_ready = 1;
    } else {
        while (!sh->token_ready) {
            if (sh->failed || now_s() > sh->deadline) { fprintf(stderr, "rank %d: gave up waiting for the root's export\n", rank); return 1; }
            usleep(100);
        }
    }
    unsigned char token[DRONE_PEER_TOKEN_BYTES];
    memcpy(token, (const void*)sh->token, sizeof(token));
    if (drone_vec_gather_init_peer(v, token, flag_page, rank, world, counts, root) != 0) {
        fprintf(stderr, "rank %d: drone_vec_gather_init_peer failed: %s\n", rank, drone_last_error());
        return 1;
    }
    int sense = 0;
    if (rank == o->die_rank) raise(SIGKILL);
    uint32_t crc = 0;
    const PeerBatch pb = {g_obs, g_rew, h_obs, h_rew, g_term, g_trunc, h_term, h_trunc, total, od};
    drone_vec_reset(v, o->seed);
    if (drone_vec_gather(v) != 0) { fprintf(stderr, "rank %d: gather failed: %s\n", rank, drone_last_error()); return 1; }
    if (rank == root && o->crc && peer_consume(v, &pb, 1, &crc) != 0) { fprintf(stderr, "rank %d: %s\n", rank, drone_last_error()); return 1; }
    const int launches = o->rollout > 0 ? (o->steps + o->rollout - 1) / o->rollout : o->steps;
    if (barrier(sh, world, &sense)) { fprintf(stderr, "rank %d: start barrier abandoned\n", rank); return 1; }
    const double t0 = now_s();
    for (int t = 0; t < launches; t++) {
        if (o->rollout > 0) {
            drone_vec_rollout(v, o->rollout);
        } else {
            drone_vec_fill_random_actions(v, act, drone_vec_gstep(v));
            drone_vec_step(v);
        }
        if (drone_vec_gather(v) != 0) { fprintf(stderr, "rank %d: gather failed: %s\n", rank, drone_last_error()); return 1; }
        if (rank == root && o->crc && peer_consume(v, &pb, 0, &crc) != 0) { fprintf(stderr, "rank %d: %s\n", rank, drone_last_error()); return 1; }
    }
    if (drone_vec_sync(v) != 0) { fprintf(stderr, "rank %d: %s\n", rank, drone_last_error()); return 1; }
    sh->batch_crc[rank] = crc;
    sh->rank_crc[rank] = 0;
    sh->rank_seconds[rank] = now_s() - t0;
    if (drone_vec_status(v)) { fprintf(stderr, "rank %d: %s\n", rank, drone_vec_status_message(v)); return 1; }
    if (barrier(sh, world, &sense)) { fprintf(stderr, "rank %d: end barrier abandoned\n", rank); return 1; }
    if (rank == 0) {
        double el = 0;
        for (int r = 0; r < world; r++) el = sh->rank_seconds[r] > el ? sh->rank_seconds[r] : el;
        const double env_steps = (double)o->total * (o->rollout > 0 ? (double)o->rollout : 1.0) * launches;
        printf("{\"mode\": \"%s + peer-store gather into rank %d's HBM batch (no collective)%s\", \"gpus\": %d, \"root\": %d, \"task\": %d, \"envs\": %d, \"launches\": %d, "
               "\"horizon\": %d, \"env_steps_per_s\": %.4g, \"ms_per_launch\": %.4f, \"crc32\": %u}\n",
               o->rollout > 0 ? "fused rollout" : "per-step", root, o->crc ? ", batch copied to the host and CRC'd every launch" : "", world, root, o->task, o->total, launches,
               o->rollout, env_steps / el, el * 1e3 / launches, sh->batch_crc[root]);
        fflush(stdout);
    }
    drone_vec_gather_close(v);
    drone_vec_close(v);
    if (rank == root) {
        drone_device_free(cfg.device, g_obs); drone_device_free(cfg.device, g_rew); drone_device_free(cfg.device, g_term); drone_device_free(cfg.device, g_trunc);
        /* (h_*: mappings of their own, released with the process) */
    }
    return 0;
}

static int run_rank(const Opts* o, int rank, Shared* sh) {
    const int world = o->gpus;
    int counts[64], offsets[64];
    for (int r = 0, off = 0; r < world; r++) {
        counts[r] = o->total / world + (r < o->total % world ? 1 : 0);
        offsets[r] = off;
        off += counts[r];
    }
    const int n = counts[rank];
    const size_t od = (size_t)drone_obs_dim(o->task), total = (size_t)o->total;
    /* the whole batch in host memory on every rank; the local buffers are this rank's slice of it */
    float* all_obs = (float*)page_alloc(sizeof(float) * total * od);
    float* all_rew = (float*)page_alloc(sizeof(float) * total);
    unsigned char* all_term = (unsigned char*)page_alloc(total);
    unsigned char* all_trunc = (unsigned char*)page_alloc(total);
    float* act = (float*)page_alloc(sizeof(float) * (size_t)n * DRONE_ACT_DIM);
    if (!all_obs || !all_rew || !all_term || !all_trunc || !act) { fprintf(stderr, "rank %d: out of memory\n", rank); return 1; }
    memset(all_obs, 0, sizeof(float) * total * od);

    DroneConfig cfg;
    drone_config_default(&cfg, o->task);
    cfg.buffer_kind = DRONE_BUFFERS_HOST;
    /* Every buffer here comes from page_alloc — a mapping of its own, never the malloc heap — which is what the flag vouches
     * for (round 5: the library no longer registers on alignment alone). The gather pins the four global buffers whole; this
     * rank's local output buffers are SLICES of them: one that does not start on a page boundary is never registered (the
     * library copies), one that does lies inside our own mapping, and the gather drops the local pins anyway. */
    cfg.host_pages_exclusive = 1;
    cfg.device = rank; /* one process per GPU */
    if (o->share) {
        const int ndev = drone_device_count();
        if (ndev < 1) { fprintf(stderr, "rank %d: no HIP device\n", rank); return 1; }
        cfg.device = rank % ndev;
    }
    cfg.env_offset = (uint32_t)offsets[rank];
    DroneVec* v = drone_vec_init(all_obs + (size_t)offsets[rank] * od, act, all_rew + offsets[rank], all_term + offsets[rank],
                                 all_trunc + offsets[rank], n, o->seed, &cfg);
    if (!v) { fprintf(stderr, "rank %d: drone_vec_init failed: %s\n", rank, drone_last_error()); return 1; }

    if (o->gather) {
        if (rank == 0) {
            if (drone_gather_unique_id(sh->id) != 0) { fprintf(stderr, "rank 0: %s\n", drone_last_error()); return 1; }
            __sync_synchronize();
            sh->id_ready = 1;
        } else {
            while (!sh->id_ready) {
                if (sh->failed || now_s() > sh->deadline) { fprintf(stderr, "rank %d: gave up waiting for the RCCL unique id\n", rank); return 1; }
                usleep(100);
            }
        }
        unsigned char id[DRONE_GATHER_ID_BYTES];
        memcpy(id, (const void*)sh->id, sizeof(id));
        if (drone_vec_gather_init_root(v, id, rank, world, counts, o->root, all_obs, all_rew, all_term, all_trunc) != 0) {
            fprintf(stderr, "rank %d: drone_vec_gather_init failed: %s\n", rank, drone_last_error());
            return 1;
        }
    }

    int sense = 0;
    if (rank == o->die_rank) raise(SIGKILL); /* tests only: a rank that dies without saying so (OOM kill, segfault) */
    drone_vec_reset(v, o->seed);
    if (o->gather && drone_vec_gather(v) != 0) { fprintf(stderr, "rank %d: gather failed: %s\n", rank, drone_last_error()); return 1; }
    uint32_t crc = 0, own = 0;
    const size_t off = (size_t)offsets[rank];
    if (o->crc) {
        crc = crc32_update(crc, all_obs, sizeof(float) * total * od);
        own = crc32_update(own, all_obs + off * od, sizeof(float) * (size_t)n * od);
    }
    const int launches = o->rollout > 0 ? (o->steps + o->rollout - 1) / o->rollout : o->steps;
    if (barrier(sh, world, &sense)) { fprintf(stderr, "rank %d: start barrier abandoned (a rank failed or the time limit passed)\n", rank); return 1; }
    const double t0 = now_s();
    for (int t = 0; t < launches; t++) {
        if (o->rollout > 0) {
            drone_vec_rollout(v, o->rollout);
        } else {
            drone_vec_fill_random_actions(v, act, drone_vec_gstep(v));
            drone_vec_step(v);
        }
        if (o->gather && drone_vec_gather(v) != 0) { fprintf(stderr, "rank %d: gather failed: %s\n", rank, drone_last_error()); return 1; }
        if (o->crc) {
            crc = crc32_update(crc, all_obs, sizeof(float) * total * od);
            crc = crc32_update(crc, all_rew, sizeof(float) * total);
            crc = crc32_update(crc, all_term, total);
            crc = crc32_update(crc, all_trunc, total);
            own = crc32_update(own, all_obs + off * od, sizeof(float) * (size_t)n * od);
            own = crc32_update(own, all_rew + off, sizeof(float) * (size_t)n);
            own = crc32_update(own, all_term + off, (size_t)n);
            own = crc32_update(own, all_trunc + off, (size_t)n);
        }
    }
    sh->rank_crc[rank] = own;
    sh->batch_crc[rank] = crc;
    sh->rank_seconds[rank] = now_s() - t0;
    if (drone_vec_status(v)) { fprintf(stderr, "rank %d: %s\n", rank, drone_vec_status_message(v)); return 1; }
    if (barrier(sh, world, &sense)) { fprintf(stderr, "rank %d: end barrier abandoned (a rank failed or the time limit passed)\n", rank); return 1; }
    if (rank == 0) {
        double el = 0;
        for (int r = 0; r < world; r++) el = sh->rank_seconds[r] > el ? sh->rank_seconds[r] : el;
        const double env_steps = (double)o->total * (o->rollout > 0 ? (double)o->rollout : 1.0) * launches;
        printf("{\"mode\": \"%s%s\", \"gpus\": %d, \"root\": %d, \"task\": %d, \"envs\": %d, \"launches\": %d, \"horizon\": %d, \"env_steps_per_s\": %.4g, "
               "\"ms_per_launch\": %.4f, \"crc32\": %u, \"rank_crc32\": [",
               o->rollout > 0 ? "fused rollout" : "per-step", !o->gather ? " (no gather)" : o->root >= 0 ? " + RCCL gather to one rank's host batch" : " + RCCL all-gather to every rank's host batch", world, o->gather ? o->root : -1, o->task,
               o->total, launches, o->rollout, env_steps / el, el * 1e3 / launches, sh->batch_crc[o->root >= 0 ? o->root : 0]);
        for (int r = 0; r < world; r++) printf("%s%u", r ? ", " : "", sh->rank_crc[r]);
        printf("]}\n");
        fflush(stdout);
    }
    if (o->gather) drone_vec_gather_close(v);
    drone_vec_close(v);
    /* (the host buffers are mappings of their own, released with the process) */
    return 0;
}

int main(int argc, char** argv) {
    Opts o = {1, 65536, 100, DRONE_TASK_HOVER, 1, 0, 0, 0, 300, -1, -1, 0, 0ull};
    for (int i = 1; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "--gpus")) o.gpus = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--envs")) o.total = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--steps")) o.steps = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--task")) o.task = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--gather")) o.gather = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--rollout")) o.rollout = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--crc")) o.crc = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--share-devices")) o.share = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--root")) o.root = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--exchange")) {
            if (!strcmp(argv[i + 1], "peer")) o.peer = 1;
            else if (strcmp(argv[i + 1], "rccl")) { fprintf(stderr, "--exchange must be rccl or peer\n"); return 2; }
        }
        else if (!strcmp(argv[i], "--timeout")) o.timeout = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--die-rank")) o.die_rank = atoi(argv[i + 1]); /* tests: that rank kills itself before the start barrier */
        else if (!strcmp(argv[i], "--seed")) o.seed = strtoull(argv[i + 1], NULL, 10);
        else { fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
    }
    if (o.timeout < 1) o.timeout = 1;
    if (o.gpus < 1 || o.gpus > 64 || o.total < o.gpus || o.steps < 1 || o.rollout < 0) { fprintf(stderr, "bad --gpus / --envs / --steps / --rollout\n"); return 2; }
    if (o.root < -1 || o.root >= o.gpus) { fprintf(stderr, "--root must be -1 or a rank\n"); return 2; }
    if (o.task < 0 || o.task > 3) { fprintf(stderr, "unknown task %d\n", o.task); return 2; }
    if (o.peer && o.root < 0) o.root = 0; /* the peer-store exchange always has one owner of the batch */
    if (o.task == DRONE_TASK_SWARM && (o.total % (8 * o.gpus))) { fprintf(stderr, "swarm task: --envs must be a multiple of 8 x --gpus\n"); return 2; }

    /* One node by construction (one process per local GPU): let RCCL's bootstrap use the loopback interface unless the
     * user chose one — containers without a resolvable hostname or a routable interface otherwise stall it. */
    setenv("NCCL_SOCKET_IFNAME", "lo", 0);
    /* shared page + fork BEFORE any HIP call: a forked child of a process that initialised the GPU is not usable */
    Shared* sh = (Shared*)mmap(NULL, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    if (sh == MAP_FAILED) { perror("mmap"); return 1; }
    memset(sh, 0, sizeof(*sh));
    sh->deadline = now_s() + (double)o.timeout;
    /* --exchange peer: the flag page of the handshake, a page of its own shared by all ranks (same rule: before the fork) */
    void* flag_page = mmap(NULL, 4096, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    if (flag_page == MAP_FAILED) { perror("mmap"); return 1; }
    memset(flag_page, 0, 4096);
    pid_t pids[64];
    int started = 0;
    for (int r = 0; r < o.gpus; r++) {
        pids[r] = fork();
        if (pids[r] < 0) { perror("fork"); sh->failed = 1; break; }
        if (pids[r] == 0) {
            const int rc = o.peer ? run_rank_peer(&o, r, sh, flag_page) : run_rank(&o, r, sh);
            if (rc) sh->failed = 1;
            fflush(stdout);
            _exit(rc);
        }
        started++;
    }
    /* Reap in ANY order, without blocking on one particular rank: a rank that dies by a signal never sets `failed`
     * itself, and its peers may be blocked inside a collective (not in one of our spins) waiting for it. The first
     * abnormal exit — or the wall-clock limit — kills the rest. */
    int rc = started == o.gpus ? 0 : 1, alive = started;
    while (alive > 0) {
        int st = 0;
        const pid_t p = waitpid(-1, &st, WNOHANG);
        if (p > 0) {
            alive--;
            for (int r = 0; r < started; r++)
                if (pids[r] == p) pids[r] = 0;
            if (!WIFEXITED(st) || WEXITSTATUS(st)) {
                if (!rc) fprintf(stderr, "drone_host_mp: a rank ended abnormally (%s %d): stopping the others\n",
                                 WIFSIGNALED(st) ? "signal" : "status", WIFSIGNALED(st) ? WTERMSIG(st) : WEXITSTATUS(st));
                rc = 1;
            }
        } else if (p < 0) {
            break; /* no children left */
        } else {
            if (!rc && now_s() > sh->deadline + 2.0) { /* the ranks' own spins give up at the deadline; whoever is still here is stuck in a call */
                fprintf(stderr, "drone_host_mp: time limit of %d s passed: stopping all ranks\n", o.timeout);
                rc = 1;
            }
            usleep(2000);
        }
        if (rc && alive > 0) {
            sh->failed = 1;
            for (int r = 0; r < started; r++)
                if (pids[r] > 0) kill(pids[r], SIGKILL);
        }
    }
    return rc;
}

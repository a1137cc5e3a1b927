/*
 * rccl_stub.c — TEST DOUBLE for librccl (tests only; never shipped, never loaded unless DRONE_RCCL_LIB points at it).
 *
 * RCCL refuses two ranks on one GPU, and the pool has 1-GPU boxes only, so the rank > 0 side of drone_vec_gather
 * (slice offsets, ragged counts, the all-gather-v branch, host staging) could never run. This stub implements the
 * ten entry points the library dlsym()s with the semantics RCCL documents, over a POSIX shared-memory segment
 * between processes that may share a device: every op is stream-synchronous (hipStreamSynchronize, copy through the
 * segment, two barriers). It checks what RCCL would require: same count on every rank for all-gather, a root in
 * range, calls in the same order (op sequence number + kind + count compared across ranks). ncclSend / ncclRecv are
 * supported inside a group that EVERY rank opens and closes (what drone_vec_gather does): they are queued and matched
 * at ncclGroupEnd — k-th send of rank a to rank b with the k-th recv of b from a, sizes compared, unmatched ones an error.
 *
 *   gcc -shared -fPIC -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ rccl_stub.c -L/opt/rocm/lib -lamdhip64 -lrt -o librccl_stub.so
 */
#define _GNU_SOURCE
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#define SLOT_BYTES ((size_t)64 << 20)
#define MAX_RANKS 16
#define MAX_P2P 64      /* queued sends (or recvs) per rank and group */
#define HEADER_BYTES ((size_t)64 << 10)

typedef struct Header {
    volatile int arrived[2];
    volatile int sense;
    volatile int failed;
    volatile unsigned long long op_sig[MAX_RANKS]; /* what each rank thinks the current op is */
    /* point-to-point: what each rank has parked in its slot for the group being closed */
    volatile int n_sends[MAX_RANKS];
    struct { volatile int peer; volatile int consumed; volatile size_t bytes, offset; } sends[MAX_RANKS][MAX_P2P];
} Header;

typedef struct P2P { int is_send, peer; size_t bytes; const void* src; void* dst; hipStream_t stream; } P2P;

struct ncclComm {
    int rank, nranks;
    char name[NCCL_UNIQUE_ID_BYTES];
    Header* h;
    unsigned char* slots;
    size_t map_bytes;
    int local_sense;
    unsigned long long seq;
    P2P queue[2 * MAX_P2P];
    int n_queue;
};

static int g_group_depth = 0;
static struct ncclComm* g_group_comm = NULL; /* the communicator whose point-to-point ops are queued in the open group */

static size_t type_size(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
        default: return 0;
    }
}

static int barrier(struct ncclComm* c) {
    c->local_sense = !c->local_sense;
    const int slot = c->local_sense;
    if (__sync_add_and_fetch(&c->h->arrived[slot], 1) == c->nranks) {
        c->h->arrived[slot] = 0;
        __sync_synchronize();
        c->h->sense = c->local_sense;
    } else {
        const time_t t0 = time(NULL);
        while (c->h->sense != c->local_sense) {
            if (c->h->failed || time(NULL) - t0 > 60) { c->h->failed = 1; return -1; }
            usleep(20);
        }
    }
    return c->h->failed ? -1 : 0;
}

/* every rank must be issuing the same op: kind, element count, root */
static int agree(struct ncclComm* c, unsigned long long sig) {
    c->seq++;
    c->h->op_sig[c->rank] = (c->seq << 40) ^ sig;
    if (barrier(c)) return -1;
    for (int r = 0; r < c->nranks; r++)
        if (c->h->op_sig[r] != c->h->op_sig[c->rank]) {
            fprintf(stderr, "rccl_stub: rank %d and rank %d disagree on op %llu\n", c->rank, r, c->seq);
            c->h->failed = 1;
        }
    return barrier(c);
}

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    memset(id, 0, sizeof(*id));
    snprintf(id->internal, sizeof(id->internal), "/drone_rccl_stub_%d_%ld", (int)getpid(), (long)time(NULL));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    struct ncclComm* c = (struct ncclComm*)calloc(1, sizeof(*c));
    c->rank = rank;
    c->nranks = nranks;
    snprintf(c->name, sizeof(c->name), "%s", id.internal);
    c->map_bytes = HEADER_BYTES + SLOT_BYTES * (size_t)nranks;
    int fd = -1;
    if (rank == 0) {
        fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) return ncclSystemError;
    } else {
        for (int tries = 0; tries < 5000 && fd < 0; tries++) {
            fd = shm_open(c->name, O_RDWR, 0600);
            if (fd < 0) usleep(1000);
        }
        if (fd < 0) return ncclSystemError;
        for (int tries = 0; tries < 5000; tries++) {  /* wait until rank 0 has sized it */
            off_t sz = lseek(fd, 0, SEEK_END);
            if (sz >= (off_t)c->map_bytes) break;
            usleep(1000);
        }
    }
    void* p = mmap(NULL, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return ncclSystemError;
    c->h = (Header*)p;
    c->slots = (unsigned char*)p + HEADER_BYTES;
    *comm = (ncclComm_t)c;
    if (barrier(c)) return ncclSystemError;  /* like the real one: returns once every rank has joined */
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    struct ncclComm* c = (struct ncclComm*)comm;
    if (!c) return ncclSuccess;
    (void)barrier(c);
    if (c->rank == 0) shm_unlink(c->name);
    munmap((void*)c->h, c->map_bytes);
    free(c);
    return ncclSuccess;
}

ncclResult_t ncclGroupStart(void) {
    g_group_depth++;
    return ncclSuccess;
}

/* run the queued sends / recvs of the group: park the sends in this rank's slot, meet, fetch what is addressed to us, meet */
static ncclResult_t flush_p2p(struct ncclComm* c) {
    Header* h = c->h;
    size_t off = 0;
    int ns = 0;
    for (int i = 0; i < c->n_queue; i++) {
        P2P* q = &c->queue[i];
        if (!q->is_send) continue;
        if (off + q->bytes > SLOT_BYTES || ns >= MAX_P2P) { h->failed = 1; break; }
        if (hipStreamSynchronize(q->stream) != hipSuccess || hipMemcpy(c->slots + SLOT_BYTES * (size_t)c->rank + off, q->src, q->bytes, hipMemcpyDeviceToHost) != hipSuccess) { h->failed = 1; break; }
        h->sends[c->rank][ns].peer = q->peer;
        h->sends[c->rank][ns].bytes = q->bytes;
        h->sends[c->rank][ns].offset = off;
        h->sends[c->rank][ns].consumed = 0;
        off += (q->bytes + 255) & ~(size_t)255;
        ns++;
    }
    h->n_sends[c->rank] = ns;
    if (barrier(c)) return ncclInternalError;
    for (int i = 0; i < c->n_queue; i++) {
        P2P* q = &c->queue[i];
        if (q->is_send) continue;
        int found = -1;
        for (int k = 0; k < h->n_sends[q->peer]; k++)
            if (h->sends[q->peer][k].peer == c->rank && !h->sends[q->peer][k].consumed) { found = k; break; }
        if (found < 0 || h->sends[q->peer][found].bytes != q->bytes) {
            fprintf(stderr, "rccl_stub: rank %d: recv of %zu bytes from rank %d has no matching send\n", c->rank, q->bytes, q->peer);
            h->failed = 1;
            break;
        }
        if (hipStreamSynchronize(q->stream) != hipSuccess ||
            hipMemcpy(q->dst, c->slots + SLOT_BYTES * (size_t)q->peer + h->sends[q->peer][found].offset, q->bytes, hipMemcpyHostToDevice) != hipSuccess) { h->failed = 1; break; }
        h->sends[q->peer][found].consumed = 1;
    }
    c->n_queue = 0;
    if (barrier(c)) return ncclInternalError;
    for (int k = 0; k < ns; k++)
        if (!h->sends[c->rank][k].consumed) {
            fprintf(stderr, "rccl_stub: rank %d: send %d to rank %d was never received\n", c->rank, k, h->sends[c->rank][k].peer);
            h->failed = 1;
        }
    return barrier(c) ? ncclInternalError : ncclSuccess;
}

ncclResult_t ncclGroupEnd(void) {
    if (g_group_depth <= 0) return ncclInvalidUsage;
    if (--g_group_depth > 0) return ncclSuccess;
    struct ncclComm* c = g_group_comm;
    g_group_comm = NULL;
    if (!c) return ncclSuccess; /* a group of collectives only: they ran as they were issued */
    /* every rank of the communicator must be closing a group with point-to-point ops (senders and receivers alike) */
    if (agree(c, 0xC000000000ull)) return ncclInternalError;
    return flush_p2p(c);
}

static ncclResult_t queue_p2p(struct ncclComm* c, int is_send, const void* src, void* dst, size_t count, ncclDataType_t datatype, int peer, hipStream_t stream) {
    const size_t bytes = count * type_size(datatype);
    if (!bytes || bytes > SLOT_BYTES || peer < 0 || peer >= c->nranks || peer == c->rank) return ncclInvalidArgument;
    if (g_group_depth <= 0) { fprintf(stderr, "rccl_stub: ncclSend / ncclRecv outside a group is not supported by this test double\n"); return ncclInvalidUsage; }
    if (g_group_comm && g_group_comm != c) return ncclInvalidUsage;
    if (c->n_queue >= 2 * MAX_P2P) return ncclInvalidUsage;
    g_group_comm = c;
    P2P* q = &c->queue[c->n_queue++];
    q->is_send = is_send; q->peer = peer; q->bytes = bytes; q->src = src; q->dst = dst; q->stream = stream;
    return ncclSuccess;
}

ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    return queue_p2p((struct ncclComm*)comm, 1, sendbuff, NULL, count, datatype, peer, stream);
}

ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    return queue_p2p((struct ncclComm*)comm, 0, NULL, recvbuff, count, datatype, peer, stream);
}
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "rccl_stub error"; }

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream) {
    struct ncclComm* c = (struct ncclComm*)comm;
    const size_t bytes = sendcount * type_size(datatype);
    if (!bytes || bytes > SLOT_BYTES) return ncclInvalidArgument;
    if (agree(c, 0xA000000000ull ^ (unsigned long long)sendcount ^ ((unsigned long long)datatype << 32))) return ncclInternalError;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpy(c->slots + SLOT_BYTES * (size_t)c->rank, sendbuff, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    if (barrier(c)) return ncclInternalError;
    for (int r = 0; r < c->nranks; r++)
        if (hipMemcpy((unsigned char*)recvbuff + bytes * (size_t)r, c->slots + SLOT_BYTES * (size_t)r, bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return barrier(c) ? ncclInternalError : ncclSuccess;
}

ncclResult_t ncclBroadcast(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, int root, ncclComm_t comm, hipStream_t stream) {
    struct ncclComm* c = (struct ncclComm*)comm;
    const size_t bytes = count * type_size(datatype);
    if (!bytes || bytes > SLOT_BYTES || root < 0 || root >= c->nranks) return ncclInvalidArgument;
    if (agree(c, 0xB000000000ull ^ (unsigned long long)count ^ ((unsigned long long)datatype << 32) ^ ((unsigned long long)root << 36))) return ncclInternalError;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (c->rank == root && hipMemcpy(c->slots + SLOT_BYTES * (size_t)root, sendbuff, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    if (barrier(c)) return ncclInternalError;
    if (hipMemcpy(recvbuff, c->slots + SLOT_BYTES * (size_t)root, bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return barrier(c) ? ncclInternalError : ncclSuccess;
}

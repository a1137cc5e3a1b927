/*
 * rccl_stub.c — TEST DOUBLE for librccl (tests only; never shipped, never loaded unless DRONE_RCCL_LIB points at it).
 *
 * RCCL refuses two ranks on one GPU, and the pool has 1-GPU boxes only, so the rank > 0 side of drone_vec_gather
 * (slice offsets, ragged counts, the all-gather-v branch, host staging) could never run. This stub implements the
 * eight entry points the library dlsym()s with the semantics RCCL documents, over a POSIX shared-memory segment
 * between processes that may share a device: every op is stream-synchronous (hipStreamSynchronize, copy through the
 * segment, two barriers). It checks what RCCL would require: same count on every rank for all-gather, a root in
 * range, calls in the same order (op sequence number + kind + count compared across ranks).
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

typedef struct Header {
    volatile int arrived[2];
    volatile int sense;
    volatile int failed;
    volatile unsigned long long op_sig[MAX_RANKS]; /* what each rank thinks the current op is */
} Header;

struct ncclComm {
    int rank, nranks;
    char name[NCCL_UNIQUE_ID_BYTES];
    Header* h;
    unsigned char* slots;
    size_t map_bytes;
    int local_sense;
    unsigned long long seq;
};

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
    c->map_bytes = 4096 + SLOT_BYTES * (size_t)nranks;
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
    c->slots = (unsigned char*)p + 4096;
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

ncclResult_t ncclGroupStart(void) { return ncclSuccess; }
ncclResult_t ncclGroupEnd(void) { return ncclSuccess; }
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

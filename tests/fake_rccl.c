/* tests/fake_rccl.c -- a STAND-IN for librccl.so, test infrastructure only (never shipped, never loaded unless MHX_RCCL_LIBRARY names it).
 *
 * RCCL refuses two ranks on one device, so on a 1-GPU box nothing of libmhx's RCCL binding (datasketch_amd/csrc/comm.hip) beyond a
 * communicator of one rank could ever execute.  This library implements the entry points comm.hip binds -- ncclGetUniqueId,
 * ncclCommInitRank, ncclAllGather, ncclBroadcast, ncclSend / ncclRecv (inside a group), ncclGroupStart / ncclGroupEnd, ncclCommDestroy, ncclCommCount, ncclCommUserRank,
 * ncclCommCuDevice, ncclGetVersion, ncclGetErrorString -- with their documented semantics, for ranks that are processes of ONE
 * node sharing ONE device: data travels device -> a /dev/shm file per rank -> device, ranks meet at a sense-reversing barrier in a
 * shared page named after the unique id.  Collectives block (the stream is synchronised first), which the semantics allow.
 * What the tests exercise with it is OUR side: the argument marshalling of mhx_comm_*, the grouped per-root broadcasts of
 * mhx_comm_allgatherv_dev with their offsets, communicator lifetimes with world > 1.  It says nothing about RCCL or xGMI. */
#include <errno.h>
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat64 = 8 } ncclDataType_t;
typedef struct { char internal[128]; } ncclUniqueId;

typedef struct {
    volatile int arrived;
    volatile int sense;
    volatile int joined;
} Meet;

struct ncclComm {
    int rank, world, device;
    char tag[40];
    Meet *meet;
    int my_sense;
    int fd;         /* this rank's data file */
    size_t cap;     /* its size */
};
typedef struct ncclComm *ncclComm_t;

#define MAX_QUEUED 64
typedef struct { const void *send; void *recv; size_t bytes; int root; ncclComm_t comm; hipStream_t stream; } Bcast;
static __thread int g_group = 0;
static __thread int g_queued = 0;
static __thread Bcast g_queue[MAX_QUEUED];
/* point-to-point messages of the open group: sends are laid out in this rank's file behind a directory (count, then peer /
 * file offset / bytes per message); a receive takes the k-th directory entry its peer addressed to this rank */
#define MAX_P2P 4096
#define P2P_DIR_BYTES (16 + 24 * (size_t)MAX_P2P)
typedef struct { int is_send; const void *send; void *recv; size_t bytes; int peer; ncclComm_t comm; hipStream_t stream; } P2p;
static __thread int g_p2p_n = 0;
static __thread P2p g_p2p[MAX_P2P];

static size_t dtype_size(ncclDataType_t t) {
    switch (t) { case ncclInt8: case ncclUint8: return 1; case ncclFloat16: return 2; case ncclInt32: case ncclUint32: case ncclFloat32: return 4; default: return 8; }
}

static void barrier(ncclComm_t c) {
    Meet *m = c->meet;
    c->my_sense = !c->my_sense;
    if (__sync_add_and_fetch(&m->arrived, 1) == c->world) {
        m->arrived = 0;
        __sync_synchronize();
        m->sense = c->my_sense;
    } else {
        while (m->sense != c->my_sense) usleep(50);
    }
    __sync_synchronize();
}

static void data_path(const ncclComm_t c, int rank, char *out, size_t n) { snprintf(out, n, "/dev/shm/fake_rccl_%s_r%d", c->tag, rank); }

static ncclResult_t publish(ncclComm_t c, const void *d_src, size_t bytes) { /* device -> this rank's file */
    if (bytes > c->cap) {
        if (ftruncate(c->fd, (off_t)bytes) != 0) return ncclSystemError;
        c->cap = bytes;
    }
    if (bytes == 0) return ncclSuccess;
    void *p = mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, c->fd, 0);
    if (p == MAP_FAILED) return ncclSystemError;
    hipError_t e = hipMemcpy(p, d_src, bytes, hipMemcpyDeviceToHost);
    munmap(p, bytes);
    return e == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}

static ncclResult_t fetch(ncclComm_t c, int rank, void *d_dst, size_t bytes) { /* rank's file -> device */
    if (bytes == 0) return ncclSuccess;
    char path[128];
    data_path(c, rank, path, sizeof path);
    int fd = open(path, O_RDONLY);
    if (fd < 0) return ncclSystemError;
    void *p = mmap(NULL, bytes, PROT_READ, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return ncclSystemError;
    hipError_t e = hipMemcpy(d_dst, p, bytes, hipMemcpyHostToDevice);
    munmap(p, bytes);
    return e == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}

ncclResult_t ncclGetVersion(int *v) { *v = 22707; return ncclSuccess; }
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : r == ncclSystemError ? "fake rccl: system error" : r == ncclUnhandledCudaError ? "fake rccl: HIP error" : "fake rccl: invalid usage"; }

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    memset(id, 0, sizeof *id);
    int fd = open("/dev/urandom", O_RDONLY);
    if (fd < 0 || read(fd, id->internal, 16) != 16) return ncclSystemError;
    close(fd);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank) {
    if (!out || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    struct ncclComm *c = (struct ncclComm *)calloc(1, sizeof *c);
    c->rank = rank, c->world = nranks;
    if (hipGetDevice(&c->device) != hipSuccess) c->device = 0;
    for (int i = 0; i < 16; ++i) snprintf(c->tag + 2 * i, 3, "%02x", (unsigned char)id.internal[i]);
    char path[128];
    snprintf(path, sizeof path, "/dev/shm/fake_rccl_%s_meet", c->tag);
    int fd = open(path, O_RDWR | O_CREAT, 0600);
    if (fd < 0 || ftruncate(fd, 4096) != 0) return ncclSystemError;
    c->meet = (Meet *)mmap(NULL, 4096, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (c->meet == MAP_FAILED) return ncclSystemError;
    data_path(c, rank, path, sizeof path);
    c->fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0600);
    if (c->fd < 0) return ncclSystemError;
    __sync_add_and_fetch(&c->meet->joined, 1);
    for (int spins = 0; c->meet->joined < nranks; ++spins) { /* ncclCommInitRank is a collective: everybody is here on return */
        if (spins > 1200000) return ncclSystemError;         /* a minute */
        usleep(50);
    }
    *out = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return ncclSuccess;
    char path[128];
    data_path(c, c->rank, path, sizeof path);
    unlink(path);
    if (c->rank == 0) {
        snprintf(path, sizeof path, "/dev/shm/fake_rccl_%s_meet", c->tag);
        unlink(path);
    }
    close(c->fd);
    munmap((void *)c->meet, 4096);
    free(c);
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t c, int *n) { *n = c->world; return ncclSuccess; }
ncclResult_t ncclCommUserRank(const ncclComm_t c, int *r) { *r = c->rank; return ncclSuccess; }
ncclResult_t ncclCommCuDevice(const ncclComm_t c, int *d) { *d = c->device; return ncclSuccess; }

ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t t, ncclComm_t c, hipStream_t stream) {
    if (g_group) return ncclInvalidUsage; /* (comm.hip does not group its all-gather) */
    const size_t bytes = count * dtype_size(t);
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    ncclResult_t r = publish(c, send, bytes);
    barrier(c);
    for (int q = 0; q < c->world && r == ncclSuccess; ++q) r = fetch(c, q, (char *)recv + (size_t)q * bytes, bytes);
    barrier(c); /* nobody overwrites its file before everybody has read it */
    return r;
}

static ncclResult_t run_bcast(const Bcast *b) {
    ncclComm_t c = b->comm;
    ncclResult_t r = ncclSuccess;
    if (hipStreamSynchronize(b->stream) != hipSuccess) return ncclUnhandledCudaError;
    if (c->rank == b->root) r = publish(c, b->send, b->bytes);
    barrier(c);
    if (r == ncclSuccess && !(c->rank == b->root && b->send == b->recv)) r = fetch(c, b->root, b->recv, b->bytes);
    barrier(c);
    return r;
}

ncclResult_t ncclBroadcast(const void *send, void *recv, size_t count, ncclDataType_t t, int root, ncclComm_t c, hipStream_t stream) {
    if (root < 0 || root >= c->world) return ncclInvalidArgument;
    Bcast b = {send, recv, count * dtype_size(t), root, c, stream};
    if (!g_group) return run_bcast(&b);
    if (g_queued >= MAX_QUEUED) return ncclInvalidUsage;
    g_queue[g_queued++] = b;
    return ncclSuccess;
}

static ncclResult_t p2p_queue(int is_send, const void *send, void *recv, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t stream) {
    if (peer < 0 || peer >= c->world) return ncclInvalidArgument;
    if (!g_group) return ncclInvalidUsage; /* (comm.hip only ever sends and receives inside one group call) */
    if (g_p2p_n >= MAX_P2P) return ncclInvalidUsage;
    P2p m = {is_send, send, recv, count * dtype_size(t), peer, c, stream};
    g_p2p[g_p2p_n++] = m;
    return ncclSuccess;
}
ncclResult_t ncclSend(const void *send, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t stream) { return p2p_queue(1, send, NULL, count, t, peer, c, stream); }
ncclResult_t ncclRecv(void *recv, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t stream) { return p2p_queue(0, NULL, recv, count, t, peer, c, stream); }

static ncclResult_t run_p2p(void) {
    if (g_p2p_n == 0) return ncclSuccess;
    ncclComm_t c = g_p2p[0].comm;
    ncclResult_t r = ncclSuccess;
    if (hipStreamSynchronize(g_p2p[0].stream) != hipSuccess) return ncclUnhandledCudaError;
    size_t total = P2P_DIR_BYTES;
    int n_send = 0;
    for (int i = 0; i < g_p2p_n; ++i) if (g_p2p[i].is_send) total += g_p2p[i].bytes, ++n_send;
    if (total > c->cap) {
        if (ftruncate(c->fd, (off_t)total) != 0) r = ncclSystemError; else c->cap = total;
    }
    char *mine = r == ncclSuccess ? (char *)mmap(NULL, total, PROT_READ | PROT_WRITE, MAP_SHARED, c->fd, 0) : (char *)MAP_FAILED;
    if (mine == (char *)MAP_FAILED) r = ncclSystemError;
    if (r == ncclSuccess) {
        uint64_t *dir = (uint64_t *)mine;
        size_t at = P2P_DIR_BYTES;
        int k = 0;
        dir[0] = (uint64_t)n_send;
        for (int i = 0; i < g_p2p_n && r == ncclSuccess; ++i) {
            if (!g_p2p[i].is_send) continue;
            dir[2 + 3 * k] = (uint64_t)g_p2p[i].peer, dir[3 + 3 * k] = at, dir[4 + 3 * k] = g_p2p[i].bytes;
            if (g_p2p[i].bytes && hipMemcpy(mine + at, g_p2p[i].send, g_p2p[i].bytes, hipMemcpyDeviceToHost) != hipSuccess) r = ncclUnhandledCudaError;
            at += g_p2p[i].bytes, ++k;
        }
        munmap(mine, total);
    }
    barrier(c); /* every rank's sends are in its file */
    for (int i = 0; i < g_p2p_n && r == ncclSuccess; ++i) {
        if (g_p2p[i].is_send) continue;
        int nth = 0; /* how many earlier receives of this group came from the same peer */
        for (int j = 0; j < i; ++j) if (!g_p2p[j].is_send && g_p2p[j].peer == g_p2p[i].peer) ++nth;
        char path[128];
        data_path(c, g_p2p[i].peer, path, sizeof path);
        int fd = open(path, O_RDONLY);
        struct stat st;
        if (fd < 0 || fstat(fd, &st) != 0 || (size_t)st.st_size < P2P_DIR_BYTES) { if (fd >= 0) close(fd); r = ncclSystemError; break; }
        char *theirs = (char *)mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_SHARED, fd, 0);
        close(fd);
        if (theirs == (char *)MAP_FAILED) { r = ncclSystemError; break; }
        const uint64_t *dir = (const uint64_t *)theirs;
        int found = 0;
        for (uint64_t k = 0; k < dir[0]; ++k) {
            if (dir[2 + 3 * k] != (uint64_t)c->rank) continue;
            if (nth-- > 0) continue;
            found = 1;
            if (dir[4 + 3 * k] != g_p2p[i].bytes) r = ncclInvalidArgument; /* the two ends disagree about a message's size */
            else if (g_p2p[i].bytes && hipMemcpy(g_p2p[i].recv, theirs + dir[3 + 3 * k], g_p2p[i].bytes, hipMemcpyHostToDevice) != hipSuccess) r = ncclUnhandledCudaError;
            break;
        }
        if (!found && r == ncclSuccess) r = ncclInvalidUsage; /* a receive without a send: real RCCL would hang here */
        munmap(theirs, (size_t)st.st_size);
    }
    barrier(c); /* nobody overwrites its file before everybody has read it */
    g_p2p_n = 0;
    return r;
}

ncclResult_t ncclGroupStart(void) { ++g_group; return ncclSuccess; }
ncclResult_t ncclGroupEnd(void) {
    if (g_group <= 0) return ncclInvalidUsage;
    if (--g_group > 0) return ncclSuccess;
    ncclResult_t r = ncclSuccess;
    for (int i = 0; i < g_queued; ++i) { /* every rank queued the same calls in the same order */
        const ncclResult_t ri = run_bcast(&g_queue[i]);
        if (r == ncclSuccess) r = ri;
    }
    g_queued = 0;
    const ncclResult_t rp = run_p2p();
    return r == ncclSuccess ? rp : r;
}

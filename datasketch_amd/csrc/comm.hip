// comm.hip -- RCCL all-gather of signature-matrix shards over xGMI (one rank per context/GPU).
// RCCL is bound lazily with dlopen so libmhx.so loads on hosts that have no librccl.so.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <mutex>

#include "mhx_internal.h"

struct mhx_comm {
    mhx_ctx *ctx = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0;
    int world = 1;
};

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommCuDevice)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // MHX_RCCL_LIBRARY: another library with RCCL's entry points (tests: tests/fake_rccl.c, which lets ranks share a device)
        const char *override_path = getenv("MHX_RCCL_LIBRARY");
        const char *names[] = {override_path && *override_path ? override_path : "librccl.so", "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char *n : names) {
            r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (r.handle) break;
            if (override_path && *override_path) break;  // (an override that does not load is an error, not a reason to take the real one)
        }
        if (!r.handle) {
            r.error = std::string("cannot load librccl.so: ") + dlerror();
            return;
        }
#define MHX_SYM(field, sym)                                              \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, sym)); \
    if (!r.field) r.error = std::string("librccl.so lacks symbol ") + sym;
        MHX_SYM(GetUniqueId, "ncclGetUniqueId")
        MHX_SYM(CommInitRank, "ncclCommInitRank")
        MHX_SYM(CommDestroy, "ncclCommDestroy")
        MHX_SYM(AllGather, "ncclAllGather")
        MHX_SYM(Broadcast, "ncclBroadcast")
        MHX_SYM(GroupStart, "ncclGroupStart")
        MHX_SYM(GroupEnd, "ncclGroupEnd")
        MHX_SYM(GetErrorString, "ncclGetErrorString")
        MHX_SYM(CommCount, "ncclCommCount")
        MHX_SYM(CommUserRank, "ncclCommUserRank")
        MHX_SYM(CommCuDevice, "ncclCommCuDevice")
        MHX_SYM(GetVersion, "ncclGetVersion")
#undef MHX_SYM
    });
    return r;
}

int rccl_ready() {
    Rccl &r = rccl();
    if (!r.error.empty()) return mhx::fail(MHX_ERR_COMM, "%s", r.error.c_str());
    return MHX_OK;
}

#define MHX_RCCL_CHECK(expr)                                                                          \
    do {                                                                                              \
        ncclResult_t _r = (expr);                                                                     \
        if (_r != ncclSuccess)                                                                        \
            return mhx::fail(MHX_ERR_COMM, "%s failed: %s", #expr, rccl().GetErrorString(_r));        \
    } while (0)

}  // namespace

static_assert(sizeof(ncclUniqueId) == MHX_COMM_ID_BYTES, "RCCL unique id size changed");

extern "C" {

int mhx_comm_unique_id(uint8_t id[MHX_COMM_ID_BYTES]) {
    if (!id) return mhx::fail(MHX_ERR_INVALID, "id is NULL");
    if (int rc = rccl_ready()) return rc;
    ncclUniqueId uid;
    MHX_RCCL_CHECK(rccl().GetUniqueId(&uid));
    memcpy(id, &uid, MHX_COMM_ID_BYTES);
    return MHX_OK;
}

int mhx_comm_create(mhx_ctx *ctx, const uint8_t id[MHX_COMM_ID_BYTES], int rank, int world_size,
                    mhx_comm **out) {
    if (!ctx || !id || !out) return mhx::fail(MHX_ERR_INVALID, "NULL argument");
    MHX_REQUIRE(world_size > 0 && rank >= 0 && rank < world_size, "bad rank %d / world_size %d", rank, world_size);
    if (int rc = rccl_ready()) return rc;
    MHX_GUARD(ctx);
    if (int rc = ctx->activate()) return rc;
    ncclUniqueId uid;
    memcpy(&uid, id, MHX_COMM_ID_BYTES);
    mhx_comm *c = new mhx_comm();
    c->ctx = ctx;
    c->rank = rank;
    c->world = world_size;
    ncclResult_t r = rccl().CommInitRank(&c->comm, world_size, uid, rank);
    if (r != ncclSuccess) {
        delete c;
        return mhx::fail(MHX_ERR_COMM, "ncclCommInitRank failed: %s", rccl().GetErrorString(r));
    }
    *out = c;
    return MHX_OK;
}

int mhx_comm_info(mhx_comm *comm, int *rank, int *world_size, int *device, int *rccl_version) {
    if (!comm) return mhx::fail(MHX_ERR_INVALID, "comm is NULL");
    // asked of RCCL itself, not echoed from mhx_comm_create's arguments: "ranks seen" is what the
    // communicator was really built with
    int v = 0;
    if (world_size) {
        MHX_RCCL_CHECK(rccl().CommCount(comm->comm, &v));
        *world_size = v;
    }
    if (rank) {
        MHX_RCCL_CHECK(rccl().CommUserRank(comm->comm, &v));
        *rank = v;
    }
    if (device) {
        MHX_RCCL_CHECK(rccl().CommCuDevice(comm->comm, &v));
        *device = v;
    }
    if (rccl_version) {
        MHX_RCCL_CHECK(rccl().GetVersion(&v));
        *rccl_version = v;
    }
    return MHX_OK;
}

int mhx_comm_destroy(mhx_comm *comm) {
    if (!comm) return MHX_OK;
    (void)hipSetDevice(comm->ctx->device);
    (void)hipStreamSynchronize(comm->ctx->stream);
    if (comm->comm) (void)rccl().CommDestroy(comm->comm);
    delete comm;
    return MHX_OK;
}

int mhx_comm_allgather_dev(mhx_comm *comm, const void *d_send, void *d_recv, size_t bytes_per_rank) {
    if (!comm) return mhx::fail(MHX_ERR_INVALID, "comm is NULL");
    if (bytes_per_rank == 0) return MHX_OK;
    MHX_REQUIRE(d_send && d_recv, "NULL device pointer");
    MHX_GUARD(comm->ctx);
    if (int rc = comm->ctx->activate()) return rc;
    MHX_RCCL_CHECK(rccl().AllGather(d_send, d_recv, bytes_per_rank, ncclUint8, comm->comm, comm->ctx->stream));
    return MHX_OK;
}

// Unequal shards, written in place: rank q's recv_bytes[q] bytes land at d_recv + recv_offsets[q] on every rank.  RCCL has
// no all-gather-v; the documented equivalent is one ncclBroadcast per root inside a group call, which RCCL fuses into one
// launch -- no padded staging buffer, no squeeze copies, no host synchronisation.
int mhx_comm_allgatherv_dev(mhx_comm *comm, const void *d_send, void *d_recv, const uint64_t *recv_offsets,
                            const uint64_t *recv_bytes) {
    if (!comm) return mhx::fail(MHX_ERR_INVALID, "comm is NULL");
    MHX_REQUIRE(recv_offsets && recv_bytes, "NULL offsets / sizes");
    uint64_t total = 0;
    for (int q = 0; q < comm->world; ++q) total += recv_bytes[q];
    if (total == 0) return MHX_OK;
    MHX_REQUIRE(d_recv && (d_send || recv_bytes[comm->rank] == 0), "NULL device pointer");
    MHX_GUARD(comm->ctx);
    if (int rc = comm->ctx->activate()) return rc;
    MHX_RCCL_CHECK(rccl().GroupStart());
    ncclResult_t first = ncclSuccess;
    for (int q = 0; q < comm->world; ++q) {
        if (recv_bytes[q] == 0) continue;  // (the same on every rank: counts are common knowledge)
        char *dst = static_cast<char *>(d_recv) + recv_offsets[q];
        const ncclResult_t r = rccl().Broadcast(q == comm->rank ? d_send : dst, dst, recv_bytes[q], ncclUint8, q, comm->comm,
                                                comm->ctx->stream);
        if (r != ncclSuccess && first == ncclSuccess) first = r;
    }
    const ncclResult_t end = rccl().GroupEnd();  // always closed: an open group would swallow the next call
    if (first != ncclSuccess) return mhx::fail(MHX_ERR_COMM, "ncclBroadcast failed: %s", rccl().GetErrorString(first));
    if (end != ncclSuccess) return mhx::fail(MHX_ERR_COMM, "ncclGroupEnd failed: %s", rccl().GetErrorString(end));
    return MHX_OK;
}

}  // extern "C"

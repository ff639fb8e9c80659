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
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
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
        MHX_SYM(Send, "ncclSend")
        MHX_SYM(Recv, "ncclRecv")
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

// A grouped point-to-point exchange -- the by-band exchange of band digests (dist.py: exchange_band_digests_dev): rank p keeps its
// rows' digests band-major [bands, n_p]; rank q, which builds the hashtables of bands [lo_q, hi_q) (ref: datasketch/lsh.py:199 -- one
// independent table per band), receives from every p the runs of those bands and places them at [band - lo_q][begin_p ...) of its
// [hi_q - lo_q, N] matrix.  One ncclSend / ncclRecv per run inside ONE group call (RCCL fuses them into one launch and pairs the k-th
// send of p to q with the k-th receive of q from p); a run a rank owes itself is a device copy on the same stream; runs of zero bytes
// are dropped (sizes are common knowledge, so both ends drop the same ones).
int mhx_comm_exchange_dev(mhx_comm *comm, const void *d_send, void *d_recv, int32_t n_send, const int32_t *send_peers,
                          const uint64_t *send_offsets, const uint64_t *send_bytes, int32_t n_recv, const int32_t *recv_peers,
                          const uint64_t *recv_offsets, const uint64_t *recv_bytes) {
    if (!comm) return mhx::fail(MHX_ERR_INVALID, "comm is NULL");
    MHX_REQUIRE(n_send >= 0 && n_recv >= 0, "negative message count");
    MHX_REQUIRE(n_send == 0 || (send_peers && send_offsets && send_bytes), "NULL send list");
    MHX_REQUIRE(n_recv == 0 || (recv_peers && recv_offsets && recv_bytes), "NULL receive list");
    uint64_t out_bytes = 0, in_bytes = 0;
    for (int i = 0; i < n_send; ++i) {
        MHX_REQUIRE(send_peers[i] >= 0 && send_peers[i] < comm->world, "send %d: peer %d outside the communicator of %d ranks", i, send_peers[i], comm->world);
        out_bytes += send_bytes[i];
    }
    for (int i = 0; i < n_recv; ++i) {
        MHX_REQUIRE(recv_peers[i] >= 0 && recv_peers[i] < comm->world, "receive %d: peer %d outside the communicator of %d ranks", i, recv_peers[i], comm->world);
        in_bytes += recv_bytes[i];
    }
    MHX_REQUIRE((d_send || out_bytes == 0) && (d_recv || in_bytes == 0), "NULL device pointer");
    // what this rank owes itself: the k-th such send goes to the k-th such receive, sizes must agree
    {
        int r = 0;
        for (int i = 0; i < n_send; ++i) {
            if (send_peers[i] != comm->rank || send_bytes[i] == 0) continue;
            while (r < n_recv && (recv_peers[r] != comm->rank || recv_bytes[r] == 0)) ++r;
            MHX_REQUIRE(r < n_recv && recv_bytes[r] == send_bytes[i], "send %d to this rank itself has no receive of the same size", i);
            ++r;
        }
        for (; r < n_recv; ++r) MHX_REQUIRE(recv_peers[r] != comm->rank || recv_bytes[r] == 0, "receive %d from this rank itself has no send", r);
    }
    if (out_bytes == 0 && in_bytes == 0) return MHX_OK;
    MHX_GUARD(comm->ctx);
    if (int rc = comm->ctx->activate()) return rc;
    const char *src = static_cast<const char *>(d_send);
    char *dst = static_cast<char *>(d_recv);
    for (int i = 0, r = 0; i < n_send; ++i) {
        if (send_peers[i] != comm->rank || send_bytes[i] == 0) continue;
        while (recv_peers[r] != comm->rank || recv_bytes[r] == 0) ++r;
        MHX_HIP_CHECK(hipMemcpyAsync(dst + recv_offsets[r], src + send_offsets[i], send_bytes[i], hipMemcpyDeviceToDevice, comm->ctx->stream));
        ++r;
    }
    MHX_RCCL_CHECK(rccl().GroupStart());
    ncclResult_t first = ncclSuccess;
    for (int i = 0; i < n_send; ++i) {
        if (send_peers[i] == comm->rank || send_bytes[i] == 0) continue;
        const ncclResult_t r = rccl().Send(src + send_offsets[i], send_bytes[i], ncclUint8, send_peers[i], comm->comm, comm->ctx->stream);
        if (r != ncclSuccess && first == ncclSuccess) first = r;
    }
    for (int i = 0; i < n_recv; ++i) {
        if (recv_peers[i] == comm->rank || recv_bytes[i] == 0) continue;
        const ncclResult_t r = rccl().Recv(dst + recv_offsets[i], recv_bytes[i], ncclUint8, recv_peers[i], comm->comm, comm->ctx->stream);
        if (r != ncclSuccess && first == ncclSuccess) first = r;
    }
    const ncclResult_t end = rccl().GroupEnd();  // always closed: an open group would swallow the next call
    if (first != ncclSuccess) return mhx::fail(MHX_ERR_COMM, "ncclSend / ncclRecv failed: %s", rccl().GetErrorString(first));
    if (end != ncclSuccess) return mhx::fail(MHX_ERR_COMM, "ncclGroupEnd failed: %s", rccl().GetErrorString(end));
    return MHX_OK;
}

}  // extern "C"

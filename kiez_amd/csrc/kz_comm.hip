// kz_comm_*: the collectives of the sharded path behind the C ABI (include/kiez_amd.h "multi-GPU"), over RCCL.
//
// One process per GPU; the HOST distributes a 128-byte unique id (rank 0: kz_comm_unique_id) by whatever it has -- MPI, a file, a
// socket, torch.distributed's store -- and every rank calls kz_comm_create(ctx, id, rank, world).  The collectives run on the
// context's stream, ordered with the kernels of the same context: what the reference's only multi-device call hands to Faiss
// (kiez/neighbors/approximate/faiss.py:138, index_cpu_to_all_gpus) is here four calls a non-torch host can bind:
//     broadcast   (the replicated target, from the rank that holds it)
//     all_to_all  (the per-shard reverse lists of the shared sweep: block r of every rank's buffer goes to rank r)
//     all_gather  (the per-target-row fit state; the source shards where a route needs them)
//     all_reduce_min_f64 (DisSimLocal's one scalar)
// librccl is NOT a link-time dependency of libkiez_amd.so: it is dlopen'ed by the first kz_comm_* call (an already loaded copy --
// torch ships one -- is taken first, so that a process never holds two), and a host without RCCL gets KZ_ERR_UNSUPPORTED from
// these calls and everything else as before.
#include <dlfcn.h>

#include <cstring>

#include "../../include/kiez_amd.h"
#include "kz_common.h"

namespace {

// (the few RCCL declarations these calls need, as rccl.h has them: the header is not required to build)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclFloat32 = 7,
               ncclFloat64 = 8 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;

struct KzRccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
KzRccl g_rccl;

int kz_rccl_load() {
    if (g_rccl.handle) return KZ_OK;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names)   // a copy the process already holds (torch's) first: never two RCCLs in one process
        if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL))) break;
    if (!h)
        for (const char* n : names)
            if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) {
        kz_set_error("kz_comm: librccl.so could not be loaded (%s): the sharded path needs RCCL", dlerror());
        return KZ_ERR_UNSUPPORTED;
    }
    KzRccl r;
    r.handle = h;
#define KZ_SYM(field, name)                                                         \
    do {                                                                            \
        *(void**)(&r.field) = dlsym(h, name);                                       \
        if (!r.field) {                                                             \
            kz_set_error("kz_comm: librccl.so does not export %s", name);           \
            return KZ_ERR_UNSUPPORTED;                                              \
        }                                                                           \
    } while (0)
    KZ_SYM(GetUniqueId, "ncclGetUniqueId");
    KZ_SYM(CommInitRank, "ncclCommInitRank");
    KZ_SYM(CommDestroy, "ncclCommDestroy");
    KZ_SYM(Broadcast, "ncclBroadcast");
    KZ_SYM(AllGather, "ncclAllGather");
    KZ_SYM(AllReduce, "ncclAllReduce");
    KZ_SYM(Send, "ncclSend");
    KZ_SYM(Recv, "ncclRecv");
    KZ_SYM(GroupStart, "ncclGroupStart");
    KZ_SYM(GroupEnd, "ncclGroupEnd");
    KZ_SYM(GetErrorString, "ncclGetErrorString");
#undef KZ_SYM
    g_rccl = r;
    return KZ_OK;
}

}  // namespace

struct kz_comm {
    kz_ctx* ctx;
    ncclComm_t comm;
    int rank, world;
};

#define KZ_NCCL(call)                                                                                        \
    do {                                                                                                     \
        const ncclResult_t r_ = (call);                                                                      \
        if (r_ != ncclSuccess) {                                                                             \
            kz_set_error("kz_comm: %s failed: %s", #call, g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?"); \
            return KZ_ERR_HIP;                                                                               \
        }                                                                                                    \
    } while (0)

extern "C" {

int kz_comm_unique_id(void* id128) {
    KZ_REQUIRE(id128, "kz_comm_unique_id: null argument");
    const int rc = kz_rccl_load();
    if (rc != KZ_OK) return rc;
    ncclUniqueId id;
    KZ_NCCL(g_rccl.GetUniqueId(&id));
    static_assert(sizeof(id) == KZ_COMM_ID_BYTES, "the unique id is 128 bytes");
    memcpy(id128, &id, sizeof(id));
    return KZ_OK;
}

int kz_comm_create(kz_ctx* ctx, const void* id128, int rank, int world, kz_comm** out) {
    KZ_REQUIRE(ctx && id128 && out, "kz_comm_create: null argument");
    KZ_REQUIRE(world >= 1 && rank >= 0 && rank < world, "kz_comm_create: rank %d of %d", rank, world);
    const int rc = kz_rccl_load();
    if (rc != KZ_OK) return rc;
    KZ_HIP(hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    KZ_NCCL(g_rccl.CommInitRank(&c, world, id, rank));
    kz_comm* k = new kz_comm();
    k->ctx = ctx;
    k->comm = c;
    k->rank = rank;
    k->world = world;
    *out = k;
    return KZ_OK;
}

int kz_comm_destroy(kz_comm* c) {
    if (!c) return KZ_OK;
    if (c->ctx) (void)hipStreamSynchronize(c->ctx->stream);
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    delete c;
    return KZ_OK;
}

int kz_comm_rank(const kz_comm* c, int* rank, int* world) {
    KZ_REQUIRE(c, "kz_comm_rank: null communicator");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    return KZ_OK;
}

// bytes of d_buf from rank `root` to every rank, in place
int kz_comm_broadcast(kz_comm* c, void* d_buf, size_t bytes, int root) {
    KZ_REQUIRE(c && (d_buf || bytes == 0), "kz_comm_broadcast: null argument");
    KZ_REQUIRE(root >= 0 && root < c->world, "kz_comm_broadcast: root %d of %d ranks", root, c->world);
    if (bytes == 0) return KZ_OK;
    KZ_HIP(hipSetDevice(c->ctx->device));
    KZ_NCCL(g_rccl.Broadcast(d_buf, d_buf, bytes, ncclUint8, root, c->comm, c->ctx->stream));
    return KZ_OK;
}

// d_recv [world][bytes_per_rank] = every rank's d_send [bytes_per_rank], in rank order
int kz_comm_all_gather(kz_comm* c, const void* d_send, void* d_recv, size_t bytes_per_rank) {
    KZ_REQUIRE(c && ((d_send && d_recv) || bytes_per_rank == 0), "kz_comm_all_gather: null argument");
    if (bytes_per_rank == 0) return KZ_OK;
    KZ_HIP(hipSetDevice(c->ctx->device));
    KZ_NCCL(g_rccl.AllGather(d_send, d_recv, bytes_per_rank, ncclUint8, c->comm, c->ctx->stream));
    return KZ_OK;
}

// block r ([send_bytes[r]] at send_offset[r]) of this rank's d_send goes to rank r; d_recv takes world blocks of recv_bytes each, in
// rank order (the exchange of the shared sweep: every rank sends rank r the reverse lists of r's slice of target rows -- slices differ
// by at most one row, the blocks a rank RECEIVES are all of its own slice's size)
int kz_comm_all_to_all(kz_comm* c, const void* d_send, const size_t* send_offset, const size_t* send_bytes, void* d_recv, size_t recv_bytes) {
    KZ_REQUIRE(c && d_send && d_recv && send_offset && send_bytes, "kz_comm_all_to_all: null argument");
    KZ_HIP(hipSetDevice(c->ctx->device));
    KZ_NCCL(g_rccl.GroupStart());
    ncclResult_t first = ncclSuccess;
    for (int r = 0; r < c->world; ++r) {
        ncclResult_t e = ncclSuccess;
        if (send_bytes[r]) e = g_rccl.Send((const char*)d_send + send_offset[r], send_bytes[r], ncclUint8, r, c->comm, c->ctx->stream);
        if (e == ncclSuccess && recv_bytes) e = g_rccl.Recv((char*)d_recv + (size_t)r * recv_bytes, recv_bytes, ncclUint8, r, c->comm, c->ctx->stream);
        if (e != ncclSuccess && first == ncclSuccess) first = e;
    }
    const ncclResult_t ge = g_rccl.GroupEnd();   // (always closed: an open group would swallow every later call)
    KZ_NCCL(first);
    KZ_NCCL(ge);
    return KZ_OK;
}

// element-wise minimum over the ranks of d_buf [count] float64, in place
int kz_comm_all_reduce_min_f64(kz_comm* c, double* d_buf, size_t count) {
    KZ_REQUIRE(c && (d_buf || count == 0), "kz_comm_all_reduce_min_f64: null argument");
    if (count == 0) return KZ_OK;
    KZ_HIP(hipSetDevice(c->ctx->device));
    KZ_NCCL(g_rccl.AllReduce(d_buf, d_buf, count, ncclFloat64, ncclMin, c->comm, c->ctx->stream));
    return KZ_OK;
}

}  // extern "C"
